/*
 * rt_oracle.cpp -- CPU parity oracle: a C++17 restatement of the reference algorithm.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  PARITY UNPINNED (see rt_oracle.h).
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -fopenmp -shared -fPIC  (oracle/Makefile)
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * Phlipsje/Raytracing @ v2 tree).  Nothing here is copied: the reference is C# and GLSL.
 */
#include "rt_oracle.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------
// vector helpers (canonical arithmetic, rt_oracle.h)
// ------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) {
    float s = 1.0f / sqrtf(dot(a, a));
    return v3(a.x * s, a.y * s, a.z * s);
}
inline float max0(float x) { return (0.0f < x) ? x : 0.0f; }
inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

// ------------------------------------------------------------------------------------------
// buffer records (PrimitiveStructsForBuffer.cs:8-74 <-> raytracer_fs.glsl:17-42, std430)
// ------------------------------------------------------------------------------------------
struct SphereRec {
    float center[3];
    float radius;
    float diffuse[3];
    uint32_t pure;
    float specular[3];
    float specularity;
};
struct PlaneRec {
    float position[3];
    float pad0;
    float normal[3];
    float pad1;
    float diffuse[3];
    uint32_t pure;
    float specular[3];
    float specularity;
};
struct TriRec {
    float a[3];
    float pad0;
    float b[3];
    float pad1;
    float c[3];
    float pad2;
    float n[3];
    float pad3;
    float diffuse[3];
    uint32_t pure;
    float specular[3];
    float specularity;
};
static_assert(sizeof(SphereRec) == 48, "SphereStruct is 48 B");
static_assert(sizeof(PlaneRec) == 64, "PlaneStruct is 64 B");
static_assert(sizeof(TriRec) == 96, "TriangleStruct is 96 B");

// ------------------------------------------------------------------------------------------
// intersection functions
// ------------------------------------------------------------------------------------------
struct V4 {
    float x, y, z, w;
};

// raytracer_fs.glsl:93-113
inline V4 IntersectSphere(V3 o, V3 d, V3 center, float radius) {
    V3 v = o - center;
    float b = 2 * (d.x * v.x + d.y * v.y + d.z * v.z);
    float c = v.x * v.x + v.y * v.y + v.z * v.z - radius * radius;
    float disc = b * b - 4 * c;
    if (disc < 0) return V4{-1, -1, -1, -1};
    float rootd = sqrtf(disc);
    float t1 = (-b + rootd) / 2;
    float t2 = (-b - rootd) / 2;
    if (t1 < 0) return V4{-1, -1, -1, -1};
    if (t2 < 0) {
        V3 n = normalize((o + t1 * d) - center);
        return V4{n.x, n.y, n.z, t1};
    }
    V3 n = normalize((o + t2 * d) - center);
    return V4{n.x, n.y, n.z, t2};
}

// raytracer_fs.glsl:114-120
inline float IntersectPlane(V3 o, V3 d, V3 center, V3 normal) {
    float den = dot(d, normal);
    if (den == 0) return -1;
    return dot(center - o, normal) / den;
}

// raytracer_fs.glsl:121-131
inline float IntersectTriangle(V3 o, V3 d, V3 A, V3 B, V3 C, V3 N) {
    float t = IntersectPlane(o, d, A, N);
    if (t <= 0) return -1;
    V3 X = o + t * d;
    bool outside = dot(cross(B - A, X - A), N) < 0 || dot(cross(C - B, X - B), N) < 0 ||
                   dot(cross(A - C, X - C), N) < 0;
    if (outside) return -1;
    return t;
}

// raytracer_fs.glsl:134-161 (no t>=0 and no t<best check; NaN comparisons fall through to hit)
inline bool IntersectBoundingBox(V3 o, V3 d, const float* b) {
    float ix = 1 / d.x, iy = 1 / d.y, iz = 1 / d.z;
    bool sx = ix < 0, sy = iy < 0, sz = iz < 0;
    // bounds[0] = min (b[0..2]), bounds[1] = max (b[3..5]); raySign picks bounds[1] for the near side
    float txmin = ((sx ? b[3] : b[0]) - o.x) * ix;
    float txmax = ((sx ? b[0] : b[3]) - o.x) * ix;
    float tymin = ((sy ? b[4] : b[1]) - o.y) * iy;
    float tymax = ((sy ? b[1] : b[4]) - o.y) * iy;
    if ((txmin > tymax) || (tymin > txmax)) return false;
    if (tymin > txmin) txmin = tymin;
    if (tymax < txmax) txmax = tymax;
    float tzmin = ((sz ? b[5] : b[2]) - o.z) * iz;
    float tzmax = ((sz ? b[2] : b[5]) - o.z) * iz;
    if ((txmin > tzmax) || (tzmin > txmax)) return false;
    return true;
}

// ------------------------------------------------------------------------------------------
// the per-pixel path (raytracer_fs.glsl:163-425)
// ------------------------------------------------------------------------------------------
struct SceneView {
    const SphereRec* spheres;
    int ns;
    const PlaneRec* planes;
    int np;
    const TriRec* tris;
    int nt;
    const float* acc;
    int64_t nacc;
    const float* lights;
    int lights_len;
};

struct Worker {
    const SceneView* s;
    uint64_t c[ORC_N_COUNTERS];
    std::vector<int64_t> stack;
    std::vector<int> sphC, triC;
    bool malformed = false;

    explicit Worker(const SceneView* sv) : s(sv) { std::memset(c, 0, sizeof c); }

    // raytracer_fs.glsl:163-199.  Counters start at 0 (the shader leaves them uninitialised).
    void GetRelevantPrimitives(V3 o, V3 d) {
        sphC.clear();
        triC.clear();
        stack.clear();
        if (s->nacc <= 0) return;  // empty scene serialises to an empty array (RTree.cs:114-115)
        stack.push_back(0);
        while (!stack.empty()) {
            if (stack.size() > c[ORC_C_MAX_STACK]) c[ORC_C_MAX_STACK] = stack.size();
            int64_t pos = stack.back();
            stack.pop_back();
            if (pos < 0 || pos + 8 > s->nacc) {
                malformed = true;
                return;
            }
            c[ORC_C_NODE_TESTS]++;
            if (!IntersectBoundingBox(o, d, s->acc + pos)) continue;
            int count = (int)s->acc[pos + 7];
            if (count < 0 || pos + 8 + count > s->nacc) {
                malformed = true;
                return;
            }
            if (s->acc[pos + 6] == 0) {
                for (int i = 0; i < count; i++) {
                    c[ORC_C_REFS_READ]++;
                    stack.push_back(pos + (int)s->acc[pos + 8 + i]);
                }
            } else {
                for (int i = 0; i < count; i++) {
                    c[ORC_C_REFS_READ]++;
                    int value = (int)s->acc[pos + 8 + i];
                    if (value >= s->ns) {
                        int ti = value - s->ns;
                        if (ti >= s->nt) {
                            malformed = true;
                            return;
                        }
                        triC.push_back(ti);
                    } else {
                        if (value < 0) {
                            malformed = true;
                            return;
                        }
                        sphC.push_back(value);
                    }
                }
            }
        }
        if (sphC.size() > c[ORC_C_MAX_SPH_CAND]) c[ORC_C_MAX_SPH_CAND] = sphC.size();
        if (triC.size() > c[ORC_C_MAX_TRI_CAND]) c[ORC_C_MAX_TRI_CAND] = triC.size();
    }

    // raytracer_fs.glsl:200-240
    bool ObjectInWayOfLight(float distanceToLight, V3 o, V3 d) {
        const float epsilon = 0.001f;
        c[ORC_C_SHADOW]++;
        GetRelevantPrimitives(o, d);
        for (int idx : sphC) {
            const SphereRec& sp = s->spheres[idx];
            c[ORC_C_SPHERE_TESTS]++;
            float r = IntersectSphere(o, d, ld3(sp.center), sp.radius).w;
            if (r > epsilon && r < distanceToLight) return true;
        }
        for (int i = 0; i < s->np; i++) {
            const PlaneRec& pl = s->planes[i];
            c[ORC_C_PLANE_TESTS]++;
            float r = IntersectPlane(o, d, ld3(pl.position), ld3(pl.normal));
            if (r > epsilon && r < distanceToLight) return true;
        }
        for (int idx : triC) {
            const TriRec& tr = s->tris[idx];
            c[ORC_C_TRI_TESTS]++;
            float r = IntersectTriangle(o, d, ld3(tr.a), ld3(tr.b), ld3(tr.c), ld3(tr.n));
            if (r > epsilon && r < distanceToLight) return true;
        }
        return false;
    }

    struct Hit {
        float t;
        V3 color;
        bool pure;
        V3 spec;
        float specularity;
        V3 normal;
        int id;
    };
    static Hit FreshHit() {
        // raytracer_fs.glsl:315-323, 371-376
        return Hit{3.402823466e+38f, v3(0, 0, 0), false, v3(0, 0, 0), 1.0f, v3(0, 0, 0), -1};
    }

    // raytracer_fs.glsl:241-297.  id: sphere i -> i, triangle j -> S+j, plane k -> S+T+k.
    void FindClosestIntersection(V3 o, V3 d, float minDistance, Hit& h) {
        GetRelevantPrimitives(o, d);
        for (int idx : sphC) {
            const SphereRec& sp = s->spheres[idx];
            c[ORC_C_SPHERE_TESTS]++;
            V4 r = IntersectSphere(o, d, ld3(sp.center), sp.radius);
            if (r.w > minDistance && r.w < h.t) {
                c[ORC_C_MATERIAL]++;
                h.t = r.w;
                h.color = ld3(sp.diffuse);
                h.pure = sp.pure != 0;
                h.spec = ld3(sp.specular);
                h.specularity = sp.specularity;
                h.normal = v3(r.x, r.y, r.z);
                h.id = idx;
            }
        }
        for (int i = 0; i < s->np; i++) {
            const PlaneRec& pl = s->planes[i];
            c[ORC_C_PLANE_TESTS]++;
            float r = IntersectPlane(o, d, ld3(pl.position), ld3(pl.normal));
            if (r > minDistance && r < h.t) {
                c[ORC_C_MATERIAL]++;
                h.t = r;
                h.color = ld3(pl.diffuse);
                h.pure = pl.pure != 0;
                h.spec = ld3(pl.specular);
                h.specularity = pl.specularity;
                h.normal = ld3(pl.normal);
                h.id = s->ns + s->nt + i;
            }
        }
        for (int idx : triC) {
            const TriRec& tr = s->tris[idx];
            c[ORC_C_TRI_TESTS]++;
            float r = IntersectTriangle(o, d, ld3(tr.a), ld3(tr.b), ld3(tr.c), ld3(tr.n));
            if (r > minDistance && r < h.t) {
                c[ORC_C_MATERIAL]++;
                h.t = r;
                h.color = ld3(tr.diffuse);
                h.pure = tr.pure != 0;
                h.spec = ld3(tr.specular);
                h.specularity = tr.specularity;
                h.normal = ld3(tr.n);
                h.id = s->ns + idx;
            }
        }
    }

    struct Pixel {
        V3 rgb;
        int primary_id, final_id, bounces;
    };

    // Bookkeeping only (ORC_C_ZERO_TERM), no effect on any pixel: would this light's term below add exactly
    // nothing whichever way its shadow ray ends?  Same expressions as the shading code of Shade(); true when every
    // component is a zero and the diffuse colour has no sign bit (x + 0 == x except for x = -0).
    static bool sign(float v) {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        return (u >> 31) != 0;
    }
    static bool TermIsZero(const Hit& h, bool mirror, V3 rayDirection, V3 L, float distanceToLight, V3 lightColor) {
        V3 term;
        if (mirror) {
            term = 1.0f / (distanceToLight * distanceToLight) * lightColor * h.color * max0(dot(h.normal, L));
        } else {
            V3 n = h.normal;
            if (dot(n, rayDirection) > 0.0f) n = -n;
            V3 vectorR = normalize(L - 2 * dot(L, n) * n);
            float a = max0(dot(n, L)), dr = max0(dot(rayDirection, vectorR));
            if (!(dr == 0.0f && h.specularity > 0.0f)) return false;  // powf(0, s) == 0 needs s > 0
            term = 1.0f / (distanceToLight * distanceToLight) * lightColor * (h.color * a + h.spec * 0.0f);
        }
        return term.x == 0.0f && term.y == 0.0f && term.z == 0.0f && !sign(h.color.x) && !sign(h.color.y) &&
               !sign(h.color.z);
    }

    // raytracer_fs.glsl:300-425.  px,py are gl_FragCoord.xy (pixel centre, origin bottom-left).
    Pixel Shade(const float* camera, float px, float py, int maxBounces) {
        const float epsilon = 0.001f;
        const V3 ambiantLight = v3(0.1f, 0.1f, 0.1f);
        const V3 skyColor = v3(0.5f, 0.6f, 1.0f);
        const float FLT_BIG = 3.402823466e+38f;

        float x = px, y = py;
        float width = camera[12], height = camera[13];
        V3 bottomLeft = ld3(camera + 3), bottomRight = ld3(camera + 6), topLeft = ld3(camera + 9);
        V3 rayOrigin = ld3(camera);
        V3 rayDirection = normalize(
            (bottomLeft + (x / width) * (bottomRight - bottomLeft) + (y / height) * (topLeft - bottomLeft)) -
            rayOrigin);
        Hit h = FreshHit();
        c[ORC_C_PRIMARY]++;
        FindClosestIntersection(rayOrigin, rayDirection, 0, h);
        Pixel out;
        out.primary_id = h.id;

        V3 mirrorColorMultiplier = v3(1, 1, 1);
        V3 finalColor = v3(0, 0, 0);
        int bounces = 0;
        while (h.pure && bounces < maxBounces) {
            c[ORC_C_SECONDARY]++;
            V3 hitPos = rayOrigin + h.t * rayDirection;
            if (h.color.x != 0 || h.color.y != 0 || h.color.z != 0) {
                V3 combinedColor = h.color * ambiantLight;
                for (int l = 0; l < s->lights_len; l += 6) {
                    V3 lightPos = ld3(s->lights + l);
                    float distanceToLight = length(lightPos - hitPos);
                    V3 shadowRayDirection = normalize(lightPos - hitPos);
                    if (TermIsZero(h, true, rayDirection, shadowRayDirection, distanceToLight, ld3(s->lights + l + 3)))
                        c[ORC_C_ZERO_TERM]++;
                    if (!ObjectInWayOfLight(distanceToLight, hitPos, shadowRayDirection)) {
                        V3 lightColor = ld3(s->lights + l + 3);
                        combinedColor = combinedColor + 1.0f / (distanceToLight * distanceToLight) * lightColor *
                                                            h.color * max0(dot(h.normal, shadowRayDirection));
                    }
                }
                finalColor = finalColor + combinedColor * mirrorColorMultiplier;
            }
            mirrorColorMultiplier = mirrorColorMultiplier * h.spec;
            V3 n = h.normal;
            if (dot(n, rayDirection) > 0.0f) n = -n;
            V3 reflectionDirection = normalize(rayDirection - 2 * dot(rayDirection, n) * n);
            rayOrigin = hitPos;
            rayDirection = reflectionDirection;
            h = FreshHit();
            FindClosestIntersection(rayOrigin, rayDirection, epsilon, h);
            bounces++;
        }
        out.bounces = bounces;
        out.final_id = h.id;
        if (h.t == FLT_BIG) {
            out.rgb = finalColor + skyColor * mirrorColorMultiplier;
            return out;
        }
        // raytracer_fs.glsl:390-394 writes outputColor but does not return: the hit is shaded anyway.
        V3 hitPos = rayOrigin + h.t * rayDirection;
        V3 combinedColor = h.color * ambiantLight;
        V3 n = h.normal;
        for (int l = 0; l < s->lights_len; l += 6) {
            V3 lightPos = ld3(s->lights + l);
            float distanceToLight = length(lightPos - hitPos);
            V3 shadowRayDirection = normalize(lightPos - hitPos);
            if (TermIsZero(h, false, rayDirection, shadowRayDirection, distanceToLight, ld3(s->lights + l + 3)))
                c[ORC_C_ZERO_TERM]++;
            if (!ObjectInWayOfLight(distanceToLight, hitPos, shadowRayDirection)) {
                V3 lightColor = ld3(s->lights + l + 3);
                if (dot(n, rayDirection) > 0.0f) n = -n;
                V3 vectorR = normalize(shadowRayDirection - 2 * dot(shadowRayDirection, n) * n);
                combinedColor =
                    combinedColor + 1.0f / (distanceToLight * distanceToLight) * lightColor *
                                        (h.color * max0(dot(n, shadowRayDirection)) +
                                         h.spec * powf(max0(dot(rayDirection, vectorR)), h.specularity));
            }
        }
        finalColor = finalColor + combinedColor * mirrorColorMultiplier;
        out.rgb = finalColor;
        return out;
    }
};

// RGBA8 rule: round(clamp(c,0,1)*255), ties to even as MathF.Round (ColorHelper.cs:16-23); NaN -> 0.
inline uint8_t quantise(float v) {
    if (!(v > 0.0f)) return 0;
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)rintf(v * 255.0f);
}

// ------------------------------------------------------------------------------------------
// scene-side restatement: BoundingBox.cs, RTree.cs, IScene.cs, Triangle.cs, Sphere.cs, ...
// ------------------------------------------------------------------------------------------
const float CS_FLOAT_MAX = 3.40282347e+38f;
// MathF.Min / MathF.Max (.NET 6): NaN-propagating, -0 < +0.
inline float MinF(float a, float b) {
    if (a < b) return a;
    if (a != a) return a;
    if (a == b) return std::signbit(a) ? a : b;
    return b;
}
inline float MaxF(float a, float b) {
    if (a > b) return a;
    if (a != a) return a;
    if (a == b) return std::signbit(a) ? b : a;
    return b;
}

struct BoundingBox {
    V3 MinimumValues, MaximumValues;
    // BoundingBox.cs:17-23
    float GetSize() const {
        float x = MaximumValues.x - MinimumValues.x;
        float y = MaximumValues.y - MinimumValues.y;
        float z = MaximumValues.z - MinimumValues.z;
        return x * y * z;
    }
    // BoundingBox.cs:25-43
    bool FullyContains(const BoundingBox& c) const {
        if (c.MinimumValues.x < MinimumValues.x) return false;
        if (c.MinimumValues.y < MinimumValues.y) return false;
        if (c.MinimumValues.z < MinimumValues.z) return false;
        if (c.MaximumValues.x > MaximumValues.x) return false;
        if (c.MaximumValues.y > MaximumValues.y) return false;
        if (c.MaximumValues.z > MaximumValues.z) return false;
        return true;
    }
    // BoundingBox.cs:45-55
    float CalculatePotentialBoundingBoxSize(BoundingBox inc) const {
        inc.MinimumValues.x = MinF(inc.MinimumValues.x, MinimumValues.x);
        inc.MinimumValues.y = MinF(inc.MinimumValues.y, MinimumValues.y);
        inc.MinimumValues.z = MinF(inc.MinimumValues.z, MinimumValues.z);
        inc.MaximumValues.x = MaxF(inc.MaximumValues.x, MaximumValues.x);
        inc.MaximumValues.y = MaxF(inc.MaximumValues.y, MaximumValues.y);
        inc.MaximumValues.z = MaxF(inc.MaximumValues.z, MaximumValues.z);
        return inc.GetSize();
    }
    // BoundingBox.cs:69-87
    bool Overlap(const BoundingBox& b) const {
        if (b.MinimumValues.x > MaximumValues.x) return false;
        if (b.MinimumValues.y > MaximumValues.y) return false;
        if (b.MinimumValues.z > MaximumValues.z) return false;
        if (b.MaximumValues.x < MinimumValues.x) return false;
        if (b.MaximumValues.y < MinimumValues.y) return false;
        if (b.MaximumValues.z < MinimumValues.z) return false;
        return true;
    }
    // BoundingBox.cs:89-102
    float OverlapSize(const BoundingBox& b) const {
        if (!Overlap(b)) return 0;
        float minX = MaxF(MinimumValues.x, b.MinimumValues.x);
        float minY = MaxF(MinimumValues.y, b.MinimumValues.y);
        float minZ = MaxF(MinimumValues.z, b.MinimumValues.z);
        float maxX = MinF(MaximumValues.x, b.MaximumValues.x);
        float maxY = MinF(MaximumValues.y, b.MaximumValues.y);
        float maxZ = MinF(MaximumValues.z, b.MaximumValues.z);
        return (maxX - minX) * (maxY - minY) * (maxZ - minZ);
    }
};

struct CsException : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// RTree.cs:57-378.  The C# code re-derives each primitive's BoundingBox on every use
// (Sphere.cs:42-51, Triangle.cs:64-80); the values never change, so they are computed once here.
class TreeNode {
  public:
    bool isEmpty = true;
    bool isLeaf = true;
    std::vector<std::unique_ptr<TreeNode>> children;
    std::vector<int> primitivePointers;
    BoundingBox boundingBox;
    TreeNode* parent = nullptr;
    int maximumChildNodes;
    const std::vector<BoundingBox>* primBoxes;

    // RTree.cs:69-84
    TreeNode(int maxChildNodes, int maxValuesPerNode, const std::vector<BoundingBox>* boxes)
        : primitivePointers(maxValuesPerNode, -1), maximumChildNodes(maxChildNodes), primBoxes(boxes) {
        UpdateBoundingBox();
    }

    const BoundingBox& PrimBox(int p) const { return (*primBoxes)[p]; }

    // RTree.cs:180-252
    void AddElement(int primitivePointer) {
        isEmpty = false;
        if (isLeaf) {
            for (size_t i = 0; i < primitivePointers.size(); i++) {
                if (primitivePointers[i] == -1) {
                    primitivePointers[i] = primitivePointer;
                    UpdateBoundingBox();
                    return;
                }
            }
            SplitNode();
            AddElement(primitivePointer);
        } else {
            BoundingBox bb = PrimBox(primitivePointer);
            for (auto& ch : children) {
                if (ch->boundingBox.FullyContains(bb)) {
                    ch->AddElement(primitivePointer);
                    UpdateBoundingBox();
                    return;
                }
            }
            int index = -1;
            float overlapWeight = 8.0f, sizeWeight = 2.0f, growthWeight = 10.0f;
            float lowestScore = 9999999999999.0f;
            float tempScore, tempSize;
            int count = 0;
            for (auto& ch : children) {
                tempScore = 0;
                tempScore += ch->boundingBox.OverlapSize(bb) * overlapWeight;
                tempSize = ch->boundingBox.CalculatePotentialBoundingBoxSize(bb);
                tempScore += tempSize * sizeWeight;
                tempScore += (tempSize - ch->boundingBox.GetSize()) * growthWeight;
                if (tempScore < lowestScore) {
                    index = count;
                    lowestScore = tempScore;
                }
                count++;
            }
            if (index < 0)
                throw CsException("IndexOutOfRangeException: no child scored below the sentinel (RTree.cs:249)");
            children[index]->AddElement(primitivePointer);
            UpdateBoundingBox();
        }
    }

    // RTree.cs:256-330
    void SplitNode() {
        isLeaf = false;
        children.clear();
        for (int i = 0; i < maximumChildNodes; i++) {
            children.emplace_back(new TreeNode(maximumChildNodes, (int)primitivePointers.size(), primBoxes));
            children.back()->parent = this;
        }
        int index0 = -1, index1 = -1;
        float distance = 0;
        int n = (int)primitivePointers.size();
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) {
                if (i == j) continue;
                const BoundingBox& b0 = PrimBox(primitivePointers[i]);
                const BoundingBox& b1 = PrimBox(primitivePointers[j]);
                float smallX = MinF(b0.MinimumValues.x, b1.MinimumValues.x);
                float smallY = MinF(b0.MinimumValues.y, b1.MinimumValues.y);
                float smallZ = MinF(b0.MinimumValues.z, b1.MinimumValues.z);
                float bigX = MaxF(b0.MaximumValues.x, b1.MaximumValues.x);
                float bigY = MaxF(b0.MaximumValues.y, b1.MaximumValues.y);
                float bigZ = MaxF(b0.MaximumValues.z, b1.MaximumValues.z);
                float newDistance = (bigX - smallX) * (bigY - smallY) * (bigZ - smallZ);
                if (distance < newDistance) {
                    distance = newDistance;
                    index0 = i;
                    index1 = j;
                }
            }
        }
        if (index0 < 0)
            throw CsException("IndexOutOfRangeException: all pair volumes are 0 in a full leaf (RTree.cs:303)");
        children[0]->AddElement(primitivePointers[index0]);
        children[1]->AddElement(primitivePointers[index1]);
        std::vector<int> pointersCopy = primitivePointers;
        primitivePointers.clear();
        for (int i = 0; i < (int)pointersCopy.size(); i++) {
            if (i == index0 || i == index1) continue;
            if (parent != nullptr) {
                if (pointersCopy[i] != -1) parent->AddElement(pointersCopy[i]);
            } else {
                if (pointersCopy[i] != -1) AddElement(pointersCopy[i]);
            }
        }
    }

    // RTree.cs:332-377
    void UpdateBoundingBox() {
        float smallX = CS_FLOAT_MAX, smallY = CS_FLOAT_MAX, smallZ = CS_FLOAT_MAX;
        float bigX = -CS_FLOAT_MAX, bigY = -CS_FLOAT_MAX, bigZ = -CS_FLOAT_MAX;
        if (isLeaf) {
            for (int p : primitivePointers) {
                if (p == -1) continue;
                const BoundingBox& b = PrimBox(p);
                smallX = MinF(smallX, b.MinimumValues.x);
                smallY = MinF(smallY, b.MinimumValues.y);
                smallZ = MinF(smallZ, b.MinimumValues.z);
                bigX = MaxF(bigX, b.MaximumValues.x);
                bigY = MaxF(bigY, b.MaximumValues.y);
                bigZ = MaxF(bigZ, b.MaximumValues.z);
            }
        } else {
            for (auto& ch : children) {
                smallX = MinF(smallX, ch->boundingBox.MinimumValues.x);
                smallY = MinF(smallY, ch->boundingBox.MinimumValues.y);
                smallZ = MinF(smallZ, ch->boundingBox.MinimumValues.z);
                bigX = MaxF(bigX, ch->boundingBox.MaximumValues.x);
                bigY = MaxF(bigY, ch->boundingBox.MaximumValues.y);
                bigZ = MaxF(bigZ, ch->boundingBox.MaximumValues.z);
            }
        }
        boundingBox.MinimumValues = v3(smallX, smallY, smallZ);
        boundingBox.MaximumValues = v3(bigX, bigY, bigZ);
        if (parent) parent->UpdateBoundingBox();
    }

    // RTree.cs:111-178.  Each child serialises into a fresh list, so the stored child offsets
    // are relative to this node's first float (raytracer_fs.glsl:184 adds them to pos).
    std::vector<float> TurnIntoFloatArray(std::vector<float> floatList) const {
        if (isEmpty) return floatList;
        floatList.push_back(boundingBox.MinimumValues.x);
        floatList.push_back(boundingBox.MinimumValues.y);
        floatList.push_back(boundingBox.MinimumValues.z);
        floatList.push_back(boundingBox.MaximumValues.x);
        floatList.push_back(boundingBox.MaximumValues.y);
        floatList.push_back(boundingBox.MaximumValues.z);
        floatList.push_back(isLeaf ? 1.0f : 0.0f);
        if (isLeaf) {
            int count = 0;
            for (int p : primitivePointers)
                if (p != -1) count++;
            floatList.push_back((float)count);
            for (int p : primitivePointers)
                if (p != -1) floatList.push_back((float)p);
        } else {
            size_t currentIndex = floatList.size();
            int count = 0;
            floatList.push_back(0);
            for (auto& ch : children) {
                if (ch->boundingBox.GetSize() >= 0) {
                    count++;
                    floatList.push_back(0);
                }
            }
            floatList[currentIndex] = (float)count;
            for (int i = 0; i < count;) {
                floatList[currentIndex + 1 + i] = (float)floatList.size();
                if (children[i]->boundingBox.GetSize() >= 0) {
                    std::vector<float> sub = children[i]->TurnIntoFloatArray(std::vector<float>());
                    floatList.insert(floatList.end(), sub.begin(), sub.end());
                    i++;
                } else {
                    // the C# loop would spin forever here (i is not advanced); cannot happen
                    // because only children[0] and children[1] are ever filled
                    throw CsException("non-terminating serialisation loop (RTree.cs:163-174)");
                }
            }
        }
        return floatList;
    }
};

// Sphere.cs:42-51
BoundingBox SphereBox(const SphereRec& s) {
    BoundingBox b;
    b.MinimumValues = v3(s.center[0] - s.radius, s.center[1] - s.radius, s.center[2] - s.radius);
    b.MaximumValues = v3(s.center[0] + s.radius, s.center[1] + s.radius, s.center[2] + s.radius);
    return b;
}
// Triangle.cs:64-80 on PointA/B/C = Center + VectorX, the same values that are uploaded (RayTracer.cs:988)
BoundingBox TriBox(const TriRec& t) {
    BoundingBox b;
    b.MinimumValues = v3(MinF(t.a[0], MinF(t.b[0], t.c[0])), MinF(t.a[1], MinF(t.b[1], t.c[1])),
                         MinF(t.a[2], MinF(t.b[2], t.c[2])));
    b.MaximumValues = v3(MaxF(t.a[0], MaxF(t.b[0], t.c[0])), MaxF(t.a[1], MaxF(t.b[1], t.c[1])),
                         MaxF(t.a[2], MaxF(t.b[2], t.c[2])));
    return b;
}

// IScene.cs:22-38 + RTree.cs:12-52 (fan-out 7, leaf capacity 5: RTree.cs:18-20)
std::vector<float> BuildAccelerationStructure(const std::vector<SphereRec>& sph, const std::vector<TriRec>& tri) {
    std::vector<BoundingBox> boxes;
    boxes.reserve(sph.size() + tri.size());
    for (auto& s : sph) boxes.push_back(SphereBox(s));
    for (auto& t : tri) boxes.push_back(TriBox(t));
    TreeNode root(7, 5, &boxes);
    for (size_t i = 0; i < sph.size(); i++) root.AddElement((int)i);
    for (size_t i = 0; i < tri.size(); i++) root.AddElement((int)(i + sph.size()));
    return root.TurnIntoFloatArray(std::vector<float>());
}

// Triangle.cs:27-36 and 14-16
struct TriGeom {
    V3 PointA, PointB, PointC, Normal;
};
TriGeom MakeTriangle(V3 A, V3 B, V3 C) {
    V3 Center = (A + B + C) / 3.0f;  // OpenTK 4.8.2 Vector3 operator/(Vector3,float): per-component division (unpinned)
    V3 VectorA = A - Center, VectorB = B - Center, VectorC = C - Center;
    V3 cr = cross(B - A, C - A);
    float len = sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
    TriGeom g;
    g.PointA = Center + VectorA;
    g.PointB = Center + VectorB;
    g.PointC = Center + VectorC;
    g.Normal = cr / len;
    return g;
}

// StreamReader.ReadLine line splitting: terminators \n, \r, \r\n; an unterminated tail is a line.
std::vector<std::string> SplitLines(const std::string& text) {
    std::vector<std::string> lines;
    size_t i = 0, n = text.size();
    while (i < n) {
        size_t j = i;
        while (j < n && text[j] != '\n' && text[j] != '\r') j++;
        lines.emplace_back(text.substr(i, j - i));
        if (j < n) {
            if (text[j] == '\r' && j + 1 < n && text[j + 1] == '\n') j++;
            j++;
        }
        i = j;
    }
    return lines;
}
std::vector<std::string> SplitOn(const std::string& s, const std::string& sep) {
    std::vector<std::string> out;
    size_t i = 0;
    for (;;) {
        size_t j = s.find(sep, i);
        if (j == std::string::npos) {
            out.push_back(s.substr(i));
            return out;
        }
        out.push_back(s.substr(i, j - i));
        i = j + sep.size();
    }
}
float ParseFloat(const std::string& s) { return strtof(s.c_str(), nullptr); }  // correctly rounded like float.Parse
int ParseInt(const std::string& s) { return (int)strtol(s.c_str(), nullptr, 10); }

// OBJImportHelper.cs:19-75.  The loop reads a line, then tests EndOfStream BEFORE processing it,
// so the last line of the file is never processed (teapot: 1024 f-lines, no trailing newline -> 1023).
bool ImportModel(const std::string& path, float scale, V3 Position, std::vector<std::array<V3, 3>>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    std::vector<std::string> lines = SplitLines(ss.str());
    std::vector<V3> verticies;
    for (size_t li = 0; li + 1 < lines.size(); li++) {
        const std::string& line = lines[li];
        if (line.empty()) continue;
        if (line.size() < 3) continue;  // C# would throw on line[1]/line[2]; no asset has such a line
        if (line[0] == 'v' && line[1] == ' ') {
            if (line[2] != ' ') {
                auto c = SplitOn(line.substr(2), " ");
                float x = ParseFloat(c.at(0)), y = ParseFloat(c.at(1)), z = ParseFloat(c.at(2));
                verticies.push_back(Position + v3(x, z, y) * scale);  // y/z swapped (OBJImportHelper.cs:39)
            } else {
                auto c = SplitOn(line.substr(3), "  ");
                float x = ParseFloat(c.at(0)), y = ParseFloat(c.at(1)), z = ParseFloat(c.at(2));
                verticies.push_back(Position + v3(x, y, z) * scale);  // OBJImportHelper.cs:48
            }
        } else if (line[0] == 'f') {
            std::vector<std::string> idx =
                (line[2] != ' ') ? SplitOn(line.substr(2), " ") : SplitOn(line.substr(3), "  ");
            V3 v1 = verticies.at(ParseInt(idx.at(0)) - 1);
            V3 v2 = verticies.at(ParseInt(idx.at(1)) - 1);
            V3 v3_ = verticies.at(ParseInt(idx.at(2)) - 1);
            out.push_back({v1, v2, v3_});
        }
    }
    return true;
}

struct Color4 {
    float R, G, B, A;
};
// OpenTK Color4(byte,byte,byte,byte): component / 255f.  Named colours are byte triples.
Color4 Bytes(int r, int g, int b, int a = 255) {
    return Color4{r / 255.0f, g / 255.0f, b / 255.0f, a / 255.0f};
}
struct Material {
    Color4 DiffuseColor, SpecularColor;
    bool IsPureSpecular;
    float SpecularWidth;
    // Material.cs:15-26
    Material(Color4 d, Color4 s, bool pure, float width)
        : DiffuseColor(d), SpecularColor(s), IsPureSpecular(pure), SpecularWidth(MaxF(width, 1.0f)) {}
};

void FillMaterial(const Material& m, float* diffuse, uint32_t* pure, float* specular, float* specularity) {
    diffuse[0] = m.DiffuseColor.R, diffuse[1] = m.DiffuseColor.G, diffuse[2] = m.DiffuseColor.B;
    specular[0] = m.SpecularColor.R, specular[1] = m.SpecularColor.G, specular[2] = m.SpecularColor.B;
    *pure = m.IsPureSpecular ? 1u : 0u;
    *specularity = m.SpecularWidth;
}

// Camera.cs:92-99 (ctor), 17 (UpDirection), 23-30 (corners); RayTracer.cs:947-954 (packing)
void CameraPack(V3 pos, V3 view, V3 right, float dist, float w, float h, float sw, float sh, float* out) {
    V3 ViewDirection = normalize(view);   // Vector3.Normalized(): scale = 1/Length, component * scale
    V3 RightDirection = normalize(right);
    V3 UpDirection = cross(ViewDirection, RightDirection);
    V3 bl = pos + (ViewDirection * dist) + (-UpDirection * h / 2.0f) - (RightDirection * w / 2.0f);
    V3 br = pos + (ViewDirection * dist) + (-UpDirection * h / 2.0f) + (RightDirection * w / 2.0f);
    V3 tl = pos + (ViewDirection * dist) + (UpDirection * h / 2.0f) + (-RightDirection * w / 2.0f);
    float v[14] = {pos.x, pos.y, pos.z, bl.x, bl.y, bl.z, br.x, br.y, br.z, tl.x, tl.y, tl.z, sw, sh};
    std::memcpy(out, v, sizeof v);
}

}  // namespace

struct orc_scene {
    std::vector<SphereRec> spheres;
    std::vector<PlaneRec> planes;
    std::vector<TriRec> tris;
    std::vector<float> accel;
    std::vector<float> lights;
    float camera[14];
    std::string error;

    void AddSphere(V3 c, float r, const Material& m) {
        SphereRec s{};
        s.center[0] = c.x, s.center[1] = c.y, s.center[2] = c.z;
        s.radius = r;
        FillMaterial(m, s.diffuse, &s.pure, s.specular, &s.specularity);
        spheres.push_back(s);
    }
    void AddPlane(V3 c, V3 n, const Material& m) {
        PlaneRec p{};
        V3 nn = normalize(n);  // Plane.cs:16
        p.position[0] = c.x, p.position[1] = c.y, p.position[2] = c.z;
        p.normal[0] = nn.x, p.normal[1] = nn.y, p.normal[2] = nn.z;
        FillMaterial(m, p.diffuse, &p.pure, p.specular, &p.specularity);
        planes.push_back(p);
    }
    void AddTriangle(V3 A, V3 B, V3 C, const Material& m) {
        TriGeom g = MakeTriangle(A, B, C);
        TriRec t{};
        t.a[0] = g.PointA.x, t.a[1] = g.PointA.y, t.a[2] = g.PointA.z;
        t.b[0] = g.PointB.x, t.b[1] = g.PointB.y, t.b[2] = g.PointB.z;
        t.c[0] = g.PointC.x, t.c[1] = g.PointC.y, t.c[2] = g.PointC.z;
        t.n[0] = g.Normal.x, t.n[1] = g.Normal.y, t.n[2] = g.Normal.z;
        FillMaterial(m, t.diffuse, &t.pure, t.specular, &t.specularity);
        tris.push_back(t);
    }
    bool AddModel(const std::string& path, float scale, V3 pos, const Material& m) {
        std::vector<std::array<V3, 3>> raw;
        if (!ImportModel(path, scale, pos, raw)) {
            error = "cannot read " + path;
            return false;
        }
        for (auto& r : raw) AddTriangle(r[0], r[1], r[2], m);
        return true;
    }
    void AddLight(V3 p, float r, float g, float b) {
        // RayTracer.cs:993-1005
        float v[6] = {p.x, p.y, p.z, r, g, b};
        lights.insert(lights.end(), v, v + 6);
    }
    void Activate() {
        try {
            accel = BuildAccelerationStructure(spheres, tris);
        } catch (const std::exception& e) {
            error = e.what();
        }
    }
};

extern "C" {

orc_scene* orc_scene_testscene1(const char* asset_dir) {
    // TestScene1.cs:16-56
    auto* s = new orc_scene();
    std::string dir = asset_dir ? asset_dir : ".";
    auto P = [&](const char* name) { return dir + "/" + name + ".KarelModel"; };
    const Color4 DimGray = Bytes(105, 105, 105), LightGray = Bytes(211, 211, 211), Red = Bytes(255, 0, 0),
                 Black = Bytes(0, 0, 0), Gray = Bytes(128, 128, 128), Gold = Bytes(255, 215, 0),
                 Purple = Bytes(128, 0, 128), DeepPink = Bytes(255, 20, 147), Yellow = Bytes(255, 255, 0),
                 Turquoise = Bytes(64, 224, 208), White = Bytes(255, 255, 255), Beige = Bytes(245, 245, 220);
    V3 offset = v3(10, 0, 0);
    CameraPack(v3(1, 5.0f, -5), v3(0, -1, 1), v3(1, 0, 0), 1.0f, 1.6f, 0.9f, 1280.0f, 720.0f, s->camera);
    s->AddPlane(v3(0, 0, 0), v3(0, 1, 0), Material(DimGray, LightGray, true, 1.0f));
    s->AddSphere(v3(-1, 0.5f, 0), 0.5f, Material(Red, Black, false, 50.0f));
    s->AddSphere(v3(0, 1, 2), 1.0f, Material(Black, Gray, true, 100.0f));
    s->AddSphere(v3(-1, 0.3f, -1), 0.3f, Material(Gold, LightGray, false, 50.0f));
    s->AddSphere(v3(-2, 0.8f, -2), 0.8f, Material(Purple, Purple, false, 15.0f));
    s->AddSphere(v3(-1, 0.5f, 0) + offset, 0.5f, Material(Red, Gray, false, 50.0f));
    s->AddSphere(v3(0, 1, 2) + offset, 1.0f, Material(DeepPink, LightGray, false, 50.0f));
    s->AddSphere(v3(-1, 0.3f, -1) + offset, 0.3f, Material(Yellow, LightGray, false, 100.0f));
    s->AddSphere(v3(-2, 0.8f, -2) + offset, 0.8f, Material(Black, LightGray, true, 5.0f));
    bool ok = s->AddModel(P("cube"), 0.02f, v3(3, 0, 1), Material(Gold, Gold, true, 1.0f)) &&
              // new Color4(50, 50, 255, 255): all-int literals bind to the byte overload
              s->AddModel(P("pyramid"), 0.03f, v3(1, 0, 0), Material(Black, Bytes(50, 50, 255, 255), true, 1.0f)) &&
              s->AddModel(P("pyramid"), 0.05f, v3(3, 0, -2), Material(Turquoise, White, false, 1.0f)) &&
              s->AddModel(P("teapot"), 0.03f, v3(0, 2, 5), Material(Beige, Gray, false, 0.01f));
    if (!ok) return s;
    // float-literal Color4 ctor: components used as given (TestScene1.cs:44-49)
    s->AddLight(v3(-5, 10, 0), 70, 70, 70);
    s->AddLight(v3(0, 3, 0), 0, 4, 4);
    s->AddLight(v3(0.5f, 2.5f, -3), 3, 0, 6);
    s->AddLight(v3(8, 5, 2), 10, 5, 0);
    s->AddLight(v3(5, 10, 5), 30, 30, 30);
    s->AddLight(v3(30, 20, 0), 300, 300, 300);
    s->Activate();
    return s;
}

orc_scene* orc_scene_from_structs(const void* spheres, int ns, const void* planes, int np, const void* tris,
                                  int nt, const float* lights, int lights_len) {
    auto* s = new orc_scene();
    std::memset(s->camera, 0, sizeof s->camera);
    if (ns > 0) s->spheres.assign((const SphereRec*)spheres, (const SphereRec*)spheres + ns);
    if (np > 0) s->planes.assign((const PlaneRec*)planes, (const PlaneRec*)planes + np);
    if (nt > 0) s->tris.assign((const TriRec*)tris, (const TriRec*)tris + nt);
    if (lights_len > 0) s->lights.assign(lights, lights + lights_len);
    s->Activate();
    return s;
}

int orc_scene_get(const orc_scene* s, int which, const void** ptr, int64_t* count) {
    switch (which) {
        case 0: *ptr = s->spheres.data(); *count = (int64_t)s->spheres.size(); return 0;
        case 1: *ptr = s->planes.data(); *count = (int64_t)s->planes.size(); return 0;
        case 2: *ptr = s->tris.data(); *count = (int64_t)s->tris.size(); return 0;
        case 3: *ptr = s->accel.data(); *count = (int64_t)s->accel.size(); return 0;
        case 4: *ptr = s->lights.data(); *count = (int64_t)s->lights.size(); return 0;
        case 5: *ptr = s->camera; *count = 14; return 0;
    }
    return -1;
}
const char* orc_scene_error(const orc_scene* s) { return s->error.c_str(); }
void orc_scene_free(orc_scene* s) { delete s; }

int orc_import_model(const char* path, float scale, float px, float py, float pz, float* raw9, int cap) {
    std::vector<std::array<V3, 3>> raw;
    if (!ImportModel(path, scale, v3(px, py, pz), raw)) return -1;
    int n = (int)raw.size();
    for (int i = 0; i < n && i < cap; i++)
        for (int k = 0; k < 3; k++) {
            raw9[i * 9 + k * 3 + 0] = raw[i][k].x;
            raw9[i * 9 + k * 3 + 1] = raw[i][k].y;
            raw9[i * 9 + k * 3 + 2] = raw[i][k].z;
        }
    return n;
}

void orc_make_triangle(const float* raw9, float* out12) {
    TriGeom g = MakeTriangle(ld3(raw9), ld3(raw9 + 3), ld3(raw9 + 6));
    const V3 v[4] = {g.PointA, g.PointB, g.PointC, g.Normal};
    for (int i = 0; i < 4; i++) out12[i * 3] = v[i].x, out12[i * 3 + 1] = v[i].y, out12[i * 3 + 2] = v[i].z;
}

void orc_camera_pack(const float* pos3, const float* view3, const float* right3, float dist, float width,
                     float height, float screen_w, float screen_h, float* out14) {
    CameraPack(ld3(pos3), ld3(view3), ld3(right3), dist, width, height, screen_w, screen_h, out14);
}

void orc_intersect_box(int n, const float* o3, const float* d3, const float* box6, int32_t* hit) {
    for (int i = 0; i < n; i++) hit[i] = IntersectBoundingBox(ld3(o3 + 3 * i), ld3(d3 + 3 * i), box6 + 6 * i) ? 1 : 0;
}
void orc_intersect_sphere(int n, const float* o3, const float* d3, const float* c3r, float* out4) {
    for (int i = 0; i < n; i++) {
        V4 r = IntersectSphere(ld3(o3 + 3 * i), ld3(d3 + 3 * i), ld3(c3r + 4 * i), c3r[4 * i + 3]);
        out4[4 * i] = r.x, out4[4 * i + 1] = r.y, out4[4 * i + 2] = r.z, out4[4 * i + 3] = r.w;
    }
}
void orc_intersect_plane(int n, const float* o3, const float* d3, const float* p3, const float* n3, float* t) {
    for (int i = 0; i < n; i++)
        t[i] = IntersectPlane(ld3(o3 + 3 * i), ld3(d3 + 3 * i), ld3(p3 + 3 * i), ld3(n3 + 3 * i));
}
void orc_intersect_triangle(int n, const float* o3, const float* d3, const float* a3, const float* b3,
                            const float* c3, const float* n3, float* t) {
    for (int i = 0; i < n; i++)
        t[i] = IntersectTriangle(ld3(o3 + 3 * i), ld3(d3 + 3 * i), ld3(a3 + 3 * i), ld3(b3 + 3 * i),
                                 ld3(c3 + 3 * i), ld3(n3 + 3 * i));
}

int orc_render(const void* spheres, int ns, const void* planes, int np, const void* triangles, int nt,
               const float* accel, int64_t accel_len, const float* lights, int lights_len,
               const float* camera14, int max_bounces, int x0, int y0, int x1, int y1, float* rgb,
               uint8_t* rgba8, int32_t* primary_id, int32_t* final_id, int32_t* bounces, uint64_t* counters,
               int n_threads) {
    SceneView sv{(const SphereRec*)spheres, ns,          (const PlaneRec*)planes, np,
                 (const TriRec*)triangles,  nt,          accel,                   accel_len,
                 lights,                    lights_len};
    if (x1 < x0 || y1 < y0) return -1;
    const int rw = x1 - x0, rh = y1 - y0;
    uint64_t total[ORC_N_COUNTERS] = {0};
    int bad = 0;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
#endif
    {
        Worker w(&sv);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int ry = 0; ry < rh; ry++) {
            for (int rx = 0; rx < rw; rx++) {
                // gl_FragCoord.xy = pixel centre
                Worker::Pixel p = w.Shade(camera14, (float)(x0 + rx) + 0.5f, (float)(y0 + ry) + 0.5f, max_bounces);
                size_t i = (size_t)ry * rw + rx;
                if (rgb) rgb[3 * i] = p.rgb.x, rgb[3 * i + 1] = p.rgb.y, rgb[3 * i + 2] = p.rgb.z;
                if (rgba8) {
                    rgba8[4 * i] = quantise(p.rgb.x), rgba8[4 * i + 1] = quantise(p.rgb.y);
                    rgba8[4 * i + 2] = quantise(p.rgb.z), rgba8[4 * i + 3] = 255;
                }
                if (primary_id) primary_id[i] = p.primary_id;
                if (final_id) final_id[i] = p.final_id;
                if (bounces) bounces[i] = p.bounces;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            for (int k = 0; k < ORC_N_COUNTERS; k++) {
                if (k >= ORC_C_MAX_STACK && k <= ORC_C_MAX_TRI_CAND)
                    total[k] = std::max(total[k], w.c[k]);
                else
                    total[k] += w.c[k];
            }
            if (w.malformed) bad = 1;
        }
    }
    if (counters) std::memcpy(counters, total, sizeof total);
    return bad ? -2 : 0;
}

}  // extern "C"
