// rt_device.h -- device-side scene view and the per-ray routines (gfx950, wave64).
//
// The arithmetic of every routine follows raytracer_fs.glsl line by line in the canonical
// binary32 rules of rt_math.h, so that hit ids and ray counts are bit-exact against the
// oracle.  What is NOT taken from the shader is the organisation: the stackless fixed-order
// traversal over the flattened tree (rt_flatten.h), streaming of leaf primitives with an
// explicit tie-break rank instead of candidate lists, and the hot/cold split of geometry and
// material.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_bvh.h"
#include "rt_flatten.h"
#include "rt_math.h"
#include "../../include/rt_mi355x.h"

#pragma clang fp contract(off)

namespace rtd {
using namespace rtm;

// The hot geometry lives in ONE contiguous image of float4s, so that the same bytes can be walked
// in HBM/L2 (GeomGlobal) or copied once into a workgroup's LDS (GeomLds):
//   [0, 2*n_nodes)                      2 x float4 per node (rtf::FlatNode)
//   [tri_off, tri_off + 3*n_tri_slots)  3 x float4 per triangle slot: A.xyz B.x | B.yz C.xy | C.z N.xyz
//   [sph_off, sph_off + n_sph_slots)    1 x float4 per sphere slot: center.xyz radius
//   [bvh_off, geom_float4s)             4 x float4 per node of the re-treed leaves (rt_bvh.h: child boxes as
//                                       per-coordinate {child 0, child 1} pairs, then the two refs)
//   [geom_float4s, ...)                 2 x uint4 per node of the tight tree, if built (large scenes; rtb::QNode)
// The part from tri_off on is what the LDS-resident kernel copies into a workgroup's LDS.
struct DevScene {
    const float4* geom;
    int32_t tri_off, sph_off, geom_float4s;
    const float4* plane_geom;   // 2 x float4 per plane: position.xyz 0 | normal.xyz 0
    const int32_t* tri_slot_id;
    const int32_t* sphere_slot_id;
    const unsigned char* spheres_aos;  // the uploaded SphereStruct / PlaneStruct / TriangleStruct arrays
    const unsigned char* planes_aos;   // (cold: only the winning hit's material is read from them)
    const unsigned char* tris_aos;
    const float* lights;
    int32_t n_nodes, n_planes, n_spheres, n_tris, lights_len;
    float extent[3];  // per-axis max |coordinate| of the root box (scale of the distance-cull pad)
    // distance-cull pad as a fraction of the coordinate scale: 2^-7, or +inf (= never cull) when the
    // upload could not verify that every primitive lies inside its leaf box and every box inside its
    // parent (hand-made accel arrays; the cull's proof needs both)
    float cull_scale;
    // re-treed leaves (rt_bvh.h): 4 x float4 per node, node 0 = root holder; null if not built.
    // They sit at the end of the geometry image (bvh == geom + bvh_off).
    const float4* bvh;
    int32_t bvh_levels;  // stack entries per lane of the staged pipeline (the deeper of the two trees)
    int32_t leaf_levels; // stack entries per lane of the megakernels, which walk the re-treed leaves only
    int32_t bvh_off;
    float root_lo[3], root_hi[3];  // the reference tree's root box (host side: tile priority)
    // sphere around the root box for misses_scene: centre, and the square of (half diagonal + 1e-5 * (|centre|_1 +
    // half diagonal)); +inf = never skip (no tree, or coordinates beyond 1e17)
    float bound_c[3], bound_r2;
    // tight per-primitive tree (rt_bvh.h build_tight), stored behind the image as 32-byte nodes with boxes on a
    // 16-bit grid (rtb::QNode: coordinate = tight_base + index * tight_step); node 0 is its root holder; null =
    // not built / not used.  Only rays with max_k(|o_k| + extent_k) <= tight_e may walk it; *_slot_leaf map a
    // slot to its reference leaf record.
    const uint4* tight;
    float tight_base[3], tight_step[3];
    float tight_e;
    const int32_t* tri_slot_leaf;
    const int32_t* sphere_slot_leaf;
    // second root holder of the tight tree (rtb::Bvh::near_root; 0 = none): rays whose origin lies inside
    // near_lo..near_hi start there -- the same triangles' tree, and the spheres' tree whose pads are sized for them
    uint32_t tight_near_root;
    float near_lo[3], near_hi[3];
};
// where a ray that walks the tight tree starts (tight_root(S, o)); 0 = the first root holder
__device__ __forceinline__ uint32_t tight_root(const DevScene& S, V3 o) {
    const bool inside = (o.x >= S.near_lo[0]) & (o.x <= S.near_hi[0]) & (o.y >= S.near_lo[1]) & (o.y <= S.near_hi[1]) &
                        (o.z >= S.near_lo[2]) & (o.z <= S.near_hi[2]);
    return inside ? S.tight_near_root : 0u;
}

struct GeomGlobal {
    const float4* base;
    int tri_off, sph_off;
    __device__ __forceinline__ float4 node_lo(int i) const { return base[2 * i]; }
    __device__ __forceinline__ float4 node_hi(int i) const { return base[2 * i + 1]; }
    __device__ __forceinline__ float4 tri(int slot, int k) const { return base[tri_off + 3 * slot + k]; }
    __device__ __forceinline__ float4 sphere(int slot) const { return base[sph_off + slot]; }
};
// geometry in HBM/L2 + the re-treed leaves, walked per lane with a stack in LDS (bvh_walk).
// UNB: the scene holds unbounded triangles (rtf::meta_nocull), so child refs must be checked for the
// never-cull mark; scenes without them run the variant that does not spend instructions on it.
template <bool UNB>
struct GeomBvhT {
    static constexpr bool unbounded = UNB;
    const float4* base;
    int tri_off, sph_off;
    const float4* bvh;
    int stack_base;  // dword index of this lane's stack inside the workgroup's LDS; entries 64 dwords apart
#ifdef RT_DIAG_STEPS
    int* diag;
#endif
    __device__ __forceinline__ float4 node_lo(int i) const { return base[2 * i]; }
    __device__ __forceinline__ float4 node_hi(int i) const { return base[2 * i + 1]; }
    __device__ __forceinline__ float4 tri(int slot, int k) const { return base[tri_off + 3 * slot + k]; }
    __device__ __forceinline__ float4 sphere(int slot) const { return base[sph_off + slot]; }
    __device__ __forceinline__ float4 bvh_node(uint32_t i, int k) const {
        return bvh[4 * (size_t)(UNB ? (i & rtb::INDEX_MASK) : i) + k];
    }
    // the two child refs: the first 8 bytes of the node's last quarter
    __device__ __forceinline__ uint2 bvh_refs(uint32_t i) const {
        return *(const uint2*)(bvh + 4 * (size_t)(UNB ? (i & rtb::INDEX_MASK) : i) + 3);
    }
    // half k of node i of the tight tree (rtb::QNode; staged pipeline only)
    const uint4* tight;
    __device__ __forceinline__ uint4 tight_node(uint32_t i, int k) const {
        return tight[2 * (size_t)(UNB ? (i & rtb::INDEX_MASK) : i) + k];
    }
};
// dynamic LDS of the kernels that use it; named directly (not through a generic pointer) so that
// every access is a ds_read / ds_write, never a flat access
extern __shared__ float4 rt_lds[];
// The same, with triangle slots, sphere slots and BVH nodes resident in the workgroup's LDS (small
// scenes): rt_lds[0 ..) holds geometry image [tri_off, geom_float4s).  The exact flat tree (only
// walked by rays with a zero direction component) stays in global memory.
struct GeomBvhLds {
    static constexpr bool unbounded = true;
    const float4* base;
    int sph_l, bvh_l;  // float4 index of the sphere slots / BVH nodes inside rt_lds
    int stack_base;
#ifdef RT_DIAG_STEPS
    int* diag;
#endif
    __device__ __forceinline__ float4 node_lo(int i) const { return base[2 * i]; }
    __device__ __forceinline__ float4 node_hi(int i) const { return base[2 * i + 1]; }
    __device__ __forceinline__ float4 tri(int slot, int k) const { return rt_lds[3 * slot + k]; }
    __device__ __forceinline__ float4 sphere(int slot) const { return rt_lds[sph_l + slot]; }
    __device__ __forceinline__ float4 bvh_node(uint32_t i, int k) const { return rt_lds[bvh_l + 4 * (int)(i & rtb::INDEX_MASK) + k]; }
    __device__ __forceinline__ uint2 bvh_refs(uint32_t i) const {
        const float4 q = rt_lds[bvh_l + 4 * (int)(i & rtb::INDEX_MASK) + 3];
        return make_uint2(__float_as_uint(q.x), __float_as_uint(q.y));
    }
};
template <class G>
struct is_bvh {
    static constexpr bool value = false;
};
template <bool UNB>
struct is_bvh<GeomBvhT<UNB>> {
    static constexpr bool value = true;
};
template <>
struct is_bvh<GeomBvhLds> {
    static constexpr bool value = true;
};

constexpr float RT_FLT_BIG = 3.402823466e+38f;  // raytracer_fs.glsl:315
constexpr float RT_EPSILON = 0.001f;            // raytracer_fs.glsl:4

// ---- raytracer_fs.glsl:93-113 ------------------------------------------------------------
__device__ __forceinline__ float4 IntersectSphere(V3 o, V3 d, V3 center, float radius) {
    V3 v = o - center;
    float b = 2 * (d.x * v.x + d.y * v.y + d.z * v.z);
    float c = v.x * v.x + v.y * v.y + v.z * v.z - radius * radius;
    float disc = b * b - 4 * c;
    if (disc < 0) return make_float4(-1, -1, -1, -1);
    float rootd = sqrtf(disc);
    float t1 = (-b + rootd) / 2;
    float t2 = (-b - rootd) / 2;
    if (t1 < 0) return make_float4(-1, -1, -1, -1);
    float t = (t2 < 0) ? t1 : t2;
    V3 n = normalize((o + t * d) - center);
    return make_float4(n.x, n.y, n.z, t);
}
// distance only (shadow rays use .w alone, raytracer_fs.glsl:213)
__device__ __forceinline__ float IntersectSphereT(V3 o, V3 d, V3 center, float radius) {
    V3 v = o - center;
    float b = 2 * (d.x * v.x + d.y * v.y + d.z * v.z);
    float c = v.x * v.x + v.y * v.y + v.z * v.z - radius * radius;
    float disc = b * b - 4 * c;
    if (disc < 0) return -1;
    float rootd = sqrtf(disc);
    float t1 = (-b + rootd) / 2;
    float t2 = (-b - rootd) / 2;
    if (t1 < 0) return -1;
    return (t2 < 0) ? t1 : t2;
}
// ---- raytracer_fs.glsl:114-120 -----------------------------------------------------------
__device__ __forceinline__ float IntersectPlane(V3 o, V3 d, V3 center, V3 normal) {
    float den = dot(d, normal);
    if (den == 0) return -1;
    return dot(center - o, normal) / den;
}
// ---- raytracer_fs.glsl:121-131 -----------------------------------------------------------
__device__ __forceinline__ float IntersectTriangle(V3 o, V3 d, V3 A, V3 B, V3 C, V3 N) {
    float t = IntersectPlane(o, d, A, N);
    if (t <= 0) return -1;
    V3 X = o + t * d;
    bool outside = dot(cross(B - A, X - A), N) < 0 || dot(cross(C - B, X - B), N) < 0 ||
                   dot(cross(A - C, X - C), N) < 0;
    return outside ? -1.0f : t;
}

// ---- raytracer_fs.glsl:134-161 -----------------------------------------------------------
// The shader recomputes 1/dir and the sign selects per node; they depend on the ray only, so
// they are computed once per ray here (same values, same comparisons).
struct BoxRay {
    V3 o, inv;
    bool sx, sy, sz;
    V3 padinv;  // distance-cull slack per axis in ray-parameter units (IntersectBoundingBoxCull);
                // for a ray that walks the tight tree: -o*inv (SlabHit)
};
__device__ __forceinline__ BoxRay make_box_ray(V3 o, V3 d) {
    BoxRay r;
    r.o = o;
    r.inv = v3(1 / d.x, 1 / d.y, 1 / d.z);
    r.sx = r.inv.x < 0, r.sy = r.inv.y < 0, r.sz = r.inv.z < 0;
    r.padinv = v3(0, 0, 0);
    return r;
}
// extent = per-axis max |coordinate| of the root box (DevScene::extent); scale = DevScene::cull_scale
__device__ __forceinline__ BoxRay make_box_ray_cull(V3 o, V3 d, V3 extent, float scale) {
    BoxRay r = make_box_ray(o, d);
    const float E = fmaxf(fmaxf(fabsf(o.x) + extent.x, fabsf(o.y) + extent.y), fabsf(o.z) + extent.z);
    const float pad = E * scale;  // 2^-7 of the scale of every coordinate the ray can meet
    r.padinv = v3(pad * fabsf(r.inv.x), pad * fabsf(r.inv.y), pad * fabsf(r.inv.z));
    return r;
}
__device__ __forceinline__ bool IntersectBoundingBox(const BoxRay& r, float4 lo, float4 hi) {
    float txmin = ((r.sx ? hi.x : lo.x) - r.o.x) * r.inv.x;
    float txmax = ((r.sx ? lo.x : hi.x) - r.o.x) * r.inv.x;
    float tymin = ((r.sy ? hi.y : lo.y) - r.o.y) * r.inv.y;
    float tymax = ((r.sy ? lo.y : hi.y) - r.o.y) * r.inv.y;
    if ((txmin > tymax) || (tymin > txmax)) return false;
    if (tymin > txmin) txmin = tymin;
    if (tymax < txmax) txmax = tymax;
    float tzmin = ((r.sz ? hi.z : lo.z) - r.o.z) * r.inv.z;
    float tzmax = ((r.sz ? lo.z : hi.z) - r.o.z) * r.inv.z;
    if ((txmin > tzmax) || (tzmin > txmax)) return false;
    return true;
}

// The reference's test above, followed by a conservative distance cull that is NOT in the
// reference: the node is also skipped when no primitive inside it can be accepted by the
// caller's window t_lo < t < t_hi (closest hit: minDistance and the best t so far; shadow ray:
// epsilon and the distance to the light).
//
// Why this cannot change a result (DESIGN.md "distance cull"): an accepted hit has a computed
// parameter t_c whose point X_c = o + t_c*d lies within a distance `pad` of the primitive, hence
// inside the node's box grown by `pad` on every side, hence t_c lies inside the slab interval of
// the grown box, [max_i(tnear_i - pad*|inv_i|), min_i(tfar_i + pad*|inv_i|)].  For a triangle the
// off-plane error of X_c is a few ulps of the scene scale E and its in-plane position passed the
// edge tests (rounded: up to rtb::sliver_reach <= max_extent/512 past a sharp vertex, or the upload
// marks the triangle unbounded and its subtree is never culled); for a sphere the error of t_c is bounded by sqrt(8*4*eps)*(|o-c|+r) < 2.8e-3*E (the
// worst case is a grazing ray, discriminant ~ 0).  pad = E/128 = 7.8e-3*E covers both.  NaN slab
// values (zero direction components) make every comparison false: no cull, as in the reference.
__device__ __forceinline__ bool IntersectBoundingBoxCull(const BoxRay& r, float4 lo, float4 hi, float t_lo,
                                                         float t_hi, float* entry = nullptr, bool nocull = false) {
    const float txn = ((r.sx ? hi.x : lo.x) - r.o.x) * r.inv.x;
    const float txf = ((r.sx ? lo.x : hi.x) - r.o.x) * r.inv.x;
    const float tyn = ((r.sy ? hi.y : lo.y) - r.o.y) * r.inv.y;
    const float tyf = ((r.sy ? lo.y : hi.y) - r.o.y) * r.inv.y;
    // straight-line form of the reference's two early returns (same comparisons, same operands): in
    // a wave-cooperative walk some lane nearly always takes the long path, so the exec-mask
    // branches of the early returns only cost scalar instructions
    const bool miss_xy = (txn > tyf) | (tyn > txf);
    const float mn = (tyn > txn) ? tyn : txn;
    const float mf = (tyf < txf) ? tyf : txf;
    const float tzn = ((r.sz ? hi.z : lo.z) - r.o.z) * r.inv.z;
    const float tzf = ((r.sz ? lo.z : hi.z) - r.o.z) * r.inv.z;
    const bool miss_z = (mn > tzf) | (tzn > mf);
    // ---- not in the reference: padded-interval cull ----
    // nocull: an unbounded triangle lies below this box (rtf::meta_nocull): its hits are not near it
    const float en = fmaxf(fmaxf(txn - r.padinv.x, tyn - r.padinv.y), tzn - r.padinv.z);
    const float ef = fminf(fminf(txf + r.padinv.x, tyf + r.padinv.y), tzf + r.padinv.z);
    const bool culled = ((en > t_hi) | (ef < t_lo)) & !nocull;
    if (entry) *entry = en;  // padded entry parameter: orders the two children of a BVH node
    return !(miss_xy | miss_z | culled);
}

// Slab test of the tight tree (rt_bvh.h build_tight, quantize).  NOT the reference's formula and not meant to
// be: the tight boxes are padded at build time so that this test, with its own rounding (grid coordinates, fma
// form: < 4 eps E in space), never rejects a box whose primitive's test can accept a t in [t_lo, t_hi];
// exactness is restored by the leaf-box check on accepted hits (rt_wavefront.h).  Requires make_tight_ray;
// finite everywhere for |1/d| <= 1e30 and coordinates <= 4e6.
// lo / hi: the node's words of the three lower / upper bounds; SIDE picks the child (its 16-bit half).
// nocull: an unbounded triangle lies below (rtb::ref_nocull): the reference accepts it wherever the ray's LINE
// passes its leaf box, so only the line test applies, not the distance window.
template <int SIDE>
__device__ __forceinline__ float grid_coord(uint32_t w) {
    return (float)(SIDE ? (w >> 16) : (w & 0xFFFFu));  // one v_cvt_f32_u32 with a word select (SDWA)
}
template <int SIDE>
__device__ __forceinline__ bool SlabHitQ(const BoxRay& r, uint32_t lox, uint32_t loy, uint32_t loz, uint32_t hix,
                                         uint32_t hiy, uint32_t hiz, float t_lo, float t_hi, float* entry, bool nocull = false) {
    const float x0 = __builtin_fmaf(grid_coord<SIDE>(lox), r.inv.x, r.padinv.x), x1 = __builtin_fmaf(grid_coord<SIDE>(hix), r.inv.x, r.padinv.x);
    const float y0 = __builtin_fmaf(grid_coord<SIDE>(loy), r.inv.y, r.padinv.y), y1 = __builtin_fmaf(grid_coord<SIDE>(hiy), r.inv.y, r.padinv.y);
    const float z0 = __builtin_fmaf(grid_coord<SIDE>(loz), r.inv.z, r.padinv.z), z1 = __builtin_fmaf(grid_coord<SIDE>(hiz), r.inv.z, r.padinv.z);
    const float en = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float ef = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    *entry = en;
    return (en <= ef) & (((en <= t_hi) & (ef >= t_lo)) | nocull);
}
// May this ray walk the tight tree?  (E_build covers its origin, 1/d small enough that no slab product
// overflows.)  If so the ray is moved onto the tree's grid: inv becomes step/d and padinv (base - o)/d, so that
// the slab parameter of grid index u is fma(u, inv, padinv); the sign flags keep their meaning (step > 0).
// The exact 1/d is gone from the ray after this: the leaf-box check re-forms it (exact_box_ray).
__device__ __forceinline__ bool make_tight_ray(BoxRay& r, V3 extent, float tight_e, const float* base, const float* step) {
    const float E = fmaxf(fmaxf(fabsf(r.o.x) + extent.x, fabsf(r.o.y) + extent.y), fabsf(r.o.z) + extent.z);
    const float big = 1e30f;
    const bool ok = (E <= tight_e) & (fabsf(r.inv.x) <= big) & (fabsf(r.inv.y) <= big) & (fabsf(r.inv.z) <= big);
    if (ok) {
        r.padinv = v3((base[0] - r.o.x) * r.inv.x, (base[1] - r.o.y) * r.inv.y, (base[2] - r.o.z) * r.inv.z);
        r.inv = v3(step[0] * r.inv.x, step[1] * r.inv.y, step[2] * r.inv.z);
    }
    return ok;
}

// child 0's box of a node as the scalar tests take it
__device__ __forceinline__ void node_child0(float4 q0, float4 q1, float4 q2, float4* lo, float4* hi) {
    *lo = make_float4(q0.x, q0.z, q1.x, 0.0f);
    *hi = make_float4(q1.z, q2.x, q2.z, 0.0f);
}

// IntersectTriangle for the fast kernels: same arithmetic, but the three edge tests are only
// evaluated for lanes whose t already lies inside the caller's acceptance window (a t outside it
// is rejected by the caller whatever the edge tests say), and the result is formed without
// nested exec-mask branches.  window(t) must be the caller's own acceptance test.
template <class Window>
__device__ __forceinline__ float IntersectTriangleWindowed(V3 o, V3 d, V3 A, V3 B, V3 C, V3 N, Window&& window) {
    const float den = dot(d, N);
    const float t = dot(A - o, N) / den;  // den == 0 gives inf/NaN: masked out below (reference returns -1)
    const bool candidate = (den != 0) & (t > 0) & window(t);
    if (!candidate) return -1.0f;
    const V3 X = o + t * d;
    const bool outside = (dot(cross(B - A, X - A), N) < 0) | (dot(cross(C - B, X - B), N) < 0) |
                         (dot(cross(A - C, X - C), N) < 0);
    return outside ? -1.0f : t;
}

// Does this ray certainly have no candidates in the tree?  A division-free test in front of the per-ray set-up
// of the per-lane kernels (three correctly rounded 1/d, the cull pads, the root box test: ~70 VALU instructions,
// and most rays of a scene like the teapot's never come near the model).  True only if the ray is regular
// (every |d_k| >= 1e-30, origin within 1e17) and its LINE passes the sphere around the root box at more than
//     sqrt(R'^2 + 1e-4 |c - o|^2),     R' = R + 1e-5 (|c|_1 + R),  R = half diagonal of the root box.
// Why that is safe: if the reference's box test (raytracer_fs.glsl:134-161; a line test, no distance bound)
// accepts the root box for a regular ray, its three computed slab intervals share a parameter t*, and each
// computed bound is off by <= 2 eps |bound - o_k| in space, so the point o + t* d lies within 2 eps E of the box
// on every axis, E = |o| + extent: the line passes the box at less than 4 eps E <= 2.4e-7 (|c - o| + |c| + extent).
// The margins above exceed that by two orders of magnitude (R' - R covers the |c| + extent part, the 1e-4 term
// the |c - o| part; the test's own rounding is ~1e-6 of its terms).  The root box holds every leaf box (nested
// trees only reach the per-lane kernels), so no leaf is accepted either.  NaNs make the comparison false.
__device__ __forceinline__ bool misses_scene(const DevScene& S, V3 o, V3 d) {
    const V3 oc = v3(S.bound_c[0] - o.x, S.bound_c[1] - o.y, S.bound_c[2] - o.z);
    const float q = dot(oc, oc), b = dot(oc, d), dd = dot(d, d);
    const float dmin = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    return (dmin >= 1e-30f) & (q <= 1e34f) & (dd <= 1e30f) & ((q - S.bound_r2) * dd - b * b > 1e-4f * q * dd);
}

// A ray whose origin and 1/d are finite: the precondition of the re-treed traversal (rt_bvh.h).
__device__ __forceinline__ bool ray_is_regular(const BoxRay& r) {
    const float big = 3.402823466e+38f;
    return fabsf(r.inv.x) <= big && fabsf(r.inv.y) <= big && fabsf(r.inv.z) <= big && fabsf(r.o.x) <= 1e18f &&
           fabsf(r.o.y) <= 1e18f && fabsf(r.o.z) <= 1e18f;
}

// ---- geometry fetch ----------------------------------------------------------------------
struct TriG {
    V3 A, B, C, N;
};
template <class G>
__device__ __forceinline__ TriG load_tri(const G& g, int slot) {
    const float4 q0 = g.tri(slot, 0), q1 = g.tri(slot, 1), q2 = g.tri(slot, 2);
    TriG t;
    t.A = v3(q0.x, q0.y, q0.z);
    t.B = v3(q0.w, q1.x, q1.y);
    t.C = v3(q1.z, q1.w, q2.x);
    t.N = v3(q2.y, q2.z, q2.w);
    return t;
}

// Hit being tracked during a closest-hit query.  kind is the reference's candidate class in
// test order (raytracer_fs.glsl:250, 268, 283): 0 sphere, 1 plane, 2 triangle; 3 = none.
struct Closest {
    float t;
    int kind;
    int index;  // sphere slot, plane index or triangle slot
    V3 n;       // sphere normal (computed at acceptance); planes/triangles fetch theirs later
};
__device__ __forceinline__ Closest fresh_closest() { return Closest{RT_FLT_BIG, 3, -1, v3(0, 0, 0)}; }

// The shader tests all spheres, then all planes, then all triangles and accepts on strict
// t < best; the winner is therefore the first candidate in (class, visit order) among those
// of minimal t.  A streaming traversal meets classes interleaved, so equal t is resolved by
// class rank, and within a class by slot (= position in the reference's visit order): an ordered
// traversal meets the leaves in a ray-dependent order, and two intersecting triangles do produce
// bit-identical t on the ray through their common line (seen on Soup1M at 4K).
__device__ __forceinline__ bool closer(float t, int kind, int index, float minDistance, const Closest& h) {
    return t > minDistance && (t < h.t || (t == h.t && (kind < h.kind || (kind == h.kind && index < h.index))));
}

// Material + normal of the winning hit (cold data, one fetch per query that hit something).
struct Surface {
    V3 diffuse, specular, normal;
    float specularity;
    bool pure;
    int id;  // S/T/P id convention of rt_mi355x.h
};
__device__ __forceinline__ Surface resolve_surface(const DevScene& S, const Closest& h) {
    Surface s;
    const unsigned char* rec;
    int mat;  // byte offset of diffuse inside the record
    if (h.kind == 0) {
        int id = S.sphere_slot_id[h.index];
        rec = S.spheres_aos + (size_t)id * 48;
        mat = 16;
        s.normal = h.n;
        s.id = id;
    } else if (h.kind == 1) {
        rec = S.planes_aos + (size_t)h.index * 64;
        mat = 32;
        const float* nf = (const float*)(rec + 16);
        s.normal = v3(nf[0], nf[1], nf[2]);
        s.id = S.n_spheres + S.n_tris + h.index;
    } else {
        int id = S.tri_slot_id[h.index];
        rec = S.tris_aos + (size_t)id * 96;
        mat = 64;
        const float* nf = (const float*)(rec + 48);
        s.normal = v3(nf[0], nf[1], nf[2]);
        s.id = S.n_spheres + id;
    }
    const float4 m0 = *(const float4*)(rec + mat);       // diffuse.rgb, isPureSpecular word
    const float4 m1 = *(const float4*)(rec + mat + 16);  // specular.rgb, specularity
    s.diffuse = v3(m0.x, m0.y, m0.z);
    s.pure = __float_as_uint(m0.w) != 0u;
    s.specular = v3(m1.x, m1.y, m1.z);
    s.specularity = m1.w;
    return s;
}

}  // namespace rtd
