// rt_kernels.h -- the gfx950 kernels: per-pixel render (reference-order and fast variants),
// geometry re-layout, band un-permute and the per-function evaluators.
#pragma once
#include "rt_device.h"

#pragma clang fp contract(off)

namespace rtd {

// counter slots of the reference-order kernel (order of rt_stats)
enum { C_PRIMARY, C_SECONDARY, C_SHADOW, C_NODE, C_REFS, C_SPHERE, C_PLANE, C_TRI, C_MAT, C_N };

// ------------------------------------------------------------------------------------------
// Stackless fixed-order traversal (rt_flatten.h).  kinds: bit 0 = test sphere leaves,
// bit 1 = test triangle leaves.  COUNT adds the reference's node/ref counters.
// ------------------------------------------------------------------------------------------
template <bool COUNT, class LeafSphere, class LeafTri>
__device__ __forceinline__ void walk(const DevScene& S, const BoxRay& br, int kinds, bool count_nodes,
                                     uint32_t* cnt, LeafSphere&& on_sphere, LeafTri&& on_tri) {
    int i = 0;
    const int n = S.n_nodes;
    while (i < n) {
        const float4 lo = S.nodes[2 * i], hi = S.nodes[2 * i + 1];
        const uint32_t meta = __float_as_uint(hi.w);
        if (COUNT && count_nodes && !(meta & rtf::META_CONT)) cnt[C_NODE]++;
        if (!IntersectBoundingBox(br, lo, hi)) {
            i = __float_as_int(lo.w);
            continue;
        }
        const int kind = (meta >> 29) & 3;
        const int count = (meta >> 24) & 31;
        const int first = meta & 0xFFFFFF;
        if (kind == rtf::KIND_BRANCH) {
            if (COUNT && count_nodes) cnt[C_REFS] += first;  // child count
        } else {
            if (COUNT && count_nodes) cnt[C_REFS] += count;
            if (kind == rtf::KIND_SPHERE) {
                if (kinds & 1)
                    for (int k = 0; k < count; k++)
                        if (on_sphere(first + k)) return;
            } else {
                if (kinds & 2)
                    for (int k = 0; k < count; k++)
                        if (on_tri(first + k)) return;
            }
        }
        i++;
    }
}

// ------------------------------------------------------------------------------------------
// FindClosestIntersection, raytracer_fs.glsl:241-297
// ------------------------------------------------------------------------------------------
template <bool REF>
__device__ __forceinline__ Closest find_closest(const DevScene& S, V3 o, V3 d, float minDistance, uint32_t* cnt) {
    Closest h = fresh_closest();
    const BoxRay br = make_box_ray(o, d);
    auto sphere = [&](int slot) {
        const float4 g = S.sphere_geom[slot];
        if (REF) cnt[C_SPHERE]++;
        const float4 r = IntersectSphere(o, d, v3(g.x, g.y, g.z), g.w);
        if (closer(r.w, 0, minDistance, h)) {
            if (REF) cnt[C_MAT]++;
            h.t = r.w, h.kind = 0, h.index = slot, h.n = v3(r.x, r.y, r.z);
        }
        return false;
    };
    auto tri = [&](int slot) {
        const TriG g = load_tri(S.tri_geom, slot);
        if (REF) cnt[C_TRI]++;
        const float t = IntersectTriangle(o, d, g.A, g.B, g.C, g.N);
        if (closer(t, 2, minDistance, h)) {
            if (REF) cnt[C_MAT]++;
            h.t = t, h.kind = 2, h.index = slot;
        }
        return false;
    };
    auto planes = [&]() {
        for (int k = 0; k < S.n_planes; k++) {
            const float4 p = S.plane_geom[2 * k], nn = S.plane_geom[2 * k + 1];
            if (REF) cnt[C_PLANE]++;
            const float t = IntersectPlane(o, d, v3(p.x, p.y, p.z), v3(nn.x, nn.y, nn.z));
            if (closer(t, 1, minDistance, h)) {
                if (REF) cnt[C_MAT]++;
                h.t = t, h.kind = 1, h.index = k;
            }
        }
    };
    if (REF) {
        // literal candidate order: spheres, planes, triangles (two walks; nodes counted once)
        walk<true>(S, br, 1, true, cnt, sphere, tri);
        planes();
        walk<true>(S, br, 2, false, cnt, sphere, tri);
    } else {
        planes();
        walk<false>(S, br, 3, false, cnt, sphere, tri);
    }
    return h;
}

// ------------------------------------------------------------------------------------------
// ObjectInWayOfLight, raytracer_fs.glsl:200-240
// ------------------------------------------------------------------------------------------
template <bool REF>
__device__ __forceinline__ bool object_in_way(const DevScene& S, float distanceToLight, V3 o, V3 d, uint32_t* cnt) {
    const BoxRay br = make_box_ray(o, d);
    bool blocked = false;
    auto in_range = [&](float t) { return t > RT_EPSILON && t < distanceToLight; };
    auto tri = [&](int slot) {
        const TriG g = load_tri(S.tri_geom, slot);
        if (REF) cnt[C_TRI]++;
        if (in_range(IntersectTriangle(o, d, g.A, g.B, g.C, g.N))) blocked = true;
        return blocked;  // stop at the first blocker
    };
    auto plane_blocks = [&]() {
        for (int k = 0; k < S.n_planes; k++) {
            const float4 p = S.plane_geom[2 * k], nn = S.plane_geom[2 * k + 1];
            if (REF) cnt[C_PLANE]++;
            if (in_range(IntersectPlane(o, d, v3(p.x, p.y, p.z), v3(nn.x, nn.y, nn.z)))) return true;
        }
        return false;
    };
    if (REF) {
        cnt[C_SHADOW]++;
        // the shader collects every candidate first (full walk), then tests spheres until the
        // first blocker: keep walking for the node counters, stop testing once blocked
        auto sphere = [&](int slot) {
            if (!blocked) {
                const float4 g = S.sphere_geom[slot];
                cnt[C_SPHERE]++;
                if (in_range(IntersectSphereT(o, d, v3(g.x, g.y, g.z), g.w))) blocked = true;
            }
            return false;
        };
        walk<true>(S, br, 1, true, cnt, sphere, tri);
        if (blocked) return true;
        if (plane_blocks()) return true;
        walk<true>(S, br, 2, false, cnt, sphere, tri);
        return blocked;
    } else {
        // any-hit is order independent: cheapest test first, leave at the first blocker
        if (plane_blocks()) return true;
        auto sphere = [&](int slot) {
            const float4 g = S.sphere_geom[slot];
            if (in_range(IntersectSphereT(o, d, v3(g.x, g.y, g.z), g.w))) blocked = true;
            return blocked;
        };
        walk<false>(S, br, 3, false, cnt, sphere, tri);
        return blocked;
    }
}

// ------------------------------------------------------------------------------------------
// main(), raytracer_fs.glsl:300-425
// ------------------------------------------------------------------------------------------
struct PixelOut {
    V3 rgb;
    int primary_id, final_id, bounces;
};

template <bool REF>
__device__ __forceinline__ V3 light_loop_diffuse(const DevScene& S, V3 hitPos, const Surface& sf, uint32_t* cnt) {
    // raytracer_fs.glsl:342-356 (mirror surfaces with a diffuse component, unflipped normal)
    V3 combined = sf.diffuse * v3(0.1f, 0.1f, 0.1f);
    for (int l = 0; l < S.lights_len; l += 6) {
        const V3 lightPos = v3(S.lights[l], S.lights[l + 1], S.lights[l + 2]);
        const float dist = length(lightPos - hitPos);
        const V3 L = normalize(lightPos - hitPos);
        if (!object_in_way<REF>(S, dist, hitPos, L, cnt)) {
            const V3 lightColor = v3(S.lights[l + 3], S.lights[l + 4], S.lights[l + 5]);
            combined = combined + 1.0f / (dist * dist) * lightColor * sf.diffuse * max0(dot(sf.normal, L));
        }
    }
    return combined;
}

template <bool REF>
__device__ __forceinline__ PixelOut shade_pixel(const DevScene& S, const float* cam, float px, float py,
                                                int maxBounces, uint32_t* cnt) {
    const float width = cam[12], height = cam[13];
    const V3 bottomLeft = v3(cam[3], cam[4], cam[5]);
    const V3 bottomRight = v3(cam[6], cam[7], cam[8]);
    const V3 topLeft = v3(cam[9], cam[10], cam[11]);
    V3 rayOrigin = v3(cam[0], cam[1], cam[2]);
    V3 rayDirection = normalize(
        (bottomLeft + (px / width) * (bottomRight - bottomLeft) + (py / height) * (topLeft - bottomLeft)) - rayOrigin);

    if (REF) cnt[C_PRIMARY]++;
    Closest h = find_closest<REF>(S, rayOrigin, rayDirection, 0.0f, cnt);
    Surface sf;
    sf.diffuse = v3(0, 0, 0), sf.specular = v3(0, 0, 0), sf.normal = v3(0, 0, 0);
    sf.specularity = 1.0f, sf.pure = false, sf.id = -1;
    if (h.kind != 3) sf = resolve_surface(S, h);
    PixelOut out;
    out.primary_id = sf.id;

    V3 mult = v3(1, 1, 1);
    V3 finalColor = v3(0, 0, 0);
    int bounces = 0;
    while (sf.pure && bounces < maxBounces) {
        if (REF) cnt[C_SECONDARY]++;
        const V3 hitPos = rayOrigin + h.t * rayDirection;
        if (sf.diffuse.x != 0 || sf.diffuse.y != 0 || sf.diffuse.z != 0) {
            const V3 combined = light_loop_diffuse<REF>(S, hitPos, sf, cnt);
            finalColor = finalColor + combined * mult;
        }
        mult = mult * sf.specular;
        V3 n = sf.normal;
        if (dot(n, rayDirection) > 0.0f) n = -n;
        const V3 refl = normalize(rayDirection - 2 * dot(rayDirection, n) * n);
        rayOrigin = hitPos;
        rayDirection = refl;
        h = find_closest<REF>(S, rayOrigin, rayDirection, RT_EPSILON, cnt);
        sf.diffuse = v3(0, 0, 0), sf.specular = v3(0, 0, 0), sf.normal = v3(0, 0, 0);
        sf.specularity = 1.0f, sf.pure = false, sf.id = -1;
        if (h.kind != 3) sf = resolve_surface(S, h);
        bounces++;
    }
    out.bounces = bounces;
    out.final_id = sf.id;
    if (h.t == RT_FLT_BIG) {
        out.rgb = finalColor + v3(0.5f, 0.6f, 1.0f) * mult;
        return out;
    }
    // raytracer_fs.glsl:398-424 (a hit that is still pure specular at the bounce cap is shaded too)
    const V3 hitPos = rayOrigin + h.t * rayDirection;
    V3 combined = sf.diffuse * v3(0.1f, 0.1f, 0.1f);
    V3 n = sf.normal;
    for (int l = 0; l < S.lights_len; l += 6) {
        const V3 lightPos = v3(S.lights[l], S.lights[l + 1], S.lights[l + 2]);
        const float dist = length(lightPos - hitPos);
        const V3 L = normalize(lightPos - hitPos);
        if (!object_in_way<REF>(S, dist, hitPos, L, cnt)) {
            const V3 lightColor = v3(S.lights[l + 3], S.lights[l + 4], S.lights[l + 5]);
            if (dot(n, rayDirection) > 0.0f) n = -n;
            const V3 R = normalize(L - 2 * dot(L, n) * n);
            combined = combined + 1.0f / (dist * dist) * lightColor *
                                      (sf.diffuse * max0(dot(n, L)) +
                                       sf.specular * powf(max0(dot(rayDirection, R)), sf.specularity));
        }
    }
    out.rgb = finalColor + combined * mult;
    return out;
}

// RGBA8 rule of rt_mi355x.h: round-half-even(clamp(c,0,1)*255); NaN -> 0
__device__ __forceinline__ uint32_t quantise(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v > 1.0f) v = 1.0f;
    return (uint32_t)rintf(v * 255.0f);
}

struct RenderArgs {
    DevScene scene;
    float cam[14];
    int max_bounces;
    int width, height;   // full frame
    int local_rows;      // rows this launch renders
    int band_rows, rank, world;
    int tiles_x, n_tiles;
    float* rgb;
    uint32_t* rgba8;
    int32_t* primary_id;
    int32_t* final_id;
    int32_t* bounces;
    unsigned long long* stats;  // C_N counters, accumulated (reference-order kernel only)
};

// blocks b and b+8 share an XCD (round-robin dispatch): give each XCD one contiguous run of
// tiles so that its L2 keeps the part of the tree its screen region walks.  Bijective for any
// n (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n / 8, r = n % 8, x = b % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
}

constexpr int TILE = 16;  // 16x16 pixels per 256-thread workgroup, 8x8 per wave

template <bool REF>
__global__ __launch_bounds__(256) void k_render(const RenderArgs a) {
    const int tile = xcd_remap(blockIdx.x, a.n_tiles);
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int r = ty * TILE + (wave >> 1) * 8 + (lane >> 3);           // local row
    const int y = ((r / a.band_rows) * a.world + a.rank) * a.band_rows + r % a.band_rows;  // global row
    const bool live = x < a.width && r < a.local_rows && y < a.height;

    uint32_t cnt[C_N];
    if (REF) {
#pragma unroll
        for (int k = 0; k < C_N; k++) cnt[k] = 0;
    }
    if (live) {
        const PixelOut p = shade_pixel<REF>(a.scene, a.cam, (float)x + 0.5f, (float)y + 0.5f, a.max_bounces, cnt);
        const size_t i = (size_t)r * a.width + x;
        if (a.rgb) a.rgb[3 * i] = p.rgb.x, a.rgb[3 * i + 1] = p.rgb.y, a.rgb[3 * i + 2] = p.rgb.z;
        if (a.rgba8)
            a.rgba8[i] = quantise(p.rgb.x) | (quantise(p.rgb.y) << 8) | (quantise(p.rgb.z) << 16) | 0xFF000000u;
        if (a.primary_id) a.primary_id[i] = p.primary_id;
        if (a.final_id) a.final_id[i] = p.final_id;
        if (a.bounces) a.bounces[i] = p.bounces;
    }
    if (REF && a.stats) {
#pragma unroll
        for (int k = 0; k < C_N; k++) {
            unsigned long long v = cnt[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&a.stats[k], v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// upload-time re-layout: gather the uploaded AoS records into traversal-order geometry slots
// ------------------------------------------------------------------------------------------
__global__ void k_gather_tris(const unsigned char* __restrict__ aos, const int32_t* __restrict__ slot_id,
                              float4* __restrict__ geom, int n_slots) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const float4* rec = (const float4*)(aos + (size_t)slot_id[s] * 96);
    const float4 A = rec[0], B = rec[1], C = rec[2], N = rec[3];
    geom[3 * s] = make_float4(A.x, A.y, A.z, B.x);
    geom[3 * s + 1] = make_float4(B.y, B.z, C.x, C.y);
    geom[3 * s + 2] = make_float4(C.z, N.x, N.y, N.z);
}
__global__ void k_gather_spheres(const unsigned char* __restrict__ aos, const int32_t* __restrict__ slot_id,
                                 float4* __restrict__ geom, int n_slots) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    geom[s] = *(const float4*)(aos + (size_t)slot_id[s] * 48);
}
__global__ void k_gather_planes(const unsigned char* __restrict__ aos, float4* __restrict__ geom, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float4* rec = (const float4*)(aos + (size_t)k * 64);
    geom[2 * k] = rec[0];
    geom[2 * k + 1] = rec[1];
}

// ------------------------------------------------------------------------------------------
// multi-GPU: rank-major gathered bands -> scanline order (elem = 4 or 12 bytes, as dwords)
// ------------------------------------------------------------------------------------------
__global__ void k_unpermute(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ out, int dwords,
                            int width, int height, int band_rows, int world, int padded_rows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // dword index in the output frame
    const size_t row_dw = (size_t)width * dwords;
    if (i >= row_dw * height) return;
    const int y = (int)(i / row_dw);
    const size_t in_row = i % row_dw;
    const int band = y / band_rows;
    const int owner = band % world;
    const int r = (band / world) * band_rows + y % band_rows;  // local row on the owner
    out[i] = gathered[((size_t)owner * padded_rows + r) * row_dw + in_row];
}

// ------------------------------------------------------------------------------------------
// per-function evaluators (parity tests pin the device arithmetic with them)
// ------------------------------------------------------------------------------------------
__global__ void k_eval_box(int n, const float* o, const float* d, const float* b, int32_t* hit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BoxRay br = make_box_ray(v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
    hit[i] = IntersectBoundingBox(br, make_float4(b[6 * i], b[6 * i + 1], b[6 * i + 2], 0),
                                  make_float4(b[6 * i + 3], b[6 * i + 4], b[6 * i + 5], 0))
                 ? 1
                 : 0;
}
__global__ void k_eval_sphere(int n, const float* o, const float* d, const float* c, float4* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = IntersectSphere(v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]),
                             v3(c[4 * i], c[4 * i + 1], c[4 * i + 2]), c[4 * i + 3]);
}
__global__ void k_eval_plane(int n, const float* o, const float* d, const float* p, const float* nn, float* t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    t[i] = IntersectPlane(v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]),
                          v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]), v3(nn[3 * i], nn[3 * i + 1], nn[3 * i + 2]));
}
__global__ void k_eval_triangle(int n, const float* o, const float* d, const float* A, const float* B,
                                const float* Cc, const float* N, float* t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto L = [&](const float* p) { return v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); };
    t[i] = IntersectTriangle(L(o), L(d), L(A), L(B), L(Cc), L(N));
}

}  // namespace rtd
