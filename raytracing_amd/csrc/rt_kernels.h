// rt_kernels.h -- the gfx950 kernels: per-pixel render (reference-order, fast-global and
// LDS-resident persistent variants), geometry re-layout, band un-permute and the per-function
// evaluators.
#pragma once
#include "rt_device.h"

#pragma clang fp contract(off)

namespace rtd {

// counter slots of the reference-order kernel (order of rt_stats)
enum { C_PRIMARY, C_SECONDARY, C_SHADOW, C_NODE, C_REFS, C_SPHERE, C_PLANE, C_TRI, C_MAT, C_ZERO_TERM, C_N };

// ------------------------------------------------------------------------------------------
// Stackless fixed-order traversal (rt_flatten.h).  kinds: bit 0 = test sphere leaves,
// bit 1 = test triangle leaves.  COUNT adds the reference's node/ref counters.
// G is the geometry source: GeomGlobal (HBM/L2) or GeomLds (workgroup LDS).
// ------------------------------------------------------------------------------------------
template <bool COUNT, class G, class LeafSphere, class LeafTri>
__device__ __forceinline__ void walk(const G& g, int n_nodes, const BoxRay& br, int kinds, bool count_nodes,
                                     uint32_t* cnt, LeafSphere&& on_sphere, LeafTri&& on_tri) {
    int i = 0;
    while (i < n_nodes) {
        const float4 lo = g.node_lo(i), hi = g.node_hi(i);
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
// Wave-cooperative form of the same walk (the fast kernels).
//
// The visit order of the flattened tree does not depend on the ray, so the lanes of a wavefront
// that enter the walk together can share ONE node index: a node is visited when ANY lane's ray
// hits its box (__ballot), and a lane whose ray misses sits out that subtree (its `resume` index
// is set to the node's skip link).  Each lane still sees exactly the nodes and primitives its own
// ray reaches in the reference, in the same order, so results are unchanged; what changes is the
// memory side: the node index and leaf slots are wave-uniform, so a node or a triangle is ONE
// scalar load (geometry in HBM/L2) or ONE LDS broadcast read per wave instead of a 64-lane
// gather of 2 KB.  Lanes that are finished (any-hit found) pass `done` and stop taking part;
// the walk ends early when no lane is left.  Convergence: every lane that enters stays in the
// loop until the wave-uniform exit, so the ballots are well defined under divergent callers.
// ------------------------------------------------------------------------------------------
// t_lo / t_hi(): the caller's acceptance window for the distance cull (IntersectBoundingBoxCull);
// t_hi is re-read at every node because a closest-hit query narrows it as it goes.
template <class G, class LeafSphere, class LeafTri, class Done, class THi>
__device__ __forceinline__ void wave_walk(const G& g, int n_nodes, const BoxRay& br, bool alive, float t_lo,
                                          THi&& t_hi, LeafSphere&& on_sphere, LeafTri&& on_tri, Done&& done) {
    int i = 0;
    int resume = alive ? 0 : 0x7FFFFFFF;  // first node index at which this lane takes part again
    if (__ballot(alive) == 0ull) return;
    while (i < n_nodes) {
        i = __builtin_amdgcn_readfirstlane(i);
        const float4 lo = g.node_lo(i), hi = g.node_hi(i);
        const uint32_t meta = __float_as_uint(hi.w);
        const int skip = __float_as_int(lo.w);
        const bool active = i >= resume;
        const bool hit = active && IntersectBoundingBoxCull(br, lo, hi, t_lo, t_hi(), nullptr, rtf::meta_nocull(meta));
        if (active && !hit) resume = skip;
        if (__ballot(hit) == 0ull) {
            i = skip;
            continue;
        }
        const int kind = (meta >> 29) & 3;
        if (kind != rtf::KIND_BRANCH) {
            const int count = (meta >> 24) & 31;
            const int first = meta & 0xFFFFFF;
            if (kind == rtf::KIND_SPHERE) {
                for (int k = 0; k < count; k++)
                    if (hit) on_sphere(first + k);
            } else {
                for (int k = 0; k < count; k++)
                    if (hit) on_tri(first + k);
            }
            if (done()) resume = 0x7FFFFFFF;
            if (__ballot(resume != 0x7FFFFFFF) == 0ull) return;
        }
        i++;
    }
}

// ------------------------------------------------------------------------------------------
// Per-lane ordered traversal of the re-treed leaves (rt_bvh.h), stack in LDS.
//
// Every lane walks its own ray: the two child boxes of a node are tested with the reference
// formula (+ the distance cull), the nearer child (smaller padded entry parameter) is entered
// first and the other is pushed.  The stack lives in LDS, one dword per level, entries of one lane
// 64 dwords apart (lane-major rows: whatever the lanes' stack depths, lane l only ever touches bank
// l % 64, so pushes and pops are conflict free).  "while-while" form: the wave first advances every
// lane to its next leaf, then all lanes test their leaf's primitives together.
// Only for regular rays of a nested tree (callers check); results do not depend on the visit
// order because acceptance carries the explicit (class, slot) rank (closer()).
// ------------------------------------------------------------------------------------------
#ifdef RT_DIAG_STEPS  // diagnostic build only: per-lane step counters of bvh_walk (tools/diag_steps.py)
#define rt_diag g.diag
#endif
// Entry of a traversal: node 0 only holds the root (box + ref) as its child 0, so one box test decides
// whether the ray walks anything at all.  Returns the root's ref or NONE.
template <class G>
__device__ __forceinline__ uint32_t bvh_enter(const G& g, const BoxRay& br, float t_lo, float t_hi) {
    float4 lo, hi;
    node_child0(g.bvh_node(0, 0), g.bvh_node(0, 1), g.bvh_node(0, 2), &lo, &hi);
    const uint32_t r0 = g.bvh_refs(0).x;
    return IntersectBoundingBoxCull(br, lo, hi, t_lo, t_hi, nullptr, G::unbounded && rtb::ref_nocull(r0)) ? r0 : 0xFFFFFFFFu;
}

template <class G, class LeafSphere, class LeafTri, class Done, class THi>
__device__ __forceinline__ void bvh_walk(const G& g, const BoxRay& br, bool alive, float t_lo, THi&& t_hi,
                                         LeafSphere&& on_sphere, LeafTri&& on_tri, Done&& done) {
    constexpr uint32_t NONE = 0xFFFFFFFFu, LEAF = 0x80000000u;
    uint32_t* const stk = (uint32_t*)rt_lds + g.stack_base;
    int sp = 0;
    uint32_t cur = NONE;
    // (scenes with unbounded triangles keep those in a second tree under node 0: take the whole node)
    if (alive) cur = G::unbounded ? 0u : bvh_enter(g, br, t_lo, t_hi());
    while (cur != NONE) {
        while (!(cur & LEAF)) {
#ifdef RT_DIAG_STEPS
            rt_diag[0]++;
            if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) rt_diag[2]++;
#endif
            const float4 q0 = g.bvh_node(cur, 0), q1 = g.bvh_node(cur, 1), q2 = g.bvh_node(cur, 2);
            const uint2 refs = g.bvh_refs(cur);
            const float th = t_hi();
            float e0, e1;
            const uint32_t r0 = refs.x, r1 = refs.y;
            const float4 lo0 = make_float4(q0.x, q0.z, q1.x, 0), hi0 = make_float4(q1.z, q2.x, q2.z, 0);
            const float4 lo1 = make_float4(q0.y, q0.w, q1.y, 0), hi1 = make_float4(q1.w, q2.y, q2.w, 0);
            const bool h0 = IntersectBoundingBoxCull(br, lo0, hi0, t_lo, th, &e0, G::unbounded && rtb::ref_nocull(r0));
            const bool h1 = IntersectBoundingBoxCull(br, lo1, hi1, t_lo, th, &e1, G::unbounded && rtb::ref_nocull(r1));
            if (h0 & h1) {
                const bool swap = e1 < e0;
                stk[64 * sp++] = swap ? r0 : r1;
                cur = swap ? r1 : r0;
            } else if (h0 | h1) {
                cur = h0 ? r0 : r1;
            } else {
                cur = sp ? stk[64 * --sp] : NONE;
            }
        }
        if (cur != NONE) {
            const int kind = (cur >> 29) & 3;
            const int count = (cur >> 24) & 31;
            const int first = cur & 0xFFFFFF;
#ifdef RT_DIAG_STEPS
            rt_diag[1] += count;
            if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) rt_diag[3]++;
#endif
            if (kind == rtf::KIND_SPHERE) {
                for (int k = 0; k < count; k++) on_sphere(first + k);
            } else {
                for (int k = 0; k < count; k++) on_tri(first + k);
            }
            cur = (!done() && sp) ? stk[64 * --sp] : NONE;
        }
    }
}

// ------------------------------------------------------------------------------------------
// FindClosestIntersection, raytracer_fs.glsl:241-297
// ------------------------------------------------------------------------------------------
// alive = false (fast kernels only): the lane has no ray (cooperative kernels keep every lane of the wave in
// step); it tests nothing and returns "no hit".
template <bool REF, class G>
__device__ __forceinline__ Closest find_closest(const DevScene& S, const G& g, V3 o, V3 d, float minDistance,
                                                uint32_t* cnt, bool alive = true, bool try_skip = true) {
    Closest h = fresh_closest();
    // (per-lane BVH kernels form the ray's box parameters only if the ray can meet the tree at all, see below)
    BoxRay br;
    if (REF)
        br = make_box_ray(o, d);
    else if (!is_bvh<G>::value)
        br = make_box_ray_cull(o, d, v3(S.extent[0], S.extent[1], S.extent[2]), S.cull_scale);
    // fast kernels: distance only per test; the normal IntersectSphere forms at the accepted t
    // (raytracer_fs.glsl:106-111) is computed once for the winner, after the walk (same expression, same bits)
    auto sphere = [&](int slot) {
        const float4 s = g.sphere(slot);
        if (REF) {
            cnt[C_SPHERE]++;
            const float4 r = IntersectSphere(o, d, v3(s.x, s.y, s.z), s.w);
            if (closer(r.w, 0, slot, minDistance, h)) {
                cnt[C_MAT]++;
                h.t = r.w, h.kind = 0, h.index = slot, h.n = v3(r.x, r.y, r.z);
            }
        } else {
            const float t = IntersectSphereT(o, d, v3(s.x, s.y, s.z), s.w);
            if (closer(t, 0, slot, minDistance, h)) h.t = t, h.kind = 0, h.index = slot;
        }
        return false;
    };
    auto tri = [&](int slot) {
        const TriG tg = load_tri(g, slot);
        if (REF) cnt[C_TRI]++;
        const float t = IntersectTriangle(o, d, tg.A, tg.B, tg.C, tg.N);
        if (closer(t, 2, slot, minDistance, h)) {
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
            if (closer(t, 1, k, minDistance, h)) {
                if (REF) cnt[C_MAT]++;
                h.t = t, h.kind = 1, h.index = k;
            }
        }
    };
    if (REF) {
        // literal candidate order: spheres, planes, triangles (two walks; nodes counted once)
        walk<true>(g, S.n_nodes, br, 1, true, cnt, sphere, tri);
        planes();
        walk<true>(g, S.n_nodes, br, 2, false, cnt, sphere, tri);
    } else {
        if (alive) planes();
        // edge tests only for a t the acceptance rule below could still take
        auto tri_fast = [&](int slot) {
            const TriG tg = load_tri(g, slot);
            const float t = IntersectTriangleWindowed(o, d, tg.A, tg.B, tg.C, tg.N,
                                                      [&](float tt) { return closer(tt, 2, slot, minDistance, h); });
            if (closer(t, 2, slot, minDistance, h)) h.t = t, h.kind = 2, h.index = slot;
        };
        if constexpr (is_bvh<G>::value) {
            // a ray that certainly misses the tree's root box has no candidates (misses_scene): no 1/d, no box test
            if (alive && !(try_skip && misses_scene(S, o, d))) {
                br = make_box_ray_cull(o, d, v3(S.extent[0], S.extent[1], S.extent[2]), S.cull_scale);
                const bool regular = ray_is_regular(br);
                if (!regular) {  // zero direction component etc.: the exact tree, exhaustively (rare)
                    auto tri_b = [&](int slot) {
                        tri_fast(slot);
                        return false;
                    };
                    walk<false>(g, S.n_nodes, br, 3, false, nullptr, sphere, tri_b);
                }
                bvh_walk(g, br, regular, minDistance, [&] { return h.t; }, sphere, tri_fast, [] { return false; });
            }
        } else {
            wave_walk(g, S.n_nodes, br, alive, minDistance, [&] { return h.t; }, sphere, tri_fast,
                      [] { return false; });
        }
        if (h.kind == 0) {
            const float4 s = g.sphere(h.index);
            h.n = normalize((o + h.t * d) - v3(s.x, s.y, s.z));
        }
    }
    return h;
}

// ------------------------------------------------------------------------------------------
// ObjectInWayOfLight, raytracer_fs.glsl:200-240
// ------------------------------------------------------------------------------------------
// alive = false (fast kernels only): the lane has no ray; nothing is tested, the result is meaningless.
template <bool REF, class G>
__device__ __forceinline__ bool object_in_way(const DevScene& S, const G& g, float distanceToLight, V3 o, V3 d,
                                              uint32_t* cnt, bool alive = true, bool try_skip = true) {
    BoxRay br;  // (per-lane BVH kernels: formed only for rays that can meet the tree, as in find_closest)
    if (REF)
        br = make_box_ray(o, d);
    else if (!is_bvh<G>::value)
        br = make_box_ray_cull(o, d, v3(S.extent[0], S.extent[1], S.extent[2]), S.cull_scale);
    bool blocked = false;
    auto in_range = [&](float t) { return t > RT_EPSILON && t < distanceToLight; };
    auto tri = [&](int slot) {
        const TriG tg = load_tri(g, slot);
        if (REF) cnt[C_TRI]++;
        if (in_range(IntersectTriangle(o, d, tg.A, tg.B, tg.C, tg.N))) blocked = true;
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
                const float4 s = g.sphere(slot);
                cnt[C_SPHERE]++;
                if (in_range(IntersectSphereT(o, d, v3(s.x, s.y, s.z), s.w))) blocked = true;
            }
            return false;
        };
        walk<true>(g, S.n_nodes, br, 1, true, cnt, sphere, tri);
        if (blocked) return true;
        if (plane_blocks()) return true;
        walk<true>(g, S.n_nodes, br, 2, false, cnt, sphere, tri);
        return blocked;
    } else {
        // any-hit is order independent: cheapest test first, leave at the first blocker
        // (a lane that is already blocked sits the rest of the walk out; so does a lane without a ray)
        blocked = alive ? plane_blocks() : true;
        auto sphere = [&](int slot) {
            if (blocked) return;
            const float4 s = g.sphere(slot);
            if (in_range(IntersectSphereT(o, d, v3(s.x, s.y, s.z), s.w))) blocked = true;
        };
        auto tri_any = [&](int slot) {
            if (blocked) return;
            const TriG tg = load_tri(g, slot);
            if (in_range(IntersectTriangleWindowed(o, d, tg.A, tg.B, tg.C, tg.N, in_range))) blocked = true;
        };
        if constexpr (is_bvh<G>::value) {
            if (!blocked && !(try_skip && misses_scene(S, o, d))) {
                br = make_box_ray_cull(o, d, v3(S.extent[0], S.extent[1], S.extent[2]), S.cull_scale);
                const bool regular = ray_is_regular(br);
                if (!regular) {
                    auto sphere_b = [&](int slot) {
                        sphere(slot);
                        return blocked;
                    };
                    auto tri_b = [&](int slot) {
                        tri_any(slot);
                        return blocked;
                    };
                    walk<false>(g, S.n_nodes, br, 3, false, nullptr, sphere_b, tri_b);
                }
                bvh_walk(g, br, regular && !blocked, RT_EPSILON, [&] { return distanceToLight; }, sphere, tri_any,
                         [&] { return blocked; });
            }
        } else {
            wave_walk(g, S.n_nodes, br, !blocked, RT_EPSILON, [&] { return distanceToLight; }, sphere, tri_any,
                      [&] { return blocked; });
        }
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

__device__ __forceinline__ Surface no_surface() {
    // raytracer_fs.glsl:315-323, 371-376
    Surface sf;
    sf.diffuse = v3(0, 0, 0), sf.specular = v3(0, 0, 0), sf.normal = v3(0, 0, 0);
    sf.specularity = 1.0f, sf.pure = false, sf.id = -1;
    return sf;
}

// one light of the scene as seen from a shaded point (raytracer_fs.glsl:344-348, 402-406)
struct LightRay {
    V3 L, colour;
    float dist;
};
__device__ __forceinline__ LightRay light_ray(const DevScene& S, int l, V3 hitPos) {
    const float* q = S.lights + 6 * l;
    const V3 lightPos = v3(q[0], q[1], q[2]);
    LightRay r;
    r.dist = length(lightPos - hitPos);
    r.L = normalize(lightPos - hitPos);
    r.colour = v3(q[3], q[4], q[5]);
    return r;
}
// Would this light's term of mirror_diffuse_term (mirror) / phong_term add exactly nothing, whatever its
// shadow ray finds?  Same expressions as those two functions, so the same values: the term is
// c = 1/d^2 * colour * (diffuse * max0(n.L) [+ specular * pow(max0(D.R), specularity)]), and
// powf(0, s) == 0 for s > 0.  True only if every component of c is a zero and adding it cannot change
// the sum (x + 0 == x for every x except x = -0 and c = +0; the sum starts from diffuse * 0.1, so a
// diffuse without sign bits rules that out).  NaN or inf anywhere makes c != 0: no skip.  The staged
// pipeline uses this to leave such shadow rays untraced (a light behind the surface whose mirror
// direction also points away from the viewer: ~40 % of the shadow rays of a triangle soup).
__device__ __forceinline__ bool light_term_is_zero(const Surface& sf, bool mirror, V3 rayDirection, const LightRay& lr) {
    V3 c;
    if (mirror) {
        c = 1.0f / (lr.dist * lr.dist) * lr.colour * sf.diffuse * max0(dot(sf.normal, lr.L));
    } else {
        V3 n = sf.normal;
        if (dot(n, rayDirection) > 0.0f) n = -n;
        const V3 R = normalize(lr.L - 2 * dot(lr.L, n) * n);
        const float a = max0(dot(n, lr.L)), dr = max0(dot(rayDirection, R));
        if (!(dr == 0.0f && sf.specularity > 0.0f)) return false;
        c = 1.0f / (lr.dist * lr.dist) * lr.colour * (sf.diffuse * a + sf.specular * 0.0f);
    }
    const bool zero = c.x == 0.0f && c.y == 0.0f && c.z == 0.0f;
    const bool plain = !(__float_as_uint(sf.diffuse.x) >> 31) && !(__float_as_uint(sf.diffuse.y) >> 31) &&
                       !(__float_as_uint(sf.diffuse.z) >> 31);
    return zero && plain;
}
// raytracer_fs.glsl:342-357: a mirror with a diffuse component (the normal is not flipped here).
// in_shadow(l, lightRay) is the caller's ObjectInWayOfLight (traced inline, or read back from the
// any-hit stage of the wavefront pipeline).
template <class Shadow>
__device__ __forceinline__ V3 mirror_diffuse_term(const DevScene& S, const Surface& sf, V3 hitPos, Shadow&& in_shadow) {
    V3 combined = sf.diffuse * v3(0.1f, 0.1f, 0.1f);
    for (int l = 0; l < S.lights_len / 6; l++) {
        const LightRay lr = light_ray(S, l, hitPos);
        if (!in_shadow(l, lr))
            combined = combined + 1.0f / (lr.dist * lr.dist) * lr.colour * sf.diffuse * max0(dot(sf.normal, lr.L));
    }
    return combined;
}
// raytracer_fs.glsl:361-366: the reflected ray
__device__ __forceinline__ V3 mirror_direction(const Surface& sf, V3 rayDirection) {
    V3 n = sf.normal;
    if (dot(n, rayDirection) > 0.0f) n = -n;
    return normalize(rayDirection - 2 * dot(rayDirection, n) * n);
}
// raytracer_fs.glsl:398-424 (a hit that is still pure specular at the bounce cap is shaded too)
template <class Shadow>
__device__ __forceinline__ V3 phong_term(const DevScene& S, const Surface& sf, V3 hitPos, V3 rayDirection,
                                         Shadow&& in_shadow) {
    V3 combined = sf.diffuse * v3(0.1f, 0.1f, 0.1f);
    V3 n = sf.normal;
    for (int l = 0; l < S.lights_len / 6; l++) {
        const LightRay lr = light_ray(S, l, hitPos);
        if (!in_shadow(l, lr)) {
            if (dot(n, rayDirection) > 0.0f) n = -n;
            const V3 R = normalize(lr.L - 2 * dot(lr.L, n) * n);
            combined = combined + 1.0f / (lr.dist * lr.dist) * lr.colour *
                                      (sf.diffuse * max0(dot(n, lr.L)) +
                                       sf.specular * powf(max0(dot(rayDirection, R)), sf.specularity));
        }
    }
    return combined;
}
// raytracer_fs.glsl:302-313
__device__ __forceinline__ V3 primary_direction(const float* cam, float px, float py) {
    const float width = cam[12], height = cam[13];
    const V3 bottomLeft = v3(cam[3], cam[4], cam[5]);
    const V3 bottomRight = v3(cam[6], cam[7], cam[8]);
    const V3 topLeft = v3(cam[9], cam[10], cam[11]);
    const V3 rayOrigin = v3(cam[0], cam[1], cam[2]);
    return normalize((bottomLeft + (px / width) * (bottomRight - bottomLeft) + (py / height) * (topLeft - bottomLeft)) -
                     rayOrigin);
}

#ifndef RT_SKIP_SEGMENTS
#define RT_SKIP_SEGMENTS 2  // path segments (primary = 0) whose rays try misses_scene first: the rays of a tile are
                            // still near-parallel there, so whole waves skip the set-up; after more bounces some lane
                            // nearly always needs the walk and the test is only overhead
#endif
template <bool REF, class G>
__device__ __forceinline__ PixelOut shade_pixel(const DevScene& S, const G& g, const float* cam, float px, float py,
                                                int maxBounces, uint32_t* cnt) {
    V3 rayOrigin = v3(cam[0], cam[1], cam[2]);
    V3 rayDirection = primary_direction(cam, px, py);

    if (REF) cnt[C_PRIMARY]++;
    Closest h = find_closest<REF>(S, g, rayOrigin, rayDirection, 0.0f, cnt);
    Surface sf = (h.kind != 3) ? resolve_surface(S, h) : no_surface();
    PixelOut out;
    out.primary_id = sf.id;

    V3 mult = v3(1, 1, 1);
    V3 finalColor = v3(0, 0, 0);
    int bounces = 0;
    while (sf.pure && bounces < maxBounces) {
        if (REF) cnt[C_SECONDARY]++;
        const V3 hitPos = rayOrigin + h.t * rayDirection;
        if (sf.diffuse.x != 0 || sf.diffuse.y != 0 || sf.diffuse.z != 0) {
            // fast kernels: a light whose term is exactly zero either way needs no shadow ray
            const V3 combined = mirror_diffuse_term(S, sf, hitPos, [&](int, const LightRay& lr) {
                const bool zero = light_term_is_zero(sf, true, rayDirection, lr);
                if (REF && zero) cnt[C_ZERO_TERM]++;  // the reference traces it; counted for rays_traced (bench)
                if (!REF && zero) return true;
                return object_in_way<REF>(S, g, lr.dist, hitPos, lr.L, cnt, true, bounces < RT_SKIP_SEGMENTS);
            });
            finalColor = finalColor + combined * mult;
        }
        mult = mult * sf.specular;
        const V3 refl = mirror_direction(sf, rayDirection);
        rayOrigin = hitPos;
        rayDirection = refl;
        h = find_closest<REF>(S, g, rayOrigin, rayDirection, RT_EPSILON, cnt, true, bounces + 1 < RT_SKIP_SEGMENTS);
        sf = (h.kind != 3) ? resolve_surface(S, h) : no_surface();
        bounces++;
    }
    out.bounces = bounces;
    out.final_id = sf.id;
    if (h.t == RT_FLT_BIG) {
        out.rgb = finalColor + v3(0.5f, 0.6f, 1.0f) * mult;
        return out;
    }
    const V3 hitPos = rayOrigin + h.t * rayDirection;
    const V3 combined = phong_term(S, sf, hitPos, rayDirection, [&](int, const LightRay& lr) {
        const bool zero = light_term_is_zero(sf, false, rayDirection, lr);
        if (REF && zero) cnt[C_ZERO_TERM]++;
        if (!REF && zero) return true;
        return object_in_way<REF>(S, g, lr.dist, hitPos, lr.L, cnt, true, bounces < RT_SKIP_SEGMENTS);
    });
    out.rgb = finalColor + combined * mult;
    return out;
}

// ------------------------------------------------------------------------------------------
// Cooperative shadow rays (per-lane BVH megakernel).
//
// shade_pixel traces the L shadow rays of a shaded point one after the other in the pixel's own lane.  In a
// scene with mirrors the lanes of a wave leave the bounce loop at different times, and the wave then runs a
// long chain of dependent traversals -- (1 bounce ray + L shadow rays) per bounce -- for the few pixels that
// are still bouncing, with most of its lanes idle (TestScene1: a dozen waves finish a 430 us chain of 70
// traversals per lane, 250 us after every other wave has ended).  Here the wave keeps all its lanes in step
// and regroups the shadow rays of the pixels that need any:
//   * every pixel lane decides per light whether a ray is needed (the exact-zero light term test) and appends
//     (pixel lane, light) to a queue in LDS -- ballot + mbcnt compaction, light-major, so that the rays traced
//     side by side go to one light from neighbouring points;
//   * all 64 lanes then trace 64 queue entries at a time, whoever's pixel they belong to (the hit point comes
//     from LDS, the light ray is formed from it by the expressions shade_pixel uses: same bits), and OR the
//     result into the pixel's mask word in LDS;
//   * every pixel lane shades from its mask, lights in index order, as before.
// With k pixels alive the shadow phase of a bounce takes ceil(k L' / 64) traversal rounds (L' = lights not
// ruled out) instead of L.  Arithmetic per pixel is unchanged; so are the results.
// LDS per wave: hit points 3 x 64 floats, 64 mask words, 64 x COOP_CH 16-bit queue entries.
// ------------------------------------------------------------------------------------------
constexpr int COOP_CH = 8;                          // lights per queue fill
constexpr int COOP_DWORDS = 192 + 64 + 32 * COOP_CH;  // per wave, after the traversal stacks
constexpr int COOP_MAX_LIGHTS = 32;                 // one mask word per pixel; scenes with more lights take shade_pixel

__device__ __forceinline__ void wave_lds_sync() {
    // producer and consumer lanes belong to one wave: LDS operations of a wave complete in issue order, the
    // fences only keep the compiler from moving accesses across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Bit l of the result = light l adds nothing at this point (blocked, or its term is an exact zero either way).
// Every lane of the wave must call this together; wants = this lane shades a point now.
template <class G>
__device__ __forceinline__ uint32_t coop_shadows(const DevScene& S, const G& g, int coop_base, bool wants, V3 hitPos,
                                                 const Surface& sf, bool mirror, V3 rayDirection) {
    if (__ballot(wants) == 0ull) return 0u;
    float* const hx = (float*)rt_lds + coop_base;
    uint32_t* const mask = (uint32_t*)rt_lds + coop_base + 192;
    unsigned short* const q = (unsigned short*)((uint32_t*)rt_lds + coop_base + 256);
    const int lane = (int)__lane_id();
    const int L = S.lights_len / 6;
    hx[lane] = hitPos.x, hx[64 + lane] = hitPos.y, hx[128 + lane] = hitPos.z;
    mask[lane] = 0u;
    uint32_t own = 0u;
    for (int l0 = 0; l0 < L; l0 += COOP_CH) {
        const int l1 = l0 + COOP_CH < L ? l0 + COOP_CH : L;
        int n = 0;  // wave-uniform
        for (int l = l0; l < l1; l++) {
            bool need = false;
            if (wants) {
                const LightRay lr = light_ray(S, l, hitPos);
                if (light_term_is_zero(sf, mirror, rayDirection, lr))
                    own |= 1u << l;
                else
                    need = true;
            }
            const unsigned long long m = __ballot(need);
            if (need) q[n + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(lane | ((l - l0) << 6));
            n += __popcll(m);
        }
        wave_lds_sync();
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool have = i < n;
            V3 o = v3(0, 0, 0), d = v3(0, 0, 0);
            float dist = 0.0f;
            int pl = 0, l = 0;
            if (have) {
                const int e = q[i];
                pl = e & 63, l = l0 + (e >> 6);
                o = v3(hx[pl], hx[64 + pl], hx[128 + pl]);
                const LightRay lr = light_ray(S, l, o);
                d = lr.L, dist = lr.dist;
            }
            const bool blocked = object_in_way<false>(S, g, dist, o, d, nullptr, have);
            if (have && blocked) atomicOr(&mask[pl], 1u << l);
        }
        wave_lds_sync();
    }
    return own | mask[lane];
}

// shade_pixel for a whole wave: every lane of the wave must call this (live = the lane has a pixel).  Per pixel
// the sequence of operations is shade_pixel's.  The mirror loop runs until no lane of the wave bounces any more;
// the shadow rays of every shading step -- each bounce's, and the final Phong term's, which all lanes compute
// together after the loop as in shade_pixel -- are regrouped over the wave (coop_shadows).  That pays even when all
// 64 lanes shade: the queue holds only the rays whose light term is not an exact zero, so a step takes
// ceil(needed rays / 64) rounds where tracing in place takes one round per light with the other lanes idle
// (measured: in-place tracing from 40 shading lanes up, TestScene1 0.443 ms, TeapotTable 0.72; always regrouped,
// final term included, 0.395 and 0.66).
template <class G>
__device__ __forceinline__ PixelOut shade_pixel_coop(const DevScene& S, const G& g, int coop_base, const float* cam, float px,
                                                     float py, int maxBounces, bool live) {
    V3 rayOrigin = v3(cam[0], cam[1], cam[2]);
    V3 rayDirection = live ? primary_direction(cam, px, py) : v3(0, 0, 0);
    Closest h = find_closest<false>(S, g, rayOrigin, rayDirection, 0.0f, nullptr, live);
    Surface sf = (h.kind != 3) ? resolve_surface(S, h) : no_surface();
    PixelOut out;
    out.primary_id = sf.id;
    V3 mult = v3(1, 1, 1);
    V3 finalColor = v3(0, 0, 0);
    int bounces = 0;
    for (;;) {
        const bool mirror = live && sf.pure && bounces < maxBounces;
        if (__ballot(mirror) == 0ull) break;
        const bool lit = mirror && (sf.diffuse.x != 0 || sf.diffuse.y != 0 || sf.diffuse.z != 0);
        const V3 hitPos = rayOrigin + h.t * rayDirection;
        {
            const uint32_t dark = coop_shadows(S, g, coop_base, lit, hitPos, sf, true, rayDirection);
            if (lit)
                finalColor = finalColor + mirror_diffuse_term(S, sf, hitPos, [&](int l, const LightRay&) {
                                              return ((dark >> l) & 1u) != 0u;
                                          }) * mult;
        }
        if (mirror) {
            mult = mult * sf.specular;
            const V3 refl = mirror_direction(sf, rayDirection);
            rayOrigin = hitPos;
            rayDirection = refl;
        }
        // the bounce rays of the lanes that go on (the others sit the walk out)
        const Closest h2 = find_closest<false>(S, g, rayOrigin, rayDirection, RT_EPSILON, nullptr, mirror);
        if (mirror) {
            h = h2;
            sf = (h.kind != 3) ? resolve_surface(S, h) : no_surface();
            bounces++;
        }
    }
    out.bounces = bounces;
    out.final_id = sf.id;
    const bool shades = live && h.t != RT_FLT_BIG;  // the terminal Phong term (the others see the sky)
    const V3 hitPos = rayOrigin + h.t * rayDirection;
    const uint32_t dark = coop_shadows(S, g, coop_base, shades, hitPos, sf, false, rayDirection);
    out.rgb = shades ? finalColor + phong_term(S, sf, hitPos, rayDirection, [&](int l, const LightRay&) {
                                        return ((dark >> l) & 1u) != 0u;
                                    }) * mult
                     : finalColor + v3(0.5f, 0.6f, 1.0f) * mult;
    return out;
}

// RGBA8 rule of rt_mi355x.h: round-half-even(clamp(c,0,1)*255); NaN -> 0
__device__ __forceinline__ uint32_t quantise(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v > 1.0f) v = 1.0f;
    return (uint32_t)rintf(v * 255.0f);
}

// Priority rectangle (super-tile coordinates, inclusive; x0 > x1 = none): its super-tiles are dealt out
// first, the others after them in row-major order.  The longest waves of a frame are those that see the
// mesh (bounce + shadow rays that traverse); started in the middle of the launch they end long after
// everything else has drained (measured on the teapot frame: all waves started by 165 us, the last one
// ended at 272 us; tools/diag_wave_times.py), started first they overlap the rest.  Pure scheduling: every
// tile is still rendered exactly once.
struct PriorityRect {
    int x0, y0, x1, y1;
};
// Division of an index below 2^31 by a divisor that is fixed at launch: n / d = (n * m) >> p with p = 31 + ceil(log2 d),
// m = ceil(2^p / d) < 2^32 (exact for every n < 2^31: m d - 2^p < d <= 2^(p-31), so the excess n (m d - 2^p) stays below
// 2^p).  The block -> tile mapping of every wave divides by five launch constants; as generic 32-bit divisions each
// costs ~30 instructions, among them a v_rcp on a wave-uniform value and a v_readfirstlane round trip (the prologue
// of a wave was ~480 instructions, three quarters of what a sky tile costs); this form is a 64-bit scalar multiply
// and a shift.  The host fills these in (make_fastdiv).
struct FastDiv {
    uint32_t d, m, p;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    if (d == 0) d = 1;
    uint32_t l = 0;
    while (l < 32 && ((uint64_t)1 << l) < d) l++;
    f.d = d, f.p = 31 + l;
    f.m = (uint32_t)((((uint64_t)1 << f.p) + d - 1) / d);
    return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) { return (uint32_t)(((unsigned long long)n * f.m) >> f.p); }
// the divisors of a launch's block -> tile mapping (block_to_tile, priority_super, select_frame, global_row)
struct TileMap {
    FastDiv per;     // tiles per super-tile (supertile^2)
    FastDiv st;      // super-tile edge in tiles
    FastDiv sxn;     // super-tiles per row of them
    FastDiv frames;  // frames of a batched launch
    FastDiv band;    // rows per band (band-sharded frames)
};
struct PriorityDivs {
    FastDiv rw, side;  // width of the priority rectangle / of the rest of a row beside it, in super-tiles
};
constexpr int MAX_BATCH = 8;  // frames one launch of the megakernels can render (rt_render_frames_device)
struct RenderArgs {
    DevScene scene;
    float cam[14];                      // frame 0
    float more_cams[MAX_BATCH - 1][14];  // frames 1 .. n_frames-1 of a batched launch
    int n_frames;                       // groups of 8 blocks go round the frames (select_frame)
    int blocks_per_frame;
    unsigned long long frame_stride;    // pixels between consecutive frames in every output plane
    PriorityRect prio[MAX_BATCH];       // per frame: super-tiles dealt out first (block_to_tile)
    PriorityDivs prio_div[MAX_BATCH];
    TileMap map;
    int max_bounces;
    int width, height;   // full frame
    int local_rows;      // rows this launch renders
    int band_rows, rank, world;
    int tiles_x, tiles_y, supertile;
    // Order by the previous frame (k_render_bvh, single frames with >= RT_HEAT_BOUNCES bounces allowed): order[k] =
    // the super-tile dealt out k-th (k_rank_supers: those where some path bounced that often last time first), or
    // null; heat_out[super-tile] is set by the waves of this launch whose paths bounce that often.
    const int32_t* order;
    uint32_t* heat_out;
    float* rgb;
    uint32_t* rgba8;
    int32_t* primary_id;
    int32_t* final_id;
    int32_t* bounces;
    unsigned long long* stats;  // C_N counters, accumulated (reference-order kernel only)
    // LDS-resident persistent kernel only: ticket counter and this launch's first ticket
    unsigned int* ticket;
    unsigned int ticket_base;
};

// Which frame of a batched launch does this workgroup render?  Everything here is wave-uniform (scalar):
// the camera is picked with compile-time indices only (a dynamic index into the by-value argument struct
// would send the whole struct to scratch memory).  Returns the frame; *block becomes the block index
// inside the frame.
__device__ __forceinline__ int select_frame(const RenderArgs& a, int* block, float cam[14], PriorityRect* prio, PriorityDivs* pdiv) {
    int f = 0;
    if (a.n_frames > 1) {  // (wave-uniform)
        // Groups of 8 consecutive blocks (one per XCD) go round the frames: every frame's priority columns
        // start at once, and every frame is spread over all XCDs.  (One frame after the other: a rank's share
        // of an 8-GPU step 0.284 ms, interleaved 0.216; block-by-block interleave would pin frame f to XCD f.)
        const uint32_t g = (uint32_t)*block >> 3;
        const uint32_t round = fd_div(g, a.map.frames);
        f = (int)(g - round * a.map.frames.d);
        *block = (int)((round << 3) | ((uint32_t)*block & 7u));
    }
    // The frame's camera and priority columns come straight from the kernel-argument segment at a wave-uniform
    // offset (scalar loads).  Indexing the by-value argument struct with a run-time index would send the whole
    // struct to scratch memory, and picking with compile-time indices only -- a chain of selects over the eight
    // frames for each of the 34 words, the earlier form -- cost every wave ~300 scalar instructions.  The render
    // kernels take RenderArgs as their only argument, so the struct starts at offset 0 of the segment; cam and
    // more_cams are adjacent arrays of 14 floats.
    static_assert(offsetof(RenderArgs, more_cams) == offsetof(RenderArgs, cam) + 14 * sizeof(float), "cameras must be contiguous");
    typedef const __attribute__((address_space(4))) char KChar;
    typedef const __attribute__((address_space(4))) float KFloat;
    typedef const __attribute__((address_space(4))) int KInt;
    typedef const __attribute__((address_space(4))) uint32_t KU32;
    KChar* const ka = (KChar*)__builtin_amdgcn_kernarg_segment_ptr();
    KFloat* const cams = (KFloat*)(ka + offsetof(RenderArgs, cam)) + 14 * f;
#pragma unroll
    for (int k = 0; k < 14; k++) cam[k] = cams[k];
    KInt* const prs = (KInt*)(ka + offsetof(RenderArgs, prio)) + 4 * f;
    PriorityRect pr;
    pr.x0 = prs[0], pr.y0 = prs[1], pr.x1 = prs[2], pr.y1 = prs[3];
    static_assert(sizeof(PriorityRect) == 16 && sizeof(PriorityDivs) == 24, "layout of the per-frame launch constants");
    KU32* const pds = (KU32*)(ka + offsetof(RenderArgs, prio_div)) + 6 * f;
    PriorityDivs pd;
    pd.rw.d = pds[0], pd.rw.m = pds[1], pd.rw.p = pds[2], pd.side.d = pds[3], pd.side.m = pds[4], pd.side.p = pds[5];
    *prio = pr, *pdiv = pd;
    return f;
}

__device__ __forceinline__ void store_pixel(const RenderArgs& a, int frame, int x, int r, const PixelOut& p) {
    const size_t i = (size_t)frame * a.frame_stride + (size_t)r * a.width + x;
    if (a.rgb) a.rgb[3 * i] = p.rgb.x, a.rgb[3 * i + 1] = p.rgb.y, a.rgb[3 * i + 2] = p.rgb.z;
    if (a.rgba8) a.rgba8[i] = quantise(p.rgb.x) | (quantise(p.rgb.y) << 8) | (quantise(p.rgb.z) << 16) | 0xFF000000u;
    if (a.primary_id) a.primary_id[i] = p.primary_id;
    if (a.final_id) a.final_id[i] = p.final_id;
    if (a.bounces) a.bounces[i] = p.bounces;
}
// local row (band-sharded frames, rt_mi355x.h) -> global row
__device__ __forceinline__ int global_row(const RenderArgs& a, int r) {
    const uint32_t band = fd_div((uint32_t)r, a.map.band);  // r / band_rows
    return (int)((band * (uint32_t)a.world + (uint32_t)a.rank) * (uint32_t)a.band_rows + ((uint32_t)r - band * (uint32_t)a.band_rows));
}

// Block -> tile mapping.  Blocks b and b+8 share an XCD (round-robin dispatch, MI355X_MICROARCH.md);
// block b is the j-th block of XCD slot x (x = b % 8, j = b / 8).  The image is cut into
// super-tiles of ST x ST tiles that are dealt round-robin to the XCD slots: consecutive blocks of
// one slot walk one super-tile (so the XCD's L2 keeps the subtree that screen region walks), and
// every slot gets super-tiles from all over the image (so sky, floor and mesh work is balanced
// across XCDs -- one contiguous run of tiles per XCD left the sky XCDs idle: 30 % wave-slot use).
// Returns false for the padding blocks of a partial super-tile.
__device__ __forceinline__ void priority_super(uint32_t k, const FastDiv& sxn, const PriorityRect& pr, const PriorityDivs& pd,
                                               uint32_t* sx, uint32_t* sy) {
    const int rw = pr.x1 - pr.x0 + 1, rh = pr.y1 - pr.y0 + 1;
    if (rw <= 0 || rh <= 0) {
        *sy = fd_div(k, sxn), *sx = k - *sy * sxn.d;
        return;
    }
    const uint32_t n_rect = (uint32_t)(rw * rh);
    if (k < n_rect) {
        const uint32_t row = fd_div(k, pd.rw);
        *sx = (uint32_t)pr.x0 + (k - row * pd.rw.d), *sy = (uint32_t)pr.y0 + row;
        return;
    }
    k -= n_rect;
    const uint32_t below = (uint32_t)pr.y0 * sxn.d, side = sxn.d - (uint32_t)rw, mid = (uint32_t)rh * side;
    if (k < below) {
        *sy = fd_div(k, sxn), *sx = k - *sy * sxn.d;
    } else if (k - below < mid) {  // side > 0 here
        k -= below;
        const uint32_t row = fd_div(k, pd.side), c = k - row * pd.side.d;
        *sx = c < (uint32_t)pr.x0 ? c : c + (uint32_t)rw, *sy = (uint32_t)pr.y0 + row;
    } else {
        k -= below + mid;
        const uint32_t row = fd_div(k, sxn);
        *sx = k - row * sxn.d, *sy = (uint32_t)pr.y1 + 1u + row;
    }
}
__device__ __forceinline__ bool block_to_tile(int b, const TileMap& tm, int tiles_x, int tiles_y, const PriorityRect& pr,
                                              const PriorityDivs& pd, int* tx, int* ty, const int32_t* order = nullptr,
                                              int* super_out = nullptr) {
    const uint32_t x = (uint32_t)b & 7u, j = (uint32_t)b >> 3;
    const uint32_t js = fd_div(j, tm.per), in = j - js * tm.per.d;
    const uint32_t super = js * 8u + x;
    uint32_t sx, sy;
    if (order) {
        const uint32_t idx = (uint32_t)order[super];  // (a permutation of the padded super-tile range)
        sy = fd_div(idx, tm.sxn), sx = idx - sy * tm.sxn.d;
    } else {
        priority_super(super, tm.sxn, pr, pd, &sx, &sy);
    }
    if (super_out) *super_out = (int)(sy * tm.sxn.d + sx);
    const uint32_t iy = fd_div(in, tm.st), ix = in - iy * tm.st.d;
    *tx = (int)(sx * tm.st.d + ix);
    *ty = (int)(sy * tm.st.d + iy);
    return *tx < tiles_x && *ty < tiles_y;
}

// Workgroup = RT_BLOCK_WAVES waves, each wave an 8x8 pixel tile.  No LDS and no barrier is used, so
// the workgroup size only sets the scheduling granularity: one wave per workgroup lets the
// dispatcher refill every freed wave slot on its own (measured against 4 waves: DESIGN.md).
#ifndef RT_BLOCK_WAVES
#define RT_BLOCK_WAVES 1
#endif
constexpr int BLOCK_WAVES = RT_BLOCK_WAVES;
constexpr int TILE_X = BLOCK_WAVES >= 2 ? 16 : 8;  // pixels per workgroup tile
constexpr int TILE_Y = BLOCK_WAVES >= 4 ? 16 : 8;
static_assert(BLOCK_WAVES == 1 || BLOCK_WAVES == 2 || BLOCK_WAVES == 4, "1, 2 or 4 waves per workgroup");

// ---- geometry in HBM/L2: one thread per pixel, one 16x16 tile per workgroup -------------------
template <bool REF>
__global__ __launch_bounds__(64 * BLOCK_WAVES) void k_render(const RenderArgs a) {
    int tx, ty, block = blockIdx.x;
    float cam[14];
    PriorityRect prio;
    PriorityDivs pdiv;
    const int frame = select_frame(a, &block, cam, &prio, &pdiv);
    const bool in_image = block_to_tile(block, a.map, a.tiles_x, a.tiles_y, prio, pdiv, &tx, &ty);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = tx * TILE_X + (wave & 1) * 8 + (lane & 7);
    const int r = ty * TILE_Y + (wave >> 1) * 8 + (lane >> 3);  // local row
    const int y = global_row(a, r);
    const bool live = in_image && x < a.width && r < a.local_rows && y < a.height;
    const GeomGlobal g{a.scene.geom, a.scene.tri_off, a.scene.sph_off};

    uint32_t cnt[C_N];
    if (REF) {
#pragma unroll
        for (int k = 0; k < C_N; k++) cnt[k] = 0;
    }
#ifdef RT_DIAG_WAVE_CYCLES  // diagnostic build only: the bounces plane receives each wave's lifetime in cycles
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (live) {
        const PixelOut p = shade_pixel<REF>(a.scene, g, cam, (float)x + 0.5f, (float)y + 0.5f, a.max_bounces, cnt);
        store_pixel(a, frame, x, r, p);
    }
#ifdef RT_DIAG_WAVE_CYCLES
    if (!REF && live && a.bounces)
        a.bounces[(size_t)frame * a.frame_stride + (size_t)r * a.width + x] = (int32_t)(__builtin_amdgcn_s_memtime() - diag_t0);
#endif
    if (REF && a.stats) {
#pragma unroll
        for (int k = 0; k < C_N; k++) {
            unsigned long long v = cnt[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&a.stats[k], v);
        }
    }
}

// ---- order by the previous frame ---------------------------------------------------------------------
// A frame with mirrors ends when its longest wave ends, and the longest waves are those whose paths keep bouncing
// (TestScene1 720p: 300 us of a 400 us launch, and dealt out bottom to top they start 100 us into it).  Where they
// are is not known before the frame is rendered -- but it is known from the frame before: waves whose paths bounce
// RT_HEAT_BOUNCES times or more mark their super-tile, and the next launch of the same size deals the marked
// super-tiles out first (one workgroup ranks them: heat -> order, a stable partition of the padded super-tile
// range).  Scheduling only: order is a permutation whatever the marks say, every tile is rendered exactly once, and
// a camera cut merely costs that frame the head start.
#ifndef RT_HEAT_BOUNCES
#define RT_HEAT_BOUNCES 3  // launches whose paths may bounce at least this often take part
#endif
#ifndef RT_HEAT_MIN_TICKS
#define RT_HEAT_MIN_TICKS 3000u  // waves that lived 30 us or longer report
#endif
// (clears `next`, the words this launch's waves will report into, on the way: one stream operation less per frame)
__global__ __launch_bounds__(1024) void k_rank_supers(const uint32_t* __restrict__ heat, int n, int32_t* __restrict__ order,
                                                      uint32_t* __restrict__ next) {
    // 16 waves, each owns a contiguous run of super-tiles (a multiple of 64 long) and walks it 64 at a time: a ballot
    // gives every lane its rank among the hot ones of the step.  Three barriers in all.
    __shared__ uint32_t longest;
    __shared__ int wave_hot[16];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int run = ((n + 16 * 64 - 1) / (16 * 64)) * 64;
    const int b = min(n, wave * run), e = min(n, b + run);
    if (t == 0) longest = 0u;
    __syncthreads();
    uint32_t mx = 0u;
    for (int i = b + lane; i < e; i += 64) mx = max(mx, heat[i]), next[i] = 0u;
    if (mx) atomicMax(&longest, mx);
    __syncthreads();
    const uint32_t bar = longest / 2u + 1u;  // hot: a wave there lived more than half as long as the longest of the frame
    int hot = 0;                              // (wave-uniform) hot super-tiles of this wave's run
    for (int i0 = b; i0 < e; i0 += 64) {
        const int i = i0 + lane;
        hot += __popcll(__ballot(i < e && heat[i] >= bar));
    }
    if (lane == 0) wave_hot[wave] = hot;
    __syncthreads();
    int hot_before = 0, total_hot = 0;
    for (int w = 0; w < 16; w++) {
        hot_before += w < wave ? wave_hot[w] : 0;
        total_hot += wave_hot[w];
    }
    for (int i0 = b; i0 < e; i0 += 64) {
        const int i = i0 + lane;
        const bool is_hot = i < e && heat[i] >= bar;
        const unsigned long long m = __ballot(is_hot);
        const int rank = hot_before + __popcll(m & ((1ull << lane) - 1ull));  // hot super-tiles before i
        if (i < e) order[is_hot ? rank : total_hot + (i - rank)] = i;           // cold ones behind all hot ones, in index order
        hot_before += __popcll(m);
    }
}

// ---- large scenes: per-lane ordered traversal of the re-treed leaves, stacks in LDS ----------------
// Same pixel mapping as k_render; dynamic LDS = BLOCK_WAVES * 64 lanes * leaf_levels dwords.
#ifndef RT_BVH_WAVES_PER_EU
#define RT_BVH_WAVES_PER_EU 4  // 128 VGPRs: left to itself the kernel takes 131 and loses the fourth wave per SIMD
                               // (teapot 4K 0.58 -> 0.65 ms)
#endif
// UNB: see GeomBvhT.  COOP: the shadow rays of a bounce are regrouped over the lanes of the wave (coop_shadows);
// dynamic LDS then also holds COOP_DWORDS per wave behind the stacks.
#ifndef RT_COOP_WAVES_PER_EU
#define RT_COOP_WAVES_PER_EU 4
#endif
template <bool UNB, bool COOP, bool HEAT = false>
__global__ __launch_bounds__(64 * BLOCK_WAVES)
    __attribute__((amdgpu_waves_per_eu(COOP ? RT_COOP_WAVES_PER_EU : RT_BVH_WAVES_PER_EU, 8))) void k_render_bvh(const RenderArgs a) {
    int tx, ty, block = blockIdx.x;
    float cam[14];
    PriorityRect prio;
    PriorityDivs pdiv;
    const int frame = select_frame(a, &block, cam, &prio, &pdiv);
    int super_idx = 0;
    unsigned long long heat_t0 = 0;
    if (HEAT) heat_t0 = __builtin_amdgcn_s_memrealtime();
    const bool in_image = HEAT ? block_to_tile(block, a.map, a.tiles_x, a.tiles_y, prio, pdiv, &tx, &ty, a.order, &super_idx)
                               : block_to_tile(block, a.map, a.tiles_x, a.tiles_y, prio, pdiv, &tx, &ty);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = tx * TILE_X + (wave & 1) * 8 + (lane & 7);
    const int r = ty * TILE_Y + (wave >> 1) * 8 + (lane >> 3);
    const int y = global_row(a, r);
    const bool live = in_image && x < a.width && r < a.local_rows && y < a.height;
#ifdef RT_DIAG_STEPS
    int diag[4] = {0, 0, 0, 0};
    const GeomBvhT<UNB> g{a.scene.geom, a.scene.tri_off, a.scene.sph_off, a.scene.bvh, wave * 64 * a.scene.leaf_levels + lane, diag};
#else
    const GeomBvhT<UNB> g{a.scene.geom, a.scene.tri_off, a.scene.sph_off, a.scene.bvh, wave * 64 * a.scene.leaf_levels + lane};
#endif
#ifdef RT_DIAG_WAVE_TIMES  // diagnostic build only: id planes receive each wave's start / end on the 100 MHz clock
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (COOP) {  // every lane takes part, with or without a pixel
        const int coop_base = BLOCK_WAVES * 64 * a.scene.leaf_levels + wave * COOP_DWORDS;
        const PixelOut p = shade_pixel_coop(a.scene, g, coop_base, cam, (float)x + 0.5f, (float)y + 0.5f, a.max_bounces, live);
        if (live) store_pixel(a, frame, x, r, p);
    } else if (live) {
        const PixelOut p =
            shade_pixel<false>(a.scene, g, cam, (float)x + 0.5f, (float)y + 0.5f, a.max_bounces, nullptr);
        store_pixel(a, frame, x, r, p);
#ifdef RT_DIAG_WAVE_TIMES
        {
            const size_t i = (size_t)frame * a.frame_stride + (size_t)r * a.width + x;
            if (a.primary_id) a.primary_id[i] = (int32_t)(uint32_t)diag_t0;
            if (a.final_id) a.final_id[i] = (int32_t)(uint32_t)__builtin_amdgcn_s_memrealtime();
        }
#endif
#ifdef RT_DIAG_STEPS  // planes receive: bounces <- node steps, final_id <- primitive tests, primary_id <- wave
                      // iterations (node << 16 | leaf) credited to the first active lane, rgb.x <- unused
        const size_t i = (size_t)r * a.width + x;
        if (a.bounces) a.bounces[i] = diag[0];
        if (a.final_id) a.final_id[i] = diag[1];
        if (a.primary_id) a.primary_id[i] = diag[2];
        if (a.rgb) a.rgb[3 * i] = (float)diag[3];
#endif
    }
    if (HEAT && in_image) {  // this wave's lifetime in 10 ns ticks, if it was a long one: its super-tile starts earlier next time
        const uint32_t ticks = (uint32_t)(__builtin_amdgcn_s_memrealtime() - heat_t0);
        if (ticks >= RT_HEAT_MIN_TICKS && (threadIdx.x & 63) == 0) atomicMax(&a.heat_out[super_idx], ticks);
    }
}

// ---- small scenes: the same per-lane traversal with triangles, spheres and BVH nodes in LDS ---------
// One 1024-thread workgroup per CU copies geometry image [tri_off, geom_float4s) into LDS once
// (teapot scene: 70 KB), the per-lane stacks follow it.  Every node and triangle fetch of the
// traversal is then a ds_read_b128 (tens of cycles) instead of a gather through L1/L2 (hundreds),
// which is what the 4 waves per SIMD of the megakernel cannot hide.
// Work distribution: each wave draws tickets (one 8x8 pixel tile each, in the XCD-interleaved
// super-tile order of block_to_tile) from one device-scope counter until the frame is exhausted --
// dynamic balance between sky and mesh tiles, and every wave reaches the exit condition
// (ticket >= n_tickets), so the grid always drains.  The counter is never reset: the host passes
// the first ticket of this launch (ticket_base) and advances it by n_tickets + number of waves,
// which is exactly what the launch consumes.
constexpr int LDS_BLOCK = 1024;
__global__ __launch_bounds__(LDS_BLOCK) void k_render_bvh_lds(const RenderArgs a) {
    const int n_copy = a.scene.geom_float4s - a.scene.tri_off;
    for (int i = threadIdx.x; i < n_copy; i += LDS_BLOCK) rt_lds[i] = a.scene.geom[a.scene.tri_off + i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const GeomBvhLds g{a.scene.geom, a.scene.sph_off - a.scene.tri_off, a.scene.bvh_off - a.scene.tri_off,
                       n_copy * 4 + wave * 64 * a.scene.leaf_levels + lane};
    const int per = a.supertile * a.supertile;
    const int n_super = ((a.tiles_x + a.supertile - 1) / a.supertile) * ((a.tiles_y + a.supertile - 1) / a.supertile);
    const unsigned int n_tickets = (unsigned int)(n_super * per);
    for (;;) {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(a.ticket, 1u) - a.ticket_base;
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= n_tickets) break;
        // ticket -> tile: consecutive tickets walk one super-tile
        const int super = (int)(t / per), in = (int)(t % per);
        const int sx_n = (a.tiles_x + a.supertile - 1) / a.supertile;
        const int tx = (super % sx_n) * a.supertile + in % a.supertile;
        const int ty = (super / sx_n) * a.supertile + in / a.supertile;
        const int x = tx * 8 + (lane & 7);
        const int r = ty * 8 + (lane >> 3);
        if (tx < a.tiles_x && ty < a.tiles_y && x < a.width && r < a.local_rows) {
            const int y = global_row(a, r);
            const PixelOut p =
                shade_pixel<false>(a.scene, g, a.cam, (float)x + 0.5f, (float)y + 0.5f, a.max_bounces, nullptr);
            store_pixel(a, 0, x, r, p);
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
// gathered: [world][n_frames][padded_rows][width] elements, out: [n_frames][height][width]
__global__ void k_unpermute(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ out, int dwords,
                            int width, int height, int band_rows, int world, int padded_rows, int n_frames) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // dword index in the output frames
    const size_t row_dw = (size_t)width * dwords;
    if (i >= row_dw * height * n_frames) return;
    const size_t row = i / row_dw;
    const int f = (int)(row / height), y = (int)(row % height);
    const size_t in_row = i % row_dw;
    const int band = y / band_rows;
    const int owner = band % world;
    const int r = (band / world) * band_rows + y % band_rows;  // local row on the owner
    out[i] = gathered[(((size_t)owner * n_frames + f) * padded_rows + r) * row_dw + in_row];
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
