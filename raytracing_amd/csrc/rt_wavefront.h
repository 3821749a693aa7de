// rt_wavefront.h -- the staged ("wavefront") form of the per-pixel path for large scenes.
//
// A megakernel keeps one pixel per lane from the primary ray to the last shadow ray.  On a scene whose
// rays need very different numbers of traversal steps (Soup1M: mean 135 node steps, heavy tail) a wave
// then runs at the pace of its slowest lane: measured lane utilisation of the traversal loop 9 %
// (tools/diag_steps.py).  Here the path is cut at every ray, and the rays of ALL pixels are regrouped:
//
//   for iteration i = 0 .. max_bounces            (i = number of mirror bounces already taken)
//     k_wf_closest  persistent waves; lanes whose ray is finished draw the next rays of the active list
//                   together (ballot + mbcnt compaction of the idle lanes, one atomic per refill of
//                   WF_REFILL lanes), so the traversal loop keeps >= 40 of 64 lanes supplied with rays
//     k_wf_any      the same for the shadow rays: work item (active path j, light l)
//     k_wf_shade    one thread per active path: the reference's shading code for this bounce
//                   (raytracer_fs.glsl:334-381 or 384-424), appends mirror paths to the next list
//
// Per-path state lives in HBM as four float4 planes indexed by the local pixel; what a stage does not
// change it does not touch.  The arithmetic of every stage is the megakernel's (same helpers, same
// order of operations per pixel, lights summed in index order), so results are identical.
#pragma once
#include "rt_kernels.h"

#pragma clang fp contract(off)

namespace rtd {

struct WfArgs {
    DevScene scene;
    float cam[14];
    int max_bounces;
    int width, height, local_rows, band_rows, rank, world;
    int tiles_x, tiles_y;  // 8x8-pixel tiles of the local frame: iteration 0 walks pixels in tile order
    FastDiv fd_band, fd_tiles_x;  // divisions by band_rows and tiles_x
    int n_lights;
    int iter;
    uint32_t n0;  // work items of iteration 0 = tiles * 64
    float* rgb;
    uint32_t* rgba8;
    int32_t* primary_id;
    int32_t* final_id;
    int32_t* bounces;
    float4* org_t;    // ray origin.xyz | t of the closest hit (RT_FLT_BIG = miss)
    float4* dir_hit;  // ray direction.xyz | packed hit (kind << 28 | slot or plane index)
    float4* mult_b;   // mult.xyz | -
    float4* final_;   // finalColor.xyz | -
    uint32_t* list[2];       // active pixels of odd / even iterations >= 1
    unsigned char* blocked;  // [active index j * n_lights + l]: shadow ray result of this iteration
    uint32_t* counters;      // per iteration i: [4i] active paths, [4i+1] closest fetch head, [4i+2] any-hit fetch head
    // binning of the continuing paths by the cell of their new origin (k_wf_bin_scan / k_wf_bin_scatter below):
    // lists of iterations >= bin_from are sorted (0: none)
    int bin_from;
    float bin_scale[3];  // cells per unit length of each axis over [-extent, +extent]
    uint32_t* list_tmp;  // the unsorted list k_wf_shade appends to
    uint32_t* keys;      // [entry of list_tmp]: its cell
    uint32_t* bins;      // [WF_BINS]: counts, then running offsets
};
constexpr int WF_BIN_BITS = 4;                    // cells per axis = 16
constexpr int WF_BINS = 8 << (3 * WF_BIN_BITS);   // x 8 direction octants  // Morton-ordered cells of the root box
__device__ __forceinline__ bool wf_binned(const WfArgs& a, int iter) { return a.bin_from > 0 && iter >= a.bin_from; }
__device__ __forceinline__ uint32_t wf_spread3(uint32_t v) {  // 5 bits -> every third bit
    v = (v | (v << 8)) & 0x0000F00Fu;
    v = (v | (v << 4)) & 0x000C30C3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
// arr[key] += 1 for every active lane, one atomic per distinct key of the wave (hit points of a plane outside the root
// box all fall into a few border cells: lane-by-lane atomics on one address serialise at ~12 ns each); returns the
// lane's place: the old value plus its rank among the wave's lanes with the same key
__device__ __forceinline__ uint32_t wf_wave_add(uint32_t* arr, uint32_t key, bool active) {
    uint32_t place = 0;
    unsigned long long todo = __ballot(active);
    const int lane = (int)__lane_id();
    while (todo) {
        const int leader = __ffsll(todo) - 1;
        const uint32_t k = (uint32_t)__shfl((int)key, leader, 64);
        const unsigned long long same = __ballot(active && key == k) & todo;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(arr + k, (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if ((same >> lane) & 1ull) place = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return place;
}
__device__ __forceinline__ uint32_t wf_cell(const WfArgs& a, V3 x) {
    const int top = (1 << WF_BIN_BITS) - 1;
    // (points of the planes lie outside the root box: clamped to the border cells; NaN converts to 0)
    const int cx = min(max((int)((x.x + a.scene.extent[0]) * a.bin_scale[0]), 0), top);
    const int cy = min(max((int)((x.y + a.scene.extent[1]) * a.bin_scale[1]), 0), top);
    const int cz = min(max((int)((x.z + a.scene.extent[2]) * a.bin_scale[2]), 0), top);
    return wf_spread3((uint32_t)cx) | (wf_spread3((uint32_t)cy) << 1) | (wf_spread3((uint32_t)cz) << 2);
}

constexpr uint32_t WF_NO_HIT = 3u << 28;
// bit 27 of a packed hit (between kind and index, masked off by every reader).  On the hit of iteration 0: the closest
// hit of this pixel's mirror bounce is known already (k_wf_closest_packet traced it: t in mult_b[p].w, hit in
// final_[p].w); on the open hit of a continuing path: org_t[p].w / this word ARE that result -- the closest-hit stage of
// the next iteration has nothing to do for it.
constexpr uint32_t WF_PRE_BIT = 1u << 27;
__device__ __forceinline__ uint32_t wf_pack_hit(int kind, int index) { return ((uint32_t)kind << 28) | (uint32_t)(index & 0xFFFFFF); }

// work item k of iteration 0 -> local pixel index, or -1 outside the image (ragged right/top tiles)
__device__ __forceinline__ int wf_pixel_of(const WfArgs& a, uint32_t k, int* x_out, int* r_out) {
    const uint32_t tile = k >> 6;
    const int in = (int)(k & 63);
    const uint32_t trow = fd_div(tile, a.fd_tiles_x);
    const int x = (int)(tile - trow * a.fd_tiles_x.d) * 8 + (in & 7);
    const int r = (int)trow * 8 + (in >> 3);
    *x_out = x, *r_out = r;
    if (x >= a.width || r >= a.local_rows) return -1;
    return r * a.width + x;
}
__device__ __forceinline__ int wf_global_row(const WfArgs& a, int r) {
    const uint32_t band = fd_div((uint32_t)r, a.fd_band);  // r / band_rows (FastDiv, rt_kernels.h)
    return (int)((band * (uint32_t)a.world + (uint32_t)a.rank) * (uint32_t)a.band_rows + ((uint32_t)r - band * (uint32_t)a.band_rows));
}
// (selected without indexing the by-value argument struct, which would force a copy of it to scratch)
__device__ __forceinline__ const uint32_t* wf_list_in(const WfArgs& a) { return (a.iter & 1) ? a.list[1] : a.list[0]; }
__device__ __forceinline__ uint32_t* wf_list_out(const WfArgs& a) { return (a.iter & 1) ? a.list[0] : a.list[1]; }
__device__ __forceinline__ uint32_t wf_active_count(const WfArgs& a) { return a.iter == 0 ? a.n0 : a.counters[4 * a.iter]; }

#ifndef RT_WF_EXACT_INLINE
#define RT_WF_EXACT_INLINE __forceinline__
#endif
#ifndef RT_WF_VOTE
#define RT_WF_VOTE 3  // a node step needs this many times as many lanes as stand at a leaf (wf_run)
#endif
#ifndef RT_WF_CHUNK
#define RT_WF_CHUNK 512  // items per run a wave reserves at once (0: every refill draws from the global head).
                         // Soup100k 1080p 14.4 -> 11.4 ms, Soup1M 4K 59.5 -> 57.5 ms with 24-lane refills
#endif
#ifndef RT_WF_NODE_REPS
#define RT_WF_NODE_REPS 4  // node steps per won vote (wf_run): on the tight tree node steps outnumber primitive tests
                           // 25:1, so re-voting after every step is mostly overhead (the loop's scalar side: ballots,
                           // counts, branches).  Soup1M 4K: 1 -> 68.4 ms, 2 -> 63.4, 4 -> 59.5, 8 -> 60.1
#endif
#ifndef RT_WF_REFILL
#define RT_WF_REFILL 24
#endif
constexpr int WF_REFILL = RT_WF_REFILL;  // idle lanes that trigger a refill.  Rays that run side by side should be neighbours
// (same light, nearby origins): when every refill drew from the global head, larger draws walked more coherently
// (16 -> 95 ms, 24 -> 83, 40 -> 78, 48 -> 83 on Soup1M 4K); with wave-owned runs (RT_WF_CHUNK) neighbours come
// by construction and smaller refills keep more lanes busy (512-item runs: 16 -> 58.2 ms, 24 -> 57.5, 40 -> 60.7)

// Draw work items for the idle lanes of the wave: one atomic per call.  Returns this lane's item or
// 0xFFFFFFFF; *exhausted becomes true (wave-uniform) once the list has been handed out completely.
__device__ __forceinline__ uint32_t wf_draw(uint32_t* head, uint32_t total, bool idle, bool* exhausted) {
    const unsigned long long mask = __ballot(idle);
    const int n_idle = __popcll(mask);
    const int lane = (int)__lane_id();
    const int leader = __ffsll(mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(head, (uint32_t)n_idle);
    base = (uint32_t)__shfl((int)base, leader, 64);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (base + (uint32_t)n_idle >= total || base >= total) *exhausted = true;
    const uint32_t k = base + (uint32_t)rank;
    return (idle && k < total) ? k : 0xFFFFFFFFu;
}

// ------------------------------------------------------------------------------------------
// The persistent, vote-scheduled traversal loop shared by the two trace stages.
//
// Every lane owns one ray (cur = its current node or leaf, lk = next primitive of the leaf, stack in
// LDS).  Rays of one wave are at unrelated points of their walks, so instead of letting the wave run
// "all node steps, then all leaves" (lanes that reached a leaf wait for the longest node chain:
// measured 25 % lane utilisation) each trip of the loop takes a vote: if at least RT_WF_VOTE times as
// many lanes stand at an inner node as at a leaf the wave executes ONE node step, otherwise ONE
// primitive test; the others wait one trip.  (On the tight tree a ray takes ~25 node steps per
// primitive test, so lanes reach leaves a few at a time: waiting for half the wave to stand at a leaf
// left a quarter of it parked; 3:1 measured best, 83 -> 80 ms on Soup1M 4K.)
// A lane whose ray ends writes its result and turns idle; once WF_REFILL lanes are idle
// they draw new rays (wf_draw).  Job supplies the ray: setup(k) -> bool (needs a traversal),
// t_lo, t_hi(), sphere(slot), tri(slot), done(), finish().
// ------------------------------------------------------------------------------------------
#ifdef RT_DIAG_WF  // diagnostic build only (tools/diag_wf.py): where the lanes of the trace stages spend their trips
__device__ unsigned long long rt_wf_diag[16];
#define WF_DIAG(i, v) diag[i] += (unsigned long long)(v)
#else
#define WF_DIAG(i, v) ((void)0)
#endif
template <class G, class Job>
__device__ __forceinline__ void wf_run(const G& g, uint32_t* head, uint32_t total, Job& job) {
#ifdef RT_DIAG_WF
    unsigned long long diag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    constexpr uint32_t NONE = 0xFFFFFFFFu, LEAF = 0x80000000u;
    uint32_t* const stk = (uint32_t*)rt_lds + g.stack_base;
    bool exhausted = false;
    uint32_t cur = NONE;
    int sp = 0, lk = 0;
#if RT_WF_CHUNK
    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform: the run of items this wave is handing out
    uint32_t chunk = (total / (gridDim.x * 4u) + 63u) & ~63u;  // short lists: at least ~4 runs per wave
    chunk = chunk < 64u ? 64u : (chunk > (uint32_t)RT_WF_CHUNK ? (uint32_t)RT_WF_CHUNK : chunk);
#endif
    for (;;) {
        const unsigned long long idle_m = __ballot(cur == NONE);
        const int n_idle = __popcll(idle_m);
        if (!exhausted && n_idle >= WF_REFILL) {
#if RT_WF_CHUNK
            // the wave owns a run of consecutive items (one atomic per run) and hands them to its idle lanes a
            // few at a time: every lane of the wave then works on neighbouring rays, however small the refill
            if (chunk_next >= chunk_end) {
                uint32_t base = 0;
                if ((int)__lane_id() == 0) base = atomicAdd(head, chunk);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                chunk_next = base, chunk_end = base + chunk < total ? base + chunk : total;
                if (base >= total) {
                    exhausted = true;
                    continue;
                }
            }
            const uint32_t left = chunk_end - chunk_next, take = (uint32_t)n_idle < left ? (uint32_t)n_idle : left;
            const uint32_t rank = (uint32_t)__popcll(idle_m & ((1ull << (int)__lane_id()) - 1ull));
            const uint32_t k = (cur == NONE && rank < take) ? chunk_next + rank : NONE;
            chunk_next += take;
#else
            const uint32_t k = wf_draw(head, total, cur == NONE, &exhausted);
#endif
            if (k != NONE && job.setup(k)) cur = job.root, sp = 0, lk = 0;  // a root holder (of the leaf tree: node 0)
            continue;
        }
        if (n_idle == 64) {
            if (exhausted) break;
            WF_DIAG(7, 1);
            continue;
        }
        const bool at_leaf = cur != NONE && (cur & LEAF);
        const int n_leaf = __popcll(__ballot(at_leaf));
        if (64 - n_idle - n_leaf >= RT_WF_VOTE * n_leaf) {
            WF_DIAG(0, 1), WF_DIAG(1, 64 - n_idle - n_leaf), WF_DIAG(2, n_idle), WF_DIAG(3, n_leaf);
#pragma unroll
            for (int rep = 0; rep < RT_WF_NODE_REPS; rep++)
            if (!(cur & LEAF)) {  // node step (NONE has the leaf bit)
                const float th = job.t_hi();
                float e0, e1;
                uint32_t r0, r1;
                bool h0, h1;
                if (job.tight) {  // conservative filter (rt_bvh.h build_tight): a 32-byte node, boxes on the 16-bit grid
                    const uint4 q0 = g.tight_node(cur, 0), q1 = g.tight_node(cur, 1);
                    r0 = q1.z, r1 = q1.w;
                    const bool nc0 = G::unbounded && rtb::ref_nocull(r0), nc1 = G::unbounded && rtb::ref_nocull(r1);
                    h0 = SlabHitQ<0>(job.br, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, job.t_lo, th, &e0, nc0);
                    h1 = SlabHitQ<1>(job.br, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, job.t_lo, th, &e1, nc1);
                    // (keeps the refs part of THIS fetch: left alone, the compiler merges the two branches' ref loads
                    // into one load behind the branch, a second dependent memory round trip per step -- 55 -> 66 ms)
                    asm volatile("" : "+v"(r0), "+v"(r1));
                } else {  // the reference's test on the reference's leaf boxes
                    const float4 q0 = g.bvh_node(cur, 0), q1 = g.bvh_node(cur, 1), q2 = g.bvh_node(cur, 2);
                    const uint2 refs = g.bvh_refs(cur);
                    r0 = refs.x, r1 = refs.y;
                    const bool nc0 = G::unbounded && rtb::ref_nocull(r0), nc1 = G::unbounded && rtb::ref_nocull(r1);
                    // (the scalar tests on the unpacked boxes: the packed pair forms cost these kernels, which run at
                    // an 80-VGPR cap, twice the spills -- Soup1M 4K 57 -> 100 ms)
                    const float4 lo0 = make_float4(q0.x, q0.z, q1.x, 0), hi0 = make_float4(q1.z, q2.x, q2.z, 0);
                    const float4 lo1 = make_float4(q0.y, q0.w, q1.y, 0), hi1 = make_float4(q1.w, q2.y, q2.w, 0);
                    h0 = IntersectBoundingBoxCull(job.br, lo0, hi0, job.t_lo, th, &e0, nc0);
                    h1 = IntersectBoundingBoxCull(job.br, lo1, hi1, job.t_lo, th, &e1, nc1);
                    asm volatile("" : "+v"(r0), "+v"(r1));
                }
                h0 &= r0 != LEAF, h1 &= r1 != LEAF;
                if (h0 & h1) {
                    const bool swap = e1 < e0;
                    stk[64 * sp++] = swap ? r0 : r1;
                    cur = swap ? r1 : r0;
                } else if (h0 | h1) {
                    cur = h0 ? r0 : r1;
                } else if (sp) {
                    cur = stk[64 * --sp];
                } else {
                    job.finish();
                    cur = NONE;
                }
            }
        } else if (WF_DIAG(4, 1), WF_DIAG(5, n_leaf), WF_DIAG(6, n_idle), at_leaf) {  // one primitive of the leaf
            const int kind = (cur >> 29) & 3;
            const int count = (cur >> 24) & 31;
            const int slot = (int)(cur & 0xFFFFFF) + lk;
            if (kind == rtf::KIND_SPHERE)
                job.sphere(slot);
            else
                job.tri(slot);
            lk++;
            const bool stop = job.done();
            if (stop || lk >= count) {
                lk = 0;
                if (!stop && sp) {
                    cur = stk[64 * --sp];
                } else {
                    job.finish();
                    cur = NONE;
                }
            }
        }
    }
#ifdef RT_DIAG_WF
    if ((int)__lane_id() == 0)
        for (int k = 0; k < 8; k++) atomicAdd(&rt_wf_diag[Job::DIAG_BASE + k], diag[k]);
#endif
}

// ------------------------------------------------------------------------------------------
// closest hit of every active path (FindClosestIntersection, raytracer_fs.glsl:241-297)
// ------------------------------------------------------------------------------------------
template <class G>
struct WfClosestJob {
    static constexpr int DIAG_BASE = 0;
    const WfArgs& a;
    const G& g;
    float t_lo;  // minDistance: 0 for the primary ray, epsilon after a bounce
    int p;
    V3 o, d;
    BoxRay br;
    Closest h;
    bool tight;  // this ray walks the tight tree: hits count only if their reference leaf box passes
    uint32_t root;  // node the walk starts at (tight_root)
    __device__ __forceinline__ float t_hi() const { return h.t; }
    __device__ __forceinline__ bool done() const { return false; }
    __device__ __forceinline__ bool leaf_passes(const int32_t* slot_leaf, int slot) const {
        if (!tight) return true;
        const int leaf = slot_leaf[slot];
        // (the tight ray lives on the tree's grid: the reference's test takes the ray as the shader forms it)
        return IntersectBoundingBox(make_box_ray(o, d), g.node_lo(leaf), g.node_hi(leaf));
    }
    __device__ __forceinline__ void sphere(int slot) { sphere_is(slot, g.sphere(slot)); }
    __device__ __forceinline__ void tri(int slot) { tri_is(slot, load_tri(g, slot)); }
    // (the record already fetched: k_wf_closest_packet reads it once for the whole wave)
    __device__ __forceinline__ void sphere_is(int slot, float4 s) {
        const float t = IntersectSphereT(o, d, v3(s.x, s.y, s.z), s.w);
        if (closer(t, 0, slot, t_lo, h) && leaf_passes(a.scene.sphere_slot_leaf, slot)) h.t = t, h.kind = 0, h.index = slot;
    }
    __device__ __forceinline__ void tri_is(int slot, const TriG& tg) {
        const float t = IntersectTriangleWindowed(o, d, tg.A, tg.B, tg.C, tg.N,
                                                  [&](float tt) { return closer(tt, 2, slot, t_lo, h); });
        if (closer(t, 2, slot, t_lo, h) && leaf_passes(a.scene.tri_slot_leaf, slot)) h.t = t, h.kind = 2, h.index = slot;
    }
    __device__ __forceinline__ void finish() const {
        if (a.iter == 0) {
            a.org_t[p] = make_float4(o.x, o.y, o.z, h.t);
            a.dir_hit[p] = make_float4(d.x, d.y, d.z, __uint_as_float(wf_pack_hit(h.kind, h.index)));
        } else {
            ((float*)(a.org_t + p))[3] = h.t;
            ((uint32_t*)(a.dir_hit + p))[3] = wf_pack_hit(h.kind, h.index);
        }
    }
    __device__ __forceinline__ bool setup(uint32_t k) {
        if (a.iter == 0) {
            int x, r;
            p = wf_pixel_of(a, k, &x, &r);
            if (p < 0) return false;
            o = v3(a.cam[0], a.cam[1], a.cam[2]);
            d = primary_direction(a.cam, (float)x + 0.5f, (float)wf_global_row(a, r) + 0.5f);
        } else {
            p = (int)wf_list_in(a)[k];
            const float4 q0 = a.org_t[p], q1 = a.dir_hit[p];
            if (__float_as_uint(q1.w) & WF_PRE_BIT) return false;  // traced with the primary packet (k_wf_closest_packet)
            o = v3(q0.x, q0.y, q0.z), d = v3(q1.x, q1.y, q1.z);
        }
        if (begin()) return true;
        // zero direction component etc.: the exact tree, exhaustively (rare)
        walk_exact();
        finish();
        return false;
    }
    __device__ RT_WF_EXACT_INLINE void walk_exact() {
        walk<false>(
            g, a.scene.n_nodes, br, 3, false, nullptr,
            [&](int slot) {
                sphere(slot);
                return false;
            },
            [&](int slot) {
                tri(slot);
                return false;
            });
    }
    // the ray (o, d, t_lo) against the planes; true: it goes on to walk a tree (tight / root say which)
    __device__ __forceinline__ bool begin() {
        const DevScene& S = a.scene;
        br = make_box_ray_cull(o, d, v3(S.extent[0], S.extent[1], S.extent[2]), S.cull_scale);
        h = fresh_closest();
        for (int q = 0; q < S.n_planes; q++) {
            const float4 pp = S.plane_geom[2 * q], nn = S.plane_geom[2 * q + 1];
            const float t = IntersectPlane(o, d, v3(pp.x, pp.y, pp.z), v3(nn.x, nn.y, nn.z));
            if (closer(t, 1, q, t_lo, h)) h.t = t, h.kind = 1, h.index = q;
        }
        tight = false, root = 0u;
        if (ray_is_regular(br) && S.n_nodes > 0) {
            tight = S.tight != nullptr && make_tight_ray(br, v3(S.extent[0], S.extent[1], S.extent[2]), S.tight_e, S.tight_base, S.tight_step);
            if (tight) root = tight_root(S, o);
            return true;
        }
        return false;
    }
};

#ifndef RT_WF_WAVES_PER_EU
#define RT_WF_WAVES_PER_EU 6  // 80 VGPRs (20 B of spill): Soup1M 4K 148.8 -> 140.3 ms; 8 waves (64 VGPRs) spill 88 B: 375 ms
                              // (round 2's kernels, Soup1M 4K: 6 waves 49.8 ms; 5 waves / 96 VGPRs 52.7; 4 waves / no spills
                              // 59.1; 7 waves / 72 VGPRs 64.9)
#endif
template <bool UNB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_WF_WAVES_PER_EU, 8))) void k_wf_closest(const WfArgs a) {
    GeomBvhT<UNB> g{a.scene.geom, a.scene.tri_off, a.scene.sph_off, a.scene.bvh, (int)(threadIdx.x & 63)};
    g.tight = a.scene.tight;
    WfClosestJob<GeomBvhT<UNB>> job{a, g, a.iter == 0 ? 0.0f : RT_EPSILON};
    wf_run(g, a.counters + 4 * a.iter + 1, wf_active_count(a), job);
}

// The primary rays of an 8x8 tile as ONE packet (iteration 0 on the tight tree).  Per-lane walks cost the same
// whether their rays are neighbours or not: every lane fetches its own node (a gather) and the wave advances at the
// pace of its votes.  The 64 primary rays of a tile share their origin and differ by a fraction of a degree, so
// here the WAVE walks the tree once -- one current node, one stack (lane i of a register = entry i), node and
// primitive addresses wave-uniform -- and every lane tests the boxes and primitives of the shared walk against its own
// ray and its own closest hit so far.  A child is entered if any lane's ray meets it; of two, the one more lanes
// enter first.  Each lane thereby tests a SUPERSET of the candidates of its own walk; results cannot differ: a hit
// counts by the exact rule (primitive test + the reference's test of the slot's leaf box, leaf_passes), and the
// closest hit is the minimum under closer()'s total order, whatever the order of the tests.
// Lanes whose ray may not walk the tight tree (a direction component below 1e-30) take the exact tree exhaustively;
// the host only launches this kernel for a camera inside E_build.
// a 16-byte record at a wave-uniform address, through the scalar cache (read only: the geometry image is never
// written while a render runs)
__device__ __forceinline__ uint4 wf_uniform_load(const void* p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) u32x4* const_ptr;
    const u32x4 v = *(const_ptr)(uintptr_t)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 wf_uniform_load_f4(const void* p) {
    const uint4 u = wf_uniform_load(p);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
// the shared walk for the lanes `mine` of the wave (rays that may walk the tight tree), once per root holder
template <bool UNB, class Job>
__device__ __forceinline__ void wf_packet_closest(const WfArgs& a, const GeomBvhT<UNB>& g, Job& job, bool mine) {
    constexpr uint32_t LEAF = 0x80000000u;
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t root = pass ? a.scene.tight_near_root : 0u;
        if (pass && root == 0u) break;
        const bool pk = mine && job.root == root;
        if (!__ballot(pk)) continue;
        int stk = 0, sp = 0;
        uint32_t cur = root;
        for (;;) {
            if (!(cur & LEAF)) {
                const uint4* const node = g.tight + 2 * (size_t)(UNB ? (cur & rtb::INDEX_MASK) : cur);
                const uint4 q0 = wf_uniform_load(node), q1 = wf_uniform_load(node + 1);
                const uint32_t r0 = q1.z, r1 = q1.w;
                const bool nc0 = UNB && rtb::ref_nocull(r0), nc1 = UNB && rtb::ref_nocull(r1);
                float e0, e1;
                const float th = job.h.t;
                const bool h0 = pk & (r0 != LEAF) & SlabHitQ<0>(job.br, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, job.t_lo, th, &e0, nc0);
                const bool h1 = pk & (r1 != LEAF) & SlabHitQ<1>(job.br, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, job.t_lo, th, &e1, nc1);
                const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
                if (m0 && m1) {
                    const int n0 = __popcll(m0 & ~m1) + __popcll(__ballot(h0 & h1 & (e0 <= e1)));
                    const int n1 = __popcll(m1 & ~m0) + __popcll(__ballot(h0 & h1 & (e1 < e0)));
                    const bool first0 = n0 >= n1;
                    stk = (int)__lane_id() == sp ? (int)(first0 ? r1 : r0) : stk;
                    sp++;
                    cur = first0 ? r0 : r1;
                    continue;
                }
                if (m0 | m1) {
                    cur = m0 ? r0 : r1;
                    continue;
                }
            } else {
                const int kind = (cur >> 29) & 3, count = (cur >> 24) & 31, first = (int)(cur & 0xFFFFFF);
                for (int k = 0; k < count; k++) {
                    const int slot = first + k;
                    if (kind == rtf::KIND_SPHERE) {
                        const float4 sp4 = wf_uniform_load_f4(g.base + g.sph_off + slot);
                        if (pk) job.sphere_is(slot, sp4);
                    } else {
                        const float4* const rec = g.base + g.tri_off + 3 * (size_t)slot;
                        const float4 t0 = wf_uniform_load_f4(rec), t1 = wf_uniform_load_f4(rec + 1), t2 = wf_uniform_load_f4(rec + 2);
                        TriG tg;  // (load_tri's unpacking)
                        tg.A = v3(t0.x, t0.y, t0.z), tg.B = v3(t0.w, t1.x, t1.y), tg.C = v3(t1.z, t1.w, t2.x), tg.N = v3(t2.y, t2.z, t2.w);
                        if (pk) job.tri_is(slot, tg);
                    }
                }
            }
            if (sp == 0) break;
            cur = (uint32_t)__builtin_amdgcn_readlane(stk, --sp);
        }
    }
}

template <bool UNB>
__global__ __launch_bounds__(64) void k_wf_closest_packet(const WfArgs a) {
    GeomBvhT<UNB> g{a.scene.geom, a.scene.tri_off, a.scene.sph_off, a.scene.bvh, 0};
    g.tight = a.scene.tight;
    WfClosestJob<GeomBvhT<UNB>> job{a, g, 0.0f};
    const bool traced = job.setup(blockIdx.x * 64u + threadIdx.x);  // false: no pixel, or done (zero direction component)
    if (traced && !job.tight) job.walk_exact();
    wf_packet_closest<UNB>(a, g, job, traced && job.tight);
    if (traced) job.finish();
    // ---- the mirror bounce of the pixels that hit ONE pure-specular plane (the reference scenes' mirror floors): the
    // reflected rays are the primary rays of the camera's mirror image -- as good a packet as the first.  Their closest
    // hits are found here and handed to k_wf_shade through WF_PRE_BIT; the next iteration's closest-hit stage skips them.
    if (a.max_bounces < 1) return;
    const bool on_plane = job.p >= 0 && job.h.kind == 1;
    const unsigned long long plane_m = __ballot(on_plane);
    if (!plane_m) return;
    const int plane = __builtin_amdgcn_readlane(job.h.index, __ffsll(plane_m) - 1);
    Closest hp = fresh_closest();
    hp.kind = 1, hp.index = plane;  // (wave-uniform: the surface's record is one scalar fetch)
    const Surface sf = resolve_surface(a.scene, hp);
    if (!sf.pure) return;
    const bool bounce = on_plane && job.h.index == plane;
    WfClosestJob<GeomBvhT<UNB>> next{a, g, RT_EPSILON};
    next.p = job.p;
    next.o = job.o + job.h.t * job.d;           // as k_wf_shade forms the continuing ray
    next.d = mirror_direction(sf, job.d);
    const bool walks = bounce && next.begin() && next.tight;  // (other rays: the next iteration's per-lane stage)
    if (!__ballot(walks)) return;
    wf_packet_closest<UNB>(a, g, next, walks);
    if (walks) {
        ((float*)(a.mult_b + job.p))[3] = next.h.t;
        ((uint32_t*)(a.final_ + job.p))[3] = wf_pack_hit(next.h.kind, next.h.index);
        ((uint32_t*)(a.dir_hit + job.p))[3] = wf_pack_hit(job.h.kind, job.h.index) | WF_PRE_BIT;
    }
}

// ------------------------------------------------------------------------------------------
// shadow rays of every active path (ObjectInWayOfLight, raytracer_fs.glsl:200-240)
// Work items are ordered (group of 64 consecutive active paths, light, path in group): the lanes
// that fetch together get neighbouring hit points and the SAME light, i.e. near-parallel rays that
// walk the same nodes (L1/L2 hits), not one point's rays to L different lights.
// ------------------------------------------------------------------------------------------
template <class G>
struct WfAnyJob {
    static constexpr int DIAG_BASE = 8;
    const WfArgs& a;
    const G& g;
    uint32_t n_active;
    float t_lo;  // epsilon
    uint32_t item;
    V3 o, d;
    float dist;
    BoxRay br;
    bool blocked;
    bool tight;  // as in WfClosestJob
    uint32_t root;
    __device__ __forceinline__ float t_hi() const { return dist; }
    __device__ __forceinline__ bool done() const { return blocked; }
    __device__ __forceinline__ bool in_range(float t) const { return t > RT_EPSILON && t < dist; }
    __device__ __forceinline__ bool leaf_passes(const int32_t* slot_leaf, int slot) const {
        if (!tight) return true;
        const int leaf = slot_leaf[slot];
        // (the tight ray lives on the tree's grid: the reference's test takes the ray as the shader forms it)
        return IntersectBoundingBox(make_box_ray(o, d), g.node_lo(leaf), g.node_hi(leaf));
    }
    __device__ __forceinline__ void sphere(int slot) {
        const float4 s = g.sphere(slot);
        if (in_range(IntersectSphereT(o, d, v3(s.x, s.y, s.z), s.w)) && leaf_passes(a.scene.sphere_slot_leaf, slot))
            blocked = true;
    }
    __device__ __forceinline__ void tri(int slot) {
        const TriG tg = load_tri(g, slot);
        if (in_range(IntersectTriangleWindowed(o, d, tg.A, tg.B, tg.C, tg.N, [&](float t) { return in_range(t); })) &&
            leaf_passes(a.scene.tri_slot_leaf, slot))
            blocked = true;
    }
    __device__ __forceinline__ void finish() const { a.blocked[item] = blocked ? 1 : 0; }
    __device__ __forceinline__ bool setup(uint32_t k) {
        const DevScene& S = a.scene;
        const uint32_t L = (uint32_t)a.n_lights;
        const uint32_t grp = k / (64u * L), in = k - grp * (64u * L);
        const uint32_t l = in >> 6, j = grp * 64u + (in & 63u);
        if (j >= n_active) return false;
        int p;
        if (a.iter == 0) {
            int x, r;
            p = wf_pixel_of(a, j, &x, &r);
            if (p < 0) return false;
        } else {
            p = (int)wf_list_in(a)[j];
        }
        const float4 q0 = a.org_t[p], q1 = a.dir_hit[p];
        const uint32_t hit = __float_as_uint(q1.w);
        const int kind = (int)(hit >> 28), index = (int)(hit & 0xFFFFFF);
        if (kind == 3) return false;
        const V3 rayOrigin = v3(q0.x, q0.y, q0.z), rayDirection = v3(q1.x, q1.y, q1.z);
        const V3 hitPos = rayOrigin + q0.w * rayDirection;
        Closest h;
        h.t = q0.w, h.kind = kind, h.index = index, h.n = v3(0, 0, 0);
        if (kind == 0) {  // as k_wf_shade forms it
            const float4 sp = S.geom[S.sph_off + index];
            h.n = normalize(hitPos - v3(sp.x, sp.y, sp.z));
        }
        const Surface sf = resolve_surface(S, h);
        const bool mirror = sf.pure && a.iter < a.max_bounces;
        // a mirror bounce only casts shadow rays for a non-black diffuse (raytracer_fs.glsl:340)
        if (mirror && !(sf.diffuse.x != 0 || sf.diffuse.y != 0 || sf.diffuse.z != 0)) return false;
        const LightRay lr = light_ray(S, (int)l, hitPos);
        o = hitPos, d = lr.L, dist = lr.dist;
        item = j * L + l;
        if (light_term_is_zero(sf, mirror, rayDirection, lr)) {  // shaded as if blocked: the term adds nothing
            blocked = true;
            finish();
            return false;
        }
        br = make_box_ray_cull(o, d, v3(S.extent[0], S.extent[1], S.extent[2]), S.cull_scale);
        blocked = false;
        for (int q = 0; q < S.n_planes && !blocked; q++) {
            const float4 pp = S.plane_geom[2 * q], nn = S.plane_geom[2 * q + 1];
            if (in_range(IntersectPlane(o, d, v3(pp.x, pp.y, pp.z), v3(nn.x, nn.y, nn.z)))) blocked = true;
        }
        tight = false, root = 0u;
        if (!blocked && ray_is_regular(br) && S.n_nodes > 0) {
            tight = S.tight != nullptr && make_tight_ray(br, v3(S.extent[0], S.extent[1], S.extent[2]), S.tight_e, S.tight_base, S.tight_step);
            if (tight) root = tight_root(S, o);
            return true;
        }
        if (!blocked) walk_exact();
        finish();
        return false;
    }
    __device__ RT_WF_EXACT_INLINE void walk_exact() {
        walk<false>(
            g, a.scene.n_nodes, br, 3, false, nullptr,
            [&](int slot) {
                sphere(slot);
                return blocked;
            },
            [&](int slot) {
                tri(slot);
                return blocked;
            });
    }
};

template <bool UNB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_WF_WAVES_PER_EU, 8))) void k_wf_any(const WfArgs a) {
    GeomBvhT<UNB> g{a.scene.geom, a.scene.tri_off, a.scene.sph_off, a.scene.bvh, (int)(threadIdx.x & 63)};
    g.tight = a.scene.tight;
    const uint32_t n_active = wf_active_count(a);
    WfAnyJob<GeomBvhT<UNB>> job{a, g, n_active, RT_EPSILON};
    // the last group of 64 paths is padded: setup() skips j >= n_active
    wf_run(g, a.counters + 4 * a.iter + 2, ((n_active + 63u) / 64u) * 64u * (uint32_t)a.n_lights, job);
}

// ------------------------------------------------------------------------------------------
// shading of every active path for this bounce (raytracer_fs.glsl:328-424), one thread per path
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wf_shade(const WfArgs a) {
    const DevScene& S = a.scene;
    const uint32_t total = wf_active_count(a);
    const uint32_t L = (uint32_t)a.n_lights;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        int p;
        if (a.iter == 0) {
            int x, r;
            p = wf_pixel_of(a, j, &x, &r);
        } else {
            p = (int)wf_list_in(a)[j];
        }
        bool continues = false;
        uint32_t cell = 0;
        if (p >= 0) {
            const float4 q0 = a.org_t[p], q1 = a.dir_hit[p];
            const V3 rayOrigin = v3(q0.x, q0.y, q0.z), rayDirection = v3(q1.x, q1.y, q1.z);
            V3 mult = v3(1, 1, 1), finalColor = v3(0, 0, 0);
            if (a.iter > 0) {
                const float4 m = a.mult_b[p], f = a.final_[p];
                mult = v3(m.x, m.y, m.z), finalColor = v3(f.x, f.y, f.z);
            }
            const uint32_t hit = __float_as_uint(q1.w);
            Closest h;
            h.t = q0.w, h.kind = (int)(hit >> 28), h.index = (int)(hit & 0xFFFFFF), h.n = v3(0, 0, 0);
            if (h.kind == 0) {  // the normal IntersectSphere forms at the accepted t (raytracer_fs.glsl:106-111)
                const float4 s = S.geom[S.sph_off + h.index];
                h.n = normalize((rayOrigin + h.t * rayDirection) - v3(s.x, s.y, s.z));
            }
            const Surface sf = (h.kind != 3) ? resolve_surface(S, h) : no_surface();
            if (a.iter == 0 && a.primary_id) a.primary_id[p] = sf.id;
            const unsigned char* blk = a.blocked + (size_t)j * L;
            auto in_shadow = [&](int l, const LightRay&) { return blk[l] != 0; };
            if (sf.pure && a.iter < a.max_bounces) {
                const V3 hitPos = rayOrigin + h.t * rayDirection;
                if (sf.diffuse.x != 0 || sf.diffuse.y != 0 || sf.diffuse.z != 0)
                    finalColor = finalColor + mirror_diffuse_term(S, sf, hitPos, in_shadow) * mult;
                mult = mult * sf.specular;
                const V3 refl = mirror_direction(sf, rayDirection);
                float next_t = RT_FLT_BIG;
                uint32_t next_hit = WF_NO_HIT;
                if (a.iter == 0 && (hit & WF_PRE_BIT))  // k_wf_closest_packet traced this bounce already
                    next_t = ((const float*)(a.mult_b + p))[3], next_hit = ((const uint32_t*)(a.final_ + p))[3] | WF_PRE_BIT;
                a.org_t[p] = make_float4(hitPos.x, hitPos.y, hitPos.z, next_t);
                a.dir_hit[p] = make_float4(refl.x, refl.y, refl.z, __uint_as_float(next_hit));
                a.mult_b[p] = make_float4(mult.x, mult.y, mult.z, 0);
                a.final_[p] = make_float4(finalColor.x, finalColor.y, finalColor.z, 0);
                continues = true;
                if (wf_binned(a, a.iter + 1)) cell = (wf_cell(a, hitPos) << 3) | (refl.x < 0 ? 1u : 0u) | (refl.y < 0 ? 2u : 0u) | (refl.z < 0 ? 4u : 0u);
            } else {
                PixelOut out;
                out.primary_id = 0;
                out.bounces = a.iter;
                out.final_id = sf.id;
                if (h.t == RT_FLT_BIG) {
                    out.rgb = finalColor + v3(0.5f, 0.6f, 1.0f) * mult;
                } else {
                    const V3 hitPos = rayOrigin + h.t * rayDirection;
                    out.rgb = finalColor + phong_term(S, sf, hitPos, rayDirection, in_shadow) * mult;
                }
                const size_t i = (size_t)p;
                if (a.rgb) a.rgb[3 * i] = out.rgb.x, a.rgb[3 * i + 1] = out.rgb.y, a.rgb[3 * i + 2] = out.rgb.z;
                if (a.rgba8)
                    a.rgba8[i] = quantise(out.rgb.x) | (quantise(out.rgb.y) << 8) | (quantise(out.rgb.z) << 16) | 0xFF000000u;
                if (a.final_id) a.final_id[i] = out.final_id;
                if (a.bounces) a.bounces[i] = out.bounces;
            }
        }
        // append the continuing paths to the next list: ballot + mbcnt compaction, one atomic per wave
        const unsigned long long mask = __ballot(continues);
        if (mask) {
            const int lane = (int)__lane_id(), leader = __ffsll(mask) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(a.counters + 4 * (a.iter + 1), (uint32_t)__popcll(mask));
            base = (uint32_t)__shfl((int)base, leader, 64);
            const uint32_t at = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (wf_binned(a, a.iter + 1)) {
                if (continues) a.list_tmp[at] = (uint32_t)p, a.keys[at] = cell;
                wf_wave_add(a.bins, cell, continues);
            } else if (continues) {
                wf_list_out(a)[at] = (uint32_t)p;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Binning of the continuing paths (not in the reference, changes no pixel: every stage treats the paths of
// a list independently).  The list k_wf_shade appends to is in compaction order: after the first bounce of
// a scene of small mirrors neighbouring entries start anywhere in the scene.  A counting sort by the cell
// of the new origin (32^3 cells of the root box in Morton order) makes the entries a wave draws together
// neighbours in space: their shadow rays to one light are near-parallel again, and their bounce rays at
// least start in the same leaves.  k_wf_shade counts the cells as it appends; one workgroup turns the
// counts into offsets; the scatter hands out places with one returning atomic per path.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_wf_bin_scan(const WfArgs a) {
    __shared__ uint32_t part[1024];
    constexpr int PER = WF_BINS / 1024;
    const int t = (int)threadIdx.x;
    uint32_t v[PER], sum = 0;
    for (int k = 0; k < PER; k++) v[k] = a.bins[t * PER + k], sum += v[k];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan of the 1024 partial sums
        const uint32_t add = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int k = 0; k < PER; k++) a.bins[t * PER + k] = run, run += v[k];
}
__global__ __launch_bounds__(256) void k_wf_bin_scatter(const WfArgs a) {
    const uint32_t total = a.counters[4 * (a.iter + 1)];
    uint32_t* const out = wf_list_out(a);
    // (whole waves stay in the loop: the places are handed out per wave)
    for (uint32_t j0 = (blockIdx.x * blockDim.x + (threadIdx.x & ~63u)); j0 < total; j0 += gridDim.x * blockDim.x) {
        const uint32_t j = j0 + (threadIdx.x & 63u);
        const bool in = j < total;
        const uint32_t key = in ? a.keys[j] : 0u;
        const uint32_t place = wf_wave_add(a.bins, key, in);
        if (in) out[place] = a.list_tmp[j];
    }
}

}  // namespace rtd
