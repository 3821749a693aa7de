// rt_gpu_bvh.h -- the binned-SAH builds of rt_bvh.h (re-treed leaves, tight tree) on the GPU, bit-identical to the
// host builder's node arrays.
//
// The host builder (rtb::detail::Builder) is a recursion: centroid bounds, 16 bins per axis, the cheapest split between
// bins (rtb::split_axis), partition, left subtree, right subtree; a subtree over m items owns m - 1 consecutive nodes
// (root, left subtree, right subtree), so where a node lives follows from the splits above it alone.  Everything a
// split depends on is independent of the ORDER of the items of the node: box unions are min/max (zeros canonical),
// bin weights are small integers, ties between candidates go by (axis, bin), and the two places where the host used
// to look at the order (two items with coincident centroids; equal keys of the median split) go by the items' refs
// (rtb::median_before).  So the device may partition with atomics and still build the same tree:
//
//   large nodes (> SMALL_MAX items), level by level over the whole array, a workgroup per 256-item chunk of ONE node:
//     k_bounds     centroid bounds of every node of the level    (wave reduction, six atomics per chunk)
//     k_scale      bin scales; clears the node's bins
//     k_bin        16 bins x 3 axes per node: {box, weight, count, never-cull count}; LDS bins per chunk, then atomics
//     k_split      one thread per node: rtb::split_axis over the bins -- the host's function, binary64 without
//                  contraction --, writes the node (child boxes are unions of bins, child refs follow from the counts),
//                  creates the children's records: large ones for the next level, small ones for k_small
//     k_partition  moves the items to their side (two atomics per chunk); an item that ends up alone writes its ref
//   small subtrees (<= 64 items): one wave each, one item per lane, the whole subtree in LDS, every node of a
//     sub-level at once (k_small): segmented bins by LDS atomics, the same split_axis, ballots for the partition;
//     the median split (coincident centroids) by ranking (key, ref) within the segment.
//   A LARGE node without a valid split (thousands of coincident centroids) makes the device decline; the host builds.
//
// Replaces the producer side of /root/reference/Helper classes/RTree.cs:180-377 for the derived trees only: the
// reference's own tree still arrives serialised and is flattened as before.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rt_bvh.h"

#pragma clang fp contract(off)

namespace rtgb {

using rtb::Node;
using rtb::detail::BINS;
using rtb::detail::Box;
using rtb::detail::Item;

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int SMALL_MAX = 64;  // subtrees of at most this many items: one wave (k_small)
constexpr int CHUNK = 256;
constexpr int BIN_WORDS = 9;   // lo.xyz, hi.xyz (ordered), weight, count, never-cull count

// order-preserving map binary32 -> uint32 (no NaN among box bounds; zeros are canonical, rt_bvh.h make_leaf_items)
__device__ __forceinline__ uint32_t f2o(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

struct BigNode {
    uint32_t b, e, base;          // items [b, e), node index
    uint32_t cb_lo[3], cb_hi[3];  // centroid bounds, ordered
    float cb_lo_f[3];
    double scale[3];
    int axis, bin;                // the chosen split
    uint32_t cursor[2];           // next free place of each side (k_partition)
    uint32_t leaf_side[2];        // the child is a single item: it writes its ref into the node itself
    uint32_t child_big[2];        // the child's record in the next level's list (NONE: small or a leaf): k_partition
                                  // collects its centroid bounds on the way
};
struct Chunk {
    uint32_t node, first, count;
};
struct SmallWork {
    uint32_t b, e, base, level, parity;
};
struct State {
    uint32_t n_big_next, n_chunks_next, n_small, declined;
    int levels;
};
struct HolderArgs {  // the three groups under the root holder(s) (rtb::detail::build_items)
    uint32_t n[3], base[3];
    Item single[3];  // the item of a group of one
    uint32_t near_holder;  // index of the second holder, 0: none
};

__device__ __forceinline__ void new_big(BigNode* nd, uint32_t b, uint32_t e, uint32_t base) {
    nd->b = b, nd->e = e, nd->base = base;
    for (int k = 0; k < 3; k++) nd->cb_lo[k] = 0xFFFFFFFFu, nd->cb_hi[k] = 0u;
}

// centroid bounds of the group roots (the levels below get theirs from k_partition)
__global__ __launch_bounds__(CHUNK) void k_bounds(const Item* items, const Chunk* chunks, BigNode* nodes) {
    __shared__ uint32_t cb[6];
    const Chunk ch = chunks[blockIdx.x];
    if (threadIdx.x < 6) cb[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    if (threadIdx.x < ch.count) {
        const Item it = items[ch.first + threadIdx.x];
        for (int k = 0; k < 3; k++) {
            const uint32_t c = f2o(0.5f * it.lo[k] + 0.5f * it.hi[k]);
            atomicMin(&cb[k], c), atomicMax(&cb[3 + k], c);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicMin(&nodes[ch.node].cb_lo[threadIdx.x], cb[threadIdx.x]);
    else if (threadIdx.x < 6)
        atomicMax(&nodes[ch.node].cb_hi[threadIdx.x - 3], cb[threadIdx.x]);
}

__device__ __forceinline__ void clear_bins(uint32_t* bins, int i) {  // word i of a node's 3 x BINS x BIN_WORDS
    const int f = i % BIN_WORDS;
    bins[i] = f < 3 ? 0xFFFFFFFFu : 0u;
}

__global__ __launch_bounds__(256) void k_scale(BigNode* nodes, uint32_t* bins) {
    BigNode& nd = nodes[blockIdx.x];
    if (threadIdx.x < 3) {
        const int k = (int)threadIdx.x;
        const float lo = o2f(nd.cb_lo[k]), hi = o2f(nd.cb_hi[k]);
        nd.cb_lo_f[k] = lo;
        nd.scale[k] = rtb::detail::bin_scale(lo, hi);
    }
    uint32_t* const mine = bins + (size_t)blockIdx.x * (3 * BINS * BIN_WORDS);
    for (int i = (int)threadIdx.x; i < 3 * BINS * BIN_WORDS; i += 256) clear_bins(mine, i);
}

// adds one item to bin (ax, k) of a table of 3 x BINS x BIN_WORDS words (LDS or global)
__device__ __forceinline__ void bin_add(uint32_t* table, int ax, int k, const Item& it) {
    uint32_t* const w = table + (ax * BINS + k) * BIN_WORDS;
    for (int c = 0; c < 3; c++) atomicMin(w + c, f2o(it.lo[c])), atomicMax(w + 3 + c, f2o(it.hi[c]));
    atomicAdd(w + 6, (uint32_t)it.weight);
    atomicAdd(w + 7, 1u);
    if (rtb::ref_nocull(it.ref)) atomicAdd(w + 8, 1u);
}

__global__ __launch_bounds__(CHUNK) void k_bin(const Item* items, const Chunk* chunks, const BigNode* nodes, uint32_t* bins) {
    __shared__ uint32_t lb[3 * BINS * BIN_WORDS];
    const Chunk ch = chunks[blockIdx.x];
    const BigNode& nd = nodes[ch.node];
    for (int i = (int)threadIdx.x; i < 3 * BINS * BIN_WORDS; i += CHUNK) clear_bins(lb, i);
    __syncthreads();
    if (threadIdx.x < ch.count) {
        const Item it = items[ch.first + threadIdx.x];
        for (int ax = 0; ax < 3; ax++) {
            const double sc = nd.scale[ax];
            if (!(sc > 0)) continue;
            bin_add(lb, ax, rtb::detail::bin_index(it.lo[ax], it.hi[ax], nd.cb_lo_f[ax], sc), it);
        }
    }
    __syncthreads();
    uint32_t* const mine = bins + (size_t)ch.node * (3 * BINS * BIN_WORDS);
    for (int i = (int)threadIdx.x; i < 3 * BINS * BIN_WORDS; i += CHUNK) {
        const int f = i % BIN_WORDS;
        if (lb[i - f + 7] == 0) continue;  // empty bin
        if (f < 3)
            atomicMin(mine + i, lb[i]);
        else if (f < 6)
            atomicMax(mine + i, lb[i]);
        else if (lb[i])
            atomicAdd(mine + i, lb[i]);
    }
}

// reads bin (ax, k) of a table as the host's Bins hold it
__device__ __forceinline__ void bin_read(const uint32_t* table, int ax, int k, Box* bb, double* bw, size_t* bn) {
    const uint32_t* const w = table + (ax * BINS + k) * BIN_WORDS;
    for (int c = 0; c < 3; c++) bb->lo[c] = o2f(w[c]), bb->hi[c] = o2f(w[3 + c]);
    *bw = (double)w[6], *bn = (size_t)w[7];
}

__device__ __forceinline__ void set_child_box(Node* nd, int c, const Box& b) {
    for (int k = 0; k < 3; k++) nd->lo[k][c] = b.lo[k], nd->hi[k][c] = b.hi[k];
}

// four lanes per node: lanes 0..2 search one axis each (rtb::split_axis), lane 0 takes the cheapest in axis order
// -- what the host's sequential search keeps -- and does the rest
__global__ __launch_bounds__(64) void k_split(BigNode* nodes, uint32_t n_nodes, const uint32_t* bins, int level, Node* out,
                                              BigNode* next, Chunk* next_chunks, SmallWork* small, uint32_t next_parity, State* st) {
    const uint32_t i = (blockIdx.x * 64 + threadIdx.x) >> 2;
    const int sub = (int)(threadIdx.x & 3), lane = (int)(threadIdx.x & 63);
    const bool live = i < n_nodes;
    BigNode& nd = nodes[live ? i : 0];
    const uint32_t* const table = bins + (size_t)(live ? i : 0) * (3 * BINS * BIN_WORDS);
    const size_t n = nd.e - nd.b;
    rtb::detail::SplitChoice mine{-1, -1};
    double mine_cost = 1e300;
    if (live && sub < 3)
        rtb::detail::split_axis(
            sub, nd.scale[sub], n, rtb::detail::child_capacity(level),
            [&](int ax, int k, Box* bb, double* bw, size_t* bn) { bin_read(table, ax, k, bb, bw, bn); }, &mine_cost, &mine);
    rtb::detail::SplitChoice ch{-1, -1};
    double best = 1e300;
    for (int ax = 0; ax < 3; ax++) {
        const double c = __shfl(mine_cost, (lane & ~3) + ax, 64);
        const int b = __shfl(mine.bin, (lane & ~3) + ax, 64);
        if (c < best) best = c, ch.axis = ax, ch.bin = b;
    }
    if (!live || sub != 0) return;
    nd.axis = ch.axis, nd.bin = ch.bin;
    if (ch.axis < 0) {  // coincident centroids in a large node: the host's median split is not built here
        atomicExch(&st->declined, 1u);
        return;
    }
    Box side[2];
    uint32_t cnt[2] = {0, 0}, nocull[2] = {0, 0};
    side[0].reset(), side[1].reset();
    for (int k = 0; k < BINS; k++) {
        const uint32_t* const w = table + (ch.axis * BINS + k) * BIN_WORDS;
        if (!w[7]) continue;
        const int s = k <= ch.bin ? 0 : 1;
        Box bb;
        for (int c = 0; c < 3; c++) bb.lo[c] = o2f(w[c]), bb.hi[c] = o2f(w[3 + c]);
        side[s].grow(bb.lo, bb.hi);
        cnt[s] += w[7], nocull[s] += w[8];
    }
    const uint32_t mid = nd.b + cnt[0];
    nd.cursor[0] = nd.b, nd.cursor[1] = mid;
    Node* const me = out + nd.base;
    me->pad[0] = me->pad[1] = 0;
    atomicMax(&st->levels, level + 1);
    for (int s = 0; s < 2; s++) {
        set_child_box(me, s, side[s]);
        const uint32_t cb = s ? mid : nd.b, ce = s ? nd.e : mid, cbase = s ? nd.base + cnt[0] : nd.base + 1;
        nd.leaf_side[s] = cnt[s] == 1;
        nd.child_big[s] = NONE;
        if (cnt[s] == 1) continue;  // (the item writes its ref)
        me->ref[s] = cbase | (nocull[s] ? rtb::NOCULL : 0u);
        if (cnt[s] <= (uint32_t)SMALL_MAX) {
            small[atomicAdd(&st->n_small, 1u)] = SmallWork{cb, ce, cbase, (uint32_t)level + 1, next_parity};
        } else {
            const uint32_t id = atomicAdd(&st->n_big_next, 1u);
            nd.child_big[s] = id;
            new_big(next + id, cb, ce, cbase);
            const uint32_t nch = (cnt[s] + CHUNK - 1) / CHUNK;
            const uint32_t at = atomicAdd(&st->n_chunks_next, nch);
            for (uint32_t c = 0; c < nch; c++)
                next_chunks[at + c] = Chunk{id, cb + c * CHUNK, min((uint32_t)CHUNK, cnt[s] - c * CHUNK)};
        }
    }
}

__global__ __launch_bounds__(CHUNK) void k_partition(const Item* items, Item* items_next, const Chunk* chunks, BigNode* nodes, Node* out,
                                                     BigNode* next) {
    __shared__ uint32_t wave_cnt[CHUNK / 64][2], side_base[2], cb[2][6];
    const Chunk ch = chunks[blockIdx.x];
    BigNode& nd = nodes[ch.node];
    const bool in = threadIdx.x < ch.count;
    if (threadIdx.x < 12) cb[threadIdx.x / 6][threadIdx.x % 6] = threadIdx.x % 6 < 3 ? 0xFFFFFFFFu : 0u;
    Item it;
    int s = 0;
    if (in) {
        it = items[ch.first + threadIdx.x];
        const int ax = nd.axis;
        s = rtb::detail::bin_index(it.lo[ax], it.hi[ax], nd.cb_lo_f[ax], nd.scale[ax]) <= nd.bin ? 0 : 1;
    }
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const unsigned long long m0 = __ballot(in && s == 0), m1 = __ballot(in && s == 1);
    if (lane == 0) wave_cnt[wave][0] = (uint32_t)__popcll(m0), wave_cnt[wave][1] = (uint32_t)__popcll(m1);
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t total = 0;
        for (int w = 0; w < CHUNK / 64; w++) total += wave_cnt[w][threadIdx.x];
        side_base[threadIdx.x] = total ? atomicAdd(&nd.cursor[threadIdx.x], total) : 0u;
    }
    // the centroid bounds of a large child, collected while its items pass through
    if (in && nd.child_big[s] != NONE)
        for (int k = 0; k < 3; k++) {
            const uint32_t c = f2o(0.5f * it.lo[k] + 0.5f * it.hi[k]);
            atomicMin(&cb[s][k], c), atomicMax(&cb[s][3 + k], c);
        }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int cs = (int)threadIdx.x / 6, f = (int)threadIdx.x % 6;
        const uint32_t child = nd.child_big[cs];
        if (child != NONE && wave_cnt[0][cs] + wave_cnt[1][cs] + wave_cnt[2][cs] + wave_cnt[3][cs]) {
            if (f < 3)
                atomicMin(&next[child].cb_lo[f], cb[cs][f]);
            else
                atomicMax(&next[child].cb_hi[f - 3], cb[cs][f]);
        }
    }
    if (!in) return;
    uint32_t at = side_base[s] + (uint32_t)__popcll((s ? m1 : m0) & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) at += wave_cnt[w][s];
    items_next[at] = it;
    if (nd.leaf_side[s]) out[nd.base].ref[s] = it.ref;
}

// ------------------------------------------------------------------------------------------
// One wave builds a subtree of <= 64 items: lane i holds the item at place i; the lanes of a node form a
// contiguous segment [sb, se); every segment of >= 2 items of a sub-level is split in the same trip of the loop.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_small(const Item* buf0, const Item* buf1, const SmallWork* work, Node* out, State* st) {
    __shared__ Item s_item[64];
    __shared__ uint32_t s_cb[32][6];
    __shared__ uint32_t s_bins[32][BINS * BIN_WORDS];
    __shared__ int s_axis[32], s_bin[32];
    __shared__ float s_key[64];
    __shared__ uint32_t s_ref[64];
    const SmallWork w = work[blockIdx.x];
    const int lane = (int)threadIdx.x, n = (int)(w.e - w.b);
    Item it{};
    if (lane < n) it = (w.parity ? buf1 : buf0)[w.b + lane];
    int sb = 0, se = lane < n ? n : 0, level = (int)w.level, deepest = 0;
    uint32_t base = w.base;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (;;) {
        const int sn = se - sb;
        const bool act = sn >= 2;
        if (!__ballot(act)) break;
        const bool leader = act && lane == sb;
        const unsigned long long lead_m = __ballot(leader);
        const int sid = __popcll(lead_m & ((1ull << sb) - 1ull));  // segments of this trip, counted from lane 0
        const bool two = act && sn == 2, gen = act && sn >= 3;
        const unsigned long long segmask = act ? (((se == 64) ? ~0ull : ((1ull << se) - 1ull)) & ~((1ull << sb) - 1ull)) : 0ull;
        if (leader) deepest = max(deepest, level + 1);
        // ---- two items: the first axis along which the centroids differ, lower first; coincident: by ref ----
        {
            Item q;  // the lane above
            q.lo[0] = __shfl_down(it.lo[0], 1, 64), q.lo[1] = __shfl_down(it.lo[1], 1, 64), q.lo[2] = __shfl_down(it.lo[2], 1, 64);
            q.hi[0] = __shfl_down(it.hi[0], 1, 64), q.hi[1] = __shfl_down(it.hi[1], 1, 64), q.hi[2] = __shfl_down(it.hi[2], 1, 64);
            q.ref = (uint32_t)__shfl_down((int)it.ref, 1, 64);
            if (two && leader) {
                bool swap = false, decided = false;
                for (int k = 0; k < 3 && !decided; k++) {
                    const float cp = 0.5f * it.lo[k] + 0.5f * it.hi[k], cq = 0.5f * q.lo[k] + 0.5f * q.hi[k];
                    if (cp == cq) continue;
                    swap = cq < cp, decided = true;
                }
                if (!decided) swap = q.ref < it.ref;
                const Item &a = swap ? q : it, &b = swap ? it : q;
                Node* const me = out + base;
                for (int k = 0; k < 3; k++) me->lo[k][0] = a.lo[k], me->hi[k][0] = a.hi[k], me->lo[k][1] = b.lo[k], me->hi[k][1] = b.hi[k];
                me->ref[0] = a.ref, me->ref[1] = b.ref, me->pad[0] = me->pad[1] = 0;
            }
        }
        // ---- three or more: centroid bounds of the segment ----
        float c[3];
        for (int k = 0; k < 3; k++) c[k] = 0.5f * it.lo[k] + 0.5f * it.hi[k];
        if (gen && leader)
            for (int k = 0; k < 3; k++) s_cb[sid][k] = 0xFFFFFFFFu, s_cb[sid][3 + k] = 0u;
        __syncthreads();
        if (gen)
            for (int k = 0; k < 3; k++) atomicMin(&s_cb[sid][k], f2o(c[k])), atomicMax(&s_cb[sid][3 + k], f2o(c[k]));
        __syncthreads();
        float cb_lo[3] = {0, 0, 0};
        double scale[3] = {0, 0, 0};
        if (gen)
            for (int k = 0; k < 3; k++) cb_lo[k] = o2f(s_cb[sid][k]), scale[k] = rtb::detail::bin_scale(cb_lo[k], o2f(s_cb[sid][3 + k]));
        // ---- the binned search, one axis at a time (the leader of each segment keeps the best) ----
        rtb::detail::SplitChoice best{-1, -1};
        double best_cost = 1e300;
        const size_t cap = rtb::detail::child_capacity(level);
        for (int ax = 0; ax < 3; ax++) {
            if (gen)  // (each segment clears its own table, its lanes taking turns)
                for (int j = lane - sb; j < BINS * BIN_WORDS; j += sn) clear_bins(&s_bins[sid][0], j);
            __syncthreads();
            if (gen && scale[ax] > 0) {
                uint32_t* const wd = &s_bins[sid][rtb::detail::bin_index(it.lo[ax], it.hi[ax], cb_lo[ax], scale[ax]) * BIN_WORDS];
                for (int k = 0; k < 3; k++) atomicMin(wd + k, f2o(it.lo[k])), atomicMax(wd + 3 + k, f2o(it.hi[k]));
                atomicAdd(wd + 6, (uint32_t)it.weight);
                atomicAdd(wd + 7, 1u);
            }
            __syncthreads();
            if (gen && leader)
                rtb::detail::split_axis(
                    ax, scale[ax], (size_t)sn, cap,
                    [&](int, int k, Box* bb, double* bw, size_t* bn) { bin_read(&s_bins[sid][0], 0, k, bb, bw, bn); }, &best_cost, &best);
            __syncthreads();
        }
        if (gen && leader) s_axis[sid] = best.axis, s_bin[sid] = best.bin;
        s_key[lane] = 0.0f, s_ref[lane] = it.ref;
        __syncthreads();
        int side = 0;
        bool median = false;
        if (gen) {
            const int ax = s_axis[sid];
            if (ax >= 0) {
                side = rtb::detail::bin_index(it.lo[ax], it.hi[ax], cb_lo[ax], scale[ax]) <= s_bin[sid] ? 0 : 1;
            } else {  // the median split along the widest axis of the centroid bounds (rt_bvh.h)
                median = true;
                int wx = 0;
                float ext[3];
                for (int k = 0; k < 3; k++) ext[k] = o2f(s_cb[sid][3 + k]) - cb_lo[k];
                for (int k = 1; k < 3; k++)
                    if (ext[k] > ext[wx]) wx = k;
                s_key[lane] = it.lo[wx] + it.hi[wx];
            }
        }
        __syncthreads();
        if (__ballot(median)) {
            int rank = 0;
            for (int j = 0; j < 64; j++)
                if (median && j >= sb && j < se && j != lane && rtb::detail::median_before(s_key[j], s_ref[j], s_key[lane], it.ref)) rank++;
            if (median) side = rank < sn / 2 ? 0 : 1;
        }
        // ---- boxes and counts of the two sides (bins 0 and 1 of the segment's table) ----
        if (gen)
            for (int j = lane - sb; j < 2 * BIN_WORDS; j += sn) clear_bins(&s_bins[sid][0], j);
        __syncthreads();
        if (gen) {
            uint32_t* const wd = &s_bins[sid][side * BIN_WORDS];
            for (int k = 0; k < 3; k++) atomicMin(wd + k, f2o(it.lo[k])), atomicMax(wd + 3 + k, f2o(it.hi[k]));
            atomicAdd(wd + 7, 1u);
            if (rtb::ref_nocull(it.ref)) atomicAdd(wd + 8, 1u);
        }
        __syncthreads();
        const unsigned long long left_m = __ballot(gen && side == 0);
        const int nl = __popcll(left_m & segmask), mid = sb + nl;
        if (gen && leader) {
            Node* const me = out + base;
            me->pad[0] = me->pad[1] = 0;
            for (int s = 0; s < 2; s++) {
                const uint32_t* const wd = &s_bins[sid][s * BIN_WORDS];
                for (int k = 0; k < 3; k++) me->lo[k][s] = o2f(wd[k]), me->hi[k][s] = o2f(wd[3 + k]);
                const int cnt = s ? sn - nl : nl;
                if (cnt >= 2) me->ref[s] = (s ? base + (uint32_t)nl : base + 1u) | (wd[8] ? rtb::NOCULL : 0u);
            }
        }
        int dest = lane;
        if (gen) {
            if (side == 0) {
                dest = sb + __popcll(left_m & segmask & below);
                if (nl == 1) out[base].ref[0] = it.ref;
            } else {
                dest = mid + __popcll(~left_m & segmask & below);
                if (sn - nl == 1) out[base].ref[1] = it.ref;
            }
        }
        __syncthreads();
        s_item[dest] = it;
        __syncthreads();
        it = s_item[lane];
        // the segment of the item now at this place
        if (two) {
            sb = se = 0;
        } else if (gen) {
            if (lane < mid)
                se = mid, base = base + 1u;
            else
                sb = mid, base = base + (uint32_t)nl;
            level++;
        }
        __syncthreads();
    }
    for (int d = 32; d > 0; d >>= 1) deepest = max(deepest, __shfl_xor(deepest, d, 64));
    if (lane == 0 && deepest) atomicMax(&st->levels, deepest);
}

// The root holder(s): node 0 = {group 0, group 1}, the optional last node = {group 0, group 2}
__global__ void k_holder(Node* out, HolderArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Box box[3];
    uint32_t ref[3];
    for (int h = 0; h < 3; h++) {
        box[h].reset();
        ref[h] = rtb::LEAF;
        if (a.n[h] == 1) {
            box[h].grow(a.single[h].lo, a.single[h].hi);
            ref[h] = a.single[h].ref;
        } else if (a.n[h] >= 2) {
            const Node& r = out[a.base[h]];
            for (int c = 0; c < 2; c++) {
                float lo[3], hi[3];
                for (int k = 0; k < 3; k++) lo[k] = r.lo[k][c], hi[k] = r.hi[k][c];
                box[h].grow(lo, hi);
            }
            ref[h] = a.base[h] | ((rtb::ref_nocull(r.ref[0]) || rtb::ref_nocull(r.ref[1])) ? rtb::NOCULL : 0u);
        }
    }
    Node* const h0 = out;
    set_child_box(h0, 0, box[0]), set_child_box(h0, 1, box[1]);
    h0->ref[0] = ref[0], h0->ref[1] = ref[1], h0->pad[0] = h0->pad[1] = 0;
    if (a.near_holder) {
        Node* const h2 = out + a.near_holder;
        set_child_box(h2, 0, box[0]), set_child_box(h2, 1, box[2]);
        h2->ref[0] = ref[0], h2->ref[1] = ref[2], h2->pad[0] = h2->pad[1] = 0;
    }
}

// The tight tree as the GPU walks it (rtb::quantize: same arithmetic, one thread per node)
__global__ __launch_bounds__(256) void k_quantize(const Node* in, rtb::QNode* out, uint32_t n, float b0, float b1, float b2, float s0,
                                                  float s1, float s2) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double B[3] = {b0, b1, b2}, S[3] = {s0, s1, s2};
    const Node nd = in[i];
    rtb::QNode q{};
    for (int side = 0; side < 2; side++) {
        const bool empty = !(nd.lo[0][side] <= nd.hi[0][side] && nd.lo[1][side] <= nd.hi[1][side] && nd.lo[2][side] <= nd.hi[2][side]);
        for (int c = 0; c < 3; c++) {
            uint32_t ul = 65535u, uh = 0u;
            if (!empty) {
                const double lo = nd.lo[c][side], hi = nd.hi[c][side];
                double l = floor((lo - B[c]) / S[c]), h = ceil((hi - B[c]) / S[c]);
                l = fmin(fmax(l, 0.0), 65535.0), h = fmin(fmax(h, 0.0), 65535.0);
                while (l > 0 && B[c] + l * S[c] > lo) l -= 1;
                while (h < 65535 && B[c] + h * S[c] < hi) h += 1;
                ul = (uint32_t)l, uh = (uint32_t)h;
            }
            q.lo[c] |= ul << (16 * side), q.hi[c] |= uh << (16 * side);
        }
        q.ref[side] = nd.ref[side];
    }
    out[i] = q;
}

// ------------------------------------------------------------------------------------------
// The items of both trees, made on the device from the flattened reference tree and the uploaded records with the
// host's own box rules (rt_bvh.h make_leaf_items / make_tight_items: binary64 without contraction, IEEE / and sqrt).
// Places are handed out by atomics -- the builds do not depend on the order of the items.
// ------------------------------------------------------------------------------------------
// place of this lane's `count` items among those of the wave's lanes, in a group whose fill count is *counter
__device__ __forceinline__ uint32_t wave_places(uint32_t* counter, uint32_t count) {
    uint32_t incl = count;
    const int lane = (int)(threadIdx.x & 63);
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += up;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(counter, total);
    base = (uint32_t)__shfl((int)base, 63, 64);
    return base + incl - count;
}

__global__ __launch_bounds__(256) void k_leaf_items(const rtf::FlatNode* nodes, uint32_t n_nodes, Item* out, uint32_t split, uint32_t* counters) {
    const uint32_t ni = blockIdx.x * 256 + threadIdx.x;
    rtf::FlatNode fn{};
    if (ni < n_nodes) fn = nodes[ni];
    const int kind = (fn.meta >> 29) & 3, count = (fn.meta >> 24) & 31;
    const bool valid = ni < n_nodes && kind != rtf::KIND_BRANCH && count != 0;
    const bool unb = kind == rtf::KIND_TRI_UNBOUNDED;
    const uint32_t p0 = wave_places(counters + 0, valid && !unb ? 1u : 0u), p1 = wave_places(counters + 1, valid && unb ? 1u : 0u);
    if (!valid) return;
    Item it;
    it.lo[0] = fn.minx + 0.0f, it.lo[1] = fn.miny + 0.0f, it.lo[2] = fn.minz + 0.0f;
    it.hi[0] = fn.maxx + 0.0f, it.hi[1] = fn.maxy + 0.0f, it.hi[2] = fn.maxz + 0.0f;
    it.ref = rtb::LEAF | (fn.meta & 0x7FFFFFFFu);
    it.weight = (float)count;
    out[unb ? split + p1 : p0] = it;
}

// groups: [0, split) triangles | [split, split2) spheres + unbounded triangles | [split2, ..) the same for near origins
__global__ __launch_bounds__(256) void k_tight_items(const rtf::FlatNode* nodes, uint32_t n_nodes, const unsigned char* tris,
                                                     const unsigned char* spheres, const int32_t* tri_slot_id, const int32_t* sphere_slot_id,
                                                     const uint8_t* tri_unbounded, rtb::TightPads pads, bool near_group, Item* out,
                                                     uint32_t split, uint32_t split2, uint32_t* counters, int32_t* tri_slot_leaf,
                                                     int32_t* sphere_slot_leaf) {
    const uint32_t ni = blockIdx.x * 256 + threadIdx.x;
    rtf::FlatNode fn{};
    if (ni < n_nodes) fn = nodes[ni];
    const int count = ni < n_nodes ? (int)((fn.meta >> 24) & 31) : 0, first = (int)(fn.meta & 0xFFFFFF);
    int kind = (fn.meta >> 29) & 3;
    const bool leaf = ni < n_nodes && kind != rtf::KIND_BRANCH;
    if (kind == rtf::KIND_TRI_UNBOUNDED) kind = rtf::KIND_TRI;  // decided per slot
    uint32_t c0 = 0, c1 = 0;
    if (leaf)
        for (int k = 0; k < count; k++) {
            if (kind == rtf::KIND_TRI && !tri_unbounded[first + k])
                c0++;
            else
                c1++;
        }
    uint32_t p0 = wave_places(counters + 0, c0), p1 = wave_places(counters + 1, c1);
    if (!leaf) return;
    for (int k = 0; k < count; k++) {
        const int slot = first + k;
        Item it;
        it.weight = 1.0f;
        it.ref = rtb::LEAF | ((uint32_t)kind << 29) | (1u << 24) | (uint32_t)slot;
        double lo[3], hi[3];
        if (kind == rtf::KIND_TRI) {
            tri_slot_leaf[slot] = (int32_t)ni;
            if (tri_unbounded[slot]) {  // its reference leaf box grown by the triangles' pad, never culled
                const double flo[3] = {fn.minx, fn.miny, fn.minz}, fhi[3] = {fn.maxx, fn.maxy, fn.maxz};
                rtb::grown_box(flo, fhi, pads.pad_tri, it.lo, it.hi);
                it.ref = rtb::LEAF | ((uint32_t)rtf::KIND_TRI_UNBOUNDED << 29) | (1u << 24) | (uint32_t)slot;
                out[split + p1] = it;
                if (near_group) out[split2 + p1] = it;
                p1++;
                continue;
            }
            const float* r = (const float*)(tris + (size_t)tri_slot_id[slot] * 96);
            rtb::projected_triangle_box(r, lo, hi);
            rtb::grown_box(lo, hi, pads.pad_tri + rtb::sliver_reach(r), it.lo, it.hi);
            out[p0++] = it;
        } else {
            sphere_slot_leaf[slot] = (int32_t)ni;
            const float* r = (const float*)(spheres + (size_t)sphere_slot_id[slot] * 48);
            for (int c = 0; c < 3; c++) lo[c] = (double)r[c] - r[3], hi[c] = (double)r[c] + r[3];
            if (near_group) {
                const double nsp = rtb::near_sphere_pad(r, pads), np = (double)pads.pad_sphere < nsp ? (double)pads.pad_sphere : nsp;
                Item nit = it;
                rtb::grown_box(lo, hi, np, nit.lo, nit.hi);
                out[split2 + p1] = nit;
            }
            rtb::grown_box(lo, hi, pads.pad_sphere, it.lo, it.hi);
            out[split + p1] = it;
            p1++;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Host side.  One build = one arena of device memory, freed when the object goes; the caller fills items() -- a copy
// of host items, or the kernels above -- and runs the build into its own node array.
// ------------------------------------------------------------------------------------------
inline size_t node_count(size_t n, size_t split, size_t split2) {
    if (n == 0) return 0;
    const size_t bounds[4] = {0, split, split2, n};
    size_t at = 1;
    for (int h = 0; h < 3; h++) {
        const size_t m = bounds[h + 1] - bounds[h];
        at += m ? m - 1 : 0;
    }
    return at + (split2 < n ? 1 : 0);
}
inline size_t node_count(const rtb::Items& src) { return node_count(src.items.size(), src.split, src.split2); }

class Build {
public:
    ~Build() {
        if (arena_) (void)hipFree(arena_);
    }
    bool alloc(size_t n_items) {
        n_ = n_items;
        cap_big_ = n_ / (SMALL_MAX + 1) + 8, cap_chunks_ = n_ / CHUNK + cap_big_ + 8, cap_small_ = n_ / 2 + 8;
        size_t at = 0;
        auto take = [&](size_t bytes) {
            const size_t here = at;
            at += (bytes + 255) & ~(size_t)255;
            return here;
        };
        const size_t o_items[2] = {take(n_ * sizeof(Item)), take(n_ * sizeof(Item))};
        const size_t o_big[2] = {take(cap_big_ * sizeof(BigNode)), take(cap_big_ * sizeof(BigNode))};
        const size_t o_chunks[2] = {take(cap_chunks_ * sizeof(Chunk)), take(cap_chunks_ * sizeof(Chunk))};
        const size_t o_bins = take(cap_big_ * 3 * BINS * BIN_WORDS * 4), o_small = take(cap_small_ * sizeof(SmallWork));
        const size_t o_state = take(sizeof(State)), o_counters = take(16);
        if (hipMalloc(&arena_, at) != hipSuccess) {
            arena_ = nullptr;
            return false;
        }
        unsigned char* const p = (unsigned char*)arena_;
        for (int k = 0; k < 2; k++)
            d_items_[k] = (Item*)(p + o_items[k]), d_big_[k] = (BigNode*)(p + o_big[k]), d_chunks_[k] = (Chunk*)(p + o_chunks[k]);
        d_bins_ = (uint32_t*)(p + o_bins), d_small_ = (SmallWork*)(p + o_small), d_state_ = (State*)(p + o_state);
        d_counters_ = (uint32_t*)(p + o_counters);
        return true;
    }
    Item* items() { return d_items_[0]; }
    uint32_t* counters() { return d_counters_; }  // four words for the item kernels' fill counts (the caller clears them)

    // Builds the tree over items()[0, n): rtb::detail::build_items's node layout into d_out[0, node_count) on the current
    // device, stream 0.  Returns 1 (built; *levels = rtb::Bvh::levels, *near_root = rtb::Bvh::near_root), 0 (declined: a
    // large node of coincident centroids -- the host builds), -1 (HIP error).
    int run(size_t split, size_t split2, Node* d_out, int* levels, uint32_t* near_root) {
        const size_t n = n_;
        *levels = 0, *near_root = 0;
        if (n == 0) return 1;
        if (n >= 0x1FFFFFFFull) return 0;
        auto ok = [](hipError_t e) { return e == hipSuccess; };
        const size_t bounds[4] = {0, split, split2, n};
        HolderArgs ha{};
        size_t at = 1;
        for (int h = 0; h < 3; h++) {
            ha.n[h] = (uint32_t)(bounds[h + 1] - bounds[h]);
            ha.base[h] = (uint32_t)at, at += ha.n[h] ? ha.n[h] - 1 : 0;
            if (ha.n[h] == 1 && !ok(hipMemcpy(&ha.single[h], d_items_[0] + bounds[h], sizeof(Item), hipMemcpyDeviceToHost))) return -1;
        }
        ha.near_holder = split2 < n ? (uint32_t)at : 0u;
        *near_root = ha.near_holder;
        // the group roots: large ones open the first level, small ones go to k_small as they are
        std::vector<BigNode> big0;
        std::vector<Chunk> chunks0;
        std::vector<SmallWork> small0;
        for (int h = 0; h < 3; h++) {
            const uint32_t b = (uint32_t)bounds[h], e = (uint32_t)bounds[h + 1];
            if (ha.n[h] < 2) continue;
            if (ha.n[h] <= (uint32_t)SMALL_MAX) {
                small0.push_back(SmallWork{b, e, ha.base[h], 1u, 0u});
                continue;
            }
            BigNode nd{};
            nd.b = b, nd.e = e, nd.base = ha.base[h];
            for (int k = 0; k < 3; k++) nd.cb_lo[k] = 0xFFFFFFFFu, nd.cb_hi[k] = 0u;
            for (uint32_t c = b; c < e; c += CHUNK) chunks0.push_back(Chunk{(uint32_t)big0.size(), c, std::min<uint32_t>(CHUNK, e - c)});
            big0.push_back(nd);
        }
        State state{};
        state.n_small = (uint32_t)small0.size();
        state.levels = 1;  // the holder
        if (!big0.empty() && (!ok(hipMemcpy(d_big_[0], big0.data(), big0.size() * sizeof(BigNode), hipMemcpyHostToDevice)) ||
                              !ok(hipMemcpy(d_chunks_[0], chunks0.data(), chunks0.size() * sizeof(Chunk), hipMemcpyHostToDevice))))
            return -1;
        if (!small0.empty() && !ok(hipMemcpy(d_small_, small0.data(), small0.size() * sizeof(SmallWork), hipMemcpyHostToDevice))) return -1;
        if (!ok(hipMemcpy(d_state_, &state, sizeof state, hipMemcpyHostToDevice))) return -1;
        uint32_t n_big = (uint32_t)big0.size(), n_chunks = (uint32_t)chunks0.size();
        int cur = 0;
        for (int level = 1; n_big > 0; level++) {
            if (level > rtb::MAX_LEVELS) return 0;
            if (level == 1) hipLaunchKernelGGL(k_bounds, dim3(n_chunks), dim3(CHUNK), 0, 0, d_items_[cur], d_chunks_[cur], d_big_[cur]);
            hipLaunchKernelGGL(k_scale, dim3(n_big), dim3(256), 0, 0, d_big_[cur], d_bins_);
            hipLaunchKernelGGL(k_bin, dim3(n_chunks), dim3(CHUNK), 0, 0, d_items_[cur], d_chunks_[cur], d_big_[cur], d_bins_);
            hipLaunchKernelGGL(k_split, dim3((n_big + 15) / 16), dim3(64), 0, 0, d_big_[cur], n_big, d_bins_, level, d_out, d_big_[cur ^ 1],
                               d_chunks_[cur ^ 1], d_small_, (uint32_t)(cur ^ 1), d_state_);
            hipLaunchKernelGGL(k_partition, dim3(n_chunks), dim3(CHUNK), 0, 0, d_items_[cur], d_items_[cur ^ 1], d_chunks_[cur], d_big_[cur],
                               d_out, d_big_[cur ^ 1]);
            if (!ok(hipMemcpy(&state, d_state_, sizeof state, hipMemcpyDeviceToHost))) return -1;
            if (state.declined) return 0;
            n_big = state.n_big_next, n_chunks = state.n_chunks_next;
            if (n_big > cap_big_ || n_chunks > cap_chunks_ || state.n_small > cap_small_) return -1;  // (cannot happen: see the caps)
            state.n_big_next = state.n_chunks_next = 0;
            if (!ok(hipMemcpy(d_state_, &state, sizeof state, hipMemcpyHostToDevice))) return -1;
            cur ^= 1;
        }
        if (state.n_small)
            hipLaunchKernelGGL(k_small, dim3(state.n_small), dim3(64), 0, 0, d_items_[0], d_items_[1], d_small_, d_out, d_state_);
        hipLaunchKernelGGL(k_holder, dim3(1), dim3(1), 0, 0, d_out, ha);
        if (!ok(hipGetLastError()) || !ok(hipMemcpy(&state, d_state_, sizeof state, hipMemcpyDeviceToHost))) return -1;
        *levels = state.levels;
        return 1;
    }

private:
    void* arena_ = nullptr;
    size_t n_ = 0, cap_big_ = 0, cap_chunks_ = 0, cap_small_ = 0;
    Item* d_items_[2] = {nullptr, nullptr};
    BigNode* d_big_[2] = {nullptr, nullptr};
    Chunk* d_chunks_[2] = {nullptr, nullptr};
    uint32_t* d_bins_ = nullptr;
    SmallWork* d_small_ = nullptr;
    State* d_state_ = nullptr;
    uint32_t* d_counters_ = nullptr;
};

// a tree over host items (the self-check's second route; an upload makes its items on the device)
inline int build_on_device(const rtb::Items& src, Node* d_out, int* levels, uint32_t* near_root) {
    Build b;
    *levels = 0, *near_root = 0;
    if (src.items.empty()) return 1;
    if (!b.alloc(src.items.size())) return 0;
    if (hipMemcpy(b.items(), src.items.data(), src.items.size() * sizeof(Item), hipMemcpyHostToDevice) != hipSuccess) return -1;
    return b.run(src.split, src.split2, d_out, levels, near_root);
}

// rtb::quantize's grid from the root holder of a tree on the device; false: no finite root box
inline bool quantize_on_device(const Node* d_nodes, size_t n_nodes, rtb::QNode* d_out, float* base, float* step) {
    Node root;
    if (n_nodes == 0 || hipMemcpy(&root, d_nodes, sizeof root, hipMemcpyDeviceToHost) != hipSuccess) return false;
    rtb::Bvh one;
    one.nodes.push_back(root);
    rtb::QBvh grid;
    if (!rtb::quantize_grid(one, &grid)) return false;
    for (int c = 0; c < 3; c++) base[c] = grid.base[c], step[c] = grid.step[c];
    hipLaunchKernelGGL(k_quantize, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, 0, d_nodes, d_out, (uint32_t)n_nodes, base[0],
                       base[1], base[2], step[0], step[1], step[2]);
    return hipGetLastError() == hipSuccess;
}

}  // namespace rtgb
