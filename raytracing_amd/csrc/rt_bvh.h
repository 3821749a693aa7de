// rt_bvh.h -- re-tree the reference's R-tree leaves into a binary BVH for per-lane ordered traversal.
//
// Why this is allowed.  In the reference a primitive is a candidate of a ray when every
// box on the path root -> its leaf passes IntersectBoundingBox (raytracer_fs.glsl:134-161, 163-199).
// That test is monotone under rounding: for a fixed ray with finite origin and finite 1/d,
//     lo' <= lo  =>  fl(lo' - o) <= fl(lo - o)  =>  fl(fl(lo' - o) * inv) <= fl(fl(lo - o) * inv)   (inv > 0;
// mirrored for inv < 0, where the test swaps the bounds), so growing a box can only lower its three
// near values and raise its three far values, and each of the test's four rejections
// (txmin > tymax, tymin > txmax, max(txmin, tymin) > tzmax, tzmin > min(txmax, tymax)) that is false for
// the smaller box is false for the larger one.  No NaN can arise (0 * inf needs an infinite 1/d).  The
// reference builds every parent box as the exact min/max of its children (RTree.cs:332-377; verified
// per upload, rtf::Flat::nested), so a leaf box that passes implies that all its ancestors pass:
//     candidates(ray) = { primitives of the leaves whose OWN box passes the reference test }.
// The inner structure therefore does not matter, as long as every inner box is an exact min/max union
// of leaf boxes (then it passes whenever a leaf below it passes) and every leaf box is tested with the
// reference formula.  This file builds such a tree with a binned surface-area heuristic; rays with a
// non-finite origin or 1/d (a zero direction component) do not use it (rt_kernels.h).
//
// Node = 64 B = 4 x float4, both child boxes in the parent (one node fetch decides both children), stored
// coordinate by coordinate as {child 0, child 1} pairs:
//     {lo.x0 lo.x1 lo.y0 lo.y1 | lo.z0 lo.z1 hi.x0 hi.x1 | hi.y0 hi.y1 hi.z0 hi.z1 | ref0 ref1 - -}
// (The pairs were laid out for packed fp32 -- one v_pk_add/mul/fma_f32 per slab parameter of both children.
// Measured and dropped, DESIGN.md 5.3: a packed instruction takes two issue slots on this GPU, tools/pk_rate.hip,
// so nothing is gained, and the pair operands cost the staged kernels twice the spills.  The layout stayed: the
// refs sit together in the last quarter, and the staged pipeline measured 4 % faster with it.)
// ref: bit 31 clear = inner node index;
// bit 31 set = leaf, bits 30..0 as rtf meta (kind, slot count, first slot); NOCULL below.  Node 0 holds the root box
// as child 0 and an empty leaf (or the tree of the unbounded leaves, NOCULL below) as child 1, so a
// traversal always starts with `cur = 0`.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "rt_flatten.h"

#if defined(__HIPCC__)
#define RT_BVH_HD __host__ __device__
#else
#define RT_BVH_HD
#endif

namespace rtb {

// A few worker threads whose failures come back to the caller.  A body that throws (std::bad_alloc on a
// multi-million-primitive scene) is caught in its own thread; every thread is joined -- also when the caller's own
// share of the work throws, or when a thread cannot be started -- and finish() rethrows the first failure on the
// calling thread, where the C ABI turns it into an error code (a joinable std::thread that is destroyed, or an
// exception that leaves a thread's body, would end the host process instead).
class Crew {
public:
    template <class F>
    void start(F f) {
        pool_.emplace_back([this, f]() mutable {
            try {
                f();
            } catch (...) {
                std::lock_guard<std::mutex> hold(mu_);
                if (!err_) err_ = std::current_exception();
            }
        });
    }
    void finish() {
        join();
        if (err_) {
            const std::exception_ptr e = err_;
            err_ = nullptr;
            std::rethrow_exception(e);
        }
    }
    ~Crew() { join(); }

private:
    void join() {
        for (std::thread& t : pool_)
            if (t.joinable()) t.join();
        pool_.clear();
    }
    std::vector<std::thread> pool_;
    std::exception_ptr err_;
    std::mutex mu_;
};

struct Node {
    float lo[3][2];  // [axis][child]   (float4 0 and the first half of float4 1)
    float hi[3][2];  // [axis][child]   (second half of float4 1 and float4 2)
    uint32_t ref[2];
    uint32_t pad[2];
    void set_child(int c, const float* l, const float* h, uint32_t r) {
        for (int k = 0; k < 3; k++) lo[k][c] = l[k], hi[k][c] = h[k];
        ref[c] = r;
    }
    void get_box(int c, float* l, float* h) const {
        for (int k = 0; k < 3; k++) l[k] = lo[k][c], h[k] = hi[k][c];
    }
};
static_assert(sizeof(Node) == 64, "BVH node is 64 B");

constexpr uint32_t LEAF = 0x80000000u;
// "never distance-cull this child": a leaf ref of kind 3 (rtf::KIND_TRI_UNBOUNDED) has bits 30 and 29 set;
// an inner ref carries both bits when such a leaf lies below it, so one mask test serves both.  Inner
// node index = ref & INDEX_MASK.
constexpr uint32_t NOCULL = 0x60000000u;
constexpr uint32_t INDEX_MASK = 0x1FFFFFFFu;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool ref_nocull(uint32_t ref) { return (ref & NOCULL) == NOCULL; }
constexpr int MAX_LEVELS = 32;  // inner nodes on any root->leaf path, holder included (= stack entries)

struct Bvh {
    std::vector<Node> nodes;
    int levels = 0;  // deepest chain of inner nodes actually built (<= MAX_LEVELS)
    int64_t leaves = 0;
    // tight tree only: index of a second root holder for rays whose origin lies inside near_lo..near_hi (its second
    // child is a spheres' tree with pads sized for such origins, build_tight); 0 = none
    uint32_t near_root = 0;
    float near_lo[3] = {0, 0, 0}, near_hi[3] = {0, 0, 0};
};

namespace detail {

struct Item {
    float lo[3], hi[3];
    uint32_t ref;
    float weight;  // primitives in the leaf
};
// (RT_BVH_HD: the device builder, rt_gpu_bvh.h, runs the same arithmetic -- binary64 without contraction, so the two
// builders decide every split alike)
struct Box {
    float lo[3], hi[3];
    RT_BVH_HD void reset() {
        for (int k = 0; k < 3; k++) lo[k] = 3.402823466e+38f, hi[k] = -3.402823466e+38f;
    }
    RT_BVH_HD void grow(const float* l, const float* h) {
        for (int k = 0; k < 3; k++) lo[k] = l[k] < lo[k] ? l[k] : lo[k], hi[k] = hi[k] < h[k] ? h[k] : hi[k];
    }
    RT_BVH_HD double half_area() const {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
};
constexpr int BINS = 16;
// bin of an item (its box along one axis) among BINS equal bins over the centroid bounds [lo, lo + BINS / scale]
RT_BVH_HD inline int bin_index(float item_lo, float item_hi, float lo, double scale) {
    const double c = 0.5f * item_lo + 0.5f * item_hi;
    int k = (int)((c - lo) * scale);
    return k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
}
RT_BVH_HD inline double bin_scale(float cb_lo, float cb_hi) {
    const double ext = (double)cb_hi - cb_lo;
    return ext > 0 ? BINS / ext : 0;
}
// The binned search: the cheapest of the <= 3 x (BINS - 1) splits between bins whose sides are non-empty and hold at
// most child_cap items each (axis -1: none).  bin_at(axis, k, &box, &weight, &count) reads one bin.  Ties keep the
// first candidate in (axis, bin) order.
struct SplitChoice {
    int axis, bin;
};
// One axis of the search (called for axis 0, 1, 2 in that order; *best_cost starts at 1e300, *best at {-1, -1}).
template <class BinAt>
RT_BVH_HD inline void split_axis(int ax, double scale_ax, size_t n, size_t child_cap, BinAt&& bin_at, double* best_cost, SplitChoice* best) {
    if (!(scale_ax > 0)) return;
    double right_cost[BINS];
    Box acc, bb;
    double bw;
    size_t bn;
    acc.reset();
    double w = 0;
    for (int k = BINS - 1; k > 0; k--) {
        bin_at(ax, k, &bb, &bw, &bn);
        if (bn) acc.grow(bb.lo, bb.hi);
        w += bw;
        right_cost[k] = w > 0 ? acc.half_area() * w : 0;
    }
    acc.reset();
    w = 0;
    size_t nl = 0;
    for (int k = 0; k + 1 < BINS; k++) {
        bin_at(ax, k, &bb, &bw, &bn);
        if (bn) acc.grow(bb.lo, bb.hi);
        w += bw;
        nl += bn;
        if (nl == 0 || nl == n) continue;
        if (nl > child_cap || n - nl > child_cap) continue;  // would break the level bound
        const double cost = acc.half_area() * w + right_cost[k + 1];
        if (cost < *best_cost) *best_cost = cost, best->axis = ax, best->bin = k;
    }
}
template <class BinAt>
RT_BVH_HD inline SplitChoice choose_split(const double* scale, size_t n, size_t child_cap, BinAt&& bin_at) {
    SplitChoice best{-1, -1};
    double best_cost = 1e300;
    for (int ax = 0; ax < 3; ax++) split_axis(ax, scale[ax], n, child_cap, bin_at, &best_cost, &best);
    return best;
}
// items a child of a node at `level` may hold so that its subtree still fits the MAX_LEVELS bound (n <= 2^room)
RT_BVH_HD inline size_t child_capacity(int level) {
    const int room = 32 /* MAX_LEVELS */ - level;
    return (room - 1 >= 31) ? (size_t)1 << 31 : (size_t)1 << (room - 1);
}
// order of two items with coincident centroids, and of equal keys of the median split: by ref (unique per item of a
// tree), so that a tree does not depend on the order its items arrive in -- the device builder partitions with
// atomics (rt_gpu_bvh.h)
RT_BVH_HD inline bool median_before(float key_x, uint32_t ref_x, float key_y, uint32_t ref_y) {
    return key_x < key_y || (key_x == key_y && ref_x < ref_y);
}

struct Builder {
    std::vector<Item>& items;
    Bvh& out;
    std::atomic<int> levels{0};
    static constexpr size_t FORK_MIN = 1 << 15;   // subtrees at least this large may be built by another thread
    static constexpr size_t SWEEP_MIN = 1 << 17;  // nodes at least this large are binned by several threads
    struct Bins {
        Box bb[3][BINS];
        double bw[3][BINS];
        size_t bn[3][BINS];
        void reset() {
            for (int ax = 0; ax < 3; ax++)
                for (int k = 0; k < BINS; k++) bb[ax][k].reset(), bw[ax][k] = 0, bn[ax][k] = 0;
        }
        void add(const Bins& o) {
            for (int ax = 0; ax < 3; ax++)
                for (int k = 0; k < BINS; k++) {
                    if (o.bn[ax][k]) bb[ax][k].grow(o.bb[ax][k].lo, o.bb[ax][k].hi);
                    bw[ax][k] += o.bw[ax][k], bn[ax][k] += o.bn[ax][k];
                }
        }
    };

    // Builds items[b, e) and returns its ref and box.  A subtree over m items has exactly m - 1 inner
    // nodes: it owns out.nodes[base, base + m - 1) (root first, then the left subtree, then the right), so
    // the layout does not depend on which thread builds what.  forks = thread forks still allowed below.
    uint32_t build(size_t b, size_t e, int level, Box* box, uint32_t base, int forks) {
        box->reset();
        if (e - b == 1) {
            box->grow(items[b].lo, items[b].hi);
            return items[b].ref;
        }
        const int room = MAX_LEVELS - level;  // levels left including this node: n <= 2^room holds
        const size_t n = e - b;
        if (n == 2) {
            // half of all nodes: what the binned search below decides for two items, without the bins -- the first
            // axis along which the centroids differ separates them (every such split costs the same), lower first
            Item &p = items[b], &q = items[b + 1];
            bool decided = false;
            for (int k = 0; k < 3 && !decided; k++) {
                const float cp = 0.5f * p.lo[k] + 0.5f * p.hi[k], cq = 0.5f * q.lo[k] + 0.5f * q.hi[k];
                if (cp == cq) continue;
                if (cq < cp) std::swap(p, q);
                decided = true;
            }
            if (!decided && q.ref < p.ref) std::swap(p, q);  // coincident centroids: by ref (median_before)
            for (int seen = levels.load(); level + 1 > seen && !levels.compare_exchange_weak(seen, level + 1);) {
            }
            Node& nd = out.nodes[base];
            nd.set_child(0, p.lo, p.hi, p.ref), nd.set_child(1, q.lo, q.hi, q.ref);
            nd.pad[0] = nd.pad[1] = 0;
            box->grow(p.lo, p.hi);
            box->grow(q.lo, q.hi);
            return base | ((ref_nocull(p.ref) || ref_nocull(q.ref)) ? NOCULL : 0u);
        }
        const size_t child_cap = child_capacity(level);
        (void)room;
        // The passes over a large node's items (centroid bounds, then the bins of all three axes in one sweep) are
        // cut into ranges for the threads this subtree may still fork: box unions are min/max and the bin weights
        // are small integers summed in double, so the result does not depend on how the range was cut.
        const int helpers = (forks > 0 && n >= SWEEP_MIN) ? (1 << std::min(forks, 4)) : 1;
        auto sweep = [&](auto&& body) {  // body(thread index, first, last)
            if (helpers == 1) return body(0, b, e);
            Crew crew;
            for (int t = 1; t < helpers; t++) crew.start([&, t] { body(t, b + n * t / helpers, b + n * (t + 1) / helpers); });
            body(0, b, b + n / helpers);
            crew.finish();
        };
        Box cb;  // centroid bounds
        cb.reset();
        {
            std::vector<Box> part(helpers);
            sweep([&](int t, size_t first, size_t last) {
                Box c_;
                c_.reset();
                for (size_t i = first; i < last; i++) {
                    float c[3];
                    for (int k = 0; k < 3; k++) c[k] = 0.5f * items[i].lo[k] + 0.5f * items[i].hi[k];
                    c_.grow(c, c);
                }
                part[t] = c_;
            });
            for (const Box& p : part) cb.grow(p.lo, p.hi);
        }
        double scale[3];
        for (int ax = 0; ax < 3; ax++) scale[ax] = bin_scale(cb.lo[ax], cb.hi[ax]);
        std::vector<Bins> bins(helpers);
        sweep([&](int t, size_t first, size_t last) {
            Bins& bs = bins[t];
            bs.reset();
            for (size_t i = first; i < last; i++)
                for (int ax = 0; ax < 3; ax++) {
                    if (!(scale[ax] > 0)) continue;
                    const int k = bin_of(items[i], ax, cb.lo[ax], scale[ax]);
                    bs.bb[ax][k].grow(items[i].lo, items[i].hi);
                    bs.bw[ax][k] += items[i].weight;
                    bs.bn[ax][k]++;
                }
        });
        for (int t = 1; t < helpers; t++) bins[0].add(bins[t]);
        const SplitChoice choice = choose_split(scale, n, child_cap, [&](int ax, int k, Box* bb, double* bw, size_t* bn) {
            *bb = bins[0].bb[ax][k], *bw = bins[0].bw[ax][k], *bn = bins[0].bn[ax][k];
        });
        const int best_axis = choice.axis, best_bin = choice.bin;
        size_t mid;
        if (best_axis >= 0) {
            const int ax = best_axis;
            const float lo = cb.lo[ax];
            const double sc = scale[ax];
            const int bin = best_bin;
            auto it = std::partition(items.begin() + b, items.begin() + e,
                                     [&](const Item& it_) { return bin_of(it_, ax, lo, sc) <= bin; });
            mid = (size_t)(it - items.begin());
        } else {
            // coincident centroids (or no split inside the level bound): median along the widest axis
            int ax = 0;
            for (int k = 1; k < 3; k++)
                if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;
            mid = b + n / 2;
            std::nth_element(items.begin() + b, items.begin() + mid, items.begin() + e, [ax](const Item& x, const Item& y) {
                return median_before(x.lo[ax] + x.hi[ax], x.ref, y.lo[ax] + y.hi[ax], y.ref);
            });
        }
        const uint32_t me = base;
        for (int seen = levels.load(); level + 1 > seen && !levels.compare_exchange_weak(seen, level + 1);) {
        }
        Box l, r;
        uint32_t rl, rr;
        const uint32_t right_base = base + (uint32_t)(mid - b);  // 1 for this node + (mid - b - 1) of the left subtree
        if (forks > 0 && n >= FORK_MIN) {
            Crew crew;
            crew.start([&] { rl = build(b, mid, level + 1, &l, base + 1, forks - 1); });
            rr = build(mid, e, level + 1, &r, right_base, forks - 1);
            crew.finish();
        } else {
            rl = build(b, mid, level + 1, &l, base + 1, 0);
            rr = build(mid, e, level + 1, &r, right_base, 0);
        }
        Node& nd = out.nodes[me];
        nd.set_child(0, l.lo, l.hi, rl), nd.set_child(1, r.lo, r.hi, rr);
        nd.pad[0] = nd.pad[1] = 0;
        box->grow(l.lo, l.hi);
        box->grow(r.lo, r.hi);
        return me | ((ref_nocull(rl) || ref_nocull(rr)) ? NOCULL : 0u);
    }
    static int bin_of(const Item& it, int ax, float lo, double scale) { return bin_index(it.lo[ax], it.hi[ax], lo, scale); }
};

}  // namespace detail

namespace detail {
// Builds items[0, split) and items[split, split2) as the two children of the root holder (node 0), so a
// traversal always starts with `cur = 0`.  split == n: one tree, the second child is an empty leaf.
// split2 < n (tight tree only): items[split2, n) become a third tree, and a second root holder -- the LAST node,
// out->near_root -- holds the first tree and that one.
inline void build_items(std::vector<Item>& items, size_t split, Bvh* out, size_t split2 = (size_t)-1) {
    out->nodes.clear();
    out->levels = 0;
    out->near_root = 0;
    if (split2 > items.size()) split2 = items.size();
    out->leaves = (int64_t)split2;
    if (items.empty()) return;
    const size_t bounds[4] = {0, split, split2, items.size()};
    const bool third = split2 < items.size();
    size_t base[3], at = 1;  // node 0: root holder
    for (int h = 0; h < 3; h++) {
        const size_t n = bounds[h + 1] - bounds[h];
        base[h] = at, at += n ? n - 1 : 0;  // a subtree over n items owns n - 1 nodes
    }
    out->nodes.assign(at + (third ? 1 : 0), Node{});
    Builder bld{items, *out};
    bld.levels = 1;
    Box box[3];
    uint32_t ref[3] = {LEAF, LEAF, LEAF};  // empty leaf (count 0): harmless even if its box were accepted
    for (int h = 0; h < 3; h++) {
        const size_t b = bounds[h], e = bounds[h + 1];
        box[h].reset();  // inverted: rejected by any finite ray
        if (e > b) ref[h] = bld.build(b, e, 1, &box[h], (uint32_t)base[h], 4);
    }
    out->levels = bld.levels.load();
    Node& h = out->nodes[0];
    h.set_child(0, box[0].lo, box[0].hi, ref[0]), h.set_child(1, box[1].lo, box[1].hi, ref[1]);
    h.pad[0] = h.pad[1] = 0;
    if (third) {
        out->near_root = (uint32_t)at;
        Node& h2 = out->nodes[at];
        h2.set_child(0, box[0].lo, box[0].hi, ref[0]), h2.set_child(1, box[2].lo, box[2].hi, ref[2]);
        h2.pad[0] = h2.pad[1] = 0;
    }
}
}  // namespace detail

// Re-tree the leaves of a flattened reference tree (flat.nested must hold).  Leaves without
// primitives are dropped (they have no candidates).  An empty tree yields no nodes.  Leaves that hold
// an unbounded triangle (kind 3) get a tree of their own under the root holder's second slot, so the
// never-cull mark stays off every path of the main tree.
// The items of a tree and how they fall into the subtrees under its root holder(s) (detail::build_items).
struct Items {
    std::vector<detail::Item> items;
    size_t split = 0, split2 = 0;  // [0, split) | [split, split2) | [split2, size): third group only in the tight tree
};
// (x + 0.0f: a bound of -0 becomes +0.  Box unions are min/max, which return either zero of a tie depending on the
// order of the operands; with one zero the unions are the same bits in any order -- the two builders visit the items
// in different orders -- and no comparison of the box tests can tell the zeros apart.)
inline void make_leaf_items(const rtf::Flat& flat, Items* out) {
    std::vector<detail::Item>& items = out->items;
    std::vector<detail::Item> unbounded;
    items.clear();
    for (const rtf::FlatNode& fn : flat.nodes) {
        const int kind = (fn.meta >> 29) & 3, count = (fn.meta >> 24) & 31;
        if (kind == rtf::KIND_BRANCH || count == 0) continue;
        detail::Item it;
        it.lo[0] = fn.minx + 0.0f, it.lo[1] = fn.miny + 0.0f, it.lo[2] = fn.minz + 0.0f;
        it.hi[0] = fn.maxx + 0.0f, it.hi[1] = fn.maxy + 0.0f, it.hi[2] = fn.maxz + 0.0f;
        it.ref = LEAF | (fn.meta & 0x7FFFFFFFu);
        it.weight = (float)count;
        (kind == rtf::KIND_TRI_UNBOUNDED ? unbounded : items).push_back(it);
    }
    out->split = items.size();
    items.insert(items.end(), unbounded.begin(), unbounded.end());
    out->split2 = items.size();
}
inline void build(const rtf::Flat& flat, Bvh* out) {
    Items it;
    make_leaf_items(flat, &it);
    detail::build_items(it.items, it.split, out);
}

// ------------------------------------------------------------------------------------------
// The tight tree: one leaf per primitive slot, boxes fitted to the primitive itself.
//
// The reference's leaf boxes are loose for a BVH (an R-tree leaf bounds up to five primitives picked
// by insertion order): on SoupScene a shadow ray passes the box of ~30 leaves and tests ~64
// primitives of which ~0.1 are hit.  The tight tree is a FILTER in front of the exact candidate
// rule, never a replacement for it:
//
//   a hit (slot, t) is accepted by the reference
//        <=>  the primitive test accepts t   AND   the slot's reference leaf box passes the
//             reference box test (first paragraph of this file; regular rays, nested tree)
//
// The traversal of the tight tree finds every slot whose primitive test can accept a t inside the
// caller's window, runs the reference's primitive test on it, and checks the reference leaf box
// (slot -> leaf record, *_slot_leaf below) only for hits it is about to accept.  For this the tight
// boxes must be CONSERVATIVE for the computed hit parameter t_c, i.e. t_c must lie inside the slab
// interval of every box on the path as the traversal computes it (rt_device.h, SlabHit):
//
//   triangle  IntersectTriangle intersects the plane through A with the STORED normal N and accepts
//             points X whose projection along N falls inside ABC, i.e. X in the triangle A B' C'
//             (B', C' = B, C projected along N onto that plane; equal to B, C for an exact normal).
//             The accepted X_c = o + t_c d misses that plane by <= eps(5|A-o| + 3 t_c): < 32 eps E with
//             E >= |o_k| + |coordinate| on every axis.  An edge test is off by up to ~4 eps |edge| at the
//             edge's far end, so where two edges meet at a small angle the accepted region reaches past the
//             vertex: by <= 8 eps L / sin(angle), L = longest edge (sliver_reach below).  The box is
//             bbox(A, B', C') grown by pad_tri = 2^-14 E_build (1000 eps E_build) PLUS that reach; triangles
//             whose reach exceeds max_extent / 512 are treated as unbounded (rt_scene_prep.h).
//   sphere    the computed root differs from an exact root (or, when rounding created the root, from
//             the closest approach) by <= sqrt(68 eps)/2 |o-c|, and the closest approach misses the
//             sphere by at most the same amount: X_c is within 2.0e-3 |o-c| <= 3.5e-3 E of the
//             sphere.  The box is c +- (r + pad_sphere), pad_sphere = 2^-8 E_build = 3.9e-3 E_build.
//   unbounded triangles (rtf::meta_nocull: the reference accepts them on a whole line or plane, wherever
//             the ray's LINE passes their leaf box -- the box test has no distance bound) get that
//             reference leaf box grown by pad_tri, hang in the spheres' tree, and carry the never-cull
//             mark (kind 3, propagated upwards): SlabHit drops its distance window for them.  If the exact
//             test accepts the leaf box, its three computed slab intervals share a point t*; the padded
//             box's intervals, as SlabHit computes them, contain every computed interval of the unpadded box
//             (pad_tri |1/d_k| >> the rounding of either formula), hence t*: the filter passes.
//   slab test errors (fma form, 1/d rounded) are < 6 eps E in space: covered by either pad.
//
// E_build is fixed at build time, so only rays with E(o) = max_k(|o_k| + extent_k) <= E_build and a
// regular direction may use this tree; the others take the leaf-box tree above.  Inner boxes are
// unions of the padded leaf boxes.
// ------------------------------------------------------------------------------------------
// Shape of a triangle record (TriangleStruct: A|_ B|_ C|_ N|_) as the edge tests see it: L = longest edge,
// ratio = (area of A B' C' doubled) / L^2 ~ the smallest angle for a sliver.  False if the stored normal is zero
// or NaN (then the triangle is never hit) -- ratio may still be 0 or NaN for collinear vertices.
RT_BVH_HD inline bool triangle_shape(const float* rec, double* ratio, double* longest) {
    const float* n = rec + 12;
    const double nn = sqrt((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2]);
    if (!(nn == nn) || !(nn > 0)) return false;
    const double e1[3] = {(double)rec[4] - rec[0], (double)rec[5] - rec[1], (double)rec[6] - rec[2]};
    const double e2[3] = {(double)rec[8] - rec[0], (double)rec[9] - rec[1], (double)rec[10] - rec[2]};
    const double e3[3] = {e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2]};
    const double cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double area2 = fabs(cr[0] * n[0] + cr[1] * n[1] + cr[2] * n[2]) / nn;
    const double l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2],
                 l3 = e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2];
    double len2 = l1;  // (std::max of the three)
    if (len2 < l2) len2 = l2;
    if (len2 < l3) len2 = l3;
    *ratio = area2 / len2, *longest = sqrt(len2);
    return true;
}
// How far past a vertex the rounding of the edge tests lets an accepted hit point lie (see the proof above);
// 0 for a triangle that is never hit, +inf for collinear vertices.
RT_BVH_HD inline double sliver_reach(const float* rec) {
    double ratio, longest;
    if (!triangle_shape(rec, &ratio, &longest)) return 0.0;
    if (!(ratio > 0)) return INFINITY;
    return 8.0 * 5.9604644775390625e-8 * longest / ratio;
}

struct TightPads {  // (the part of TightInput the per-slot box rules need; a kernel argument of rt_gpu_bvh.h)
    float pad_tri = 0, pad_sphere = 0;
    // Origins of many rays of a frame -- shadow and bounce rays from hit points on the primitives -- lie inside the
    // scene's root box (grown a little: near_lo..near_hi).  For such an origin |o - c| is bounded per sphere by the
    // distance D from its centre to the farthest corner of that region, so its pad can be NEAR_REL * D (+ pad_tri)
    // instead of pad_sphere, which is sized for origins up to E_build away (Soup1M: 0.13-0.25 against 0.625 around
    // spheres of radius 0.05-0.4).  near_on: build that second spheres' tree (Bvh::near_root).
    bool near_on = false;
    float near_lo[3] = {0, 0, 0}, near_hi[3] = {0, 0, 0};
};
struct TightInput : TightPads {
    const unsigned char* tris = nullptr;     // TriangleStruct records, 96 B: A|_ B|_ C|_ N|_ ...
    const unsigned char* spheres = nullptr;  // SphereStruct records, 48 B: center, radius
    const std::vector<uint8_t>* tri_unbounded = nullptr;  // per triangle slot: accepted on a line / plane (rtf::meta_nocull)
};
constexpr double NEAR_REL = 2.05e-3;  // the sphere bound of the proof above (X_c within 2.0e-3 |o-c| of the sphere), with room
// pad of a sphere's box in the tree for near origins
RT_BVH_HD inline double near_sphere_pad(const float* rec, const TightPads& in) {
    double d2 = 0;
    for (int c = 0; c < 3; c++) {
        const double a = fabs((double)rec[c] - in.near_lo[c]), b = fabs((double)in.near_hi[c] - rec[c]);
        const double far = a < b ? b : a;
        d2 += far * far;
    }
    return NEAR_REL * sqrt(d2) + in.pad_tri;
}
// The boxes of the tight tree's items (the rules of the comment above), shared with the device's item kernels.
// (+ 0.0f: make_leaf_items)
RT_BVH_HD inline float round_down(double x) {
    const float f = (float)x;
    return ((double)f > x ? nextafterf(f, -INFINITY) : f) + 0.0f;
}
RT_BVH_HD inline float round_up(double x) {
    const float f = (float)x;
    return ((double)f < x ? nextafterf(f, INFINITY) : f) + 0.0f;
}
RT_BVH_HD inline void grown_box(const double* lo, const double* hi, double pad, float* out_lo, float* out_hi) {
    for (int c = 0; c < 3; c++) out_lo[c] = round_down(lo[c] - pad), out_hi[c] = round_up(hi[c] + pad);
}
// bbox(A, B', C'): B', C' = B, C projected along the stored normal onto the plane through A
RT_BVH_HD inline void projected_triangle_box(const float* r, double* lo, double* hi) {
    const double n[3] = {r[12], r[13], r[14]};
    const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    for (int c = 0; c < 3; c++) lo[c] = hi[c] = r[c];
    for (int v = 1; v < 3; v++) {
        const double e[3] = {(double)r[4 * v] - r[0], (double)r[4 * v + 1] - r[1], (double)r[4 * v + 2] - r[2]};
        // projection along N onto the plane through A; a zero or NaN normal never yields a hit
        const double s = (e[0] * n[0] + e[1] * n[1] + e[2] * n[2]) / nn;
        const bool project = nn > 0 && s == s;
        for (int c = 0; c < 3; c++) {
            const double x = project ? (double)r[4 * v + c] - s * n[c] : (double)r[4 * v + c];
            lo[c] = x < lo[c] ? x : lo[c], hi[c] = hi[c] < x ? x : hi[c];
        }
    }
}

inline void make_tight_items(const rtf::Flat& flat, const TightInput& in, Items* out, std::vector<int32_t>* tri_slot_leaf,
                             std::vector<int32_t>* sphere_slot_leaf) {
    // triangles and spheres get a tree each (the two children of the root holder): the spheres' pad is
    // 64 x the triangles', mixed in they would inflate every inner box above them
    std::vector<detail::Item>& items = out->items;
    items.clear();
    std::vector<detail::Item> sphere_items, near_items;
    items.reserve(flat.tri_slot_id.size() + 2 * flat.sphere_slot_id.size());
    tri_slot_leaf->assign(flat.tri_slot_id.size(), 0);
    sphere_slot_leaf->assign(flat.sphere_slot_id.size(), 0);
    auto fill = [&](size_t n0, size_t n1, std::vector<detail::Item>& items, std::vector<detail::Item>& sphere_items,
                    std::vector<detail::Item>& near_items) {
    for (size_t ni = n0; ni < n1; ni++) {
        const rtf::FlatNode& fn = flat.nodes[ni];
        const int count = (fn.meta >> 24) & 31, first = (int)(fn.meta & 0xFFFFFF);
        int kind = (fn.meta >> 29) & 3;
        if (kind == rtf::KIND_BRANCH) continue;
        if (kind == rtf::KIND_TRI_UNBOUNDED) kind = rtf::KIND_TRI;  // decided per slot below
        for (int k = 0; k < count; k++) {
            const int slot = first + k;
            detail::Item it;
            it.weight = 1.0f;
            it.ref = LEAF | ((uint32_t)kind << 29) | (1u << 24) | (uint32_t)slot;
            double lo[3], hi[3], pad;
            if (kind == rtf::KIND_TRI && in.tri_unbounded && (*in.tri_unbounded)[slot]) {
                // hits anywhere along a line or plane -- but only where the ray's LINE passes the reference leaf
                // box (the exact rule): that box grown by the triangles' pad, walked without a distance window
                // (kind 3 = never cull, propagated to the inner refs above).  Kept out of the triangles' tree.
                (*tri_slot_leaf)[slot] = (int32_t)ni;
                const double flo[3] = {fn.minx, fn.miny, fn.minz}, fhi[3] = {fn.maxx, fn.maxy, fn.maxz};
                grown_box(flo, fhi, in.pad_tri, it.lo, it.hi);
                it.ref = LEAF | ((uint32_t)rtf::KIND_TRI_UNBOUNDED << 29) | (1u << 24) | (uint32_t)slot;
                sphere_items.push_back(it);
                if (in.near_on) near_items.push_back(it);  // (reachable from either root holder)
                continue;
            }
            if (kind == rtf::KIND_TRI) {
                (*tri_slot_leaf)[slot] = (int32_t)ni;
                const float* r = (const float*)(in.tris + (size_t)flat.tri_slot_id[slot] * 96);
                projected_triangle_box(r, lo, hi);
                pad = in.pad_tri + sliver_reach(r);  // (bounded: <= max_extent / 512 for every triangle that gets here)
            } else {
                (*sphere_slot_leaf)[slot] = (int32_t)ni;
                const float* r = (const float*)(in.spheres + (size_t)flat.sphere_slot_id[slot] * 48);
                for (int c = 0; c < 3; c++) lo[c] = (double)r[c] - r[3], hi[c] = (double)r[c] + r[3];
                pad = in.pad_sphere;
                if (in.near_on) {
                    const double nsp = near_sphere_pad(r, in), np = (double)in.pad_sphere < nsp ? (double)in.pad_sphere : nsp;
                    detail::Item nit = it;
                    grown_box(lo, hi, np, nit.lo, nit.hi);
                    near_items.push_back(nit);
                }
            }
            grown_box(lo, hi, pad, it.lo, it.hi);
            (kind == rtf::KIND_TRI ? items : sphere_items).push_back(it);
        }
    }
    };
    // leaf records are independent (each slot belongs to one leaf): node ranges on up to 16 threads, results
    // concatenated in range order, so the item order -- and with it the tree -- does not depend on the thread count
    const size_t n_nodes = flat.nodes.size();
    const int n_threads = n_nodes < (1u << 16) ? 1 : (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (n_threads == 1) {
        fill(0, n_nodes, items, sphere_items, near_items);
    } else {
        std::vector<std::vector<detail::Item>> part(n_threads), sphere_part(n_threads), near_part(n_threads);
        Crew crew;
        for (int t = 0; t < n_threads; t++)
            crew.start([&, t] { fill(n_nodes * t / n_threads, n_nodes * (t + 1) / n_threads, part[t], sphere_part[t], near_part[t]); });
        crew.finish();
        for (int t = 0; t < n_threads; t++) {
            items.insert(items.end(), part[t].begin(), part[t].end());
            sphere_items.insert(sphere_items.end(), sphere_part[t].begin(), sphere_part[t].end());
            near_items.insert(near_items.end(), near_part[t].begin(), near_part[t].end());
        }
    }
    // (a near tree that holds no sphere -- only unbounded triangles, or nothing -- would be the far tree again)
    bool any_sphere = false;
    for (const detail::Item& it : near_items) any_sphere |= ((it.ref >> 29) & 3) == (uint32_t)rtf::KIND_SPHERE;
    if (!any_sphere) near_items.clear();
    out->split = items.size();
    items.insert(items.end(), sphere_items.begin(), sphere_items.end());
    out->split2 = items.size();
    items.insert(items.end(), near_items.begin(), near_items.end());
}
inline void build_tight(const rtf::Flat& flat, const TightInput& in, Bvh* out, std::vector<int32_t>* tri_slot_leaf,
                        std::vector<int32_t>* sphere_slot_leaf) {
    Items it;
    make_tight_items(flat, in, &it, tri_slot_leaf, sphere_slot_leaf);
    detail::build_items(it.items, it.split, out, it.split2);
    for (int c = 0; c < 3; c++) out->near_lo[c] = in.near_lo[c], out->near_hi[c] = in.near_hi[c];
}

// ------------------------------------------------------------------------------------------
// The tight tree as the GPU walks it: 32-byte nodes, child boxes on a 16-bit grid.
//
// A node step of the staged pipeline is a gather: every lane of a wave fetches its own node, and the
// cost of a gather instruction is the number of cache lines it touches (64 for 64 rays), whatever its
// width.  A 64-byte fp32 node takes four such instructions per step; the tight tree is only a filter
// (see above), so its boxes need not be exact, only conservative: each coordinate is stored as a 16-bit
// index of a uniform grid over the tree's root box, lower bounds rounded down, upper bounds up
// (coordinate = base + index * step per axis).  Two 16-byte loads fetch a node, the tree takes half the
// cache, and a box grows by less than one grid step (extent / 65535 <= 2^-16 E) per side: a quarter of
// pad_tri.  The traversal never forms the coordinate: with inv_q = step/d and pad_q = (base - o)/d per
// ray, the slab parameter is fma(index, inv_q, pad_q) (rt_device.h SlabHitQ; rounding < 4 eps E in
// space, inside the pads' budget like the fp32 form's).
//   word k (k = 0..2): lower bound of axis k, child 0 in bits 0..15, child 1 in bits 16..31
//   word 3 + k:        upper bound of axis k, same packing;    words 6, 7: the two refs
// ------------------------------------------------------------------------------------------
struct QNode {
    uint32_t lo[3], hi[3], ref[2];
};
static_assert(sizeof(QNode) == 32, "QNode is two 16-byte loads");
struct QBvh {
    std::vector<QNode> nodes;
    float base[3] = {0, 0, 0}, step[3] = {1, 1, 1};
    uint32_t near_root = 0;  // Bvh::near_root
    // the box child `side` of node i stands for (what the traversal tests against)
    void get_box(size_t i, int side, double* lo, double* hi) const {
        for (int c = 0; c < 3; c++) {
            lo[c] = (double)base[c] + (double)((nodes[i].lo[c] >> (16 * side)) & 0xFFFFu) * (double)step[c];
            hi[c] = (double)base[c] + (double)((nodes[i].hi[c] >> (16 * side)) & 0xFFFFu) * (double)step[c];
        }
    }
};

// Quantises a tree built by build_tight.  Empty child slots (inverted boxes) stay inverted (lo index 65535,
// hi index 0).  Returns false if the tree has no finite root box (then there is nothing to walk).
// the grid of a tree: base and step per axis from the boxes of its root holder (node 0)
inline bool quantize_grid(const Bvh& in, QBvh* out) {
    out->nodes.clear();
    if (in.nodes.empty()) return false;
    double rlo[3] = {1e300, 1e300, 1e300}, rhi[3] = {-1e300, -1e300, -1e300};
    for (int side = 0; side < 2; side++) {
        float lo[3], hi[3];
        in.nodes[0].get_box(side, lo, hi);
        if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) continue;
        for (int c = 0; c < 3; c++) rlo[c] = std::min(rlo[c], (double)lo[c]), rhi[c] = std::max(rhi[c], (double)hi[c]);
    }
    for (int c = 0; c < 3; c++) {
        if (!(rlo[c] <= rhi[c]) || !std::isfinite(rlo[c]) || !std::isfinite(rhi[c])) return false;
        out->base[c] = std::nextafterf((float)rlo[c], -INFINITY);
        // a step that reaches the upper bound with room to spare in index 65535 (never zero: flat scenes)
        float st = (float)((rhi[c] - (double)out->base[c]) / 65534.0);
        st = std::nextafterf(st, INFINITY);
        if (!(st > 0)) st = 1e-30f;
        out->step[c] = st;
    }
    return true;
}
inline bool quantize(const Bvh& in, QBvh* out) {
    if (!quantize_grid(in, out)) return false;
    out->nodes.resize(in.nodes.size());
    out->near_root = in.near_root;
    const size_t n = in.nodes.size();
    auto run = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            QNode q{};
            for (int side = 0; side < 2; side++) {
                float lo[3], hi[3];
                in.nodes[i].get_box(side, lo, hi);
                const bool empty = !(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]);
                for (int c = 0; c < 3; c++) {
                    uint32_t ul = 65535u, uh = 0u;
                    if (!empty) {
                        const double B = out->base[c], S = out->step[c];
                        double l = std::floor(((double)lo[c] - B) / S), h = std::ceil(((double)hi[c] - B) / S);
                        l = std::min(std::max(l, 0.0), 65535.0), h = std::min(std::max(h, 0.0), 65535.0);
                        while (l > 0 && B + l * S > (double)lo[c]) l -= 1;
                        while (h < 65535 && B + h * S < (double)hi[c]) h += 1;
                        ul = (uint32_t)l, uh = (uint32_t)h;
                    }
                    q.lo[c] |= ul << (16 * side), q.hi[c] |= uh << (16 * side);
                }
                q.ref[side] = in.nodes[i].ref[side];
            }
            out->nodes[i] = q;
        }
    };
    const int n_threads = n < (1u << 16) ? 1 : (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (n_threads == 1) {
        run(0, n);
    } else {
        Crew crew;
        for (int t = 0; t < n_threads; t++) crew.start([&, t] { run(n * t / n_threads, n * (t + 1) / n_threads); });
        crew.finish();
    }
    return true;
}

}  // namespace rtb
