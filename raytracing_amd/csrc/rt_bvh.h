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
// Node = 64 B = 4 x float4: {lo0.xyz, ref0 | hi0.xyz, ref1 | lo1.xyz, - | hi1.xyz, -}: both child boxes
// sit in the parent, so one node fetch decides both children.  ref: bit 31 clear = inner node index;
// bit 31 set = leaf, bits 30..0 as rtf meta (kind, slot count, first slot).  Node 0 holds the root box
// as child 0 and an empty leaf as child 1, so a traversal always starts with `cur = 0`.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "rt_flatten.h"

namespace rtb {

struct Node {
    float lo0[3];
    uint32_t ref0;
    float hi0[3];
    uint32_t ref1;
    float lo1[3];
    uint32_t pad0;
    float hi1[3];
    uint32_t pad1;
};
static_assert(sizeof(Node) == 64, "BVH node is 64 B");

constexpr uint32_t LEAF = 0x80000000u;
constexpr int MAX_LEVELS = 32;  // inner nodes on any root->leaf path, holder included (= stack entries)

struct Bvh {
    std::vector<Node> nodes;
    int levels = 0;  // deepest chain of inner nodes actually built (<= MAX_LEVELS)
    int64_t leaves = 0;
};

namespace detail {

struct Item {
    float lo[3], hi[3];
    uint32_t ref;
    float weight;  // primitives in the leaf
};
struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; k++) lo[k] = 3.402823466e+38f, hi[k] = -3.402823466e+38f;
    }
    void grow(const float* l, const float* h) {
        for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], l[k]), hi[k] = std::max(hi[k], h[k]);
    }
    double half_area() const {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
};

struct Builder {
    std::vector<Item>& items;
    Bvh& out;
    static constexpr int BINS = 16;

    // builds items[b, e), returns its ref and box; `room` = inner levels still available
    uint32_t build(size_t b, size_t e, int level, Box* box) {
        box->reset();
        if (e - b == 1) {
            box->grow(items[b].lo, items[b].hi);
            return items[b].ref;
        }
        const int room = MAX_LEVELS - level;  // levels left including this node: n <= 2^room holds
        const size_t n = e - b;
        const size_t child_cap = (room - 1 >= 31) ? (size_t)1 << 31 : (size_t)1 << (room - 1);
        Box cb;  // centroid bounds
        cb.reset();
        for (size_t i = b; i < e; i++) {
            float c[3];
            for (int k = 0; k < 3; k++) c[k] = 0.5f * items[i].lo[k] + 0.5f * items[i].hi[k];
            cb.grow(c, c);
        }
        int best_axis = -1, best_bin = -1;
        double best_cost = 1e300;
        double scale[3];
        for (int ax = 0; ax < 3; ax++) {
            const double ext = (double)cb.hi[ax] - cb.lo[ax];
            scale[ax] = ext > 0 ? BINS / ext : 0;
            if (!(ext > 0)) continue;
            Box bb[BINS];
            double bw[BINS];
            size_t bn[BINS];
            for (int k = 0; k < BINS; k++) bb[k].reset(), bw[k] = 0, bn[k] = 0;
            for (size_t i = b; i < e; i++) {
                const int k = bin_of(items[i], ax, cb.lo[ax], scale[ax]);
                bb[k].grow(items[i].lo, items[i].hi);
                bw[k] += items[i].weight;
                bn[k]++;
            }
            double right_cost[BINS];
            Box acc;
            acc.reset();
            double w = 0;
            for (int k = BINS - 1; k > 0; k--) {
                if (bn[k]) acc.grow(bb[k].lo, bb[k].hi);
                w += bw[k];
                right_cost[k] = w > 0 ? acc.half_area() * w : 0;
            }
            acc.reset();
            w = 0;
            size_t nl = 0;
            for (int k = 0; k + 1 < BINS; k++) {
                if (bn[k]) acc.grow(bb[k].lo, bb[k].hi);
                w += bw[k];
                nl += bn[k];
                if (nl == 0 || nl == n) continue;
                if (nl > child_cap || n - nl > child_cap) continue;  // would break the level bound
                const double cost = acc.half_area() * w + right_cost[k + 1];
                if (cost < best_cost) best_cost = cost, best_axis = ax, best_bin = k;
            }
        }
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
            std::nth_element(items.begin() + b, items.begin() + mid, items.begin() + e,
                             [ax](const Item& x, const Item& y) { return x.lo[ax] + x.hi[ax] < y.lo[ax] + y.hi[ax]; });
        }
        const uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        if (level + 1 > out.levels) out.levels = level + 1;
        Box l, r;
        const uint32_t rl = build(b, mid, level + 1, &l);
        const uint32_t rr = build(mid, e, level + 1, &r);
        Node& nd = out.nodes[me];
        for (int k = 0; k < 3; k++) nd.lo0[k] = l.lo[k], nd.hi0[k] = l.hi[k], nd.lo1[k] = r.lo[k], nd.hi1[k] = r.hi[k];
        nd.ref0 = rl, nd.ref1 = rr, nd.pad0 = nd.pad1 = 0;
        box->grow(l.lo, l.hi);
        box->grow(r.lo, r.hi);
        return me;
    }
    static int bin_of(const Item& it, int ax, float lo, double scale) {
        const double c = 0.5f * it.lo[ax] + 0.5f * it.hi[ax];
        int k = (int)((c - lo) * scale);
        return k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
    }
};

}  // namespace detail

// Re-tree the leaves of a flattened reference tree (flat.nested must hold).  Leaves without
// primitives are dropped (they have no candidates).  An empty tree yields no nodes.
inline void build(const rtf::Flat& flat, Bvh* out) {
    out->nodes.clear();
    out->levels = 0;
    out->leaves = 0;
    std::vector<detail::Item> items;
    for (const rtf::FlatNode& fn : flat.nodes) {
        const int kind = (fn.meta >> 29) & 3, count = (fn.meta >> 24) & 31;
        if (kind == rtf::KIND_BRANCH || count == 0) continue;
        detail::Item it;
        it.lo[0] = fn.minx, it.lo[1] = fn.miny, it.lo[2] = fn.minz;
        it.hi[0] = fn.maxx, it.hi[1] = fn.maxy, it.hi[2] = fn.maxz;
        it.ref = LEAF | (fn.meta & 0x7FFFFFFFu);
        it.weight = (float)count;
        items.push_back(it);
    }
    out->leaves = (int64_t)items.size();
    if (items.empty()) return;
    out->nodes.reserve(items.size() + 1);
    out->nodes.emplace_back();  // node 0: root holder
    out->levels = 1;
    detail::Builder bld{items, *out};
    detail::Box root;
    const uint32_t ref = bld.build(0, items.size(), 1, &root);
    Node& h = out->nodes[0];
    for (int k = 0; k < 3; k++) {
        h.lo0[k] = root.lo[k], h.hi0[k] = root.hi[k];
        h.lo1[k] = 3.402823466e+38f, h.hi1[k] = -3.402823466e+38f;  // inverted: rejected by any finite ray
    }
    h.ref0 = ref;
    h.ref1 = LEAF;  // empty leaf (count 0): harmless even if its box were accepted
    h.pad0 = h.pad1 = 0;
}

}  // namespace rtb
