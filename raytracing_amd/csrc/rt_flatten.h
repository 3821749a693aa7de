// rt_flatten.h -- validate the reference's serialised R-tree and flatten it for the GPU.
//
// Input format (RTree.cs:111-178, read back by raytracer_fs.glsl:163-199): pre-order records
//   [minX,minY,minZ,maxX,maxY,maxZ, isLeaf(0/1), count, count x (primitive id | child offset)]
// child offsets are relative to the node's own first float (shader line 184: pos + offset).
//
// The shader visits nodes with a stack: children are pushed 0..count-1 and popped last first.
// That visit order is FIXED by the tree (it does not depend on the ray), so the traversal can
// be made stackless: nodes are stored in visit order, a box hit continues at i+1 and a box
// miss jumps to skip[i] (the first node after the subtree).  Leaf primitives are stored in
// the same order ("slots"), which makes a leaf's geometry contiguous in memory and makes the
// slot index the reference's tie-break rank (first candidate of equal t wins).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace rtf {

// One 32-byte node record = two 16-byte vector loads.
struct FlatNode {
    float minx, miny, minz;
    int32_t skip;  // index of the first node after this node's subtree
    float maxx, maxy, maxz;
    uint32_t meta;  // see below
};
static_assert(sizeof(FlatNode) == 32, "node record is 32 B");

// meta layout
//   bit 31      continuation: same logical reference node as the previous record (a mixed or
//               oversized leaf is cut into homogeneous pieces of <= 31 slots)
//   bits 30..29 kind: 0 branch, 1 sphere leaf, 2 triangle leaf, 3 triangle leaf that holds an unbounded
//               triangle (set by the upload, see meta_nocull)
//   bits 28..24 leaf: slot count (0..31); branch: bit 24 = an unbounded triangle somewhere below
//   bits 23..0  leaf: first slot; branch: child count (for the refs_read counter)
constexpr uint32_t META_CONT = 1u << 31;
constexpr int KIND_BRANCH = 0, KIND_SPHERE = 1, KIND_TRI = 2, KIND_TRI_UNBOUNDED = 3;
constexpr uint32_t META_BRANCH_NOCULL = 1u << 24;
// A triangle whose vertices are (nearly) collinear or coincident as seen along its stored normal is
// accepted by the reference on a whole line or plane (its three edge tests degenerate), wherever the
// ray passes the leaf's box test -- which has no distance bound.  Subtrees that hold such a triangle
// must never be skipped by the distance cull (rt_device.h IntersectBoundingBoxCull).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool meta_nocull(uint32_t meta) {
    const uint32_t kind = (meta >> 29) & 3u;
    return kind == (uint32_t)KIND_TRI_UNBOUNDED || (kind == (uint32_t)KIND_BRANCH && (meta & META_BRANCH_NOCULL));
}
constexpr int MAX_LEAF_SLOTS = 31;
constexpr int64_t SLOT_LIMIT = 1 << 24;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t make_meta(bool cont, int kind, int count, int64_t first) {
    return (cont ? META_CONT : 0u) | ((uint32_t)kind << 29) | ((uint32_t)count << 24) | (uint32_t)first;
}

struct Flat {
    std::vector<FlatNode> nodes;
    std::vector<int32_t> tri_slot_id;     // slot -> triangle index (0..T-1)
    std::vector<int32_t> sphere_slot_id;  // slot -> sphere index (0..S-1)
    int64_t max_depth = 0;
    int64_t logical_nodes = 0;
    // every node's box is finite and lies inside its parent's box, component by component (true for
    // every tree RTree.cs builds: UpdateBoundingBox takes the exact min/max of the children,
    // RTree.cs:332-377).  rt_bvh.h may re-tree the leaves only when this holds.
    bool nested = true;
    std::vector<int32_t> parent;  // per record: index of the parent branch, -1 for the root, -2 for continuation records
};

enum { FLATTEN_OK = 0, FLATTEN_MALFORMED = -2, FLATTEN_TOO_LARGE = -6 };

// float -> int as GLSL int(x) for the values the format can hold; false if not an exact
// non-negative integer below 2^24 (the format stores indices as floats, RTree.cs:138-165).
inline bool exact_index(float f, int64_t* out) {
    if (!(f >= 0.0f) || !(f < 16777216.0f)) return false;
    int64_t v = (int64_t)f;
    if ((float)v != f) return false;
    *out = v;
    return true;
}

inline int flatten(const float* acc, int64_t len, int64_t n_spheres, int64_t n_tris, Flat* out, std::string* err) {
    out->nodes.clear();
    out->tri_slot_id.clear();
    out->sphere_slot_id.clear();
    out->max_depth = 0;
    out->logical_nodes = 0;
    out->nested = true;
    out->parent.clear();
    if (len == 0) return FLATTEN_OK;  // empty scene (RTree.cs:114-115): nothing to traverse
    if (len < 8 || acc == nullptr) {
        *err = "accel array shorter than one node header";
        return FLATTEN_MALFORMED;
    }
    struct Pending {
        int64_t pos;
        int32_t parent;
        int32_t depth;
    };
    std::vector<Pending> stack;
    std::vector<int32_t>& parent = out->parent;
    parent.clear();
    std::vector<int32_t> size;  // subtree size in flat records
    stack.push_back({0, -1, 0});
    const int64_t node_budget = len / 8 + 1;  // every logical node owns >= 8 floats
    std::vector<int64_t> sph, tri;
    while (!stack.empty()) {
        Pending p = stack.back();
        stack.pop_back();
        if (p.pos < 0 || p.pos + 8 > len) {
            *err = "node header out of bounds at float " + std::to_string(p.pos);
            return FLATTEN_MALFORMED;
        }
        if (++out->logical_nodes > node_budget) {
            *err = "more nodes than the array can hold (shared or cyclic children)";
            return FLATTEN_MALFORMED;
        }
        if (p.depth > out->max_depth) out->max_depth = p.depth;
        const float* n = acc + p.pos;
        int64_t count;
        if (!exact_index(n[7], &count) || p.pos + 8 + count > len) {
            *err = "bad count at float " + std::to_string(p.pos + 7);
            return FLATTEN_MALFORMED;
        }
        FlatNode fn;
        fn.minx = n[0], fn.miny = n[1], fn.minz = n[2];
        fn.maxx = n[3], fn.maxy = n[4], fn.maxz = n[5];
        fn.skip = 0;
        const int32_t me = (int32_t)out->nodes.size();
        {
            bool ok = true;
            for (int k = 0; k < 6; k++) ok = ok && std::isfinite(n[k]) && std::fabs(n[k]) < 1e18f;
            if (ok && p.parent >= 0) {
                const FlatNode& up = out->nodes[p.parent];
                ok = n[0] >= up.minx && n[1] >= up.miny && n[2] >= up.minz && n[3] <= up.maxx && n[4] <= up.maxy &&
                     n[5] <= up.maxz;
            }
            if (!ok) out->nested = false;
        }
        if (n[6] == 0.0f) {  // branch (raytracer_fs.glsl:180)
            if (count >= SLOT_LIMIT) return FLATTEN_TOO_LARGE;
            fn.meta = make_meta(false, KIND_BRANCH, 0, count);
            out->nodes.push_back(fn);
            parent.push_back(p.parent);
            size.push_back(1);
            for (int64_t i = 0; i < count; i++) {
                int64_t off;
                // a child must start after this node's own header + pointer list
                if (!exact_index(n[8 + i], &off) || off < 8 + count || p.pos + off + 8 > len) {
                    *err = "bad child offset at float " + std::to_string(p.pos + 8 + i);
                    return FLATTEN_MALFORMED;
                }
                stack.push_back({p.pos + off, me, p.depth + 1});
            }
        } else {  // leaf (raytracer_fs.glsl:187-196): id < S is a sphere, else triangle id - S
            sph.clear();
            tri.clear();
            for (int64_t i = 0; i < count; i++) {
                int64_t id;
                if (!exact_index(n[8 + i], &id) || id >= n_spheres + n_tris) {
                    *err = "bad primitive id at float " + std::to_string(p.pos + 8 + i);
                    return FLATTEN_MALFORMED;
                }
                if (id >= n_spheres)
                    tri.push_back(id - n_spheres);
                else
                    sph.push_back(id);
            }
            int pieces = 0;
            auto emit = [&](int kind, const std::vector<int64_t>& ids, std::vector<int32_t>& slots) -> int {
                size_t done = 0;
                do {
                    size_t k = std::min<size_t>(ids.size() - done, (size_t)MAX_LEAF_SLOTS);
                    int64_t first = (int64_t)slots.size();
                    if (first + (int64_t)k >= SLOT_LIMIT) return FLATTEN_TOO_LARGE;
                    fn.meta = make_meta(pieces > 0, kind, (int)k, first);
                    for (size_t j = 0; j < k; j++) slots.push_back((int32_t)ids[done + j]);
                    out->nodes.push_back(fn);
                    parent.push_back(pieces > 0 ? -2 : p.parent);
                    size.push_back(1);
                    pieces++;
                    done += k;
                } while (done < ids.size());
                return FLATTEN_OK;
            };
            int rc = FLATTEN_OK;
            if (!sph.empty()) rc = emit(KIND_SPHERE, sph, out->sphere_slot_id);
            if (rc == FLATTEN_OK && (!tri.empty() || sph.empty())) rc = emit(KIND_TRI, tri, out->tri_slot_id);
            if (rc != FLATTEN_OK) {
                *err = "scene too large for the packed node format";
                return rc;
            }
            // a box miss on the first piece skips the other pieces as well
            for (int k = 0; k < pieces; k++) size[me + k] = pieces - k;
        }
        if ((int64_t)out->nodes.size() >= (int64_t)INT32_MAX - 64) return FLATTEN_TOO_LARGE;
    }
    // subtree sizes: visit order is a pre-order, so children come after their parent
    for (int64_t i = (int64_t)out->nodes.size() - 1; i >= 0; i--) {
        if (parent[i] >= 0) size[parent[i]] += size[i];
    }
    for (size_t i = 0; i < out->nodes.size(); i++) out->nodes[i].skip = (int32_t)(i + size[i]);
    return FLATTEN_OK;
}

// Mark the triangle-leaf record `leaf` and all its ancestors as "never distance-cull" (meta_nocull).
inline void mark_unbounded(Flat* flat, int32_t leaf) {
    FlatNode& fn = flat->nodes[leaf];
    fn.meta = (fn.meta & ~(3u << 29)) | ((uint32_t)KIND_TRI_UNBOUNDED << 29);
    int32_t i = leaf;
    while (flat->parent[i] == -2) i--;  // continuation records hang under the first piece's parent
    for (i = flat->parent[i]; i >= 0; i = flat->parent[i]) {
        if (flat->nodes[i].meta & META_BRANCH_NOCULL) break;  // the rest of the path is marked already
        flat->nodes[i].meta |= META_BRANCH_NOCULL;
    }
}

}  // namespace rtf
