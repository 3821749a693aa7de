// rt_scene_prep.h -- everything rt_upload_scene decides on the host before a byte goes to the GPU:
// flatten + validate the serialised R-tree, check the preconditions of the distance cull, mark the
// triangles the reference accepts on a whole line or plane, build the re-treed leaves and (large
// scenes) the tight tree.  No HIP in here: rt_check_scene runs the same code without a device, which
// is what the CPU tests use.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "rt_bvh.h"
#include "rt_flatten.h"

namespace rtp {

struct Options {
    bool build_tight = true;
    int tight_min_nodes = 1000;  // flat node records from which a scene gets the tight tree
    // the containment check somewhere else (rt_mi355x.hip: on the GPU): (flat, extent, &tri_unbounded, &unbounded_leaves)
    // -> 1 contained, 0 not, -1 declined: the host threads check
    std::function<int(const rtf::Flat&, const float*, std::vector<uint8_t>*, std::vector<int32_t>*)> containment;
    bool defer_trees = false;    // stop before the items of both trees: the caller makes and builds them (rt_gpu_bvh.h)
                                 // or calls make_items_on_host + finish_trees_on_host
};

struct Prepared {
    rtf::Flat flat;
    rtb::Bvh bvh;    // re-treed leaves; empty if the tree is not nested
    rtb::Bvh tight;  // per-primitive tree (inner refs relative to its own node 0); empty if not built
    rtb::QBvh tight_q;  // the same tree as the GPU walks it: 32-byte nodes, boxes on a 16-bit grid (rt_bvh.h quantize)
    std::vector<int32_t> tri_slot_leaf, sphere_slot_leaf;  // slot -> reference leaf record (tight tree only)
    std::vector<uint8_t> tri_unbounded;                    // per triangle slot (rtf::meta_nocull)
    float extent[3] = {0, 0, 0};  // per-axis max |coordinate| of the root box
    bool contained = false;       // nested + primitives inside their leaves + normals match: the cull is allowed
    float tight_e = 0;            // E_build of the tight tree's pads
    float pad_tri = 0, pad_sphere = 0;
    int64_t unbounded = 0;
    // what the two trees are built from (consumed by finish_trees_on_host; with Options::defer_trees left to the caller)
    rtb::Items leaf_items, tight_items;
    bool want_tight = false;
    rtb::TightPads pads;  // pads and near region of the tight items (rt_bvh.h)
    float near_lo[3] = {0, 0, 0}, near_hi[3] = {0, 0, 0};  // = pads.near_lo / near_hi
    // Trees the caller built on its GPU instead (rt_gpu_bvh.h): device arrays in the layouts of bvh.nodes and
    // tight_q.nodes; `bvh`, `tight` and `tight_q` above then stay empty.
    struct DeviceTrees {
        bool on = false;
        const void* bvh_nodes = nullptr;
        size_t n_bvh = 0;
        int bvh_levels = 0;
        const void* tight_q_nodes = nullptr;
        size_t n_tight = 0;
        int tight_levels = 0;
        uint32_t near_root = 0;
        float base[3] = {0, 0, 0}, step[3] = {1, 1, 1};
        // also on the device already (the item kernels read them): the flattened nodes, the caller's sphere / triangle
        // records and slot id maps (in the context's own buffers), and -- if the tight tree was built -- the slot -> leaf maps
        const void* flat_nodes = nullptr;
        bool slot_leaf_on_device = false;
    } dev;
};

// The items of both trees on host threads (side by side).  spheres / triangles: the arrays prepare saw.
inline void make_items_on_host(Prepared* out, const unsigned char* spheres, const unsigned char* triangles) {
    const rtf::Flat& flat = out->flat;
    out->leaf_items = rtb::Items(), out->tight_items = rtb::Items();
    rtb::Crew tight_items;  // (joined whatever happens below; a failure in either thread is rethrown by finish())
    if (out->want_tight)
        tight_items.start([&] {
            rtb::TightInput in;
            static_cast<rtb::TightPads&>(in) = out->pads;
            in.tris = triangles, in.spheres = spheres;
            in.tri_unbounded = &out->tri_unbounded;
            rtb::make_tight_items(flat, in, &out->tight_items, &out->tri_slot_leaf, &out->sphere_slot_leaf);
        });
    if (flat.nested) rtb::make_leaf_items(flat, &out->leaf_items);
    tight_items.finish();
}

// The SAH builds over Prepared::leaf_items / tight_items on host threads (side by side: both builders fork further
// below their first levels, neither reads what the other writes), then the 16-bit grid form of the tight tree.
inline void finish_trees_on_host(Prepared* out) {
    out->bvh = rtb::Bvh();
    out->tight = rtb::Bvh();
    rtb::Crew tight_builder;  // (joined whatever happens below; a failure in either builder is rethrown by finish())
    if (out->want_tight)
        tight_builder.start([&] {
            rtb::detail::build_items(out->tight_items.items, out->tight_items.split, &out->tight, out->tight_items.split2);
            for (int c = 0; c < 3; c++) out->tight.near_lo[c] = out->near_lo[c], out->tight.near_hi[c] = out->near_hi[c];
        });
    if (!out->leaf_items.items.empty()) rtb::detail::build_items(out->leaf_items.items, out->leaf_items.split, &out->bvh);
    if (out->want_tight) {
        tight_builder.finish();
        if (out->bvh.nodes.empty() || out->tight.levels > rtb::MAX_LEVELS) {  // (no leaf holds a primitive / too deep)
            out->tight = rtb::Bvh();
            out->tri_slot_leaf.clear(), out->sphere_slot_leaf.clear();
        }
    }
    out->tight_q = rtb::QBvh();
    if (!out->tight.nodes.empty() && !rtb::quantize(out->tight, &out->tight_q)) out->tight = rtb::Bvh();
    out->leaf_items = rtb::Items(), out->tight_items = rtb::Items();
}

// Is a triangle record (TriangleStruct, 96 B: A|_ B|_ C|_ N|_ ...) accepted on a line or plane?  Its three
// edge tests accept the intersection of three half-planes of the plane through A with the stored normal;
// for vertices that are collinear or coincident as seen along that normal this is a whole line or the
// whole plane, wherever the ray passes the leaf's box test (which has no distance bound).  Width/length
// below 1e-5 is within rounding of that case (the reference's table model has 19 such triangles).
// The same holds, to a degree, for any sliver: rounding lets the edge tests accept points up to
// rtb::sliver_reach past a sharp vertex.  The distance cull (pad E/128) and the tight boxes (which add the reach
// to their pad) assume hits near the primitive, so a triangle whose reach exceeds max_extent / 512 -- a quarter
// of the cull's pad; a sliver with angle below 2.4e-4 * longest edge / max_extent -- is treated as unbounded
// too.  A NaN or zero normal never yields a hit and is not "unbounded".
RT_BVH_HD inline bool triangle_is_unbounded(const float* rec, float max_extent) {
    double ratio, longest;
    if (!rtb::triangle_shape(rec, &ratio, &longest)) return false;
    if (!(ratio > 0) || !(ratio >= 1e-5)) return true;
    return !(rtb::sliver_reach(rec) <= (double)max_extent * (1.0 / 512.0));
}

// Distance cull precondition: boxes nested, every primitive inside its leaf box (within a slack far
// below the cull's pad), stored triangle normals perpendicular to their triangles.  Also collects the
// unbounded triangles.  Returns false if the cull must stay off.
// Leaf records are independent: the upload runs this over node ranges on up to 16 threads (a million primitives
// take 0.17 s on one core).
// One leaf record (shared with the device's form of the check, rt_mi355x.hip k_containment): false = a primitive
// outside its leaf box or a normal that is not the triangle's; on_unbounded(slot) for every unbounded triangle.
template <class OnUnbounded>
RT_BVH_HD inline bool leaf_contained(const rtf::FlatNode& fn, const int32_t* sphere_slot_id, const int32_t* tri_slot_id,
                                     const unsigned char* spheres, const unsigned char* triangles, float slack, float max_extent,
                                     OnUnbounded&& on_unbounded) {
    const int kind = (fn.meta >> 29) & 3, count = (fn.meta >> 24) & 31, first = fn.meta & 0xFFFFFF;
    if (kind == rtf::KIND_BRANCH) return true;
    const float lo[3] = {fn.minx - slack, fn.miny - slack, fn.minz - slack};
    const float hi[3] = {fn.maxx + slack, fn.maxy + slack, fn.maxz + slack};
    for (int k = 0; k < count; k++) {
        if (kind == rtf::KIND_SPHERE) {
            const float* rec = (const float*)(spheres + (size_t)sphere_slot_id[first + k] * 48);
            for (int c = 0; c < 3; c++)
                if (!(rec[c] - rec[3] >= lo[c] && rec[c] + rec[3] <= hi[c]) || !(rec[3] >= 0)) return false;
            continue;
        }
        const float* rec = (const float*)(triangles + (size_t)tri_slot_id[first + k] * 96);
        for (int v = 0; v < 3; v++)
            for (int c = 0; c < 3; c++) {
                const float x = rec[4 * v + c];
                if (!(x >= lo[c] && x <= hi[c])) return false;
            }
        // the stored normal must be the triangle's: IntersectTriangle intersects the plane through A
        // with THAT normal, and the cull's proof needs the hit point on the triangle.  Measured as the
        // distance of B and C from that plane, allowed up to 1/8 of the cull's pad (the reference's own
        // sliver triangles reach 3e-3 rad, 0.003 units).  A NaN normal (zero-area triangle, 32 of them
        // in the teapot) or a zero normal never produces a hit and is fine.
        const float* n = rec + 12;
        const double nn = sqrt((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2]);
        for (int v = 1; v < 3 && nn == nn && nn > 0; v++) {
            const double off = fabs(((double)rec[4 * v] - rec[0]) * n[0] + ((double)rec[4 * v + 1] - rec[1]) * n[1] +
                                    ((double)rec[4 * v + 2] - rec[2]) * n[2]);
            if (!(off <= 4.0 * slack * nn)) return false;
        }
        if (triangle_is_unbounded(rec, max_extent)) on_unbounded(first + k);
    }
    return true;
}
inline bool check_containment_range(const rtf::Flat& flat, const unsigned char* spheres, const unsigned char* triangles,
                                    float slack, float max_extent, size_t n0, size_t n1, std::vector<uint8_t>* tri_unbounded,
                                    std::vector<int32_t>* unbounded_leaves) {
    for (size_t ni = n0; ni < n1; ni++)
        if (!leaf_contained(flat.nodes[ni], flat.sphere_slot_id.data(), flat.tri_slot_id.data(), spheres, triangles, slack, max_extent,
                            [&](int slot) {
                                (*tri_unbounded)[slot] = 1;  // slots of different leaves: no two threads write the same byte
                                unbounded_leaves->push_back((int32_t)ni);
                            }))
            return false;
    return true;
}
inline bool check_containment(const rtf::Flat& flat, const unsigned char* spheres, const unsigned char* triangles,
                              const float extent[3], std::vector<uint8_t>* tri_unbounded,
                              std::vector<int32_t>* unbounded_leaves) {
    if (!flat.nested) return false;
    const float max_extent = std::fmax(extent[0], std::fmax(extent[1], extent[2]));
    const float slack = max_extent * (1.0f / 4096.0f);
    const size_t n = flat.nodes.size();
    const int n_threads = n < (1u << 16) ? 1 : (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (n_threads == 1) return check_containment_range(flat, spheres, triangles, slack, max_extent, 0, n, tri_unbounded, unbounded_leaves);
    std::vector<std::vector<int32_t>> found(n_threads);
    std::vector<char> ok(n_threads, 1);
    rtb::Crew crew;
    for (int t = 0; t < n_threads; t++)
        crew.start([&, t] {
            ok[t] = check_containment_range(flat, spheres, triangles, slack, max_extent, n * t / n_threads, n * (t + 1) / n_threads,
                                            tri_unbounded, &found[t]) ? 1 : 0;
        });
    crew.finish();
    for (int t = 0; t < n_threads; t++) {
        if (!ok[t]) return false;
        unbounded_leaves->insert(unbounded_leaves->end(), found[t].begin(), found[t].end());  // ranges in order: sorted
    }
    return true;
}

// Returns rtf::FLATTEN_* (0 = ok).  spheres / triangles: the uploaded SphereStruct / TriangleStruct arrays.
// flat_given: out->flat already holds the flattened tree (rt_gpu_flatten.h made it on the device).
inline int prepare(const unsigned char* spheres, int n_spheres, const unsigned char* triangles, int n_triangles,
                   const float* accel, int64_t accel_len, const Options& opt, Prepared* out, std::string* err,
                   bool flat_given = false) {
    // RT_PREP_TIMING=1 (development): wall time of each phase on stderr
    const bool timing = std::getenv("RT_PREP_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rt prepare] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    rtf::Flat& flat = out->flat;
    const int rc = flat_given ? (int)rtf::FLATTEN_OK : rtf::flatten(accel, accel_len, n_spheres, n_triangles, &flat, err);
    if (rc != rtf::FLATTEN_OK) return rc;
    lap("validate + flatten");
    for (int k = 0; k < 3; k++) out->extent[k] = 0.0f;
    if (!flat.nodes.empty()) {
        const rtf::FlatNode& root = flat.nodes[0];
        const float lo[3] = {root.minx, root.miny, root.minz}, hi[3] = {root.maxx, root.maxy, root.maxz};
        for (int k = 0; k < 3; k++) out->extent[k] = std::fmax(std::fabs(lo[k]), std::fabs(hi[k]));
    }
    out->tri_unbounded.assign(flat.tri_slot_id.size(), 0);
    std::vector<int32_t> unbounded_leaves;
    int took = opt.containment && flat.nested ? opt.containment(flat, out->extent, &out->tri_unbounded, &unbounded_leaves) : -1;
    if (took < 0) {
        std::fill(out->tri_unbounded.begin(), out->tri_unbounded.end(), 0), unbounded_leaves.clear();
        took = check_containment(flat, spheres, triangles, out->extent, &out->tri_unbounded, &unbounded_leaves) ? 1 : 0;
    }
    out->contained = took == 1;
    if (!out->contained) std::fill(out->tri_unbounded.begin(), out->tri_unbounded.end(), 0);  // no cull: nothing to protect
    out->unbounded = (int64_t)std::count(out->tri_unbounded.begin(), out->tri_unbounded.end(), 1);
    if (out->contained)
        for (int32_t leaf : unbounded_leaves) rtf::mark_unbounded(&flat, leaf);
    lap("cull preconditions");
    // re-treed leaves for the per-lane kernels (only for nested trees, rt_bvh.h) and, side by side on a second
    // thread (both builders fork further below their first levels; neither reads what the other writes), the
    // tight per-primitive tree for the staged pipeline's rays (rt_bvh.h build_tight): large scenes whose
    // primitives were just verified to sit inside their leaves with matching normals
    out->bvh = rtb::Bvh();
    out->tight = rtb::Bvh();
    out->tri_slot_leaf.clear(), out->sphere_slot_leaf.clear();
    const float max_extent = std::fmax(out->extent[0], std::fmax(out->extent[1], out->extent[2]));
    out->tight_e = 4.0f * max_extent;  // origins up to 3 scene radii from the world origin
    out->pad_tri = out->tight_e * (1.0f / 16384.0f), out->pad_sphere = out->tight_e * (1.0f / 256.0f);
    const bool want_tight = opt.build_tight && out->contained && flat.nested && (int64_t)flat.nodes.size() >= opt.tight_min_nodes &&
                            max_extent <= 1e6f;
    out->want_tight = want_tight;
    out->leaf_items = rtb::Items(), out->tight_items = rtb::Items();
    out->pads = rtb::TightPads();
    out->pads.pad_tri = out->pad_tri, out->pads.pad_sphere = out->pad_sphere;
    // origins inside the root box grown by max_extent / 32 (hit points lie within the cull's pad, <= max_extent / 64
    // there, of a primitive) walk the spheres' tree with the per-sphere pads (rt_bvh.h TightPads)
    out->pads.near_on = !flat.nodes.empty();
    if (out->pads.near_on) {
        const rtf::FlatNode& root = flat.nodes[0];
        const float rlo[3] = {root.minx, root.miny, root.minz}, rhi[3] = {root.maxx, root.maxy, root.maxz};
        const float grow = 1.0f / 32.0f;  // (measured on Soup1M 4K: 1/32 48.0 ms, 1/2 47.8, 1 48.5, 2 50.2)
        for (int k = 0; k < 3; k++) out->pads.near_lo[k] = rlo[k] - max_extent * grow, out->pads.near_hi[k] = rhi[k] + max_extent * grow;
    }
    for (int k = 0; k < 3; k++) out->near_lo[k] = out->pads.near_lo[k], out->near_hi[k] = out->pads.near_hi[k];
    if (opt.defer_trees) return rtf::FLATTEN_OK;
    make_items_on_host(out, spheres, triangles);
    lap("items of both trees");
    finish_trees_on_host(out);
    lap("both trees (SAH), 16-bit grid");
    return rtf::FLATTEN_OK;
}

// Structural self-check of a tree built by rt_bvh.h (used by rt_check_scene, i.e. by the CPU tests):
// every inner box is the exact min/max union of its children, every leaf ref is reached exactly once,
// the depth bound holds, never-cull marks are present exactly above kind-3 leaves.  For the tight tree
// also: each primitive's accepted region (A, B', C' / the sphere) lies inside its leaf box shrunk by the
// pad.  Returns an empty string or a description of the first violation.
inline std::string verify_tree(const rtb::Bvh& t, const Prepared* tight_of, const unsigned char* spheres,
                               const unsigned char* triangles) {
    if (t.nodes.empty()) return "";
    std::vector<uint8_t> visited_node(t.nodes.size(), 0);
    // the tight tree may have a second root holder (rtb::Bvh::near_root: the triangles' tree again + the spheres' tree
    // for near origins): every slot must be reached exactly once from EITHER holder, with that holder's pads
    for (int pass = 0; pass < (t.near_root ? 2 : 1); pass++) {
    const bool near = pass == 1;
    rtb::TightInput near_in;
    if (tight_of) {
        near_in.pad_tri = tight_of->pad_tri, near_in.pad_sphere = tight_of->pad_sphere;
        for (int k = 0; k < 3; k++) near_in.near_lo[k] = t.near_lo[k], near_in.near_hi[k] = t.near_hi[k];
    }
    std::vector<uint8_t> seen_tri, seen_sph;
    if (tight_of) seen_tri.assign(tight_of->flat.tri_slot_id.size(), 0), seen_sph.assign(tight_of->flat.sphere_slot_id.size(), 0);
    struct Todo {
        uint32_t ref;
        float lo[3], hi[3];
        int level;
    };
    std::vector<Todo> stack;
    auto push = [&](uint32_t ref, const float* lo, const float* hi, int level) {
        Todo td;
        td.ref = ref, td.level = level;
        for (int k = 0; k < 3; k++) td.lo[k] = lo[k], td.hi[k] = hi[k];
        stack.push_back(td);
    };
    auto push_children = [&](const rtb::Node& nd, int level) {
        for (int c = 0; c < 2; c++) {
            float lo[3], hi[3];
            nd.get_box(c, lo, hi);
            push(nd.ref[c], lo, hi, level);
        }
    };
    const uint32_t holder = near ? t.near_root : 0u;
    if (holder >= t.nodes.size()) return "second root holder out of range";
    push_children(t.nodes[holder], 1);
    visited_node[holder] = 1;
    while (!stack.empty()) {
        const Todo td = stack.back();
        stack.pop_back();
        if (td.ref & rtb::LEAF) {
            const int kind = (td.ref >> 29) & 3, count = (td.ref >> 24) & 31, first = (int)(td.ref & 0xFFFFFF);
            if (!tight_of || count == 0) continue;
            if (count != 1) return "tight leaf with more than one slot";
            if (kind == rtf::KIND_SPHERE) {
                if (seen_sph[first]++) return "sphere slot reached twice";
                const float* rec = (const float*)(spheres + (size_t)tight_of->flat.sphere_slot_id[first] * 48);
                const double sphere_pad = near ? std::min(rtb::near_sphere_pad(rec, near_in), (double)tight_of->pad_sphere) : (double)tight_of->pad_sphere;
                if (near) {  // the pad covers 2.0e-3 |o - c| for every origin of the near region (checked at its corners)
                    double worst = 0;
                    for (int corner = 0; corner < 8; corner++) {
                        double d2 = 0;
                        for (int c = 0; c < 3; c++) {
                            const double x = (corner >> c) & 1 ? t.near_hi[c] : t.near_lo[c];
                            d2 += (x - rec[c]) * (x - rec[c]);
                        }
                        worst = std::max(worst, std::sqrt(d2));
                    }
                    if (!(sphere_pad >= std::min(2.0e-3 * worst, (double)tight_of->pad_sphere))) return "near pad of a sphere below its error bound";
                }
                for (int c = 0; c < 3; c++)
                    if (!((double)rec[c] - rec[3] - sphere_pad >= td.lo[c] - 1e-6 * std::fabs(td.lo[c]) &&
                          (double)rec[c] + rec[3] + sphere_pad <= td.hi[c] + 1e-6 * std::fabs(td.hi[c])))
                        return "sphere sticks out of its padded tight box";
            } else {
                if (seen_tri[first]++) return "triangle slot reached twice";
                if (tight_of->tri_unbounded[first]) {  // its reference leaf box, grown by the pad, marked never-cull
                    const rtf::FlatNode& fn = tight_of->flat.nodes[tight_of->tri_slot_leaf[first]];
                    const float flo[3] = {fn.minx, fn.miny, fn.minz}, fhi[3] = {fn.maxx, fn.maxy, fn.maxz};
                    if (kind != rtf::KIND_TRI_UNBOUNDED) return "unbounded triangle without the never-cull kind";
                    for (int c = 0; c < 3; c++)
                        if (!((double)flo[c] - tight_of->pad_tri >= td.lo[c] - 1e-6 * std::fabs(td.lo[c]) &&
                              (double)fhi[c] + tight_of->pad_tri <= td.hi[c] + 1e-6 * std::fabs(td.hi[c])))
                            return "unbounded triangle's box does not contain its padded reference leaf box";
                    continue;
                }
                const float* rec = (const float*)(triangles + (size_t)tight_of->flat.tri_slot_id[first] * 96);
                const double n[3] = {rec[12], rec[13], rec[14]};
                const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                for (int v = 0; v < 3; v++) {  // A, B', C': the region IntersectTriangle accepts (rt_bvh.h)
                    const double e[3] = {(double)rec[4 * v] - rec[0], (double)rec[4 * v + 1] - rec[1], (double)rec[4 * v + 2] - rec[2]};
                    const double sc = (e[0] * n[0] + e[1] * n[1] + e[2] * n[2]) / nn;
                    const bool project = nn > 0 && sc == sc;
                    for (int c = 0; c < 3; c++) {
                        const double x = project ? (double)rec[4 * v + c] - sc * n[c] : (double)rec[4 * v + c];
                        const double pad = tight_of->pad_tri + rtb::sliver_reach(rec);
                        if (!(x - pad >= td.lo[c] - 1e-6 * std::fabs(td.lo[c]) && x + pad <= td.hi[c] + 1e-6 * std::fabs(td.hi[c])))
                            return "triangle region sticks out of its padded tight box";
                    }
                }
            }
            continue;
        }
        const uint32_t idx = td.ref & rtb::INDEX_MASK;
        if (idx == 0 || idx >= t.nodes.size() || (t.near_root && idx == t.near_root)) return "inner ref out of range";
        if (td.level + 1 > rtb::MAX_LEVELS || td.level + 1 > t.levels) return "deeper than the recorded level count";
        if (visited_node[idx] > (near ? 1 : 0)) return "shared nodes";
        visited_node[idx] = near ? 2 : 1;
        const rtb::Node& nd = t.nodes[idx];
        for (int k = 0; k < 3; k++)
            if (std::fmin(nd.lo[k][0], nd.lo[k][1]) != td.lo[k] || std::fmax(nd.hi[k][0], nd.hi[k][1]) != td.hi[k])
                return "inner box is not the exact union of its children";
        if (rtb::ref_nocull(td.ref) != (rtb::ref_nocull(nd.ref[0]) || rtb::ref_nocull(nd.ref[1]))) return "never-cull mark not propagated";
        push_children(nd, td.level + 1);
    }
    if (tight_of) {
        for (uint8_t s : seen_tri)
            if (s != 1) return "triangle slot missing from the tight tree";
        for (uint8_t s : seen_sph)
            if (s != 1) return "sphere slot missing from the tight tree";
    }
    }  // pass
    for (uint8_t v : visited_node)
        if (!v) return "unreachable nodes";
    return "";
}

// The 32-byte form against the tree it was made from: same refs, and every grid box contains its fp32 box
// (empty slots stay inverted) while growing by less than 1.01 grid steps per side.
inline std::string verify_quantized(const rtb::Bvh& t, const rtb::QBvh& q) {
    if (t.nodes.size() != q.nodes.size()) return "quantised tree has a different node count";
    for (size_t i = 0; i < t.nodes.size(); i++)
        for (int side = 0; side < 2; side++) {
            if (q.nodes[i].ref[side] != t.nodes[i].ref[side]) return "quantised tree has a different ref";
            float lo[3], hi[3];
            double qlo[3], qhi[3];
            t.nodes[i].get_box(side, lo, hi);
            q.get_box(i, side, qlo, qhi);
            const bool empty = !(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]);
            for (int c = 0; c < 3; c++) {
                if (empty) {
                    if (!(qlo[c] > qhi[c])) return "empty child slot is not inverted on the grid";
                    continue;
                }
                if (!(qlo[c] <= lo[c] && qhi[c] >= hi[c])) return "grid box does not contain its fp32 box";
                if (!(lo[c] - qlo[c] <= 1.01 * q.step[c] && qhi[c] - hi[c] <= 1.01 * q.step[c])) return "grid box looser than one step";
            }
        }
    return "";
}

}  // namespace rtp
