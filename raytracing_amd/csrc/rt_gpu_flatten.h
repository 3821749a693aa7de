// rt_gpu_flatten.h -- rt_flatten.h's flatten() on the GPU, for large accel arrays.
//
// Same input (the reference's serialised R-tree, RTree.cs:111-178), same output (rtf::Flat: records in the
// shader's visit order with skip links, slot -> primitive maps, parent links, nested flag), bit for bit --
// tests/test_gpu_flatten.py compares the two on every scene class and on hand-made arrays.  The host version is
// a depth-first walk, one node after the other (Soup1M: 579 k records, 50-90 ms on one core); here the tree is
// discovered level by level and the visit order is computed, not walked:
//
//   1. discovery (k_discover, one launch per level, ~20 levels): every node of the frontier checks its header
//      exactly as flatten() does, reserves ids for its children (one atomic) and writes their positions; a leaf
//      counts its sphere and triangle ids.  Child offsets point strictly forward (flatten() demands
//      off >= 8 + count), so there are no cycles; shared children are caught by the node budget (len/8 + 1 ids).
//   2. subtree sums, deepest level first (k_sums): flat records (a branch is one record, a leaf one per 31 slots
//      and primitive class), sphere slots and triangle slots below every node.
//   3. positions, root first (k_place): the shader pushes children 0..n-1 and pops the last first, so child k's
//      first record is parent + 1 + sum of the subtrees of the children after it; slots are numbered in the same
//      order.
//   4. k_emit writes every node's records, slot ids and parent links at their final places.
//
// Anything flatten() would refuse raises a flag; the caller then runs flatten() on the host, which reports the
// first problem in ITS visit order with its message -- so malformed input behaves exactly as before and this file
// never has to explain an error.  Every index read from the array is bounds-checked before use: a hostile array
// cannot make a kernel read or write outside its buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "rt_flatten.h"

namespace rtgf {

struct Work {  // per logical node id (ids of one level are consecutive; a node's children are consecutive)
    int32_t* pos;          // first float of the node's record in the accel array
    int32_t* parent;       // logical id of the parent, -1 for the root
    int32_t* count;        // children (branch) or primitive ids (leaf); 0 if the node was refused
    int32_t* child_first;  // branch: logical id of child 0
    int32_t* nsph;         // leaf: sphere ids / triangle ids in its list; -1 = branch
    int32_t* ntri;
    int32_t* sub_rec;      // flat records / sphere slots / triangle slots of the subtree
    int32_t* sub_sph;
    int32_t* sub_tri;
    int32_t* fpos;         // first flat record of the node
    int32_t* sph_first;    // first sphere / triangle slot of the subtree
    int32_t* tri_first;
};
constexpr int N_WORK = 12;
enum { FLAG_REFUSED = 1, FLAG_NOT_NESTED = 2 };

__device__ __forceinline__ bool exact_index_dev(float f, int64_t* out) {
    if (!(f >= 0.0f) || !(f < 16777216.0f)) return false;
    const int64_t v = (int64_t)f;
    if ((float)v != f) return false;
    *out = v;
    return true;
}
__host__ __device__ inline int leaf_pieces(int nsph, int ntri) {
    const int ps = nsph > 0 ? (nsph + rtf::MAX_LEAF_SLOTS - 1) / rtf::MAX_LEAF_SLOTS : 0;
    const int pt = (ntri > 0 || nsph == 0) ? (ntri > 0 ? (ntri + rtf::MAX_LEAF_SLOTS - 1) / rtf::MAX_LEAF_SLOTS : 1) : 0;
    return ps + pt;
}

// state[0] = next free logical id, state[1] = flags
__global__ void k_discover(const float* __restrict__ acc, int64_t len, int64_t n_spheres, int64_t n_tris, Work w, uint32_t fb,
                           uint32_t fe, uint32_t budget, uint32_t* state) {
    const uint32_t id = fb + blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= fe) return;
    w.count[id] = 0, w.child_first[id] = 0, w.nsph[id] = 0, w.ntri[id] = 0;
    const int64_t pos = w.pos[id];
    if (pos < 0 || pos + 8 > len) {
        atomicOr(&state[1], FLAG_REFUSED);
        return;
    }
    const float* n = acc + pos;
    int64_t count;
    if (!exact_index_dev(n[7], &count) || pos + 8 + count > len) {
        atomicOr(&state[1], FLAG_REFUSED);
        return;
    }
    if (n[6] == 0.0f) {  // branch
        w.nsph[id] = -1;
        if (count >= rtf::SLOT_LIMIT) {
            atomicOr(&state[1], FLAG_REFUSED);
            return;
        }
        const uint32_t cf = atomicAdd(&state[0], (uint32_t)count);
        if ((uint64_t)cf + (uint64_t)count > (uint64_t)budget) {  // more nodes than the array can hold
            atomicOr(&state[1], FLAG_REFUSED);
            return;
        }
        w.count[id] = (int32_t)count, w.child_first[id] = (int32_t)cf;
        for (int64_t i = 0; i < count; i++) {
            int64_t off;
            int32_t child_pos = -1;
            if (!exact_index_dev(n[8 + i], &off) || off < 8 + count || pos + off + 8 > len)
                atomicOr(&state[1], FLAG_REFUSED);
            else
                child_pos = (int32_t)(pos + off);
            w.pos[cf + i] = child_pos, w.parent[cf + i] = (int32_t)id;
        }
    } else {  // leaf
        int ns = 0, nt = 0;
        for (int64_t i = 0; i < count; i++) {
            int64_t pid;
            if (!exact_index_dev(n[8 + i], &pid) || pid >= n_spheres + n_tris) {
                atomicOr(&state[1], FLAG_REFUSED);
                return;
            }
            if (pid >= n_spheres)
                nt++;
            else
                ns++;
        }
        w.count[id] = (int32_t)count, w.nsph[id] = ns, w.ntri[id] = nt;
    }
}

__global__ void k_sums(Work w, uint32_t fb, uint32_t fe, uint32_t* state) {
    const uint32_t id = fb + blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= fe) return;
    int64_t rec, sph, tri;
    if (w.nsph[id] >= 0) {
        rec = leaf_pieces(w.nsph[id], w.ntri[id]), sph = w.nsph[id], tri = w.ntri[id];
    } else {
        rec = 1, sph = 0, tri = 0;
        const int32_t cf = w.child_first[id];
        for (int32_t k = 0; k < w.count[id]; k++) rec += w.sub_rec[cf + k], sph += w.sub_sph[cf + k], tri += w.sub_tri[cf + k];
    }
    // flatten()'s limits: slots below 2^24 per class, records below INT32_MAX - 64
    if (sph >= rtf::SLOT_LIMIT || tri >= rtf::SLOT_LIMIT || rec >= (int64_t)INT32_MAX - 64) {
        atomicOr(&state[1], FLAG_REFUSED);
        rec = 1, sph = 0, tri = 0;
    }
    w.sub_rec[id] = (int32_t)rec, w.sub_sph[id] = (int32_t)sph, w.sub_tri[id] = (int32_t)tri;
}

__global__ void k_place(Work w, uint32_t fb, uint32_t fe) {
    const uint32_t id = fb + blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= fe) return;
    if (id == 0) w.fpos[0] = 0, w.sph_first[0] = 0, w.tri_first[0] = 0;
    if (w.nsph[id] >= 0) return;
    // visit order: the last child first (raytracer_fs.glsl:181-186 pushes 0..n-1, pops from the top)
    int32_t rec = w.fpos[id] + 1, sph = w.sph_first[id], tri = w.tri_first[id];
    const int32_t cf = w.child_first[id];
    for (int32_t k = w.count[id] - 1; k >= 0; k--) {
        const int32_t c = cf + k;
        w.fpos[c] = rec, w.sph_first[c] = sph, w.tri_first[c] = tri;
        rec += w.sub_rec[c], sph += w.sub_sph[c], tri += w.sub_tri[c];
    }
}

__global__ void k_emit(const float* __restrict__ acc, int64_t n_spheres, Work w, uint32_t n_logical, rtf::FlatNode* nodes,
                       int32_t* flat_parent, int32_t* sphere_slot_id, int32_t* tri_slot_id, uint32_t* state) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_logical) return;
    const float* n = acc + w.pos[id];
    rtf::FlatNode fn;
    fn.minx = n[0], fn.miny = n[1], fn.minz = n[2];
    fn.maxx = n[3], fn.maxy = n[4], fn.maxz = n[5];
    const int32_t me = w.fpos[id];
    const int32_t up = w.parent[id];
    {
        bool ok = true;
        for (int k = 0; k < 6; k++) ok = ok && isfinite(n[k]) && fabsf(n[k]) < 1e18f;
        if (ok && up >= 0) {
            const float* u = acc + w.pos[up];
            ok = n[0] >= u[0] && n[1] >= u[1] && n[2] >= u[2] && n[3] <= u[3] && n[4] <= u[4] && n[5] <= u[5];
        }
        if (!ok) atomicOr(&state[1], FLAG_NOT_NESTED);
    }
    const int32_t parent_rec = up >= 0 ? w.fpos[up] : -1;
    if (w.nsph[id] < 0) {
        fn.skip = me + w.sub_rec[id];
        fn.meta = rtf::make_meta(false, rtf::KIND_BRANCH, 0, w.count[id]);
        nodes[me] = fn, flat_parent[me] = parent_rec;
        return;
    }
    const int ns = w.nsph[id], nt = w.ntri[id], count = w.count[id];
    const int pieces = leaf_pieces(ns, nt);
    fn.skip = me + pieces;  // a box miss on the first piece skips the other pieces as well
    // slot ids in the order of the leaf's list, per class
    int32_t s_at = w.sph_first[id], t_at = w.tri_first[id];
    for (int i = 0; i < count; i++) {
        const int64_t pid = (int64_t)n[8 + i];
        if (pid >= n_spheres)
            tri_slot_id[t_at++] = (int32_t)(pid - n_spheres);
        else
            sphere_slot_id[s_at++] = (int32_t)pid;
    }
    int piece = 0;
    for (int done = 0; done < ns; done += rtf::MAX_LEAF_SLOTS, piece++) {
        const int k = ns - done < rtf::MAX_LEAF_SLOTS ? ns - done : rtf::MAX_LEAF_SLOTS;
        fn.meta = rtf::make_meta(piece > 0, rtf::KIND_SPHERE, k, w.sph_first[id] + done);
        nodes[me + piece] = fn, flat_parent[me + piece] = piece > 0 ? -2 : parent_rec;
    }
    if (nt > 0 || ns == 0) {
        int done = 0;
        do {
            const int k = nt - done < rtf::MAX_LEAF_SLOTS ? nt - done : rtf::MAX_LEAF_SLOTS;
            fn.meta = rtf::make_meta(piece > 0, rtf::KIND_TRI, k, w.tri_first[id] + done);
            nodes[me + piece] = fn, flat_parent[me + piece] = piece > 0 ? -2 : parent_rec;
            piece++, done += k;
        } while (done < nt);
    }
}

// One launch and one 8-byte read-back per level: a tree deeper than this (a chain of one-child branches can be
// len / 9 levels deep, and the host walks it in linear time) is left to the host.
constexpr size_t MAX_DEVICE_LEVELS = 1024;

// Returns 1 and fills *out if the array was flattened on the current device, 0 if it must go through flatten()
// on the host instead (refused input, a HIP error, an array too large for 32-bit positions, a tree deeper than
// MAX_DEVICE_LEVELS -- *too_deep then says so).  d_acc: the array on the device (len floats).  Temporary device
// memory: 48 bytes per possible node + the outputs.
inline int flatten_on_device(const float* d_acc, int64_t len, int64_t n_spheres, int64_t n_tris, rtf::Flat* out,
                             bool* too_deep = nullptr) {
    if (too_deep) *too_deep = false;
    if (len < 8 || len >= (int64_t)1 << 31) return 0;
    const uint32_t budget = (uint32_t)(len / 8 + 1);
    int32_t* pool = nullptr;
    uint32_t* state = nullptr;
    rtf::FlatNode* d_nodes = nullptr;
    int32_t *d_parent = nullptr, *d_sph = nullptr, *d_tri = nullptr;
    auto release = [&]() {
        (void)hipFree(pool), (void)hipFree(state), (void)hipFree(d_nodes), (void)hipFree(d_parent), (void)hipFree(d_sph), (void)hipFree(d_tri);
        (void)hipGetLastError();
    };
    if (hipMalloc(&pool, (size_t)budget * N_WORK * 4) != hipSuccess || hipMalloc(&state, 8) != hipSuccess) {
        release();
        return 0;
    }
    Work w;
    int32_t** fields[N_WORK] = {&w.pos,     &w.parent,  &w.count,   &w.child_first, &w.nsph,      &w.ntri,
                                &w.sub_rec, &w.sub_sph, &w.sub_tri, &w.fpos,        &w.sph_first, &w.tri_first};
    for (int k = 0; k < N_WORK; k++) *fields[k] = pool + (size_t)k * budget;
    const int32_t root[2] = {0, -1};
    uint32_t h_state[2] = {1, 0};
    bool ok = hipMemcpy(w.pos, &root[0], 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(w.parent, &root[1], 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(state, h_state, 8, hipMemcpyHostToDevice) == hipSuccess;
    std::vector<std::pair<uint32_t, uint32_t>> levels;
    uint32_t fb = 0, fe = 1;
    const auto grid = [](uint32_t n) { return dim3((n + 255) / 256); };
    while (ok && fb < fe) {
        hipLaunchKernelGGL(k_discover, grid(fe - fb), dim3(256), 0, 0, d_acc, len, n_spheres, n_tris, w, fb, fe, budget, state);
        ok = hipMemcpy(h_state, state, 8, hipMemcpyDeviceToHost) == hipSuccess && !(h_state[1] & FLAG_REFUSED) && h_state[0] <= budget;
        levels.emplace_back(fb, fe);
        fb = fe, fe = h_state[0];
        if (ok && fb < fe && levels.size() >= MAX_DEVICE_LEVELS) {
            if (too_deep) *too_deep = true;
            ok = false;
        }
    }
    const uint32_t n_logical = fe;
    for (size_t l = levels.size(); ok && l-- > 0;)
        hipLaunchKernelGGL(k_sums, grid(levels[l].second - levels[l].first), dim3(256), 0, 0, w, levels[l].first, levels[l].second, state);
    int32_t totals[3] = {0, 0, 0};
    ok = ok && hipMemcpy(&totals[0], w.sub_rec, 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(&totals[1], w.sub_sph, 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(&totals[2], w.sub_tri, 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(h_state, state, 8, hipMemcpyDeviceToHost) == hipSuccess && !(h_state[1] & FLAG_REFUSED);
    if (ok) {
        for (size_t l = 0; l < levels.size(); l++)
            hipLaunchKernelGGL(k_place, grid(levels[l].second - levels[l].first), dim3(256), 0, 0, w, levels[l].first, levels[l].second);
        const size_t n_rec = (size_t)totals[0], n_s = (size_t)totals[1], n_t = (size_t)totals[2];
        ok = hipMalloc(&d_nodes, std::max<size_t>(n_rec, 1) * sizeof(rtf::FlatNode)) == hipSuccess &&
             hipMalloc(&d_parent, std::max<size_t>(n_rec, 1) * 4) == hipSuccess && hipMalloc(&d_sph, std::max<size_t>(n_s, 1) * 4) == hipSuccess &&
             hipMalloc(&d_tri, std::max<size_t>(n_t, 1) * 4) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_emit, grid(n_logical), dim3(256), 0, 0, d_acc, n_spheres, w, n_logical, d_nodes, d_parent, d_sph, d_tri, state);
            out->nodes.resize(n_rec), out->parent.resize(n_rec), out->sphere_slot_id.resize(n_s), out->tri_slot_id.resize(n_t);
            ok = hipMemcpy(out->nodes.data(), d_nodes, n_rec * sizeof(rtf::FlatNode), hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(out->parent.data(), d_parent, n_rec * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                 (n_s == 0 || hipMemcpy(out->sphere_slot_id.data(), d_sph, n_s * 4, hipMemcpyDeviceToHost) == hipSuccess) &&
                 (n_t == 0 || hipMemcpy(out->tri_slot_id.data(), d_tri, n_t * 4, hipMemcpyDeviceToHost) == hipSuccess) &&
                 hipMemcpy(h_state, state, 8, hipMemcpyDeviceToHost) == hipSuccess && hipGetLastError() == hipSuccess;
        }
    }
    if (ok) {
        out->max_depth = (int64_t)levels.size() - 1;
        out->logical_nodes = n_logical;
        out->nested = !(h_state[1] & FLAG_NOT_NESTED);
    }
    release();
    return ok ? 1 : 0;
}

}  // namespace rtgf
