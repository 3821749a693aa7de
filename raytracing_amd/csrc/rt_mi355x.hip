// rt_mi355x.hip -- C ABI of include/rt_mi355x.h: context, scene upload (validate + flatten +
// GPU re-layout), render launches, band un-permute, per-function evaluators.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_bvh.h"
#include "rt_gpu_bvh.h"
#include "rt_gpu_flatten.h"
#include "rt_kernels.h"
#include "rt_scene_prep.h"
#include "rt_wavefront.h"

using namespace rtd;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// Path state of the staged pipeline for rt_render_device_noalloc (include/rt_mi355x.h): a fixed array in
// the code object instead of hipMalloc'ed buffers.  Every GPU the library is loaded on holds its own copy
// at its own address until the process ends.  On the host the name is only a stand-in for that symbol:
// the per-device address comes from hipGetSymbolAddress (rt_create) and is what kernels and memsets get.
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)RT_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

}  // namespace

struct rt_multi;
struct rt_ctx {
    int device = 0;
    rt_multi* multi = nullptr;  // rt_create_multi: this context is the root of a multi-GPU context (rt_multi.h)
    std::string error;
    bool has_scene = false;
    int supertile = 2;  // workgroup tiles per super-tile edge: 2 x 8-pixel tiles = 16 px (block_to_tile).  With the priority
                        // columns: 8 -> 0.246 ms, 4 -> 0.235, 2 -> 0.225, 1 -> 0.229 on the teapot frame (8 was best without)
    DevScene scene{};
    rt_scene_info info{};
    // scene buffers
    DevBuf geom, plane_geom, tri_slot_id, sphere_slot_id, spheres_aos, planes_aos, tris_aos, lights;
    DevBuf tri_slot_leaf, sphere_slot_leaf;  // tight tree only (rt_bvh.h)
    DevBuf accel_tmp;                        // the accel array while rt_gpu_flatten.h works on it (grow only)
    // Which fast kernel renders a frame is decided by measurement where the scene does not tell (tune_begin): the
    // staged pipeline for mid-sized scenes (a triangle soup is 1.8x faster with it at 5.7 k nodes, a field of meshes
    // 3x slower at 18 k), the cooperative shadow rays of the megakernel for any scene with bounces (TestScene1 720p
    // d10 0.50 -> 0.44 ms, teapot 4K 0.58 -> 0.62 ms)
    struct Tune {
        static constexpr int N = 3;    // [0] megakernel, [1] staged pipeline, [2] megakernel with cooperative shadow rays
        long long scene_serial = -1;   // (upload, frame size, bounces) the measurements belong to
        int w = 0, rows = 0, bounces = -1;
        bool cand[N] = {false, false, false};  // the kernels in the running for this key
        float ms[N] = {-1.0f, -1.0f, -1.0f};   // measured GPU time: the faster of two timed renders
        int runs[N] = {0, 0, 0};       // renders started with each: the first is a warm-up (first-use allocation of the
                                       // path state, code load), the next two run between events
        int samples[N] = {0, 0, 0};
        hipEvent_t ev[N][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
        bool pending[N] = {false, false, false};
        bool decided() const {
            for (int k = 0; k < N; k++)
                if (cand[k] && samples[k] < 2) return false;
            return true;
        }
        int best() const {
            int b = 0;
            for (int k = 1; k < N; k++)
                if (cand[k] && ms[k] < ms[b]) b = k;
            return b;
        }
    } tune;
    long long scene_serial = 0;
    bool prioritise = true;   // RT_PRIORITISE=0 (development): deal the tiles out in plain row-major order
    int coop_shadows = 2;  // RT_OPTION_COOP_SHADOWS: the BVH megakernel regroups the shadow rays of sparse bounces over the
                           // wave: 0 never, 1 always, 2 (default) where it measures faster (tune_begin)
    bool build_tight = true;  // RT_OPTION_TIGHT_TREE (takes effect at the next upload and at once for renders)
    int gpu_build_min = 20000;    // RT_OPTION_GPU_BUILD_MIN: scenes whose two trees have at least this many items together get
                                  // them built on the GPU (rt_gpu_bvh.h); -1: never
    DevBuf gb_bvh, gb_tight, gb_tight_q;  // the device builder's results until upload_prepared has copied them
    DevBuf gb_flat, gb_unb;               // ... and two of its inputs: the flattened nodes, the unbounded-triangle flags
    DevBuf gb_list;                       // containment_on_device: leaves that hold an unbounded triangle, its state words
    // what containment_on_device left on the GPU for build_trees_on_device (one upload; cleared when it starts)
    bool staged_records = false, staged_flat = false, staged_unb = false;
    int gpu_flatten_min = 65536;  // RT_OPTION_GPU_FLATTEN_MIN: accel arrays of at least this many floats are flattened on
                                  // the GPU (rt_gpu_flatten.h), smaller ones on the host (a launch per tree level costs
                                  // more than the whole host walk there); -1 = never
    int tight_min_nodes = 1000;  // RT_OPTION_TIGHT_MIN_NODES: scenes with fewer flat nodes get no tight tree (only the
                                 // staged pipeline walks it: measured slower in the megakernel)
    int traversal = RT_TRAVERSAL_AUTO;  // RT_OPTION_TRAVERSAL
    int bvh_min_nodes = 0;              // auto: scenes with at least this many flat nodes take the BVH kernel
                                        // (measured faster than the wave walk on every scene)
    // LDS-resident persistent kernel (k_render_bvh_lds)
    DevBuf ticket;
    unsigned int ticket_base = 0;
    int n_cus = 256;
    bool lds_capable = false;   // triangle/sphere slots + BVH nodes + the per-lane stacks fit the CU's LDS
    size_t lds_bytes = 0;       // dynamic LDS the LDS-resident kernel needs for this scene
    int lds_limit = 160 * 1024;  // bytes of dynamic LDS one workgroup may take (160 KB per CU)
    // order by the previous frame (rt_kernels.h k_rank_supers): marks of the last launch / of this one, the order table,
    // and what the marks belong to (scene, frame size, rows, super-tile size, bounces)
    DevBuf heat[2], order;
    int heat_cur = 0;
    bool heat_valid = false, use_heat = true;  // RT_HEAT=0 (development): off
    long long heat_key[6] = {0, 0, 0, 0, 0, 0};
    bool allow_lds = false;     // RT_OPTION_USE_LDS: opt-in, measured slower than the L2-resident walk
    // path state of the wavefront pipeline (rt_wavefront.h; grow only)
    DevBuf wf_org, wf_dir, wf_mult, wf_final, wf_list0, wf_list1, wf_blocked, wf_counters;
    DevBuf wf_list_tmp, wf_keys, wf_bins;  // binning of the continuing paths (k_wf_bin_*)
    int wf_bin_from = 0;  // lists of iterations >= this are sorted by origin cell (0: none)
    bool wf_packet = true;  // primary rays of the staged pipeline as one packet per tile (RT_WF_PACKET=0, development: off)
    unsigned char* noalloc_scratch = nullptr;  // this device's address of `scratch` (hipGetSymbolAddress, rt_create)
    int wf_min_nodes = 1000;   // scenes with at least this many flat nodes get the tight tree and are candidates for
                               // the staged pipeline under AUTO (decided by measurement, tune_begin)
    int wf_sure_nodes = 100000;  // ... and from here on they take it without asking
    int wf_max_bounces = 64;   // beyond this the per-bounce launches stop paying: megakernel instead
    int wf_waves_per_cu = 24;  // persistent waves of the two trace stages
    // The wavefront path state and the LDS kernel's ticket counter are per context: renders that use
    // them are chained on the device (each waits for the previous one's event), whatever streams the
    // caller launches on.  The megakernels keep no per-context state and overlap freely.
    hipEvent_t chain_done = nullptr;
    bool chain_pending = false;
    // frame buffers of rt_render (grow only)
    DevBuf fb_rgb, fb_rgba8, fb_pid, fb_fid, fb_stats;
    // scratch of the evaluators
    DevBuf ev[8];
};

namespace {

int fail(rt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    return code;
}
int hip_fail(rt_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_DEVICE,
                std::string(what) + ": " + hipGetErrorString(e));
}
// No exception crosses the ABI (rt_mi355x.h): host-side preparation allocates (std::bad_alloc on a scene that does not
// fit host memory) and runs worker threads (rt_bvh.h Crew rethrows their failures here).
template <class F>
int no_throw(rt_ctx* ctx, std::string* msg, F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        if (msg) *msg = "out of host memory";
        return fail(ctx, RT_ERR_OUT_OF_MEMORY, "out of host memory");
    } catch (const std::exception& e) {
        if (msg) *msg = std::string("internal error: ") + e.what();
        return fail(ctx, RT_ERR_DEVICE, std::string("internal error: ") + e.what());
    } catch (...) {
        if (msg) *msg = "internal error";
        return fail(ctx, RT_ERR_DEVICE, "internal error");
    }
}
#define HIP_TRY(ctx, expr)                                     \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #expr); \
    } while (0)

int ensure(rt_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.bytes && b.p) return RT_OK;
    if (b.p) {
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    if (bytes == 0) bytes = 16;
    HIP_TRY(ctx, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return RT_OK;
}
int upload(rt_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
    int rc = ensure(ctx, b, bytes);
    if (rc != RT_OK) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}
void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

// Which fast kernel renders this scene (RT_TRAVERSAL_WAVE / BVH / WAVEFRONT) under the context's
// RT_OPTION_TRAVERSAL: the re-treed kernels need the BVH (nested trees only); the staged pipeline is
// for large scenes and a bounded number of bounces.
int pick_traversal(const rt_ctx* ctx, const DevScene& sc, int max_bounces) {
    const bool bvh_ok = sc.bvh != nullptr && sc.bvh_levels <= rtb::MAX_LEVELS;
    if (!bvh_ok || ctx->traversal == RT_TRAVERSAL_WAVE) return RT_TRAVERSAL_WAVE;
    const bool wf_ok = max_bounces <= ctx->wf_max_bounces;
    if (ctx->traversal == RT_TRAVERSAL_WAVEFRONT) return wf_ok ? RT_TRAVERSAL_WAVEFRONT : RT_TRAVERSAL_BVH;
    if (ctx->traversal == RT_TRAVERSAL_BVH) return RT_TRAVERSAL_BVH;
    if (wf_ok && sc.n_nodes >= std::max(ctx->wf_sure_nodes, ctx->wf_min_nodes)) return RT_TRAVERSAL_WAVEFRONT;  // between: tune_begin
    return sc.n_nodes >= ctx->bvh_min_nodes ? RT_TRAVERSAL_BVH : RT_TRAVERSAL_WAVE;
}

int camera_size(const float* cam, int* w, int* h) {
    if (!cam) return RT_ERR_INVALID_ARGUMENT;
    const float fw = cam[12], fh = cam[13];
    if (!(fw >= 1.0f) || !(fh >= 1.0f) || !(fw <= 65536.0f) || !(fh <= 65536.0f)) return RT_ERR_INVALID_ARGUMENT;
    if (fw != (float)(int)fw || fh != (float)(int)fh) return RT_ERR_INVALID_ARGUMENT;
    *w = (int)fw;
    *h = (int)fh;
    return RT_OK;
}

// Order a render that uses per-context device state behind the previous such render (see rt_ctx).
int chain_begin(rt_ctx* ctx, hipStream_t st) {
    if (ctx->chain_pending) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->chain_done, 0));
    return RT_OK;
}
int chain_end(rt_ctx* ctx, hipStream_t st) {
    HIP_TRY(ctx, hipEventRecord(ctx->chain_done, st));
    ctx->chain_pending = true;
    return RT_OK;
}

// The staged pipeline of rt_wavefront.h: per bounce one closest-hit stage, one shadow stage, one
// shading stage, all on `st`; list sizes stay on the device (persistent / grid-stride kernels read
// them), so the host never synchronises.
// Bytes of path state one render of the staged pipeline needs, laid out as consecutive 256-byte aligned
// blocks: org | dir | mult | final (16 B per local pixel each), two index lists (4 B per pixel), one byte
// per (work item of iteration 0, light), the per-iteration counters.
struct WfLayout {
    size_t org, dir, mult, final_, list0, list1, blocked, counters, total, n_counters;
};
WfLayout wf_layout(size_t n_px, size_t n0, int n_lights, int max_bounces) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    WfLayout l;
    size_t at = 0;
    l.org = at, at += up(n_px * 16);
    l.dir = at, at += up(n_px * 16);
    l.mult = at, at += up(n_px * 16);
    l.final_ = at, at += up(n_px * 16);
    l.list0 = at, at += up(n_px * 4);
    l.list1 = at, at += up(n_px * 4);
    l.blocked = at, at += up(n0 * (size_t)std::max(n_lights, 1));
    l.n_counters = 4 * ((size_t)max_bounces + 2);
    l.counters = at, at += up(l.n_counters * 4);
    l.total = at;
    return l;
}

// noalloc: the path state lives in this device's static `scratch` array (rt_render_device_noalloc) instead
// of the context's grow-only buffers; a frame that does not fit is rejected, nothing is allocated.
int launch_wavefront(rt_ctx* ctx, const RenderArgs& r, hipStream_t st, bool noalloc = false) {
    WfArgs a;
    a.scene = r.scene;
    if (!ctx->build_tight) a.scene.tight = nullptr;
    std::memcpy(a.cam, r.cam, sizeof a.cam);
    a.max_bounces = r.max_bounces;
    a.width = r.width, a.height = r.height, a.local_rows = r.local_rows;
    a.band_rows = r.band_rows, a.rank = r.rank, a.world = r.world;
    a.tiles_x = (r.width + 7) / 8, a.tiles_y = (r.local_rows + 7) / 8;
    a.fd_band = make_fastdiv((uint32_t)r.band_rows), a.fd_tiles_x = make_fastdiv((uint32_t)a.tiles_x);
    a.n_lights = r.scene.lights_len / 6;
    a.iter = 0;
    const size_t n0 = (size_t)a.tiles_x * a.tiles_y * 64;
    if (n0 * (size_t)std::max(a.n_lights, 1) >= 0xFFFFFFF0ull)
        return fail(ctx, RT_ERR_TOO_LARGE, "frame x lights exceeds the 32-bit work index of the wavefront pipeline");
    a.n0 = (uint32_t)n0;
    a.rgb = r.rgb, a.rgba8 = r.rgba8, a.primary_id = r.primary_id, a.final_id = r.final_id, a.bounces = r.bounces;
    const size_t n_px = (size_t)r.width * r.local_rows;
    const WfLayout lay = wf_layout(n_px, n0, a.n_lights, r.max_bounces);
    int rc;
    if (noalloc) {
        if (!ctx->noalloc_scratch) return fail(ctx, RT_ERR_DEVICE, "the static scratch array has no address on this device");
        if (lay.total > NOALLOC_SCRATCH_BYTES)
            return fail(ctx, RT_ERR_TOO_LARGE,
                        "rt_render_device_noalloc: this frame needs " + std::to_string(lay.total) +
                            " bytes of path state, the static scratch array holds " + std::to_string(NOALLOC_SCRATCH_BYTES));
        unsigned char* const base = ctx->noalloc_scratch;
        a.org_t = (float4*)(base + lay.org), a.dir_hit = (float4*)(base + lay.dir);
        a.mult_b = (float4*)(base + lay.mult), a.final_ = (float4*)(base + lay.final_);
        a.list[0] = (uint32_t*)(base + lay.list0), a.list[1] = (uint32_t*)(base + lay.list1);
        a.blocked = base + lay.blocked;
        a.counters = (uint32_t*)(base + lay.counters);
    } else {
        if ((rc = ensure(ctx, ctx->wf_org, n_px * 16)) || (rc = ensure(ctx, ctx->wf_dir, n_px * 16)) ||
            (rc = ensure(ctx, ctx->wf_mult, n_px * 16)) || (rc = ensure(ctx, ctx->wf_final, n_px * 16)) ||
            (rc = ensure(ctx, ctx->wf_list0, n_px * 4)) || (rc = ensure(ctx, ctx->wf_list1, n_px * 4)) ||
            (rc = ensure(ctx, ctx->wf_blocked, n0 * (size_t)std::max(a.n_lights, 1))) ||
            (rc = ensure(ctx, ctx->wf_counters, lay.n_counters * 4)))
            return rc;
        a.org_t = (float4*)ctx->wf_org.p, a.dir_hit = (float4*)ctx->wf_dir.p;
        a.mult_b = (float4*)ctx->wf_mult.p, a.final_ = (float4*)ctx->wf_final.p;
        a.list[0] = (uint32_t*)ctx->wf_list0.p, a.list[1] = (uint32_t*)ctx->wf_list1.p;
        a.blocked = (unsigned char*)ctx->wf_blocked.p;
        a.counters = (uint32_t*)ctx->wf_counters.p;
    }
    // (the static array's layout and frame limit stay as documented: a noalloc render does not bin)
    a.bin_from = noalloc || ctx->wf_bin_from > r.max_bounces ? 0 : ctx->wf_bin_from;
    a.list_tmp = a.keys = a.bins = nullptr;
    if (a.bin_from > 0) {
        if ((rc = ensure(ctx, ctx->wf_list_tmp, n_px * 4)) || (rc = ensure(ctx, ctx->wf_keys, n_px * 4)) ||
            (rc = ensure(ctx, ctx->wf_bins, (size_t)WF_BINS * 4)))
            return rc;
        a.list_tmp = (uint32_t*)ctx->wf_list_tmp.p, a.keys = (uint32_t*)ctx->wf_keys.p, a.bins = (uint32_t*)ctx->wf_bins.p;
        for (int c = 0; c < 3; c++) {
            const float e = a.scene.extent[c];
            a.bin_scale[c] = e > 0 && e < 1e30f ? (float)(1 << WF_BIN_BITS) / (2 * e) : 0.0f;
        }
    }
    // (noalloc: no event work at all -- the caller orders its calls on one GPU, rt_mi355x.h -- so the whole render is
    // memset + kernel launches on `st` and can be recorded into a HIP graph)
    if (!noalloc && (rc = chain_begin(ctx, st))) return rc;
    // the only part of the state that must start at zero (the array keeps the previous call's contents)
    HIP_TRY(ctx, hipMemsetAsync(a.counters, 0, lay.n_counters * 4, st));
    const int trace_grid = ctx->n_cus * ctx->wf_waves_per_cu;
    const size_t lds = (size_t)64 * std::max(a.scene.bvh_levels, 1) * 4;
    const int shade_grid = ctx->n_cus * 8;
    // primary rays as one packet per tile (k_wf_closest_packet): the tight tree, and a camera its pads cover
    bool packet = ctx->wf_packet && a.scene.tight != nullptr;
    if (packet) {
        float E = 0;
        for (int k = 0; k < 3; k++) E = std::fmax(E, std::fabs(a.cam[k]) + a.scene.extent[k]);
        packet = E <= a.scene.tight_e;
    }
    for (int i = 0; i <= r.max_bounces; i++) {
        a.iter = i;
        // iteration 0 knows its size; later lists are at most as long as the frame
        const int g0 = (int)std::min<size_t>((size_t)trace_grid, (n0 + 63) / 64);
        // <true>: the scene holds unbounded triangles, refs are checked for the never-cull mark
        const bool unb = ctx->info.unbounded_triangles > 0;
        if (i == 0 && packet)
            hipLaunchKernelGGL(unb ? k_wf_closest_packet<true> : k_wf_closest_packet<false>, dim3((unsigned)(n0 / 64)), dim3(64), 0, st, a);
        else
            hipLaunchKernelGGL(unb ? k_wf_closest<true> : k_wf_closest<false>, dim3(i == 0 ? g0 : trace_grid), dim3(64), lds, st, a);
        if (a.n_lights > 0)
            hipLaunchKernelGGL(unb ? k_wf_any<true> : k_wf_any<false>, dim3(trace_grid), dim3(64), lds, st, a);
        const bool bin = a.bin_from > 0 && i + 1 >= a.bin_from && i < r.max_bounces;  // this iteration's output list
        if (bin) HIP_TRY(ctx, hipMemsetAsync(a.bins, 0, (size_t)WF_BINS * 4, st));
        hipLaunchKernelGGL(k_wf_shade, dim3(shade_grid), dim3(256), 0, st, a);
        if (bin) {
            hipLaunchKernelGGL(k_wf_bin_scan, dim3(1), dim3(1024), 0, st, a);
            hipLaunchKernelGGL(k_wf_bin_scatter, dim3(shade_grid), dim3(256), 0, st, a);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return noalloc ? (int)RT_OK : chain_end(ctx, st);
}

}  // namespace

// rt_multi.h (included at the end of this file)
static int multi_upload(rt_ctx* root, const rtp::Prepared& prep, const void* spheres, int n_spheres, const void* planes,
                        int n_planes, const void* triangles, int n_triangles, const float* lights, int lights_len);
static int multi_render(rt_ctx* root, const float* camera, int max_bounces, float* rgb, uint8_t* rgba8, int32_t* primary_id,
                        int32_t* final_id, rt_stats* stats);
static int multi_render_device(rt_ctx* root, const float* camera, int max_bounces, float* d_rgb, uint8_t* d_rgba8,
                               int32_t* d_primary_id, int32_t* d_final_id, void* stream);
static void multi_release(rt_ctx* root);

extern "C" {

int rt_local_rows(int height, int band_rows, int rank, int world) {
    if (height <= 0 || band_rows <= 0 || world <= 0 || rank < 0 || rank >= world) return 0;
    const int n_bands = (height + band_rows - 1) / band_rows;
    int rows = 0;
    for (int b = rank; b < n_bands; b += world) rows += std::min(band_rows, height - b * band_rows);
    return rows;
}

int rt_padded_rows(int height, int band_rows, int world) {
    if (height <= 0 || band_rows <= 0 || world <= 0) return 0;
    const int n_bands = (height + band_rows - 1) / band_rows;
    return ((n_bands + world - 1) / world) * band_rows;
}

int rt_create(int device_id, rt_ctx** out) {
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device_id < 0 || device_id >= n) return RT_ERR_DEVICE;
    rt_ctx* ctx = new rt_ctx();
    ctx->device = device_id;
    if (const char* v = getenv("RT_SUPERTILE")) {
        const int st = atoi(v);
        if (st >= 1 && st <= 64) ctx->supertile = st;
    }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        delete ctx;
        return RT_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        if (prop.multiProcessorCount > 0) ctx->n_cus = prop.multiProcessorCount;
        if (prop.maxSharedMemoryPerMultiProcessor > 0)
            ctx->lds_limit = std::min<int>(ctx->lds_limit, (int)prop.maxSharedMemoryPerMultiProcessor);
    }
    if (const char* v = getenv("RT_USE_LDS")) ctx->allow_lds = atoi(v) != 0;
    if (const char* v = getenv("RT_WF_WAVES")) {  // development: persistent waves per CU of the trace stages
        const int t = atoi(v);
        if (t >= 1 && t <= 64) ctx->wf_waves_per_cu = t;
    }
    if (const char* v = getenv("RT_WF_PACKET")) ctx->wf_packet = atoi(v) != 0;  // development
    if (const char* v = getenv("RT_WF_BIN_FROM")) ctx->wf_bin_from = std::max(atoi(v), 0);  // development
    if (const char* v = getenv("RT_PRIORITISE")) ctx->prioritise = atoi(v) != 0;
    if (const char* v = getenv("RT_TIGHT")) ctx->build_tight = atoi(v) != 0;  // development overrides
    if (const char* v = getenv("RT_GPU_FLATTEN_MIN")) ctx->gpu_flatten_min = atoi(v);
    if (const char* v = getenv("RT_GPU_BUILD_MIN")) ctx->gpu_build_min = std::max(atoi(v), -1);
    if (const char* v = getenv("RT_COOP")) ctx->coop_shadows = std::min(std::max(atoi(v), 0), 2);
    if (const char* v = getenv("RT_HEAT")) ctx->use_heat = atoi(v) != 0;
    if (const char* v = getenv("RT_WF_MIN_NODES")) ctx->wf_min_nodes = std::max(0, atoi(v));
    if (const char* v = getenv("RT_TRAVERSAL")) {  // development override of RT_OPTION_TRAVERSAL
        const int t = atoi(v);
        if (t >= RT_TRAVERSAL_AUTO && t <= RT_TRAVERSAL_WAVEFRONT) ctx->traversal = t;
    }
    // set-up that later calls must not repeat: the event that chains renders sharing per-context state, and
    // this device's address of the static scratch array (device `device_id` is current here)
    void* sym = nullptr;
    if (hipEventCreateWithFlags(&ctx->chain_done, hipEventDisableTiming) != hipSuccess ||
        hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)) != hipSuccess || !sym) {
        (void)hipGetLastError();
        if (ctx->chain_done) (void)hipEventDestroy(ctx->chain_done);
        delete ctx;
        return RT_ERR_DEVICE;
    }
    ctx->noalloc_scratch = (unsigned char*)sym;
    // the persistent kernel's ticket counter: zeroed once, never reset (rt_kernels.h)
    if (hipMalloc(&ctx->ticket.p, 256) != hipSuccess || hipMemset(ctx->ticket.p, 0, 256) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_render_bvh_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                            ctx->lds_limit) != hipSuccess) {
        (void)hipGetLastError();
        ctx->allow_lds = false;  // the global-memory kernel needs none of this
    } else {
        ctx->ticket.bytes = 256;
    }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx* ctx) {
    if (!ctx) return;
    if (ctx->multi) multi_release(ctx);  // the other devices' contexts, streams, communicators
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->chain_done) (void)hipEventDestroy(ctx->chain_done);
    for (auto& pair : ctx->tune.ev)
        for (hipEvent_t e : pair)
            if (e) (void)hipEventDestroy(e);
    DevBuf* wf[] = {&ctx->wf_org,   &ctx->wf_dir,   &ctx->wf_mult,    &ctx->wf_final,
                    &ctx->wf_list0, &ctx->wf_list1, &ctx->wf_blocked, &ctx->wf_counters,
                    &ctx->wf_list_tmp, &ctx->wf_keys, &ctx->wf_bins};
    for (DevBuf* b : wf) release(*b);
    DevBuf* all[] = {&ctx->geom,       &ctx->plane_geom, &ctx->tri_slot_id, &ctx->sphere_slot_id, &ctx->spheres_aos,
                     &ctx->planes_aos, &ctx->tris_aos,   &ctx->lights,      &ctx->ticket,         &ctx->fb_rgb,
                     &ctx->fb_rgba8,   &ctx->fb_pid,     &ctx->fb_fid,      &ctx->fb_stats,
                     &ctx->tri_slot_leaf, &ctx->sphere_slot_leaf, &ctx->accel_tmp, &ctx->heat[0], &ctx->heat[1], &ctx->order,
                     &ctx->gb_bvh,     &ctx->gb_tight,   &ctx->gb_tight_q,  &ctx->gb_flat,        &ctx->gb_unb,
                     &ctx->gb_list};
    for (DevBuf* b : all) release(*b);
    for (DevBuf& b : ctx->ev) release(b);
    delete ctx;
}

int rt_set_option(rt_ctx* ctx, int option, int value) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    switch (option) {
        case RT_OPTION_USE_LDS:
            ctx->allow_lds = value != 0 && ctx->ticket.p != nullptr;
            return RT_OK;
        case RT_OPTION_TRAVERSAL:
            if (value < RT_TRAVERSAL_AUTO || value > RT_TRAVERSAL_WAVEFRONT)
                return fail(ctx, RT_ERR_INVALID_ARGUMENT, "traversal must be one of RT_TRAVERSAL_*");
            ctx->traversal = value;
            return RT_OK;
        case RT_OPTION_WAVEFRONT_MIN_NODES:
            if (value < 0) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "node threshold must be >= 0");
            ctx->wf_min_nodes = value;
            return RT_OK;
        case RT_OPTION_TIGHT_TREE:
            ctx->build_tight = value != 0;
            return RT_OK;
        case RT_OPTION_COOP_SHADOWS:
            if (value < 0 || value > 2) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "cooperative shadow rays: 0 never, 1 always, 2 measured");
            ctx->coop_shadows = value;
            return RT_OK;
        case RT_OPTION_FRAME_FEEDBACK:
            ctx->use_heat = value != 0;
            ctx->heat_valid = false;
            return RT_OK;
        case RT_OPTION_BIN_PATHS:
            if (value < 0) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "first sorted iteration must be >= 0 (0: no binning)");
            ctx->wf_bin_from = value;
            return RT_OK;
        case RT_OPTION_PACKET_WALKS:
            ctx->wf_packet = value != 0;
            return RT_OK;
        case RT_OPTION_GPU_BUILD_MIN:
            if (value < -1) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "GPU build threshold must be >= -1");
            ctx->gpu_build_min = value;
            return RT_OK;
        case RT_OPTION_GPU_FLATTEN_MIN:
            if (value < -1) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "GPU flatten threshold must be >= -1");
            ctx->gpu_flatten_min = value;
            return RT_OK;
        case RT_OPTION_TIGHT_MIN_NODES:
            if (value < 0) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "node threshold must be >= 0");
            ctx->tight_min_nodes = value;
            return RT_OK;
        case RT_OPTION_SUPERTILE:
            if (value < 1 || value > 64) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "supertile must be 1..64");
            ctx->supertile = value;
            return RT_OK;
    }
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown option");
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

// Device half of rt_upload_scene for ONE context: the prepared host structures (rt_scene_prep.h) and the caller's
// arrays go to ctx's GPU.  A multi-GPU context prepares once and runs this for every device (the scene is
// replicated).
static int upload_prepared(rt_ctx* ctx, const rtp::Prepared& prep, const void* spheres, int n_spheres, const void* planes,
                           int n_planes, const void* triangles, int n_triangles, const float* lights, int lights_len) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // renders are asynchronous on caller streams (torch's are non-blocking: the copies below would not wait
    // for them), and a buffer that does not grow is overwritten in place: drain the device first
    HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->has_scene = false;
    const rtf::Flat& flat = prep.flat;
    const rtb::Bvh& bvh = prep.bvh;
    const rtb::Bvh& tight = prep.tight;
    const rtb::QBvh& tight_q = prep.tight_q;  // what the GPU gets of it: 32-byte nodes (rt_bvh.h quantize)
    const std::vector<int32_t>&tri_slot_leaf = prep.tri_slot_leaf, &sphere_slot_leaf = prep.sphere_slot_leaf;
    const bool contained = prep.contained;
    const float* extent = prep.extent;
    const float tight_e = prep.tight_e;
    // the two derived trees: host vectors, or arrays the device builder left on this GPU (rt_gpu_bvh.h)
    const rtp::Prepared::DeviceTrees& dev = prep.dev;
    const size_t n_bvh = dev.on ? dev.n_bvh : bvh.nodes.size(), n_tight = dev.on ? dev.n_tight : tight_q.nodes.size();
    const int bvh_levels = dev.on ? dev.bvh_levels : bvh.levels;
    const int tight_levels = n_tight == 0 ? 0 : (dev.on ? dev.tight_levels : tight.levels);
    int rc;
    const bool records_there = dev.on && dev.slot_leaf_on_device;  // (the tight tree's item kernel wrote the slot -> leaf maps)
    if (!ctx->staged_records) {  // (else: the device's containment check or item kernel read them from these buffers)
        if ((rc = upload(ctx, ctx->spheres_aos, spheres, (size_t)n_spheres * 48))) return rc;
        if ((rc = upload(ctx, ctx->tris_aos, triangles, (size_t)n_triangles * 96))) return rc;
        if ((rc = upload(ctx, ctx->tri_slot_id, flat.tri_slot_id.data(), flat.tri_slot_id.size() * 4))) return rc;
        if ((rc = upload(ctx, ctx->sphere_slot_id, flat.sphere_slot_id.data(), flat.sphere_slot_id.size() * 4))) return rc;
    }
    if ((rc = upload(ctx, ctx->planes_aos, planes, (size_t)n_planes * 64))) return rc;
    if ((rc = upload(ctx, ctx->lights, lights, (size_t)lights_len * 4))) return rc;
    const int n_ts = (int)flat.tri_slot_id.size(), n_ss = (int)flat.sphere_slot_id.size();
    // one contiguous geometry image: nodes | triangle slots | sphere slots | BVH nodes (rt_device.h)
    const size_t tri_off = flat.nodes.size() * 2, sph_off = tri_off + (size_t)n_ts * 3;
    const size_t bvh_off = sph_off + (size_t)n_ss;
    const size_t geom_f4 = bvh_off + n_bvh * 4;
    const size_t all_f4 = geom_f4 + n_tight * 2;
    if (all_f4 >= (size_t)INT32_MAX / 2) return fail(ctx, RT_ERR_TOO_LARGE, "geometry image too large");
    if ((rc = ensure(ctx, ctx->geom, all_f4 * 16))) return rc;
    if (!records_there) {
        if ((rc = upload(ctx, ctx->tri_slot_leaf, tri_slot_leaf.data(), tri_slot_leaf.size() * 4))) return rc;
        if ((rc = upload(ctx, ctx->sphere_slot_leaf, sphere_slot_leaf.data(), sphere_slot_leaf.size() * 4))) return rc;
    }
    if (n_tight)
        HIP_TRY(ctx, hipMemcpy((float4*)ctx->geom.p + geom_f4, dev.on ? dev.tight_q_nodes : (const void*)tight_q.nodes.data(),
                               n_tight * sizeof(rtb::QNode), dev.on ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    if (!flat.nodes.empty())
        HIP_TRY(ctx, hipMemcpy(ctx->geom.p, dev.on ? dev.flat_nodes : (const void*)flat.nodes.data(), flat.nodes.size() * 32,
                               dev.on ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    if (n_bvh)
        HIP_TRY(ctx, hipMemcpy((float4*)ctx->geom.p + bvh_off, dev.on ? dev.bvh_nodes : (const void*)bvh.nodes.data(),
                               n_bvh * sizeof(rtb::Node), dev.on ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    if ((rc = ensure(ctx, ctx->plane_geom, (size_t)n_planes * 32))) return rc;
    float4* const tri_geom = (float4*)ctx->geom.p + tri_off;
    float4* const sphere_geom = (float4*)ctx->geom.p + sph_off;

    // GPU re-layout: AoS upload -> traversal-order geometry slots
    if (n_ts)
        hipLaunchKernelGGL(k_gather_tris, dim3((n_ts + 255) / 256), dim3(256), 0, 0,
                           (const unsigned char*)ctx->tris_aos.p, (const int32_t*)ctx->tri_slot_id.p, tri_geom,
                           n_ts);
    if (n_ss)
        hipLaunchKernelGGL(k_gather_spheres, dim3((n_ss + 255) / 256), dim3(256), 0, 0,
                           (const unsigned char*)ctx->spheres_aos.p, (const int32_t*)ctx->sphere_slot_id.p,
                           sphere_geom, n_ss);
    if (n_planes)
        hipLaunchKernelGGL(k_gather_planes, dim3((n_planes + 255) / 256), dim3(256), 0, 0,
                           (const unsigned char*)ctx->planes_aos.p, (float4*)ctx->plane_geom.p, n_planes);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());

    DevScene& s = ctx->scene;
    s.geom = (const float4*)ctx->geom.p;
    s.tri_off = (int32_t)tri_off;
    s.sph_off = (int32_t)sph_off;
    s.geom_float4s = (int32_t)geom_f4;
    s.plane_geom = (const float4*)ctx->plane_geom.p;
    s.tri_slot_id = (const int32_t*)ctx->tri_slot_id.p;
    s.sphere_slot_id = (const int32_t*)ctx->sphere_slot_id.p;
    s.spheres_aos = (const unsigned char*)ctx->spheres_aos.p;
    s.planes_aos = (const unsigned char*)ctx->planes_aos.p;
    s.tris_aos = (const unsigned char*)ctx->tris_aos.p;
    s.lights = (const float*)ctx->lights.p;
    s.n_nodes = (int32_t)flat.nodes.size();
    s.n_planes = n_planes;
    s.n_spheres = n_spheres;
    s.n_tris = n_triangles;
    s.lights_len = lights_len;
    for (int k = 0; k < 3; k++) s.extent[k] = extent[k];
    s.cull_scale = contained ? 0.0078125f : INFINITY;

    s.bvh = n_bvh == 0 ? nullptr : (const float4*)ctx->geom.p + bvh_off;
    s.bvh_levels = std::max(bvh_levels, tight_levels);
    s.leaf_levels = bvh_levels;
    s.bvh_off = (int32_t)bvh_off;
    for (int k = 0; k < 3; k++) s.root_lo[k] = s.root_hi[k] = 0.0f;
    if (!flat.nodes.empty()) {
        const rtf::FlatNode& root = flat.nodes[0];
        s.root_lo[0] = root.minx, s.root_lo[1] = root.miny, s.root_lo[2] = root.minz;
        s.root_hi[0] = root.maxx, s.root_hi[1] = root.maxy, s.root_hi[2] = root.maxz;
    }
    {  // bounding sphere of the root box (rt_device.h misses_scene)
        double c[3], r2 = 0, c1 = 0;
        for (int k = 0; k < 3; k++) {
            c[k] = 0.5 * ((double)s.root_lo[k] + s.root_hi[k]);
            const double h = 0.5 * ((double)s.root_hi[k] - s.root_lo[k]);
            r2 += h * h, c1 += std::fabs(c[k]);
            s.bound_c[k] = (float)c[k];
        }
        const double R = std::sqrt(r2), Rp = R + 1e-5 * (c1 + R) + 1e-30;
        const bool usable = !flat.nodes.empty() && std::isfinite(R) && std::isfinite(c1) && c1 + R <= 1e17;
        s.bound_r2 = usable ? std::nextafterf((float)(Rp * Rp), INFINITY) : INFINITY;
    }
    s.tight = n_tight == 0 ? nullptr : (const uint4*)((const float4*)ctx->geom.p + geom_f4);
    for (int k = 0; k < 3; k++)
        s.tight_base[k] = dev.on ? dev.base[k] : tight_q.base[k], s.tight_step[k] = dev.on ? dev.step[k] : tight_q.step[k];
    s.tight_e = tight_e;
    s.tight_near_root = n_tight == 0 ? 0u : (dev.on ? dev.near_root : tight_q.near_root);
    for (int k = 0; k < 3; k++) s.near_lo[k] = dev.on ? prep.near_lo[k] : tight.near_lo[k], s.near_hi[k] = dev.on ? prep.near_hi[k] : tight.near_hi[k];
    s.tri_slot_leaf = (const int32_t*)ctx->tri_slot_leaf.p;
    s.sphere_slot_leaf = (const int32_t*)ctx->sphere_slot_leaf.p;

    rt_scene_info& inf = ctx->info;
    std::memset(&inf, 0, sizeof inf);
    inf.n_nodes = (int64_t)flat.nodes.size();
    inf.n_tri_slots = n_ts;
    inf.n_sphere_slots = n_ss;
    inf.max_depth = flat.max_depth;
    inf.device_bytes = (int64_t)(flat.nodes.size() * 32 + (size_t)n_ts * 52 + (size_t)n_ss * 20 +
                                 (size_t)n_planes * 96 + (size_t)n_spheres * 48 + (size_t)n_triangles * 96 +
                                 (size_t)lights_len * 4);
    // k_render_bvh_lds strides its per-lane stacks by scene.leaf_levels (it walks the re-treed leaves only)
    ctx->lds_bytes = (geom_f4 - tri_off) * 16 + (size_t)LDS_BLOCK * std::max(s.leaf_levels, 1) * 4;
    inf.lds_scene_bytes = (int64_t)ctx->lds_bytes;
    inf.bvh_nodes = (int64_t)n_bvh;
    inf.bvh_levels = bvh_levels;
    inf.flags = (flat.nested ? RT_SCENE_NESTED : 0) | (contained ? RT_SCENE_CULL : 0);
    inf.unbounded_triangles = prep.unbounded;
    inf.device_bytes += (int64_t)(n_bvh * sizeof(rtb::Node));
    inf.tight_nodes = (int64_t)n_tight;
    inf.tight_levels = tight_levels;
    inf.device_bytes += (int64_t)(n_tight * sizeof(rtb::QNode) +
                                  (records_there ? (size_t)n_ts + (size_t)n_ss : tri_slot_leaf.size() + sphere_slot_leaf.size()) * 4);
    ctx->lds_capable = n_bvh != 0 && ctx->lds_bytes <= (size_t)ctx->lds_limit;
    ctx->has_scene = true;
    ctx->scene_serial++;
    return RT_OK;
}

// The cull's preconditions (rt_scene_prep.h check_containment) on the GPU: one thread per leaf record runs the host's
// own rule (rtp::leaf_contained).  state: [0] violated, [1] unbounded triangles, [2] entries of `list`.
__global__ __launch_bounds__(256) void k_containment(const rtf::FlatNode* nodes, uint32_t n_nodes, const unsigned char* spheres,
                                                     const unsigned char* triangles, const int32_t* sphere_slot_id, const int32_t* tri_slot_id,
                                                     float slack, float max_extent, uint8_t* tri_unbounded, int32_t* list, uint32_t list_cap,
                                                     uint32_t* state) {
    const uint32_t ni = blockIdx.x * 256 + threadIdx.x;
    if (ni >= n_nodes) return;
    const bool fine = rtp::leaf_contained(nodes[ni], sphere_slot_id, tri_slot_id, spheres, triangles, slack, max_extent, [&](int slot) {
        tri_unbounded[slot] = 1;
        atomicAdd(state + 1, 1u);
        const uint32_t at = atomicAdd(state + 2, 1u);
        if (at < list_cap) list[at] = (int32_t)ni;
    });
    if (!fine) atomicExch(state, 1u);
}
// rtp::Options::containment for a single-GPU upload.  Leaves the records, slot id maps, flattened nodes and the
// unbounded flags on the device for build_trees_on_device.  -1: declined (small scene, HIP error): the host checks.
static int containment_on_device(rt_ctx* ctx, const rtf::Flat& flat, const float* extent, const void* spheres, int n_spheres,
                                 const void* triangles, int n_triangles, std::vector<uint8_t>* tri_unbounded, std::vector<int32_t>* leaves) {
    const size_t n_nodes = flat.nodes.size(), n_ts = flat.tri_slot_id.size(), n_ss = flat.sphere_slot_id.size();
    if ((int64_t)(n_ts + n_ss) < (int64_t)ctx->gpu_build_min || n_nodes >= 0x7FFFFFFFull || hipSetDevice(ctx->device) != hipSuccess) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;  // (renders still reading the buffers below: upload_prepared's rule)
    const size_t list_cap = 1 << 16;
    if (upload(ctx, ctx->spheres_aos, spheres, (size_t)n_spheres * 48) != RT_OK || upload(ctx, ctx->tris_aos, triangles, (size_t)n_triangles * 96) != RT_OK ||
        upload(ctx, ctx->tri_slot_id, flat.tri_slot_id.data(), n_ts * 4) != RT_OK ||
        upload(ctx, ctx->sphere_slot_id, flat.sphere_slot_id.data(), n_ss * 4) != RT_OK ||
        upload(ctx, ctx->gb_flat, flat.nodes.data(), n_nodes * sizeof(rtf::FlatNode)) != RT_OK || ensure(ctx, ctx->gb_unb, n_ts) != RT_OK ||
        ensure(ctx, ctx->gb_list, 16 + list_cap * 4) != RT_OK)
        return -1;
    ctx->staged_records = true;
    uint32_t* const d_state = (uint32_t*)ctx->gb_list.p;
    if (hipMemset(ctx->gb_unb.p, 0, std::max<size_t>(n_ts, 1)) != hipSuccess || hipMemset(d_state, 0, 16) != hipSuccess) return -1;
    const float max_extent = std::fmax(extent[0], std::fmax(extent[1], extent[2]));
    hipLaunchKernelGGL(k_containment, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, 0, (const rtf::FlatNode*)ctx->gb_flat.p,
                       (uint32_t)n_nodes, (const unsigned char*)ctx->spheres_aos.p, (const unsigned char*)ctx->tris_aos.p,
                       (const int32_t*)ctx->sphere_slot_id.p, (const int32_t*)ctx->tri_slot_id.p, max_extent * (1.0f / 4096.0f), max_extent,
                       (uint8_t*)ctx->gb_unb.p, (int32_t*)(d_state + 4), (uint32_t)list_cap, d_state);
    uint32_t state[4];
    if (hipMemcpy(state, d_state, 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (state[0]) return 0;
    if (state[2] > list_cap) return -1;  // (tens of thousands of unbounded triangles: the host's walk)
    if (state[1]) {  // rare: the host marks the leaves in its copy of the tree and needs the flags for its own item rules
        leaves->resize(state[2]);
        if (hipMemcpy(tri_unbounded->data(), ctx->gb_unb.p, n_ts, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(leaves->data(), d_state + 4, (size_t)state[2] * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        std::sort(leaves->begin(), leaves->end());
    }
    ctx->staged_flat = state[1] == 0;  // (marks change the host's nodes: build_trees_on_device sends them again)
    ctx->staged_unb = true;
    return 1;
}

// Both derived trees on ctx's GPU (rt_gpu_bvh.h): items made by kernels from the flattened nodes and the caller's
// records, the SAH builds, the 16-bit grid form -- with prepare's post-conditions (rtp::finish_trees_on_host).
// prep: as rtp::prepare leaves it with Options::defer_trees.  False: too small for min_items, or the device declined
// or failed -- the host makes the items and builds.  keep_fp32: leave the tight tree's fp32 nodes in ctx->gb_tight
// (the self-check compares them).
static bool build_trees_on_device(rt_ctx* ctx, rtp::Prepared* prep, const void* spheres, int n_spheres, const void* triangles,
                                  int n_triangles, int64_t min_items, bool keep_fp32 = false) {
    const rtf::Flat& flat = prep->flat;
    const size_t n_nodes = flat.nodes.size(), n_ts = flat.tri_slot_id.size(), n_ss = flat.sphere_slot_id.size();
    // how many items each group of each tree gets (rt_bvh.h make_leaf_items / make_tight_items)
    size_t leaf0 = 0, leaf1 = 0;
    if (flat.nested)
        for (const rtf::FlatNode& fn : flat.nodes) {
            const int kind = (fn.meta >> 29) & 3, count = (fn.meta >> 24) & 31;
            if (kind == rtf::KIND_BRANCH || count == 0) continue;
            (kind == rtf::KIND_TRI_UNBOUNDED ? leaf1 : leaf0)++;
        }
    const size_t leaf_n = leaf0 + leaf1;
    const bool want_tight = prep->want_tight && leaf_n > 0;  // (no leaf holds a primitive: no tight tree either)
    const size_t unb = (size_t)prep->unbounded, g0 = n_ts - unb, g1 = n_ss + unb;
    const bool near_group = prep->pads.near_on && n_ss > 0;  // (a near tree without a sphere would be the far tree again)
    const size_t tight_n = want_tight ? g0 + g1 + (near_group ? g1 : 0) : 0;
    if ((int64_t)(leaf_n + tight_n) < min_items || n_nodes >= 0x7FFFFFFFull) return false;
    rtp::Prepared::DeviceTrees dev;
    auto ok = [](hipError_t e) { return e == hipSuccess; };
    if (!ok(hipDeviceSynchronize())) return false;  // (renders still reading the buffers below: upload_prepared's rule)
    if (!ctx->staged_flat && upload(ctx, ctx->gb_flat, flat.nodes.data(), n_nodes * sizeof(rtf::FlatNode)) != RT_OK) return false;
    const unsigned grid = (unsigned)((n_nodes + 255) / 256);
    const size_t n_bvh = rtgb::node_count(leaf_n, leaf0, leaf_n);
    if (ensure(ctx, ctx->gb_bvh, n_bvh * sizeof(rtb::Node)) != RT_OK) return false;
    if (leaf_n) {
        rtgb::Build b;
        if (!b.alloc(leaf_n) || !ok(hipMemset(b.counters(), 0, 16))) return false;
        hipLaunchKernelGGL(rtgb::k_leaf_items, dim3(grid), dim3(256), 0, 0, (const rtf::FlatNode*)ctx->gb_flat.p, (uint32_t)n_nodes, b.items(),
                           (uint32_t)leaf0, b.counters());
        if (b.run(leaf0, leaf_n, (rtb::Node*)ctx->gb_bvh.p, &dev.bvh_levels, &dev.near_root) != 1) return false;
    }
    dev.bvh_nodes = ctx->gb_bvh.p, dev.n_bvh = n_bvh;
    bool tight = tight_n > 0;
    if (tight) {
        const size_t n_tight = rtgb::node_count(tight_n, g0, g0 + g1);
        if (!ctx->staged_records &&
            (upload(ctx, ctx->spheres_aos, spheres, (size_t)n_spheres * 48) != RT_OK || upload(ctx, ctx->tris_aos, triangles, (size_t)n_triangles * 96) != RT_OK ||
             upload(ctx, ctx->tri_slot_id, flat.tri_slot_id.data(), n_ts * 4) != RT_OK ||
             upload(ctx, ctx->sphere_slot_id, flat.sphere_slot_id.data(), n_ss * 4) != RT_OK))
            return false;
        ctx->staged_records = true;
        if ((!ctx->staged_unb && upload(ctx, ctx->gb_unb, prep->tri_unbounded.data(), n_ts) != RT_OK) || ensure(ctx, ctx->tri_slot_leaf, n_ts * 4) != RT_OK ||
            ensure(ctx, ctx->sphere_slot_leaf, n_ss * 4) != RT_OK || ensure(ctx, ctx->gb_tight, n_tight * sizeof(rtb::Node)) != RT_OK ||
            ensure(ctx, ctx->gb_tight_q, n_tight * sizeof(rtb::QNode)) != RT_OK)
            return false;
        {
            rtgb::Build b;
            if (!b.alloc(tight_n) || !ok(hipMemset(b.counters(), 0, 16))) return false;
            hipLaunchKernelGGL(rtgb::k_tight_items, dim3(grid), dim3(256), 0, 0, (const rtf::FlatNode*)ctx->gb_flat.p, (uint32_t)n_nodes,
                               (const unsigned char*)ctx->tris_aos.p, (const unsigned char*)ctx->spheres_aos.p, (const int32_t*)ctx->tri_slot_id.p,
                               (const int32_t*)ctx->sphere_slot_id.p, (const uint8_t*)ctx->gb_unb.p, prep->pads, near_group, b.items(),
                               (uint32_t)g0, (uint32_t)(g0 + g1), b.counters(), (int32_t*)ctx->tri_slot_leaf.p, (int32_t*)ctx->sphere_slot_leaf.p);
            if (b.run(g0, g0 + g1, (rtb::Node*)ctx->gb_tight.p, &dev.tight_levels, &dev.near_root) != 1) return false;
        }
        tight = dev.tight_levels <= rtb::MAX_LEVELS &&
                rtgb::quantize_on_device((const rtb::Node*)ctx->gb_tight.p, n_tight, (rtb::QNode*)ctx->gb_tight_q.p, dev.base, dev.step);
        if (!ok(hipDeviceSynchronize())) return false;
        if (!keep_fp32) release(ctx->gb_tight);  // (the fp32 form is only the quantiser's input)
        if (tight) dev.tight_q_nodes = ctx->gb_tight_q.p, dev.n_tight = n_tight, dev.slot_leaf_on_device = true;
    }
    if (!tight) dev.near_root = 0, dev.tight_levels = 0;
    dev.flat_nodes = ctx->gb_flat.p;
    dev.on = true;
    prep->dev = dev;
    return true;
}

// A multi-GPU context replicates the scene from host arrays: the trees built on the first device come back to the
// host's vectors (what rtp::finish_trees_on_host would have left there).
static bool device_trees_to_host(rt_ctx* ctx, rtp::Prepared* prep) {
    const rtp::Prepared::DeviceTrees dev = prep->dev;
    auto get = [](void* dst, const void* src, size_t bytes) { return bytes == 0 || hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
    prep->bvh = rtb::Bvh(), prep->tight = rtb::Bvh(), prep->tight_q = rtb::QBvh();
    prep->bvh.nodes.resize(dev.n_bvh), prep->bvh.levels = dev.bvh_levels;
    prep->tight_q.nodes.resize(dev.n_tight), prep->tight_q.near_root = dev.near_root;
    prep->tight.levels = dev.tight_levels, prep->tight.near_root = dev.near_root;
    for (int k = 0; k < 3; k++) {
        prep->tight_q.base[k] = dev.base[k], prep->tight_q.step[k] = dev.step[k];
        prep->tight.near_lo[k] = prep->near_lo[k], prep->tight.near_hi[k] = prep->near_hi[k];
    }
    prep->tri_slot_leaf.clear(), prep->sphere_slot_leaf.clear();
    if (dev.slot_leaf_on_device) prep->tri_slot_leaf.resize(prep->flat.tri_slot_id.size()), prep->sphere_slot_leaf.resize(prep->flat.sphere_slot_id.size());
    const bool ok = get(prep->bvh.nodes.data(), dev.bvh_nodes, dev.n_bvh * sizeof(rtb::Node)) &&
                    get(prep->tight_q.nodes.data(), dev.tight_q_nodes, dev.n_tight * sizeof(rtb::QNode)) &&
                    get(prep->tri_slot_leaf.data(), ctx->tri_slot_leaf.p, prep->tri_slot_leaf.size() * 4) &&
                    get(prep->sphere_slot_leaf.data(), ctx->sphere_slot_leaf.p, prep->sphere_slot_leaf.size() * 4);
    prep->dev = rtp::Prepared::DeviceTrees();
    return ok;
}

static int upload_scene(rt_ctx* ctx, const void* spheres, int n_spheres, const void* planes, int n_planes,
                        const void* triangles, int n_triangles, const float* accel, int64_t accel_len,
                        const float* lights, int lights_len);
int rt_upload_scene(rt_ctx* ctx, const void* spheres, int n_spheres, const void* planes, int n_planes,
                    const void* triangles, int n_triangles, const float* accel, int64_t accel_len,
                    const float* lights, int lights_len) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    const int rc = no_throw(ctx, nullptr, [&] {
        return upload_scene(ctx, spheres, n_spheres, planes, n_planes, triangles, n_triangles, accel, accel_len, lights, lights_len);
    });
    // the copy of the accel array the GPU flatten worked on is not needed once the scene is up (or refused), nor are
    // the device builder's arrays (upload_prepared has copied them into the geometry image)
    if ((ctx->accel_tmp.p || ctx->gb_bvh.p || ctx->gb_tight.p || ctx->gb_tight_q.p || ctx->gb_flat.p || ctx->gb_unb.p) &&
        hipSetDevice(ctx->device) == hipSuccess)
        release(ctx->accel_tmp), release(ctx->gb_bvh), release(ctx->gb_tight), release(ctx->gb_tight_q), release(ctx->gb_flat),
            release(ctx->gb_unb), release(ctx->gb_list);
    return rc;
}
static int upload_scene(rt_ctx* ctx, const void* spheres, int n_spheres, const void* planes, int n_planes,
                        const void* triangles, int n_triangles, const float* accel, int64_t accel_len,
                        const float* lights, int lights_len) {
    if (n_spheres < 0 || n_planes < 0 || n_triangles < 0 || accel_len < 0 || lights_len < 0 || lights_len % 6 != 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "negative count or lights_len not a multiple of 6");
    if ((n_spheres && !spheres) || (n_planes && !planes) || (n_triangles && !triangles) || (accel_len && !accel) ||
        (lights_len && !lights))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null array with non-zero count");
    ctx->has_scene = false;

    // Validation + flattening of a large array runs on the GPU (rt_gpu_flatten.h: level by level instead of node by
    // node; bit-identical to the host's, tests/test_gpu_flatten.py).  If it declines -- an array it would have to
    // refuse, or a HIP error -- the host does it, and reports what is wrong with the array in its own words.
    rtp::Prepared prep;
    bool flat_given = false;
    const auto t0 = std::chrono::steady_clock::now();
    if (ctx->gpu_flatten_min >= 0 && accel_len >= ctx->gpu_flatten_min && accel_len >= 8 &&
        hipSetDevice(ctx->device) == hipSuccess && upload(ctx, ctx->accel_tmp, accel, (size_t)accel_len * 4) == RT_OK)
        flat_given = rtgf::flatten_on_device((const float*)ctx->accel_tmp.p, accel_len, n_spheres, n_triangles, &prep.flat) == 1;
    if (std::getenv("RT_PREP_TIMING"))
        std::fprintf(stderr, "[rt prepare] %-28s %8.1f ms\n", flat_given ? "validate + flatten (GPU)" : "(GPU flatten not used)",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    // everything else the upload decides is decided on the host (rt_scene_prep.h): cull preconditions, both trees
    std::string err;
    rtp::Options opt;
    opt.build_tight = ctx->build_tight, opt.tight_min_nodes = ctx->tight_min_nodes;
    // the two derived trees of a large scene are built on the GPU (rt_gpu_bvh.h; bit-identical to the host builder's,
    // tests/test_gpu_build.py); a multi-GPU context builds on its first device and replicates from host copies
    opt.defer_trees = ctx->gpu_build_min >= 0;
    ctx->staged_records = ctx->staged_flat = ctx->staged_unb = false;
    if (opt.defer_trees)  // ... and so are the cull's preconditions (same rule, rtp::leaf_contained)
        opt.containment = [&](const rtf::Flat& flat, const float* extent, std::vector<uint8_t>* tri_unbounded, std::vector<int32_t>* leaves) {
            return containment_on_device(ctx, flat, extent, spheres, n_spheres, triangles, n_triangles, tri_unbounded, leaves);
        };
    int rc = rtp::prepare((const unsigned char*)spheres, n_spheres, (const unsigned char*)triangles, n_triangles, accel,
                          accel_len, opt, &prep, &err, flat_given);
    if (rc != rtf::FLATTEN_OK) return fail(ctx, rc, err);
    if (opt.defer_trees) {
        const auto t1 = std::chrono::steady_clock::now();
        bool on_device = hipSetDevice(ctx->device) == hipSuccess &&
                         build_trees_on_device(ctx, &prep, spheres, n_spheres, triangles, n_triangles, ctx->gpu_build_min);
        if (on_device && ctx->multi && !device_trees_to_host(ctx, &prep)) on_device = false;
        if (!on_device) {
            rtp::make_items_on_host(&prep, (const unsigned char*)spheres, (const unsigned char*)triangles);
            rtp::finish_trees_on_host(&prep);
        }
        if (std::getenv("RT_PREP_TIMING"))
            std::fprintf(stderr, "[rt prepare] %-28s %8.1f ms\n", on_device ? "both trees (SAH) on the GPU" : "both trees (SAH), host",
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    }
    if (ctx->multi) return multi_upload(ctx, prep, spheres, n_spheres, planes, n_planes, triangles, n_triangles, lights, lights_len);
    return upload_prepared(ctx, prep, spheres, n_spheres, planes, n_planes, triangles, n_triangles, lights, lights_len);
}

int rt_check_accel(const float* accel, int64_t accel_len, int n_spheres, int n_triangles, rt_scene_info* info,
                   char* err, size_t err_cap) {
    rtf::Flat flat;
    std::string msg;
    int rc = (accel_len < 0 || n_spheres < 0 || n_triangles < 0 || (accel_len && !accel)) ? RT_ERR_INVALID_ARGUMENT
                                                                                         : RT_OK;
    if (rc != RT_OK) msg = "negative size or null array";
    rtb::Bvh bvh;
    if (rc == RT_OK)
        rc = no_throw(nullptr, &msg, [&] {
            const int r = rtf::flatten(accel, accel_len, n_spheres, n_triangles, &flat, &msg);
            if (r == RT_OK && info && flat.nested) rtb::build(flat, &bvh);
            return r;
        });
    if (err && err_cap) {
        std::strncpy(err, msg.c_str(), err_cap - 1);
        err[err_cap - 1] = 0;
    }
    if (rc == RT_OK && info) {
        std::memset(info, 0, sizeof *info);
        info->n_nodes = (int64_t)flat.nodes.size();
        info->n_tri_slots = (int64_t)flat.tri_slot_id.size();
        info->n_sphere_slots = (int64_t)flat.sphere_slot_id.size();
        info->max_depth = flat.max_depth;
        info->lds_scene_bytes =
            (int64_t)(flat.nodes.size() * 32 + flat.tri_slot_id.size() * 48 + flat.sphere_slot_id.size() * 16);
        info->flags = flat.nested ? RT_SCENE_NESTED : 0;  // RT_SCENE_CULL needs the primitives: upload only
        if (flat.nested) {
            info->bvh_nodes = (int64_t)bvh.nodes.size();
            info->bvh_levels = bvh.levels;
        }
    }
    return rc;
}

int rt_check_scene(const void* spheres, int n_spheres, const void* triangles, int n_triangles, const float* accel,
                   int64_t accel_len, int tight_min_nodes, rt_scene_info* info, char* err, size_t err_cap) {
    std::string msg;
    int rc = RT_OK;
    if (accel_len < 0 || n_spheres < 0 || n_triangles < 0 || tight_min_nodes < 0 || (accel_len && !accel) ||
        (n_spheres && !spheres) || (n_triangles && !triangles)) {
        rc = RT_ERR_INVALID_ARGUMENT;
        msg = "negative size or null array";
    }
    rtp::Prepared prep;
    if (rc == RT_OK)
        rc = no_throw(nullptr, &msg, [&] {
            rtp::Options opt;
            opt.tight_min_nodes = tight_min_nodes;
            int r = rtp::prepare((const unsigned char*)spheres, n_spheres, (const unsigned char*)triangles, n_triangles, accel,
                                 accel_len, opt, &prep, &msg);
            if (r != RT_OK) return r;
            msg = rtp::verify_tree(prep.bvh, nullptr, nullptr, nullptr);
            if (msg.empty())
                msg = rtp::verify_tree(prep.tight, &prep, (const unsigned char*)spheres, (const unsigned char*)triangles);
            if (msg.empty()) msg = rtp::verify_quantized(prep.tight, prep.tight_q);
            if (msg.empty()) {
                // the trees must not depend on the order their items arrive in (the device builder partitions with
                // atomics, rt_gpu_bvh.h): the same items reversed and rotated give the same node arrays
                rtp::make_items_on_host(&prep, (const unsigned char*)spheres, (const unsigned char*)triangles);
                rtb::Items* src[2] = {&prep.leaf_items, &prep.tight_items};
                const rtb::Bvh* want[2] = {&prep.bvh, &prep.tight};
                for (int t = 0; t < 2 && msg.empty(); t++) {
                    if (want[t]->nodes.empty()) continue;
                    std::vector<rtb::detail::Item>& it = src[t]->items;
                    const size_t bounds[4] = {0, src[t]->split, src[t]->split2, it.size()};
                    for (int h = 0; h < 3; h++) {
                        std::reverse(it.begin() + bounds[h], it.begin() + bounds[h + 1]);
                        if (bounds[h + 1] - bounds[h] > 3) std::rotate(it.begin() + bounds[h], it.begin() + bounds[h] + (bounds[h + 1] - bounds[h]) / 3, it.begin() + bounds[h + 1]);
                    }
                    rtb::Bvh again;
                    rtb::detail::build_items(it, src[t]->split, &again, src[t]->split2);
                    if (again.nodes.size() != want[t]->nodes.size() || again.levels != want[t]->levels ||
                        std::memcmp(again.nodes.data(), want[t]->nodes.data(), again.nodes.size() * sizeof(rtb::Node)) != 0)
                        msg = t ? "the tight tree depends on the order of its items" : "the re-treed leaves depend on the order of their items";
                }
            }
            return msg.empty() ? (int)RT_OK : (int)RT_ERR_DEVICE;  // an internal inconsistency, never expected
        });
    if (err && err_cap) {
        std::strncpy(err, msg.c_str(), err_cap - 1);
        err[err_cap - 1] = 0;
    }
    if (rc == RT_OK && info) {
        std::memset(info, 0, sizeof *info);
        info->n_nodes = (int64_t)prep.flat.nodes.size();
        info->n_tri_slots = (int64_t)prep.flat.tri_slot_id.size();
        info->n_sphere_slots = (int64_t)prep.flat.sphere_slot_id.size();
        info->max_depth = prep.flat.max_depth;
        info->flags = (prep.flat.nested ? RT_SCENE_NESTED : 0) | (prep.contained ? RT_SCENE_CULL : 0);
        info->bvh_nodes = (int64_t)prep.bvh.nodes.size();
        info->bvh_levels = prep.bvh.levels;
        info->tight_nodes = (int64_t)prep.tight.nodes.size();
        info->tight_levels = prep.tight.nodes.empty() ? 0 : prep.tight.levels;
        info->unbounded_triangles = prep.unbounded;
    }
    return rc;
}

int rt_check_gpu_build(rt_ctx* ctx, const void* spheres, int n_spheres, const void* triangles, int n_triangles, const float* accel,
                       int64_t accel_len, int tight_min_nodes, char* err, size_t err_cap) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (accel_len < 0 || n_spheres < 0 || n_triangles < 0 || tight_min_nodes < 0 || (accel_len && !accel) || (n_spheres && !spheres) ||
        (n_triangles && !triangles))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "negative size or null array");
    std::string msg;
    int result = 1;
    const int rc = no_throw(ctx, &msg, [&]() -> int {
        // the host's trees, then the upload's own device route on a second preparation of the same arrays
        rtp::Prepared host, prep;
        rtp::Options opt;
        opt.tight_min_nodes = tight_min_nodes;
        int r = rtp::prepare((const unsigned char*)spheres, n_spheres, (const unsigned char*)triangles, n_triangles, accel, accel_len,
                             opt, &host, &msg);
        if (r != RT_OK) return r;
        opt.defer_trees = true;
        ctx->has_scene = false;  // (the route writes the context's record and slot buffers)
        ctx->staged_records = ctx->staged_flat = ctx->staged_unb = false;
        const int min_before = ctx->gpu_build_min;
        ctx->gpu_build_min = 0;
        int device_checked = -1;
        opt.containment = [&](const rtf::Flat& flat, const float* extent, std::vector<uint8_t>* tri_unbounded, std::vector<int32_t>* leaves) {
            return device_checked = containment_on_device(ctx, flat, extent, spheres, n_spheres, triangles, n_triangles, tri_unbounded, leaves);
        };
        r = rtp::prepare((const unsigned char*)spheres, n_spheres, (const unsigned char*)triangles, n_triangles, accel, accel_len, opt,
                         &prep, &msg);
        ctx->gpu_build_min = min_before;
        if (r != RT_OK) return r;
        if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return RT_ERR_DEVICE;
        if (prep.flat.nested && device_checked < 0) {
            msg = "the device declined the containment check";
            return RT_ERR_DEVICE;
        }
        if (prep.contained != host.contained || prep.unbounded != host.unbounded || prep.tri_unbounded != host.tri_unbounded ||
            prep.flat.nodes.size() != host.flat.nodes.size() ||
            std::memcmp(prep.flat.nodes.data(), host.flat.nodes.data(), host.flat.nodes.size() * sizeof(rtf::FlatNode)) != 0) {
            msg = "containment check: the device's verdict, unbounded triangles or never-cull marks differ from the host's";
            return RT_ERR_DEVICE;
        }
        if (!build_trees_on_device(ctx, &prep, spheres, n_spheres, triangles, n_triangles, 0, true)) {
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) {
                msg = std::string("HIP error in the device build: ") + hipGetErrorString(e);
                return RT_ERR_DEVICE;
            }
            result = 0;  // declined
            return RT_OK;
        }
        const rtp::Prepared::DeviceTrees& dev = prep.dev;
        auto differs = [&](const char* what, const void* d_ptr, const void* h_ptr, size_t bytes, size_t record) {
            std::vector<unsigned char> got(bytes);
            if (bytes && hipMemcpy(got.data(), d_ptr, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
                msg = std::string(what) + ": cannot read the device array";
                return true;
            }
            if (bytes == 0 || std::memcmp(got.data(), h_ptr, bytes) == 0) return false;
            size_t at = 0;
            while (std::memcmp(got.data() + at * record, (const unsigned char*)h_ptr + at * record, record) == 0) at++;
            msg = std::string(what) + " differ at record " + std::to_string(at) + " of " + std::to_string(bytes / record);
            return true;
        };
        if (dev.n_bvh != host.bvh.nodes.size() || dev.n_tight != host.tight_q.nodes.size()) {
            msg = "node counts differ: leaves " + std::to_string(dev.n_bvh) + " / " + std::to_string(host.bvh.nodes.size()) + ", tight " +
                  std::to_string(dev.n_tight) + " / " + std::to_string(host.tight_q.nodes.size());
            return RT_ERR_DEVICE;
        }
        if (dev.n_bvh && dev.bvh_levels != host.bvh.levels) {
            msg = "depth of the re-treed leaves differs";
            return RT_ERR_DEVICE;
        }
        if (differs("nodes of the re-treed leaves", dev.bvh_nodes, host.bvh.nodes.data(), dev.n_bvh * sizeof(rtb::Node), sizeof(rtb::Node)))
            return RT_ERR_DEVICE;
        if (dev.n_tight) {
            if (dev.tight_levels != host.tight.levels || dev.near_root != host.tight.near_root) {
                msg = "depth / second holder of the tight tree differ";
                return RT_ERR_DEVICE;
            }
            if (std::memcmp(dev.base, host.tight_q.base, 12) != 0 || std::memcmp(dev.step, host.tight_q.step, 12) != 0) {
                msg = "grids of the tight tree differ";
                return RT_ERR_DEVICE;
            }
            if (differs("fp32 nodes of the tight tree", ctx->gb_tight.p, host.tight.nodes.data(), dev.n_tight * sizeof(rtb::Node), sizeof(rtb::Node)) ||
                differs("16-bit nodes of the tight tree", dev.tight_q_nodes, host.tight_q.nodes.data(), dev.n_tight * sizeof(rtb::QNode),
                        sizeof(rtb::QNode)) ||
                differs("triangle slot -> leaf maps", ctx->tri_slot_leaf.p, host.tri_slot_leaf.data(), host.tri_slot_leaf.size() * 4, 4) ||
                differs("sphere slot -> leaf maps", ctx->sphere_slot_leaf.p, host.sphere_slot_leaf.data(), host.sphere_slot_leaf.size() * 4, 4))
                return RT_ERR_DEVICE;
        }
        // second route: the host's items, in the host's order, through the device builder
        rtp::make_items_on_host(&prep, (const unsigned char*)spheres, (const unsigned char*)triangles);
        const rtb::Items* src[2] = {&prep.leaf_items, &prep.tight_items};
        const rtb::Bvh* want[2] = {&host.bvh, &host.tight};
        for (int t = 0; t < 2; t++) {
            const size_t n_nodes = rtgb::node_count(*src[t]);
            if (n_nodes == 0 || want[t]->nodes.empty()) continue;
            DevBuf& out = t ? ctx->gb_tight : ctx->gb_bvh;
            if (ensure(ctx, out, n_nodes * sizeof(rtb::Node)) != RT_OK) return RT_ERR_DEVICE;
            int levels = 0;
            uint32_t near_root = 0;
            if (rtgb::build_on_device(*src[t], (rtb::Node*)out.p, &levels, &near_root) != 1) {
                msg = "the device builder refused host items it took from its own kernels";
                return RT_ERR_DEVICE;
            }
            if (levels != want[t]->levels || differs(t ? "tight tree from host items" : "re-treed leaves from host items", out.p,
                                                      want[t]->nodes.data(), n_nodes * sizeof(rtb::Node), sizeof(rtb::Node))) {
                if (msg.empty()) msg = "depths differ (host items)";
                return RT_ERR_DEVICE;
            }
        }
        return RT_OK;
    });
    ctx->staged_records = ctx->staged_flat = ctx->staged_unb = false;
    if (hipSetDevice(ctx->device) == hipSuccess)
        release(ctx->gb_bvh), release(ctx->gb_tight), release(ctx->gb_tight_q), release(ctx->gb_flat), release(ctx->gb_unb), release(ctx->gb_list);
    if (err && err_cap) {
        std::strncpy(err, msg.c_str(), err_cap - 1);
        err[err_cap - 1] = 0;
    }
    if (rc != RT_OK) return fail(ctx, rc, msg);
    return result;
}

int rt_check_gpu_flatten(rt_ctx* ctx, const float* accel, int64_t accel_len, int n_spheres, int n_triangles, char* err,
                         size_t err_cap) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (accel_len < 0 || n_spheres < 0 || n_triangles < 0 || (accel_len && !accel))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "negative size or null array");
    rtf::Flat host, dev;
    std::string msg, host_msg;
    const int host_rc = rtf::flatten(accel, accel_len, n_spheres, n_triangles, &host, &host_msg);
    int took = 0;
    bool too_deep = false;
    if (accel_len >= 8) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        const int rc = upload(ctx, ctx->accel_tmp, accel, (size_t)accel_len * 4);
        if (rc != RT_OK) return rc;
        took = rtgf::flatten_on_device((const float*)ctx->accel_tmp.p, accel_len, n_spheres, n_triangles, &dev, &too_deep);
    }
    auto differ = [&](const char* what, size_t at) { msg = std::string(what) + " differ at " + std::to_string(at); };
    if (host_rc != rtf::FLATTEN_OK) {
        if (took) msg = "the device flattened an array the host refuses: " + host_msg;
    } else if (accel_len >= 8 && !took && !too_deep) {
        msg = "the device declined an array the host accepts";
    } else if (took) {
        if (dev.nodes.size() != host.nodes.size() || dev.tri_slot_id.size() != host.tri_slot_id.size() ||
            dev.sphere_slot_id.size() != host.sphere_slot_id.size() || dev.parent.size() != host.parent.size())
            msg = "array sizes differ";
        else if (dev.max_depth != host.max_depth || dev.logical_nodes != host.logical_nodes || dev.nested != host.nested)
            msg = "max_depth / logical_nodes / nested differ";
        for (size_t i = 0; msg.empty() && i < host.nodes.size(); i++)
            if (std::memcmp(&host.nodes[i], &dev.nodes[i], sizeof(rtf::FlatNode)) != 0) differ("node records", i);
        for (size_t i = 0; msg.empty() && i < host.parent.size(); i++)
            if (host.parent[i] != dev.parent[i]) differ("parent links", i);
        for (size_t i = 0; msg.empty() && i < host.tri_slot_id.size(); i++)
            if (host.tri_slot_id[i] != dev.tri_slot_id[i]) differ("triangle slot ids", i);
        for (size_t i = 0; msg.empty() && i < host.sphere_slot_id.size(); i++)
            if (host.sphere_slot_id[i] != dev.sphere_slot_id[i]) differ("sphere slot ids", i);
    }
    if (err && err_cap) {
        std::strncpy(err, msg.c_str(), err_cap - 1);
        err[err_cap - 1] = 0;
    }
    if (!msg.empty()) return fail(ctx, RT_ERR_DEVICE, msg);
    if (host_rc != rtf::FLATTEN_OK) {  // refused by both: what an upload would report
        if (err && err_cap) {
            std::strncpy(err, host_msg.c_str(), err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return fail(ctx, host_rc, host_msg);
    }
    return took ? 1 : 0;
}

int rt_get_scene_info(const rt_ctx* ctx, rt_scene_info* out) {
    if (!ctx || !out) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return RT_ERR_NO_SCENE;
    *out = ctx->info;
    out->traversal = pick_traversal(ctx, ctx->scene, 10);
    // a mid-sized scene under AUTO: the measured verdict once there is one (tune_begin)
    const rt_ctx::Tune& t = ctx->tune;
    if (ctx->traversal == RT_TRAVERSAL_AUTO && out->traversal == RT_TRAVERSAL_BVH && t.scene_serial == ctx->scene_serial &&
        t.cand[1] && t.decided())
        out->traversal = t.best() == 1 ? RT_TRAVERSAL_WAVEFRONT : RT_TRAVERSAL_BVH;
    return RT_OK;
}

// Screen columns (in super-tiles) that the scene's root box projects to: the tiles the kernels deal out
// first (PriorityRect, rt_kernels.h).  The camera array holds the eye and three corners of the image
// plane; a point P is seen at (u, v) with eye + s (P - eye) on the plane, i.e.
// u e1 + v e2 - w (P - eye) = eye - BL with w = 1/s.  Any corner behind the eye, a degenerate camera or a
// box wider than 60 % of the frame give "no priority".  Scheduling only: a wrong answer costs time, never
// pixels.
static PriorityRect priority_rect(const DevScene& sc, const float* cam, int width, int local_rows, int super_px) {
    PriorityRect none{1, 1, 0, 0};
    if (sc.n_nodes <= 0) return none;
    const double eye[3] = {cam[0], cam[1], cam[2]};
    double e1[3], e2[3], c[3];
    for (int k = 0; k < 3; k++) e1[k] = (double)cam[6 + k] - cam[3 + k], e2[k] = (double)cam[9 + k] - cam[3 + k], c[k] = eye[k] - cam[3 + k];
    double umin = 1e300, umax = -1e300;
    for (int corner = 0; corner < 8; corner++) {
        double q[3];
        for (int k = 0; k < 3; k++) q[k] = ((corner >> k) & 1 ? sc.root_hi[k] : sc.root_lo[k]) - eye[k];
        // Cramer: columns e1, e2, -q; right-hand side c
        auto det = [](const double* a, const double* b, const double* d) {
            return a[0] * (b[1] * d[2] - b[2] * d[1]) - b[0] * (a[1] * d[2] - a[2] * d[1]) + d[0] * (a[1] * b[2] - a[2] * b[1]);
        };
        const double nq[3] = {-q[0], -q[1], -q[2]};
        const double D = det(e1, e2, nq);
        if (!(std::fabs(D) > 1e-300)) return none;
        const double u = det(c, e2, nq) / D, v = det(e1, c, nq) / D, w = det(e1, e2, c) / D;
        if (!(w > 1e-9) || !(u == u) || !(v == v)) return none;  // behind the eye or at it
        umin = std::min(umin, u), umax = std::max(umax, u);
    }
    const double x0 = std::max(0.0, umin * width), x1 = std::min((double)width - 1, umax * width);
    if (!(x0 <= x1)) return none;  // off screen
    if ((x1 - x0 + 1) > 0.6 * width) return none;  // most of the frame: nothing to prioritise
    // The whole column range, not just the box's rows: the other long waves of a frame (the mesh's mirror
    // image in a floor, the floor points whose shadow rays cross it) sit above and below it on the screen.
    // Measured on the teapot frame: box rectangle 0.264 ms, full-height strip 0.244 ms, full-width strip
    // 0.275 ms (= no priority).
    PriorityRect pr;
    pr.x0 = (int)x0 / super_px, pr.x1 = (int)x1 / super_px;
    pr.y0 = 0, pr.y1 = std::max(local_rows - 1, 0) / super_px;
    return pr;
}

// Kernel choice by measurement.  For a (scene, frame size, bounces) the kernels in the running are
//   [0] the per-lane megakernel                       always
//   [1] the staged pipeline                           AUTO, mid-sized scenes (RT_OPTION_WAVEFRONT_MIN_NODES .. 100 k nodes)
//   [2] the megakernel with cooperative shadow rays   RT_OPTION_COOP_SHADOWS = 2, bounces >= 1, 1..32 lights
// Each in turn renders three frames -- a warm-up (first-use allocation of the path state, code load), then two
// between events -- and from then on the fastest is used (results are identical either way).  Nothing waits: a
// measurement is read when its events have completed, until then renders use the plain megakernel.  Returns the
// kernel to force (RT_TRAVERSAL_BVH / WAVEFRONT, | FORCE_COOP) or 0 for "no opinion"; *timed = slot whose stop event
// tune_end must record, or -1.
constexpr int FORCE_COOP = 0x100;
static int tune_code(int k) { return k == 1 ? RT_TRAVERSAL_WAVEFRONT : (k == 2 ? (RT_TRAVERSAL_BVH | FORCE_COOP) : RT_TRAVERSAL_BVH); }
static bool tune_candidates(const rt_ctx* ctx, int max_bounces, bool cand[rt_ctx::Tune::N]) {
    const DevScene& sc = ctx->scene;
    cand[0] = true, cand[1] = cand[2] = false;
    if (!ctx->has_scene || sc.bvh == nullptr || sc.bvh_levels > rtb::MAX_LEVELS) return false;
    if (ctx->traversal != RT_TRAVERSAL_AUTO && ctx->traversal != RT_TRAVERSAL_BVH) return false;
    const bool large = ctx->traversal == RT_TRAVERSAL_AUTO && max_bounces <= ctx->wf_max_bounces &&
                       sc.n_nodes >= std::max(ctx->wf_sure_nodes, ctx->wf_min_nodes);
    if (large) return false;  // the staged pipeline, without asking
    cand[1] = ctx->traversal == RT_TRAVERSAL_AUTO && max_bounces <= ctx->wf_max_bounces && sc.n_nodes >= ctx->wf_min_nodes;
    cand[2] = ctx->coop_shadows == 2 && max_bounces >= 1 && sc.lights_len > 0 && sc.lights_len / 6 <= COOP_MAX_LIGHTS;
    return cand[1] || cand[2];
}
// the verdict for this scene and bounce count if there is one (batches and the entry point that must not create
// events reuse it), else 0
static int tune_verdict(const rt_ctx* ctx, int max_bounces) {
    const rt_ctx::Tune& t = ctx->tune;
    bool cand[rt_ctx::Tune::N];
    if (!tune_candidates(ctx, max_bounces, cand)) return 0;
    if (t.scene_serial != ctx->scene_serial || t.bounces != max_bounces || !t.decided()) return 0;
    for (int k = 0; k < rt_ctx::Tune::N; k++)
        if (cand[k] != t.cand[k]) return 0;
    return tune_code(t.best());
}
static int tune_begin(rt_ctx* ctx, int w, int local_rows, int max_bounces, hipStream_t st, int* timed) {
    *timed = -1;
    constexpr int N = rt_ctx::Tune::N;
    bool cand[N];
    if (!tune_candidates(ctx, max_bounces, cand)) return 0;
    rt_ctx::Tune& t = ctx->tune;
    bool same = t.scene_serial == ctx->scene_serial && t.w == w && t.rows == local_rows && t.bounces == max_bounces;
    for (int k = 0; k < N; k++) same = same && cand[k] == t.cand[k];
    if (!same) {
        t.scene_serial = ctx->scene_serial, t.w = w, t.rows = local_rows, t.bounces = max_bounces;
        for (int k = 0; k < N; k++) t.cand[k] = cand[k], t.ms[k] = -1.0f, t.runs[k] = 0, t.samples[k] = 0, t.pending[k] = false;
    }
    for (int k = 0; k < N; k++)
        if (t.pending[k] && hipEventQuery(t.ev[k][1]) == hipSuccess) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, t.ev[k][0], t.ev[k][1]) == hipSuccess) {
                t.ms[k] = t.samples[k] == 0 ? ms : std::min(t.ms[k], ms);
                t.samples[k]++;
            }
            t.pending[k] = false;
        }
    (void)hipGetLastError();  // hipEventQuery reports "not ready" as an error
    for (int k = 0; k < N; k++)
        if (t.cand[k] && t.samples[k] < 2 && !t.pending[k]) {
            if (t.runs[k]++ == 0) return tune_code(k);  // warm-up, not timed
            for (int e = 0; e < 2; e++)
                if (!t.ev[k][e] && hipEventCreate(&t.ev[k][e]) != hipSuccess) return 0;
            if (hipEventRecord(t.ev[k][0], st) != hipSuccess) return 0;
            *timed = k;
            return tune_code(k);
        }
    if (t.decided()) return tune_code(t.best());
    return RT_TRAVERSAL_BVH;
}
static void tune_end(rt_ctx* ctx, int timed, hipStream_t st) {
    if (timed < 0) return;
    if (hipEventRecord(ctx->tune.ev[timed][1], st) == hipSuccess) ctx->tune.pending[timed] = true;
}

// Core of rt_render_device / rt_render_frames_device: n_frames (<= MAX_BATCH) cameras of one size in one
// launch of a megakernel; the staged pipeline, the LDS kernel and the reference-order kernel take one
// frame per launch (callers loop).
static int render_frames(rt_ctx* ctx, const float* camera, int n_frames, uint64_t frame_stride, int max_bounces,
                         int flags, int band_rows, int rank, int world, float* d_rgb, uint8_t* d_rgba8,
                         int32_t* d_primary_id, int32_t* d_final_id, int32_t* d_bounces, rt_stats* d_stats,
                         void* stream, int force_trav = 0, bool noalloc = false) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_upload_scene has not succeeded on this context");
    int w, h;
    if (camera_size(camera, &w, &h) != RT_OK)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "camera[12..13] must hold an integral width/height in 1..65536");
    if (max_bounces < 0 || band_rows <= 0 || world <= 0 || rank < 0 || rank >= world)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad max_bounces / band_rows / rank / world");
    if (((uintptr_t)d_rgba8 & 3) != 0) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "d_rgba8 must be 4-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    for (int f = 1; f < n_frames; f++)
        if (camera[14 * f + 12] != camera[12] || camera[14 * f + 13] != camera[13])
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "the frames of one batch must have one size");
    RenderArgs a;
    a.scene = ctx->scene;
    std::memcpy(a.cam, camera, sizeof a.cam);
    std::memset(a.more_cams, 0, sizeof a.more_cams);
    for (int f = 1; f < n_frames; f++) std::memcpy(a.more_cams[f - 1], camera + 14 * f, sizeof a.cam);
    a.n_frames = n_frames;
    a.frame_stride = frame_stride;
    a.max_bounces = max_bounces;
    a.width = w, a.height = h;
    a.local_rows = rt_local_rows(h, band_rows, rank, world);
    a.band_rows = band_rows, a.rank = rank, a.world = world;
    a.tiles_x = (w + TILE_X - 1) / TILE_X;
    a.tiles_y = (a.local_rows + TILE_Y - 1) / TILE_Y;
    a.supertile = ctx->supertile;
    // grid: whole super-tiles, their count padded to a multiple of the 8 XCD slots
    const int n_super = ((a.tiles_x + a.supertile - 1) / a.supertile) * ((a.tiles_y + a.supertile - 1) / a.supertile);
    const int n_blocks = ((n_super + 7) / 8) * 8 * a.supertile * a.supertile;
    a.blocks_per_frame = n_blocks;
    const int sx_n = (a.tiles_x + a.supertile - 1) / a.supertile;
    a.map.per = make_fastdiv((uint32_t)(a.supertile * a.supertile)), a.map.st = make_fastdiv((uint32_t)a.supertile);
    a.map.sxn = make_fastdiv((uint32_t)sx_n), a.map.frames = make_fastdiv((uint32_t)n_frames);
    a.map.band = make_fastdiv((uint32_t)band_rows);
    for (int f = 0; f < MAX_BATCH; f++) a.prio[f] = PriorityRect{1, 1, 0, 0};
    if (ctx->prioritise)
        for (int f = 0; f < n_frames; f++)
            a.prio[f] = priority_rect(a.scene, camera + 14 * f, w, a.local_rows, a.supertile * TILE_X);
    for (int f = 0; f < MAX_BATCH; f++) {
        const int rw = a.prio[f].x1 - a.prio[f].x0 + 1;
        a.prio_div[f].rw = make_fastdiv((uint32_t)std::max(rw, 1));
        a.prio_div[f].side = make_fastdiv((uint32_t)std::max(sx_n - rw, 1));
    }
    a.rgb = d_rgb;
    a.rgba8 = (uint32_t*)d_rgba8;
    a.primary_id = d_primary_id;
    a.final_id = d_final_id;
    a.bounces = d_bounces;
    a.stats = (unsigned long long*)d_stats;
    a.ticket = nullptr;
    a.ticket_base = 0;
    a.order = nullptr, a.heat_out = nullptr;
    if (a.tiles_y == 0) return RT_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool force_coop = (force_trav & FORCE_COOP) != 0;
    int trav = (force_trav & 0xFF) ? (force_trav & 0xFF) : pick_traversal(ctx, a.scene, max_bounces);
    // the staged pipeline indexes (pixel, light) work items with 32 bits: beyond that, the megakernel
    if (trav == RT_TRAVERSAL_WAVEFRONT &&
        (size_t)((w + 7) / 8) * ((a.local_rows + 7) / 8) * 64 * (size_t)std::max(a.scene.lights_len / 6, 1) >= 0xFFFFFFF0ull)
        trav = RT_TRAVERSAL_BVH;
    const bool use_wf = trav == RT_TRAVERSAL_WAVEFRONT, use_bvh = trav == RT_TRAVERSAL_BVH;
    const bool use_lds = use_bvh && !noalloc && !(flags & RT_RENDER_GLOBAL_MEMORY) && ctx->allow_lds && ctx->lds_capable && ctx->ticket.p;
    if (n_frames > 1 && ((flags & RT_RENDER_REFERENCE_ORDER) || use_wf || use_lds || (size_t)n_blocks * n_frames > 0x7FFFFFFFull)) {
        // one frame per launch for these paths
        for (int f = 0; f < n_frames; f++) {
            const size_t off = (size_t)f * frame_stride;
            const int rc = render_frames(ctx, camera + 14 * f, 1, 0, max_bounces, flags, band_rows, rank, world,
                                         d_rgb ? d_rgb + 3 * off : nullptr, d_rgba8 ? d_rgba8 + 4 * off : nullptr,
                                         d_primary_id ? d_primary_id + off : nullptr, d_final_id ? d_final_id + off : nullptr,
                                         d_bounces ? d_bounces + off : nullptr, d_stats, stream, force_trav, noalloc);
            if (rc != RT_OK) return rc;
        }
        return RT_OK;
    }
    if (flags & RT_RENDER_REFERENCE_ORDER)
        hipLaunchKernelGGL(k_render<true>, dim3(n_blocks), dim3(64 * BLOCK_WAVES), 0, st, a);
    else if (use_wf)
        return launch_wavefront(ctx, a, st, noalloc);
    else if (use_lds) {
        // LDS-resident persistent kernel: one workgroup per CU; each of its waves draws one ticket
        // past the end, so the launch consumes exactly n_tickets + n_waves tickets
        const int per = a.supertile * a.supertile;
        const unsigned int n_tickets = (unsigned int)(n_super * per);
        const int grid = (int)std::min<unsigned int>((unsigned int)ctx->n_cus, (n_tickets + 15) / 16);
        a.ticket = (unsigned int*)ctx->ticket.p;
        a.ticket_base = ctx->ticket_base;
        ctx->ticket_base += n_tickets + (unsigned int)grid * (LDS_BLOCK / 64);
        int rc = chain_begin(ctx, st);  // the ticket arithmetic assumes launches run in issue order
        if (rc != RT_OK) return rc;
        hipLaunchKernelGGL(k_render_bvh_lds, dim3(grid), dim3(LDS_BLOCK), ctx->lds_bytes, st, a);
        if ((rc = chain_end(ctx, st)) != RT_OK) return rc;
    } else if (use_bvh) {
        const bool unb = ctx->info.unbounded_triangles > 0;
        const size_t stacks = (size_t)BLOCK_WAVES * 64 * a.scene.leaf_levels * 4;
        const bool coop = (force_coop || ctx->coop_shadows == 1) && a.scene.lights_len / 6 <= COOP_MAX_LIGHTS && a.scene.lights_len > 0;
        void (*kern)(const RenderArgs) = coop ? (unb ? k_render_bvh<true, true> : k_render_bvh<false, true>)
                                              : (unb ? k_render_bvh<true, false> : k_render_bvh<false, false>);
        // order by the previous frame (rt_kernels.h): single frames whose paths may bounce RT_HEAT_BOUNCES times
        // -- where the root box gives no priority columns (it spans the frame: TestScene1 0.398 -> 0.314 ms; with the columns
        // TeapotTable's long waves start first anyway and the bookkeeping costs 1 %, and on the teapot frames, whose
        // paths bounce once, the columns beat it: 0.221 / 0.568 ms against 0.250 / 0.704)
        if (ctx->use_heat && !noalloc && n_frames == 1 && max_bounces >= RT_HEAT_BOUNCES && BLOCK_WAVES == 1 &&
            a.prio[0].x0 > a.prio[0].x1) {
            const int n_pad = ((n_super + 7) / 8) * 8;
            const long long key[6] = {(long long)ctx->scene_serial, w, a.local_rows, a.supertile, max_bounces, (long long)rank * 65536 + world};
            const bool same = ctx->heat_valid && std::memcmp(key, ctx->heat_key, sizeof key) == 0;
            int rc;
            if ((rc = ensure(ctx, ctx->heat[0], (size_t)n_pad * 4)) || (rc = ensure(ctx, ctx->heat[1], (size_t)n_pad * 4)) ||
                (rc = ensure(ctx, ctx->order, (size_t)n_pad * 4)))
                return rc;
            if ((rc = chain_begin(ctx, st)) != RT_OK) return rc;  // the marks and the table belong to one launch at a time
            const int prev = ctx->heat_cur, cur = prev ^ 1;
            if (same) {
                hipLaunchKernelGGL(k_rank_supers, dim3(1), dim3(1024), 0, st, (const uint32_t*)ctx->heat[prev].p, n_pad, (int32_t*)ctx->order.p,
                                   (uint32_t*)ctx->heat[cur].p);
                a.order = (const int32_t*)ctx->order.p;
            } else {
                HIP_TRY(ctx, hipMemsetAsync(ctx->heat[cur].p, 0, (size_t)n_pad * 4, st));
            }
            a.heat_out = (uint32_t*)ctx->heat[cur].p;
            ctx->heat_cur = cur, ctx->heat_valid = true;
            std::memcpy(ctx->heat_key, key, sizeof key);
            void (*hkern)(const RenderArgs) = coop ? (unb ? k_render_bvh<true, true, true> : k_render_bvh<false, true, true>)
                                                   : (unb ? k_render_bvh<true, false, true> : k_render_bvh<false, false, true>);
            hipLaunchKernelGGL(hkern, dim3(n_blocks * n_frames), dim3(64 * BLOCK_WAVES),
                               stacks + (coop ? (size_t)BLOCK_WAVES * COOP_DWORDS * 4 : 0), st, a);
            HIP_TRY(ctx, hipGetLastError());
            return chain_end(ctx, st);
        }
        hipLaunchKernelGGL(kern, dim3(n_blocks * n_frames), dim3(64 * BLOCK_WAVES),
                           stacks + (coop ? (size_t)BLOCK_WAVES * COOP_DWORDS * 4 : 0), st, a);
    }
    else
        hipLaunchKernelGGL(k_render<false>, dim3(n_blocks * n_frames), dim3(64 * BLOCK_WAVES), 0, st, a);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

static int render_device_one(rt_ctx* ctx, const float camera[14], int max_bounces, int flags, int band_rows, int rank,
                             int world, float* d_rgb, uint8_t* d_rgba8, int32_t* d_primary_id, int32_t* d_final_id,
                             int32_t* d_bounces, rt_stats* d_stats, void* stream);
int rt_render_device(rt_ctx* ctx, const float camera[14], int max_bounces, int flags, int band_rows, int rank,
                     int world, float* d_rgb, uint8_t* d_rgba8, int32_t* d_primary_id, int32_t* d_final_id,
                     int32_t* d_bounces, rt_stats* d_stats, void* stream) {
    // a multi-GPU context asked for a WHOLE frame (rank 0 of world 1, no bounce plane, no counters): every device renders
    // its bands, the planes are assembled in the caller's buffers on the root device (rt_multi.h).  The per-device
    // contexts inside it are plain contexts (multi == nullptr) and take the path below.
    if (ctx && ctx->multi && rank == 0 && world == 1 && !(flags & RT_RENDER_REFERENCE_ORDER) && !d_bounces && !d_stats)
        return multi_render_device(ctx, camera, max_bounces, d_rgb, d_rgba8, d_primary_id, d_final_id, stream);
    return render_device_one(ctx, camera, max_bounces, flags, band_rows, rank, world, d_rgb, d_rgba8, d_primary_id, d_final_id,
                             d_bounces, d_stats, stream);
}
// rt_render_device on ONE device (also what a multi-GPU context runs on each of its devices, the root included)
static int render_device_one(rt_ctx* ctx, const float camera[14], int max_bounces, int flags, int band_rows, int rank,
                             int world, float* d_rgb, uint8_t* d_rgba8, int32_t* d_primary_id, int32_t* d_final_id,
                             int32_t* d_bounces, rt_stats* d_stats, void* stream) {
    int timed = -1, force = 0, w, h;
    if (ctx && ctx->has_scene && !(flags & RT_RENDER_REFERENCE_ORDER) && camera_size(camera, &w, &h) == RT_OK &&
        band_rows > 0 && world > 0 && rank >= 0 && rank < world && hipSetDevice(ctx->device) == hipSuccess)
        force = tune_begin(ctx, w, rt_local_rows(h, band_rows, rank, world), max_bounces, (hipStream_t)stream, &timed);
    const int rc = render_frames(ctx, camera, 1, 0, max_bounces, flags, band_rows, rank, world, d_rgb, d_rgba8,
                                 d_primary_id, d_final_id, d_bounces, d_stats, stream, force);
    tune_end(ctx, timed, (hipStream_t)stream);
    return rc;
}

int rt_render_device_noalloc(rt_ctx* ctx, const float camera[14], int max_bounces, int band_rows, int rank, int world,
                             float* d_rgb, uint8_t* d_rgba8, int32_t* d_primary_id, int32_t* d_final_id,
                             int32_t* d_bounces, void* stream) {
    // The traversal is the one rt_render_device would take, without its timing of mid-sized scenes (a verdict
    // measured earlier is honoured): nothing here creates events, buffers or anything else.
    const int force = ctx ? tune_verdict(ctx, max_bounces) : 0;
    return render_frames(ctx, camera, 1, 0, max_bounces, RT_RENDER_DEFAULT, band_rows, rank, world, d_rgb, d_rgba8,
                         d_primary_id, d_final_id, d_bounces, nullptr, stream, force, true);
}

int rt_render_frames_device(rt_ctx* ctx, const float* cameras, int n_frames, int max_bounces, int flags, int band_rows,
                            int rank, int world, int64_t frame_stride, float* d_rgb, uint8_t* d_rgba8,
                            int32_t* d_primary_id, int32_t* d_final_id, int32_t* d_bounces, void* stream) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!cameras || n_frames <= 0 || frame_stride < 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null cameras, no frames or a negative frame stride");
    int w, h;
    if (camera_size(cameras, &w, &h) != RT_OK)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "camera[12..13] must hold an integral width/height in 1..65536");
    if (band_rows <= 0 || world <= 0 || rank < 0 || rank >= world)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad band_rows / rank / world");
    if (n_frames > 1 && (uint64_t)frame_stride < (uint64_t)rt_local_rows(h, band_rows, rank, world) * (uint64_t)w)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "frame_stride is smaller than one frame's local pixels");
    int timed = -1, force = 0;
    if (ctx->has_scene && !(flags & RT_RENDER_REFERENCE_ORDER) && n_frames == 1 && hipSetDevice(ctx->device) == hipSuccess)
        force = tune_begin(ctx, w, rt_local_rows(h, band_rows, rank, world), max_bounces, (hipStream_t)stream, &timed);
    else if (ctx->has_scene && !(flags & RT_RENDER_REFERENCE_ORDER))
        // Batches reuse a verdict -- except the cooperative megakernel's: it wins single frames by shortening their
        // tail, and the frames of a batch fill each other's tails (measured, 8 frames per launch: TeapotTable 0.370 ms
        // per frame plain / 0.412 cooperative, TestScene1 0.133 / 0.150, teapot 0.187 / 0.233; tools/sweep_probe.py)
        force = tune_verdict(ctx, max_bounces) & ~FORCE_COOP;
    for (int f0 = 0; f0 < n_frames; f0 += MAX_BATCH) {  // MAX_BATCH frames per launch
        const int n = std::min(MAX_BATCH, n_frames - f0);
        const size_t off = (size_t)f0 * (size_t)frame_stride;
        const int rc = render_frames(ctx, cameras + 14 * f0, n, (uint64_t)frame_stride, max_bounces, flags, band_rows, rank,
                                     world, d_rgb ? d_rgb + 3 * off : nullptr, d_rgba8 ? d_rgba8 + 4 * off : nullptr,
                                     d_primary_id ? d_primary_id + off : nullptr, d_final_id ? d_final_id + off : nullptr,
                                     d_bounces ? d_bounces + off : nullptr, nullptr, stream, force);
        if (rc != RT_OK) return rc;
    }
    tune_end(ctx, timed, (hipStream_t)stream);
    return RT_OK;
}

int rt_render(rt_ctx* ctx, const float camera[14], int max_bounces, float* rgb, uint8_t* rgba8,
              int32_t* primary_id, int32_t* final_id, rt_stats* stats) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (ctx->multi) return multi_render(ctx, camera, max_bounces, rgb, rgba8, primary_id, final_id, stats);
    int w, h;
    if (camera_size(camera, &w, &h) != RT_OK)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "camera[12..13] must hold an integral width/height in 1..65536");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)w * h;
    int rc;
    if (rgb && (rc = ensure(ctx, ctx->fb_rgb, px * 12))) return rc;
    if (rgba8 && (rc = ensure(ctx, ctx->fb_rgba8, px * 4))) return rc;
    if (primary_id && (rc = ensure(ctx, ctx->fb_pid, px * 4))) return rc;
    if (final_id && (rc = ensure(ctx, ctx->fb_fid, px * 4))) return rc;
    if (stats) {
        if ((rc = ensure(ctx, ctx->fb_stats, sizeof(rt_stats)))) return rc;
        HIP_TRY(ctx, hipMemset(ctx->fb_stats.p, 0, sizeof(rt_stats)));
    }
    rc = rt_render_device(ctx, camera, max_bounces, stats ? RT_RENDER_REFERENCE_ORDER : RT_RENDER_DEFAULT, h, 0, 1,
                          rgb ? (float*)ctx->fb_rgb.p : nullptr, rgba8 ? (uint8_t*)ctx->fb_rgba8.p : nullptr,
                          primary_id ? (int32_t*)ctx->fb_pid.p : nullptr,
                          final_id ? (int32_t*)ctx->fb_fid.p : nullptr, nullptr,
                          stats ? (rt_stats*)ctx->fb_stats.p : nullptr, nullptr);
    if (rc != RT_OK) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (rgb) HIP_TRY(ctx, hipMemcpy(rgb, ctx->fb_rgb.p, px * 12, hipMemcpyDeviceToHost));
    if (rgba8) HIP_TRY(ctx, hipMemcpy(rgba8, ctx->fb_rgba8.p, px * 4, hipMemcpyDeviceToHost));
    if (primary_id) HIP_TRY(ctx, hipMemcpy(primary_id, ctx->fb_pid.p, px * 4, hipMemcpyDeviceToHost));
    if (final_id) HIP_TRY(ctx, hipMemcpy(final_id, ctx->fb_fid.p, px * 4, hipMemcpyDeviceToHost));
    if (stats) HIP_TRY(ctx, hipMemcpy(stats, ctx->fb_stats.p, sizeof(rt_stats), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_unpermute_frames(rt_ctx* ctx, const void* d_gathered, void* d_out, int elem_size, int width, int height,
                        int band_rows, int world, int n_frames, void* stream) {
    if (!ctx || !d_gathered || !d_out) return RT_ERR_INVALID_ARGUMENT;
    if ((elem_size != 4 && elem_size != 12) || width <= 0 || height <= 0 || band_rows <= 0 || world <= 0 || n_frames <= 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "elem_size must be 4 or 12; sizes must be positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int dwords = elem_size / 4;
    const size_t total = (size_t)width * height * dwords * n_frames;
    if ((total + 255) / 256 > 0x7FFFFFFFull) return fail(ctx, RT_ERR_TOO_LARGE, "too many pixels for one un-permute launch");
    const int padded = rt_padded_rows(height, band_rows, world);
    hipLaunchKernelGGL(k_unpermute, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)d_gathered, (uint32_t*)d_out, dwords, width, height, band_rows, world,
                       padded, n_frames);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int rt_unpermute_bands(rt_ctx* ctx, const void* d_gathered, void* d_out, int elem_size, int width, int height,
                       int band_rows, int world, void* stream) {
    return rt_unpermute_frames(ctx, d_gathered, d_out, elem_size, width, height, band_rows, world, 1, stream);
}

// ---- per-function evaluators -------------------------------------------------------------
static int eval_common(rt_ctx* ctx, int n, const void* const* in, const int* in_floats, int n_in, void* out,
                       int out_bytes_per, int which) {
    if (!ctx || n < 0) return RT_ERR_INVALID_ARGUMENT;
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    for (int k = 0; k < n_in; k++) {
        if (!in[k]) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null input");
        if ((rc = upload(ctx, ctx->ev[k], in[k], (size_t)n * in_floats[k] * 4))) return rc;
    }
    if ((rc = ensure(ctx, ctx->ev[7], (size_t)n * out_bytes_per))) return rc;
    const dim3 g((n + 255) / 256), b(256);
    auto F = [&](int k) { return (const float*)ctx->ev[k].p; };
    switch (which) {
        case 0: hipLaunchKernelGGL(k_eval_box, g, b, 0, 0, n, F(0), F(1), F(2), (int32_t*)ctx->ev[7].p); break;
        case 1: hipLaunchKernelGGL(k_eval_sphere, g, b, 0, 0, n, F(0), F(1), F(2), (float4*)ctx->ev[7].p); break;
        case 2: hipLaunchKernelGGL(k_eval_plane, g, b, 0, 0, n, F(0), F(1), F(2), F(3), (float*)ctx->ev[7].p); break;
        case 3:
            hipLaunchKernelGGL(k_eval_triangle, g, b, 0, 0, n, F(0), F(1), F(2), F(3), F(4), F(5),
                               (float*)ctx->ev[7].p);
            break;
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(out, ctx->ev[7].p, (size_t)n * out_bytes_per, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_eval_box(rt_ctx* ctx, int n, const float* o3, const float* d3, const float* box6, int32_t* hit) {
    const void* in[] = {o3, d3, box6};
    const int fl[] = {3, 3, 6};
    return eval_common(ctx, n, in, fl, 3, hit, 4, 0);
}
int rt_eval_sphere(rt_ctx* ctx, int n, const float* o3, const float* d3, const float* c3r, float* out4) {
    const void* in[] = {o3, d3, c3r};
    const int fl[] = {3, 3, 4};
    return eval_common(ctx, n, in, fl, 3, out4, 16, 1);
}
int rt_eval_plane(rt_ctx* ctx, int n, const float* o3, const float* d3, const float* p3, const float* n3, float* t) {
    const void* in[] = {o3, d3, p3, n3};
    const int fl[] = {3, 3, 3, 3};
    return eval_common(ctx, n, in, fl, 4, t, 4, 2);
}
int rt_eval_triangle(rt_ctx* ctx, int n, const float* o3, const float* d3, const float* a3, const float* b3,
                     const float* c3, const float* n3, float* t) {
    const void* in[] = {o3, d3, a3, b3, c3, n3};
    const int fl[] = {3, 3, 3, 3, 3, 3};
    return eval_common(ctx, n, in, fl, 6, t, 4, 3);
}

}  // extern "C"

#include "rt_multi.h"

#ifdef RT_DIAG_WF  // diagnostic build only: read and clear the trace stages' trip counters (tools/diag_wf.py)
extern "C" int rt_debug_wf_diag(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rtd::rt_wf_diag), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long zero[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(rtd::rt_wf_diag), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif
