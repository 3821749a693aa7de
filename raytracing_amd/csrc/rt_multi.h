// rt_multi.h -- one context over several GPUs of a node (rt_create_multi, include/rt_mi355x.h).
// Included by rt_mi355x.hip after the single-GPU render core; not a stand-alone header.
//
// the reference's host draws one frame per call on one GPU (RayTracer.cs:125-137).  Here one
// call still returns one finished frame, but the frame is cut into 8-row bands dealt round-robin to the
// context's devices (the scene is replicated: read-only, <= ~130 MB), every device renders its bands on its own
// stream, and the bands are gathered on the first device over RCCL -- grouped ncclSend / ncclRecv, so each
// peer's bands travel that peer's own xGMI link to the root -- where the un-permute kernel restores scanline
// order.  Single process, single host thread, ncclCommInitAll.
//
// RCCL is loaded with dlopen when the first multi-GPU context is created: single-GPU users of the library
// never load it, and a host process that already holds an RCCL (PyTorch ships its own) is not handed a second
// copy through the link line.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool load() {
        if (lib) return true;
        // RT_RCCL_LIBRARY names the library to load instead (tests use it to force the failure path)
        const char* forced = std::getenv("RT_RCCL_LIBRARY");
        if (forced && *forced) {
            lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (lib) break;
            }
        }
        if (!lib) {
            const char* e = dlerror();  // (one call: dlerror clears the state it reports)
            error = std::string("cannot load librccl.so: ") + (e ? e : "?");
            return false;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(lib, n);
            if (!p) error = std::string("librccl.so lacks ") + n;
            return p;
        };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;  // process-wide; contexts are created from one host thread at a time (rt_mi355x.h)

constexpr int MULTI_BAND_ROWS = 8;  // one row of 8x8-pixel tiles per band: finest interleave of sky / mesh / floor rows

}  // namespace

// Per-device state of a multi-GPU context.  sub[0] is the owning context itself (the root: frames are assembled
// on its device); sub[1..] are ordinary single-GPU contexts of the other devices.
struct rt_multi {
    int n = 0;
    std::vector<rt_ctx*> sub;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;  // empty when the exchange does not go over RCCL (see use_rccl)
    // device ids all different -> RCCL.  A list that names one device more than once cannot form a communicator
    // ("duplicate GPU"); it is accepted so that every rank can be played on one GPU (tests), and its bands then move
    // with device-to-device copies in stream order.  Everything else -- partition, launches, assembly -- is the same.
    bool use_rccl = false;
    struct Planes {
        DevBuf rgb, rgba8, pid, fid, stats;
    };
    std::vector<Planes> local;  // each device's own bands: rt_padded_rows() x W pixels per plane
    DevBuf gathered;            // root: [n][padded rows][W] elements of the plane being assembled
    std::vector<hipEvent_t> done;  // copy path: device d's render has finished
    hipEvent_t enter = nullptr, leave = nullptr;  // root device: rt_render_device on this context joins the caller's stream
};

namespace {

int nccl_fail(rt_ctx* ctx, ncclResult_t r, const char* what) {
    return fail(ctx, RT_ERR_DEVICE, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}
#define NCCL_TRY(ctx, expr)                                       \
    do {                                                          \
        ncclResult_t r_ = (expr);                                 \
        if (r_ != ncclSuccess) return nccl_fail(ctx, r_, #expr);  \
    } while (0)

}  // namespace

static void multi_release(rt_ctx* root) {
    rt_multi* m = root->multi;
    if (!m) return;
    for (int d = 0; d < (int)m->sub.size(); d++) {
        rt_ctx* c = m->sub[d];
        if (!c) continue;
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        if (d < (int)m->comm.size() && m->comm[d]) (void)g_rccl.CommDestroy(m->comm[d]);
        if (d < (int)m->local.size()) {
            release(m->local[d].rgb), release(m->local[d].rgba8), release(m->local[d].pid), release(m->local[d].fid);
            release(m->local[d].stats);
        }
        if (d < (int)m->done.size() && m->done[d]) (void)hipEventDestroy(m->done[d]);
        if (d < (int)m->stream.size() && m->stream[d]) (void)hipStreamDestroy(m->stream[d]);
    }
    (void)hipSetDevice(root->device);
    if (m->enter) (void)hipEventDestroy(m->enter);
    if (m->leave) (void)hipEventDestroy(m->leave);
    release(m->gathered);
    for (int d = 1; d < (int)m->sub.size(); d++)
        if (m->sub[d]) rt_destroy(m->sub[d]);
    root->multi = nullptr;
    delete m;
}

static int multi_upload(rt_ctx* root, const rtp::Prepared& prep, const void* spheres, int n_spheres, const void* planes, int n_planes,
                        const void* triangles, int n_triangles, const float* lights, int lights_len) {
    rt_multi* m = root->multi;
    for (int d = 0; d < m->n; d++) {
        rt_ctx* c = m->sub[d];
        c->build_tight = root->build_tight, c->wf_min_nodes = root->wf_min_nodes, c->tight_min_nodes = root->tight_min_nodes;
        const int rc = upload_prepared(c, prep, spheres, n_spheres, planes, n_planes, triangles, n_triangles, lights, lights_len);
        if (rc != RT_OK) {
            if (c != root) root->error = "device " + std::to_string(c->device) + ": " + c->error;
            root->has_scene = false;
            return rc;
        }
    }
    return RT_OK;
}

// The requested planes' bands from every device to the root in ONE exchange, then scanline order into the root
// device buffers out[k].  RCCL: a single group holds every (plane, device) send/receive pair -- each device sends its
// chunk of each plane to rank 0, rank 0 receives n chunks per plane (its own included: a send to self inside a group is
// a local copy), all progressing concurrently on their own links.
struct MultiPlane {
    size_t elem;                       // bytes per pixel
    DevBuf rt_multi::Planes::*plane;   // where every device rendered it
    void* out;                         // root device buffer, H x W pixels
};
static int multi_gather_planes(rt_ctx* root, const MultiPlane* planes, int n_planes, int w, int h) {
    rt_multi* m = root->multi;
    const int padded = rt_padded_rows(h, MULTI_BAND_ROWS, m->n);
    size_t total = 0;
    for (int k = 0; k < n_planes; k++) total += (size_t)padded * w * planes[k].elem * m->n;
    HIP_TRY(root, hipSetDevice(root->device));
    int rc = ensure(root, m->gathered, total);
    if (rc != RT_OK) return rc;
    unsigned char* g = (unsigned char*)m->gathered.p;
    if (m->use_rccl) {
        NCCL_TRY(root, g_rccl.GroupStart());
        ncclResult_t bad = ncclSuccess;
        const char* what = "";
        size_t at = 0;
        for (int k = 0; k < n_planes && bad == ncclSuccess; k++) {
            const size_t chunk = (size_t)padded * w * planes[k].elem;
            for (int d = 0; d < m->n && bad == ncclSuccess; d++) {
                const void* src = (m->local[d].*(planes[k].plane)).p;
                if ((bad = g_rccl.Send(src, chunk, ncclUint8, 0, m->comm[d], m->stream[d])) != ncclSuccess) what = "ncclSend";
                else if ((bad = g_rccl.Recv(g + at + chunk * d, chunk, ncclUint8, d, m->comm[0], m->stream[0])) != ncclSuccess) what = "ncclRecv";
            }
            at += chunk * m->n;
        }
        // a group that was opened is always closed, whatever happened inside it
        const ncclResult_t end = g_rccl.GroupEnd();
        if (bad != ncclSuccess) return nccl_fail(root, bad, what);
        if (end != ncclSuccess) return nccl_fail(root, end, "ncclGroupEnd");
    } else {
        for (int d = 0; d < m->n; d++) HIP_TRY(root, hipStreamWaitEvent(m->stream[0], m->done[d], 0));
        size_t at = 0;
        for (int k = 0; k < n_planes; k++) {
            const size_t chunk = (size_t)padded * w * planes[k].elem;
            for (int d = 0; d < m->n; d++)
                HIP_TRY(root, hipMemcpyAsync(g + at + chunk * d, (m->local[d].*(planes[k].plane)).p, chunk, hipMemcpyDeviceToDevice, m->stream[0]));
            at += chunk * m->n;
        }
    }
    size_t at = 0;
    for (int k = 0; k < n_planes; k++) {
        rc = rt_unpermute_bands(root, g + at, planes[k].out, (int)planes[k].elem, w, h, MULTI_BAND_ROWS, m->n, m->stream[0]);
        if (rc != RT_OK) return rc;
        at += (size_t)padded * w * planes[k].elem * m->n;
    }
    return RT_OK;
}

// Every device renders its bands into its own compact planes, on its own stream (after `after`, if given: an event
// on the ROOT device that the caller's stream has recorded).
static int multi_launch_bands(rt_ctx* root, const float* camera, int max_bounces, int w, int h, bool rgb, bool rgba8, bool pid,
                              bool fid, bool stats, hipEvent_t after) {
    rt_multi* m = root->multi;
    const int padded = rt_padded_rows(h, MULTI_BAND_ROWS, m->n);
    const size_t lpx = (size_t)padded * w;
    int rc;
    for (int d = 0; d < m->n; d++) {
        rt_ctx* c = m->sub[d];
        rt_multi::Planes& L = m->local[d];
        HIP_TRY(root, hipSetDevice(c->device));
        if ((rgb && (rc = ensure(c, L.rgb, lpx * 12))) || (rgba8 && (rc = ensure(c, L.rgba8, lpx * 4))) ||
            (pid && (rc = ensure(c, L.pid, lpx * 4))) || (fid && (rc = ensure(c, L.fid, lpx * 4))) ||
            (stats && (rc = ensure(c, L.stats, sizeof(rt_stats))))) {
            root->error = c->error;
            return rc;
        }
        if (after) HIP_TRY(root, hipStreamWaitEvent(m->stream[d], after, 0));
        if (stats) HIP_TRY(root, hipMemsetAsync(L.stats.p, 0, sizeof(rt_stats), m->stream[d]));
        c->traversal = root->traversal, c->supertile = root->supertile, c->allow_lds = false, c->coop_shadows = root->coop_shadows;
        c->use_heat = root->use_heat;
        rc = render_device_one(c, camera, max_bounces, stats ? RT_RENDER_REFERENCE_ORDER : RT_RENDER_DEFAULT, MULTI_BAND_ROWS, d,
                              m->n, rgb ? (float*)L.rgb.p : nullptr, rgba8 ? (uint8_t*)L.rgba8.p : nullptr,
                              pid ? (int32_t*)L.pid.p : nullptr, fid ? (int32_t*)L.fid.p : nullptr, nullptr,
                              stats ? (rt_stats*)L.stats.p : nullptr, m->stream[d]);
        if (rc != RT_OK) {
            if (c != root) root->error = "device " + std::to_string(c->device) + ": " + c->error;
            return rc;
        }
        if (!m->use_rccl) HIP_TRY(root, hipEventRecord(m->done[d], m->stream[d]));
    }
    HIP_TRY(root, hipSetDevice(root->device));
    return RT_OK;
}

static void multi_drain(rt_ctx* root) {
    rt_multi* m = root->multi;
    const std::string first = root->error;  // keep the first failure's message
    for (int d = 0; d < m->n; d++)
        if (m->sub[d] && m->stream[d] && hipSetDevice(m->sub[d]->device) == hipSuccess) (void)hipStreamSynchronize(m->stream[d]);
    (void)hipGetLastError();
    (void)hipSetDevice(root->device);
    root->error = first;
}

// rt_render_device on a multi-GPU context (whole frame: rank 0 of world 1): the finished planes land in the caller's
// buffers on the ROOT device, and the call is asynchronous on `stream` (a stream of the root device): the devices'
// own streams start after what `stream` holds so far, and `stream` continues after the assembled planes are in place.
static int multi_render_device_impl(rt_ctx* root, const float* camera, int max_bounces, float* d_rgb, uint8_t* d_rgba8,
                                    int32_t* d_primary_id, int32_t* d_final_id, hipStream_t stream) {
    rt_multi* m = root->multi;
    int w, h;
    if (camera_size(camera, &w, &h) != RT_OK)
        return fail(root, RT_ERR_INVALID_ARGUMENT, "camera[12..13] must hold an integral width/height in 1..65536");
    if (!root->has_scene) return fail(root, RT_ERR_NO_SCENE, "rt_upload_scene has not succeeded on this context");
    HIP_TRY(root, hipSetDevice(root->device));
    HIP_TRY(root, hipEventRecord(m->enter, stream));
    int rc = multi_launch_bands(root, camera, max_bounces, w, h, d_rgb != nullptr, d_rgba8 != nullptr, d_primary_id != nullptr,
                                d_final_id != nullptr, false, m->enter);
    if (rc != RT_OK) return rc;
    MultiPlane planes[4];
    int n = 0;
    if (d_rgb) planes[n++] = MultiPlane{12, &rt_multi::Planes::rgb, d_rgb};
    if (d_rgba8) planes[n++] = MultiPlane{4, &rt_multi::Planes::rgba8, d_rgba8};
    if (d_primary_id) planes[n++] = MultiPlane{4, &rt_multi::Planes::pid, d_primary_id};
    if (d_final_id) planes[n++] = MultiPlane{4, &rt_multi::Planes::fid, d_final_id};
    if (n && (rc = multi_gather_planes(root, planes, n, w, h)) != RT_OK) return rc;
    HIP_TRY(root, hipSetDevice(root->device));
    HIP_TRY(root, hipEventRecord(m->leave, m->stream[0]));
    HIP_TRY(root, hipStreamWaitEvent(stream, m->leave, 0));
    // (the other devices' streams have nothing after their sends; the root stream's receive + assembly is what `stream` waits for)
    return RT_OK;
}
static int multi_render_device(rt_ctx* root, const float* camera, int max_bounces, float* d_rgb, uint8_t* d_rgba8,
                               int32_t* d_primary_id, int32_t* d_final_id, void* stream) {
    const int rc = multi_render_device_impl(root, camera, max_bounces, d_rgb, d_rgba8, d_primary_id, d_final_id, (hipStream_t)stream);
    if (rc != RT_OK) multi_drain(root);
    return rc;
}

static int multi_render_impl(rt_ctx* root, const float* camera, int max_bounces, float* rgb, uint8_t* rgba8, int32_t* primary_id,
                             int32_t* final_id, rt_stats* stats);
// rt_render on a multi-GPU context.  Whatever goes wrong on the way, the call does not return while any device's
// stream still has work of this frame in flight (the caller may free or reuse its arrays at once).
static int multi_render(rt_ctx* root, const float* camera, int max_bounces, float* rgb, uint8_t* rgba8, int32_t* primary_id,
                        int32_t* final_id, rt_stats* stats) {
    const int rc = multi_render_impl(root, camera, max_bounces, rgb, rgba8, primary_id, final_id, stats);
    if (rc != RT_OK) multi_drain(root);
    return rc;
}
static int multi_render_impl(rt_ctx* root, const float* camera, int max_bounces, float* rgb, uint8_t* rgba8, int32_t* primary_id,
                             int32_t* final_id, rt_stats* stats) {
    rt_multi* m = root->multi;
    int w, h;
    if (camera_size(camera, &w, &h) != RT_OK)
        return fail(root, RT_ERR_INVALID_ARGUMENT, "camera[12..13] must hold an integral width/height in 1..65536");
    if (!root->has_scene) return fail(root, RT_ERR_NO_SCENE, "rt_upload_scene has not succeeded on this context");
    const size_t px = (size_t)w * h;
    int rc;
    HIP_TRY(root, hipSetDevice(root->device));
    if ((rgb && (rc = ensure(root, root->fb_rgb, px * 12))) || (rgba8 && (rc = ensure(root, root->fb_rgba8, px * 4))) ||
        (primary_id && (rc = ensure(root, root->fb_pid, px * 4))) || (final_id && (rc = ensure(root, root->fb_fid, px * 4))))
        return rc;
    if ((rc = multi_launch_bands(root, camera, max_bounces, w, h, rgb != nullptr, rgba8 != nullptr, primary_id != nullptr,
                                 final_id != nullptr, stats != nullptr, nullptr)) != RT_OK)
        return rc;
    // gather + assemble the requested planes on the root (one exchange), then hand them to the host
    struct Job {
        bool want;
        size_t elem;
        DevBuf rt_multi::Planes::*plane;
        DevBuf* fb;
        void* host;
    } jobs[] = {{rgb != nullptr, 12, &rt_multi::Planes::rgb, &root->fb_rgb, rgb},
                {rgba8 != nullptr, 4, &rt_multi::Planes::rgba8, &root->fb_rgba8, rgba8},
                {primary_id != nullptr, 4, &rt_multi::Planes::pid, &root->fb_pid, primary_id},
                {final_id != nullptr, 4, &rt_multi::Planes::fid, &root->fb_fid, final_id}};
    MultiPlane planes[4];
    int n = 0;
    for (const Job& j : jobs)
        if (j.want) planes[n++] = MultiPlane{j.elem, j.plane, j.fb->p};
    if (n && (rc = multi_gather_planes(root, planes, n, w, h)) != RT_OK) return rc;
    HIP_TRY(root, hipSetDevice(root->device));
    for (const Job& j : jobs)
        if (j.want) HIP_TRY(root, hipMemcpyAsync(j.host, j.fb->p, px * j.elem, hipMemcpyDeviceToHost, m->stream[0]));
    if (stats) {  // every device counted its own bands: the frame's counters are their sum
        std::memset(stats, 0, sizeof *stats);
        for (int d = 0; d < m->n; d++) {
            rt_stats part;
            HIP_TRY(root, hipSetDevice(m->sub[d]->device));
            HIP_TRY(root, hipStreamSynchronize(m->stream[d]));
            HIP_TRY(root, hipMemcpy(&part, m->local[d].stats.p, sizeof part, hipMemcpyDeviceToHost));
            const uint64_t* s = (const uint64_t*)&part;
            uint64_t* t = (uint64_t*)stats;
            for (size_t k = 0; k < sizeof(rt_stats) / 8; k++) t[k] += s[k];
        }
    }
    for (int d = 0; d < m->n; d++) {  // the call returns a finished frame (and idle devices), like the reference's draw
        HIP_TRY(root, hipSetDevice(m->sub[d]->device));
        HIP_TRY(root, hipStreamSynchronize(m->stream[d]));
    }
    HIP_TRY(root, hipSetDevice(root->device));
    return RT_OK;
}

extern "C" int rt_create_multi(const int* device_ids, int n_devices, rt_ctx** out) {
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return RT_ERR_INVALID_ARGUMENT;
    rt_ctx* root = nullptr;
    int rc = rt_create(device_ids[0], &root);
    if (rc != RT_OK) return rc;
    rt_multi* m = new rt_multi();
    root->multi = m;
    m->n = n_devices;
    m->sub.assign(n_devices, nullptr), m->stream.assign(n_devices, nullptr), m->done.assign(n_devices, nullptr);
    m->local.resize(n_devices);
    m->sub[0] = root;
    bool distinct = true;
    for (int d = 0; d < n_devices; d++)
        for (int e = 0; e < d; e++) distinct &= device_ids[d] != device_ids[e];
    m->use_rccl = distinct;
    for (int d = 0; d < n_devices && rc == RT_OK; d++) {
        if (d > 0) rc = rt_create(device_ids[d], &m->sub[d]);
        if (rc != RT_OK) break;
        if (hipSetDevice(device_ids[d]) != hipSuccess ||
            hipStreamCreateWithFlags(&m->stream[d], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->done[d], hipEventDisableTiming) != hipSuccess)
            rc = RT_ERR_DEVICE;
    }
    if (rc == RT_OK && (hipSetDevice(root->device) != hipSuccess ||
                        hipEventCreateWithFlags(&m->enter, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&m->leave, hipEventDisableTiming) != hipSuccess))
        rc = RT_ERR_DEVICE;
    if (rc == RT_OK && m->use_rccl) {
        if (!g_rccl.load()) {
            rc = RT_ERR_DEVICE;
            std::fprintf(stderr, "rt_create_multi: %s\n", g_rccl.error.c_str());
        } else {
            m->comm.assign(n_devices, nullptr);
            const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n_devices, device_ids);
            if (r != ncclSuccess) {
                std::fprintf(stderr, "rt_create_multi: ncclCommInitAll: %s\n", g_rccl.GetErrorString(r));
                m->comm.clear();
                rc = RT_ERR_DEVICE;
            }
        }
    }
    if (rc != RT_OK) {
        multi_release(root);
        rt_destroy(root);
        return rc;
    }
    (void)hipSetDevice(root->device);
    *out = root;
    return RT_OK;
}

extern "C" int rt_multi_info(const rt_ctx* ctx, int* n_devices, int* uses_rccl) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (n_devices) *n_devices = ctx->multi ? ctx->multi->n : 1;
    if (uses_rccl) *uses_rccl = ctx->multi && ctx->multi->use_rccl ? 1 : 0;
    return RT_OK;
}
