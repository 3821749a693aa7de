"""RayTracer: the host object of the reference (RayTracer.cs), with its three GL call sites
replaced by the C ABI of include/rt_mi355x.h.

    reference (RayTracer.cs)                     here
    Init(): CreateProgram/LoadShader  49-62   -> rt_create
    SendPrimitivesToShader()         957-1046 -> rt_upload_scene
    PrepareRenderOpenGL()            943-956  -> keeps the 14-float camera array
    RenderGL(): DrawArrays           125-137  -> rt_render / rt_render_device

The product path has no CPU fallback: if librt_mi355x.so is missing the import of the library
fails, and without a GPU rt_create fails.
"""
import ctypes as C

import numpy as np

from . import _native as N


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt error {code}: {message}")
        self.code = code


def check_accel(accel, n_spheres, n_triangles):
    """rt_check_accel: host-only validation of a serialised R-tree.  Returns the flattened sizes
    or raises RtError (no GPU needed)."""
    lib = N.device_lib()
    a = np.ascontiguousarray(accel, dtype=np.float32)
    info = N.rt_scene_info()
    err = C.create_string_buffer(256)
    rc = lib.rt_check_accel(N.ptr(a) if len(a) else None, len(a), int(n_spheres), int(n_triangles), C.byref(info),
                            err, 256)
    if rc != 0:
        raise RtError(rc, err.value.decode())
    return {n: int(getattr(info, n)) for n, _ in N.rt_scene_info._fields_ if n != "reserved"}


def check_scene(scene, tight_min_nodes=1000):
    """rt_check_scene: the host half of rt_upload_scene (validation, cull preconditions, both trees) plus a
    structural self-check of the trees.  Returns the scene info or raises RtError (no GPU needed)."""
    lib = N.device_lib()
    info = N.rt_scene_info()
    err = C.create_string_buffer(256)
    sp, tr, a = scene.spheresData, scene.trianglesData, scene.AccelerationStructureData
    rc = lib.rt_check_scene(N.ptr(sp) if len(sp) else None, len(sp), N.ptr(tr) if len(tr) else None, len(tr),
                            N.ptr(a) if len(a) else None, len(a), int(tight_min_nodes), C.byref(info), err, 256)
    if rc != 0:
        raise RtError(rc, err.value.decode())
    return {n: int(getattr(info, n)) for n, _ in N.rt_scene_info._fields_ if n != "reserved"}


class RayTracer:
    def __init__(self, device_id=0):
        """device_id: a HIP device ordinal (rt_create), or a list of them (rt_create_multi: one context over
        several GPUs, frames assembled on the first; RenderGL is then the multi-GPU render)."""
        self._lib = N.device_lib()
        self._ctx = C.c_void_p()
        if isinstance(device_id, (list, tuple)):
            ids = (C.c_int * len(device_id))(*[int(d) for d in device_id])
            rc = self._lib.rt_create_multi(ids, len(device_id), C.byref(self._ctx))
            if rc != 0:
                self._ctx = C.c_void_p()
                raise RtError(rc, f"rt_create_multi(devices {list(device_id)}) failed")
            device_id = device_id[0]
        else:
            rc = self._lib.rt_create(int(device_id), C.byref(self._ctx))
            if rc != 0:
                self._ctx = C.c_void_p()
                raise RtError(rc, f"rt_create(device {device_id}) failed (no MI355X visible?)")
        self.device_id = int(device_id)
        self.cameraData = None
        self.scene = None

    # ---- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(self._ctx).decode())

    OPTION_USE_LDS, OPTION_SUPERTILE, OPTION_TRAVERSAL, OPTION_TIGHT_TREE, OPTION_WAVEFRONT_MIN_NODES = 1, 2, 3, 4, 5
    OPTION_TIGHT_MIN_NODES, OPTION_COOP_SHADOWS, OPTION_GPU_FLATTEN_MIN, OPTION_FRAME_FEEDBACK, OPTION_BIN_PATHS, OPTION_GPU_BUILD_MIN, OPTION_PACKET_WALKS = 6, 7, 8, 9, 10, 11, 12
    TRAVERSAL_AUTO, TRAVERSAL_WAVE, TRAVERSAL_BVH, TRAVERSAL_WAVEFRONT = 0, 1, 2, 3

    def set_option(self, option, value):
        self._check(self._lib.rt_set_option(self._ctx, int(option), int(value)))

    # ---- RayTracer.cs:957-1046 ------------------------------------------------------------
    def SendPrimitivesToShader(self, scene):
        """scene: raytracing_amd.host.Scene (or any object with the same six buffers)."""
        s = np.ascontiguousarray(scene.spheresData, dtype=N.SPHERE_DT)
        p = np.ascontiguousarray(scene.planesData, dtype=N.PLANE_DT)
        t = np.ascontiguousarray(scene.trianglesData, dtype=N.TRIANGLE_DT)
        a = np.ascontiguousarray(scene.AccelerationStructureData, dtype=np.float32)
        li = np.ascontiguousarray(scene.lightsData, dtype=np.float32)
        self._check(self._lib.rt_upload_scene(self._ctx, N.ptr(s), len(s), N.ptr(p), len(p), N.ptr(t), len(t),
                                              N.ptr(a), len(a), N.ptr(li), len(li)))
        self.scene = scene
        if self.cameraData is None and getattr(scene, "cameraData", None) is not None:
            self.cameraData = np.ascontiguousarray(scene.cameraData, dtype=np.float32).copy()

    def upload_raw(self, spheres, planes, triangles, accel, lights):
        """rt_upload_scene with caller-owned arrays exactly as given (used by the negative tests)."""
        s = np.ascontiguousarray(spheres, dtype=N.SPHERE_DT)
        p = np.ascontiguousarray(planes, dtype=N.PLANE_DT)
        t = np.ascontiguousarray(triangles, dtype=N.TRIANGLE_DT)
        a = np.ascontiguousarray(accel, dtype=np.float32)
        li = np.ascontiguousarray(lights, dtype=np.float32)
        self._check(self._lib.rt_upload_scene(self._ctx, N.ptr(s), len(s), N.ptr(p), len(p), N.ptr(t), len(t),
                                              N.ptr(a), len(a), N.ptr(li), len(li)))

    # ---- RayTracer.cs:943-956 -------------------------------------------------------------
    def PrepareRenderOpenGL(self, cameraData):
        cam = np.ascontiguousarray(cameraData, dtype=np.float32)
        if cam.shape != (14,):
            raise ValueError("cameraData must hold 14 floats (RayTracer.cs:947-954)")
        self.cameraData = cam.copy()

    # ---- RayTracer.cs:125-137 -------------------------------------------------------------
    def RenderGL(self, max_bounces=10, want=("rgb", "rgba8", "primary_id", "final_id"), stats=False):
        """One frame to host arrays shaped [H, W, ...], row 0 = bottom row.  Returns a dict."""
        cam = self.cameraData
        # an invalid size is the library's to reject (RT_ERR_INVALID_ARGUMENT before anything is written)
        ok = np.isfinite(cam[12:14]).all() and cam[12] >= 1 and cam[13] >= 1 and cam[12] <= 65536 and cam[13] <= 65536
        W, H = (int(cam[12]), int(cam[13])) if ok else (1, 1)
        out = {}
        if "rgb" in want:
            out["rgb"] = np.empty((H, W, 3), np.float32)
        if "rgba8" in want:
            out["rgba8"] = np.empty((H, W, 4), np.uint8)
        if "primary_id" in want:
            out["primary_id"] = np.empty((H, W), np.int32)
        if "final_id" in want:
            out["final_id"] = np.empty((H, W), np.int32)
        st = N.rt_stats() if stats else None
        self._check(self._lib.rt_render(self._ctx, N.ptr(cam), int(max_bounces), N.ptr(out.get("rgb")),
                                        N.ptr(out.get("rgba8")), N.ptr(out.get("primary_id")),
                                        N.ptr(out.get("final_id")), C.byref(st) if stats else None))
        if stats:
            out["stats"] = {k: int(getattr(st, k)) for k in N.STATS_FIELDS}
        return out

    # ---- device-resident frame (bench, multi-GPU) -------------------------------------------
    def render_device(self, cameraData, max_bounces, *, rgb=None, rgba8=None, primary_id=None, final_id=None,
                      bounces=None, stats=None, reference_order=False, band_rows=None, rank=0, world=1, stream=0):
        """Asynchronous launch; the outputs are DEVICE pointers (ints, e.g. torch tensor.data_ptr())."""
        cam = np.ascontiguousarray(cameraData, dtype=np.float32)
        if band_rows is None:
            band_rows = int(cam[13])
        flags = 1 if reference_order else 0
        vp = lambda v: C.c_void_p(int(v)) if v else None  # noqa: E731
        self._check(self._lib.rt_render_device(self._ctx, N.ptr(cam), int(max_bounces), flags, int(band_rows),
                                               int(rank), int(world), vp(rgb), vp(rgba8), vp(primary_id),
                                               vp(final_id), vp(bounces), vp(stats), vp(stream)))

    NOALLOC_SCRATCH_BYTES = 96 << 20

    def render_device_noalloc(self, cameraData, max_bounces, *, rgb=None, rgba8=None, primary_id=None, final_id=None,
                              bounces=None, band_rows=None, rank=0, world=1, stream=0):
        """rt_render_device_noalloc: as render_device, but the staged pipeline's path state lives in the
        library's static device array (no hipMalloc); oversized frames raise RtError(RT_ERR_TOO_LARGE)."""
        cam = np.ascontiguousarray(cameraData, dtype=np.float32)
        if band_rows is None:
            band_rows = int(cam[13])
        vp = lambda v: C.c_void_p(int(v)) if v else None  # noqa: E731
        self._check(self._lib.rt_render_device_noalloc(self._ctx, N.ptr(cam), int(max_bounces), int(band_rows),
                                                       int(rank), int(world), vp(rgb), vp(rgba8), vp(primary_id),
                                                       vp(final_id), vp(bounces), vp(stream)))

    def render_frames_device(self, cameras, max_bounces, frame_stride, *, rgb=None, rgba8=None, primary_id=None,
                             final_id=None, bounces=None, reference_order=False, band_rows=None, rank=0, world=1,
                             stream=0):
        """rt_render_frames_device: several frames of one size, frame f's planes f * frame_stride pixels after
        frame 0's (device pointers); the megakernels take up to 8 frames per launch."""
        cams = np.ascontiguousarray(cameras, dtype=np.float32).reshape(-1, 14)
        if band_rows is None:
            band_rows = int(cams[0, 13])
        vp = lambda v: C.c_void_p(int(v)) if v else None  # noqa: E731
        self._check(self._lib.rt_render_frames_device(self._ctx, N.ptr(cams), len(cams), int(max_bounces),
                                                      1 if reference_order else 0, int(band_rows), int(rank),
                                                      int(world), int(frame_stride), vp(rgb), vp(rgba8),
                                                      vp(primary_id), vp(final_id), vp(bounces), vp(stream)))

    def unpermute_bands(self, gathered, out, elem_size, width, height, band_rows, world, stream=0):
        vp = lambda v: C.c_void_p(int(v)) if v else None  # noqa: E731
        self._check(self._lib.rt_unpermute_bands(self._ctx, vp(gathered), vp(out), int(elem_size), int(width),
                                                 int(height), int(band_rows), int(world), vp(stream)))

    def unpermute_frames(self, gathered, out, elem_size, width, height, band_rows, world, n_frames, stream=0):
        vp = lambda v: C.c_void_p(int(v)) if v else None  # noqa: E731
        self._check(self._lib.rt_unpermute_frames(self._ctx, vp(gathered), vp(out), int(elem_size), int(width),
                                                  int(height), int(band_rows), int(world), int(n_frames), vp(stream)))

    def local_rows(self, height, band_rows, rank, world):
        return self._lib.rt_local_rows(height, band_rows, rank, world)

    def padded_rows(self, height, band_rows, world):
        return self._lib.rt_padded_rows(height, band_rows, world)

    def multi_info(self):
        n, r = C.c_int(), C.c_int()
        self._check(self._lib.rt_multi_info(self._ctx, C.byref(n), C.byref(r)))
        return {"n_devices": n.value, "uses_rccl": bool(r.value)}

    def check_gpu_flatten(self, accel, n_spheres, n_triangles):
        """rt_check_gpu_flatten: host and device flatten of the array compared bit for bit.  Returns 1 (identical,
        flattened on the device) or 0 (too short for the device); raises RtError with the host's refusal when both
        refuse the array, or with the first difference."""
        a = np.ascontiguousarray(accel, dtype=np.float32)
        err = C.create_string_buffer(512)
        rc = self._lib.rt_check_gpu_flatten(self._ctx, N.ptr(a) if len(a) else None, len(a), int(n_spheres), int(n_triangles),
                                            err, len(err))
        if rc < 0:
            raise RtError(rc, err.value.decode() or self._lib.rt_last_error(self._ctx).decode())
        return rc

    def check_gpu_build(self, scene, tight_min_nodes=0):
        """rt_check_gpu_build: both trees of the scene built on the host and on the device, compared bit for bit.
        Returns 1 (identical) or 0 (the device declined a tree); raises RtError with the first difference."""
        s = np.ascontiguousarray(scene.spheresData, dtype=N.SPHERE_DT)
        t = np.ascontiguousarray(scene.trianglesData, dtype=N.TRIANGLE_DT)
        a = np.ascontiguousarray(scene.AccelerationStructureData, dtype=np.float32)
        err = C.create_string_buffer(512)
        rc = self._lib.rt_check_gpu_build(self._ctx, N.ptr(s) if len(s) else None, len(s), N.ptr(t) if len(t) else None, len(t),
                                          N.ptr(a) if len(a) else None, len(a), int(tight_min_nodes), err, len(err))
        if rc < 0:
            raise RtError(rc, err.value.decode() or self._lib.rt_last_error(self._ctx).decode())
        return rc

    def scene_info(self):
        info = N.rt_scene_info()
        self._check(self._lib.rt_get_scene_info(self._ctx, C.byref(info)))
        return {n: int(getattr(info, n)) for n, _ in N.rt_scene_info._fields_ if n != "reserved"}

    # ---- per-function evaluators ------------------------------------------------------------
    def _f32(self, a, cols):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] == cols
        return a

    def eval_box(self, o, d, box):
        o, d, box = self._f32(o, 3), self._f32(d, 3), self._f32(box, 6)
        hit = np.zeros(len(o), np.int32)
        self._check(self._lib.rt_eval_box(self._ctx, len(o), N.ptr(o), N.ptr(d), N.ptr(box), N.ptr(hit)))
        return hit

    def eval_sphere(self, o, d, c3r):
        o, d, c3r = self._f32(o, 3), self._f32(d, 3), self._f32(c3r, 4)
        out = np.zeros((len(o), 4), np.float32)
        self._check(self._lib.rt_eval_sphere(self._ctx, len(o), N.ptr(o), N.ptr(d), N.ptr(c3r), N.ptr(out)))
        return out

    def eval_plane(self, o, d, p, n):
        o, d, p, n = self._f32(o, 3), self._f32(d, 3), self._f32(p, 3), self._f32(n, 3)
        t = np.zeros(len(o), np.float32)
        self._check(self._lib.rt_eval_plane(self._ctx, len(o), N.ptr(o), N.ptr(d), N.ptr(p), N.ptr(n), N.ptr(t)))
        return t

    def eval_triangle(self, o, d, a, b, c, n):
        o, d, a, b, c, n = [self._f32(v, 3) for v in (o, d, a, b, c, n)]
        t = np.zeros(len(o), np.float32)
        self._check(self._lib.rt_eval_triangle(self._ctx, len(o), N.ptr(o), N.ptr(d), N.ptr(a), N.ptr(b), N.ptr(c),
                                               N.ptr(n), N.ptr(t)))
        return t
