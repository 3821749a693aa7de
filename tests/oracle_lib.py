"""ctypes wrapper around oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

SPHERE_DT = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("diffuse", "<f4", 3), ("pure", "<u4"),
                      ("specular", "<f4", 3), ("specularity", "<f4")])
PLANE_DT = np.dtype([("position", "<f4", 3), ("pad0", "<f4"), ("normal", "<f4", 3), ("pad1", "<f4"),
                     ("diffuse", "<f4", 3), ("pure", "<u4"), ("specular", "<f4", 3), ("specularity", "<f4")])
TRI_DT = np.dtype([("a", "<f4", 3), ("pad0", "<f4"), ("b", "<f4", 3), ("pad1", "<f4"), ("c", "<f4", 3),
                   ("pad2", "<f4"), ("n", "<f4", 3), ("pad3", "<f4"), ("diffuse", "<f4", 3), ("pure", "<u4"),
                   ("specular", "<f4", 3), ("specularity", "<f4")])
assert SPHERE_DT.itemsize == 48 and PLANE_DT.itemsize == 64 and TRI_DT.itemsize == 96

COUNTER_NAMES = ["primary", "secondary", "shadow", "node_tests", "refs_read", "sphere_tests", "plane_tests",
                 "tri_tests", "material", "max_stack", "max_sph_cand", "max_tri_cand", "zero_term"]

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int]
        L.orc_scene_testscene1.restype = C.c_void_p
        L.orc_scene_testscene1.argtypes = [C.c_char_p]
        L.orc_scene_from_structs.restype = C.c_void_p
        L.orc_scene_from_structs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_int]
        L.orc_scene_get.restype = C.c_int
        L.orc_scene_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.orc_scene_error.restype = C.c_char_p
        L.orc_scene_error.argtypes = [C.c_void_p]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_import_model.restype = C.c_int
        L.orc_import_model.argtypes = [C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int]
        L.orc_make_triangle.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_camera_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_float, C.c_void_p]
        L.orc_intersect_box.argtypes = [C.c_int] + [C.c_void_p] * 4
        L.orc_intersect_sphere.argtypes = [C.c_int] + [C.c_void_p] * 4
        L.orc_intersect_plane.argtypes = [C.c_int] + [C.c_void_p] * 5
        L.orc_intersect_triangle.argtypes = [C.c_int] + [C.c_void_p] * 7
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleScene:
    """Packed scene buffers as numpy arrays (copies)."""

    def __init__(self, handle):
        L = lib()
        err = L.orc_scene_error(handle).decode()
        self.error = err
        dts = [SPHERE_DT, PLANE_DT, TRI_DT, np.float32, np.float32, np.float32]
        out = []
        for which, dt in enumerate(dts):
            ptr, cnt = C.c_void_p(), C.c_int64()
            L.orc_scene_get(handle, which, C.byref(ptr), C.byref(cnt))
            n = cnt.value
            if n == 0:
                out.append(np.zeros(0, dtype=dt))
            else:
                nbytes = n * np.dtype(dt).itemsize
                buf = (C.c_char * nbytes).from_address(ptr.value)
                out.append(np.frombuffer(buf, dtype=dt).copy())
        self.spheres, self.planes, self.triangles, self.accel, self.lights, self.camera = out
        L.orc_scene_free(handle)


def testscene1(asset_dir):
    return OracleScene(lib().orc_scene_testscene1(str(asset_dir).encode()))


def scene_from_structs(spheres, planes, triangles, lights):
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DT)
    planes = np.ascontiguousarray(planes, dtype=PLANE_DT)
    triangles = np.ascontiguousarray(triangles, dtype=TRI_DT)
    lights = _f32(lights)
    h = lib().orc_scene_from_structs(_p(spheres), len(spheres), _p(planes), len(planes), _p(triangles),
                                     len(triangles), _p(lights), len(lights))
    return OracleScene(h)


def render(spheres, planes, triangles, accel, lights, camera, max_bounces, region=None, n_threads=1,
           want=("rgb", "rgba8", "primary_id", "final_id", "bounces")):
    """Returns dict of arrays shaped [h, w, ...] (row 0 = bottom row) plus 'counters' dict."""
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DT)
    planes = np.ascontiguousarray(planes, dtype=PLANE_DT)
    triangles = np.ascontiguousarray(triangles, dtype=TRI_DT)
    accel, lights, camera = _f32(accel), _f32(lights), _f32(camera)
    W, H = int(camera[12]), int(camera[13])
    x0, y0, x1, y1 = region if region is not None else (0, 0, W, H)
    w, h = x1 - x0, y1 - y0
    out = {}
    if "rgb" in want:
        out["rgb"] = np.zeros((h, w, 3), np.float32)
    if "rgba8" in want:
        out["rgba8"] = np.zeros((h, w, 4), np.uint8)
    for k in ("primary_id", "final_id", "bounces"):
        if k in want:
            out[k] = np.zeros((h, w), np.int32)
    counters = np.zeros(16, np.uint64)
    rc = lib().orc_render(_p(spheres), len(spheres), _p(planes), len(planes), _p(triangles), len(triangles),
                          _p(accel), len(accel), _p(lights), len(lights), _p(camera), int(max_bounces), x0, y0,
                          x1, y1, _p(out.get("rgb")), _p(out.get("rgba8")), _p(out.get("primary_id")),
                          _p(out.get("final_id")), _p(out.get("bounces")), _p(counters), int(n_threads))
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    out["counters"] = {n: int(counters[i]) for i, n in enumerate(COUNTER_NAMES)}
    return out


def import_model(path, scale, position):
    cap = 1 << 16
    raw = np.zeros((cap, 9), np.float32)
    n = lib().orc_import_model(str(path).encode(), scale, *[float(v) for v in position], _p(raw), cap)
    if n < 0:
        raise IOError(path)
    return raw[:n].copy()


def make_triangle(raw9):
    out = np.zeros(12, np.float32)
    raw9 = _f32(raw9)
    lib().orc_make_triangle(_p(raw9), _p(out))
    return out.reshape(4, 3)


def camera_pack(pos, view, right, dist, width, height, screen_w, screen_h):
    out = np.zeros(14, np.float32)
    pos, view, right = _f32(pos), _f32(view), _f32(right)
    lib().orc_camera_pack(_p(pos), _p(view), _p(right), dist, width, height, screen_w, screen_h, _p(out))
    return out


def intersect_box(o, d, box):
    o, d, box = _f32(o), _f32(d), _f32(box)
    hit = np.zeros(len(o), np.int32)
    lib().orc_intersect_box(len(o), _p(o), _p(d), _p(box), _p(hit))
    return hit


def intersect_sphere(o, d, c3r):
    o, d, c3r = _f32(o), _f32(d), _f32(c3r)
    out = np.zeros((len(o), 4), np.float32)
    lib().orc_intersect_sphere(len(o), _p(o), _p(d), _p(c3r), _p(out))
    return out


def intersect_plane(o, d, p, n):
    o, d, p, n = _f32(o), _f32(d), _f32(p), _f32(n)
    t = np.zeros(len(o), np.float32)
    lib().orc_intersect_plane(len(o), _p(o), _p(d), _p(p), _p(n), _p(t))
    return t


def intersect_triangle(o, d, a, b, c, n):
    o, d, a, b, c, n = [_f32(v) for v in (o, d, a, b, c, n)]
    t = np.zeros(len(o), np.float32)
    lib().orc_intersect_triangle(len(o), _p(o), _p(d), _p(a), _p(b), _p(c), _p(n), _p(t))
    return t
