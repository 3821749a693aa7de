"""Python view of the C++ host mirror (csrc/host): scenes and cameras as packed numpy buffers.

A `Scene` holds exactly what RayTracer.SendPrimitivesToShader uploads (RayTracer.cs:957-1046)
plus the 14-float camera array of PrepareRenderOpenGL (RayTracer.cs:943-956).
"""
import ctypes as C

import numpy as np

from . import _native as N


class Scene:
    def __init__(self, spheres, planes, triangles, accel, lights, camera, name=""):
        self.name = name
        self.spheresData = np.ascontiguousarray(spheres, dtype=N.SPHERE_DT)
        self.planesData = np.ascontiguousarray(planes, dtype=N.PLANE_DT)
        self.trianglesData = np.ascontiguousarray(triangles, dtype=N.TRIANGLE_DT)
        self.AccelerationStructureData = np.ascontiguousarray(accel, dtype=np.float32)
        self.lightsData = np.ascontiguousarray(lights, dtype=np.float32)
        self.cameraData = np.ascontiguousarray(camera, dtype=np.float32)

    @property
    def n_lights(self):
        return len(self.lightsData) // 6

    def camera_for(self, width, height):
        """The scene's camera with a different screen size in slots 12, 13 (RayTracer.cs:953)."""
        cam = self.cameraData.copy()
        cam[12], cam[13] = float(width), float(height)
        return cam


def _take(handle, name):
    L = N.host_lib()
    err = L.rth_scene_error(handle).decode()
    if err:
        L.rth_scene_free(handle)
        raise RuntimeError(f"scene {name!r}: {err}")
    dts = [N.SPHERE_DT, N.PLANE_DT, N.TRIANGLE_DT, np.float32, np.float32, np.float32]
    out = []
    for which, dt in enumerate(dts):
        p, cnt = C.c_void_p(), C.c_int64()
        L.rth_scene_get(handle, which, C.byref(p), C.byref(cnt))
        n = cnt.value
        if n == 0:
            out.append(np.zeros(0, dtype=dt))
        else:
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(p.value)
            out.append(np.frombuffer(buf, dtype=dt).copy())
    L.rth_scene_free(handle)
    return Scene(*out, name=name)


def make_scene(name, screen=(1280, 720), params=None):
    """name: TestScene1 | TeapotScene | TeapotTable | Soup (params: n_triangles, n_spheres, n_lights, seed)."""
    L = N.host_lib()
    p = np.asarray(params if params is not None else [], dtype=np.float64)
    h = L.rth_scene_create(name.encode(), float(screen[0]), float(screen[1]), N.ptr(p) if len(p) else None, len(p))
    return _take(h, name)


def accel_from_structs(spheres, triangles):
    """IScene.ActivateAccelerationStructure on packed structs; returns the float array."""
    L = N.host_lib()
    spheres = np.ascontiguousarray(spheres, dtype=N.SPHERE_DT)
    triangles = np.ascontiguousarray(triangles, dtype=N.TRIANGLE_DT)
    h = L.rth_scene_from_structs(N.ptr(spheres), len(spheres), N.ptr(triangles), len(triangles))
    return _take(h, "from_structs").AccelerationStructureData


def camera_pack(position, view, right, distance, width, height, screen_w, screen_h):
    out = np.zeros(14, np.float32)
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (position, view, right)]
    N.host_lib().rth_camera_pack(N.ptr(a[0]), N.ptr(a[1]), N.ptr(a[2]), distance, width, height, screen_w, screen_h,
                                 N.ptr(out))
    return out


def camera_orbit(frame, n_frames, screen_w, screen_h):
    out = np.zeros(14, np.float32)
    N.host_lib().rth_camera_orbit(frame, n_frames, screen_w, screen_h, N.ptr(out))
    return out


def import_model(path, scale, position):
    """OBJImportHelper.ImportModel -> [n, 4, 3] floats: PointA, PointB, PointC, Normal."""
    L = N.host_lib()
    cap = 1 << 16
    out = np.zeros((cap, 12), np.float32)
    n = L.rth_import_model(str(path).encode(), scale, *[float(v) for v in position], N.ptr(out), cap)
    if n < 0:
        raise RuntimeError(L.rth_last_error().decode())
    return out[:n].reshape(n, 4, 3).copy()


def convert_model(text_path, mesh_path):
    L = N.host_lib()
    if L.rth_convert_model(str(text_path).encode(), str(mesh_path).encode()) != 0:
        raise RuntimeError(L.rth_last_error().decode())
