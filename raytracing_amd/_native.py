"""ctypes bindings of the two native libraries.  No fallbacks: a missing library raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_DEVICE_LIB: development override used to A/B kernel variants in one process tree
DEVICE_LIB = os.environ.get("RT_DEVICE_LIB") or os.path.join(HERE, "librt_mi355x.so")
HOST_LIB = os.path.join(HERE, "librt_host.so")
ASSET_DIR = os.path.join(HERE, "assets")

# PrimitiveStructsForBuffer.cs:8-74 as numpy record dtypes (the bool is one 32-bit word)
SPHERE_DT = np.dtype([("Center", "<f4", 3), ("Radius", "<f4"), ("DiffuseColor", "<f4", 3), ("IsPureSpecular", "<u4"),
                      ("SpecularColor", "<f4", 3), ("Specularity", "<f4")])
PLANE_DT = np.dtype([("Position", "<f4", 3), ("padding0", "<f4"), ("Normal", "<f4", 3), ("padding1", "<f4"),
                     ("DiffuseColor", "<f4", 3), ("IsPureSpecular", "<u4"), ("SpecularColor", "<f4", 3),
                     ("Specularity", "<f4")])
TRIANGLE_DT = np.dtype([("PointA", "<f4", 3), ("padding0", "<f4"), ("PointB", "<f4", 3), ("padding1", "<f4"),
                        ("PointC", "<f4", 3), ("padding2", "<f4"), ("Normal", "<f4", 3), ("padding3", "<f4"),
                        ("DiffuseColor", "<f4", 3), ("IsPureSpecular", "<u4"), ("SpecularColor", "<f4", 3),
                        ("Specularity", "<f4")])
assert SPHERE_DT.itemsize == 48 and PLANE_DT.itemsize == 64 and TRIANGLE_DT.itemsize == 96

STATS_FIELDS = ["primary_rays", "secondary_rays", "shadow_rays", "node_tests", "refs_read", "sphere_tests",
                "plane_tests", "triangle_tests", "material_fetches", "shadow_rays_zero_term"]


class rt_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STATS_FIELDS] + [("reserved", C.c_uint64 * 6)]


class rt_scene_info(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("n_nodes", "n_tri_slots", "n_sphere_slots", "max_depth", "device_bytes", "lds_scene_bytes",
                 "bvh_nodes", "bvh_levels", "flags", "traversal")] + \
               [("tight_nodes", C.c_int64), ("tight_levels", C.c_int64), ("unbounded_triangles", C.c_int64),
                ("reserved", C.c_int64 * 3)]


# every symbol include/rt_mi355x.h declares: name -> (restype, argtypes)
_VP, _I, _I64, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
DEVICE_API = {
    "rt_create": (_I, [_I, C.POINTER(_VP)]),
    "rt_create_multi": (_I, [C.POINTER(_I), _I, C.POINTER(_VP)]),
    "rt_multi_info": (_I, [_VP, C.POINTER(_I), C.POINTER(_I)]),
    "rt_destroy": (None, [_VP]),
    "rt_last_error": (C.c_char_p, [_VP]),
    "rt_set_option": (_I, [_VP, _I, _I]),
    "rt_upload_scene": (_I, [_VP, _VP, _I, _VP, _I, _VP, _I, _VP, _I64, _VP, _I]),
    "rt_render": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "rt_render_device": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rt_render_device_noalloc": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rt_render_frames_device": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _I, _I64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rt_local_rows": (_I, [_I, _I, _I, _I]),
    "rt_padded_rows": (_I, [_I, _I, _I]),
    "rt_unpermute_bands": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "rt_unpermute_frames": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP]),
    "rt_get_scene_info": (_I, [_VP, C.POINTER(rt_scene_info)]),
    "rt_check_accel": (_I, [_VP, _I64, _I, _I, C.POINTER(rt_scene_info), C.c_char_p, C.c_size_t]),
    "rt_check_scene": (_I, [_VP, _I, _VP, _I, _VP, _I64, _I, C.POINTER(rt_scene_info), C.c_char_p, C.c_size_t]),
    "rt_check_gpu_flatten": (_I, [_VP, _VP, _I64, _I, _I, C.c_char_p, C.c_size_t]),
    "rt_check_gpu_build": (_I, [_VP, _VP, _I, _VP, _I, _VP, _I64, _I, C.c_char_p, C.c_size_t]),
    "rt_eval_box": (_I, [_VP, _I, _VP, _VP, _VP, _VP]),
    "rt_eval_sphere": (_I, [_VP, _I, _VP, _VP, _VP, _VP]),
    "rt_eval_plane": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "rt_eval_triangle": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
}
HOST_API = {
    "rth_set_asset_directory": (None, [C.c_char_p]),
    "rth_scene_create": (_VP, [C.c_char_p, _F, _F, _VP, _I]),
    "rth_scene_from_structs": (_VP, [_VP, _I, _VP, _I]),
    "rth_scene_get": (_I, [_VP, _I, C.POINTER(_VP), C.POINTER(_I64)]),
    "rth_scene_error": (C.c_char_p, [_VP]),
    "rth_scene_free": (None, [_VP]),
    "rth_camera_pack": (None, [_VP, _VP, _VP, _F, _F, _F, _F, _F, _VP]),
    "rth_camera_orbit": (None, [_I, _I, _F, _F, _VP]),
    "rth_import_model": (_I, [C.c_char_p, _F, _F, _F, _F, _VP, _I]),
    "rth_convert_model": (_I, [C.c_char_p, C.c_char_p]),
    "rth_last_error": (C.c_char_p, []),
}

_device = None
_host = None


def _load(path, api, what):
    if not os.path.exists(path):
        raise ImportError(f"{what} is not built: {path} is missing.  Run `python -m raytracing_amd.build` "
                          f"(or __graft_entry__.build()).  There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    for name, (res, args) in api.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


def device_lib():
    """librt_mi355x.so (HIP).  Loads without a GPU; every compute call needs one."""
    global _device
    if _device is None:
        _device = _load(DEVICE_LIB, DEVICE_API, "the HIP ray caster")
    return _device


def host_lib():
    global _host
    if _host is None:
        _host = _load(HOST_LIB, HOST_API, "the host scene library")
        _host.rth_set_asset_directory(ASSET_DIR.encode())
    return _host


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
