"""raytracing_amd: MI355X-native per-pixel ray caster behind the Phlipsje/Raytracing host API.

  raytracer.RayTracer   the reference's host object with its GL call sites bound to the C ABI
  host                  scenes/cameras from the C++ mirror of IScene/Camera/RTree/OBJImportHelper
  build                 builds librt_mi355x.so (hipcc, gfx950) and librt_host.so (g++)
"""
from .raytracer import RayTracer, RtError  # noqa: F401
from . import host  # noqa: F401

__all__ = ["RayTracer", "RtError", "host"]
