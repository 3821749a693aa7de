"""Wall time of rt_upload_scene on a Soup scene (RT_PREP_TIMING=1 prints the host phases).
usage: python tools/time_upload.py [n_triangles n_spheres n_lights]"""
import sys
import time

import torch  # noqa: F401  (before the library loads the system HIP runtime)

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from raytracing_amd import RayTracer, host  # noqa: E402

params = [int(v) for v in sys.argv[1:4]] or [1000000, 10000, 8]
t = time.time()
scene = host.make_scene("Soup", params)
print(f"scene build {time.time() - t:.2f} s")
with RayTracer(0) as rt:
    for i in range(4):
        t = time.time()
        rt.SendPrimitivesToShader(scene)
        print(f"upload {i}: {(time.time() - t) * 1e3:.1f} ms")
