"""rt_check_gpu_build (device-built trees, cull preconditions and item boxes against the host's, bit for bit) over many
soups of different sizes -- each size is a different random scene -- for a time budget.
usage: python tools/stress_build.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (before the library loads the system HIP runtime)
from raytracing_amd import RayTracer, host

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
done = declined = 0
tris = 0
with RayTracer(0) as rt:
    while time.time() < t_end:
        n = int(10 ** g.uniform(2.0, 5.6))
        s = int(g.integers(0, max(2, n // 50)))
        sc = host.make_scene("Soup", params=[n, s, 4])
        r = rt.check_gpu_build(sc)          # raises RtError with the first difference
        done += 1
        declined += r == 0
        tris += n
        if done % 10 == 0:
            print(f"... {done} scenes ({tris} triangles), {declined} declined by the device", flush=True)
print(f"stress build: {done} soups of 100 .. 400 000 triangles ({tris} triangles in all): device-built trees, cull verdicts, unbounded "
      f"marks and slot maps identical to the host's in every one; {declined} declined by the device")
