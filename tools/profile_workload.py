"""Render exactly N frames of one bench workload with the fast path and nothing else (no reference-order
kernel, no CPU work): the program rocprofv3 wraps for the per-workload kernel stats and PMC passes
(tools/gpu_pmc.sh).  Prints the frame count and the live ms/frame.  The library times its candidate kernels over
the first renders of a scene; for a profile of ONE kernel pin the choice: RT_TRAVERSAL=2|3 (RT_OPTION_TRAVERSAL) and
RT_COOP=0|1 (read by the library itself) in the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from raytracing_amd import RayTracer, host

name, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
sn, pr, w, h, d = B.WORKLOADS[name]
rt = RayTracer(0)
if os.environ.get("RT_TRAVERSAL"):
    rt.set_option(rt.OPTION_TRAVERSAL, int(os.environ["RT_TRAVERSAL"]))
sc = host.make_scene(sn, params=pr)
rt.SendPrimitivesToShader(sc)
cam = sc.camera_for(w, h)
out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in range(n):
    if k == n // 2:
        e0.record(s)
    rt.render_device(cam, d, rgba8=out.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
e1.record(s)
torch.cuda.synchronize()
print(f"{name} frames={n} ms_per_frame={e0.elapsed_time(e1) / (n - n // 2):.4f} traversal={rt.scene_info()['traversal']}")
