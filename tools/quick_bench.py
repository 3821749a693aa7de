"""Quick single-GPU timing of named workloads (development aid): ms/frame and Mrays/s."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from raytracing_amd import RayTracer, host
names = sys.argv[1:] or ["teapot_1080p_d1", "teapot_4k_d1", "testscene1_720p_d10", "teapottable_1080p_d8", "soup100k_1080p_d4"]
rt = RayTracer(0)
for name in names:
    sn, pr, w, h, d = B.WORKLOADS[name] if name in B.WORKLOADS else B.parse_workload(name)
    sc = host.make_scene(sn, params=pr)
    rt.SendPrimitivesToShader(sc)
    cam = sc.camera_for(w, h)
    st = B.device_stats(rt, torch, cam, d)
    big = name.startswith("soup")
    ms = B.time_single(rt, torch, cam, d, w, h, 3 if big else 30, 2 if big else 5)
    print(f"{name:24s} {ms:9.4f} ms  {B.total_rays(st)/ms/1e3:10.1f} Mrays/s  alg {B.algorithmic_bytes(st,w,h)/ms/1e6:9.1f} GB/s", flush=True)
