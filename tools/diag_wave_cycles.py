"""Diagnostic (needs build/librt_diag.so, -DRT_DIAG_WAVE_CYCLES): distribution of per-wave lifetimes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from raytracing_amd import RayTracer, host
rt = RayTracer(0)
for name in sys.argv[1:] or ["teapot_1080p_d1"]:
    sn, pr, w, h, d = B.WORKLOADS[name]
    sc = host.make_scene(sn, params=pr); rt.SendPrimitivesToShader(sc)
    cam = sc.camera_for(w, h)
    cyc = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        rt.render_device(cam, d, rgba8=out.data_ptr(), bounces=cyc.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ms = B.time_single(rt, torch, cam, d, w, h, 20, 3)
    c = cyc.cpu().numpy()[: h // 8 * 8, : w // 8 * 8].reshape(h // 8, 8, w // 8, 8)[:, 0, :, 0].astype(np.float64)
    q = np.percentile(c, [50, 90, 99, 99.9, 100])
    print(f"{name}: {ms:.3f} ms/frame; wave lifetimes (cycles @100MHz ticks?) median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {q[4]:.0f}; sum {c.sum():.3e}; waves {c.size}")
