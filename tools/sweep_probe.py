"""Orbit sweep, 8 frames per launch: ms per frame (RT_COOP=0/1 picks the kernel).
usage: python tools/sweep_probe.py [frames [scene width height depth]]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import RayTracer, host
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = sys.argv[2] if len(sys.argv) > 2 else "TeapotTable"
w, h, d = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (1920, 1080, 8)
sc = host.make_scene(name)
cams = np.stack([host.camera_orbit(f, F, w, h) for f in range(F)])
st = torch.cuda.current_stream()
with RayTracer(0) as rt:
    rt.SendPrimitivesToShader(sc)
    out = torch.empty((8, h, w, 4), dtype=torch.uint8, device="cuda")
    for _ in range(12):
        rt.render_device(cams[0], d, rgba8=out.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for f0 in range(0, F, 8):
            rt.render_frames_device(cams[f0:f0 + 8], d, h * w, rgba8=out.data_ptr(), stream=st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
    print(f"{name} {w}x{h} d{d} RT_COOP={os.environ.get('RT_COOP', 'default')}: {e0.elapsed_time(e1) / F:.4f} ms per frame, traversal {rt.scene_info()['traversal']}")
