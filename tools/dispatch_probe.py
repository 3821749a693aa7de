"""How long does a frame of trivial workgroups take?  The teapot scene seen by a camera that looks at the sky: every
tile's rays miss everything.  usage: python tools/dispatch_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import RayTracer, host
sc = host.make_scene("TeapotScene")
st = torch.cuda.current_stream()
with RayTracer(0) as rt:
    rt.SendPrimitivesToShader(sc)
    for (w, h) in ((1280, 720), (1920, 1080), (3840, 2160), (7680, 4320)):
        out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
        for label, view in (("sky", [0.0, 1.0, 0.05]), ("floor", [0.0, -1.0, 0.05])):
            pos = np.array([0.0, 30.0, -40.0]) if label == "floor" else np.array([0.0, 5.0, 0.0])
            if label == "floor":
                pos = np.array([200.0, 30.0, 200.0])   # far from the teapot: plain floor, no mesh in view
            cam = host.camera_pack(pos, np.array(view), np.array([1.0, 0, 0]), 1.0, 1.6, 0.9, w, h)
            for _ in range(15):
                rt.render_device(cam, 1, rgba8=out.data_ptr(), stream=st.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record(st)
            for _ in range(n):
                rt.render_device(cam, 1, rgba8=out.data_ptr(), stream=st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            tiles = ((w + 7) // 8) * ((h + 7) // 8)
            print(f"{w}x{h} {label:5s}: {ms:.4f} ms, {tiles} tiles, {ms * 1e6 / tiles:.2f} ns per tile")
