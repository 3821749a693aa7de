"""Per-ray-class cost of the per-lane megakernel on the teapot scene at 4K: three cameras --
every ray misses everything ("sky"), every pixel is plain mirror floor far from the mesh ("floor": primary + 6 shadow
rays + bounce per pixel, none near the tree), and the bench's own teapot frame ("teapot") -- each rendered K times in a
row by the plain megakernel (cooperative shadow rays and the measured kernel choice switched off, so every dispatch is
k_render_bvh<false, false, false>).  Prints ms per frame and the dispatch order; under
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM
the i-th k_render_bvh row of counter_collection.csv belongs to the i-th line of the printed order
(tools/class_probe_join.py folds the two).
usage: python tools/class_probe.py [K] [WIDTH HEIGHT]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import RayTracer, host

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
sc = host.make_scene("TeapotScene")
st = torch.cuda.current_stream()
cams = {
    "sky": host.camera_pack(np.array([0.0, 5.0, 0.0]), np.array([0.0, 1.0, 0.05]), np.array([1.0, 0, 0]), 1.0, 1.6, 0.9, W, H),
    "floor": host.camera_pack(np.array([200.0, 30.0, 200.0]), np.array([0.0, -1.0, 0.05]), np.array([1.0, 0, 0]), 1.0, 1.6, 0.9, W, H),
    "teapot": sc.camera_for(W, H),
}
with RayTracer(0) as rt:
    rt.set_option(rt.OPTION_COOP_SHADOWS, 0)
    rt.set_option(rt.OPTION_TRAVERSAL, 2)
    rt.SendPrimitivesToShader(sc)
    out = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    order = []
    for label, cam in cams.items():
        for _ in range(2):
            rt.render_device(cam, 1, rgba8=out.data_ptr(), stream=st.cuda_stream)
            order.append(label + ":warm")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(K):
            rt.render_device(cam, 1, rgba8=out.data_ptr(), stream=st.cuda_stream)
            order.append(label)
        e1.record(st)
        torch.cuda.synchronize()
        print(f"class {label:7s} {W}x{H}: {e0.elapsed_time(e1) / K:.4f} ms per frame, {((W + 7) // 8) * ((H + 7) // 8)} waves")
print("order " + " ".join(order))
