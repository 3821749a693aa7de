"""Randomised parity sweep (development aid, run on the GPU box): every fast traversal against the
reference-order kernel on random cameras, including cameras inside the geometry, grazing views along the
floor and axis-aligned views.  Reports the number of differing pixels per plane (must be 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import RayTracer, host

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rt = RayTracer(0)
rt.set_option(rt.OPTION_WAVEFRONT_MIN_NODES, 0)   # every scene gets the tight tree: mode 3 walks it
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
scenes = [("TestScene1", None, 10, 12.0), ("TeapotTable", None, 8, 12.0), ("Soup", [20000, 200, 8], 4, 60.0),
          ("Soup", [300000, 3000, 8], 3, 60.0)]
W, H = 256, 144
bad = total = 0
t_end = time.time() + budget
st = torch.cuda.current_stream().cuda_stream
while time.time() < t_end:
    for name, params, depth, ext in scenes:
        sc = host.make_scene(name, params=params)
        rt.SendPrimitivesToShader(sc)
        for _ in range(6):
            kind = g.integers(0, 4)
            pos = g.uniform([-ext, 0.05, -ext], [ext, ext * 0.6, ext])
            if kind == 0:    # looking at the middle of the scene
                view = g.uniform([-2, 0, -2], [2, 4, 6]) - pos
            elif kind == 1:  # grazing along the floor
                pos[1] = g.uniform(0.01, 0.3); view = np.array([g.normal(), g.uniform(-0.02, 0.02), g.normal()])
            elif kind == 2:  # axis aligned, integer position
                pos = np.round(pos); view = np.eye(3)[g.integers(0, 3)] * g.choice([-1.0, 1.0])
            else:            # anywhere
                view = g.normal(size=3)
            right = np.cross(view, [0, -1, 0])
            if np.linalg.norm(right) < 1e-6:
                right = np.array([1.0, 0, 0])
            cam = host.camera_pack(pos, view, right, 1.0, 1.6, 0.9, W, H)
            planes = {}
            for mode in (None, 1, 2, 3, 5):
                pid = torch.empty((H, W), dtype=torch.int32, device="cuda"); fid = torch.empty_like(pid)
                rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
                if mode is not None:   # 5 = the BVH megakernel with cooperative shadow rays
                    rt.set_option(rt.OPTION_TRAVERSAL, 2 if mode == 5 else mode)
                    rt.set_option(rt.OPTION_COOP_SHADOWS, 1 if mode == 5 else 0)
                rt.render_device(cam, depth, rgb=rgb.data_ptr(), primary_id=pid.data_ptr(), final_id=fid.data_ptr(),
                                 reference_order=mode is None, stream=st)
                torch.cuda.synchronize()
                planes[mode] = (pid, fid, rgb.view(torch.int32))
            total += 1
            for mode in (1, 2, 3, 5):
                d = [int((planes[mode][k] != planes[None][k]).sum()) for k in range(3)]
                if any(d):
                    bad += 1
                    print(f"MISMATCH {name} {params} mode {mode} kind {kind} pos {pos.tolist()} view {view.tolist()}: "
                          f"primary {d[0]} final {d[1]} rgb words {d[2]}", flush=True)
    print(f"... {total} cameras, {bad} mismatching (camera, traversal) pairs", flush=True)
print(f"stress parity: {total} cameras x 4 traversals (wave, bvh, wavefront + tight tree, bvh + cooperative shadow rays) "
      f"against the reference-order kernel, every float of rgb and both id planes: {bad} mismatches")
