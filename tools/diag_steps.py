"""Diagnostic (needs build/librt_diag.so built with -DRT_DIAG_STEPS, loaded through RT_DEVICE_LIB): work per ray
and SIMD efficiency of bvh_walk.  usage: RT_DEVICE_LIB=build/librt_diag.so python tools/diag_steps.py [workload ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from raytracing_amd import RayTracer, host
rt = RayTracer(0)
for name in sys.argv[1:] or ["soup100k_1080p_d4"]:
    sn, pr, w, h, d = B.WORKLOADS[name]
    sc = host.make_scene(sn, params=pr); rt.SendPrimitivesToShader(sc)
    cam = sc.camera_for(w, h)
    rt.set_option(rt.OPTION_TRAVERSAL, 2)
    z = lambda: torch.zeros((h, w), dtype=torch.int32, device="cuda")
    nodes, prims, wnode = z(), z(), z()
    rgb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    rt.render_device(cam, d, rgb=rgb.data_ptr(), primary_id=wnode.data_ptr(), final_id=prims.data_ptr(), bounces=nodes.data_ptr(),
                     stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = B.device_stats(rt, torch, cam, d)  # reference-order kernel: unaffected by the diag build
    rays = B.total_rays(st)
    n, p, wn, wl = float(nodes.sum()), float(prims.sum()), float(wnode.sum()), float(rgb[..., 0].sum())
    print(f"{name}: rays {rays}  node steps/ray {n/rays:.1f}  prim tests/ray {p/rays:.1f}  "
          f"(reference rule: nodes/ray {st['node_tests']/rays:.1f} tris/ray {st['triangle_tests']/rays:.1f})  "
          f"wave node iterations {wn:.3e} -> lane utilisation {n/(wn*64):.2f}; wave leaf iterations {wl:.3e} "
          f"-> leaves/lane-slot {p/(wl*64):.2f} prims per lane-slot")
