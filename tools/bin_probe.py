"""Does sorting the continuing paths of the staged pipeline by the cell of their new origin pay?  
Renders one workload with RT_WF_BIN_FROM = 0 (off), 1, 2, 3 -- lists of iterations >= that value are sorted -- on one
upload each, K frames between events, and compares every output plane with the unsorted run's bit for bit.
usage: python tools/bin_probe.py [Scene:params:WxH:depth] [K] [settings, e.g. 0,1,2]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import RayTracer, host
import bench

spec = sys.argv[1] if len(sys.argv) > 1 else "Soup:1000000,10000,8:3840x2160:4"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
settings = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,3").split(",")]
name, params, W, H, depth = bench.parse_workload(spec)
sc = host.make_scene(name, (W, H), params)
cam = sc.camera_for(W, H)
st = torch.cuda.current_stream()
base = None
for b in settings:
    os.environ["RT_WF_BIN_FROM"] = str(b)
    with RayTracer(0) as rt:
        rt.set_option(rt.OPTION_TRAVERSAL, 3)  # the staged pipeline
        rt.SendPrimitivesToShader(sc)
        rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        pid = torch.empty((H, W), dtype=torch.int32, device="cuda")
        fid = torch.empty((H, W), dtype=torch.int32, device="cuda")
        bn = torch.empty((H, W), dtype=torch.int32, device="cuda")
        kw = dict(rgb=rgb.data_ptr(), primary_id=pid.data_ptr(), final_id=fid.data_ptr(), bounces=bn.data_ptr(), stream=st.cuda_stream)
        for _ in range(2):
            rt.render_device(cam, depth, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(K):
            rt.render_device(cam, depth, **kw)
        e1.record(st)
        torch.cuda.synchronize()
        out = [t.cpu() for t in (rgb.view(torch.int32), pid, fid, bn)]
        same = "reference run" if base is None else ("identical planes" if all(torch.equal(x, y) for x, y in zip(out, base)) else "PLANES DIFFER")
        if base is None:
            base = out
        print(f"RT_WF_BIN_FROM={b}: {e0.elapsed_time(e1) / K:.3f} ms per frame  ({same})", flush=True)
