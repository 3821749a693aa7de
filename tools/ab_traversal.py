"""A/B of the two fast kernels on named workloads (development aid): id planes, bounce counts and RGBA8
must be identical; prints ms/frame of each.  usage: ab_traversal.py [workload ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from raytracing_amd import RayTracer, host

names = sys.argv[1:] or ["teapot_1080p_d1", "testscene1_720p_d10", "teapottable_1080p_d8", "soup100k_1080p_d4"]
rt = RayTracer(0)
for name in names:
    sn, pr, w, h, d = B.WORKLOADS[name] if name in B.WORKLOADS else B.parse_workload(name)
    t0 = time.time()
    sc = host.make_scene(sn, params=pr)
    t1 = time.time()
    rt.SendPrimitivesToShader(sc)
    t2 = time.time()
    info = rt.scene_info()
    cam = sc.camera_for(w, h)
    res = {}
    for mode, label in ((1, "wave"), (2, "bvh"), (3, "wf")):
        rt.set_option(rt.OPTION_TRAVERSAL, mode)
        pid = torch.empty((h, w), dtype=torch.int32, device="cuda")
        fid = torch.empty((h, w), dtype=torch.int32, device="cuda")
        bnc = torch.empty((h, w), dtype=torch.int32, device="cuda")
        rgba = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
        rt.render_device(cam, d, rgba8=rgba.data_ptr(), primary_id=pid.data_ptr(), final_id=fid.data_ptr(),
                         bounces=bnc.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        big = name.startswith("soup")
        ms = B.time_single(rt, torch, cam, d, w, h, 3 if big else 30, 2 if big else 5)
        res[label] = (pid, fid, bnc, rgba, ms)
    if os.environ.get("AB_REF"):
        pid = torch.empty((h, w), dtype=torch.int32, device="cuda"); fid = torch.empty_like(pid); bnc = torch.empty_like(pid)
        rgba = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
        rt.render_device(cam, d, rgba8=rgba.data_ptr(), primary_id=pid.data_ptr(), final_id=fid.data_ptr(),
                         bounces=bnc.data_ptr(), reference_order=True, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for label in ("wave", "bvh", "wf"):
            r = res[label]
            dp = (r[0] != pid); df = (r[1] != fid)
            print(f"  {label} vs reference-order: primary_id diffs {int(dp.sum())}, final_id diffs {int(df.sum())}, "
                  f"bounce diffs {int((r[2] != bnc).sum())}, rgba8 diffs {int((r[3] != rgba).any(-1).sum())}")
            if int(dp.sum()):
                idx = dp.nonzero()[:5].cpu().numpy()
                for (yy, xx) in idx:
                    print(f"    px ({xx},{yy}): {label} {int(r[0][yy, xx])} ref {int(pid[yy, xx])}")
    a, b, c = res["wave"], res["bvh"], res["wf"]
    same = [bool(torch.equal(a[k], b[k])) and bool(torch.equal(a[k], c[k])) for k in range(4)]
    st = B.device_stats(rt, torch, cam, d)
    rays = B.total_rays(st)
    print(f"{name:22s} build {t1-t0:5.2f}s upload {t2-t1:5.2f}s nodes {info['n_nodes']} bvh {info['bvh_nodes']} "
          f"levels {info['bvh_levels']} flags {info['flags']} | wave {a[4]:9.3f} ms  bvh {b[4]:9.3f} ms  wf {c[4]:9.3f} ms "
          f"({rays/min(b[4],c[4])/1e3:9.1f} Mrays/s) | identical pid/fid/bounces/rgba8: {same}", flush=True)
