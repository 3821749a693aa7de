"""Where do the lanes of the staged pipeline's trace stages spend their loop trips?  (development aid)
Needs a diagnostic build:  hipcc ... -DRT_DIAG_WF -o build/librt_diag_wf.so; run with RT_DEVICE_LIB pointing to it.
Per stage: node trips / leaf trips, and in each kind the mean number of lanes working, parked (at a leaf during a
node trip / at a node during a leaf trip) and without a ray."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from raytracing_amd import RayTracer, host, _native as N

name = sys.argv[1] if len(sys.argv) > 1 else "soup1m_4k_d4"
sn, pr, w, h, d = B.WORKLOADS[name]
rt = RayTracer(0)
sc = host.make_scene(sn, params=pr)
rt.SendPrimitivesToShader(sc)
cam = sc.camera_for(w, h)
out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
lib = C.CDLL(N.DEVICE_LIB)
buf = (C.c_ulonglong * 16)()
for k in range(2):
    rt.render_device(cam, d, rgba8=out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert lib.rt_debug_wf_diag(buf) == 0
v = list(buf)
for base, stage in ((0, "closest"), (8, "any-hit")):
    nt, nw, ni, nl, lt, lw, li, spin = v[base:base + 8]
    print(f"{stage}: node trips {nt:.3e} (x{B.__dict__.get('X', 4)} steps) lanes working {nw / max(nt, 1):.1f} idle {ni / max(nt, 1):.1f} parked at leaf {nl / max(nt, 1):.1f} | "
          f"leaf trips {lt:.3e} lanes working {lw / max(lt, 1):.1f} idle {li / max(lt, 1):.1f} parked at node {64 - (lw + li) / max(lt, 1):.1f} | empty spins {spin:.3e}")
