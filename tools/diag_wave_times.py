"""Development aid (needs a library built with -DRT_DIAG_WAVE_TIMES, RT_DEVICE_LIB=...): when do the waves of
one megakernel launch start and end?  Prints the launch's occupancy profile on the 100 MHz real-time clock."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from raytracing_amd import RayTracer, host
name = sys.argv[1] if len(sys.argv) > 1 else "teapot_1080p_d1"
sn, pr, w, h, d = B.WORKLOADS[name]
rt = RayTracer(0)
sc = host.make_scene(sn, params=pr); rt.SendPrimitivesToShader(sc)
cam = sc.camera_for(w, h)
st = torch.cuda.current_stream().cuda_stream
t0 = torch.zeros((h, w), dtype=torch.int32, device="cuda"); t1 = torch.zeros_like(t0)
for _ in range(3):
    rt.render_device(cam, d, primary_id=t0.data_ptr(), final_id=t1.data_ptr(), stream=st)
torch.cuda.synchronize()
a = t0.cpu().numpy().astype(np.int64) & 0xFFFFFFFF; b = t1.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
# one value per 8x8 wave tile
a = a[::8, ::8].ravel(); b = b[::8, ::8].ravel()
base = a.min(); a = (a - base) * 0.01; b = (b - base) * 0.01      # microseconds
life = b - a
print(f"{name}: {len(a)} waves, launch spans {b.max():.1f} us; wave lifetime mean {life.mean():.1f} us, "
      f"p50 {np.percentile(life,50):.1f}, p90 {np.percentile(life,90):.1f}, p99 {np.percentile(life,99):.1f}, max {life.max():.1f}")
print(f"sum of lifetimes / (4096 slots) = {life.sum()/4096:.1f} us")
for q in (10, 25, 50, 75, 90, 95, 99, 100):
    print(f"  {q:3d} % of the waves have started by {np.percentile(a,q):7.1f} us, ended by {np.percentile(b,q):7.1f} us")
edges = np.arange(0, b.max() + 20, 20)
for t in edges:
    print(f"  t={t:6.0f} us: {int(((a <= t) & (b > t)).sum()):5d} waves resident")
late = np.argsort(b)[-12:]
tw = (w + 7) // 8
print("last waves to end: (tile x, tile y from the bottom) start / end / lifetime")
for i in late:
    print(f"   ({i % tw:3d},{i // tw:3d})  {a[i]:7.1f} {b[i]:7.1f} {life[i]:7.1f}")
longest = np.argsort(life)[-12:]
print("longest waves:")
for i in longest:
    print(f"   ({i % tw:3d},{i // tw:3d})  {a[i]:7.1f} {b[i]:7.1f} {life[i]:7.1f}")
