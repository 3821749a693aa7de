"""Development aid: what one rank of an N-rank bench step costs on this GPU, without the collective:
N frames' local bands (rank 0 of N), batched (ShardedFrames.render -> rt_render_frames_device) against one
launch per frame over four streams, against one whole frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from raytracing_amd import RayTracer, host
from raytracing_amd.parallel import ShardedFrames

name = sys.argv[1] if len(sys.argv) > 1 else "teapot_1080p_d1"
sn, pr, w, h, d = B.WORKLOADS[name]
rt = RayTracer(0)
sc = host.make_scene(sn, params=pr)
rt.SendPrimitivesToShader(sc)
cam = sc.camera_for(w, h)


def timed(fn, K=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for world in (1, 2, 4, 8):
    sf = ShardedFrames(rt, w, h, 0, world, B.BAND_ROWS, world)
    cams = B.orbit_cameras(sc, w, h, world)   # the bench's cameras: different frames, not one frame N times
    out = sf.local if world > 1 else sf.frames
    streams = [torch.cuda.Stream() for _ in range(4)]

    def per_frame():
        main = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(main)
        for f in range(world):
            rt.render_device(cams[f], d, rgba8=out[f].data_ptr(), band_rows=B.BAND_ROWS, rank=0, world=world,
                             stream=streams[f % 4].cuda_stream)
        for s in streams:
            main.wait_stream(s)

    print(f"{name} world {world}: batched {timed(lambda: sf.render(cams, d)):.4f} ms per step, "
          f"one launch per frame over 4 streams {timed(per_frame):.4f} ms", flush=True)
