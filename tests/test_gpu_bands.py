"""GPU: band-sharded rendering and the un-permute kernel reproduce the single-GPU frame."""
import numpy as np
import pytest
import torch

from raytracing_amd import parallel as P

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1, 3, 5], ids=["auto", "wave", "wavefront", "bvh+coop-shadows"])
def traversal(request, rt):
    """5 = the BVH megakernel with cooperative shadow rays forced (by default it is used where it measures faster)."""
    rt.set_option(rt.OPTION_TRAVERSAL, 2 if request.param == 5 else request.param)
    rt.set_option(rt.OPTION_COOP_SHADOWS, 1 if request.param == 5 else 2)
    yield request.param
    rt.set_option(rt.OPTION_TRAVERSAL, rt.TRAVERSAL_AUTO)
    rt.set_option(rt.OPTION_COOP_SHADOWS, 2)


@pytest.mark.parametrize("w,h,band,world", [(160, 90, 16, 2), (123, 77, 8, 3), (320, 180, 16, 8), (64, 20, 32, 2)])
def test_bands_assemble_to_the_full_frame(w, h, band, world, rt, scene_of, traversal):
    sc = scene_of("TestScene1")
    cam = sc.camera_for(w, h)
    rt.SendPrimitivesToShader(sc)
    rt.PrepareRenderOpenGL(cam)
    full = rt.RenderGL(10, want=("rgba8", "final_id"))
    dev = torch.device("cuda", 0)
    shards = [P.ShardedFrame(rt, w, h, r, world, band, device=dev) for r in range(world)]
    fid = [torch.full((s.pr, w), -7, dtype=torch.int32, device=dev) for s in shards]
    for r, s in enumerate(shards):  # every "rank" renders on the one GPU, in turn
        assert rt.local_rows(h, band, r, world) == P.local_rows(h, band, r, world)
        assert rt.padded_rows(h, band, world) == s.pr
        rt.render_device(cam, 10, rgba8=s.local.data_ptr(), final_id=fid[r].data_ptr(), band_rows=band, rank=r,
                         world=world, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    gathered = torch.stack([s.local for s in shards])
    root = shards[0]
    root.gathered.copy_(gathered)
    frame = root.assemble()
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), full["rgba8"])
    assert np.array_equal(P.assemble_rows(gathered, h, band, world).cpu().numpy(), full["rgba8"])
    ids = P.assemble_rows(torch.stack(fid), h, band, world).cpu().numpy()
    assert np.array_equal(ids, full["final_id"])
    # rows beyond a rank's share are never written
    for r in range(world):
        assert (fid[r][P.local_rows(h, band, r, world):] == -7).all()


def test_batched_frames_unpermute(rt, scene_of):
    """bench.py's exchange: several frames' bands gathered together, one un-permute launch."""
    sc = scene_of("TeapotScene")
    w, h, band, world, F = 96, 54, 8, 4, 3
    cams = [sc.camera_for(w, h)] + [None] * (F - 1)
    from raytracing_amd import host
    for f in range(1, F):
        cams[f] = host.camera_orbit(f * 20, 256, w, h)
    rt.SendPrimitivesToShader(sc)
    dev = torch.device("cuda", 0)
    pr = P.padded_rows(h, band, world)
    gathered = torch.zeros((world, F, pr, w, 4), dtype=torch.uint8, device=dev)
    full = torch.zeros((F, h, w, 4), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for f, cam in enumerate(cams):
        rt.render_device(cam, 1, rgba8=full[f].data_ptr(), stream=st)
        for r in range(world):
            rt.render_device(cam, 1, rgba8=gathered[r, f].data_ptr(), band_rows=band, rank=r, world=world, stream=st)
    out = torch.empty_like(full)
    rt.unpermute_frames(gathered.data_ptr(), out.data_ptr(), 4, w, h, band, world, F, stream=st)
    torch.cuda.synchronize()
    assert torch.equal(out, full)
    assert torch.equal(P.assemble_frames(gathered, h, band, world), full)
    assert len(torch.unique(full[1].view(-1, 4), dim=0)) > 50   # a real picture, not a blank frame


def test_unpermute_float3(rt):
    w, h, band, world = 50, 37, 8, 3
    pr = P.padded_rows(h, band, world)
    g = torch.randn((world, pr, w, 3), device="cuda")
    out = torch.empty((h, w, 3), device="cuda")
    rt.unpermute_bands(g.data_ptr(), out.data_ptr(), 12, w, h, band, world)
    torch.cuda.synchronize()
    assert torch.equal(out, P.assemble_rows(g, h, band, world))


def test_nccl_world_size_one(rt, scene_of):
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        sc = scene_of("TestScene1")
        cam = sc.camera_for(96, 54)
        rt.SendPrimitivesToShader(sc)
        sf = P.ShardedFrame(rt, 96, 54, 0, 1, 16)
        sf.render(cam, 10)
        sf.gather()
        frame = sf.assemble()
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        rt.PrepareRenderOpenGL(cam)
        assert np.array_equal(frame.cpu().numpy(), rt.RenderGL(10, want=("rgba8",))["rgba8"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,rank", [(1, 0), (3, 1), (8, 7)])
def test_batched_launch_equals_one_launch_per_frame(world, rank, rt, scene_of, traversal):
    """rt_render_frames_device: 11 cameras (more than the 8 of one launch) into strided planes must give,
    bit for bit, what 11 rt_render_device calls give -- for the megakernels (one launch per 8 frames) and
    for the staged pipeline (one launch sequence per frame)."""
    import torch
    from raytracing_amd import host
    sc = scene_of("TestScene1")
    rt.SendPrimitivesToShader(sc)
    w, h, band, depth, F = 200, 120, 8, 3, 11
    cams = np.stack([host.camera_orbit(7 * f, 96, w, h) for f in range(F)])
    rows = rt.local_rows(h, band, rank, world)
    stride = rows * w + 64                     # padded plane stride
    st = torch.cuda.current_stream().cuda_stream
    planes = {k: torch.full((F, stride) + shp, 77, dtype=dt, device="cuda")
              for k, shp, dt in (("rgba8", (4,), torch.uint8), ("rgb", (3,), torch.float32),
                                 ("primary_id", (), torch.int32), ("final_id", (), torch.int32),
                                 ("bounces", (), torch.int32))}
    rt.render_frames_device(cams, depth, stride, band_rows=band, rank=rank, world=world, stream=st,
                            **{k: v.data_ptr() for k, v in planes.items()})
    torch.cuda.synchronize()
    for f in range(F):
        one = {k: torch.empty_like(v[0]) for k, v in planes.items()}
        rt.render_device(cams[f], depth, band_rows=band, rank=rank, world=world, stream=st,
                         **{k: v.data_ptr() for k, v in one.items()})
        torch.cuda.synchronize()
        for k in planes:
            a, b = planes[k][f, :rows * w], one[k][:rows * w]
            assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), (k, f)
            assert bool((planes[k][f, rows * w:] == 77).all()), "padding touched"
    assert not torch.equal(planes["rgba8"][0], planes["rgba8"][5])
