"""World-size-2 gloo test of the multi-GPU frame assembly (no GPU: each rank's rows come from the
oracle, the exchange and the un-permute are the product's torch.distributed path)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import parallel as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("h,band,world", [(90, 16, 2), (77, 8, 3), (1080, 16, 8), (5, 16, 2), (64, 16, 4)])
def test_row_maps_are_a_partition(h, band, world):
    idx = P.gather_index(h, band, world)
    pr = P.padded_rows(h, band, world)
    assert len(set(idx.tolist())) == h
    seen = np.concatenate([P.global_rows_of(h, band, r, world) for r in range(world)])
    assert sorted(seen.tolist()) == list(range(h))
    for r in range(world):
        assert P.local_rows(h, band, r, world) <= pr
    # round-robin bands balance the load: no rank owns more than one band above the mean
    rows = [P.local_rows(h, band, r, world) for r in range(world)]
    assert max(rows) - min(rows) <= band


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, band, depth, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_render
    from raytracing_amd import host
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = host.make_scene("TestScene1")
        cam = sc.camera_for(w, h)
        pr = P.padded_rows(h, band, world)
        local = torch.zeros((pr, w, 4), dtype=torch.uint8)
        rows = P.global_rows_of(h, band, rank, world)
        # stand-in for rt_render_device on this rank: the oracle renders exactly the rows it owns
        for lr, y in enumerate(rows):
            r = oracle_render(sc, cam, depth, threads=1, region=(0, int(y), w, int(y) + 1), want=("rgba8",))
            local[lr] = torch.from_numpy(r["rgba8"][0])
        gathered = P.all_gather_bands(local, world)
        frame = P.assemble_rows(gathered, h, band, world)
        # the batched form bench.py uses: two frames (the second without bounces) in ONE collective
        local2 = torch.zeros((2, pr, w, 4), dtype=torch.uint8)
        local2[0] = local
        for lr, y in enumerate(rows):
            r = oracle_render(sc, cam, 0, threads=1, region=(0, int(y), w, int(y) + 1), want=("rgba8",))
            local2[1, lr] = torch.from_numpy(r["rgba8"][0])
        g2 = torch.empty((world, 2, pr, w, 4), dtype=torch.uint8)
        dist.all_gather_into_tensor(g2.view(world * 2 * pr, w, 4), local2.view(2 * pr, w, 4))
        frames = P.assemble_frames(g2, h, band, world)
        if rank == 0:
            full = oracle_render(sc, cam, depth, threads=2, want=("rgba8",))["rgba8"]
            full0 = oracle_render(sc, cam, 0, threads=2, want=("rgba8",))["rgba8"]
            q.put(bool(np.array_equal(frame.numpy(), full)) and bool(np.array_equal(frames[0].numpy(), full))
                  and bool(np.array_equal(frames[1].numpy(), full0)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("w,h,band", [(48, 27, 4), (40, 37, 16)])
def test_two_rank_gloo_frame_assembly(w, h, band):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, band, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


# ---- ShardedFrames (bench.py's exchange) with a stand-in renderer: the three assembly modes over gloo --------
class _OracleRenderer:
    """Stand-in for RayTracer on a host without a GPU: same two methods ShardedFrames calls, same buffers (host
    tensors addressed by data_ptr), rows rendered by the oracle, un-permute by the torch index path."""
    device_id = 0

    def __init__(self, sc, w, h):
        self.sc, self.w, self.h = sc, w, h

    @staticmethod
    def _view(ptr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr))

    def render_frames_device(self, cameras, max_bounces, frame_stride, *, rgba8, band_rows, rank, world, stream=0):
        from helpers import oracle_render
        rows = P.global_rows_of(self.h, band_rows, rank, world)
        for f, cam in enumerate(cameras):
            dst = self._view(rgba8 + f * frame_stride * 4, len(rows) * self.w * 4).reshape(len(rows), self.w, 4)
            for lr, y in enumerate(rows):
                dst[lr] = oracle_render(self.sc, cam, max_bounces, threads=1, region=(0, int(y), self.w, int(y) + 1),
                                        want=("rgba8",))["rgba8"][0]

    def unpermute_frames(self, gathered, out, elem_size, width, height, band_rows, world, n_frames, stream=0):
        pr = P.padded_rows(height, band_rows, world)
        g = torch.from_numpy(self._view(gathered, world * n_frames * pr * width * 4).reshape(world, n_frames, pr, width, 4))
        o = self._view(out, n_frames * height * width * 4).reshape(n_frames, height, width, 4)
        o[:] = P.assemble_frames(g, height, band_rows, world).numpy()


def _sharded_worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_render
    from raytracing_amd import host
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h, band, depth = 40, 27, 4, 2
        sc = host.make_scene("TestScene1")
        n_frames = 1 if mode == "root" else world
        cams = [host.camera_orbit(40 * f, 256, w, h) for f in range(n_frames)]
        bufs = [P.ShardedFrames(_OracleRenderer(sc, w, h), w, h, rank, world, band, n_frames, assemble=mode,
                                device=torch.device("cpu")) for _ in range(2)]
        assert not bufs[0].staged
        for step in range(3):            # bench.py's double-buffered step
            b, other = bufs[step % 2], bufs[(step + 1) % 2]
            b.finish()
            b.render(cams, depth)
            other.wait_gather()
            b.gather_async()
        ok = True
        for b in bufs:
            b.finish()
            owned = b.owned_frames()
            expect = {"all": list(range(n_frames)), "owner": [rank], "root": [0] if rank == 0 else []}[mode]
            ok = ok and [f for f, _ in owned] == expect
            for f, frame in owned:
                full = oracle_render(sc, cams[f], depth, threads=2, want=("rgba8",))["rgba8"]
                ok = ok and bool(np.array_equal(frame.numpy(), full))
        q.put((rank, ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["owner", "root", "all"])
def test_two_rank_gloo_sharded_frames(mode):
    """bench.py's step on two ranks: "owner" = one all-to-all, frame f assembled on rank f (weak scaling);
    "root" = one frame gathered on rank 0 (strong scaling); "all" = all-gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True), (1, True)]
