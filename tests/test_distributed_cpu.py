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
