#!/usr/bin/env python3
"""bench.py -- Mrays/s of the per-pixel ray-casting hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): TeapotScene -- assets/teapot via OBJImportHelper + RTree over
the mirror floor with TestScene1's six lights -- 1920x1080, shadows + 1 reflection bounce.
A step renders world-size frames of a camera orbit; every frame is cut into 8-row bands dealt
round-robin to the ranks, rendered into HBM, and the whole batch is assembled on every rank with ONE
RCCL all-gather per step (double buffered, so the exchange of step s-1 overlaps the render of step
s).  Per-GPU work per step is therefore one frame's worth at any N ("weak").
Rays are counted by the reference's rule (primary + executed mirror bounces + shadow rays) with
the reference-order kernel outside the timed region; the count is identical to the oracle's.

One JSON line is printed by rank 0 (contract in the task description) with two extra objects:
  roofline      algorithmic bytes of the reference algorithm per launch / measured kernel time
  cpu_baseline  the CPU oracle ("port") timed single-threaded on a bounded sample of the same frame
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KERNEL_OF = {1: "k_render<false> (wave-cooperative walk)", 2: "k_render_bvh (per-lane BVH, LDS stacks)",
             3: "k_wf_closest + k_wf_any + k_wf_shade per bounce (wavefront pipeline, whole frame)"}
FP32_VECTOR_PEAK_TF = 157.3  # MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BAND_ROWS = 8  # one 8x8 tile row per band: finest interleave of sky / mesh / floor rows over the ranks

WORKLOADS = {
    # name: (scene, params, width, height, max_bounces)
    "teapot_1080p_d1": ("TeapotScene", None, 1920, 1080, 1),
    "teapot_4k_d1": ("TeapotScene", None, 3840, 2160, 1),
    "testscene1_720p_d10": ("TestScene1", None, 1280, 720, 10),
    "teapottable_1080p_d8": ("TeapotTable", None, 1920, 1080, 8),
    "soup1m_4k_d4": ("Soup", [1_000_000, 10_000, 8], 3840, 2160, 4),
    "soup1m_8k_d4": ("Soup", [1_000_000, 10_000, 8], 7680, 4320, 4),
    "soup100k_1080p_d4": ("Soup", [100_000, 1_000, 8], 1920, 1080, 4),
}


def parse_workload(spec):
    """Ad-hoc workload for the development tools: 'Scene:p1,p2,..:WxH:depth', e.g. 'Soup:20000,200,8:1920x1080:4'."""
    scene, params, size, depth = spec.split(":")
    w, h = size.split("x")
    return scene, ([int(v) for v in params.split(",")] if params else None), int(w), int(h), int(depth)


def algorithmic_bytes(stats, width, height):
    """bytes the reference algorithm needs per frame (every candidate tested)."""
    return (32 * stats["node_tests"] + 4 * stats["refs_read"] + 16 * stats["sphere_tests"] +
            64 * stats["triangle_tests"] + 32 * stats["plane_tests"] + 32 * stats["material_fetches"] +
            4 * width * height)


def algorithmic_flops(stats):
    """flops of the reference algorithm per frame (27 per box test, 30 per sphere, 75 per
    triangle, 8 per plane test) -- the VALU-side figure for cache-resident scenes."""
    return (27 * stats["node_tests"] + 30 * stats["sphere_tests"] + 75 * stats["triangle_tests"] +
            8 * stats["plane_tests"])


def total_rays(stats):
    return stats["primary_rays"] + stats["secondary_rays"] + stats["shadow_rays"]


def orbit_cameras(sc, width, height, n):
    """n cameras on a slow orbit around the scene camera's own view (frame 0 = the scene camera)."""
    from raytracing_amd import host
    cams = [sc.camera_for(width, height)]
    pos = sc.cameraData[0:3].astype(np.float64)
    centre = (sc.cameraData[3:6] + (sc.cameraData[6:9] - sc.cameraData[3:6]) / 2 +
              (sc.cameraData[9:12] - sc.cameraData[3:6]) / 2).astype(np.float64)
    view = centre - pos
    target = pos + view / np.linalg.norm(view) * 8.0
    for k in range(1, n):
        ang = np.pi / 180.0 * k   # one degree per frame: a slow camera move, so that every frame of a step costs
                                  # about what frame 0 (the scene's own camera, the N = 1 workload) costs
        rel = pos - target
        c, s = np.cos(ang), np.sin(ang)
        p = target + np.array([c * rel[0] + s * rel[2], rel[1], -s * rel[0] + c * rel[2]])
        v = target - p
        right = np.cross(v, [0, -1, 0])
        cams.append(host.camera_pack(p, v, right, 1.0, 1.6, 0.9, width, height))
    return cams


def device_stats(rt, torch, cam, depth):
    """Reference-rule counters of one frame from the reference-order kernel (not timed)."""
    from raytracing_amd import _native as N
    st = torch.zeros(16, dtype=torch.int64, device="cuda")
    rt.render_device(cam, depth, stats=st.data_ptr(), reference_order=True,
                     stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    v = st.cpu().numpy()
    return {k: int(v[i]) for i, k in enumerate(N.STATS_FIELDS)}


def time_single(rt, torch, cam, depth, width, height, steps, warmup):
    """ms per frame of the fast kernel, device-resident RGBA8 output, HIP events on the launch stream."""
    out = torch.empty((height, width, 4), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()
    for _ in range(warmup):
        rt.render_device(cam, depth, rgba8=out.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(steps):
        rt.render_device(cam, depth, rgba8=out.data_ptr(), stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def cpu_baseline(sc, cam, depth, width, height, budget_s=12.0):
    """The oracle (a C++ port of the reference algorithm; the reference itself cannot run here),
    single-threaded, on evenly spaced full rows of the same frame, sized to ~budget_s of CPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    def rows(ys, threads):
        rays, t0 = 0, time.perf_counter()
        for y in ys:
            r = O.render(sc.spheresData.view(O.SPHERE_DT), sc.planesData.view(O.PLANE_DT),
                         sc.trianglesData.view(O.TRI_DT), sc.AccelerationStructureData, sc.lightsData, cam, depth,
                         region=(0, int(y), width, int(y) + 1), n_threads=threads, want=())
            c = r["counters"]
            rays += c["primary"] + c["secondary"] + c["shadow"]
        return rays, time.perf_counter() - t0

    # size the sample to ~budget_s of single-thread work: evenly spaced rows if the frame is slower
    # than the budget, otherwise the whole frame repeated
    probe = np.linspace(0, height - 1, 8).astype(int)
    rays, dt = rows(probe, 1)
    per_row = dt / len(probe)
    if per_row * height > budget_s:
        n_rows = int(max(8, budget_s / max(per_row, 1e-6)))
        ys = np.unique(np.linspace(0, height - 1, n_rows).astype(int))
        what = f"{len(ys)} evenly spaced full rows of the {width}x{height} frame"
    else:
        reps = int(min(64, max(1, round(budget_s / max(per_row * height, 1e-6)))))
        ys = np.tile(np.arange(height), reps)
        what = f"the whole {width}x{height} frame, {reps} passes"
    rays, dt = rows(ys, 1)
    one = rays / dt / 1e6
    # the same sample on all host cores (one row per call, rows spread over an OpenMP team is not
    # possible with 1-row regions, so the multi-core figure renders a band per call)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16) if os.environ.get("OMP_NUM_THREADS") is None else min(cores, int(os.environ["OMP_NUM_THREADS"]))
    def frame_all():
        r = O.render(sc.spheresData.view(O.SPHERE_DT), sc.planesData.view(O.PLANE_DT), sc.trianglesData.view(O.TRI_DT),
                     sc.AccelerationStructureData, sc.lightsData, cam, depth, region=(0, 0, width, height),
                     n_threads=cores, want=())
        c = r["counters"]
        return c["primary"] + c["secondary"] + c["shadow"]

    all_what = f"the whole {width}x{height} frame"
    if per_row * height / cores > 4.0:      # a big scene: a centred band of rows instead of whole frames
        band = min(height, max(cores * 4, int(4.0 * cores / per_row)))
        y0 = (height - band) // 2
        all_what = f"rows {y0}..{y0 + band - 1}"

        def frame_all():  # noqa: F811
            r = O.render(sc.spheresData.view(O.SPHERE_DT), sc.planesData.view(O.PLANE_DT),
                         sc.trianglesData.view(O.TRI_DT), sc.AccelerationStructureData, sc.lightsData, cam, depth,
                         region=(0, y0, width, y0 + band), n_threads=cores, want=())
            c = r["counters"]
            return c["primary"] + c["secondary"] + c["shadow"]
    frame_all()                             # the thread team starts outside the timed region
    rays_all, passes, t0 = 0, 0, time.perf_counter()
    while passes == 0 or time.perf_counter() - t0 < 3.0:
        rays_all += frame_all()
        passes += 1
    dt_all = time.perf_counter() - t0
    allc = rays_all / dt_all / 1e6
    return {"value": round(one, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"{what} ({rays} rays, {dt:.1f} s, C++ restatement of the reference algorithm, "
                      f"1 thread, row by row)",
            "all_cores": {"value": round(allc, 3), "cores": cores,
                          "sample": f"{all_what}, {passes} passes, OpenMP row loop, {dt_all:.1f} s"}}


def dbg(*a):
    if os.environ.get("BENCH_DEBUG"):
        print(f"[rank {os.environ.get('RANK', '0')}]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="teapot_1080p_d1", choices=sorted(WORKLOADS))
    ap.add_argument("--no-extras", action="store_true", help="skip the extra single-GPU workloads")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--verify", action="store_true", help="check the assembled frames against whole-frame renders")
    ap.add_argument("--soup1m", action="store_true", help="include the 1M-triangle soup in the extras")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracing_amd import RayTracer, host
    from raytracing_amd import parallel as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # BENCH_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the N > 1 code path on a one-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    dbg("process group up")
    scene_name, params, W, H, depth = WORKLOADS[args.workload]
    sc = host.make_scene(scene_name, params=params)
    rt = RayTracer(local_rank)
    rt.SendPrimitivesToShader(sc)
    cams = orbit_cameras(sc, W, H, world)

    # rays per step: every frame of the batch, counted by the reference-order kernel (untimed)
    frame_stats = [device_stats(rt, torch, cam, depth) for cam in cams]
    rays_per_step = sum(total_rays(s) for s in frame_stats)
    dbg("stats done", rays_per_step)

    # ---- timed region: K steps, each = `world` band-sharded frames + ONE all-gather for the batch ----
    stream = torch.cuda.current_stream()
    bufs = [P.ShardedFrames(rt, W, H, rank, world, BAND_ROWS, len(cams)) for _ in range(2)]  # double buffered

    step_no = [0]

    def step():
        # step s renders while the all-gather of step s-1 is in flight on RCCL's stream
        b, other = bufs[step_no[0] % 2], bufs[(step_no[0] + 1) % 2]
        b.finish()               # this buffer's previous exchange: scanline order, on every rank
        b.render(cams, depth)
        other.wait_gather()      # at most one collective in flight (it has overlapped this step's render)
        b.gather_async()
        step_no[0] += 1

    def drain():
        for b in bufs:
            b.finish()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
        dbg("warmup step")
    drain()
    dbg("drained")
    fence()
    dbg("fenced")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        step()
    drain()
    e1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    dbg("timed region done", elapsed)
    if args.verify:  # the assembled frames of the last step equal whole-frame renders on this rank
        last = bufs[(step_no[0] - 1) % 2].frames
        for f, cam in enumerate(cams):
            ref = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
            rt.render_device(cam, depth, rgba8=ref.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize()
            if not torch.equal(ref, last[f]):
                raise SystemExit(f"rank {rank}: assembled frame {f} differs from the single-GPU render")
        dbg("verified", len(cams), "assembled frames")
    kernel_ms = e0.elapsed_time(e1) / (args.steps * len(cams))  # per render launch, on the launch stream
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_step / (elapsed / args.steps) / 1e6
        st0 = frame_stats[0]
        # one launch renders 1/world of a frame; algorithmic bytes scale the same way
        alg_bytes = sum(algorithmic_bytes(s, W, H) for s in frame_stats) / len(frame_stats) / world
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        alg_flops = sum(algorithmic_flops(s) for s in frame_stats) / len(frame_stats) / world
        valu_tf = alg_flops / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(args.workload, {}).get("hbm_bytes_per_launch")
        result = {
            "metric": "Mrays/s (primary+shadow+secondary)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (teapot mesh fixture of the reference asset, builder-defined scene, orbit cameras)",
            "config": {"workload": args.workload, "scene": scene_name, "width": W, "height": H,
                       "max_bounces": depth, "lights": sc.n_lights, "triangles": len(sc.trianglesData),
                       "frames_per_step": len(cams), "rays_per_step": rays_per_step,
                       "sharding": f"{BAND_ROWS}-row bands round-robin over {world} rank(s), "
                                   f"one RCCL all-gather per step" if world > 1 else "single GPU, whole frame"},
            "ray_breakdown_frame0": {k: st0[k] for k in ("primary_rays", "secondary_rays", "shadow_rays")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": KERNEL_OF.get(rt.scene_info()["traversal"], "?"), "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "note": "algorithmic bytes of the REFERENCE algorithm (SURVEY 8d: every candidate "
                                 "tested, no distance cull); the kernel culls and orders, so it is credited with "
                                 "more bytes than it moves (frac can exceed 1) -- `traffic` is what HBM saw.  "
                                 "This scene is L2-resident: the binding limit is VALU issue, see `valu`",
                         "valu": {"algorithmic_flops_per_launch": int(alg_flops),
                                  "achieved_tflops": round(valu_tf, 2), "peak_tflops": FP32_VECTOR_PEAK_TF,
                                  "frac": round(valu_tf / FP32_VECTOR_PEAK_TF, 4),
                                  "note": "peak counts FMA as 2 flops; the canonical arithmetic forbids "
                                          "contraction, so half of it is reachable"}},
        }

    # ---- N = 1 only: extra workloads and the CPU baseline (outside the contract's timed region) ----
    if world == 1 and rank == 0:
        if not args.no_extras:
            extras = {}
            names = ["teapot_4k_d1", "testscene1_720p_d10", "teapottable_1080p_d8", "soup100k_1080p_d4"]
            if args.soup1m:
                names.append("soup1m_4k_d4")
            for name in names:
                if name == args.workload:
                    continue
                sn, pr, w, h, d = WORKLOADS[name]
                t_build = time.perf_counter()
                s2 = host.make_scene(sn, params=pr)
                t_build = time.perf_counter() - t_build
                rt.SendPrimitivesToShader(s2)
                cam = s2.camera_for(w, h)
                st = device_stats(rt, torch, cam, d)
                big = name.startswith("soup")
                ms = time_single(rt, torch, cam, d, w, h, 3 if big else 20, 2 if big else 3)  # >= 2 warm-up frames: AUTO times both kernels
                ab = algorithmic_bytes(st, w, h)
                extras[name] = {"traversal": KERNEL_OF.get(rt.scene_info()["traversal"], "?").split(" ")[0],
                                "ms_per_frame": round(ms, 4), "Mrays_per_s": round(total_rays(st) / ms / 1e3, 1),
                                "rays": total_rays(st), "algorithmic_GBps": round(ab / ms / 1e6, 1),
                                "scene_build_s": round(t_build, 2)}
            rt.SendPrimitivesToShader(sc)
            # the headline workload as a throughput job: 8 orbit frames per launch (rt_render_frames_device).
            # A single 1080p frame leaves ~0.1 ms of ramp and tail per launch on the table; `value` above is
            # the single-frame figure (one step = one frame, as the reference draws them).
            batch = np.stack(orbit_cameras(sc, W, H, 8))
            out8 = torch.empty((8, H, W, 4), dtype=torch.uint8, device="cuda")
            s = torch.cuda.current_stream()
            for _ in range(3):
                rt.render_frames_device(batch, depth, H * W, rgba8=out8.data_ptr(), stream=s.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10):
                rt.render_frames_device(batch, depth, H * W, rgba8=out8.data_ptr(), stream=s.cuda_stream)
            e1.record(s)
            torch.cuda.synchronize()
            ms8 = e0.elapsed_time(e1) / 10
            rays8 = sum(total_rays(device_stats(rt, torch, c, depth)) for c in batch)
            extras[args.workload + "_8_frames_per_launch"] = {
                "ms_per_launch": round(ms8, 4), "ms_per_frame": round(ms8 / 8, 4),
                "Mrays_per_s": round(rays8 / ms8 / 1e3, 1), "rays": rays8}
            result["extra"] = extras
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(sc, cams[0], depth, W, H)
            result["gpu_over_cpu_1thread"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(result), flush=True)
    rt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
