#!/usr/bin/env python3
"""bench.py -- Mrays/s of the per-pixel ray-casting hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--strong]

`--gpus N` with N > 1 starts its own N ranks (one process per GPU, RANK/LOCAL_RANK/WORLD_SIZE set, rendezvous
on 127.0.0.1) unless it already runs under a launcher (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` sets WORLD_SIZE), in which case it is one of the ranks.  The parent never touches the GPU.

Headline workload: TeapotScene -- BASELINE.json configs[1]'s scene (assets/teapot via OBJImportHelper + RTree
over the mirror floor, TestScene1's six lights, shadows + 1 reflection bounce) at 3840x2160, the size
BASELINE.json's metric ("ms/frame at 4K") and north_star's target ("teapot scene at 4K") are quoted on.
configs[1]'s own 1920x1080 frame, configs[0] (TestScene1 720p), configs[2] (Soup1M 4K: HBM-side traversal)
and configs[4]'s frame (TeapotTable 1080p depth 8) are measured in the same run under "extra", each with its
own roofline object.

Steps.  weak (default): a step renders N frames of a camera orbit; every frame is cut into 8-row bands dealt
round-robin to the N ranks, every rank renders its bands of all N frames in one launch, and ONE RCCL
all-to-all per step delivers frame f's bands to rank f, which restores scanline order: N concurrent gathers,
one root each, so per-GPU work AND per-GPU traffic stay one frame's worth at any N.  The exchange of step s-1
overlaps the render of step s (double buffered, at most one collective in flight).
strong (`--strong`): ONE frame per step, its bands split over the N ranks and gathered on rank 0
(BASELINE.json configs[3]: `--strong --workload soup1m_8k_d4`).

Rays are counted by the reference's rule (primary + executed mirror bounces + shadow rays) with the
reference-order kernel outside the timed region; the count is identical to the oracle's.  `rays_traced` is
what the fast kernels actually trace: they leave out shadow rays whose light term is an exact zero
either way (same pixels), which the reference traces.

`--workload teapottable_sweep256_d8` is BASELINE.json configs[4] as one command: the 256 frames of the camera sweep
(host.camera_orbit) at 1920x1080, depth 8; a step renders 8 x N consecutive frames of it (8 frames per launch on
every rank: its bands of them), frame f is assembled on rank f % N by one all-to-all per step, and the steps go
round the sweep (32 / N steps = one sweep; the default number of steps).
`--capi-multi --gpus N` measures the multi-GPU render BEHIND the C ABI instead (single process, rt_create_multi over
devices 0..N-1: one rt_render_device call = one frame, bands on every device, grouped ncclSend/ncclRecv gather on
the root, device-resident result; `--strong` semantics).  BENCH_MULTI_DEVICES=0,0,0 plays the devices on one GPU.

One JSON line is printed by rank 0 (contract in the task description) with two extra objects:
  roofline      the bound that applies to the dominant kernel (<= 1 by construction), see roofline_of()
  cpu_baseline  the CPU oracle ("port") timed on a bounded sample of the same frame
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KERNEL_OF = {1: "k_render<false> (wave-cooperative walk)", 2: "k_render_bvh (per-lane BVH, LDS stacks)",
             3: "k_wf_closest + k_wf_any + k_wf_shade per bounce (wavefront pipeline, whole frame)"}
FP32_VECTOR_PEAK_TF = 157.3  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 lanes x 2 (FMA) x 2.4 GHz
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD, CLOCK_HZ = 1024, 2.4e9
N_SIMD_TOTAL = 1024          # 256 CUs x 4 SIMDs: the VALU pipes that profiles/valu_model.json's per-SIMD issue intervals refer to
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
    "teapottable_sweep256_d8": ("TeapotTable", None, 1920, 1080, 8),   # configs[4]: the 256-frame orbit, 8 x N frames per step
}
SWEEP = "teapottable_sweep256_d8"
SWEEP_FRAMES, SWEEP_BATCH = 256, 8
HEADLINE = "teapot_4k_d1"
EXTRAS = ["teapot_1080p_d1", "testscene1_720p_d10", "teapottable_1080p_d8", "soup100k_1080p_d4", "soup1m_4k_d4"]


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
    triangle, 8 per plane test) -- the VALU-side figure."""
    return (27 * stats["node_tests"] + 30 * stats["sphere_tests"] + 75 * stats["triangle_tests"] +
            8 * stats["plane_tests"])


def total_rays(stats):
    return stats["primary_rays"] + stats["secondary_rays"] + stats["shadow_rays"]


def traced_rays(stats):
    return total_rays(stats) - stats.get("shadow_rays_zero_term", 0)


def orbit_cameras(sc, width, height, n):
    """n cameras on a slow orbit around the scene camera's own view (frame 0 = the scene camera)."""
    import numpy as np
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


TUNE_FRAMES = 12  # renders after which the library's kernel choice by measurement is final (rt_mi355x.hip tune_begin)


def time_single(rt, torch, cam, depth, width, height, steps, warmup):
    """ms per frame of the fast kernel, device-resident RGBA8 output, HIP events on the launch stream.
    (>= TUNE_FRAMES warm-up frames: the library times its candidate kernels over the first frames of a (scene, size,
    bounces) -- three frames each for up to three candidates -- and keeps the fastest.)"""
    out = torch.empty((height, width, 4), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()
    for _ in range(max(warmup, TUNE_FRAMES)):
        rt.render_device(cam, depth, rgba8=out.data_ptr(), stream=s.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(steps):
        rt.render_device(cam, depth, rgba8=out.data_ptr(), stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


_profiles = None
_valu_model = None
_src_sha = None


def kernel_src_sha():
    """Fingerprint of the sources the kernels are built from (the same rule as tools/update_traffic.py, which stores
    it with every folded PMC run): a committed profile with another fingerprint was measured on other kernels."""
    global _src_sha
    if _src_sha is None:
        h = hashlib.sha256()
        files = sorted(glob.glob(os.path.join(ROOT, "raytracing_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "raytracing_amd", "csrc", "*.hip")))
        files += [os.path.join(ROOT, "include", "rt_mi355x.h"), os.path.join(ROOT, "raytracing_amd", "build.py")]
        for f in files:
            h.update(os.path.basename(f).encode() + b"\0")
            with open(f, "rb") as fh:
                h.update(fh.read())
        _src_sha = h.hexdigest()[:16]
    return _src_sha


def profile_of(workload):
    """Counters of a committed rocprofv3 --pmc run of this workload (profiles/traffic.json), or {}."""
    global _profiles
    if _profiles is None:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                _profiles = json.load(f)
        except (OSError, ValueError):
            _profiles = {}
    return _profiles.get(workload, {})


def valu_pipe_seconds(prof):
    """Seconds of VALU pipe time (summed over the SIMDs) the profiled frame's VALU instructions take at the issue
    intervals tools/valu_rate.hip measured per instruction KIND on a saturated SIMD (profiles/valu_model.json:
    plain fp/int 1.2 ns, compares / min / max / conversions / division helpers ~1.8, transcendental 3.4, packed 2.4),
    weighted per kernel by its static instruction mix; the transcendental share is the dynamic counter's."""
    global _valu_model
    if _valu_model is None:
        try:
            with open(os.path.join(ROOT, "profiles", "valu_model.json")) as f:
                _valu_model = json.load(f)
        except (OSError, ValueError):
            _valu_model = {}
    ns = _valu_model.get("ns_per_wave_instruction_per_simd", {})
    total = 0.0
    for name, k in prof.get("kernels", {}).items():
        model = _valu_model.get("kernels", {}).get(name)
        insts = k.get("SQ_INSTS_VALU")
        if not model or not insts or "trans" not in ns:
            return None
        trans = k.get("SQ_INSTS_VALU_TRANS_F32", model["static_transcendental_share"] * insts)
        total += ((insts - trans) * model["ns_per_non_transcendental_instruction"] + trans * ns["trans"]) * 1e-9
    return total or None


def roofline_of(workload, stats, width, height, ms, traversal, share=1.0):
    """The bound of the dominant kernel as a fraction <= 1.

    One "launch" = one frame (the megakernels: one kernel; the staged pipeline: its 3 x (max_bounces + 1) kernels
    issued back to back), `ms` = its duration measured live with HIP events on the launch stream.
    Scenes that live in L2 (every reference-sized scene) are bound by VALU issue: achieved = flops of the REFERENCE
    algorithm (27 per box test, 30 per sphere, 75 per triangle, 8 per plane test; every candidate
    tested) / ms against the fp32 vector peak, plus -- from the committed PMC pass -- the share of VALU issue slots
    in use and the active lanes per instruction.  Soup1M (113 MB of hot nodes and triangles, nothing of it in
    L2) is bound by the memory side: achieved = bytes that crossed the L2 -> fabric boundary per frame (PMC,
    gfx950-corrected: 2 x FETCH_SIZE + WRITE_SIZE) / ms against the HBM peak.  The byte count of the reference
    algorithm is kept beside it (`reference_algorithm_bytes_per_launch`): the kernels cull, order and stop early,
    so they move far fewer bytes than it, and dividing it by the time exceeds every physical bandwidth.
    share: the part of a frame one launch renders (1/N on a rank of a strong-scaled frame)."""
    prof = profile_of(workload)
    # a profile measured on other kernel sources says nothing about this build's instruction counts or traffic
    stale = bool(prof) and prof.get("kernel_src_sha") != kernel_src_sha()
    alg_flops = algorithmic_flops(stats) * share
    alg_bytes = algorithmic_bytes(stats, width, height) * share
    tf = alg_flops / (ms * 1e-3) / 1e12
    traffic = prof.get("hbm_bytes_per_launch")
    valu = {"achieved_tflops": round(tf, 2), "peak_tflops": FP32_VECTOR_PEAK_TF, "frac": round(tf / FP32_VECTOR_PEAK_TF, 4),
            "algorithmic_flops_per_launch": int(alg_flops)}
    pipe_s = None if stale else valu_pipe_seconds(prof)
    if pipe_s:
        # the VALU pipes' time for this frame's instructions (measured per-kind issue intervals) over the SIMD-time of the launch
        valu["issue_occupancy"] = round(pipe_s * share / (N_SIMD_TOTAL * ms * 1e-3), 3)
    if not stale and prof.get("SQ_THREAD_CYCLES_VALU") and prof.get("SQ_ACTIVE_INST_VALU"):
        # (SQ_ACTIVE_INST_VALU counts 1 per instruction and 2 per transcendental one, SQ_THREAD_CYCLES_VALU the same times
        # the active lanes -- calibrated with tools/valu_rate.hip on full waves: 64 and 128 per instruction)
        valu["active_lanes"] = round(prof["SQ_THREAD_CYCLES_VALU"] / (64.0 * prof["SQ_ACTIVE_INST_VALU"]), 3)
    hbm = None
    if traffic:
        gbs = traffic * share / (ms * 1e-3) / 1e9
        hbm = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    memory_bound = bool(prof.get("memory_bound")) and hbm is not None
    r = {"bound": "hbm" if memory_bound else "valu"}
    if memory_bound:
        r.update(hbm)
        r["valu"] = valu
    else:
        r.update({"achieved": valu["achieved_tflops"], "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s", "frac": valu["frac"]})
        for k in ("issue_occupancy", "active_lanes", "algorithmic_flops_per_launch"):
            if k in valu:
                r[k] = valu[k]
        if hbm:
            r["hbm"] = hbm
    r["traffic"] = int(traffic * share) if traffic else None
    r["kernel"] = KERNEL_OF.get(traversal, "?")
    r["kernel_ms"] = round(ms, 4)
    r["reference_algorithm_bytes_per_launch"] = int(alg_bytes)
    if prof.get("source"):
        r["pmc_source"] = prof["source"]
        r["pmc_commit"] = prof.get("git_commit")
        r["pmc_kernel_ms"] = prof.get("kernel_ms_per_frame")
    r["pmc_stale"] = stale if prof else None
    return r


def cpu_baseline(sc, cam, depth, width, height, budget_s=12.0):
    """The oracle (a C++ port of the reference algorithm; the reference itself cannot run here),
    single-threaded, on evenly spaced full rows of the same frame, sized to ~budget_s of CPU; then the same
    code with an OpenMP row loop on every core this process may use."""
    import numpy as np
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
    # every core this process may run on (the GPU box exports OMP_NUM_THREADS = its CPU share; honour it)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        cores = max(1, min(cores, int(os.environ["OMP_NUM_THREADS"])))

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
    while passes == 0 or time.perf_counter() - t0 < 5.0:
        rays_all += frame_all()
        passes += 1
    dt_all = time.perf_counter() - t0
    allc = rays_all / dt_all / 1e6
    return {"value": round(one, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"{what} ({rays} rays, {dt:.1f} s, C++ restatement of the reference algorithm, "
                      f"1 thread, row by row)",
            "all_cores": {"value": round(allc, 3), "cores": cores,
                          "sample": f"{all_what}, {passes} passes, OpenMP row loop over {cores} threads "
                                    f"(every core this process may use), {dt_all:.1f} s"}}


def dbg(*a):
    if os.environ.get("BENCH_DEBUG"):
        print(f"[rank {os.environ.get('RANK', '0')}]", *a, file=sys.stderr, flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N fresh rank processes (this process has not
    touched the GPU and never does), wait for them, return the worst exit code.  Rank 0's stdout is ours."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0:
                        rc = rc or code
                        for q in pending:          # one rank failed: the others would wait in a collective forever
                            procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def measure_extra(rt, torch, host, name):
    """One single-GPU workload outside the contract's timed region: ms/frame, rays and its roofline."""
    sn, pr, w, h, d = WORKLOADS[name]
    t_build = time.perf_counter()
    s2 = host.make_scene(sn, params=pr)
    t_build = time.perf_counter() - t_build
    t_up = time.perf_counter()
    rt.SendPrimitivesToShader(s2)
    t_up = time.perf_counter() - t_up
    cam = s2.camera_for(w, h)
    st = device_stats(rt, torch, cam, d)
    big = name.startswith("soup1m")
    ms = time_single(rt, torch, cam, d, w, h, 5 if big else 20, 5)
    trav = rt.scene_info()["traversal"]
    out = {"ms_per_frame": round(ms, 4), "Mrays_per_s": round(total_rays(st) / ms / 1e3, 1),
           "Mrays_per_s_traced": round(traced_rays(st) / ms / 1e3, 1),
           "rays": total_rays(st), "rays_traced": traced_rays(st),
           "scene_build_s": round(t_build, 2), "upload_s": round(t_up, 2),
           "roofline": roofline_of(name, st, w, h, ms, trav)}
    if name == "testscene1_720p_d10":
        # This scene's frames are dealt out in an order ranked from the PREVIOUS launch's wave lifetimes (its root box spans
        # the view, paths bounce ten times: RT_OPTION_FRAME_FEEDBACK), so ms_per_frame above is the steady state of a
        # repeated view.  The same frame with the feedback off = the first frame of a view, or any frame after a cut.
        rt.set_option(rt.OPTION_FRAME_FEEDBACK, 0)
        cold = time_single(rt, torch, cam, d, w, h, 20, 5)
        rt.set_option(rt.OPTION_FRAME_FEEDBACK, 1)
        out["ms_per_frame_is"] = "steady state of a repeated view (tiles ordered by the previous frame's wave lifetimes)"
        out["ms_first_frame_of_a_view"] = round(cold, 4)
        out["Mrays_per_s_first_frame_of_a_view"] = round(total_rays(st) / cold / 1e3, 1)
    return out


def capi_multi(args):
    """`--capi-multi --gpus N`: the multi-GPU render behind the C ABI (rt_create_multi), as the C# host would drive it --
    one process, one context over N devices, one rt_render_device call = one finished frame on the root device
    (bands on every device's own stream, all planes in one grouped ncclSend/ncclRecv gather, un-permute on the root).
    Strong scaling by construction: one frame per step, split over the devices."""
    import torch
    from raytracing_amd import RayTracer, host
    devs = os.environ.get("BENCH_MULTI_DEVICES")   # e.g. "0,0,0": the devices played on one GPU (copies instead of RCCL)
    devices = [int(x) for x in devs.split(",")] if devs else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but BENCH_MULTI_DEVICES names {len(devices)} devices")
    scene_name, params, W, H, depth = WORKLOADS[args.workload]
    if args.workload == SWEEP:
        raise SystemExit("--capi-multi renders single frames; the sweep is a per-process workload")
    torch.cuda.set_device(devices[0])
    sc = host.make_scene(scene_name, params=params)
    rt = RayTracer(devices)
    info = rt.multi_info()
    rt.SendPrimitivesToShader(sc)
    cam = sc.camera_for(W, H)
    rt.PrepareRenderOpenGL(cam)
    st = rt.RenderGL(depth, want=(), stats=True)["stats"]   # reference-order kernel on every device, counters summed
    out = torch.empty((H, W, 4), dtype=torch.uint8, device=torch.device("cuda", devices[0]))
    s = torch.cuda.current_stream()
    for _ in range(max(args.warmup, TUNE_FRAMES)):
        rt.render_device(cam, depth, rgba8=out.data_ptr(), stream=s.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(s)
    for _ in range(args.steps):
        rt.render_device(cam, depth, rgba8=out.data_ptr(), stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if args.verify:
        with RayTracer(devices[0]) as one:
            one.SendPrimitivesToShader(sc)
            ref = torch.empty_like(out)
            one.render_device(cam, depth, rgba8=ref.data_ptr(), stream=s.cuda_stream)
            torch.cuda.synchronize()
            if not torch.equal(ref, out):
                raise SystemExit("the frame assembled by the multi-GPU context differs from the single-GPU render")
    n = len(devices)
    ms = elapsed / args.steps * 1e3
    result = {
        "metric": "Mrays/s (primary+shadow+secondary)", "value": round(total_rays(st) / (elapsed / args.steps) / 1e6, 2),
        "value_traced": round(traced_rays(st) / (elapsed / args.steps) / 1e6, 2), "unit": "Mrays/s",
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (mesh fixtures of the reference assets / seeded soup, builder-defined scenes)",
        "config": {"workload": args.workload, "scene": scene_name, "width": W, "height": H, "max_bounces": depth,
                   "lights": sc.n_lights, "triangles": len(sc.trianglesData), "spheres": len(sc.spheresData),
                   "frames_per_step": 1, "rays_per_step": total_rays(st), "rays_traced_per_step": traced_rays(st),
                   "sharding": f"C ABI, one process: rt_create_multi over devices {devices}; {BAND_ROWS}-row bands round-robin, "
                               f"one frame per rt_render_device call, assembled on device {devices[0]}"},
        "n_ranks": n, "backend": "rccl (ncclSend/ncclRecv, in-library)" if info["uses_rccl"] else "device-to-device copies (devices played on one GPU)",
        "n_ranks_rccl": n if info["uses_rccl"] else 0,
        "ms_per_frame": round(ms, 4), "ms_per_frame_gpu_events": round(e0.elapsed_time(e1) / args.steps, 4),
        "verified": bool(args.verify),
        "roofline": roofline_of(args.workload, st, W, H, e0.elapsed_time(e1) / args.steps, rt.scene_info()["traversal"], share=1.0),
    }
    result["roofline"]["note"] = "one launch = one frame over all devices; the committed PMC pass (if any) is the single-GPU frame's"
    print(json.dumps(result), flush=True)
    rt.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--strong", action="store_true",
                    help="ONE frame per step split over the ranks and gathered on rank 0 (default: weak, N frames per step)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra single-GPU workloads")
    ap.add_argument("--no-soup1m", action="store_true", help="skip the 1M-triangle soup among the extras")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--verify", action="store_true", help="check the assembled frames against whole-frame renders")
    ap.add_argument("--capi-multi", action="store_true",
                    help="single process: one rt_create_multi context over devices 0..N-1, one rt_render_device call per frame")
    args = ap.parse_args()
    if args.steps == 50 and args.workload == SWEEP and "--steps" not in " ".join(sys.argv):
        args.steps = SWEEP_FRAMES // (SWEEP_BATCH * max(args.gpus, 1))   # one pass over the sweep

    if args.capi_multi:
        return capi_multi(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))   # before anything touches the GPU

    import numpy as np
    import torch
    import torch.distributed as dist
    from raytracing_amd import RayTracer, host
    from raytracing_amd import parallel as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # BENCH_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the N > 1 code path on a one-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    n_ranks_rccl = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        n_ranks_rccl = dist.get_world_size()

    dbg("process group up")
    scene_name, params, W, H, depth = WORKLOADS[args.workload]
    sc = host.make_scene(scene_name, params=params)
    rt = RayTracer(local_rank)
    rt.SendPrimitivesToShader(sc)
    sweep = args.workload == SWEEP
    if sweep and (args.strong or SWEEP_FRAMES % (SWEEP_BATCH * world) != 0):
        raise SystemExit(f"{SWEEP}: weak mode only, and {SWEEP_BATCH} x the number of GPUs must divide {SWEEP_FRAMES}")
    n_frames = 1 if args.strong else (SWEEP_BATCH * world if sweep else world)
    if sweep:   # step s renders frames [s * n_frames, (s + 1) * n_frames) of the sweep, round and round
        all_cams = np.stack([host.camera_orbit(f, SWEEP_FRAMES, W, H) for f in range(SWEEP_FRAMES)])
        step_cams = [all_cams[k * n_frames:(k + 1) * n_frames] for k in range(SWEEP_FRAMES // n_frames)]
    else:
        step_cams = [orbit_cameras(sc, W, H, n_frames)]
    cams = step_cams[0]

    # rays per step: every frame of the step, counted by the reference-order kernel (untimed); the sweep's steps differ,
    # so every frame the timed region renders is counted (rank 0 needs them, the others only frame 0's for the roofline)
    n_distinct = min(len(step_cams), args.steps)
    counted = [[device_stats(rt, torch, cam, depth) for cam in step_cams[k]] if (rank == 0 or k == 0) else None for k in range(n_distinct)]
    frame_stats = counted[0]
    rays_per_step = sum(total_rays(s) for s in frame_stats)
    traced_per_step = sum(traced_rays(s) for s in frame_stats)
    rays_timed = traced_timed = None
    if rank == 0:
        rays_timed = sum(sum(total_rays(s) for s in counted[k % n_distinct]) for k in range(args.steps))
        traced_timed = sum(sum(traced_rays(s) for s in counted[k % n_distinct]) for k in range(args.steps))
    dbg("stats done", rays_per_step)

    # ---- timed region: K steps, each = this rank's bands of n_frames frames + ONE collective ----
    stream = torch.cuda.current_stream()
    assemble = "root" if args.strong else "owner"
    bufs = [P.ShardedFrames(rt, W, H, rank, world, BAND_ROWS, n_frames, assemble=assemble) for _ in range(2)]  # double buffered
    if bufs[0].staged and rank == 0:
        print(f"bench: backend {backend} cannot move device tensors in this collective; staging through host memory "
              f"(rehearsal only)", file=sys.stderr)
    for _ in range(TUNE_FRAMES):  # the library settles on a kernel for this (scene, local frame size, bounces) before anything is timed
        bufs[0].render(cams, depth)
        torch.cuda.synchronize()
    timed_step = [0]   # which step of the sweep a timed step renders (warm-up steps render step 0's frames)

    step_no, timing = [0], [False]
    # N > 1: HIP events around each render launch of the timed region (the first 512 steps), on the launch stream --
    # the roofline's kernel time must not include what the stream waits for between launches (the exchange).  N = 1:
    # nothing lies between the launches, one pair around the whole region measures them without touching the stream.
    launch_events = []

    def step(timed=False):
        # step s renders while the exchange of step s-1 is in flight on RCCL's stream
        b, other = bufs[step_no[0] % 2], bufs[(step_no[0] + 1) % 2]
        b.finish()               # this buffer's previous exchange: scanline order on the frame's owner
        my_cams = step_cams[timed_step[0] % len(step_cams)] if timing[0] else cams
        if timing[0]:
            timed_step[0] += 1
        if timed and len(launch_events) < 512:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record(stream)
            b.render(my_cams, depth)
            pair[1].record(stream)
            launch_events.append(pair)
        else:
            b.render(my_cams, depth)
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
    fence()
    dbg("fenced")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    timing[0] = True
    for _ in range(args.steps):
        step(timed=world > 1)
    timing[0] = False
    drain()
    e1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    dbg("timed region done", elapsed)
    if args.verify:  # the frames this rank assembled in the last step equal whole-frame renders
        last = bufs[(step_no[0] - 1) % 2]
        last_cams = step_cams[(timed_step[0] - 1) % len(step_cams)]
        if sweep:   # frames 0 / 64 / 128 / 192 of the sweep, whichever steps hold them: render those steps once more
            for f_abs in (0, 64, 128, 192):
                k = f_abs // n_frames
                last.render(step_cams[k], depth)
                last.gather_async()
                last.finish()
                torch.cuda.synchronize()
                for f, frame in last.owned_frames():
                    if k * n_frames + f != f_abs:
                        continue
                    ref = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
                    rt.render_device(step_cams[k][f], depth, rgba8=ref.data_ptr(), stream=stream.cuda_stream)
                    torch.cuda.synchronize()
                    if not torch.equal(ref, frame):
                        raise SystemExit(f"rank {rank}: sweep frame {f_abs} differs from the single-GPU render")
                    dbg("verified sweep frame", f_abs)
            last_cams = step_cams[192 // n_frames]
        for f, frame in last.owned_frames():
            ref = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
            rt.render_device(last_cams[f], depth, rgba8=ref.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize()
            if not torch.equal(ref, frame):
                raise SystemExit(f"rank {rank}: assembled frame {f} differs from the single-GPU render")
        dbg("verified", [f for f, _ in last.owned_frames()])
    # per render launch (one per step), on the launch stream
    kernel_ms = (sum(a.elapsed_time(b) for a, b in launch_events) / len(launch_events)) if launch_events else e0.elapsed_time(e1) / args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_timed / elapsed / 1e6
        st0 = frame_stats[0]
        trav = rt.scene_info()["traversal"]
        # one launch renders this rank's bands of every frame of the step: n_frames / world frames' worth
        mean_stats = {k: sum(s[k] for s in frame_stats) / len(frame_stats) for k in st0}
        result = {
            "metric": "Mrays/s (primary+shadow+secondary)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (mesh fixtures of the reference assets / seeded soup, builder-defined scenes, orbit cameras)",
            "config": {"workload": args.workload, "scene": scene_name, "width": W, "height": H,
                       "max_bounces": depth, "lights": sc.n_lights, "triangles": len(sc.trianglesData),
                       "spheres": len(sc.spheresData), "frames_per_step": n_frames, "rays_per_step": rays_per_step,
                       "rays_traced_per_step": traced_per_step,
                       "rays_note": "value counts the reference's rays (every shadow ray of its algorithm); "
                                    "rays_traced leaves out the shadow rays whose light term is an exact zero "
                                    "either way, which the fast kernels do not trace (identical pixels)",
                       "sharding": ("single GPU, whole frame" if world == 1 else
                                    f"{BAND_ROWS}-row bands round-robin over {world} ranks; " +
                                    ("one frame per step gathered on rank 0" if args.strong else
                                     f"{n_frames} frames per step, one all-to-all: frame f assembled on rank f % {world}")),
                       "steps_note": (f"step s renders frames [{n_frames} s, {n_frames} (s + 1)) of the {SWEEP_FRAMES}-frame sweep, "
                                      f"{SWEEP_BATCH} frames per launch; {SWEEP_FRAMES // n_frames} steps = one sweep" if sweep else None)},
            "value_traced": round(traced_timed / elapsed / 1e6, 2),   # the same, counting only the rays the fast kernels trace
            "n_ranks": world, "backend": backend if world > 1 else None,
            "n_ranks_rccl": (world if backend == "nccl" else 0) if world > 1 else 1,
            "ms_per_frame": round(ms_per_step / n_frames, 4),
            "ray_breakdown_frame0": {k: st0[k] for k in ("primary_rays", "secondary_rays", "shadow_rays",
                                                         "shadow_rays_zero_term")},
            "roofline": roofline_of(args.workload, mean_stats, W, H, kernel_ms, trav, share=n_frames / world),
        }

    # ---- N = 1 only: extra workloads and the CPU baseline (outside the contract's timed region) ----
    if world == 1 and rank == 0:
        if not args.no_extras:
            extras = {}
            for name in EXTRAS:
                if name == args.workload or (args.no_soup1m and name.startswith("soup1m")):
                    continue
                extras[name] = measure_extra(rt, torch, host, name)
            rt.SendPrimitivesToShader(sc)
            # configs[1]'s frames as a throughput job: 8 orbit frames per launch (rt_render_frames_device).  The
            # frames of a launch fill each other's ramp and tail (1080p: ~0.06 ms of a single-frame launch).
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for wl8 in ("teapot_1080p_d1", "teapot_4k_d1"):
                sn, pr, w8, h8, d8 = WORKLOADS[wl8]
                s8 = sc if sn == scene_name else host.make_scene(sn, params=pr)
                rt.SendPrimitivesToShader(s8)
                batch = np.stack(orbit_cameras(s8, w8, h8, 8))
                out8 = torch.empty((8, h8, w8, 4), dtype=torch.uint8, device="cuda")
                for _ in range(3):
                    rt.render_frames_device(batch, d8, h8 * w8, rgba8=out8.data_ptr(), stream=s.cuda_stream)
                e0.record(s)
                for _ in range(10):
                    rt.render_frames_device(batch, d8, h8 * w8, rgba8=out8.data_ptr(), stream=s.cuda_stream)
                e1.record(s)
                torch.cuda.synchronize()
                ms8 = e0.elapsed_time(e1) / 10
                rays8 = sum(total_rays(device_stats(rt, torch, c, d8)) for c in batch)
                extras[wl8 + "_8_frames_per_launch"] = {
                    "ms_per_launch": round(ms8, 4), "ms_per_frame": round(ms8 / 8, 4),
                    "Mrays_per_s": round(rays8 / ms8 / 1e3, 1), "rays": rays8}
                del out8
            # configs[4]: the 256-frame camera sweep of the teapot+table scene at depth 8 (host.camera_orbit, the sweep
            # tests/test_gpu_configs.py checks four frames of against the oracle), 8 frames per launch.  Rays are counted
            # on every 32nd frame (the reference-order kernel is ~10x slower than the sweep itself).
            sn, pr, w5, h5, d5 = WORKLOADS["teapottable_1080p_d8"]
            s5 = host.make_scene(sn, params=pr)
            rt.SendPrimitivesToShader(s5)
            sweep = np.stack([host.camera_orbit(f, 256, w5, h5) for f in range(256)])
            out5 = torch.empty((8, h5, w5, 4), dtype=torch.uint8, device="cuda")
            for f0 in (0, 8):
                rt.render_frames_device(sweep[f0:f0 + 8], d5, h5 * w5, rgba8=out5.data_ptr(), stream=s.cuda_stream)
            e0.record(s)
            for f0 in range(0, 256, 8):
                rt.render_frames_device(sweep[f0:f0 + 8], d5, h5 * w5, rgba8=out5.data_ptr(), stream=s.cuda_stream)
            e1.record(s)
            torch.cuda.synchronize()
            ms5 = e0.elapsed_time(e1)
            rays5 = sum(total_rays(device_stats(rt, torch, sweep[f], d5)) for f in range(0, 256, 32)) * 32
            extras["teapottable_1080p_d8_sweep_256_frames"] = {
                "ms_total": round(ms5, 2), "ms_per_frame": round(ms5 / 256, 4), "frames_per_launch": 8,
                "Mrays_per_s": round(rays5 / ms5 / 1e3, 1), "rays_estimate": rays5,
                "rays_note": "32 x the rays of frames 0, 32, ..., 224"}
            rt.SendPrimitivesToShader(sc)
            result["extra"] = extras
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(sc, cams[0], depth, W, H)
            result["gpu_over_cpu_1thread"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
            result["gpu_over_cpu_all_cores"] = round(result["value"] / result["cpu_baseline"]["all_cores"]["value"], 1)
    if rank == 0:
        print(json.dumps(result), flush=True)
    rt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
