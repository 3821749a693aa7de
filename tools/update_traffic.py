"""Fold one tools/gpu_pmc.sh result directory into profiles/: per-FRAME totals of the fast-path kernels
(sum over all their dispatches / frames rendered) go to profiles/traffic.json under the workload's name --
what bench.py's roofline objects read -- and the kernel stats + PMC summary are copied beside it.

    python tools/update_traffic.py [--dry-run] WORKLOAD gpurun_out/TAG FRAMES [ROUND_TAG] [--memory-bound] [--gather]

hbm_bytes_per_launch = (factor * FETCH_SIZE + WRITE_SIZE) KB * 1024 per frame.  gfx950's FETCH_SIZE tallies each
128-byte fabric read request of a wide streaming read as 64 bytes (MI355X_MICROARCH.md, HBM section): factor 2.
tools/fetch_calib.hip measured it for this code's other pattern, whole 64-byte records gathered at random from a
2 GiB table, each read once: FETCH_SIZE = 1.01 x the bytes read, factor 1 (--gather; profiles/r02_fetch_calibration.txt).
WRITE_SIZE is exact."""
import csv, glob, hashlib, json, os, shutil, subprocess, sys
from collections import defaultdict


def kernel_src_sha(root):
    """Fingerprint of what the measured kernels were built from (bench.py forms the same and flags a profile whose
    fingerprint differs from the tree's as stale): the device sources, the ABI header and the build flags."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "raytracing_amd", "csrc", "*.h")) + glob.glob(os.path.join(root, "raytracing_amd", "csrc", "*.hip")))
    files += [os.path.join(root, "include", "rt_mi355x.h"), os.path.join(root, "raytracing_amd", "build.py")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]

args = [a for a in sys.argv[1:] if not a.startswith("--")]
dry = "--dry-run" in sys.argv
wl, src, frames = args[0], args[1], int(args[2])
tag = args[3] if len(args) > 3 else "r03"
FAST = ("k_render_bvh", "k_render<false>", "k_wf_")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

tot, per_kernel = defaultdict(float), defaultdict(lambda: defaultdict(float))
for path in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if any(s in k for s in FAST):
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                per_kernel[k.split("(")[0].replace("void ", "").replace("rtd::", "")][row["Counter_Name"]] += float(row["Counter_Value"])
kernel_ns, calls = defaultdict(float), defaultdict(int)
for path in glob.glob(src + "/prof/**/*kernel_stats.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Name"]
            if any(s in k for s in FAST):
                kk = k.split("(")[0].replace("void ", "").replace("rtd::", "")
                kernel_ns[kk] += float(row["TotalDurationNs"])
                calls[kk] += int(row["Calls"])
per_frame = {c: v / frames for c, v in tot.items()}
entry = {"round": tag, "frames_profiled": frames,
         "kernels": {k: {"calls_per_frame": calls[k] / frames, "ms_per_frame": round(kernel_ns[k] / frames / 1e6, 4),
                         **{c: round(v / frames, 1) for c, v in per_kernel.get(k, {}).items()
                            if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "TA_TA_BUSY_sum")}}
                     for k in sorted(kernel_ns)},
         "kernel_ms_per_frame": round(sum(kernel_ns.values()) / frames / 1e6, 4)}
# --gather: the launch's reads are 64-byte record gathers (BVH nodes, triangle slots), for which FETCH_SIZE is exact
# (profiles/r02_fetch_calibration.txt); otherwise wide streaming reads, which it reports at half
factor = 1 if "--gather" in sys.argv else 2
if "FETCH_SIZE" in per_frame and "WRITE_SIZE" in per_frame:
    entry["FETCH_SIZE_KB"], entry["WRITE_SIZE_KB"] = round(per_frame["FETCH_SIZE"], 1), round(per_frame["WRITE_SIZE"], 1)
    entry["fetch_factor"] = factor
    entry["hbm_bytes_per_launch"] = int((factor * per_frame["FETCH_SIZE"] + per_frame["WRITE_SIZE"]) * 1024)
for c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS",
          "SQ_INSTS_SMEM", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum",
          "TCC_EA0_RDREQ_sum", "TA_TA_BUSY_sum"):
    if c in per_frame:
        entry[c] = round(per_frame[c], 1)
if "--memory-bound" in sys.argv:
    entry["memory_bound"] = True
for fname, key in (("run_unprofiled.txt", "ms_per_frame_unprofiled"), ("run.txt", "ms_per_frame_under_kernel_trace")):
    try:
        with open(os.path.join(src, fname)) as f:
            for line in f:
                if "ms_per_frame=" in line:
                    entry[key] = float(line.split("ms_per_frame=")[1].split()[0])   # (the second half of the frames, synchronised per frame)
    except OSError:
        pass
if "ms_per_frame_unprofiled" in entry and "ms_per_frame_under_kernel_trace" in entry:
    entry["profiler_overhead"] = round(entry["ms_per_frame_under_kernel_trace"] / entry["ms_per_frame_unprofiled"] - 1.0, 3)
entry["kernel_src_sha"] = kernel_src_sha(ROOT)   # (fold a run into profiles/ before touching the sources again)
try:
    entry["git_commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    entry["git_dirty"] = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "raytracing_amd", "include"],
                                             capture_output=True, text=True).stdout.strip())
except OSError:
    pass
short = wl.replace("_", "")
entry["source"] = f"profiles/{tag}_{short}_pmc_summary.txt, profiles/{tag}_{short}_kernel_stats.csv"
print(json.dumps({wl: entry}, indent=1))
if not dry:
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    with open(tpath) as f:
        t = json.load(f)
    t[wl] = entry
    t["_how"] = ("tools/gpu_pmc.sh WORKLOAD: rocprofv3 --kernel-trace --stats and one --pmc pass per counter group over "
                 "`tools/profile_workload.py WORKLOAD FRAMES`; per-frame totals of the fast-path kernels; hbm_bytes = "
                 "(fetch_factor*FETCH_SIZE + WRITE_SIZE)*1024; fetch_factor 2 for streaming reads (gfx950 FETCH_SIZE counts 64 B "
                 "per 128-B request, MI355X_MICROARCH.md HBM section), 1 for launches whose reads are 64-byte record gathers "
                 "(measured: profiles/r02_fetch_calibration.txt)")
    with open(tpath, "w") as f:
        json.dump(t, f, indent=1)
    shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(ROOT, "profiles", f"{tag}_{short}_pmc_summary.txt"))
    for p in glob.glob(src + "/prof/**/*kernel_stats.csv", recursive=True):
        shutil.copy(p, os.path.join(ROOT, "profiles", f"{tag}_{short}_kernel_stats.csv"))
