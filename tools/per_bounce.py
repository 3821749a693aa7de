"""Per-bounce table of the staged pipeline from rocprofv3 output directories:
    python tools/per_bounce.py DEPTH TRACE_DIR [PMC_DIR ...]
TRACE_DIR holds a --kernel-trace run (…_kernel_trace.csv), each PMC_DIR a --pmc pass (…_counter_collection.csv) of the
same program.  Dispatches are taken in start order; a frame is DEPTH+1 iterations, each opened by a k_wf_closest.
Prints per iteration the mean ms of each stage over the frames and the per-frame means of the counters."""
import csv, glob, os, sys
from collections import defaultdict

depth = int(sys.argv[1])
STAGES = ("k_wf_closest", "k_wf_any", "k_wf_shade", "k_wf_bin_scan", "k_wf_bin_scatter")


def stage_of(name):
    for s in STAGES:
        if s in name:
            return s
    return None


def rows(d, suffix):
    for path in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        with open(path, newline="") as f:
            yield from csv.DictReader(f)


def by_iteration(items):
    """items: (order key, stage, value dict) -> {(iteration, stage): [value dict per frame]}"""
    out, it = defaultdict(list), -1
    for _, stage, val in sorted(items, key=lambda r: r[0]):
        if stage == "k_wf_closest":
            it = (it + 1) % (depth + 1)
        out[(it, stage)].append(val)
    return out


trace = by_iteration((int(r["Start_Timestamp"]), stage_of(r["Kernel_Name"]), {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
                     for r in rows(sys.argv[2], "kernel_trace.csv") if stage_of(r["Kernel_Name"]))
counters = defaultdict(dict)
for d in sys.argv[3:]:
    per = defaultdict(dict)
    for r in rows(d, "counter_collection.csv"):
        if stage_of(r["Kernel_Name"]):
            per[int(r["Dispatch_Id"])].setdefault("_stage", stage_of(r["Kernel_Name"]))
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for key, vals in by_iteration((i, v.pop("_stage"), v) for i, v in per.items()).items():
        for c in vals[0]:
            counters[key][c] = sum(v.get(c, 0.0) for v in vals) / len(vals)
mean = lambda key: sum(v["ms"] for v in trace[key]) / len(trace[key]) if trace.get(key) else 0.0
frames = len(trace[(0, "k_wf_closest")])
print(f"{frames} frames; ms per stage and iteration (mean over the frames)")
print(f"{'iter':>4} {'closest':>8} {'any':>8} {'shade':>8} {'bin':>8}   L2 hit rate closest / any   active lanes closest / any")
total = 0.0
for it in range(depth + 1):
    ms = [mean((it, s)) for s in STAGES]
    total += sum(ms)
    extra = ""
    for s in ("k_wf_closest", "k_wf_any"):
        c = counters.get((it, s), {})
        hit = f"{c['TCC_HIT_sum'] / c['TCC_REQ_sum']:.3f}" if c.get("TCC_REQ_sum") else "  -  "
        lanes = f"{c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_INSTS_VALU']):.3f}" if c.get("SQ_INSTS_VALU") else "  -  "
        extra += f"   {s[5:]:>7}: L2 {hit} lanes {lanes}"
    print(f"{it:>4} {ms[0]:>8.3f} {ms[1]:>8.3f} {ms[2]:>8.3f} {ms[3] + ms[4]:>8.3f}{extra}")
print(f"sum of the stages: {total:.3f} ms per frame")
