"""Join tools/class_probe.py's printed dispatch order with the counter_collection.csv of a rocprofv3 --pmc run of it:
mean counters per ray class and per wave.   usage: python tools/class_probe_join.py RUN.txt PMC_DIR [WAVES]"""
import csv, glob, sys
from collections import defaultdict

run, pmc = sys.argv[1], sys.argv[2]
order, ms = [], {}
for line in open(run):
    if line.startswith("order "):
        order = line.split()[1:]
    if line.startswith("class "):
        p = line.split()
        ms[p[1]] = float(p[3])
        waves = int(p[7])
rows = defaultdict(dict)  # dispatch id -> counter -> value
for path in glob.glob(pmc + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if "k_render_bvh" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
assert len(ids) == len(order), (len(ids), len(order))
acc = defaultdict(lambda: defaultdict(list))
for i, lab in zip(ids, order):
    if lab.endswith(":warm"):
        continue
    for c, v in rows[i].items():
        acc[lab][c].append(v)
for lab in acc:
    print(f"== {lab}: {ms.get(lab, float('nan')):.4f} ms per frame (unprofiled run), {waves} waves")
    for c in sorted(acc[lab]):
        v = sum(acc[lab][c]) / len(acc[lab][c])
        print(f"   {c:28s} {v:16.0f}   per wave {v / waves:10.1f}")
