"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of each counter, per kernel."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                acc[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{path.split('/')[-3]:12s} {k:42s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
