#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection CSVs under <dir> -> per-kernel table of the counters' per-dispatch means.
    pmc_kernel_table.py <dir> [substring of the kernel names to keep]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if flt and flt not in k:
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, cs in sorted(agg.items()):
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"    {c:32s} {v / n:16.1f}   ({n} dispatches)")
