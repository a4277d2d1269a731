#!/usr/bin/env python3
"""Condense rocprofv3 --kernel-trace --stats CSV output into a small per-kernel table
(name, calls, total ms, avg us, %).  usage: prof_csv_summary.py <dir> [top]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    print(f"# {os.path.relpath(stats[0], d)}")
    print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'%':>6}  kernel")
    for r in rows[:top]:
        print(f"{int(r['Calls']):8d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} "
              f"{float(r['Percentage']):6.2f}  {r['Name'][:120]}")
    sys.exit(0)
tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not tr:
    print("no rocprofv3 csv found under", d); sys.exit(1)
agg = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(tr[0])):
    a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'%':>6}  kernel")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[0]:8d} {v[1]/1e6:10.3f} {v[1]/v[0]/1e3:9.2f} {100*v[1]/tot:6.2f}  {n[:120]}")
