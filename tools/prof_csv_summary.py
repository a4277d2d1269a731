#!/usr/bin/env python3
"""Condense rocprofv3 --kernel-trace --stats CSV output into a small per-kernel table.

    prof_csv_summary.py <dir> [top] [--last-ms X]

Without --last-ms: the tool's own *_kernel_stats.csv (whole process, includes warm-up / MIOpen find).
With --last-ms X : only dispatches that start in the last X ms of the trace (the timed steady-state
steps of bench.py), aggregated from *_kernel_trace.csv."""
import csv, glob, os, sys
from collections import defaultdict
argv = [a for a in sys.argv[1:] if a != "--by-grid"]
last_ms = None
if "--last-ms" in argv:
    i = argv.index("--last-ms")
    last_ms = float(argv[i + 1])
    del argv[i:i + 2]
d = argv[0]
top = int(argv[1]) if len(argv) > 1 else 40
stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if stats and last_ms is None:
    rows = list(csv.DictReader(open(stats[0])))
    print(f"# {os.path.relpath(stats[0], d)} (whole process)")
    print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'%':>6}  kernel")
    for r in rows[:top]:
        print(f"{int(r['Calls']):8d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} "
              f"{float(r['Percentage']):6.2f}  {r['Name'][:120]}")
    sys.exit(0)
tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not tr:
    print("no rocprofv3 csv found under", d); sys.exit(1)
by_grid = "--by-grid" in sys.argv        # split a symbol by launch geometry (e.g. the MIOpen kernels of different layers)
rows = [(r["Kernel_Name"] + (f"  grid={r.get('Grid_Size_X', r.get('Grid_Size', '?'))}" if by_grid else ""),
         float(r["Start_Timestamp"]), float(r["End_Timestamp"])) for r in csv.DictReader(open(tr[0]))]
t_end = max(r[2] for r in rows)
if last_ms is not None:
    rows = [r for r in rows if r[1] >= t_end - last_ms * 1e6]
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    a = agg[n]; a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
span = (t_end - min(r[1] for r in rows)) / 1e6
print(f"# steady-state window: last {last_ms} ms of the trace; {len(rows)} dispatches, GPU busy {tot/1e6:.2f} ms of {span:.2f} ms")
print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'%':>6}  kernel")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[0]:8d} {v[1]/1e6:10.3f} {v[1]/v[0]/1e3:9.2f} {100*v[1]/tot:6.2f}  {n[:120]}")
