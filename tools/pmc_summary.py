#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; units of 1 KiB... see below).
usage: pmc_summary.py <dir containing FETCH_SIZE/ and WRITE_SIZE/>
Per (kernel name, grid size): average counter value per dispatch over the whole run (the eager steps are
identical).  WRITE_SIZE is in KiB and was checked exact on this box (resize forward writing 651.5 MB reports
636192 KiB).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by
exactly 2x, so the read column is shown raw and doubled."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
def load(counter):
    f = glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return None
    rows = list(csv.DictReader(open(f[0])))
    return rows
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = load(counter)
    if rows is None:
        print("missing", counter); continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:                      # every eager step is identical: average over all dispatches
        if r["Counter_Name"] != counter:
            continue
        key = (r["Kernel_Name"][:100], r.get("Grid_Size", ""))
        a = agg[key]; a[0] += 1; a[1] += float(r["Counter_Value"])
    out[counter] = agg
keys = set()
for a in out.values():
    keys |= set(a)
tot = lambda k: sum(out[c][k][1] for c in out if k in out[c])
print(f"{'calls':>6} {'read_MB(raw)':>13} {'read_MB(x2)':>12} {'write_MB':>10}  kernel  grid")
for k in sorted(keys, key=lambda k: -tot(k))[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    f = out.get("FETCH_SIZE", {}).get(k, [0, 0.0]); w = out.get("WRITE_SIZE", {}).get(k, [0, 0.0])
    rf = f[1] / max(f[0], 1) * 1024 / 1e6
    print(f"{max(f[0], w[0]):6d} {rf:13.1f} {2*rf:12.1f} {w[1]/max(w[0],1)*1024/1e6:10.1f}  {k[0]}  {k[1]}")
