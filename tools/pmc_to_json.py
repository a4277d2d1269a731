#!/usr/bin/env python3
"""rocprofv3 --pmc passes of one eager training step (tools/gpu_measure.sh) -> per-kernel JSON.

    pmc_to_json.py <dir with FETCH_SIZE/ WRITE_SIZE/ [SQ/]> <out.json>

Per kernel symbol (all dispatches of the steady eager steps averaged): read_bytes (FETCH_SIZE x 1 KiB, DOUBLED -- the
gfx950 correction of MI355X_MICROARCH.md, HBM section: the counter tallies 128-byte requests at 64 bytes), write_bytes
(WRITE_SIZE x 1 KiB, exact), and from the SQ pass mfma_busy_cycles (SQ_VALU_MFMA_BUSY_CYCLES, summed over the SIMDs)
with the dispatch's duration in the same (profiled) pass -> mfma_util = busy / (1024 SIMDs x duration x 2.4 GHz)."""
import csv, glob, json, os, sys
from collections import defaultdict

d, out_path = sys.argv[1], sys.argv[2]


def rows(sub, name):
    f = glob.glob(os.path.join(d, sub, "**", name), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def short(n):
    return n.replace("void ", "").split("(")[0][:96]


res = defaultdict(dict)
by_grid = defaultdict(dict)          # "symbol|grid size": one launch geometry of a symbol (a template instance serves many)
for counter, key, scale in (("FETCH_SIZE", "read_bytes", 2048.0), ("WRITE_SIZE", "write_bytes", 1024.0)):
    agg = defaultdict(lambda: [0, 0.0])
    aggg = defaultdict(lambda: [0, 0.0])
    for r in rows(counter, "*counter_collection.csv"):
        if r["Counter_Name"] == counter:
            for a in (agg[short(r["Kernel_Name"])], aggg[short(r["Kernel_Name"]) + "|" + r.get("Grid_Size", "")]):
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        res[k][key] = v / n * scale
        res[k]["calls_seen"] = n
    for k, (n, v) in aggg.items():
        by_grid[k][key] = v / n * scale
        by_grid[k]["calls_seen"] = n
agg = defaultdict(lambda: [0, 0.0])
disp = {}
for r in rows("SQ", "*counter_collection.csv"):
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
dur = defaultdict(lambda: [0, 0.0])
for r in rows("SQ", "*kernel_trace.csv"):
    a = dur[short(r["Kernel_Name"])]
    a[0] += 1
    a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, (n, v) in agg.items():
    res[k]["mfma_busy_cycles"] = v / n
    if k in dur and dur[k][0]:
        ns = dur[k][1] / dur[k][0]
        res[k]["profiled_duration_us"] = ns / 1e3
        res[k]["mfma_util_at_2.4GHz"] = (v / n) / (1024.0 * ns * 2.4)
meta = {"_source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES, separate passes, --kernel-trace only, "
                   "over `bench.py --steps 2 --warmup 1 --no-graph` (tools/gpu_measure.sh); FETCH_SIZE doubled (gfx950)"}
meta["_by_grid"] = {k: v for k, v in sorted(by_grid.items()) if v.get("read_bytes", 0) + v.get("write_bytes", 0) > 32e6}
json.dump({**meta, **dict(sorted(res.items(), key=lambda kv: -(kv[1].get('read_bytes', 0) + kv[1].get('write_bytes', 0))))},
          open(out_path, "w"), indent=1)
print(f"{len(res)} kernels -> {out_path}")
