#!/usr/bin/env python3
"""Steady-state per-kernel summary from a rocprofv3 (rocpd sqlite) kernel trace.
usage: prof_summary.py results.db n_steps step_ms [top]"""
import sqlite3, sys
from collections import defaultdict
db, nsteps, step_ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 30
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
t_end = rows[-1][2]
win = nsteps * step_ms * 1e6
sel = [r for r in rows if r[1] > t_end - win]
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in sel:
    agg[n][0] += 1
    agg[n][1] += (e - s)
tot = sum(v[1] for v in agg.values())
print(f"window {win/1e6:.1f} ms, {len(sel)} dispatches, busy {tot/1e6:.2f} ms = {tot/1e6/nsteps:.3f} ms/step")
print(f"{'ms/step':>9} {'%':>6} {'calls/step':>10} {'avg us':>9}  kernel")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1]/1e6/nsteps:9.3f} {100*v[1]/tot:6.1f} {v[0]/nsteps:10.1f} {v[1]/v[0]/1e3:9.1f}  {n[:110]}")
