#!/usr/bin/env python3
"""The accuracy leg over many dropout seeds: runs tools/accuracy_leg.run() for `--n` dropout seeds
(same data, same initial weights, a different realisation of the training noise per seed) and writes every run plus
mean / standard error of the final validation rel-L2.  --impl reference (build container, CPU) is committed as
profiles/accuracy_reference_cpu_seeds.json; --impl hip runs on cuda:0 (bench.py quotes both and a Welch t-test)."""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import accuracy_leg as AL


def summarize(vals):
    n = len(vals)
    m = sum(vals) / n
    var = sum((v - m) ** 2 for v in vals) / max(n - 1, 1)
    return dict(n=n, mean=m, std=math.sqrt(var), sem=math.sqrt(var / n), values=vals)


def welch(a, b):
    """Welch's t statistic, degrees of freedom and the two-sided p-value (normal tail for df > 30, else a t series)."""
    va, vb = a["std"] ** 2 / a["n"], b["std"] ** 2 / b["n"]
    t = (a["mean"] - b["mean"]) / math.sqrt(va + vb)
    df = (va + vb) ** 2 / (va ** 2 / (a["n"] - 1) + vb ** 2 / (b["n"] - 1))
    try:
        from scipy import stats
        p = float(2 * stats.t.sf(abs(t), df))
    except Exception:
        p = float(math.erfc(abs(t) / math.sqrt(2)))
    return dict(t=t, df=df, p_two_sided=p)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", default="hip", choices=["hip", "reference"])
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--first-seed", type=int, default=1000)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    runs = []
    for i in range(a.n):
        r = AL.run(a.impl, dropout_seed=a.first_seed + i, log=None)
        r.pop("history", None)
        runs.append(r)
        print(json.dumps({"seed": r["dropout_seed"], "val_rel_l2": r["val_rel_l2"], "seconds": r["seconds"]}), flush=True)
        with open(a.out, "w") as f:
            json.dump({"impl": a.impl, "summary": summarize([x["val_rel_l2"] for x in runs]), "runs": runs}, f, indent=1)
