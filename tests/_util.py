"""Shared helpers for the test-suite (fixture loading, tolerances)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# parity bar from BASELINE.json north_star: 1e-5 relative L2 (fp32 mode)
TOL = 1e-5


class Golden:
    """One tests/golden/<name>.npz fixture (see tests/golden/make_golden.py)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.meta = json.loads(bytes(z["meta"]).decode())
        base = np.load(os.path.join(GOLDEN, self.meta["base"] + ".npz")) if "base" in self.meta else z
        t = lambda a: torch.from_numpy(np.array(a))
        self.sd = {k[3:]: t(base[k]) for k in base.files if k.startswith("sd/")}
        self.inputs = {k[3:]: t(base[k]) for k in base.files if k.startswith("in/")}
        self.out = t(z["out"])
        self.cot = t(z["cot"])
        self.din = {k[4:]: t(z[k]) for k in z.files if k.startswith("din/")}
        self.dparam = {k[7:]: t(z[k]) for k in z.files if k.startswith("dparam/")}
        self.masks = [t(z[f"mask/{i}"]) for i in range(len([k for k in z.files if k.startswith("mask/")]))]


def all_golden(prefix=""):
    """Module fixtures of make_golden.py (the host_* files of make_golden_host.py have their own tests)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and f.startswith(prefix) and not f.startswith("host_"))


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)
