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


class replay_audit:
    """``with replay_audit() as a: <float64 oracle run with replayed ReLU decisions>`` then ``a.check()``: the decisions the
    device run handed to the checker may differ from the float64 oracle's own ``pre > 0`` on at most ``max_frac`` of a gate's
    elements, and only where ``|pre| < MAX_REL * rms(pre)`` -- i.e. only where the pre-activation lies within rounding of
    the kink.  A systematically wrong mask (a transposed tile, a stale buffer) fails here instead of being replayed into
    the reference."""

    # |pre| / rms(pre) below which a replayed decision may differ.  FeedForward gates: 1e-5 (fp32-class products on both
    # sides).  Down-scaler gates: 1e-4 -- F.interpolate(align_corners=True) evaluates its source coordinates in the tensor's
    # dtype, so every float32 evaluation of the first resize (the reference's own included) sits ~1e-4 relative from the
    # float64 one (measured on the replay-relu run: 4 + 7 decisions of 9.4 M, |pre| <= 1.7e-5 rms), and the
    # chain's pre-activations inherit that offset.
    MAX_REL = {"ff": 1e-5, "scaler": 1e-4}

    def __init__(self, max_frac=1e-5, max_rel=None):
        self.max_frac, self.records = max_frac, []
        self.max_rel = dict(self.MAX_REL) if max_rel is None else {"ff": max_rel, "scaler": max_rel}

    def __enter__(self):
        from oracle import galerkin_oracle as O
        O.set_replay_audit(self.records)
        return self

    def __exit__(self, *exc):
        from oracle import galerkin_oracle as O
        O.set_replay_audit(None)

    def summary(self):
        return {"gates": len(self.records), "elements": sum(r["n"] for r in self.records),
                "flipped": sum(r["flipped"] for r in self.records),
                "worst_frac": max((r["flipped"] / r["n"] for r in self.records), default=0.0),
                "worst_rel": max((r["max_rel"] for r in self.records), default=0.0)}

    def check(self, min_gates=1):
        assert len(self.records) >= min_gates, (len(self.records), min_gates)
        bad = [r for r in self.records
               if r["flipped"] > self.max_frac * r["n"] + 1 or r["max_rel"] > self.max_rel[r["where"].split(".")[0]]]
        assert not bad, bad[:8]
        return self.summary()


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


def spectral_conv2d_dft_math(x, wlin, blin, w0, w1, modes, act=torch.nn.functional.silu):
    """The S1..S4 truncated-DFT pipeline of galerkin_transformer/spectral.py restated in plain torch einsums (any
    device / dtype) on the product's own bases (spectral.dft_bases): pins the formulation against torch.fft on the CPU."""
    from galerkin_transformer.spectral import dft_bases

    B, n, _, C = x.shape
    m = modes
    F1, G2, G3, F4 = (t.to(x.dtype).to(x.device) for t in dft_bases(n, m))
    X1 = torch.einsum("yk,bxyc->bxkc", F1, x)                                   # [B,n,2m,C]
    X2 = torch.einsum("rq,brkc->bqkc", G2, X1.reshape(B, 2 * n, m, C))          # [B,4m,m,C]
    X2 = X2.reshape(B, 2, 2 * m * m, C)
    xc = torch.complex(X2[:, 0], X2[:, 1])
    W = torch.cat([w0, w1], dim=2).reshape(w0.shape[0], w0.shape[1], 2 * m * m, 2)
    yc = torch.einsum("bqi,ioq->bqo", xc, torch.complex(W[..., 0], W[..., 1]))
    Y = torch.stack([yc.real, yc.imag], 1)                                      # [B,2,2m*m,Co]
    Co = Y.shape[-1]
    Z = torch.einsum("rq,bqkc->brkc", G3, Y.reshape(B, 4 * m, m, Co))           # [B,2n,m,Co]
    Z = Z.reshape(B, n, 2 * m, Co)
    out = torch.einsum("yk,bxko->bxyo", F4, Z)
    return act(out + torch.nn.functional.linear(x, wlin, blin))
