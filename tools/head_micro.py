#!/usr/bin/env python3
"""gt_mlp_head_fwd / gt_mlp_head_bwd in isolation at the ex2 B = 128 size (T = 128 * 141 * 141 grid points, 32 -> 128 -> 1,
SiLU), in the two arithmetics of the entry points (f16x2: two-term fp16 kernels; f32: fp32-MFMA kernels), rotating over
three input sets; errors against fp64 on a 64k-row prefix.

    python tools/head_micro.py [B]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import torch
from galerkin_transformer import _hip as H


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    T = B * 141 * 141
    w1 = torch.randn(128, 32, device=dev) * 0.2
    b1 = torch.randn(128, device=dev) * 0.3
    w2 = torch.randn(1, 128, device=dev) * 0.2
    b2 = torch.randn(1, device=dev)
    sets = [(torch.randn(T, 32, device=dev), torch.randn(T, 1, device=dev) * 1e-6) for _ in range(3)]
    out = torch.empty(T, 1, device=dev)
    dx = torch.empty(T, 32, device=dev)
    dw1, db1, dw2, db2 = (torch.empty(s, device=dev) for s in ((128, 32), (128,), (1, 128), (1,)))
    res = {"T": T}
    n = 65536
    xd, gd = sets[0][0][:n].double().requires_grad_(True), sets[0][1][:n].double()
    w1d, b1d, w2d, b2d = (t.double().requires_grad_(True) for t in (w1, b1, w2, b2))
    ref = torch.nn.functional.linear(torch.nn.functional.silu(torch.nn.functional.linear(xd, w1d, b1d)), w2d, b2d)
    gref = torch.autograd.grad(ref, (xd, w1d, b1d, w2d, b2d), gd)
    for prec in ("f32", "f16x2"):
        def fwd(i):
            H.mlp_head_fwd(sets[i % 3][0], w1, b1, w2, b2, H.ACT_SILU, out, precision=prec)

        def bwd(i):
            H.mlp_head_bwd(sets[i % 3][0], w1, b1, w2, H.ACT_SILU, sets[i % 3][1], dx, dw1, db1, dw2, db2, precision=prec)
        r = {}
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(30):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            r[name + "_us"] = round(e0.elapsed_time(e1) / 30 * 1e3, 1)
        o = torch.empty(n, 1, device=dev)
        H.mlp_head_fwd(sets[0][0][:n], w1, b1, w2, b2, H.ACT_SILU, o, precision=prec)
        dxs = torch.empty(n, 32, device=dev)
        H.mlp_head_bwd(sets[0][0][:n], w1, b1, w2, H.ACT_SILU, sets[0][1][:n].contiguous(), dxs, dw1, db1, dw2, db2, precision=prec)
        torch.cuda.synchronize()
        r["err"] = {k: float("%.2e" % rel(a, b)) for k, a, b in zip(("out", "dx", "dw1", "db1", "dw2", "db2"),
                                                                    (o, dxs, dw1, db1, dw2, db2), (ref,) + tuple(gref))}
        res[prec] = r
    print(json.dumps(res))


if __name__ == "__main__":
    main()
