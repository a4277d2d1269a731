#!/usr/bin/env python3
"""Signed error of the split-operand product against float64 -- bias or random walk?  

C = A B^T with same-sign operands (|randn|: every partial sum grows monotonically, so a rounding that truncates shows up
as a NEGATIVE mean relative error growing with the chain length) and with random signs (mean signed error relative to
sum |a||b|), for the bf16x3 and f32 kernels, on token-row shapes (K = 128 / 384: chains of 8 / 24 stages) and on the
token-contracted weight-gradient shape (K = tokens; split_k = 0 is the library's own K split, split_k = 1 one long chain).
Prints one JSON line per case."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
from galerkin_transformer import _hip as H

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
CASES = (  # M, N, K, layout_a, layout_b, split_k
    (33282, 128, 128, 0, 0, 1), (33282, 128, 384, 0, 0, 1), (16641, 256, 1152, 0, 0, 1),
    (256, 128, 33282, 1, 1, 0), (256, 128, 33282, 1, 1, 1), (256, 128, 236672, 1, 1, 0), (256, 128, 236672, 1, 1, 1))
precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f32", "bf16x3"]
for signs in ("positive", "random"):
    for (M, N, K, la, lb, sk) in CASES:
        A = torch.randn(M, K, generator=g)
        B = torch.randn(N, K, generator=g)
        if signs == "positive":
            A, B = A.abs(), B.abs()
        Ad = (A if la == 0 else A.t().contiguous()).to(dev)
        Bd = (B if lb == 0 else B.t().contiguous()).to(dev)
        ref = (Ad.double() @ Bd.double().t()) if la == 0 else (Ad.double().t() @ Bd.double())
        scale = (Ad.double().abs() @ Bd.double().abs().t()) if la == 0 else (Ad.double().abs().t() @ Bd.double().abs())
        rec = {"signs": signs, "MNK": [M, N, K], "split_k": sk}
        for prec in precs:
            C = torch.empty(M, N, device=dev)
            H.gemm(Ad, Bd, C, M, N, K, layout_a=la, layout_b=lb, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, split_k=sk,
                   precision=prec)
            torch.cuda.synchronize()
            e = (C.double() - ref) / scale
            rec[prec] = {"mean_signed": float(e.mean()), "rms": float(e.pow(2).mean().sqrt()), "max_abs": float(e.abs().max()),
                         "rel_l2": float((C.double() - ref).norm() / ref.norm())}
            if prec != "f32":
                rec[prec]["kernel"] = H.gemm_kernel_name(Ad, Bd, M, N, K, layout_a=la, layout_b=lb, lda=Ad.shape[1],
                                                         ldb=Bd.shape[1], ldc=N, split_k=sk, precision=prec)
        del A, B, Ad, Bd, ref, scale
        print(json.dumps(rec), flush=True)
