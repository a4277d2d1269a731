#!/usr/bin/env python3
"""Accuracy of the split-operand arithmetic under cancellation: C = A B^T with A = offset + randn (a large common component
that the zero-mean B cancels), error against float64 relative to ||C|| and to the un-cancelled scale sum|a||b|, for the
bf16x3 / f32 kernels, token-row (K = 128) and token-contracted (K = T, split-K) shapes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
from galerkin_transformer import _hip as H

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
out = []
for offset in (0.0, 10.0, 100.0):
    for (M, N, K, la, lb, sk) in ((33282, 128, 128, 0, 0, 1), (256, 128, 33282, 1, 1, 0)):
        A = (torch.randn(M, K, generator=g) + offset)
        B = torch.randn(N, K, generator=g)
        ref = A.double() @ B.double().t()
        scale = (A.double().abs() @ B.double().abs().t())
        Ad = (A if la == 0 else A.t().contiguous()).to(dev)
        Bd = (B if lb == 0 else B.t().contiguous()).to(dev)
        rec = {"offset": offset, "MNK": [M, N, K]}
        for prec in ("f32", "bf16x3"):
            C = torch.empty(M, N, device=dev)
            H.gemm(Ad, Bd, C, M, N, K, layout_a=la, layout_b=lb, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, split_k=sk,
                   precision=prec)
            torch.cuda.synchronize()
            d = C.double().cpu() - ref
            rec[prec] = {"rel_l2": float(d.norm() / ref.norm()), "max_err_over_sum_abs": float((d.abs() / scale).max()),
                         "rms_err_over_sum_abs": float((d / scale).pow(2).mean().sqrt())}
        out.append(rec)
        print(json.dumps(rec))
