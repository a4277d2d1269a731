#!/usr/bin/env python3
"""One-shape GEMM timing probe: python tools/gemm_probe.py M N K la lb [batch]   (dense operands, batch-strided)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
from galerkin_transformer import _hip as H
M, N, K, la, lb = map(int, sys.argv[1:6])
nb = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda:0")
A = torch.randn((nb, M, K) if la == 0 else (nb, K, M), device=dev)
B = torch.randn((nb, N, K) if lb == 0 else (nb, K, N), device=dev)
C = torch.empty(nb, M, N, device=dev)
sk = int(os.environ.get("SPLITK", 0 if la == 1 else 1))
f = lambda: H.gemm(A, B, C, M, N, K, layout_a=la, layout_b=lb, lda=A.shape[2], ldb=B.shape[2], ldc=N, split_k=sk,
                   batch=(nb, 1), a_bs=(A.shape[1] * A.shape[2], 0), b_bs=(B.shape[1] * B.shape[2], 0), c_bs=(M * N, 0))
f(); torch.cuda.synchronize()
os.environ.pop("GT_GEMM_DEBUG", None)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
env = {k: v for k, v in os.environ.items() if k.startswith("GT_GEMM") or k == "SPLITK"}
print(f"M={M} N={N} K={K} la={la} lb={lb} batch={nb} {env} plan={H.gemm_plan(M, N, K, (nb, 1), sk)}: "
      f"{2.0*M*N*K*nb/t/1e12:6.1f} TF  {t*1e6:7.1f} us")
