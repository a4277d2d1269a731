#!/usr/bin/env python3
"""One-shape GEMM timing probe: python tools/gemm_probe.py M N K la lb [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
from galerkin_transformer import _hip as H
M, N, K, la, lb = map(int, sys.argv[1:6])
dev = torch.device("cuda:0")
A = torch.randn((M, K) if la == 0 else (K, M), device=dev)
B = torch.randn((N, K) if lb == 0 else (K, N), device=dev)
C = torch.empty(M, N, device=dev)
f = lambda: H.gemm(A, B, C, M, N, K, layout_a=la, layout_b=lb, lda=A.shape[1], ldb=B.shape[1], ldc=N, split_k=0 if la == 1 else 1)
f(); torch.cuda.synchronize()
os.environ.pop("GT_GEMM_DEBUG", None)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
print(f"M={M} N={N} K={K} la={la} lb={lb} blocks={os.environ.get('GT_GEMM_BLOCKS')} stream={os.environ.get('GT_GEMM_STREAM')}: {2.0*M*N*K/t/1e12:6.1f} TF  {t*1e6:7.1f} us")
