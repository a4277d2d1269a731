"""Time the per-grid-line DFT stages (gt_dft_*) against the same products on gt_gemm.  usage: dft_probe.py [B] [n]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
from galerkin_transformer import _hip as H
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 141
m, C = 12, 32
nb, P = B * n, 2 * m
F = torch.randn(n, P, device=dev); X = torch.randn(nb, n, C, device=dev); Y = torch.empty(nb, P, C, device=dev)
Z = torch.randn(nb, P, C, device=dev); W2 = torch.randn(C, C, device=dev); bias = torch.randn(C, device=dev)
out = torch.empty(nb, n, C, device=dev); pre = torch.empty(nb, n, C, device=dev)
def t(fn, k=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / k * 1e3
ba = 4.0 * nb * (n + P) * C
bs = 4.0 * nb * (n * C * 3 + P * C)
for name, fn, by in (
    ("analysis  dft ", lambda: H.dft_analysis(F, X, Y, nb, n, P, C), ba),
    ("analysis  gemm", lambda: H.gemm(F, X, Y, P, C, n, layout_a=1, layout_b=1, lda=P, ldb=C, ldc=C, batch=(nb, 1),
                                      b_bs=(n * C, 0), c_bs=(P * C, 0)), ba),
    ("synthesis dft ", lambda: H.dft_synthesis(F, Z, out, nb, n, P, C, X, W2, C, bias=bias, act=2, pre=pre), bs),
    ("synthesis gemm", lambda: H.gemm(F, Z, out, n, C, P, layout_b=1, lda=P, ldb=C, ldc=C, batch=(nb, 1),
                                      b_bs=(P * C, 0), c_bs=(n * C, 0), bias=bias, act=2, pre=pre, ldpre=C,
                                      K2=C, A2=X, lda2=C, a2_bs=(n * C, 0), B2=W2, ldb2=C), bs),
    ("synth-bwd dft ", lambda: H.dft_synthesis(F, Z, out, nb, n, P, C, X, W2, C), bs - 4.0 * nb * n * C)):
    us = t(fn)
    print(f"{name} B={B} n={n}: {us:8.1f} us   {by / us * 1e-6:6.2f} TB/s")
