"""Time the tall-skinny weight-gradient products with and without the tsmm path.  usage: tsmm_probe.py [K]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
from galerkin_transformer import _hip as H
dev = torch.device("cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2544768
def t(fn, k=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / k * 1e3
for M, N in ((32, 2), (32, 32), (128, 32)):
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); Cc = torch.empty(M, N, device=dev)
    cs = torch.empty(M, device=dev)
    us = t(lambda: H.gemm(A, B, Cc, M, N, K, layout_a=1, layout_b=1, lda=M, ldb=N, ldc=N, split_k=0, a_colsum=cs))
    nm = H.gemm_kernel_name(A, B, M, N, K, layout_a=1, layout_b=1, lda=M, ldb=N, ldc=N, split_k=0)
    print(f"M={M} N={N} K={K}: {us:8.1f} us  {4.0 * K * (M + N) / us * 1e-6:5.2f} TB/s  {nm[:60]}")
