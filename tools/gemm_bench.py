#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM engine at the hot-path shapes (GPU box only).
Prints TFLOP/s per shape next to torch.matmul (rocBLAS/hipBLASLt fp32) as a yardstick."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import torch
from galerkin_transformer import _hip as H


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = torch.device("cuda:0")
    B = int(os.environ.get("BATCH", 16))
    n = 1849
    T = B * n
    shapes = [
        ("qkv   NT", T, 384, 128, 0, 0),
        ("ffn1  NT", T, 256, 128, 0, 0),
        ("ffn2  NT", T, 128, 256, 0, 0),
        ("dx    NN", T, 128, 384, 0, 1),
        ("dW    TN", 384, 128, T, 1, 1),
        ("dWffn TN", 256, 128, T, 1, 1),
        ("big   NT", 8192, 8192, 1024, 0, 0),
    ]
    for name, M, N, K, la, lb in shapes:
        A = torch.randn((M, K) if la == 0 else (K, M), device=dev)
        Bm = torch.randn((N, K) if lb == 0 else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        fl_ = 2.0 * M * N * K
        split = 0 if la == 1 else 1
        f = lambda: H.gemm(A, Bm, C, M, N, K, layout_a=la, layout_b=lb, lda=A.shape[1], ldb=Bm.shape[1],
                           ldc=N, split_k=split)
        t = timeit(f)
        sweep = ""
        if os.environ.get("SWEEP"):
            for bk in (16, 32):
                os.environ["GT_GEMM_BK"] = str(bk)
                sweep += f"\n      bk{bk}:"
                for c in range(5):
                    os.environ["GT_GEMM_CFG"] = str(c)
                    tg = (256, 512, 1024) if la == 1 else (512,)
                    for t_ in tg:
                        os.environ["GT_GEMM_TARGET"] = str(t_)
                        sweep += f" c{c}" + (f"/t{t_}" if la == 1 else "") + f":{fl_ / timeit(f) / 1e12:5.1f}"
            for k_ in ("GT_GEMM_CFG", "GT_GEMM_BK", "GT_GEMM_TARGET"):
                del os.environ[k_]
        a2 = A if la == 0 else A.t()
        b2 = Bm.t() if lb == 0 else Bm
        t2 = 1.0 if os.environ.get("NOTORCH") else timeit(lambda: torch.matmul(a2, b2))
        fl = 2.0 * M * N * K
        print(f"{name}  M={M:7d} N={N:5d} K={K:6d}  plan={H.gemm_plan(M, N, K, split_k=split)}  "
              f"gt {fl / t / 1e12:7.2f} TF/s ({t * 1e6:8.1f} us)   torch {fl / t2 / 1e12:7.2f} TF/s "
              f"({t2 * 1e6:8.1f} us)" + sweep, flush=True)


if __name__ == "__main__":
    main()
