#!/usr/bin/env python3
"""Token-contracted weight gradients (gemm_x3w_kernel + its split-K reduce) in isolation at the ex2 B = 128 shapes:
dW[M][N] = A[T][M]^T B[T][N], T = 128 * 43 * 43, rotating over three operand sets.  Environment switches read by the
library: GT_X3W_PF (1 | 2: stages of operand values in flight per thread), GT_X3W_MAP (0 tile-major grid, 1 chunk-major
inside an XCD, 2 = that for two output tiles only).

    python tools/x3w_micro.py [B]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import torch
from galerkin_transformer import _hip as H


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    T = B * 43 * 43
    out = {"PF": os.environ.get("GT_X3W_PF", "2"), "MAP": os.environ.get("GT_X3W_MAP", "2"), "T": T}
    for M, N in ((128, 128), (128, 256), (256, 128), (384, 128)):
        sets = [(torch.randn(T, M, device=dev), torch.randn(T, N, device=dev), torch.empty(M, N, device=dev),
                 torch.empty(M, device=dev)) for _ in range(3)]
        name = H.gemm_kernel_name(sets[0][0], sets[0][1], M, N, T, layout_a=1, layout_b=1, lda=M, ldb=N, ldc=N, split_k=0)
        fn = lambda i: H.gemm(sets[i % 3][0], sets[i % 3][1], sets[i % 3][2], M, N, T, layout_a=1, layout_b=1, lda=M, ldb=N,
                              ldc=N, split_k=0, a_colsum=sets[i % 3][3])
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        ref = sets[0][0].double().t() @ sets[0][1].double()
        err = float((sets[0][2].double() - ref).norm() / ref.norm())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        out[f"{M}x{N}"] = dict(us=round(us, 1), gbs=round(4.0 * T * (M + N) / us / 1e3, 0), rel_err=float("%.2e" % err),
                               kernel=name.replace("(gt::GemmP)", ""))
        del sets
    print(json.dumps(out))


if __name__ == "__main__":
    main()
