#!/usr/bin/env python3
"""Where a block of gemm_x3w_kernel spends its time (profiling build `-DGT_X3W_PROF`: wave 0 of every block accumulates
shader-clock cycles per phase of a stage):

    python -c "import sys; sys.path.insert(0, 'galerkin-transformer_amd'); import build; \
               build.build(tag='_x3wprof', defines=['GT_X3W_PROF'], only=['gt_gemm_x3.hip'])"
    GT_HIP_LIB=libgt_hip_x3wprof.so python tools/x3w_prof.py

phases: 0 values of the stage have arrived (vmcnt wait) + amax | 1 barrier | 2 exponents, split, plane stores | 3 request of
stage s + PF, barrier | 4 fragment reads + MFMA issue.  Cycles per stage (median over blocks), and the share of a block's life.
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import numpy as np
import torch
from galerkin_transformer import _hip as H


def main():
    dev = torch.device("cuda:0")
    T = 128 * 43 * 43
    L = H.lib()
    L.gt_debug_x3w_prof.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    L.gt_debug_x3w_prof.restype = ctypes.c_int
    out = {"PF": os.environ.get("GT_X3W_PF", "2"), "MAP": os.environ.get("GT_X3W_MAP", "2")}
    for M, N in ((128, 128), (128, 256), (384, 128)):
        A, B, C = torch.randn(T, M, device=dev), torch.randn(T, N, device=dev), torch.empty(M, N, device=dev)
        for _ in range(3):
            H.gemm(A, B, C, M, N, T, layout_a=1, layout_b=1, lda=M, ldb=N, ldc=N, split_k=0)
        torch.cuda.synchronize()
        buf = np.zeros((4096, 8), dtype=np.uint64)
        assert L.gt_debug_x3w_prof(buf.ctypes.data, buf.nbytes) == 0
        b = buf[buf[:, 6] > 0].astype(np.float64)
        stages = b[:, 6]
        per = b[:, :5] / stages[:, None]
        life = b[:, 5]
        span = (b[:, 7].max() - b[:, 7].min()) * 0.01
        out[f"{M}x{N}"] = {"blocks": int(len(b)), "stages_per_block": float(np.median(stages)),
                           "cycles_per_stage_median": [round(float(v)) for v in np.median(per, axis=0)],
                           "share_of_block_life": [round(float(v), 3) for v in (b[:, :5].sum(0) / life.sum())],
                           "block_life_cycles_median": round(float(np.median(life))),
                           "last_block_end_minus_first_block_end_us": round(float(span), 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
