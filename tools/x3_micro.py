#!/usr/bin/env python3
"""Split-operand token GEMMs in isolation at the ex2 B = 128 shapes (T = 128 * 43 * 43 rows): microseconds, useful
TFLOP/s and algorithmic GB/s per launch, rotating over three buffer sets so the 256 MB infinity cache does not flatter
the numbers.  GT_HIP_LIB selects a library variant (tools/ablate_x3.sh builds: the ablations give wrong results, only the
time is read).

    python tools/x3_micro.py [B]
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
    shapes = [("plain   N128 K128", 128, 128, 0, {}), ("bias+res N128 K128", 128, 128, 0, dict(bias=1, res=1)),
              ("silu+pre N256 K128", 256, 128, 0, dict(bias=1, act=1, pre=1)), ("plain   N128 K256", 128, 256, 0, {}),
              ("dgrad   N128 K384", 128, 384, 1, {}), ("dgrad+aux N128 K256", 128, 256, 1, dict(aux=1)),
              ("plain   N384 K128", 384, 128, 0, dict(bias=1)),
              ("A^T     N128 K128", 128, 128, 0, dict(la=1)), ("A^T     N128 K384", 128, 384, 0, dict(la=1))]
    out = {"lib": os.environ.get("GT_HIP_LIB", "libgt_hip.so"), "T": T}
    for name, N, K, lb, ep in shapes:
        sets = []
        for r in range(3):
            A = torch.randn((K, T) if ep.get("la") else (T, K), device=dev)
            Bm = torch.randn((N, K) if lb == 0 else (K, N), device=dev) * 0.1
            C = torch.empty(T, N, device=dev)
            kw = {}
            if ep.get("bias"):
                kw["bias"] = torch.randn(N, device=dev)
            if ep.get("res"):
                kw.update(res=torch.randn(T, N, device=dev), ldr=N)
            if ep.get("pre"):
                kw.update(pre=torch.empty(T, N, device=dev), ldpre=N)
            if ep.get("act"):
                kw["act"] = H.ACT_SILU
            if ep.get("aux"):
                kw.update(aux_op=H.AUX_DSILU, aux=torch.randn(T, N, device=dev), ldaux=N)
            sets.append((A, Bm, C, kw))
        fn = lambda i: H.gemm(sets[i % 3][0], sets[i % 3][1], sets[i % 3][2], T, N, K, layout_a=int(bool(ep.get("la"))),
                              layout_b=lb, lda=(T if ep.get("la") else K),
                              ldb=sets[i % 3][1].shape[1], ldc=N, **sets[i % 3][3])
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        nbytes = 4.0 * T * (K + N * (1 + sum(bool(ep.get(k)) for k in ("res", "pre", "aux"))))
        out[name] = dict(us=round(us, 1), tflops=round(2.0 * T * N * K / us / 1e6, 1), gbs=round(nbytes / us / 1e3, 0))
        del sets
    print(json.dumps(out))


if __name__ == "__main__":
    main()
