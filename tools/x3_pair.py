#!/usr/bin/env python3
"""How much do two token GEMMs overlap when they run side by side?  Times the same product on two disjoint buffer sets,
back to back on one stream and concurrently on two streams (what a kernel that overlapped its own load and store phases
could at best approach).   python tools/x3_pair.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import torch
from galerkin_transformer import _hip as H


def main():
    dev = torch.device("cuda:0")
    T = 128 * 43 * 43
    out = {}
    for name, N, K, lb in (("N128 K128", 128, 128, 0), ("N128 K384 dgrad", 128, 384, 1), ("N256 K128", 256, 128, 0)):
        sets = []
        for r in range(4):
            A = torch.randn(T, K, device=dev)
            Bm = torch.randn((N, K) if lb == 0 else (K, N), device=dev) * 0.1
            sets.append((A, Bm, torch.empty(T, N, device=dev)))
        run = lambda i: H.gemm(sets[i][0], sets[i][1], sets[i][2], T, N, K, layout_b=lb, lda=K, ldb=sets[i][1].shape[1], ldc=N)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for i in range(4):
            run(i)
        with torch.cuda.stream(s1):
            run(0)
        with torch.cuda.stream(s2):
            run(1)
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            run(2 * (r & 1)); run(2 * (r & 1) + 1)
        e1.record()
        torch.cuda.synchronize()
        seq = e0.elapsed_time(e1) / reps * 1e3
        cur = torch.cuda.current_stream()
        e0.record()
        for r in range(reps):
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                run(2 * (r & 1))
            with torch.cuda.stream(s2):
                run(2 * (r & 1) + 1)
            cur.wait_stream(s1); cur.wait_stream(s2)
        e1.record()
        torch.cuda.synchronize()
        par = e0.elapsed_time(e1) / reps * 1e3
        out[name] = dict(two_sequential_us=round(seq, 1), two_concurrent_us=round(par, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
