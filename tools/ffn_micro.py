#!/usr/bin/env python3
"""Round 6: FFN forward at the C2 B = 128 shape -- two packed-B launches vs gt_ffn_fwd, with and without the dropout masks
(how much of a token GEMM's time is the stateless RNG of its epilogue?).  python tools/ffn_micro.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "galerkin-transformer_amd"))
import torch
from galerkin_transformer import _hip as H

dev = torch.device("cuda:0")
T, d, f = 236672, 128, 256
x = torch.randn(T, d, device=dev)
w1, b1 = torch.randn(f, d, device=dev) * 0.1, torch.randn(f, device=dev) * 0.1
w2, b2 = torch.randn(d, f, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1
hid, out = torch.empty(T, f, device=dev), torch.empty(T, d, device=dev)
H.set_seed(1, dev)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for p in (0.05, 0.0):
    dh = H.dropout_desc(p, 500, dev) if p > 0 else None
    do = H.dropout_desc(p, 501, dev) if p > 0 else None
    t1 = timeit(lambda: H.gemm(x, w1, hid, T, f, d, lda=d, ldb=d, ldc=f, bias=b1, act=H.ACT_RELU, drop=dh))
    t2 = timeit(lambda: H.gemm(hid, w2, out, T, d, f, lda=f, ldb=f, ldc=d, bias=b2, drop=do, res=x, ldr=d))
    tf = timeit(lambda: H.ffn_fwd(x, w1, b1, w2, b2, x, dh, do, H.ACT_RELU, hid, out))
    print(f"p={p}: FFN1 {t1:.1f} us  FFN2 {t2:.1f} us  sum {t1 + t2:.1f}  fused {tf:.1f} us")
t0 = timeit(lambda: H.gemm(x, w1, hid, T, f, d, lda=d, ldb=d, ldc=f))
print(f"FFN1 shape, no epilogue at all: {t0:.1f} us")
tc = timeit(lambda: hid.copy_(hid))
print(f"[T,256] copy (242 MB r + 242 MB w): {tc:.1f} us")
