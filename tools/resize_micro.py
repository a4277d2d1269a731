"""Channels-last bilinear resizes of the headline step (C2, B = 128) timed back to back on a cold working set:
the regressor-side resize 77 -> 141 at 32 channels with the affine epilogue (gt_bilinear2d_fwd_affine), its backward, the
up-scaler's 43 -> 77 at 128 channels forward / backward.  python tools/resize_micro.py  (one line per kernel: us, GB/s)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "galerkin-transformer_amd"))
from galerkin_transformer import _hip as H

dev = torch.device("cuda:0")
B = int(os.environ.get("RM_B", "128"))


def timed(fn, reps=20, copies=4):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % copies)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(name, C, Hi, Ho, affine):
    xs = [torch.randn(B, Hi, Hi, C, device=dev) for _ in range(4)]
    bias = torch.randn(C, device=dev) if affine else None
    grid = torch.randn(B, Ho, Ho, 2, device=dev) if affine else None
    w = torch.randn(C, 2, device=dev) if affine else None
    us = timed(lambda i: H.bilinear2d_fwd(xs[i], (Ho, Ho), True, True, H.ACT_NONE, bias=bias, rp_a=grid, rp_b=w, rp_ldb=2))
    nb = 4.0 * B * C * (Hi * Hi + Ho * Ho)
    print(f"{name} fwd  C={C} {Hi}->{Ho}: {us:7.1f} us  {nb / us / 1e3:7.1f} GB/s")
    gs = [torch.randn(B, Ho, Ho, C, device=dev) for _ in range(4)]
    us = timed(lambda i: H.bilinear2d_bwd(gs[i], None, (Hi, Hi), True, True, H.ACT_NONE))
    print(f"{name} bwd  C={C} {Ho}->{Hi}: {us:7.1f} us  {nb / us / 1e3:7.1f} GB/s")


case("regressor", 32, 77, 141, True)
case("up-scaler", 128, 43, 77, False)
case("plain", 32, 77, 141, False)
# a plain copy of the same bytes for scale
a = [torch.randn(B * 141 * 141 * 32, device=dev) for _ in range(4)]
o = torch.empty_like(a[0])
us = timed(lambda i: o.copy_(a[i]))
print(f"copy {a[0].numel() * 8 / 1e6:.0f} MB: {us:7.1f} us  {a[0].numel() * 8 / us / 1e3:7.1f} GB/s")
