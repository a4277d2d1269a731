#!/usr/bin/env python3
"""GPU probe: torch/MIOpen conv and bilinear-interp timings at the Darcy-141 scaler shapes (B=16),
dense NCHW vs channels_last.  Informational (decides which layout the scalers keep)."""
import torch, time, sys
import torch.nn.functional as F
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for cl in (False, True):
    mf = torch.channels_last if cl else torch.contiguous_format
    print(f"--- channels_last={cl}")
    for (ci, co, n) in [(1, 128, 141), (128, 42, 78), (42, 42, 78), (42, 44, 78), (128, 128, 77)]:
        x = torch.randn(B, ci, n, n, device=dev).contiguous(memory_format=mf).requires_grad_(True)
        conv = torch.nn.Conv2d(ci, co, 3, padding=1, bias=False).to(dev).to(memory_format=mf)
        y = conv(x); g = torch.randn_like(y)
        tf = timeit(lambda: conv(x))
        def fb():
            x.grad = None; conv.weight.grad = None
            conv(x).backward(g)
        tfb = timeit(fb)
        print(f"conv {ci:3d}->{co:3d} @{n}^2: fwd {tf:8.1f} us   fwd+bwd {tfb:8.1f} us  out_cl={y.is_contiguous(memory_format=torch.channels_last)}")
    for (c, ni, no) in [(128, 141, 78), (128, 78, 43), (128, 43, 77), (128, 77, 141)]:
        x = torch.randn(B, c, ni, ni, device=dev).contiguous(memory_format=mf).requires_grad_(True)
        y = F.interpolate(x, size=(no, no), mode="bilinear", align_corners=True); g = torch.randn_like(y)
        tf = timeit(lambda: F.interpolate(x, size=(no, no), mode="bilinear", align_corners=True))
        def fb():
            x.grad = None
            F.interpolate(x, size=(no, no), mode="bilinear", align_corners=True).backward(g)
        tfb = timeit(fb)
        mb = B * c * (ni * ni + no * no) * 4 / 1e6
        print(f"interp c={c} {ni}->{no}: fwd {tf:8.1f} us ({mb/tf*1e6/1e6:6.2f} TB/s)  fwd+bwd {tfb:8.1f} us")
x = torch.randn(B, 128, 141, 141, device=dev)
print("permute copy NCHW->NHWC 141^2x128:", timeit(lambda: x.permute(0, 2, 3, 1).contiguous()), "us")
