#!/usr/bin/env python3
"""rocFFT A/B for the spectral block (north_star: "rocFFT for the spectral path"; SURVEY 7 hard part 5).

Times ONE SpectralConv2d (reference layers.py:1153-1197) forward + backward on the GPU, two ways, same input:

  A  rocFFT formulation -- the reference's own arithmetic on the device: Linear, torch.fft.rfft2 (hipFFT / rocFFT
     plans under torch on ROCm), the two corner mode mixes as real einsums, zero-filled half spectrum, torch.fft.irfft2,
     activation;  141 = 3 * 47 and 211 is prime, so rocFFT runs its Bluestein / Rader paths;
  B  this repo's truncated DFT (galerkin_transformer.SpectralConv2d: S1..S4 + gt_modemix, csrc/gt_dft.hip).

    python tools/rocfft_probe.py [B] [n ...]          (default B = 128 for n = 141, B = 8 for n = 211)

Prints one JSON line per grid size; run it under `rocprofv3 --kernel-trace --stats` for the kernel-level view
(profiles/r02_rocfft_probe_*.txt).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import torch
import torch.nn.functional as F


def reference_block(x, lin_w, lin_b, w0, w1, modes, n):
    """layers.py:1172-1189 with torch ops on the device (no reference code is imported: the five lines of math)."""
    B, C = x.shape[0], x.shape[-1]
    res = F.linear(x, lin_w, lin_b)
    xf = torch.fft.rfft2(x.permute(0, 3, 1, 2), s=(n, n), norm="ortho")
    xf = torch.stack([xf.real, xf.imag], dim=-1)

    def cmul(a, b):                                                    # (b,i,x,y,2) x (i,o,x,y,2) -> (b,o,x,y,2)
        op = lambda u, v: torch.einsum("bixy,ioxy->boxy", u, v)
        return torch.stack([op(a[..., 0], b[..., 0]) - op(a[..., 1], b[..., 1]),
                            op(a[..., 1], b[..., 0]) + op(a[..., 0], b[..., 1])], dim=-1)

    out = torch.zeros(B, w0.shape[1], n, n // 2 + 1, 2, device=x.device)
    out[:, :, :modes, :modes] = cmul(xf[:, :, :modes, :modes], w0)
    out[:, :, -modes:, :modes] = cmul(xf[:, :, -modes:, :modes], w1)
    y = torch.fft.irfft2(torch.complex(out[..., 0], out[..., 1]), s=(n, n), norm="ortho")
    return F.silu(y.permute(0, 2, 3, 1) + res)


def time_fwd_bwd(fn, x, reps):
    def once():
        xx = x.detach().requires_grad_(True)
        y = fn(xx)
        y.backward(torch.ones_like(y))
    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    import galerkin_transformer as gt
    dev = torch.device("cuda:0")
    args = [int(a) for a in sys.argv[1:]]
    sizes = args[1:] or [141, 211]
    C, modes = 32, 12
    for n in sizes:
        B = args[0] if args else (128 if n <= 141 else 8)
        torch.manual_seed(0)
        conv = gt.SpectralConv2d(C, C, modes, dropout=0.0).to(dev)
        x = torch.randn(B, n, n, C, device=dev)
        sd = {k: v.detach() for k, v in conv.state_dict().items()}
        params = [sd[k].clone().requires_grad_(True) for k in ("linear.weight", "linear.bias", "fourier_weight.0",
                                                               "fourier_weight.1")]
        fa = lambda xx: reference_block(xx, *params, modes, n)
        fb = lambda xx: conv(xx)
        with torch.no_grad():
            ya, yb = fa(x), fb(x)
        err = float((ya - yb).norm() / ya.norm())
        ta = time_fwd_bwd(fa, x, 5)
        tb = time_fwd_bwd(fb, x, 20)
        print(json.dumps({"grid": n, "batch": B, "channels": C, "modes": modes,
                          "rocfft_formulation_ms": round(ta, 3), "truncated_dft_ms": round(tb, 3),
                          "speedup": round(ta / tb, 2), "rel_l2_between_them": err}), flush=True)


if __name__ == "__main__":
    main()
