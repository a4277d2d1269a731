"""Spectral convolution operators (SpectralConv1d / SpectralConv2d, reference layers.py:1040-1197).

The reference computes  irfft2( pad( W (x) rfft2(x)[low modes] ) ).  Only ``modes`` (12 of 71) frequencies
per axis are kept, and the grids are awkward FFT lengths (141 = 3*47, 211 prime), so the transform is
evaluated as a *truncated DFT*: four small real GEMMs against precomputed cos/sin bases, which run
on the MFMA GEMM engine, never materialise the zero-padded half spectrum, and read the activations
exactly once in each direction:

  S1  X1[b,x,(ri,ky),c] = sum_y  F1[y,(ri,ky)]        x[b,x,y,c]          (rfft along y, ky < m)
  S2  X2[b,(ro,j),(ky,c)] = sum_(x,ri) G2[(x,ri),(ro,j)] X1[b,(x,ri),(ky,c)] (fft along x, 2m kept rows)
  mix Y[b,.,q,o] = sum_i X2[b,.,q,i] (x) W[i,o,q]                           (complex, real-pair weights)
  S3  Z[b,(x,ro),(ky,o)] = sum_(ri,j) G3[(x,ro),(ri,j)] Y[b,(ri,j),(ky,o)]  (ifft along x)
  S4  out[b,x,y,o] = act( sum_(ro,ky) F4[y,(ro,ky)] Z[b,x,(ro,ky),o] + Linear(x) )   (c2r along y)

``norm='ortho'`` (1/sqrt(n) per axis).  The c2r stage follows irfft semantics exactly: the imaginary
part of the ky=0 (and Nyquist) column is ignored, interior columns count twice.  Backward is the
transposed pipeline through the same bases.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
from torch.autograd import Function

from . import _hip as H
from . import ops
from .ops import _c

_basis_cache: Dict[Tuple, Tuple[torch.Tensor, ...]] = {}


def dft_bases(n: int, m: int, dtype=torch.float64):
    """(F1 [n,2m], G2 [2n,4m], G3 [2n,4m], F4 [n,2m]) in ``dtype`` on the CPU (see module docstring)."""
    if 2 * m > n:
        raise ValueError(f"modes={m} needs n >= 2*modes (n={n})")
    if m > n // 2 + 1:
        raise ValueError("modes exceeds the half spectrum")
    t = torch.arange(n, dtype=torch.float64)
    ky = torch.arange(m, dtype=torch.float64)
    s = 1.0 / math.sqrt(n)
    ang = 2 * math.pi * torch.outer(t, ky) / n                       # [n, m]
    F1 = torch.cat([torch.cos(ang), -torch.sin(ang)], dim=1) * s      # (ri, ky)
    c = torch.full((m,), 2.0, dtype=torch.float64)
    c[0] = 1.0
    if n % 2 == 0 and m - 1 == n // 2:
        c[m - 1] = 1.0
    F4 = torch.cat([torch.cos(ang) * c, -torch.sin(ang) * c], dim=1) * s
    j = torch.arange(2 * m)
    f = torch.where(j < m, j, n - 2 * m + j).to(torch.float64)       # kept x-frequencies
    phi = 2 * math.pi * torch.outer(t, f) / n                         # [n, 2m]
    cs, sn = torch.cos(phi) * s, torch.sin(phi) * s
    G2 = torch.empty(n, 2, 2, 2 * m, dtype=torch.float64)             # [x, ri, ro, j]
    G2[:, 0, 0], G2[:, 1, 0], G2[:, 0, 1], G2[:, 1, 1] = cs, sn, -sn, cs
    G3 = torch.empty(n, 2, 2, 2 * m, dtype=torch.float64)             # [x, ro, ri, j]
    G3[:, 0, 0], G3[:, 0, 1], G3[:, 1, 0], G3[:, 1, 1] = cs, -sn, sn, cs
    return (F1.to(dtype), G2.reshape(2 * n, 4 * m).to(dtype), G3.reshape(2 * n, 4 * m).to(dtype),
            F4.to(dtype))


def _bases(n: int, m: int, device: torch.device):
    key = (n, m, device.type, device.index)
    b = _basis_cache.get(key)
    if b is None:
        b = tuple(t.to(device=device, dtype=torch.float32).contiguous() for t in dft_bases(n, m))
        _basis_cache[key] = b
    return b


class SpectralConv2dFn(Function):
    @staticmethod
    def forward(ctx, x, wlin, blin, w0, w1, modes: int, act: int, want_freq: bool = False):
        H.need_f32_cuda(x, wlin, blin, w0, w1)
        B, n, n2, C = x.shape
        assert n == n2
        m, Co = modes, wlin.shape[0]
        dev = x.device
        F1, G2, G3, F4 = _bases(n, m, dev)
        xc, wl, w0c, w1c = _c(x), _c(wlin), _c(w0), _c(w1)
        T = B * n * n
        f32 = dict(dtype=torch.float32, device=dev)
        wlT = wl.t().contiguous()          # [C, Co]: second-product operand of the last stage (tiny)
        X1 = torch.empty(B * n, 2 * m, C, **f32)
        line = H.dft_supported(n, 2 * m, C, Co)      # per-grid-line stages on the dedicated streaming kernels
        if line:
            H.dft_analysis(F1, xc, X1, B * n, n, 2 * m, C)
        else:
            H.gemm(F1, xc, X1, 2 * m, C, n, layout_a=1, layout_b=1, lda=2 * m, ldb=C, ldc=C, batch=(B * n, 1),
                   b_bs=(n * C, 0), c_bs=(2 * m * C, 0))
        Q = 2 * m * m
        X2 = torch.empty(B, 2, Q, C, **f32)
        H.gemm(G2, X1, X2, 4 * m, m * C, 2 * n, layout_a=1, layout_b=1, lda=4 * m, ldb=m * C, ldc=m * C,
               batch=(B, 1), b_bs=(2 * n * m * C, 0), c_bs=(2 * Q * C, 0))
        Y = torch.empty(B, 2, Q, Co, **f32)
        H.modemix_fwd(X2, w0c, Y, B, m * m, C, Co, Q, 0)
        H.modemix_fwd(X2, w1c, Y, B, m * m, C, Co, Q, m * m)
        Z = torch.empty(B * n, 2 * m, Co, **f32)
        H.gemm(G3, Y, Z, 2 * n, m * Co, 4 * m, layout_b=1, lda=4 * m, ldb=m * Co, ldc=m * Co, batch=(B, 1),
               b_bs=(2 * Q * Co, 0), c_bs=(2 * n * m * Co, 0))
        out = torch.empty(B, n, n, Co, **f32)
        pre = torch.empty(B, n, n, Co, **f32) if act != H.ACT_NONE else None
        # out = act( c2r-stage(Z) + x Wl^T + b ): the residual Linear is the second product of the same launch
        if line:
            H.dft_synthesis(F4, Z, out, B * n, n, 2 * m, Co, xc, wlT, C, bias=blin, act=act, pre=pre)
        else:
            H.gemm(F4, Z, out, n, Co, 2 * m, layout_b=1, lda=2 * m, ldb=Co, ldc=Co, batch=(B * n, 1),
                   b_bs=(2 * m * Co, 0), c_bs=(n * Co, 0), bias=blin, act=act, pre=pre, ldpre=Co,
                   K2=C, A2=xc, lda2=C, a2_bs=(n * C, 0), B2=wlT, ldb2=Co)
        gate = ops._take_gate(xc) if ctx.needs_input_grad[0] else None       # x = silu(gate) of the layer in front (ops.silu_gate_scope)
        ctx.save_for_backward(xc, wl, w0c, w1c, X2, pre, gate)
        ctx.cfg = (B, n, C, Co, m, act, blin is not None)
        ops._offer_gate(ctx, out, pre if act == H.ACT_SILU else None)
        Yout = Y if want_freq else Y.new_empty(0)        # the mixed retained modes [B, 2, 2 m m, Co] (detached)
        ctx.mark_non_differentiable(Yout)
        return out, Yout

    @staticmethod
    def backward(ctx, gy, _gfreq=None):
        xc, wl, w0c, w1c, X2, pre, gate = ctx.saved_tensors
        B, n, C, Co, m, act, has_b = ctx.cfg
        dev = gy.device
        F1, G2, G3, F4 = _bases(n, m, dev)
        T, Q = B * n * n, 2 * m * m
        f32 = dict(dtype=torch.float32, device=dev)
        g = _c(gy)
        # (g_gated: the consumer of this layer's result took its pre-activation and already multiplied by silu')
        dpre = g if act == H.ACT_NONE or getattr(ctx, "g_gated", False) else H.act_bwd(g, pre, act)
        dZ = torch.empty(B * n, 2 * m, Co, **f32)
        line = H.dft_supported(n, 2 * m, C, Co)
        if line:
            H.dft_analysis(F4, dpre, dZ, B * n, n, 2 * m, Co)
        else:
            H.gemm(F4, dpre, dZ, 2 * m, Co, n, layout_a=1, layout_b=1, lda=2 * m, ldb=Co, ldc=Co,
                   batch=(B * n, 1), b_bs=(n * Co, 0), c_bs=(2 * m * Co, 0))
        dY = torch.empty(B, 2, Q, Co, **f32)
        H.gemm(G3, dZ, dY, 4 * m, m * Co, 2 * n, layout_a=1, layout_b=1, lda=4 * m, ldb=m * Co, ldc=m * Co,
               batch=(B, 1), b_bs=(2 * n * m * Co, 0), c_bs=(2 * Q * Co, 0))
        dX2 = torch.empty(B, 2, Q, C, **f32)
        dw0, dw1 = torch.empty_like(w0c), torch.empty_like(w1c)
        H.modemix_bwd(X2, w0c, dY, dX2, dw0, B, m * m, C, Co, Q, 0)
        H.modemix_bwd(X2, w1c, dY, dX2, dw1, B, m * m, C, Co, Q, m * m)
        dX1 = torch.empty(B * n, 2 * m, C, **f32)
        H.gemm(G2, dX2, dX1, 2 * n, m * C, 4 * m, layout_b=1, lda=4 * m, ldb=m * C, ldc=m * C, batch=(B, 1),
               b_bs=(2 * Q * C, 0), c_bs=(2 * n * m * C, 0))
        # dx = r2c-stage^T(dX1) + dpre Wl  (second product of the same launch)
        dx = torch.empty(B, n, n, C, **f32)
        if line:
            H.dft_synthesis(F1, dX1, dx, B * n, n, 2 * m, C, dpre, wl, Co, out_gate=gate)
        else:
            H.gemm(F1, dX1, dx, n, C, 2 * m, layout_b=1, lda=2 * m, ldb=C, ldc=C, batch=(B * n, 1),
                   b_bs=(2 * m * C, 0), c_bs=(n * C, 0),
                   K2=Co, A2=dpre, lda2=Co, a2_bs=(n * Co, 0), B2=wl, ldb2=C)
            if gate is not None:
                dx = H.act_bwd(dx, gate.reshape(dx.shape), H.ACT_SILU)
        dwl = torch.empty(Co, C, **f32)
        dbl = torch.empty(Co, **f32) if has_b else None
        H.gemm(dpre, xc, dwl, Co, C, T, layout_a=1, layout_b=1, lda=Co, ldb=C, ldc=C, split_k=0, a_colsum=dbl)
        return dx, dwl, dbl, dw0, dw1, None, None, None


class SpectralConv1dFn(Function):
    @staticmethod
    def forward(ctx, x, wlin, blin, w, modes: int, act: int, want_freq: bool = False):
        H.need_f32_cuda(x, wlin, blin, w)
        B, n, C = x.shape
        m, Co = modes, wlin.shape[0]
        dev = x.device
        F1, _, _, F4 = _bases(n, m, dev)
        xc, wl, wc = _c(x), _c(wlin), _c(w)
        T = B * n
        f32 = dict(dtype=torch.float32, device=dev)
        wlT = wl.t().contiguous()
        X = torch.empty(B, 2, m, C, **f32)
        H.gemm(F1, xc, X, 2 * m, C, n, layout_a=1, layout_b=1, lda=2 * m, ldb=C, ldc=C, batch=(B, 1),
               b_bs=(n * C, 0), c_bs=(2 * m * C, 0), split_k=0)
        Y = torch.empty(B, 2, m, Co, **f32)
        H.modemix_fwd(X, wc, Y, B, m, C, Co, m, 0)
        out = torch.empty(B, n, Co, **f32)
        pre = torch.empty(B, n, Co, **f32) if act != H.ACT_NONE else None
        H.gemm(F4, Y, out, n, Co, 2 * m, layout_b=1, lda=2 * m, ldb=Co, ldc=Co, batch=(B, 1),
               b_bs=(2 * m * Co, 0), c_bs=(n * Co, 0), bias=blin, act=act, pre=pre, ldpre=Co,
               K2=C, A2=xc, lda2=C, a2_bs=(n * C, 0), B2=wlT, ldb2=Co)
        ctx.save_for_backward(xc, wl, wc, X, pre)
        ctx.cfg = (B, n, C, Co, m, act, blin is not None)
        Yout = Y if want_freq else Y.new_empty(0)
        ctx.mark_non_differentiable(Yout)
        return out, Yout

    @staticmethod
    def backward(ctx, gy, _gfreq=None):
        xc, wl, wc, X, pre = ctx.saved_tensors
        B, n, C, Co, m, act, has_b = ctx.cfg
        dev = gy.device
        F1, _, _, F4 = _bases(n, m, dev)
        T = B * n
        f32 = dict(dtype=torch.float32, device=dev)
        g = _c(gy)
        dpre = H.act_bwd(g, pre, act) if act != H.ACT_NONE else g
        dY = torch.empty(B, 2, m, Co, **f32)
        H.gemm(F4, dpre, dY, 2 * m, Co, n, layout_a=1, layout_b=1, lda=2 * m, ldb=Co, ldc=Co, batch=(B, 1),
               b_bs=(n * Co, 0), c_bs=(2 * m * Co, 0), split_k=0)
        dX = torch.empty(B, 2, m, C, **f32)
        dw = torch.empty_like(wc)
        H.modemix_bwd(X, wc, dY, dX, dw, B, m, C, Co, m, 0)
        dx = torch.empty(B, n, C, **f32)
        H.gemm(F1, dX, dx, n, C, 2 * m, layout_b=1, lda=2 * m, ldb=C, ldc=C, batch=(B, 1),
               b_bs=(2 * m * C, 0), c_bs=(n * C, 0),
               K2=Co, A2=dpre, lda2=Co, a2_bs=(n * Co, 0), B2=wl, ldb2=C)
        dwl = torch.empty(Co, C, **f32)
        dbl = torch.empty(Co, **f32) if has_b else None
        H.gemm(dpre, xc, dwl, Co, C, T, layout_a=1, layout_b=1, lda=Co, ldb=C, ldc=C, split_k=0, a_colsum=dbl)
        return dx, dwl, dbl, dw, None, None, None


def spectral_conv2d(x, wlin, blin, w0, w1, modes: int, act: str = "silu", return_freq: bool = False):
    """return_freq: also the zero-padded half spectrum of the mixed modes, (B, Cout, n, n//2 + 1) complex64, as the reference
    returns it (layers.py:1179-1197) -- assembled from the retained coefficients, detached (a diagnostic output).
    DEVIATION from the reference: the returned spectrum is DETACHED (marked non-differentiable).  In the reference `out_ft`
    is part of the autograd graph (layers.py:1179-1196), so a loss or regulariser built on it reaches the weights; here it
    would silently get zero gradient.  No script of the reference differentiates through it (it is a debug / plotting output);
    build such a term from the module's output instead."""
    out, Y = SpectralConv2dFn.apply(x, wlin, blin, w0, w1, int(modes), H.ACT_CODE[act], bool(return_freq))
    if not return_freq:
        return out
    B, n, m, Co = x.shape[0], x.shape[1], int(modes), wlin.shape[0]
    blk = torch.complex(Y[:, 0], Y[:, 1]).view(B, 2 * m, m, Co).permute(0, 3, 1, 2)     # [B, Co, (low | high) rows, ky]
    ft = torch.zeros(B, Co, n, n // 2 + 1, dtype=torch.complex64, device=x.device)
    ft[:, :, :m, :m] = blk[:, :, :m]
    ft[:, :, n - m:, :m] = blk[:, :, m:]
    return out, ft


def spectral_conv1d(x, wlin, blin, w, modes: int, act: str = "silu", return_freq: bool = False):
    out, Y = SpectralConv1dFn.apply(x, wlin, blin, w, int(modes), H.ACT_CODE[act], bool(return_freq))
    if not return_freq:
        return out
    B, n, m, Co = x.shape[0], x.shape[1], int(modes), wlin.shape[0]
    ft = torch.zeros(B, Co, n // 2 + 1, dtype=torch.complex64, device=x.device)
    ft[:, :, :m] = torch.complex(Y[:, 0], Y[:, 1]).permute(0, 2, 1)                        # [B, m, Co] -> [B, Co, m]
    return out, ft
