#!/usr/bin/env python3
"""CPU study for DESIGN §7 item 0: activations STORED as two fp16 terms + a per-row power-of-two exponent (written once by
the producing epilogue, 4 bytes per value like fp32), consumed without a split:

  * row-contracted product  Y = X W^T  (token GEMM, X stored split, W packed as today): the row exponent factors out of
    every dot product -- nothing to do;
  * token-contracted product  dW = G^T X  (weight gradient, BOTH operands stored split with their own row exponents e_j, f_j):
    the term of token j carries 2^(e_j + f_j), which depends on the contraction index.  The consumer therefore multiplies the
    stored fp16 terms of ONE operand by 2^(c - e_j - f_j) (a packed fp16 multiply by a power of two: exact until the result
    leaves fp16's normal range; c = one exponent for the block, chosen so that the largest shifted magnitude stays below 2^15).
    Rows whose terms are small against the largest lose low bits in fp16's subnormal range.

numpy emulation: float16 terms (round to nearest even, subnormals as the hardware keeps them), products and sums in float64
(the fp32 accumulation of the MFMA is common to every scheme and left out); three plane products (hi hi, hi lo, lo hi) as in
the shipped kernels; error against the float64 product of the float32 operands; the shipped on-the-fly split (running
exponents replaced by the exact per-row / per-block amax) beside it.

    python tools/presplit_study.py
"""
import json

import numpy as np


def rel(x, ref):
    return float(np.linalg.norm(x - ref) / np.linalg.norm(ref))


def row_exp(a, target=14):
    """e such that the row's amax * 2^e lies in [2^target, 2^(target+1)); 0 for an all-zero row."""
    m = np.abs(a).max(axis=1)
    return np.where(m > 0, target - np.floor(np.log2(np.where(m > 0, m, 1.0))), 0.0)


def split2(a):
    """float32 -> (hi, lo) float16 terms (as float64 values), hi + lo = a up to 2^-22 of the value (or the subnormal quantum)."""
    a = a.astype(np.float32)
    with np.errstate(over="ignore"):
        hi = a.astype(np.float16)
        lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def three(ah, al, bh, bl, contract):
    return contract(al, bh) + contract(ah, bl) + contract(ah, bh)


def main():
    rng = np.random.default_rng(1)
    T, K, N = 8192, 128, 256
    out = {}
    # activations with rows over four decades, gradient rows over six (what a loss gradient field looks like)
    X = (rng.standard_normal((T, K)) * np.exp2(rng.integers(-6, 7, size=(T, 1)))).astype(np.float32)
    G = (rng.standard_normal((T, N)) * np.exp2(rng.integers(-30, -10, size=(T, 1)))).astype(np.float32)
    W = (0.1 * rng.standard_normal((N, K))).astype(np.float32)

    # ---- row-contracted: Y = X W^T
    ref = X.astype(np.float64) @ W.astype(np.float64).T
    ex = row_exp(X)
    xh, xl = split2(X * np.exp2(ex)[:, None].astype(np.float32))            # STORED terms of X
    ew = 14 - np.floor(np.log2(np.abs(W).max()))
    wh, wl = split2(W * np.float32(2.0 ** ew))
    y = three(xh, xl, wh, wl, lambda a, b: a @ b.T) * np.exp2(-ex)[:, None] * 2.0 ** -ew
    out["row-contracted X W^T, X stored split (per-row exponent)"] = rel(y, ref)

    # ---- token-contracted: dW = G^T X, both stored split with their own row exponents
    ref = G.astype(np.float64).T @ X.astype(np.float64)
    eg = row_exp(G)
    gh, gl = split2(G * np.exp2(eg)[:, None].astype(np.float32))            # STORED terms of G
    # consumer: shift the stored terms of X by 2^(c - e_j - f_j) in fp16 (exact multiply, then fp16 rounding / flush)
    shift = -(ex + eg)                                                       # per row, before the block exponent
    mag = np.abs(X).max(axis=1) * np.exp2(ex) * np.exp2(shift)               # magnitude of the shifted row
    c = 13 - np.floor(np.log2(mag.max()))
    f = np.exp2(c + shift)[:, None]
    with np.errstate(over="ignore", under="ignore"):
        sh = (xh * f).astype(np.float16).astype(np.float64)
        sl = (xl * f).astype(np.float16).astype(np.float64)
    dw = three(gh, gl, sh, sl, lambda a, b: a.T @ b) * 2.0 ** -c
    out["token-contracted G^T X, both stored split, X re-scaled by 2^(c-e_j-f_j) in fp16"] = rel(dw, ref)
    # the same with the shift applied BEFORE the split (what a consumer that still splits from fp32 would get)
    xs_h, xs_l = split2((X.astype(np.float64) * np.exp2(ex)[:, None] * f).astype(np.float32))
    dw2 = three(gh, gl, xs_h, xs_l, lambda a, b: a.T @ b) * 2.0 ** -c
    out["token-contracted, X split on the fly after the row shift"] = rel(dw2, ref)
    # shipped scheme: one exponent per operand and block (exact amax), both operands split on the fly
    sg, sx = 2.0 ** (13 - np.floor(np.log2(np.abs(G).max()))), 2.0 ** (13 - np.floor(np.log2(np.abs(X).max())))
    a_h, a_l = split2(G * np.float32(sg))
    b_h, b_l = split2(X * np.float32(sx))
    dw3 = three(a_h, a_l, b_h, b_l, lambda a, b: a.T @ b) / (sg * sx)
    out["token-contracted, shipped: one exponent per operand and block, split on the fly"] = rel(dw3, ref)
    # how many stored values the shift pushes into fp16's subnormal range / to zero
    hi_shift = np.abs(xh * f)
    out["share of X values whose hi term leaves the normal range under the shift"] = float(((hi_shift < 2.0 ** -14) & (xh != 0)).mean())
    out["share flushed to zero"] = float(((hi_shift < 2.0 ** -25) & (xh != 0)).mean())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
