#!/usr/bin/env python3
"""CPU study for DESIGN §7 item 1: fp32-class GEMM results from TWO fp16 terms per operand (three plane products) against
the shipped THREE bf16 terms (six products).  numpy emulation: terms rounded to the 16-bit format (round to nearest even),
products and sums in float64 (the MFMA's fp32 accumulation error is common to both schemes and left out), error measured
against the float64 product of the float32 operands.

fp16 has 11 significand bits but a narrow exponent range (normal >= 6.1e-5, subnormal quantum 6e-8, max 65504), so the
operands need a per-tensor power-of-two scale; the study reports the error with the scale derived from the tensor's amax
(`exact`), from an amax that is off by 2^-6 / 2^+6 (a stale / delayed estimate), and without any scale.

    python tools/f16x2_study.py
"""
import json

import numpy as np


def split_bf16(a, terms):
    """a (float32) -> list of bf16 terms (as float32 values), residuals exact."""
    out, r = [], a.astype(np.float32)
    for _ in range(terms):
        u = r.view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16).astype(np.uint32)      # RNE to bf16
        h = u.view(np.float32)
        out.append(h)
        r = (r - h).astype(np.float32)
    return out


def split_f16(a, terms, scale):
    out, r = [], (a.astype(np.float32) * np.float32(scale))
    with np.errstate(over="ignore"):
        for _ in range(terms):
            h = r.astype(np.float16).astype(np.float32)
            out.append(h)
            r = (r - h).astype(np.float32)
    return out


def gemm_terms(at, bt, pairs):
    acc = 0.0
    for i, j in pairs:
        acc = acc + at[i].astype(np.float64) @ bt[j].astype(np.float64)
    return acc


def rel(x, ref):
    return float(np.linalg.norm(x - ref) / np.linalg.norm(ref))


def pow2_scale(amax, target=2.0 ** 13):
    return 2.0 ** np.floor(np.log2(target / amax))


def main():
    rng = np.random.default_rng(0)
    T, K, N = 4096, 128, 256
    cases = {
        "activations ~ N(0,1) x weights ~ N(0,0.1)": (rng.standard_normal((T, K)), 0.1 * rng.standard_normal((K, N))),
        "gradients ~ 1e-5 * lognormal spread x weights": (1e-5 * rng.standard_normal((T, K)) * np.exp(rng.standard_normal((T, K))),
                                                          0.1 * rng.standard_normal((K, N))),
        "activations with outliers (1% x 100)": (rng.standard_normal((T, K)) * np.where(rng.random((T, K)) < 0.01, 100.0, 1.0),
                                                 0.1 * rng.standard_normal((K, N))),
        "common offset 100 (cancellation)": (100.0 + rng.standard_normal((T, K)), rng.standard_normal((K, N)) - 0.0),
    }
    six = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
    three = [(0, 0), (0, 1), (1, 0)]
    res = {}
    for name, (A, B) in cases.items():
        A, B = A.astype(np.float32), B.astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        r = {"f32_fma_chain": rel((A @ B).astype(np.float64), ref),
             "bf16x3_six_products": rel(gemm_terms(split_bf16(A, 3), split_bf16(B, 3), six), ref),
             "bf16x2_three_products": rel(gemm_terms(split_bf16(A, 2), split_bf16(B, 2), three), ref)}
        sa, sb = pow2_scale(np.abs(A).max()), pow2_scale(np.abs(B).max())
        for tag, fa in (("exact_amax", 1.0), ("amax_estimate_64x_too_large", 2.0 ** -6), ("amax_estimate_64x_too_small", 2.0 ** 6),
                        ("no_scale", None)):
            s1, s2 = (1.0, 1.0) if fa is None else (sa * fa, sb)
            at, bt = split_f16(A, 2, s1), split_f16(B, 2, s2)
            out = gemm_terms(at, bt, three) / (s1 * s2)
            r["f16x2_three_products/" + tag] = rel(out, ref) if np.isfinite(out).all() else float("inf")
        res[name] = r
        print(name)
        for k, v in r.items():
            print(f"    {k:48s} {v:.2e}")
    # how wide is the window for the scale?  error vs the factor by which the amax estimate is off (gradient-like operand)
    A, B = (c.astype(np.float32) for c in cases["gradients ~ 1e-5 * lognormal spread x weights"])
    ref = A.astype(np.float64) @ B.astype(np.float64)
    sa, sb = pow2_scale(np.abs(A).max()), pow2_scale(np.abs(B).max())
    window = {}
    for e in (3, 2, 0, -4, -8, -12, -16, -20, -24):
        s1 = sa * 2.0 ** e
        out = gemm_terms(split_f16(A, 2, s1), split_f16(B, 2, sb), three) / (s1 * sb)
        window[f"scale x 2^{e}"] = rel(out, ref) if np.isfinite(out).all() else float("inf")
    res["scale_window(gradient operand)"] = window
    print("scale window (scale = 2^13 / amax, times the factor):")
    for k, v in window.items():
        print(f"    {k:24s} {v:.2e}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
