#!/usr/bin/env python3
"""Finer bisect of HIP-graph capture: raw C-ABI launch, forward-only op, forward+backward op."""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
from galerkin_transformer import _hip as H, ops

dev = torch.device("cuda:0")


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def capture(fn, warm=2):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize()
    return g, out


print("stage 1: raw gt_gemm", flush=True)
A, W = torch.randn(1000, 128, device=dev), torch.randn(256, 128, device=dev)
C = torch.empty(1000, 256, device=dev)
ref = A @ W.t()
g, _ = capture(lambda: H.gemm(A, W, C, 1000, 256, 128, lda=128, ldb=128, ldc=256))
C.zero_()
g.replay()
torch.cuda.synchronize()
print("  rel", rel(C, ref), flush=True)

print("stage 2: split-k gt_gemm + colsum", flush=True)
X = torch.randn(5000, 128, device=dev)
G = torch.randn(5000, 256, device=dev)
dW = torch.empty(256, 128, device=dev)


def f2():
    H.gemm(G, X, dW, 256, 128, 5000, layout_a=1, layout_b=1, lda=256, ldb=128, ldc=128, split_k=0)
    return H.colsum(G, 5000, 256, 256)


g, cs = capture(f2)
dW.zero_()
g.replay()
torch.cuda.synchronize()
print("  rel dW", rel(dW, G.t() @ X), "colsum", rel(cs, G.sum(0)), flush=True)

print("stage 3: ops.linear forward (no grad)", flush=True)
b = torch.randn(256, device=dev)
with torch.no_grad():
    g, y = capture(lambda: ops.linear(A, W, b, act="relu"))
    g.replay()
    torch.cuda.synchronize()
print("  rel", rel(y, torch.relu(A @ W.t() + b)), flush=True)

print("stage 4: ops.linear forward+backward", flush=True)
Wp = W.clone().requires_grad_(True)
bp = b.clone().requires_grad_(True)
Ap = A.clone().requires_grad_(True)


def f4():
    for t in (Wp, bp, Ap):
        t.grad = None
    y = ops.linear(Ap, Wp, bp, act="relu")
    y.square().mean().backward()
    return y


f4()
torch.cuda.synchronize()
ge = [t.grad.clone() for t in (Wp, bp, Ap)]
g, y = capture(f4)
g.replay()
torch.cuda.synchronize()
print("  grads rel", [rel(t.grad, e) for t, e in zip((Wp, bp, Ap), ge)], flush=True)

print("stage 5: feed_forward fwd+bwd with dropout", flush=True)
W2 = torch.randn(128, 256, device=dev).requires_grad_(True)
b2 = torch.randn(128, device=dev).requires_grad_(True)


def f5():
    for t in (Wp, bp, Ap, W2, b2):
        t.grad = None
    H.set_seed(7, dev)
    ops._salt[0] = 50
    y = ops.feed_forward(Ap, Wp, bp, W2, b2, res=Ap, act="relu", p_h=0.1, p_out=0.1)
    y.square().mean().backward()
    return y


f5()
torch.cuda.synchronize()
ge = [t.grad.clone() for t in (Wp, bp, Ap, W2, b2)]
g, y = capture(f5)
g.replay()
torch.cuda.synchronize()
print("  grads rel", [rel(t.grad, e) for t, e in zip((Wp, bp, Ap, W2, b2), ge)], flush=True)

import galerkin_transformer as gt
gt.set_attention_dropout("off")
print("stage 6: encoder layer fwd only (no grad)", flush=True)
layer = gt.SimpleTransformerEncoderLayer(d_model=128, pos_dim=2, n_head=4, dim_feedforward=256,
                                         attention_type="galerkin", layer_norm=False, attn_norm=True,
                                         norm_eps=1e-7, dropout=0.0, ffn_dropout=0.0).to(dev).eval()
x = torch.randn(4, 1849, 128, device=dev)
pos = torch.rand(4, 1849, 2, device=dev)
with torch.no_grad():
    ye = layer(x, pos).clone()
    g, y = capture(lambda: layer(x, pos))
    g.replay()
    torch.cuda.synchronize()
print("  rel", rel(y, ye), flush=True)

print("stage 7: encoder layer fwd+bwd", flush=True)
params = list(layer.parameters())


def f7():
    for t in params:
        t.grad = None
    y = layer(x, pos)
    y.square().mean().backward()
    return y


f7()
torch.cuda.synchronize()
ge = [t.grad.clone() for t in params]
g, y = capture(f7)
g.replay()
torch.cuda.synchronize()
print("  worst grad rel", max(rel(t.grad, e) for t, e in zip(params, ge)), flush=True)

print("stage 8: spectral conv fwd+bwd", flush=True)
sc = gt.SpectralConv2d(32, 32, 12, dropout=0.0).to(dev)
xs = torch.randn(4, 141, 141, 32, device=dev)
params = list(sc.parameters())


def f8():
    for t in params:
        t.grad = None
    y = sc(xs)
    y.square().mean().backward()
    return y


f8()
torch.cuda.synchronize()
ge = [t.grad.clone() for t in params]
g, y = capture(f8)
g.replay()
torch.cuda.synchronize()
print("  worst grad rel", max(rel(t.grad, e) for t, e in zip(params, ge)), flush=True)
print("done", flush=True)
