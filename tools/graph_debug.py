#!/usr/bin/env python3
"""Bisect HIP-graph capture problems at model level: eager vs graph-replay gradients per sub-module."""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
import bench
import galerkin_transformer as gt

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def check(name, fn, params):
    def run():
        for p in params:
            p.grad = None
        l = fn()
        l.backward()
        return l.detach()

    le = float(run())
    torch.cuda.synchronize()
    ge = [p.grad.clone() for p in params]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    g = torch.cuda.CUDAGraph()
    lbuf = torch.zeros((), device=dev)
    with torch.cuda.graph(g):
        lbuf.copy_(run())
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    worst = max(rel(p.grad, e) for p, e in zip(params, ge))
    print(f"{name:28s} loss eager {le:+.6e} graph {float(lbuf):+.6e}  worst grad rel diff {worst:.2e}", flush=True)


torch.manual_seed(0)
cfg = bench.darcy_config()
model = gt.FourierTransformer2D(**cfg).to(dev).eval()
gt.set_attention_dropout("off")
node, pos, grid, target = bench.synthetic_batch(B, dev, 1)
which = sys.argv[2] if len(sys.argv) > 2 else "all"

if which in ("all", "reg"):
    x_u = torch.randn(B, 141, 141, 128, device=dev)
    reg = model.regressor
    check("spectral regressor", lambda: reg(x_u, grid=grid).square().mean(), list(reg.parameters()))
if which in ("all", "down"):
    ds = model.downscaler
    check("downscaler (torch)", lambda: ds(node).square().mean(), list(ds.parameters()))
if which in ("all", "up"):
    us = model.upscaler
    x_c = torch.randn(B, 43, 43, 128, device=dev)
    check("upscaler (torch)", lambda: us(x_c).square().mean(), list(us.parameters()))
if which in ("all", "model"):
    check("whole model", lambda: ((model(node, None, pos, grid)["preds"] - target) ** 2).mean(),
          list(model.parameters()))
