#!/usr/bin/env python3
"""One encoder layer (C2 sizes) at batch B through the HIP path vs the CPU oracle: rel-L2 of output, input gradient and
every parameter gradient as one JSON line.  Run it under different switches (GT_X3_PACKED, GT_DUAL_STREAM, GT_PRECISION,
GT_PLAIN_TILES, GT_DKV_LN, GT_X3Q ...) to find which kernel a parity gap belongs to.  usage: parity_probe.py [B] [f64]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import galerkin_transformer as gt
from oracle import galerkin_oracle as O
from _util import rel_l2

B = int(sys.argv[1]) if len(sys.argv) > 1 else 18
dt = torch.float64 if "f64" in sys.argv else torch.float32
n, d, h, p, ff, eps = 1849, 128, 4, 2, 256, 1e-7
torch.manual_seed(31)
layer = gt.SimpleTransformerEncoderLayer(d_model=d, pos_dim=p, n_head=h, dim_feedforward=ff, attention_type="galerkin",
                                         layer_norm=False, attn_norm=True, norm_eps=eps, dropout=0.0, ffn_dropout=0.0)
with torch.no_grad():
    for prm in layer.parameters():
        prm.add_(0.02 * torch.randn_like(prm))
x, pos, cot = torch.randn(B, n, d), torch.rand(B, n, p), torch.randn(B, n, d)
sd = {k: v.clone().to(dt) for k, v in layer.state_dict().items()}
ref_y, (ref_dx,), ref_dp = O.grads_of(
    lambda s, xx: O.encoder_layer(s, xx, pos.to(dt), n_head=h, attention_type="galerkin", layer_norm=False, attn_norm=True,
                                  norm_eps=eps), sd, [x.to(dt)], cot.to(dt))
dev = torch.device("cuda:0")
layer = layer.to(dev)
gt.set_attention_dropout("off")
xg = x.to(dev).requires_grad_(True)
y = layer(xg, pos.to(dev))
y.backward(cot.to(dev))
torch.cuda.synchronize()
errs = {"out": rel_l2(y, ref_y), "dx": rel_l2(xg.grad, ref_dx)}
for k, v in dict(layer.named_parameters()).items():
    errs[k] = rel_l2(v.grad, ref_dp[k])
env = {k: v for k, v in os.environ.items() if k.startswith("GT_")}
print(json.dumps({"B": B, "oracle": str(dt), "env": env, "worst": max(errs.values()),
                  "errs": {k: float("%.3g" % v) for k, v in errs.items()}}))
