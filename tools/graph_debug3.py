#!/usr/bin/env python3
"""Which ingredient of the captured training step breaks?  usage: graph_debug3.py B variant
variants: eval_noopt | train_noopt | eval_clip | eval_adam_foreach | eval_adam_fused | full"""
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
B, variant = int(sys.argv[1]), sys.argv[2]
torch.manual_seed(0)
cfg = bench.darcy_config()
for kv in filter(None, os.environ.get("GT_CFG", "").split(",")):
    k, v = kv.split("=")
    cfg[k] = float(v)
model = gt.FourierTransformer2D(**cfg).to(dev)
train = variant.startswith("train") or variant == "full"
model.train(train)
gt.set_attention_dropout(os.environ.get("GT_ATTN", "reference" if train else "off"))
batch = bench.synthetic_batch(B, dev, 1)
tr = bench.Trainer(model, batch, 1)
if "foreach" in variant:
    tr.opt = torch.optim.Adam(tr.params, lr=1e-3, capturable=True, foreach=True)
do_clip = variant in ("eval_clip", "full") or "adam" in variant
do_adam = "adam" in variant or variant == "full"


def opt_step():
    if do_clip:
        torch.nn.utils.clip_grad_norm_(tr.params, 0.99, foreach=True)
    if do_adam:
        tr.opt.step()


tr.opt_step = opt_step
ok = tr.capture()
for i in range(5):
    tr.step()
torch.cuda.synchronize()
print(f"{variant:20s} B={B} graphed={ok} loss after 5 replays = {float(tr.loss):+.6f}", flush=True)
