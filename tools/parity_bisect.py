#!/usr/bin/env python3
"""Which class of contraction carries the exact-math gradient excess of the default arithmetic?  

The Darcy 141^2 model of the bench, attention dropout OFF, every activation smooth (SiLU down-scaler AND SiLU FeedForward:
no ReLU kinks, so ONE float64 oracle run serves every HIP variant), B = 9 by default.  The HIP model is run
  * in the default arithmetic (bf16x3) and in f32,
  * with exactly one launch class switched to f32 (`_hip.set_precision_classes`), and
  * in f32 with exactly one launch class left on bf16x3,
and every run's parameter gradients are compared with the float64 oracle (max / median relative L2 over the encoder
parameters, over all parameters, and the prediction).  The float32 oracle's own distance is printed beside them.

usage: parity_bisect.py [B] [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "galerkin-transformer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
import galerkin_transformer as gt
from galerkin_transformer import _hip as H, ops
from oracle import galerkin_oracle as O
from _util import rel_l2

B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "parity_bisect.json")
FFN_ACT = os.environ.get("BISECT_FFN_ACT", "silu")

cfg = bench.darcy_config()
for k in ("dropout", "downscaler_dropout", "upscaler_dropout", "ffn_dropout", "encoder_dropout", "decoder_dropout"):
    cfg[k] = 0.0
cfg["downscaler_activation"] = "silu"
torch.manual_seed(41)
model = gt.FourierTransformer2D(**cfg)
with torch.no_grad():
    for prm in model.parameters():
        prm.add_(0.02 * torch.randn_like(prm))
for layer in model.encoder_layers:
    layer.ff.activation = torch.nn.SiLU() if FFN_ACT == "silu" else torch.nn.ReLU()
b = bench.synthetic_batch(B, torch.device("cpu"), seed=77)
cot = torch.randn(B, bench.N_FINE, bench.N_FINE, 1)
sd0 = {k: v.clone() for k, v in model.state_dict().items()}


def oracle(dt):
    sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    return O.grads_of(lambda s: O.fourier_transformer_2d(s, cfg, b["node"].to(dt), b["pos"].to(dt), b["grid"].to(dt),
                                                         ffn_activation=FFN_ACT), sd, [], cot.to(dt))


t0 = time.time()
ref, _, ref_dp = oracle(torch.float64)
y32, _, dp32 = oracle(torch.float32)
t_oracle = time.time() - t0

dev = torch.device("cuda:0")
model = model.to(dev).train()
bd = {k: v.to(dev) for k, v in b.items()}
cotd = cot.to(dev)
gt.set_attention_dropout("off")


def summarize(out, grads):
    errs = {k: rel_l2(grads[k], ref_dp[k]) for k in ref_dp}
    enc = sorted(v for k, v in errs.items() if k.startswith("encoder_layers."))
    allv = sorted(errs.values())
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    per_layer = [max(v for k, v in errs.items() if k.startswith(f"encoder_layers.{li}.")) for li in range(cfg["num_encoder_layers"])]
    return {"out": rel_l2(out, ref), "enc_max": enc[-1], "enc_med": enc[len(enc) // 2], "all_max": allv[-1],
            "all_med": allv[len(allv) // 2], "per_layer_max": [float("%.3g" % v) for v in per_layer],
            "worst": [(k, float("%.3g" % v)) for k, v in worst]}


DEFAULT = os.environ.get("GT_PRECISION", "bf16x3")


def run(classes=None, base=None, fused_qkv=True, conv_implicit=True):
    old = gt.set_precision(base or DEFAULT)
    H.set_precision_classes(classes)
    ops._qkvnorm_fused[0] = fused_qkv
    ops._conv_implicit[0] = conv_implicit
    ops._scaler_chain[0] = conv_implicit
    try:
        model.zero_grad(set_to_none=True)
        out = model(bd["node"], None, bd["pos"], bd["grid"])["preds"]
        out.backward(cotd)
        torch.cuda.synchronize()
        return summarize(out.cpu(), {k: v.grad.cpu() for k, v in model.named_parameters()})
    finally:
        gt.set_precision(old)
        H.set_precision_classes(None)
        ops._qkvnorm_fused[0] = True
        ops._conv_implicit[0] = True
        ops._scaler_chain[0] = True


res = {"B": B, "ffn_activation": FFN_ACT, "oracle_seconds": t_oracle,
       "oracle_f32": summarize(y32, dp32), "runs": {}}
R = res["runs"]
R[f"{DEFAULT} (default)"] = run()
R["f32 everywhere"] = run(base="f32")
CLS = ("conv",) if os.environ.get("BISECT_QUICK") else ("tok", "batched", "wgrad", "hn", "conv")
for c in CLS:
    if c == "hn":       # the fused epilogue exists on the split engine only: unfused projection (class tok) on fp32 MFMA
        R["bf16x3, QKV projection unfused"] = run(fused_qkv=False)
        continue
    if c == "conv":     # implicit convolutions exist on the split engine only: the library's fp32 convolutions instead
        R[f"{DEFAULT}, convolutions -> library fp32"] = run(conv_implicit=False)
        continue
    R[f"bf16x3, {c} -> f32"] = run(classes={c: "f32"})
if not os.environ.get("BISECT_QUICK"):
    for c in ("tok", "batched", "wgrad"):
        R[f"f32, {c} -> bf16x3"] = run(classes={c: "bf16x3"}, base="f32")
    R["f32, hn+conv -> bf16x3 (tok/batched/wgrad f32)"] = run(classes={"tok": "f32", "batched": "f32", "wgrad": "f32"})
    R["bf16x2 (for scale)"] = run(base="bf16x2")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w") as f:
    json.dump(res, f, indent=1)
print("oracle f32 vs f64:", json.dumps(res["oracle_f32"]))
for k, v in R.items():
    print(f"{k:48s} out {v['out']:.2e}  enc max {v['enc_max']:.2e} med {v['enc_med']:.2e}  all max {v['all_max']:.2e}  layers {v['per_layer_max']}")
