#!/usr/bin/env python3
"""Fused conv0 + resize (gt_conv3x3_resize_fwd/bwd_nhwc) in isolation at the ex2 B = 128 shape: microseconds per launch.
With and without the recorded decisions (relu_bits).  GT_HIP_LIB selects a library variant (GT_CRB_WAVES builds).

    python tools/crb_micro.py [B]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import torch
from galerkin_transformer import _hip as H


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    x = torch.randn(B, 1, 141, 141, device=dev)
    w = torch.randn(128, 1, 3, 3, device=dev) * 0.3
    drop = H.dropout_desc(0.05, 7, dev)
    H.set_seed(1234, dev)
    y, bits = H.conv3x3_resize_fwd(x, w, (78, 78), drop, out_nhwc=True, want_bits=True)
    g = torch.randn_like(y)
    out = {"lib": os.environ.get("GT_HIP_LIB", "libgt_hip.so"), "B": B}
    dw0 = H.conv3x3_resize_bwd(g, y, x, w, drop, out_nhwc=True)
    dw1 = H.conv3x3_resize_bwd(g, y, x, w, drop, out_nhwc=True, bits=bits)
    out["bits_vs_reevaluated_rel"] = float("%.2e" % float((dw1 - dw0).norm() / dw0.norm()))
    for name, fn in (("fwd", lambda: H.conv3x3_resize_fwd(x, w, (78, 78), drop, out_nhwc=True)),
                     ("fwd_bits", lambda: H.conv3x3_resize_fwd(x, w, (78, 78), drop, out_nhwc=True, want_bits=True)),
                     ("bwd", lambda: H.conv3x3_resize_bwd(g, y, x, w, drop, out_nhwc=True)),
                     ("bwd_bits", lambda: H.conv3x3_resize_bwd(g, y, x, w, drop, out_nhwc=True, bits=bits))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name + "_us"] = round(e0.elapsed_time(e1) * 1000 / 20, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
