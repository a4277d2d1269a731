#!/usr/bin/env python3
"""Drive the gt_fourier16 kernels at C3's layer shape (for rocprofv3 --pmc runs): B = 8, n = 3721, h = 4, DP = 36."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
from galerkin_transformer import _hip as H
dev = torch.device("cuda:0")
B, n, h, DP = int(os.environ.get("FB", 8)), 3721, 4, 36
g = torch.Generator().manual_seed(1)
Q, K, V, dO = (torch.randn(B * n, h, DP, generator=g).to(dev) for _ in range(4))
H.set_seed(5, dev)
drop = H.dropout_desc(0.5, 3, dev)
scale = 1.0 / math.sqrt(34) / n
imgs = H.fourier16_presplit((Q, K, V, dO), B, n, h, DP)
for _ in range(int(os.environ.get("REPS", 3))):
    H.fourier16_attn(imgs[0], None, imgs[1], imgs[2], B, n, h, DP, scale, None, drop, False)
    H.fourier16_attn(imgs[1], imgs[2], imgs[0], imgs[3], B, n, h, DP, scale, None, drop, True)
    H.fourier16_attn(imgs[0], None, imgs[1], imgs[2], B, n, h, DP, scale, None, None, False)
torch.cuda.synchronize()
