#!/usr/bin/env python3
"""Which ATen kernels does one training step of the headline workload launch, and from where?  (dispatch diet: every one of
them is a candidate for folding into a library launch or a persistent view)   python tools/aten_in_step.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
import bench
import galerkin_transformer as gt
from torch.profiler import profile, ProfilerActivity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = sys.argv[2] if len(sys.argv) > 2 else "ex2_darcy141"
dev = torch.device("cuda:0")
torch.manual_seed(1)
model, cfg = bench.build_model(wl)
model = model.to(dev).train()
batch = bench.synthetic_batch(B, dev, seed=3, workload=wl)
tr = bench.Trainer(model, batch, 1, use_graph=False, workload=wl)
for _ in range(3):
    tr.eager_step()
torch.cuda.synchronize()
import traceback
from collections import Counter
from torch.utils._python_dispatch import TorchDispatchMode

WATCH = ("copy_", "fill_", "zero_", "cat", "stack", "clone", "flip", "add", "mul", "sub", "div", "mean", "pow", "_foreach_copy_",
         "zeros", "zeros_like", "index_select", "sum", "neg", "where", "slice_backward", "select_backward", "constant_pad_nd",
         "convolution", "convolution_backward", "miopen_convolution", "upsample_bilinear2d", "upsample_bilinear2d_backward", "silu", "silu_backward", "permute", "native_dropout")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        out = func(*args, **(kwargs or {}))
        if name in WATCH:
            dev = None
            for a in list(args) + [out]:
                if isinstance(a, torch.Tensor):
                    dev = a.device.type
                    break
                if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
                    dev = a[0].device.type
                    break
            if dev == "cuda":
                fr = [f for f in traceback.extract_stack() if ("galerkin_transformer" in f.filename or f.filename.endswith("bench.py"))
                      and "aten_in_step" not in f.filename]
                where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line.strip()[:90]}" if fr else "(autograd engine)"
                self.c[(name, where)] += 1
        return out


spy = Spy()
with spy:
    tr.eager_step()
torch.cuda.synchronize()
tot = 0
for (name, where), n in sorted(spy.c.items(), key=lambda kv: -kv[1]):
    tot += n
    print(f"{n:4d}  {name:16s} {where}")
print("watched aten ops on device tensors in one step:", tot)
