#!/usr/bin/env python3
"""Per-layer cost of the scaler 3x3 convolutions under MIOpen (the Winograd kernel is persistent: one grid size for every
layer, so the kernel trace cannot split it).  Times forward, data gradient and weight gradient of each ex2 conv shape at
batch 128, NCHW and channels_last.

    python tools/conv_probe.py [B]
"""
import json
import sys

import torch

SHAPES = [("down.conv1 128->42 @78", 128, 42, 78), ("down.conv2 42->42 @78", 42, 42, 78),
          ("down.conv3 42->44 @78", 42, 44, 78), ("up.conv 128->128 @77", 128, 128, 77)]


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda:0")
    for name, ci, co, n in SHAPES:
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = torch.randn(B, ci, n, n, device=dev).contiguous(memory_format=fmt)
            w = torch.randn(co, ci, 3, 3, device=dev).contiguous(memory_format=fmt)
            y = torch.nn.functional.conv2d(x, w, padding=1)
            gy = torch.randn_like(y)
            cb = lambda mask: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                                  mask)
            res = dict(layer=name, batch=B, layout="nhwc" if fmt == torch.channels_last else "nchw",
                       gflop=round(2 * B * n * n * ci * co * 9 / 1e9, 1),
                       fwd_us=round(timeit(lambda: torch.nn.functional.conv2d(x, w, padding=1)), 1),
                       dgrad_us=round(timeit(lambda: cb([True, False, False])), 1),
                       wgrad_us=round(timeit(lambda: cb([False, True, False])), 1))
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
