#!/usr/bin/env python3
"""Block-phase timeline of the packed-B token GEMM (gemm_x3p_kernel) at the ex2 B = 128 shapes.

Needs the profiling build of the library (wall-clock stamps per block, `-DGT_X3P_PROF`):

    python -c "import sys; sys.path.insert(0, 'galerkin-transformer_amd'); import build; \
               build.build(tag='_x3pprof', defines=['GT_X3P_PROF'], only=['gt_gemm_x3.hip'])"
    GT_HIP_LIB=libgt_hip_x3pprof.so python tools/x3p_prof.py [out.json]

Every block records s_memrealtime (100 MHz) at entry, when its first stage has landed, at the end of the K loop and after
its stores have left, plus HW_ID / XCC_ID.  Printed: per-phase durations, how many blocks of the chip are in which phase
over the launch (is the chip loading / computing / storing in lockstep?), and blocks per CU.
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import numpy as np
import torch
from galerkin_transformer import _hip as H


def stamps(nblocks):
    L = H.lib()
    L.gt_debug_x3p_prof.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    L.gt_debug_x3p_prof.restype = ctypes.c_int
    buf = np.zeros((8192, 8), dtype=np.uint64)
    rc = L.gt_debug_x3p_prof(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    return buf[:min(nblocks, 8192)]


def analyse(b):
    t = b[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01                                     # microseconds
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    span = float(t[:, 3].max())
    q = lambda x: [round(float(v), 2) for v in np.percentile(x, [5, 50, 95])]
    wait = b[:, 7].astype(np.int64) * 0.01                   # microseconds of the K loop spent in its vmcnt + barrier waits
    hw, xcc = b[:, 4].astype(np.int64), b[:, 5].astype(np.int64) & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    ids, counts = np.unique(cu, return_counts=True)
    grid = np.linspace(0, span, 41)[:-1] + span / 80
    tl = []
    for x in grid:
        tl.append([int(((t[:, 0] <= x) & (x < t[:, 1])).sum()), int(((t[:, 1] <= x) & (x < t[:, 2])).sum()),
                   int(((t[:, 2] <= x) & (x < t[:, 3])).sum())])
    # resident blocks of one CU over time: the busiest CU's block intervals
    one = np.where(cu == ids[np.argmax(counts)])[0]
    one = one[np.argsort(t[one, 0])]
    return {"blocks": int(len(b)), "span_us": round(span, 2), "prologue_us_p5_50_95": q(pro), "loop_us_p5_50_95": q(loop),
            "epilogue_us_p5_50_95": q(epi), "loop_wait_us_p5_50_95": q(wait), "block_life_us_p5_50_95": q(t[:, 3] - t[:, 0]),
            "sum_over_blocks_us": {"prologue": round(float(pro.sum()), 1), "loop": round(float(loop.sum()), 1),
                                   "epilogue": round(float(epi.sum()), 1)},
            "cus_seen": int(len(ids)), "blocks_per_cu_min_max": [int(counts.min()), int(counts.max())],
            "timeline_t_us": [round(float(x), 1) for x in grid], "timeline_prologue_loop_epilogue": tl,
            "one_cu_blocks_t0_t1_t2_t3": [[round(float(v), 2) for v in t[i]] for i in one[:24]]}


def main():
    dev = torch.device("cuda:0")
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    T = 128 * 43 * 43
    res = {}
    for name, N, K, kw in (("ffn1 N256 K128 bias+relu", 256, 128, dict(bias=True, act=H.ACT_RELU)),
                           ("plain N128 K128", 128, 128, {}), ("ffn2 N128 K256 bias+res", 128, 256, dict(bias=True, res=True))):
        sets = []
        for r in range(3):
            A = torch.randn(T, K, device=dev)
            Bm = torch.randn(N, K, device=dev) * 0.1
            C = torch.empty(T, N, device=dev)
            k = {}
            if kw.get("bias"):
                k["bias"] = torch.randn(N, device=dev)
            if kw.get("act"):
                k["act"] = kw["act"]
            if kw.get("res"):
                k.update(res=torch.randn(T, N, device=dev), ldr=N)
            sets.append((A, Bm, C, k))
        run = lambda i: H.gemm(sets[i % 3][0], sets[i % 3][1], sets[i % 3][2], T, N, K, lda=K, ldb=K, ldc=N, **sets[i % 3][3])
        for i in range(6):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        nblocks = ((T + 127) // 128) * ((N + 127) // 128)
        r = analyse(stamps(nblocks))
        r["us_per_launch_events"] = round(e0.elapsed_time(e1) * 1000 / 12, 1)
        res[name] = r
        print(name, {k: v for k, v in r.items() if not k.startswith("timeline") and not k.startswith("one_cu")})
        print("  timeline (prologue, loop, epilogue blocks):", " ".join("%d/%d/%d" % tuple(x) for x in r["timeline_prologue_loop_epilogue"][::2]))
    # the QKV projection with the head-norm epilogue (plain K' / V' tiles, no raw projection): the product path's launch
    h, dk, pd = 4, 32, 2
    d = h * dk
    DP = H.round4(dk + pd)
    sets = []
    for r in range(3):
        sets.append(dict(x=torch.randn(T, d, device=dev), w=torch.randn(3 * d, d, device=dev) * 0.1,
                         b=torch.randn(3 * d, device=dev), pos=torch.rand(T, pd, device=dev),
                         out=torch.empty(3, T, h, DP, device=dev), st=torch.zeros(2, T, h, 2, device=dev),
                         g=torch.ones(2, h, dk, device=dev), bt=torch.zeros(2, h, dk, device=dev)))

    def run_qkv(i):
        q = sets[i % 3]
        H.gemm(q["x"], q["w"], None, T, 3 * d, d, lda=d, ldb=d, ldc=3 * d, bias=q["b"],
               hn=dict(gamma=q["g"], beta=q["bt"], pos=q["pos"], out=q["out"], stats=q["st"], h=h, dk=dk, p=pd,
                       norm_mask=0b110, eps=1e-7, skip_raw=7, plain=True))
    for i in range(6):
        run_qkv(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(12):
        run_qkv(i)
    e1.record()
    torch.cuda.synchronize()
    r = analyse(stamps(((T + 127) // 128) * 3))
    r["us_per_launch_events"] = round(e0.elapsed_time(e1) * 1000 / 12, 1)
    res["qkv+headnorm N384 K128 plain tiles"] = r
    print("qkv+headnorm", {k: v for k, v in r.items() if not k.startswith("timeline") and not k.startswith("one_cu")})
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
