#!/usr/bin/env python3
"""Headline benchmark: training samples/s of the Darcy 141x141 Galerkin-encoder model
(BASELINE.json configs[1]: ex2_darcy.py FourierTransformer2D, 6 x (d=128, 4 heads, d_k'=34) encoder
layers on the 43x43 coarse grid, two SpectralConv2d decoder layers at 141x141).

    python bench.py --gpus N --steps K --warmup W [--batch B_per_gpu] [--scaling weak|strong --global-batch G]
                    [--workload ...] [--loss mse|weighted_l2] [--precision f16x2|bf16x3|f32|bf16x2|bf16]

A step = forward + loss + backward + (gradient all-reduce) + clip_grad_norm_(0.99) + Adam on one batch of
synthetic tensors already resident in HBM (the recipe of the reference's examples/ex2_memory_profile.py:58-71
plus the optimizer of examples/ex2_darcy.py).  Every dropout of config.yml is active (train mode), including
the reference's always-on p=0.5 attention dropout.  Operands, accumulators and results are fp32; the contractions
run in the library's default arithmetic (``f16x2``: every fp32 operand value as two fp16 terms under an in-kernel
power-of-two scale, three products on the f16 MFMA pipe, fp32 accumulation -- the mode the 1e-5 parity tests run in;
``bf16x3`` = three bf16 terms / six products, the default of rounds 2-3, is still selectable); ``--precision f32``
selects the bit-exact fp32 MFMA kernels, and the default run reports both.

One process per GPU (launched by torch.distributed.run for N>1), batch-sharded data parallel: the per-GPU batch is
fixed (``--scaling weak``, the default) or the global batch is (``--scaling strong --global-batch G``); one flat
gradient all-reduce over RCCL per step.  Rank 0 prints ONE JSON line (see the bench contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import torch
import torch.distributed as dist

N_FINE, N_COARSE = 141, 43          # 421 grid subsampled by 3 / by 10 (ex2_darcy.py defaults)
PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
PEAK_HBM_GBS = 8000.0
PLANE_PRODUCTS = {"bf16x3": 6, "bf16x2": 3, "bf16": 1, "f16x2": 3}


# name -> (config.yml section, fine subsample, coarse subsample, config overrides, where the target lives)
# The headline (BASELINE.json configs[1]) is ex2_darcy141; the others are informational end-to-end runs of
# the remaining BASELINE configurations through the same training step.
WORKLOADS = {
    "ex2_darcy141": ("ex2_darcy", 3, 10, dict(), "fine"),
    "ex2_darcy211_fourier": ("ex2_darcy", 2, 7, dict(attention_type="fourier", xavier_init=0.001), "fine"),
    "ex3_darcy_inv": ("ex3_darcy_inv", 3, 12, dict(), "coarse"),
    "ex4_ns": None,                  # FourierTransformer2DLite, 64x64, 10-step rollout (ns_lite.py:205-238)
    "ex1_burgers": None,             # SimpleTransformer, n = 8192, d_model 64, 4 heads (BASELINE configs[0])
}
DEFAULT_BATCH = {"ex2_darcy141": 128, "ex2_darcy211_fourier": 8, "ex3_darcy_inv": 128, "ex4_ns": 16, "ex1_burgers": 32}

WORKLOAD_TEXT = {
    "ex2_darcy141": "ex2_darcy FourierTransformer2D 141x141 fine / 43x43 coarse, galerkin, "
                    "6 layers d=128 h=4 ffn=256, 2x SpectralConv2d(32, modes 12)",
    "ex2_darcy211_fourier": "ex2_darcy FourierTransformer2D 211x211 fine / 61x61 coarse, fourier (QK^T)V, "
                            "6 layers d=128 h=4 ffn=256, 2x SpectralConv2d(32, modes 12) [informational]",
    "ex3_darcy_inv": "ex3_darcy_inv FourierTransformer2D 141x141 -> 36x36, galerkin, 6 layers d=192 h=4 ffn=384, "
                     "pointwise decoder, input noise 0.1 [informational]",
    "ex4_ns": "ex4 FourierTransformer2DLite 64x64, galerkin 1 head d=48 (LayerNorm, no per-head norm), 4 layers, "
              "2x SpectralConv2d(20, modes 12); one step = 10-step autoregressive rollout + one backward [informational]",
    "ex1_burgers": "ex1_burgers SimpleTransformer n=8192, galerkin, 4 layers d=64 h=4 ffn=128, 2x SpectralConv1d "
                   "[informational]",
}


def _n(sub):
    return (421 - 1) // sub + 1


def darcy_config(workload="ex2_darcy141"):
    import yaml
    from galerkin_transformer.ft import DarcyDataset
    section, sf, sc, over, where = WORKLOADS[workload]
    with open(os.path.join(ROOT, "galerkin-transformer_amd", "config.yml")) as f:
        cfg = yaml.full_load(f)[section]
    n_f, n_c = _n(sf), _n(sc)
    down, up = DarcyDataset.get_scaler_sizes(n_f, n_c)
    if where == "coarse":                      # ex3: the decoder works on the coarse grid (ex3_darcy_inv.py)
        up = ((n_c, n_c), (n_c, n_c))
    cfg.update(downscaler_size=down, upscaler_size=up, attn_norm=True, norm_eps=1e-7, normalizer=None)
    cfg.update(over)
    return cfg


def ns_config():
    """examples/ex4_navier_stokes_2+1d.py:29-58."""
    return dict(node_feats=10 + 2, pos_dim=2, n_targets=1, n_hidden=48, num_feat_layers=0, num_encoder_layers=4,
                n_head=1, dim_feedforward=96, attention_type="galerkin", feat_extract_type=None, xavier_init=0.01,
                diagonal_weight=0.01, layer_norm=True, attn_norm=False, return_attn_weight=False,
                return_latent=False, decoder_type="ifft", freq_dim=20, num_regressor_layers=2, fourier_modes=12,
                spacial_dim=2, spacial_fc=False, dropout=0.0, encoder_dropout=0.0, decoder_dropout=0.0,
                ffn_dropout=0.05, debug=False)


def burgers_config():
    """config.yml:ex1_burgers with the BASELINE.json configs[0] override (galerkin, d_model 64, 4 heads)."""
    import yaml
    with open(os.path.join(ROOT, "galerkin-transformer_amd", "config.yml")) as f:
        cfg = yaml.full_load(f)["ex1_burgers"]
    cfg.update(attention_type="galerkin", n_hidden=64, n_head=4, dim_feedforward=128)
    return cfg


def build_model(workload):
    import galerkin_transformer as gt
    if workload == "ex4_ns":
        cfg = ns_config()
        return gt.FourierTransformer2DLite(**cfg), cfg
    if workload == "ex1_burgers":
        cfg = burgers_config()
        return gt.SimpleTransformer(**cfg), cfg
    cfg = darcy_config(workload)
    return gt.FourierTransformer2D(**cfg), cfg


def _central_diff2d(u, h):
    """(B, n, n, 1) -> (B, n, n, 2): dilation-2 central differences on the zero-padded field (what the datasets
    store as target_grad)."""
    import torch.nn.functional as F
    p = F.pad(u[..., 0], (1, 1, 1, 1))
    gx = (p[:, 2:, 1:-1] - p[:, :-2, 1:-1]) / 2
    gy = (p[:, 1:-1, 2:] - p[:, 1:-1, :-2]) / 2
    return torch.stack([gx, gy], dim=-1) / h


def synthetic_batch(B, device, seed, workload="ex2_darcy141"):
    """Dict of HBM-resident tensors of the workload's shapes (randn fields on the true grids)."""
    from galerkin_transformer.ft import DarcyDataset
    g = torch.Generator().manual_seed(seed)
    if workload == "ex4_ns":
        n = 64
        ax = torch.linspace(0, 1, n)
        gx, gy = torch.meshgrid(ax, ax, indexing="xy")
        grid = torch.stack([gx, gy], -1)
        x, u = torch.randn(B, n, n, 10, generator=g), torch.randn(B, n, n, 10, generator=g)
        gradu = torch.stack([_central_diff2d(u[..., t:t + 1], 1 / n) for t in range(10)], dim=-1)
        out = dict(node=x, pos=grid.reshape(1, -1, 2).repeat(B, 1, 1), grid=grid.unsqueeze(0).repeat(B, 1, 1, 1),
                   target=u, target_grad=gradu)
    elif workload == "ex1_burgers":
        n = 8192
        pos = torch.linspace(0, 1, n)[None, :, None].repeat(B, 1, 1)
        out = dict(node=torch.randn(B, n, 1, generator=g), pos=pos, grid=pos.clone(),
                   target=torch.randn(B, n, 1, generator=g))
    else:
        _, sf, sc, _, where = WORKLOADS[workload]
        n_f, n_t = _n(sf), (_n(sf) if where == "fine" else _n(sc))
        node = torch.randn(B, n_f, n_f, 1, generator=g)
        target = torch.randn(B, n_t, n_t, 1, generator=g)
        if workload == "ex3_darcy_inv":                  # ex3_darcy_inv.py --noise 0.1: noisy measurements in
            node = node + 0.1 * torch.randn(B, n_f, n_f, 1, generator=g)
        pos = torch.from_numpy(DarcyDataset.get_grid(421, subsample=sc, return_elem=False)).float()
        grid = torch.from_numpy(DarcyDataset.get_grid(421, subsample=(sf if where == "fine" else sc),
                                                      return_elem=False)).float()
        out = dict(node=node, pos=pos.reshape(1, -1, 2).repeat(B, 1, 1), grid=grid.unsqueeze(0).repeat(B, 1, 1, 1),
                   target=target, target_grad=_central_diff2d(target, 1.0 / n_t),
                   coeff=torch.rand(B, n_t, n_t, 1, generator=g) + 0.5)
    return {k: v.to(device).contiguous() for k, v in out.items()}


def make_loss(workload, kind):
    """batch, model -> scalar loss tensor, with no host read-back (the step is captured into a HIP graph).
    mse: the reference profilers' objective (ex2_memory_profile.py:58-71).  weighted_l2: the training objective of
    the example scripts, WeightedL2Loss2d(regularizer=True, h, gamma) loss + regulariser (ex2_darcy.py:118,
    utils_ft.py:656-681; ns_lite.py:205-238 for the rollout)."""
    from galerkin_transformer.ft import WeightedL2Loss2d
    if workload == "ex4_ns":
        lf = WeightedL2Loss2d(regularizer=True, h=1 / 64, gamma=0.1)

        def rollout(model, b):
            x, total = b["node"], 0
            for t in range(x.size(-1)):
                up = model(x, None, pos=b["pos"], grid=b["grid"])["preds"]
                if kind == "mse":
                    total = total + ((up[..., 0] - b["target"][..., t]) ** 2).mean()
                else:
                    l, r, _ = lf.terms(up[..., 0], b["target"][..., t], targets_prime=b["target_grad"][..., t])
                    total = total + lf.reduce(l) + lf.reduce(r)
                x = torch.cat((x[..., 1:], up), dim=-1)
            return total
        return rollout
    if workload == "ex1_burgers":
        if kind != "mse":
            raise SystemExit("--loss weighted_l2 is implemented for the 2-D workloads")
        return lambda model, b: ((model(b["node"], None, b["pos"], b["grid"])["preds"][..., :1] - b["target"]) ** 2).mean()
    if kind == "mse":
        return lambda model, b: ((model(b["node"], None, b["pos"], b["grid"])["preds"] - b["target"]) ** 2).mean()
    gamma = 0.5                                           # examples/ex2_darcy.py default --gamma

    def wl2(model, b):
        out = model(b["node"], None, b["pos"], b["grid"])["preds"]
        lf = WeightedL2Loss2d(regularizer=True, h=1.0 / b["target"].shape[1], gamma=gamma)
        l, r, _ = lf.terms(out[..., 0], b["target"][..., 0], targets_prime=b["target_grad"], K=b["coeff"])
        return lf.reduce(l) + lf.reduce(r)
    return wl2


class Trainer:
    """fwd+loss+bwd(+grad gather) | all-reduce | clip+Adam, each compute leg captured in a HIP graph."""

    def __init__(self, model, batch, world, lr=1e-3, clip=0.99, use_graph=True, workload="ex2_darcy141", loss="mse",
                 optimizer="flat"):
        import galerkin_transformer as gt
        from galerkin_transformer import _hip
        self._hip = _hip
        self.model, self.world, self.clip = model, world, clip
        self.batch = batch if isinstance(batch, dict) else dict(zip(("node", "pos", "grid", "target"), batch))
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.dev = self.batch["node"].device
        self.loss_fn = make_loss(workload, loss)
        self.flat = optimizer == "flat"
        if self.flat:
            self.opt = gt.FlatClipAdam(self.params, lr=lr, max_norm=clip, model=model)
            self.opt_kind = "FlatClipAdam (gt_grad_sqnorm + gt_adam_clip_step on one flat bucket)"
            self.reducer = None
        else:
            self.opt = torch.optim.Adam(self.params, lr=lr, capturable=True, fused=True)
            self.opt_kind = "torch clip_grad_norm_(foreach) + adam(fused,capturable)"
            from galerkin_transformer.distributed import FlatGradAllReducer
            self.reducer = FlatGradAllReducer(self.params)
        self.loss = torch.zeros((), device=self.dev)
        self.comm_enabled = True
        self.g_fb = self.g_opt = None
        self.use_graph = use_graph
        self.comm_in_graph = False           # ask for the collective inside the step graph (capture(); falls back)
        self.one_graph = False               # ... and whether the current capture is that form
        self.comm_events = None              # list: step() appends an event pair around every host-issued collective

    def fwd_bwd(self):
        self._hip.weight_packs.refresh()         # every weight of the step packed in ONE launch (they changed in opt_step)
        self._hip.advance_seed(self.dev)
        loss = self.loss_fn(self.model, self.batch)
        loss.backward()
        self.loss.copy_(loss.detach())
        if self.flat:
            self.opt.gather_grads()          # grads -> the flat bucket the collective and the optimizer work on

    def comm(self):
        if not self.comm_enabled:            # rank-local profiling step (roofline leg): no collective
            return
        if self.flat:
            self.opt.all_reduce()            # ONE flat RCCL sum-all-reduce per step, in place, no copy back
        else:
            self.reducer.reduce()

    def opt_step(self):
        self._hip.weight_packs.invalidate()      # the weights are about to change
        if self.flat:
            self.opt.apply()
        else:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip, foreach=True)
            self.opt.step()

    def eager_step(self):
        for p in self.params:
            p.grad = None
        self.fwd_bwd()
        self.comm()
        self.opt_step()

    _warm_stream = None                                  # one warm-up stream per process (scratch is kept per stream)

    def capture(self, warm=3):
        if Trainer._warm_stream is None:
            Trainer._warm_stream = torch.cuda.Stream()
        s = Trainer._warm_stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warm):
                self.eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if not self.use_graph:
            return False
        self.one_graph = False
        if (self.world > 1 and self.comm_in_graph and self.flat and self.comm_enabled
                and dist.get_backend() == "nccl"):       # (a gloo collective is a host copy: not capturable, and not recoverable)
            # the RCCL all-reduce captured between the two compute legs: ONE replay per step, no host round trip between
            # backward and optimizer.  Any failure (a backend that cannot be captured -- gloo --, an
            # RCCL build without graph support) leaves the two-graph form below.
            try:
                for p in self.params:
                    p.grad = None
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.fwd_bwd()
                    self.comm()
                    self.opt_step()
                torch.cuda.synchronize()
                self.g_fb, self.g_opt, self.one_graph = g, None, True
                return True
            except Exception as e:
                print(f"[bench] collective-in-graph capture failed ({type(e).__name__}: {e}); two graphs", file=sys.stderr)
                self.g_fb = self.g_opt = None
                torch.cuda.synchronize()
        try:
            for p in self.params:
                p.grad = None
            # with a process group alive its watchdog thread touches the runtime (event queries) while this thread
            # captures: "thread_local" keeps those calls from invalidating the capture
            mode = dict(capture_error_mode="thread_local") if self.world > 1 else {}
            self.g_fb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fb, **mode):
                self.fwd_bwd()
                if self.world == 1:
                    self.opt_step()
            if self.world > 1:
                self.g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_opt, pool=self.g_fb.pool(), **mode):
                    self.opt_step()
            torch.cuda.synchronize()
            return True
        except Exception as e:                           # graph capture unavailable: stay eager
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            self.g_fb = self.g_opt = None
            torch.cuda.synchronize()
            return False

    def step(self):
        if self.g_fb is None:
            self.eager_step()
            return
        self.g_fb.replay()
        if self.world > 1 and not self.one_graph:
            if self.comm_events is not None and self.comm_enabled:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.comm()
                e1.record()
                self.comm_events.append((e0, e1))
            else:
                self.comm()
            self.g_opt.replay()


def timed_run(tr, steps, warmup, world, comm_rec=None):
    """comm_rec (dict, N > 1): filled with the device time of the timed steps' host-issued collectives (event pairs on the
    stream the step runs on)."""
    for _ in range(warmup):
        tr.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tr.comm_events = [] if (comm_rec is not None and world > 1) else None
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if tr.comm_events is not None:           # (read back outside the timed region)
        ev, tr.comm_events = tr.comm_events, None
        us = [a.elapsed_time(b) * 1e3 for a, b in ev]
        comm_rec.update(allreduce_us_per_step=(round(sum(us) / len(us), 1) if us else None), timed_collectives=len(us))
    if world > 1:
        t = torch.tensor([elapsed], device=tr.dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


FOURIER_KEYS = ("gt_fourier16_attn", "gt_fourier_attn")
FFN_ENTRY_KEYS = ("gt_ffn_fwd", "gt_ffn_bwd")
FFN_KEY = "ffn_fwd16_kernel<true>"      # the kernel symbol both entry points launch (gt_ffn.hip)
PROFILE_ROUND = "r07"               # prefix of the profiles/ records this line quotes (the measurement pass of build round 6)


def roofline_leg(trainer, precision, workload="ex2_darcy141"):
    """Per-launch HIP-event timing of one eager step, then the dominant hot-path kernel re-timed back-to-back on the
    launch stream; plus the per-leg table (head norm, K^T V, FFN, Q.P) the north star asks for."""
    from galerkin_transformer import _hip
    trainer.comm_enabled = False             # only rank 0 runs this leg: its profiled step must not enter a collective
    try:
        with _hip.Profile() as prof:
            trainer.eager_step()
    finally:
        trainer.comm_enabled = True
    torch.cuda.synchronize()
    table = prof.table()
    if not table:
        return None, {}
    trainer.shape_table = prof.table(by_shape=True)
    # dominant kernel of the hand-written path = the GEMM template instance with the largest share of
    # the step; within it, the launch shape that accounts for most of that time
    # (the fused FeedForward launches -- forward and data half of the backward -- are ONE kernel symbol, the two entry points
    # its two launch kinds: counted together, as a profiler's per-kernel table does)
    total_ms = sum(v["ms"] for v in table.values())
    table = dict(table)
    if any(k in table for k in FFN_ENTRY_KEYS):
        table[FFN_KEY] = {f: sum(table[k][f] for k in FFN_ENTRY_KEYS if k in table) for f in ("calls", "ms", "flops", "bytes")}
    gemm_keys = [k for k in table if k.startswith("gemm_") or k in FOURIER_KEYS or k == FFN_KEY]
    dom = max(gemm_keys, key=lambda k: table[k]["ms"])
    recs = [r for r in prof.records if (r[0] == dom or (dom == FFN_KEY and r[0] in FFN_ENTRY_KEYS))
            and (r[5] is not None or dom in FOURIER_KEYS)]
    by_shape, n_shape = {}, {}
    for r in recs:
        by_shape[r[6]] = by_shape.get(r[6], 0.0) + r[3].elapsed_time(r[4])
        n_shape[r[6]] = n_shape.get(r[6], 0) + 1
    top_shape = max(by_shape, key=by_shape.get)
    best = next(r for r in recs if r[6] == top_shape)
    # duration of that launch INSIDE the step (HIP events around each launch of the profiled eager step, on the launch
    # stream): operands come from wherever the producing kernel left them (L2 / Infinity Cache / HBM), as in the timed
    # region.  The same launch replayed back-to-back on a cold 0.6 GB working set is reported next to it.
    dur_s = by_shape[top_shape] / n_shape[top_shape] * 1e-3
    # the launches of one output shape differ in their epilogue operands (FFN1 forward: A, B, C; the hidden-gradient launch
    # of the same shape also reads the saved activation as `aux`): time AND algorithmic bytes are averaged over the same
    # launches, and the kinds are listed (the counter passes average over the same set)
    same = [r for r in recs if r[6] == top_shape]
    mean_flops = sum(r[1] for r in same) / len(same)
    mean_bytes = sum(r[2] for r in same) / len(same)
    kinds = {}
    for r in same:
        k = kinds.setdefault(int(r[2]), [0, 0.0])
        k[0] += 1
        k[1] += r[3].elapsed_time(r[4])
    launch_kinds = [dict(algorithmic_bytes=b, launches=n, avg_launch_us=round(ms / n * 1e3, 2),
                         hbm_gbs=round(b / (ms / n * 1e-3) / 1e9, 1)) for b, (n, ms) in sorted(kinds.items())]
    replay_s = None
    if best[5] is not None:
        call, _keep = best[5]
        reps = 50
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        replay_s = e0.elapsed_time(e1) / reps * 1e-3
    x3 = "x3" in dom or dom == "gt_fourier16_attn" or dom == FFN_KEY
    products = (3 if dom == "gt_fourier16_attn" else PLANE_PRODUCTS.get(precision, 1)) if x3 else 1
    peak = PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS
    useful = mean_flops / dur_s / 1e12                   # 2 M N K per launch: what the caller asked for
    executed = useful * products                         # flops the matrix pipe actually retires
    hbm = mean_bytes / dur_s / 1e9
    # which roofline bounds this launch: the larger of (executed flops / MFMA peak, bytes / HBM peak)
    bound = "mfma" if executed / peak >= hbm / PEAK_HBM_GBS else "hbm"
    # HBM traffic of this kernel + launch shape from the rocprofv3 --pmc passes (tools/gpu_pmc.sh), when a
    # measurement for exactly this launch is on file under profiles/
    traffic, tsrc = None, None
    kernel_symbol = "gt::" + dom.replace("+splitk", "")
    pmc_file = "pmc_step.json" if workload == "ex2_darcy141" else f"pmc_step_{workload}.json"
    try:
        with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
            pj = json.load(f)
        rec, grid = None, None
        if dom in FOURIER_KEYS:
            # the template instance of this head-tile width and pass kind (single score / dual), whatever its mask mode
            want = f"{'fourier16' if dom == 'gt_fourier16_attn' else 'fourier_core'}_kernel<"
            dual = "true" if best[6][4] == 2 else "false"
            cands = [k for k in pj if k.startswith("gt::" + want) and f", {dual}," in k]
            if cands:
                kernel_symbol = max(cands, key=lambda k: pj[k].get("calls_seen", 0))
                rec = pj[kernel_symbol]
        elif dom == FFN_KEY:                      # 256 threads per 64 token rows
            grid = -(-best[6][0] // 64) * 256
            rec = pj.get("_by_grid", {}).get(f"{kernel_symbol}|{grid}")
        elif x3_name(dom):
            # one launch geometry of the symbol: the split-operand kernels run 256 threads per 128 x 128 output tile
            Ms, Ns = best[6][0], best[6][1]
            tiles = -(-Ms // 128) * -(-Ns // 128)
            grid = tiles * 256 * best[6][3]
            rec = pj.get("_by_grid", {}).get(f"gt::{dom.replace('+splitk', '')}|{grid}")
            if rec is None and dom.endswith("+splitk"):
                # split-K launch: tiles x K chunks workgroups, the chunk count is the library's choice -- the ONE launch
                # geometry of this symbol in the profiled steps whose workgroup count is a multiple of this tile count and
                # of no other tile count the symbol ran with
                others = {-(-r[6][0] // 128) * -(-r[6][1] // 128) for r in recs} - {tiles}
                cands = []
                for k in pj.get("_by_grid", {}):
                    if k.startswith(kernel_symbol + "|"):
                        wg = int(k.rsplit("|", 1)[1]) // 256
                        if wg % tiles == 0 and not any(wg % o == 0 for o in others if o > tiles or tiles % o):
                            cands.append(k)
                if len(cands) == 1:
                    grid = int(cands[0].rsplit("|", 1)[1])
                    rec = pj["_by_grid"][cands[0]]
        if rec is None and kernel_symbol in pj and len([k for k in pj.get("_by_grid", {}) if k.startswith(kernel_symbol + "|")]) == 1:
            rec = pj[kernel_symbol]                  # a symbol with ONE launch geometry in the profiled steps: its average is this launch
        if rec and "read_bytes" in rec and "write_bytes" in rec:
            traffic = int(rec["read_bytes"] + rec["write_bytes"])
            tsrc = (pj.get("_source", "") + f"; the {rec['calls_seen']} launches of this symbol" +
                    (f" with grid size {grid}" if grid else "") + " (every launch of this shape in the profiled steps, averaged)")
    except (OSError, ValueError):
        pass
    roof = dict(bound=bound, kernel=kernel_symbol, launch_shape_MNKb=list(best[6]),
                includes_splitk_reduce=dom.endswith("+splitk"), launches_per_step=len(recs),
                share_of_hip_path=round(table[dom]["ms"] / total_ms, 3),
                avg_launch_us=round(dur_s * 1e6, 2), launches_of_this_shape=n_shape[top_shape],
                replay_back_to_back_us=(round(replay_s * 1e6, 2) if replay_s else None),
                # all launch shapes of this kernel symbol in one step (what a profiler's per-kernel average mixes)
                symbol_avg_launch_us_all_shapes=round(table[dom]["ms"] / table[dom]["calls"] * 1e3, 2),
                achieved=round(executed if bound == "mfma" else hbm, 2),
                peak=peak if bound == "mfma" else PEAK_HBM_GBS, unit="TFLOP/s" if bound == "mfma" else "GB/s",
                frac=round(executed / peak if bound == "mfma" else hbm / PEAK_HBM_GBS, 4),
                mfma={"useful_tflops": round(useful, 2), "plane_products": products,
                      "executed_tflops": round(executed, 2), "peak_tflops": peak, "frac": round(executed / peak, 4),
                      "useful_vs_f32_mfma_peak": round(useful / PEAK_F32_MFMA_TFLOPS, 4)},
                hbm={"algorithmic_gbs": round(hbm, 1), "peak_gbs": PEAK_HBM_GBS, "frac": round(hbm / PEAK_HBM_GBS, 4)},
                traffic=traffic, traffic_source=tsrc, traffic_recorded_at=(_profiles_commit() if traffic else None),
                traffic_ratio=(round(traffic / mean_bytes, 3) if traffic else None),
                algorithmic_flops_per_launch=mean_flops, algorithmic_bytes_per_launch=mean_bytes,
                launch_kinds=launch_kinds,
                # the launch kind with the fewest operands on its own (FFN1 forward: A, B, C -- no saved activation read
                # back as a 1-bit-per-element ReLU decision): the fraction that does not credit that re-read as useful bytes
                frac_leanest_kind=(round(launch_kinds[0]["hbm_gbs"] / PEAK_HBM_GBS, 4) if bound == "hbm" and launch_kinds else None))
    # legs the north star names: live per-launch HIP-event time of this step + the HBM bytes / matrix-pipe busy cycles
    # of the same kernel symbol from the rocprofv3 --pmc passes on file (profiles/pmc_step.json: FETCH_SIZE doubled
    # for gfx950, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES in separate passes -- tools/gpu_measure.sh, tools/pmc_to_json.py)
    try:
        with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        pmc = {}
    if traffic and kernel_symbol in pmc and "mfma_util_at_2.4GHz" in pmc[kernel_symbol]:
        roof["mfma"]["busy_frac_profiled_pass"] = round(pmc[kernel_symbol]["mfma_util_at_2.4GHz"], 3)
    legs = {}
    pk = "gemm_x3h_kernel" if precision == "f16x2" else "gemm_x3p_kernel"     # the packed-B instances of the arithmetic
    shape_rows = prof.table(by_shape=True)
    for label, key, sym, pred in LEGS:
        key, sym = key.replace("gemm_x3p_kernel", pk), sym.replace("gemm_x3p_kernel", pk)
        leg = leg_record(table, shape_rows, pmc, key, sym, pred)
        if leg:
            legs[label] = leg
    roof["legs"] = legs
    return roof, table


def leg_record(table, shape_rows, pmc, key, sym, pred):
    """One roofline leg: mean HIP-event time of the launches of host entry `key` (only those whose recorded shape satisfies
    `pred`, when given) and the counter bytes of kernel symbol `sym` from profiles/pmc_step.json.  Time and bytes must be
    averages over THE SAME launches: a host entry that launches several kernel instances (gt_conv3x3_wgrad_nhwc: one
    template instance per channel geometry) is split into one leg per instance by `pred` -- round 4's single leg divided the
    widest instance's bytes by the mean time of four launches of three shapes."""
    if pred is None:
        t = table.get(key) or table.get(key.replace("+splitk", ""))
    else:
        t = dict(calls=0, ms=0.0, flops=0.0, bytes=0.0)
        for k, r in shape_rows.items():
            name, _, shp = k.partition(" (")
            if name in (key, key.replace("+splitk", "")) and shp and pred(tuple(int(v) for v in shp.rstrip(")").split(",") if v.strip())):
                for f in t:
                    t[f] += r[f]
    if not t or t["calls"] == 0 or t["ms"] <= 0:
        return None
    us = t["ms"] / t["calls"] * 1e3
    leg = dict(us=round(us, 1), launches=t["calls"], kernel=sym)
    if t["flops"] > 0:
        leg["useful_tflops"] = round(t["flops"] / (t["ms"] * 1e-3) / 1e12, 2)
    # roofline fraction against the ALGORITHMIC bytes of the launches (what the operator has to move: operands + result, each
    # once); the counter bytes of the same symbol (what it did move) and their ratio are carried separately -- a ratio well
    # above 1 is re-read traffic, not bandwidth achieved
    alg = t["bytes"] / t["calls"] if t.get("bytes", 0) > 0 else None
    if alg:
        leg.update(algorithmic_bytes_per_launch=int(alg), hbm_gbs_algorithmic=round(alg / us / 1e3, 1),
                   hbm_frac=round(alg / us / 1e3 / PEAK_HBM_GBS, 4))
    c = pmc.get(sym)
    if c and "read_bytes" in c and "write_bytes" in c:
        by = c["read_bytes"] + c["write_bytes"]
        leg.update(counter_bytes_per_launch=int(by), counter_gbs=round(by / us / 1e3, 1),
                   counter_over_algorithmic=(round(by / alg, 3) if alg else None), counter_launches_seen=c.get("calls_seen"))
        if not alg:
            leg["hbm_frac"] = round(by / us / 1e3 / PEAK_HBM_GBS, 4)
        if "mfma_util_at_2.4GHz" in c:
            leg["mfma_busy_frac_profiled_pass"] = round(c["mfma_util_at_2.4GHz"], 3)
    return leg


# roofline legs: label, key of the HIP-event table (_hip.Profile), kernel symbol in profiles/pmc_step.json, and -- where one
# host entry launches several kernel instances -- a predicate on the launch's recorded shape that selects the launches of
# THAT symbol (tests/test_host_cpu.py: every symbol exists in the counter file, every hbm_gbs of the committed line is
# recomputed from the counter file and the per-shape event table)
LEGS = (("qkv_proj+headnorm_fwd", "gemm_x3p_kernel<0, 32, 0, 128>", "gt::gemm_x3p_kernel<0, 32, 0, 128>", None),
        ("headnorm_fwd", "gt_headnorm_fwd", "gt::headnorm_fwd_v2_kernel", None),
        ("headnorm_bwd", "gt_headnorm_bwd", "gt::headnorm_bwd_v2_kernel", None),
        ("galerkin_ktv", "gt_galerkin_ktv", "gt::galerkin_ktv_lds_kernel<2>", None),
        ("galerkin_dkv", "gt_galerkin_dkv", "gt::galerkin_dkv_kernel<2>", None),
        ("galerkin_dkv+headnorm_bwd", "gt_galerkin_dkv_ln", "gt::galerkin_dkv_ln_kernel<2, true>", None),
        ("galerkin_qp(Q'.P with fc folded)", "gemm_x3r_kernel<0, 1, 3, 3, 0, 0>", "gt::gemm_x3r_kernel<0, 1, 3, 3, 0, 0>", None),
        ("feed_forward fused (forward, data half of the backward)", FFN_KEY, "gt::" + FFN_KEY, None),
        ("token_gemms(packed B)", "gemm_x3p_kernel<0, 0, 0, 128>", "gt::gemm_x3p_kernel<0, 0, 0, 128>", None),
        ("conv3x3_implicit", "gemm_x3p_kernel<0, 0, 1, 128>", "gt::gemm_x3p_kernel<0, 0, 1, 128>", None),
        ("conv3x3_implicit_narrow(down-scaler)", "gemm_x3p_kernel<0, 0, 1, 64>", "gt::gemm_x3p_kernel<0, 0, 1, 64>", None),
        # gt_conv3x3_wgrad_nhwc, shape (B, H, W, Cin, Cout): one kernel instance per channel geometry
        ("conv3x3_wgrad 128->128 (up-scaler)", "gt_conv3x3_wgrad_nhwc", "gt::convw_kernel<4, 2, 1>", lambda sh: sh[3:5] == (128, 128)),
        ("conv3x3_wgrad 128->48 (down-scaler)", "gt_conv3x3_wgrad_nhwc", "gt::convw_kernel<4, 3, 1>", lambda sh: sh[3:5] == (128, 48)),
        ("conv3x3_wgrad 48->48 (down-scaler)", "gt_conv3x3_wgrad_nhwc", "gt::convw_kernel<3, 3, 1>", lambda sh: sh[3:5] == (48, 48)),
        ("weight_gradients(f16x2 planes in LDS)", "gemm_x3w_kernel<2>+splitk", "gt::gemm_x3w_kernel<2>", None),
        ("weight_gradients(ring)", "gemm_x3r_kernel<1, 1, 3, 3, 0, 0>+splitk", "gt::gemm_x3r_kernel<1, 1, 3, 3, 0, 0>", None),
        ("fourier_attention(f16x2)", "gt_fourier16_attn", "gt::fourier16_kernel<36, false, 4, 8>", lambda sh: sh[4] == 1),
        ("fourier_attention_dual(f16x2)", "gt_fourier16_attn", "gt::fourier16_kernel<36, true, 4, 4>", lambda sh: sh[4] == 2))


def x3_name(key: str) -> bool:
    return key.startswith("gemm_x3")


def cpu_baseline_leg(model_cpu_sd, cfg, budget_s=20.0):
    """CPU training step on this host's cores, B=4 (the reference default batch): 1 warm-up + as many steps as fit
    the budget (>= 3).  kind "reference" = the reference's own modules (only where /root/reference exists, i.e. the
    build container); kind "port" = oracle/galerkin_oracle.py, the pinned CPU restatement -- its step is slightly
    LIGHTER than the reference's (it treats every nn.Dropout as identity; the always-on attention dropout is kept)."""
    B = 4
    # torch's CPU kernels stop scaling (and then regress) well below the core count of a GPU host:
    # cap the thread pool; "cores" in the output is the number of threads actually used
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    b = synthetic_batch(B, torch.device("cpu"), seed=7)
    node, pos, grid, target = b["node"], b["pos"], b["grid"], b["target"]
    kind = "port"
    if os.path.isdir("/root/reference/libs"):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
            from make_golden import import_reference
            _, M, _ = import_reference()
            mc = dict(cfg)
            model = M.FourierTransformer2D(**mc).train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)

            def step():
                opt.zero_grad()
                out = model(node, None, pos, grid)["preds"]
                ((out - target) ** 2).mean().backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), 0.99)
                opt.step()
            kind = "reference"
        except Exception as e:                                        # fall back to the port
            print(f"[bench] reference import failed ({type(e).__name__}: {e}); timing the oracle", file=sys.stderr)
            kind = "port"
    if kind == "port":
        from oracle import galerkin_oracle as O
        sd = {k: v.detach().clone().float().requires_grad_(v.is_floating_point()) for k, v in model_cpu_sd.items()}
        state = {}
        step = lambda: O.model_train_step_cpu(sd, cfg, node, pos, grid, target, state)
    step()
    t0, n = time.perf_counter(), 0
    while n < 3 or (time.perf_counter() - t0 < budget_s and n < 40):
        step()
        n += 1
    dt = time.perf_counter() - t0
    src = ("the reference's FourierTransformer2D imported from /root/reference" if kind == "reference" else
           "oracle/galerkin_oracle.py (every nn.Dropout = identity, attention dropout kept: a slightly lighter step "
           "than the reference's)")
    return dict(value=round(B * n / dt, 3), unit="samples/s", cores=torch.get_num_threads(), kind=kind,
                sample=f"{n} steps of batch {B} (fwd+MSE+bwd+clip+Adam, {src}), {dt:.1f} s of CPU time")


def accuracy_leg(precision, n_seeds=10, first_seed=1000):
    """Second half of the metric: validation rel-L2 after the short synthetic-Darcy training run of
    tools/accuracy_leg.py on this GPU, over `n_seeds` dropout seeds, next to the reference's own CPU runs of the same
    recipe, data, initial weights and dropout seeds (profiles/accuracy_reference_cpu_seeds.json: twelve seeds recorded in
    the build container, where /root/reference exists).  The 128-step run peaks at lr 1e-3 and is noise-sensitive --
    individual runs land between 0.08 and 0.26 for BOTH implementations -- so the two samples are compared as
    distributions: mean +- standard error and Welch's two-sample t-test."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import accuracy_leg as AL
    import accuracy_seeds as AS
    seeds = [first_seed + i for i in range(n_seeds)]
    runs = [AL.run("hip", dropout_seed=sd) for sd in seeds]
    hs = AS.summarize([r["val_rel_l2"] for r in runs])
    out = {"metric": "validation relative L2 error after %d epochs (%d steps of batch %d) on the synthetic Darcy set"
                     % (runs[0]["epochs"], runs[0]["steps"], runs[0]["batch"]),
           "hip": {"val_rel_l2_mean": round(hs["mean"], 5), "val_rel_l2_sem": round(hs["sem"], 5),
                   "val_rel_l2_std": round(hs["std"], 5), "val_rel_l2_runs": [round(v, 5) for v in hs["values"]],
                   "dropout_seeds": seeds, "seconds_per_run": runs[0]["seconds"], "precision": precision},
           "reference_cpu": None, "welch_t_test": None, "data": runs[0]["data"]}
    try:
        with open(os.path.join(ROOT, "profiles", "accuracy_reference_cpu_seeds.json")) as f:
            ref = json.load(f)
        rs = AS.summarize([r["val_rel_l2"] for r in ref["runs"]])
        out["reference_cpu"] = {"val_rel_l2_mean": round(rs["mean"], 5), "val_rel_l2_sem": round(rs["sem"], 5),
                                "val_rel_l2_std": round(rs["std"], 5), "val_rel_l2_runs": [round(v, 5) for v in rs["values"]],
                                "dropout_seeds": [r["dropout_seed"] for r in ref["runs"]],
                                "seconds_per_run": ref["runs"][0]["seconds"],
                                "source": "profiles/accuracy_reference_cpu_seeds.json (tools/accuracy_seeds.py --impl reference)"}
        w = AS.welch(hs, rs)
        p2 = w["p_two_sided"]
        reading = ("p > 0.05: the two error distributions are statistically indistinguishable" if p2 > 0.05 else
                   "p <= 0.05 on this draw of seeds: the HIP runs' mean error is %s than the reference's (the spread between "
                   "seeds is 0.08 ... 0.26 for both; earlier rounds' draws gave p = 0.08 ... 0.24)"
                   % ("LOWER" if hs["mean"] < rs["mean"] else "HIGHER"))
        out["welch_t_test"] = {"t": round(w["t"], 3), "df": round(w["df"], 1), "p_two_sided": round(p2, 4), "reading": reading}
    except (OSError, ValueError, KeyError):
        pass
    return out


def source_hash():
    """sha256 (16 hex digits) over the product sources -- csrc/*.hip, *.h, include/*.h, galerkin_transformer/*.py -- the
    staleness key of the quoted records: .git does not travel to the GPU box, file contents do.  (bench.py itself is the
    measuring tool: a change of its reporting does not invalidate counter or parity records.)"""
    import glob
    import hashlib
    pkg = os.path.join(ROOT, "galerkin-transformer_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(pkg, "galerkin_transformer", "*.py")))
    hsh = hashlib.sha256()
    for fn in files:
        hsh.update(os.path.relpath(fn, ROOT).encode())
        with open(fn, "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()[:16]


def _profiles_commit():
    """Where the quoted profiles/ records come from (written by the measurement pass: profiles/SOURCE.json), and whether the
    product sources are still the ones they were measured with (quoted records went stale silently)."""
    try:
        with open(os.path.join(ROOT, "profiles", "SOURCE.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    if isinstance(rec, dict):
        rec = dict(rec)
        rec["sources_now_sha16"] = source_hash()
        rec["records_match_current_sources"] = rec.get("source_sha16") == rec["sources_now_sha16"]
    return rec


def parity_record():
    """The deterministic half of the accuracy metric, recorded by the GPU test-suite (tests/test_bench_kernels_gpu.py,
    tests/test_fullsize_models_gpu.py; committed under profiles/): whole-model prediction / gradient parity at full size
    and the 5-step training trajectory against the float64 oracle.  Quoted here, not recomputed (the bench does not run
    the checker).  gradients_rel_l2_max is the maximum over ALL parameters; the float32 oracle's own maximum distance from
    the float64 one (same replayed decisions) is given next to it -- the down-scaler filters dominate both (float32
    interpolation coordinates, see the test's docstring) and are listed separately."""
    out = {}
    for key, fn in (("trajectory_5_steps", PROFILE_ROUND + "_parity_trajectory.json"),
                    ("whole_model_replay", PROFILE_ROUND + "_parity_whole_model_replay_relu.json"),
                    ("whole_model_exact_math", PROFILE_ROUND + "_parity_whole_model_off_silu.json"),
                    ("full_size_C4_ex3_darcy_inv", PROFILE_ROUND + "_parity_whole_model_full_ex3_darcy_inv.json"),
                    ("full_size_C3_darcy211_fourier", PROFILE_ROUND + "_parity_whole_model_full_ex2_darcy211_fourier.json"),
                    ("full_size_C5_ns_rollout", PROFILE_ROUND + "_parity_whole_model_full_ex4_ns.json")):
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                r = json.load(f)
        except (OSError, ValueError):
            continue
        if key == "trajectory_5_steps":
            out[key] = {k: r[k] for k in ("steps", "batch", "loss_rel_err_max", "param_rel_l2_hip_vs_f64",
                                          "param_rel_l2_oracle_f32_vs_f64", "precision") if k in r}
        elif key.startswith("full_size"):
            out[key] = {k: r[k] for k in ("B", "prediction", "prediction_worst_step", "grad_max", "grad_max_oracle_f32",
                                          "precision", "replay_audit") if k in r}
        else:
            e, n = r.get("hip_vs_f64", {}), r.get("oracle_f32_vs_f64", {})
            if e:
                allv = {k: v for k, v in e.items() if k != "out"}
                enc = [v for k, v in allv.items() if not k.startswith("downscaler.")]
                ds = {k: v for k, v in allv.items() if k.startswith("downscaler.")}
                out[key] = {"prediction_rel_l2": e.get("out"), "gradients_rel_l2_max": max(allv.values()),
                            "gradients_rel_l2_max_oracle_f32": (max(v for k, v in n.items() if k != "out") if n else None),
                            "gradients_rel_l2_max_outside_downscaler": max(enc),
                            "gradients_rel_l2_median": sorted(allv.values())[len(allv) // 2],
                            "downscaler_filters_hip_vs_f64": {k.split(".")[2]: v for k, v in ds.items()},
                            "downscaler_filters_hip_vs_oracle_f32": {k.split(".")[2]: v for k, v in
                                                                     r.get("hip_vs_oracle_f32", {}).items() if k.startswith("downscaler.")},
                            "precision": r.get("precision"), "replay_audit": r.get("replay_audit")}
    out["source"] = (f"profiles/{PROFILE_ROUND}_parity_*.json, written by tests/test_bench_kernels_gpu.py / "
                     "test_fullsize_models_gpu.py on MI355X")
    out["recorded_at"] = _profiles_commit()
    return out if len(out) > 2 else None


DTYPE_TEXT = {
    "f32": "f32",
    "bf16x3": "f32 (operands split exactly into 3 bf16 terms, 6 plane products on the bf16 MFMA pipe, f32 accumulate; "
              "fp32-class results: the 1e-5 parity gate is tested in this mode)",
    "f16x2": "f32 (operands split into 2 fp16 terms under a running per-row / per-tile power-of-two scale, 3 products on the "
             "f16 MFMA pipe, f32 accumulate; fp32-class results: the 1e-5 parity suite passes in this mode; packed-B token GEMMs, "
             "convolutions, weight gradients, regression head and the Fourier attention run it, the batched per-sample products "
             "(gemm_x3r: Q'P, dQ, dP^T) run bf16x3 = 3 bf16 terms, 6 products)",
    "bf16x2": "f32 storage / bf16x2 split MFMA (~2^-16 relative; throughput mode, own gate)",
    "bf16": "f32 storage / bf16-rounded MFMA operands, f32 accumulate (throughput mode, own 3e-3 gate)",
}


def strong_leg(tr, G, per_gpu, steps, warmup, world, rank, workload, dev):
    """The same step at a FIXED global batch G split evenly over the ranks (north_star: strong scaling) -- a second timed
    region of the same process on a fresh synthetic batch of G / world samples per rank, same model and optimizer state
    (re-captured graphs).  At N = 1 this is the one-GPU time of G samples, the denominator of the strong-scaling speed-up."""
    old = tr.batch
    try:
        tr.g_fb = tr.g_opt = None
        torch.cuda.empty_cache()
        tr.batch = synthetic_batch(per_gpu, dev, seed=2000 + rank, workload=workload)
        graphed = tr.capture(warm=2)
        comm = {}
        e = timed_run(tr, steps, min(warmup, 3), world, comm_rec=comm)
        rec = {"global_batch": G, "per_gpu_batch": per_gpu, "value": round(G * steps / e, 2), "unit": "samples/s",
               "ms_per_step": round(e / steps * 1e3, 3), "steps": steps, "hip_graph": bool(graphed)}
        if comm:
            rec["allreduce_us_per_step"] = comm.get("allreduce_us_per_step")
        return rec
    finally:
        tr.batch = old
        tr.g_fb = tr.g_opt = None
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default per workload")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--global-batch", type=int, default=None, help="--scaling strong: fixed total batch, split evenly")
    ap.add_argument("--workload", default="ex2_darcy141", choices=sorted(WORKLOADS),
                    help="ex2_darcy141 is the headline metric; the others are informational")
    ap.add_argument("--loss", default="mse", choices=["mse", "weighted_l2"])
    ap.add_argument("--precision", default=None, choices=["f32", "bf16x3", "bf16x2", "bf16", "f16x2"],
                    help="arithmetic of the contractions (default: the library default, f16x2)")
    ap.add_argument("--optimizer", default="flat", choices=["flat", "torch"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets the "
                    "multi-rank path be exercised with several ranks on one GPU in tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the short convergence run (val rel-L2)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra timed run in the exact fp32 MFMA mode")
    ap.add_argument("--table", default=None, help="write the per-kernel event table to this JSON file")
    ap.add_argument("--strong-global-batch", default="512", help="comma list: global batches of the extra strong-scaling "
                    "legs every --scaling weak run adds to its line (each split evenly over the ranks; '' or 0: none)")
    ap.add_argument("--comm-in-graph", action="store_true", help="N > 1: capture the RCCL all-reduce inside the step graph "
                    "(one replay per step); falls back to the two-graph form when the capture fails")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    if a.scaling == "strong":
        gb = a.global_batch or DEFAULT_BATCH[a.workload] * 8
        if gb % world:
            raise SystemExit(f"--global-batch {gb} is not divisible by {world} ranks")
        per_gpu = gb // world
    else:
        per_gpu = a.batch or DEFAULT_BATCH[a.workload]
        gb = per_gpu * world

    import galerkin_transformer as gt
    from galerkin_transformer import _hip
    _hip.lib()                                           # fail loudly if the HIP library is missing
    if a.precision:
        gt.set_precision(a.precision)
    precision = gt.get_precision()
    torch.manual_seed(1127802)                           # identical init on every rank
    model, cfg = build_model(a.workload)
    cpu_sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    gt.set_attention_dropout("reference")
    from galerkin_transformer.distributed import rank_seed
    _hip.set_seed(rank_seed(1127802, rank), dev)         # per-rank dropout streams
    batch = synthetic_batch(per_gpu, dev, seed=1000 + rank, workload=a.workload)
    tr = Trainer(model, batch, world, use_graph=not a.no_graph, workload=a.workload, loss=a.loss,
                 optimizer=a.optimizer)
    tr.comm_in_graph = bool(a.comm_in_graph)
    graphed = tr.capture()
    comm = None
    if world > 1:
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)                            # every rank contributes 1: the ranks the collective really spans
        comm = {"bytes": 4 * sum(p.numel() for p in tr.params), "backend": a.backend, "ranks_seen": int(seen.item()),
                "collective": "one flat fp32 sum-all-reduce of the gradient bucket per step, 1/world folded into clip + Adam",
                "in_graph": bool(tr.one_graph)}
    elapsed = timed_run(tr, a.steps, a.warmup, world, comm_rec=comm)
    loss = float(tr.loss.item())

    # the same timed region once more in the bit-exact fp32 MFMA arithmetic (new captures, same weights trajectory)
    f32_leg = None
    if precision != "f32" and not a.no_f32_leg:
        gt.set_precision("f32")
        tr.g_fb = tr.g_opt = None
        g2 = tr.capture(warm=1)
        e2 = timed_run(tr, a.steps, min(a.warmup, 2), world)
        f32_leg = {"value": round(gb * a.steps / e2, 2), "ms_per_step": round(e2 / a.steps * 1e3, 3),
                   "hip_graph": bool(g2), "arithmetic": "v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 FMA chain)"}
        gt.set_precision(precision)

    roof, table = (None, {})
    if rank == 0 and not a.no_roofline:
        try:
            roof, table = roofline_leg(tr, precision, a.workload)
        except Exception as e:
            print(f"[bench] roofline leg failed: {type(e).__name__}: {e}", file=sys.stderr)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.workload == "ex2_darcy141":
        try:
            cpu = cpu_baseline_leg(cpu_sd, cfg)
            try:        # the reference implementation itself, timed where /root/reference exists (tools/cpu_baseline_reference.py)
                with open(os.path.join(ROOT, "profiles", "cpu_baseline_reference.json")) as f:
                    cpu["reference_in_build_container"] = json.load(f)
            except (OSError, ValueError):
                pass
        except Exception as e:
            print(f"[bench] cpu baseline failed: {type(e).__name__}: {e}", file=sys.stderr)
    acc = None
    if rank == 0 and world == 1 and not a.no_accuracy and a.workload == "ex2_darcy141":
        try:
            acc = accuracy_leg(precision)
        except Exception as e:
            print(f"[bench] accuracy leg failed: {type(e).__name__}: {e}", file=sys.stderr)
    if world > 1:
        dist.barrier()
    # strong-scaling legs (every rank takes part): the fixed global batches G next to the weak-scaling value of the line
    strong = []
    if a.scaling == "weak" and a.strong_global_batch not in ("", "0"):
        for G in [int(x) for x in a.strong_global_batch.split(",") if x.strip()]:
            if G <= 0 or G % world:
                continue
            try:
                strong.append(strong_leg(tr, G, G // world, min(a.steps, 10), a.warmup, world, rank, a.workload, dev))
            except Exception as e:                       # (out of memory at N = 1, ...): the line says so instead of dying
                print(f"[bench] strong leg G={G} failed: {type(e).__name__}: {e}", file=sys.stderr)
                strong.append({"global_batch": G, "per_gpu_batch": G // world, "value": None, "error": type(e).__name__})

    if rank == 0:
        value = gb * a.steps / elapsed
        out = {
            "metric": ("training samples/s, Darcy 141x141 Galerkin encoder (fwd+loss+bwd+clip+Adam)"
                       if a.workload == "ex2_darcy141" else
                       f"training samples/s, {a.workload} (informational, not the headline metric)"),
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": DTYPE_TEXT[precision],
            "data": "synthetic (randn fields of the workload's shapes on the true grids, random init)",
            "config": {"workload": WORKLOAD_TEXT[a.workload],
                       "params": sum(p.numel() for p in model.parameters()),
                       "global_batch": gb, "per_gpu_batch": per_gpu, "parallelism": f"dp{world}",
                       "hip_graph": bool(graphed), "optimizer": tr.opt_kind, "loss": a.loss, "precision": precision,
                       "dropout": "config.yml (train mode) + reference attention dropout p=0.5",
                       "final_loss": round(loss, 6)},
            "f32_mfma_exact": f32_leg,
            # algorithmic work of the whole step (19.3 GFLOP per sample fwd + bwd) over the step time
            "useful_tflops": (round(19.3e9 * gb / (elapsed / a.steps) / 1e12, 2) if a.workload == "ex2_darcy141" else None),
            "roofline": roof, "cpu_baseline": cpu, "parity": parity_record(), "accuracy": acc,
            # N > 1: where the exchange step's time went; every N: the same step at fixed global batches (strong scaling:
            # speed-up at N GPUs = strong[i].value of the N-GPU line / strong[i].value of the 1-GPU line)
            "comm": comm, "strong": strong or None,
        }
        if a.table and table:
            with open(a.table, "w") as f:
                json.dump({"by_kernel": dict(sorted(table.items(), key=lambda kv: -kv[1]["ms"])),
                           "by_shape": dict(sorted(getattr(tr, "shape_table", {}).items(),
                                                   key=lambda kv: -kv[1]["ms"]))}, f, indent=1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
