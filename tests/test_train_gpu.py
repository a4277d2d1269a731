"""End-to-end (-m gpu): the reference-style training driver on synthetic Darcy data through the HIP
path -- loss goes down, the best checkpoint round-trips through state_dict, and a HIP-graph-captured
step reproduces the eager step bit-for-bit."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small_cfg(gt, n_f, n_c):
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "galerkin-transformer_amd", "config.yml")) as f:
        cfg = yaml.full_load(f)["ex2_darcy"]
    down, up = gt.DarcyDataset.get_scaler_sizes(n_f, n_c, scale_factor=False)
    cfg.update(n_hidden=32, n_head=2, dim_feedforward=64, num_encoder_layers=2, freq_dim=16, fourier_modes=4,
               downscaler_size=down, upscaler_size=up, norm_eps=1e-7)
    return cfg


def test_run_train_synthetic_darcy(gpu_device, tmp_path):
    import galerkin_transformer as gt
    from libs import (DarcyDataset, DataLoader, OneCycleLR, WeightedL2Loss2d, get_seed, run_train,
                      train_batch_darcy, validate_epoch_darcy)
    get_seed(1127802, printout=False)
    kw = dict(subsample_attn=30, subsample_nodes=10, synthetic=True, n_samples_synthetic=40)
    train = DarcyDataset(train_data=True, train_len=32, **kw)
    valid = DarcyDataset(train_data=False, valid_len=8, normalizer_x=train.normalizer_x, **kw)
    n_f, n_c = train.n_f, train.n_grid
    cfg = _small_cfg(gt, n_f, n_c)
    cfg["normalizer"] = train.normalizer_y.to(gpu_device)
    model = gt.FourierTransformer2D(**cfg).to(gpu_device)
    tl = DataLoader(train, batch_size=4, shuffle=True, drop_last=True)
    vl = DataLoader(valid, batch_size=4, shuffle=False)
    epochs = 6
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    sched = OneCycleLR(opt, max_lr=2e-3, div_factor=1e2, final_div_factor=1e2, pct_start=0.3,
                       steps_per_epoch=len(tl), epochs=epochs)
    h = 1 / n_f
    res = run_train(model, WeightedL2Loss2d(regularizer=True, h=h, gamma=0.5),
                    WeightedL2Loss2d(regularizer=False, h=h), tl, vl, opt, sched,
                    train_batch=train_batch_darcy, validate_epoch=validate_epoch_darcy, epochs=epochs,
                    patience=None, tqdm_mode='epoch', model_name='m.pt', result_name='r.pkl',
                    model_save_path=str(tmp_path), device=gpu_device)
    lt = res["loss_train"][:, 0]
    assert np.all(np.isfinite(lt)) and lt[-1] < 0.8 * lt[0]
    fresh = gt.FourierTransformer2D(**cfg).to(gpu_device)
    fresh.load_state_dict(torch.load(os.path.join(str(tmp_path), "m.pt")))
    m = validate_epoch_darcy(fresh, WeightedL2Loss2d(regularizer=False, h=h), vl, gpu_device)["metric"]
    assert np.isfinite(m)


def test_run_train_ns_rollout_device_loader_flat_adam(gpu_device, tmp_path):
    """ex4 flow end to end (examples/ex4_navier_stokes_2+1d.py:9-106 at reduced size): synthetic NS trajectories ->
    DeviceResidentLoader (dataset uploaded once) -> FourierTransformer2DLite -> FlatClipAdam + OneCycleLR ->
    run_train with train_batch_ns / validate_epoch_ns (10-step rollout, one backward)."""
    import galerkin_transformer as gt
    from libs import OneCycleLR, WeightedL2Loss2d, get_seed, run_train
    from libs.ns_lite import (DeviceResidentLoader, FourierTransformer2DLite, NavierStokesDatasetLite, train_batch_ns,
                              validate_epoch_ns)
    get_seed(1127802, printout=False)
    train = NavierStokesDatasetLite(train_data=True, train_len=16, valid_len=4, synthetic_len=20)
    valid = NavierStokesDatasetLite(train_data=False, train_len=16, valid_len=4, synthetic_len=20)
    cfg = dict(node_feats=12, pos_dim=2, n_targets=1, n_hidden=24, num_encoder_layers=2, n_head=1, dim_feedforward=48,
               attention_type="galerkin", layer_norm=True, attn_norm=False, xavier_init=0.01, diagonal_weight=0.01,
               encoder_dropout=0.0, ffn_dropout=0.05, dropout=0.0, decoder_dropout=0.0, decoder_type="ifft2",
               freq_dim=12, num_regressor_layers=2, fourier_modes=8, spacial_dim=2, spacial_fc=False, debug=False)
    model = FourierTransformer2DLite(**cfg).to(gpu_device)
    tl = DeviceResidentLoader(train, 4, gpu_device, shuffle=True, drop_last=True)
    vl = DeviceResidentLoader(valid, 4, gpu_device)
    assert next(iter(tl))["node"].is_cuda and len(tl) == 4
    epochs = 5
    opt = gt.FlatClipAdam(model.parameters(), lr=2e-3, max_norm=0.99)
    sched = OneCycleLR(opt, max_lr=2e-3, div_factor=1e2, final_div_factor=1e2, steps_per_epoch=len(tl), epochs=epochs)
    res = run_train(model, WeightedL2Loss2d(regularizer=True, h=1 / 64, gamma=0.1),
                    WeightedL2Loss2d(regularizer=False, h=1 / 64), tl, vl, opt, sched, train_batch=train_batch_ns,
                    validate_epoch=validate_epoch_ns, epochs=epochs, patience=None, tqdm_mode='epoch',
                    model_name='ns.pt', result_name='ns.pkl', model_save_path=str(tmp_path), device=gpu_device)
    lt = res["loss_train"][:, 0]
    assert np.all(np.isfinite(lt)) and lt[-1] < lt[0]        # 20 steps of a tiny model: the objective moves down
    assert np.isfinite(res["best_val_metric"])


def test_packed_parameter_views_equal_the_copies(gpu_device):
    """FlatClipAdam(model=...) lays SimpleAttention's pack groups out back to back, so the packed QKV weight / bias and the
    per-head LayerNorm parameters are VIEWS of the bucket (no torch.cat / torch.stack launches); without `model` they are
    copies.  Both must train identically, and the views must really alias the parameters."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import galerkin_transformer as gt
    from galerkin_transformer import _hip
    outs = []
    for with_model in (False, True):
        torch.manual_seed(3)
        model = gt.FourierTransformer2D(**bench.darcy_config()).to(gpu_device).train()
        gt.set_attention_dropout("reference")
        batch = bench.synthetic_batch(2, gpu_device, 5)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = gt.FlatClipAdam(params, lr=1e-3, max_norm=0.99, model=model if with_model else None)
        attn = model.encoder_layers[0].attn
        wqkv, bqkv, gamma, beta, _ = attn._packed()
        aliased = wqkv.data_ptr() == attn.linears[0].weight.data_ptr()
        assert aliased == with_model
        if with_model:
            assert bqkv.data_ptr() == attn.linears[0].bias.data_ptr() and gamma.data_ptr() == attn.norm_K[0].weight.data_ptr()
            assert torch.equal(wqkv, torch.cat([l.weight for l in attn.linears])) and tuple(gamma.shape) == (2, attn.n_head, attn.d_k)
            assert torch.equal(beta.reshape(-1), torch.cat([m.bias for m in list(attn.norm_K) + list(attn.norm_V)]))
        _hip.set_seed(50, gpu_device)
        _hip._salt[0] = 7
        for _ in range(2):
            for p in params:
                p.grad = None
            _hip.advance_seed(gpu_device)
            out = model(batch["node"], None, batch["pos"], batch["grid"])["preds"]
            ((out - batch["target"]) ** 2).mean().backward()
            opt.step()
        torch.cuda.synchronize()
        outs.append([p.detach().clone() for p in model.parameters()])
    worst = max(float((a - b).abs().max()) for a, b in zip(*outs))
    assert worst < 1e-6, worst


def test_weights_packed_once_per_step_equal_per_call_packs(gpu_device):
    """_hip.weight_packs: from the third step on the Trainer packs every Linear weight of the step in ONE launch
    (gt_gemm_pack_b_many) and the token products take the buffers through gt_gemm_desc.b_packed; the parameters after five
    steps must equal those of the per-call packs (GT_PACK_ALL=0 behaviour) bit for bit, and the per-call pack launches of
    those products must be gone from the step."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import galerkin_transformer as gt
    from galerkin_transformer import _hip
    if gt.get_precision() != "f16x2":
        pytest.skip("weights are packed ahead in the two-term fp16 arithmetic only")
    outs, packs = [], []
    try:
        for enabled in (False, True):
            _hip.weight_packs.clear()
            _hip.weight_packs.enabled = enabled
            torch.manual_seed(3)
            model = gt.FourierTransformer2D(**bench.darcy_config()).to(gpu_device).train()
            gt.set_attention_dropout("reference")
            batch = bench.synthetic_batch(16, gpu_device, 5)           # 16 x 1849 token rows: the packed-B kernels
            tr = bench.Trainer(model, batch, 1, use_graph=False)
            _hip.set_seed(50, gpu_device)
            _hip._salt[0] = 7
            for _ in range(4):
                tr.eager_step()
            with _hip.Profile() as prof:
                tr.eager_step()
            torch.cuda.synchronize()
            t = prof.table()
            packs.append((t.get("gt_gemm_pack_b_many", {}).get("calls", 0),
                          sum(1 for e in _hip.weight_packs.entries.values() if e[1] is not None),
                          sum(v["calls"] for k, v in t.items() if k.startswith("gemm_x3h"))))
            outs.append([p.detach().clone() for p in model.parameters()])
    finally:
        _hip.weight_packs.enabled = os.environ.get("GT_PACK_ALL", "1") != "0"
        _hip.weight_packs.clear()
    assert packs[0][:2] == (0, 0)
    assert packs[1][0] == 1 and packs[1][1] >= 30, packs           # 6 layers x (QKV, FFN1, FFN2) x (forward, data gradient)
    assert packs[0][2] == packs[1][2] > 0                          # the same products ran on the packed-B kernels
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("workload,B,library_convs", [("ex2_darcy141", 8, False), ("ex2_darcy141", 1, False),
                                                       ("ex2_darcy211_fourier", 2, False), ("ex2_darcy211_fourier", 1, False),
                                                       ("ex4_ns", 2, False), ("ex1_burgers", 2, False),
                                                       ("ex3_darcy_inv", 8, False), ("ex3_darcy_inv", 1, False)])
def test_no_library_convolution_in_the_training_step(gpu_device, workload, B, library_convs):
    """one eager training step of every BASELINE workload under a dispatch spy -- no aten::convolution /
    convolution_backward (MIOpen) may be reached for C1, C2, C3, C5 (C3's 113 / 114-pixel rows run gt_conv3x3_wgrad_nhwc in two
    x-segments since round 5; the down-scaler chain runs from 1 024 pixel rows, i.e. at batch 1 too).  Round 6: C4 (ex3),
    whose down-scaler config.yml leaves on the SiLU default, was the pinned exception until the fused conv0 + resize and the
    segment chain got their SiLU forms (GT_ACT_DROP_SILU, gt_hip.h v20) -- no workload reaches the library any more."""
    import sys
    from torch.utils._python_dispatch import TorchDispatchMode
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import galerkin_transformer as gt

    class Spy(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.seen = []

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func.__name__.split(".")[0]
            if "convolution" in name or name.startswith("miopen") or name.startswith("cudnn"):
                self.seen.append(name)
            return func(*args, **(kwargs or {}))

    torch.manual_seed(1)
    model, _ = bench.build_model(workload)
    model = model.to(gpu_device).train()
    gt.set_attention_dropout("reference")
    batch = bench.synthetic_batch(B, gpu_device, seed=3, workload=workload)
    tr = bench.Trainer(model, batch, 1, use_graph=False, workload=workload)
    tr.eager_step()
    spy = Spy()
    with spy:
        tr.eager_step()
    torch.cuda.synchronize()
    assert np.isfinite(float(tr.loss.item()))
    assert bool(spy.seen) == library_convs, (workload, spy.seen)


@pytest.mark.parametrize("workload,B", [("ex2_darcy141", 10), ("ex3_darcy_inv", 14), ("ex4_ns", 5)])
def test_masked_gradient_twins_change_nothing_but_the_launch_count(gpu_device, workload, B):
    """Round 6 (the elementwise folds): the dropout-masked copy of the data gradient between two encoder
    blocks (nn.Dropout of reference model.py:125,132 backwards) is written by the product that computes the gradient
    (gt_gemm_desc.c_masked, ops._masked_twins) instead of by one gt_dropout_apply pass per block.  Same seed => bit-identical
    parameter gradients with the fold on and off; with it the step launches fewer gt_dropout_apply (ex4's LayerNorm layers sit
    between the blocks: nothing to fold there, and nothing may change)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import galerkin_transformer as gt
    from galerkin_transformer import _hip, ops
    grads, drops = [], []
    old = ops._fold_masks[0]
    try:
        for fold in (False, True):
            ops._fold_masks[0] = fold
            torch.manual_seed(11)
            model, _ = bench.build_model(workload)
            model = model.to(gpu_device).train()
            gt.set_attention_dropout("reference")
            batch = bench.synthetic_batch(B, gpu_device, seed=7, workload=workload)
            tr = bench.Trainer(model, batch, 1, use_graph=False, workload=workload)
            _hip.set_seed(123, gpu_device)
            for p in tr.params:
                p.grad = None
            with _hip.Profile() as prof:
                tr.fwd_bwd()
            torch.cuda.synchronize()
            grads.append(tr.opt.flat_grad.clone())
            drops.append(prof.table().get("gt_dropout_apply", {}).get("calls", 0))
    finally:
        ops._fold_masks[0] = old
    assert torch.equal(grads[0], grads[1])
    L = 6 if workload != "ex4_ns" else 0
    assert drops[0] - drops[1] == (2 * L - 1 if L else 0), drops       # every block but the LAST one (its gradient comes from the up-scaler)


@pytest.mark.parametrize("workload,B", [("ex2_darcy141", 2), ("ex2_darcy211_fourier", 1)])
def test_upscaler_activations_on_the_convolution_epilogue(gpu_device, workload, B):
    """Round 6: the conv -> SiLU -> SiLU tail of Interp2dUpsample (reference layers.py:642-650; ex2: upscaler_dropout 0) runs
    on the epilogue of the implicit-GEMM convolution (GT_ACT_SILU2) and its derivative on the epilogue of the product that
    forms the features' gradient (ops.upsample_fc(in_factor=...)) instead of in gt_dropact_fwd / gt_dropact_bwd.  Same
    seed => the same loss and parameter gradients as the unfused passes to rounding, and two elementwise launches fewer."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import galerkin_transformer as gt
    from galerkin_transformer import _hip, layers
    grads, calls, losses = [], [], []
    old = layers._fuse_act2[0]
    try:
        for fuse in (False, True):
            layers._fuse_act2[0] = fuse
            torch.manual_seed(11)
            model, _ = bench.build_model(workload)
            model = model.to(gpu_device).train()
            gt.set_attention_dropout("reference")
            batch = bench.synthetic_batch(B, gpu_device, seed=7, workload=workload)
            tr = bench.Trainer(model, batch, 1, use_graph=False, workload=workload)
            _hip.set_seed(123, gpu_device)
            for p in tr.params:
                p.grad = None
            with _hip.Profile() as prof:
                tr.fwd_bwd()
            torch.cuda.synchronize()
            grads.append(tr.opt.flat_grad.clone())
            losses.append(float(tr.loss.item()))
            t = prof.table()
            calls.append(t.get("gt_dropact_fwd", {}).get("calls", 0) + t.get("gt_dropact_bwd", {}).get("calls", 0))
    finally:
        layers._fuse_act2[0] = old
    assert calls[0] - calls[1] == 2, calls
    assert abs(losses[0] - losses[1]) <= 1e-6 * abs(losses[0])
    rel = float((grads[0] - grads[1]).norm() / grads[0].norm())
    assert rel < 2e-6, rel


@pytest.mark.parametrize("workload,B", [("ex2_darcy141", 2), ("ex2_darcy211_fourier", 1)])
def test_silu_backward_of_the_spectral_layers_on_the_gradient_kernels(gpu_device, workload, B):
    """Round 6: inside SpectralRegressor (reference model.py:569-580) the SiLU backward of a SpectralConv2d rides on the store
    of the kernel that forms its result's gradient (the next layer's synthesis, the regression head's backward:
    ops.silu_gate_scope) instead of a gt_act_bwd pass of its own.  Same seed => bit-identical parameter gradients, two
    gt_act_bwd launches fewer."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import galerkin_transformer as gt
    from galerkin_transformer import _hip, ops
    grads, calls = [], []
    old = ops._gate_fold[0]
    try:
        for fold in (False, True):
            ops._gate_fold[0] = fold
            torch.manual_seed(11)
            model, _ = bench.build_model(workload)
            model = model.to(gpu_device).train()
            gt.set_attention_dropout("reference")
            batch = bench.synthetic_batch(B, gpu_device, seed=7, workload=workload)
            tr = bench.Trainer(model, batch, 1, use_graph=False, workload=workload)
            _hip.set_seed(123, gpu_device)
            for p in tr.params:
                p.grad = None
            with _hip.Profile() as prof:
                tr.fwd_bwd()
            torch.cuda.synchronize()
            grads.append(tr.opt.flat_grad.clone())
            calls.append(prof.table().get("gt_act_bwd", {}).get("calls", 0))
    finally:
        ops._gate_fold[0] = old
    assert calls[0] - calls[1] == 2, calls
    assert torch.equal(grads[0], grads[1])
    assert not ops._silu_gates and ops._gate_depth[0] == 0


def test_graph_step_equals_eager_step(gpu_device):
    """Same seed => the captured training step (fwd+bwd+clip+Adam, all dropouts on) updates the
    parameters exactly like the eager step."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import galerkin_transformer as gt
    from galerkin_transformer import _hip, ops
    outs = []
    for graph in (False, True):
        torch.manual_seed(3)
        model = gt.FourierTransformer2D(**bench.darcy_config()).to(gpu_device).train()
        gt.set_attention_dropout("reference")
        batch = bench.synthetic_batch(2, gpu_device, 5)
        tr = bench.Trainer(model, batch, 1, use_graph=graph)
        _hip.set_seed(50, gpu_device)
        _hip._salt[0] = 7
        assert tr.capture(warm=1) == graph      # one eager warm-up step either way, then (maybe) capture
        _hip.set_seed(99, gpu_device, rewind_salts=False)
        tr.step()                                # eager: salts continue where the capture trace started
        torch.cuda.synchronize()
        outs.append([p.detach().clone() for p in model.parameters()])
    worst = max(float((a - b).abs().max()) for a, b in zip(*outs))
    assert worst < 1e-6, worst


def test_bench_two_ranks_on_one_gpu_gloo(gpu_device):
    """The multi-rank path of bench.py (split graphs, flat gradient all-reduce between them, barrier +
    max-over-ranks timing, one JSON line from rank 0) with two ranks sharing cuda:0 over gloo.  RCCL itself
    needs >= 2 GPUs and is exercised by the driver's scaling run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--batch", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--strong-global-batch", "8"]      # roofline + f32 legs stay on:
    # rank 0's profiling step must not enter a collective the other rank never joins
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["config"]["hip_graph"]
    assert rec["value"] > 0 and np.isfinite(rec["config"]["final_loss"])
    assert rec["roofline"]
    if rec["config"]["precision"] != "f32":               # the fp32-MFMA leg is the comparison run of the other modes
        assert rec["f32_mfma_exact"]["value"] > 0
    # round 6: where the exchange step's time went, and the strong-scaling leg next to the weak value
    c = rec["comm"]
    assert c["ranks_seen"] == 2 and c["backend"] == "gloo" and c["bytes"] == 4 * rec["config"]["params"] and not c["in_graph"]
    assert c["timed_collectives"] == 2 and c["allreduce_us_per_step"] > 0
    (st,) = rec["strong"]
    assert st["global_batch"] == 8 and st["per_gpu_batch"] == 4 and st["value"] > 0 and st["allreduce_us_per_step"] > 0


def _run_bench_ranks(extra, port, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-f32-leg", "--strong-global-batch", "0"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_eight_ranks_on_one_gpu_gloo(gpu_device):
    """the exact launch the driver uses for the 8-GPU run (torch.distributed.run, 8 ranks, --gpus 8)
    with the ranks sharing cuda:0 over gloo at per-rank batch 2: split graphs, the flat all-reduce between them, barrier +
    max-over-ranks timing, ONE JSON line from rank 0.  (RCCL itself needs the 8-GPU node.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29551", os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--batch", "2",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-f32-leg", "--no-accuracy",
           "--strong-global-batch", "8", "--comm-in-graph"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["config"]["global_batch"] == 16 and rec["config"]["parallelism"] == "dp8"
    assert rec["scaling"] == "weak" and rec["value"] > 0 and np.isfinite(rec["config"]["final_loss"])
    # the line answers the strong-scaling target too and says where the exchange step's time went.
    # (--comm-in-graph: gloo cannot be captured, so this also exercises the fallback to the two-graph step.)
    c = rec["comm"]
    assert c["ranks_seen"] == 8 and c["bytes"] == 4 * rec["config"]["params"] and c["backend"] == "gloo"
    assert c["in_graph"] is False and c["timed_collectives"] == 2 and c["allreduce_us_per_step"] > 0
    (st,) = rec["strong"]
    assert st["global_batch"] == 8 and st["per_gpu_batch"] == 1 and st["value"] > 0 and st["hip_graph"]


def test_bench_strong_scaling_two_ranks_gloo(gpu_device):
    """--scaling strong: the global batch is fixed and split over the ranks (two ranks sharing cuda:0 over gloo); also
    the ex3 inverse-problem workload (BASELINE configs[3]: the DDP configuration) through the same path."""
    rec = _run_bench_ranks(["--backend", "gloo", "--scaling", "strong", "--global-batch", "4"], 29541)
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2
    assert rec["config"]["global_batch"] == 4 and rec["config"]["per_gpu_batch"] == 2
    rec = _run_bench_ranks(["--backend", "gloo", "--workload", "ex3_darcy_inv", "--batch", "2"], 29542)
    assert rec["config"]["global_batch"] == 4 and np.isfinite(rec["config"]["final_loss"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs two GPUs (the driver's scaling run has them)")
def test_bench_two_ranks_rccl(gpu_device):
    """The same multi-rank step over RCCL (backend nccl), one rank per GPU, whenever the box has two GPUs."""
    rec = _run_bench_ranks(["--backend", "nccl", "--batch", "2"], 29543)
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["value"] > 0
    assert np.isfinite(rec["config"]["final_loss"])


def test_flat_clip_adam_matches_torch(gpu_device):
    """optim.FlatClipAdam (gt_grad_sqnorm + gt_adam_clip_step on one flat bucket) == clip_grad_norm_ + torch.optim.Adam
    (reference libs/utils_ft.py:676-681) over several steps, odd tensor sizes, with and without clipping."""
    import galerkin_transformer as gt
    dev = gpu_device
    shapes = [(7, 13), (129,), (3, 5, 2), (1,), (64, 64)]
    for max_norm, wd in ((0.99, 0.0), (None, 0.0), (0.05, 1e-2)):
        g = torch.Generator().manual_seed(5)
        init = [torch.randn(*s, generator=g) for s in shapes]
        pa = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        pb = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        opt_a = gt.FlatClipAdam(pa, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, max_norm=max_norm)
        opt_b = torch.optim.Adam(pb, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        sched_a = torch.optim.lr_scheduler.OneCycleLR(opt_a, max_lr=3e-3, total_steps=8)
        sched_b = torch.optim.lr_scheduler.OneCycleLR(opt_b, max_lr=3e-3, total_steps=8)
        for it in range(6):
            grads = [torch.randn(*s, generator=g) * (3.0 if it % 2 else 0.1) for s in shapes]
            for p, q, gr in zip(pa, pb, grads):
                p.grad, q.grad = gr.to(dev).clone(), gr.to(dev).clone()
            if max_norm:
                torch.nn.utils.clip_grad_norm_(pb, max_norm)
            opt_a.step(); opt_b.step()
            sched_a.step(); sched_b.step()
        torch.cuda.synchronize()
        for p, q in zip(pa, pb):
            assert (p - q).abs().max().item() < 2e-6 * (1 + q.abs().max().item()), (max_norm, wd)
        if max_norm:
            ref = torch.sqrt(sum((gr ** 2).sum() for gr in grads)).item()
            assert abs(opt_a.grad_norm() - ref) < 1e-5 * ref


def test_ddp_step_equals_single_process(gpu_device, tmp_path):
    """SURVEY section 4(iv) / reference libs/utils_ft.py:656-681 under DDP: two ranks (gloo, sharing cuda:0) with B samples
    each, FlatClipAdam's in-place flat all-reduce with the 1/world average folded into gt_grad_sqnorm / gt_adam_clip_step,
    dropout off == ONE process on the concatenated 2B batch: the averaged gradient and its clip norm to 1e-5, the
    parameters after the step to 1e-4 (Adam's first step is ~lr * sign(g): an element whose gradient sits in the fp32
    noise of the two summation orders may move the other way)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "_ddp_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    f2, f1 = str(tmp_path / "ddp.pt"), str(tmp_path / "single.pt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", worker, f2, "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    out = subprocess.run([sys.executable, worker, f1, "2"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    a, b = torch.load(f2), torch.load(f1)
    assert a["world"] == 2 and b["world"] == 1
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    assert rel(a["grad"], b["grad"]) < 1e-5, rel(a["grad"], b["grad"])
    assert abs(a["norm"] - b["norm"]) < 1e-5 * b["norm"]
    assert rel(a["param"], b["param"]) < 1e-4, rel(a["param"], b["param"])


def test_flat_clip_adam_state_dict_roundtrip(gpu_device):
    """optimizer.state_dict() (what run_train pickles as `optimizer_state`) carries the moments and the step count: a
    fresh FlatClipAdam that loads it continues exactly like the original; a torch.optim.Adam state loads too; and a
    parameter re-bound after construction (model.to / .float) is refused instead of silently updating a stale bucket."""
    import copy
    import galerkin_transformer as gt
    dev = gpu_device
    g = torch.Generator().manual_seed(9)
    shapes = [(5, 7), (33,), (2, 3, 4)]
    init = [torch.randn(*s, generator=g) for s in shapes]
    grads = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(5)]

    def mk(vals):
        ps = [torch.nn.Parameter(v.clone().to(dev)) for v in vals]
        return ps, gt.FlatClipAdam(ps, lr=2e-3, max_norm=0.9)

    pa, oa = mk(init)
    for it in range(3):
        for p, gr in zip(pa, grads[it]):
            p.grad = gr.to(dev).clone()
        oa.step()
    sd = copy.deepcopy(oa.state_dict())
    assert int(sd["state"][0]["step"].item()) == 3 and sd["state"][1]["exp_avg"].abs().sum() > 0
    pb, ob = mk([p.detach().cpu() for p in pa])
    ob.load_state_dict(sd)
    for it in range(3, 5):
        for p, q, gr in zip(pa, pb, grads[it]):
            p.grad, q.grad = gr.to(dev).clone(), gr.to(dev).clone()
        oa.step(); ob.step()
    torch.cuda.synchronize()
    for p, q in zip(pa, pb):
        assert torch.equal(p, q)
    # torch.optim.Adam's state has the same layout
    pt = [torch.nn.Parameter(v.clone().to(dev)) for v in init]
    ot = torch.optim.Adam(pt, lr=2e-3)
    for p, gr in zip(pt, grads[0]):
        p.grad = gr.to(dev).clone()
    ot.step()
    pc, oc = mk([p.detach().cpu() for p in pt])
    oc.load_state_dict(ot.state_dict())
    assert int(oc.step_count.item()) == 1 and torch.allclose(oc.state[pc[0]]["exp_avg"], ot.state[pt[0]]["exp_avg"])
    # re-bound parameter
    pa[0].data = pa[0].data.clone()
    pa[0].grad = grads[0][0].to(dev)
    with pytest.raises(RuntimeError):
        oa.step()
    with pytest.raises(NotImplementedError):
        oa.add_param_group({"params": [torch.nn.Parameter(torch.zeros(3, device=dev))]})
