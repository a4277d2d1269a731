#!/usr/bin/env python3
"""Accuracy leg of the headline metric ("training samples/s + rel-L2 error"): train the ex2 Darcy-141 Galerkin model
for a few epochs on the deterministic synthetic Darcy set (galerkin_transformer.ft.DarcyDataset(synthetic=True): the
real piececonst_*.mat files are not on this box) with the example script's recipe -- batch 4, Adam 1e-3, OneCycleLR,
WeightedL2Loss2d(regularizer=True, gamma=0.5), clip 0.99 (examples/ex2_darcy.py:109-131, utils_ft.py:656-712) -- and
report the validation relative-L2 error.

    python tools/accuracy_leg.py --impl reference   # the reference's own model / loss / train step on CPU
                                                    # (build container only: imports /root/reference);
                                                    # result committed as profiles/accuracy_reference_cpu.json
    python tools/accuracy_leg.py --impl hip         # this repo's model on cuda:0 (bench.py runs this and prints both)

Both runs see the same data, the same initial weights (same seed; the initialisers are bit-equal, tests/
test_host_golden_cpu.py) and the same schedule; the dropout streams differ (torch CPU generator vs the stateless
device RNG), so the two errors agree statistically, not digit for digit.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))

import numpy as np
import torch

SEED = 1127802
N_TRAIN, N_VALID, BATCH, EPOCHS = 64, 16, 4, 8


_DATASETS = None


def datasets():
    """(train, valid), built once per process (6 s of CPU for the 421-grid synthesis)."""
    global _DATASETS
    if _DATASETS is None:
        _DATASETS = _build_datasets()
    return _DATASETS


def _build_datasets():
    from galerkin_transformer.ft import DarcyDataset
    kw = dict(subsample_attn=10, subsample_nodes=3, synthetic=True, n_samples_synthetic=N_TRAIN + N_VALID,
              random_state=SEED)
    train = DarcyDataset(train_data=True, train_len=N_TRAIN, **kw)
    valid = DarcyDataset(train_data=False, valid_len=N_VALID, normalizer_x=train.normalizer_x, **kw)
    return train, valid


def config(train):
    import yaml
    from galerkin_transformer.ft import DarcyDataset
    with open(os.path.join(ROOT, "galerkin-transformer_amd", "config.yml")) as f:
        cfg = yaml.full_load(f)["ex2_darcy"]
    down, up = DarcyDataset.get_scaler_sizes(141, 43)
    cfg.update(downscaler_size=down, upscaler_size=up, attn_norm=True, norm_eps=1e-7)
    return cfg


def run(impl: str, epochs: int = EPOCHS, log=None, optimizer: str = "flat", attn_dropout: str = "reference",
        dropout_seed: int = SEED):
    """dropout_seed reseeds ONLY the dropout streams (after the model is built from SEED): same data order, same
    initial weights, a different realisation of the training noise."""
    from torch.utils.data import DataLoader
    train, valid = datasets()
    cfg = config(train)
    if impl == "reference":
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from make_golden import import_reference
        L, M, FT = import_reference()
        import utils_ft as UF                                     # /root/reference/libs (import_reference put it on the path)
        device = torch.device("cpu")
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        # the reference's normaliser class on the reference side (same statistics)
        nz = FT.UnitGaussianNormalizer()
        nz.mean, nz.std = np.asarray(train.normalizer_y.mean), np.asarray(train.normalizer_y.std)
        cfg = dict(cfg, normalizer=nz.to(device))                 # ex2_darcy.py:73: normalizer_y.to(device)
        torch.manual_seed(SEED)
        model = M.FourierTransformer2D(**cfg)
        torch.manual_seed(dropout_seed)
        Loss, train_batch, validate = FT.WeightedL2Loss2d, UF.train_batch_darcy, UF.validate_epoch_darcy
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    else:
        import galerkin_transformer as gt
        from galerkin_transformer import utils_ft as UF
        device = torch.device("cuda:0")
        cfg = dict(cfg, normalizer=train.normalizer_y.to(device))
        torch.manual_seed(SEED)
        model = gt.FourierTransformer2D(**cfg).to(device)
        gt.set_attention_dropout(attn_dropout)
        gt.get_seed(dropout_seed, printout=False)
        Loss, train_batch, validate = gt.WeightedL2Loss2d, UF.train_batch_darcy, UF.validate_epoch_darcy
        opt = (gt.FlatClipAdam(model.parameters(), lr=1e-3, max_norm=0.99, model=model) if optimizer == "flat"
               else torch.optim.Adam(model.parameters(), lr=1e-3))
    g = torch.Generator().manual_seed(SEED)
    tl = DataLoader(train, batch_size=BATCH, shuffle=True, drop_last=True, generator=g)
    vl = DataLoader(valid, batch_size=BATCH, shuffle=False, drop_last=False)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, div_factor=1e4, final_div_factor=1e4, pct_start=0.3,
                                                steps_per_epoch=len(tl), epochs=epochs)
    h = 1 / 141
    loss_func, metric_func = Loss(regularizer=True, h=h, gamma=0.5), Loss(regularizer=False, h=h)
    t0, hist, steps = time.perf_counter(), [], 0
    for ep in range(epochs):
        model.train()
        losses = []
        for batch in tl:
            (loss, reg), _, _ = train_batch(model, loss_func, batch, opt, sched, device, grad_clip=0.99)
            losses.append(loss)
            steps += 1
        val = float(validate(model, metric_func, vl, device)["metric"])
        hist.append(dict(epoch=ep + 1, train_loss=float(np.mean(losses)), val_rel_l2=val))
        if log:
            print(f"[{impl}] epoch {ep + 1}/{epochs}  train loss {np.mean(losses):.4f}  val rel-L2 {val:.4f}", file=log,
                  flush=True)
    return dict(impl=impl, device=str(device), epochs=epochs, steps=steps, batch=BATCH, n_train=N_TRAIN, n_valid=N_VALID,
                seed=SEED, dropout_seed=dropout_seed, val_rel_l2=hist[-1]["val_rel_l2"], train_loss_last=hist[-1]["train_loss"], history=hist,
                seconds=round(time.perf_counter() - t0, 1),
                data="DarcyDataset(synthetic=True) 141x141 fine / 43x43 coarse; recipe of examples/ex2_darcy.py")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", default="hip", choices=["hip", "reference"])
    ap.add_argument("--epochs", type=int, default=EPOCHS)
    ap.add_argument("--out", default=None)
    ap.add_argument("--optimizer", default="flat", choices=["flat", "torch"], help="hip only (debugging)")
    ap.add_argument("--attn-dropout", default="reference", choices=["reference", "off"], help="hip only (debugging)")
    ap.add_argument("--dropout-seed", type=int, default=SEED)
    a = ap.parse_args()
    res = run(a.impl, a.epochs, log=sys.stderr, optimizer=a.optimizer, attn_dropout=a.attn_dropout,
              dropout_seed=a.dropout_seed)
    txt = json.dumps(res, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    print(json.dumps({k: v for k, v in res.items() if k != "history"}))
