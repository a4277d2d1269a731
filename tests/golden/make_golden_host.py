#!/usr/bin/env python3
"""Golden fixtures for the HOST mirror (losses, normaliser, initialisers, the ex4 rollout), recorded from the REAL
reference.  Companion of make_golden.py (whose fixtures stay byte-reproducible on their own); same rules: runs only
in the build container (imports /root/reference by path), writes tests/golden/host_*.npz / host_init_hashes.json.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_host.py
"""
import hashlib
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import AttnDropCtl, import_reference, perturb  # noqa: E402

SEED = 1127802


def tensor_hash(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()[:16]


def loss_cases(FT, blob, index):
    g = torch.Generator().manual_seed(SEED)
    rn = lambda *s: torch.randn(*s, generator=g)
    # ---- 1-D (ft.py:848-980)
    N, L, E = 3, 64, 8
    preds, targets = rn(N, L), rn(N, L)
    tprime, pprime = rn(N, L), rn(N, L)
    Kc = torch.rand(N, L, generator=g) + 0.5
    lat = [rn(N, L, E) * 0.3, rn(N, L, E) * 0.3]
    blob.update({"l1/preds": preds, "l1/targets": targets, "l1/tprime": tprime, "l1/pprime": pprime, "l1/K": Kc,
                 "l1/lat0": lat[0], "l1/lat1": lat[1]})
    cases1 = [
        dict(kw=dict(regularizer=False, h=1 / 64), use=()),
        dict(kw=dict(regularizer=True, h=1 / 64, gamma=0.5), use=("tprime",)),
        dict(kw=dict(regularizer=True, h=1 / 512, gamma=0.1, metric_reduction="L2"), use=("tprime",)),
        dict(kw=dict(regularizer=True, h=1 / 64, metric_reduction="Linf", return_norm=False), use=("tprime",)),
        dict(kw=dict(regularizer=True, h=1 / 64, alpha=0.3), use=("tprime", "pprime", "K")),
        dict(kw=dict(regularizer=True, h=1 / 64, orthogonal_reg=1, orthogonal_mode="global", delta=1e-2),
             use=("tprime", "lat")),
        dict(kw=dict(regularizer=False, h=1 / 64, orthogonal_reg=1, orthogonal_mode="local", delta=1e-2),
             use=("tprime", "lat")),
    ]
    for i, c in enumerate(cases1):
        p = preds.clone().requires_grad_(True)
        f = FT.WeightedL2Loss(**c["kw"])
        kwargs = {}
        if "tprime" in c["use"]:
            kwargs["targets_prime"] = tprime
        if "pprime" in c["use"]:
            kwargs["preds_prime"] = pprime
        if "K" in c["use"]:
            kwargs["K"] = Kc
        if "lat" in c["use"]:
            kwargs["preds_latent"] = lat
        loss, reg, ortho, metric = f(p, targets, **kwargs)
        total = loss + reg + ortho
        (gp,) = torch.autograd.grad(total.sum(), p)
        blob[f"l1/{i}/out"] = torch.stack([loss.detach().reshape(()), reg.detach().reshape(()),
                                           ortho.detach().reshape(()), torch.tensor(float(metric))])
        blob[f"l1/{i}/dpreds"] = gp
        index["loss1d"].append(dict(kw=c["kw"], use=list(c["use"])))
    # ---- 2-D (ft.py:983-1105)
    N, n = 2, 17
    preds, targets = rn(N, n, n), rn(N, n, n)
    tprime, pprime = rn(N, n, n, 2), rn(N, n, n, 2)
    Kc = torch.rand(N, n, n, 1, generator=g) + 0.5
    blob.update({"l2/preds": preds, "l2/targets": targets, "l2/tprime": tprime, "l2/pprime": pprime, "l2/K": Kc})
    cases2 = [
        dict(kw=dict(regularizer=False, h=1 / 141), use=()),
        dict(kw=dict(regularizer=True, h=1 / 141, gamma=0.5), use=("tprime",)),
        dict(kw=dict(regularizer=True, h=1 / 64, gamma=0.1, metric_reduction="L2"), use=("tprime",)),
        dict(kw=dict(regularizer=True, h=1 / 141, metric_reduction="Linf", return_norm=False), use=("tprime", "K")),
        dict(kw=dict(regularizer=True, h=1 / 141, alpha=0.2), use=("tprime", "pprime", "K")),
    ]
    for i, c in enumerate(cases2):
        p = preds.clone().requires_grad_(True)
        f = FT.WeightedL2Loss2d(**c["kw"])
        kwargs = {}
        if "tprime" in c["use"]:
            kwargs["targets_prime"] = tprime
        if "pprime" in c["use"]:
            kwargs["preds_prime"] = pprime
        if "K" in c["use"]:
            kwargs["K"] = Kc
        loss, reg, metric, norms = f(p, targets, **kwargs)
        (gp,) = torch.autograd.grad((loss + reg).sum(), p)
        blob[f"l2/{i}/out"] = torch.stack([loss.detach().reshape(()), reg.detach().reshape(()),
                                           torch.tensor(float(metric))])
        blob[f"l2/{i}/L2"] = norms["L2"].detach()
        blob[f"l2/{i}/dpreds"] = gp
        index["loss2d"].append(dict(kw=c["kw"], use=list(c["use"])))


def normalizer_case(FT, blob):
    g = torch.Generator().manual_seed(SEED + 1)
    x = (torch.randn(12, 9, 9, generator=g) * 3.0 + 1.5).numpy()
    nz = FT.UnitGaussianNormalizer()
    y = nz.fit_transform(x)
    blob["nz/x"] = torch.from_numpy(x)
    blob["nz/y"] = torch.from_numpy(np.asarray(y))
    blob["nz/mean"] = torch.from_numpy(np.asarray(nz.mean))
    blob["nz/std"] = torch.from_numpy(np.asarray(nz.std))
    xt = torch.randn(4, 9, 9, generator=g)
    blob["nz/xt"] = xt
    blob["nz/inv_np"] = torch.from_numpy(np.asarray(nz.inverse_transform(xt.numpy())))


def init_hashes(M, FT):
    """Same-seed construction of the example models in the reference: per-parameter sha256 prefixes."""
    import yaml
    with open("/root/reference/config.yml") as f:
        cfgs = yaml.full_load(f)
    out = {}
    # ex1 as shipped, and the BASELINE override (d=64, 4 heads)
    for name, upd in (("ex1_burgers", {}), ("ex1_burgers_d64h4", dict(n_hidden=64, n_head=4, dim_feedforward=128,
                                                                      attention_type="galerkin"))):
        cfg = dict(cfgs["ex1_burgers"])
        cfg.update(upd)
        torch.manual_seed(SEED)
        m = M.SimpleTransformer(**cfg)
        out[name] = dict(config=cfg, seed=SEED, hashes={k: tensor_hash(v) for k, v in m.state_dict().items()})
    down, up = FT.DarcyDataset.get_scaler_sizes(141, 43)
    cfg = dict(cfgs["ex2_darcy"])
    cfg.update(downscaler_size=[float(v) for v in down], upscaler_size=[list(map(int, s)) for s in up], norm_eps=1e-7)
    torch.manual_seed(SEED)
    mc = dict(cfg, downscaler_size=tuple(cfg["downscaler_size"]),
              upscaler_size=tuple(tuple(s) for s in cfg["upscaler_size"]))
    m = M.FourierTransformer2D(**mc)
    out["ex2_darcy"] = dict(config=cfg, seed=SEED, hashes={k: tensor_hash(v) for k, v in m.state_dict().items()})
    down, _ = FT.DarcyDataset.get_scaler_sizes(141, 36, scale_factor=False)
    cfg = dict(cfgs["ex3_darcy_inv"])
    cfg.update(downscaler_size=[list(map(int, s)) for s in down], upscaler_size=[[36, 36], [36, 36]], norm_eps=1e-7)
    torch.manual_seed(SEED)
    mc = dict(cfg, downscaler_size=tuple(tuple(s) for s in cfg["downscaler_size"]),
              upscaler_size=tuple(tuple(s) for s in cfg["upscaler_size"]))
    m = M.FourierTransformer2D(**mc)
    out["ex3_darcy_inv"] = dict(config=cfg, seed=SEED, hashes={k: tensor_hash(v) for k, v in m.state_dict().items()})
    cfg = dict(node_feats=12, pos_dim=2, n_targets=1, n_hidden=48, num_feat_layers=0, num_encoder_layers=4, n_head=1,
               dim_feedforward=96, attention_type="galerkin", feat_extract_type=None, xavier_init=0.01,
               diagonal_weight=0.01, layer_norm=True, attn_norm=False, return_attn_weight=False, return_latent=False,
               decoder_type="ifft", freq_dim=20, num_regressor_layers=2, fourier_modes=12, spacial_dim=2,
               spacial_fc=False, dropout=0.0, encoder_dropout=0.0, decoder_dropout=0.0, ffn_dropout=0.05, debug=False)
    torch.manual_seed(SEED)
    m = M.FourierTransformer2DLite(**cfg)
    out["ex4_ns_lite"] = dict(config=cfg, seed=SEED, n_params=sum(p.numel() for p in m.parameters()),
                              hashes={k: tensor_hash(v) for k, v in m.state_dict().items()})
    return out


def rollout_case(M, FT, ctl):
    """The ex4 training objective (ns_lite.py:205-238): 10-step autoregressive rollout, loss + regulariser summed,
    one backward; small Lite model, attention dropout neutralised."""
    g = torch.Generator().manual_seed(SEED + 2)
    cfg = dict(node_feats=10 + 2, pos_dim=2, n_targets=1, n_hidden=24, num_encoder_layers=2, n_head=1,
               dim_feedforward=48, attention_type="galerkin", layer_norm=True, attn_norm=False, xavier_init=0.01,
               diagonal_weight=0.01, encoder_dropout=0.0, ffn_dropout=0.0, dropout=0.0, decoder_dropout=0.0,
               decoder_type="ifft2", freq_dim=12, num_regressor_layers=2, fourier_modes=5, spacial_dim=2,
               spacial_fc=False, regressor_activation="silu", debug=False)
    torch.manual_seed(15)
    model = M.FourierTransformer2DLite(**cfg)
    perturb(model, g, 0.02)
    B, ng, T = 2, 16, 10
    x = torch.randn(B, ng, ng, T, generator=g)
    u = torch.randn(B, ng, ng, T, generator=g)
    gradu = torch.randn(B, ng, ng, 2, T, generator=g)
    pos = torch.rand(B, ng * ng, 2, generator=g)
    grid = torch.rand(B, ng, ng, 2, generator=g)
    loss_func = FT.WeightedL2Loss2d(regularizer=True, h=1 / ng, gamma=0.1)
    with ctl.active(None):
        xx, total, regs, preds = x, 0, [], []
        for t in range(T):
            u_pred = model(xx, None, pos=pos, grid=grid)["preds"]
            loss, reg, _, _ = loss_func(u_pred[..., 0], u[..., t:t + 1][..., 0],
                                        targets_prime=gradu[..., t:t + 1][..., 0])
            total = total + loss + reg
            regs.append(reg.item())
            xx = torch.cat((xx[..., 1:], u_pred), dim=-1)
            preds.append(u_pred)
    params = dict(model.named_parameters())
    grads = torch.autograd.grad(total, list(params.values()))
    blob = {"meta": np.frombuffer(json.dumps(dict(kind="ns_rollout", config=cfg, B=B, ng=ng, T=T, h=1 / ng,
                                                  gamma=0.1)).encode(), dtype=np.uint8),
            "loss_total": total.detach().numpy(), "reg_total": np.float64(sum(regs)),
            "preds": torch.cat(preds, -1).detach().numpy(),
            "in/x": x.numpy(), "in/u": u.numpy(), "in/gradu": gradu.numpy(), "in/pos": pos.numpy(),
            "in/grid": grid.numpy()}
    for k, v in model.state_dict().items():
        blob["sd/" + k] = v.detach().numpy()
    for k, gv in zip(params.keys(), grads):
        blob["dparam/" + k] = gv.numpy()
    path = os.path.join(HERE, "host_ns_rollout.npz")
    np.savez_compressed(path, **blob)
    print(f"host_ns_rollout: loss_total={float(total):.6f}  {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    L, M, FT = import_reference()
    blob, index = {}, dict(loss1d=[], loss2d=[])
    loss_cases(FT, blob, index)
    normalizer_case(FT, blob)
    arrays = {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in blob.items()}
    arrays["meta"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "host_losses.npz")
    np.savez_compressed(path, **arrays)
    print(f"host_losses: {len(index['loss1d'])} 1-D + {len(index['loss2d'])} 2-D cases, normaliser  "
          f"{os.path.getsize(path) / 1024:.0f} KiB")
    with open(os.path.join(HERE, "host_init_hashes.json"), "w") as f:
        json.dump(init_hashes(M, FT), f, indent=0, sort_keys=True)
    print("host_init_hashes.json written")
    rollout_case(M, FT, AttnDropCtl())


if __name__ == "__main__":
    main()
