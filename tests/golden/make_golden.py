#!/usr/bin/env python3
"""Generate golden fixtures from the REAL reference implementation.

Runs only in the build container (needs /root/reference, read-only).  It imports the
reference's ``libs/{layers,model}.py`` by path, builds the hot-path modules at small
sizes, perturbs every parameter (so the 6 deep-copied layers are not identical /
near-identity), feeds seeded inputs, and records outputs + gradients w.r.t. inputs
and every parameter into ``tests/golden/<case>.npz``.

Dropout handling: every ``nn.Dropout`` is built with p=0; the attention-matrix
``F.dropout(p_attn)`` (p=0.5, always on -- reference layers.py:700-701, 730-731) is
intercepted: identity for ``attn_drop=None`` cases, mask replay (mask stored in the
fixture) for ``*_replay`` cases.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import contextlib
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference/libs"
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name, attrs in (("torchinfo", dict(summary=lambda *a, **k: None)),
                        ("IPython", dict(get_ipython=lambda: None))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    sys.path.insert(0, REF)
    import layers as ref_layers      # noqa
    import model as ref_model        # noqa
    import ft as ref_ft              # noqa
    return ref_layers, ref_model, ref_ft


class AttnDropCtl:
    """Intercepts the 1-argument F.dropout(p_attn) calls of the reference."""

    def __init__(self):
        self.masks = None        # None -> identity ; list -> replay in call order
        self.i = 0

    @contextlib.contextmanager
    def active(self, masks=None):
        import torch.nn.functional as F
        orig = F.dropout
        self.masks, self.i = masks, 0

        def patched(inp, *args, **kwargs):
            if not args and not kwargs:          # the attention call site
                if self.masks is None:
                    return inp
                m = self.masks[self.i]
                self.i += 1
                return inp * m
            return orig(inp, *args, **kwargs)

        F.dropout = patched
        try:
            yield
        finally:
            F.dropout = orig


def perturb(module, gen, scale=0.02):
    with torch.no_grad():
        for p in module.parameters():
            p.add_(scale * torch.randn(p.shape, generator=gen))


def record(name, module, inputs, run, meta, ctl, masks=None, const_inputs=None, base=None):
    """inputs: dict name->tensor (float, get grads).  run(module, **inputs, **const) -> out.
    base: name of a sibling fixture holding the same sd/ and in/ arrays (not stored again)."""
    const_inputs = const_inputs or {}
    ins = {k: v.clone().requires_grad_(True) for k, v in inputs.items()}
    with ctl.active(masks):
        out = run(module, **ins, **const_inputs)
    gen = torch.Generator().manual_seed(99)
    cot = torch.randn(out.shape, generator=gen)
    params = dict(module.named_parameters())
    leaves = list(ins.values()) + list(params.values())
    grads = torch.autograd.grad(out, leaves, cot, allow_unused=True)
    if base is not None:
        meta = dict(meta, base=base)
    blob = {"out": out.detach().numpy(), "cot": cot.numpy(),
            "meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)}
    if base is None:
        for k, v in module.state_dict().items():
            blob["sd/" + k] = v.detach().numpy()
    for (k, v), g in zip(ins.items(), grads[:len(ins)]):
        if base is None:
            blob["in/" + k] = v.detach().numpy()
        if g is not None:
            blob["din/" + k] = g.numpy()
    for k, v in const_inputs.items():
        if torch.is_tensor(v) and base is None:
            blob["in/" + k] = v.numpy()
    for k, g in zip(params.keys(), grads[len(ins):]):
        if g is not None:
            blob["dparam/" + k] = g.numpy()
    if masks is not None:
        for i, m in enumerate(masks):
            blob[f"mask/{i}"] = m.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **blob)
    print(f"{name:34s} out{tuple(out.shape)}  |out|={float(out.norm()):.4f}  "
          f"{os.path.getsize(path)/1024:.0f} KiB")


def main():
    L, M, FT = import_reference()
    ctl = AttnDropCtl()
    g = torch.Generator().manual_seed(1127802)      # reference default seed (utils_ft.py:46)

    def rn(*shape):
        return torch.randn(*shape, generator=g)

    # ---------------- encoder layers -------------------------------------------------
    def enc_case(name, B, n, masks_shape=None, **kw):
        torch.manual_seed(1127802)
        layer = M.SimpleTransformerEncoderLayer(dropout=0.0, ffn_dropout=0.0, **kw)
        perturb(layer, g)
        p = kw.get("pos_dim", 1)
        x, pos = rn(B, n, kw["d_model"]), torch.rand(B, n, p, generator=g)
        meta = dict(kind="encoder_layer", B=B, n=n, **kw)
        run = lambda m, x, pos: m(x, pos)
        record(name, layer, dict(x=x), run, meta, ctl, const_inputs=dict(pos=pos))
        if masks_shape is not None:
            mask = (torch.rand(*masks_shape, generator=g) >= 0.5).float() * 2.0
            record(name + "_replay", layer, dict(x=x), run, meta, ctl, masks=[mask],
                   const_inputs=dict(pos=pos), base=name)

    # C2-like: Darcy galerkin, 4 heads x (32+2)
    enc_case("enc_galerkin_c2", 2, 150, masks_shape=(2, 4, 34, 34), d_model=128, pos_dim=2, n_head=4,
             dim_feedforward=256, attention_type="galerkin", layer_norm=False, attn_norm=True,
             norm_eps=1e-7, xavier_init=1e-2, diagonal_weight=1e-2)
    # C3-like: fourier, LN on Q,K
    enc_case("enc_fourier_c3", 2, 131, masks_shape=(2, 4, 131, 131), d_model=64, pos_dim=2, n_head=4,
             dim_feedforward=128, attention_type="fourier", layer_norm=False, attn_norm=True,
             norm_eps=1e-7, xavier_init=1e-3, diagonal_weight=1e-2)
    # C4-like: d_k' = 50
    enc_case("enc_galerkin_c4", 1, 144, d_model=96, pos_dim=2, n_head=2, dim_feedforward=192,
             attention_type="galerkin", layer_norm=False, attn_norm=True, norm_eps=1e-7)
    # C5-like: 1 head, conventional LayerNorm, no per-head norm
    enc_case("enc_galerkin_c5_ln", 2, 256, masks_shape=(2, 1, 50, 50), d_model=48, pos_dim=2, n_head=1,
             dim_feedforward=96, attention_type="galerkin", layer_norm=True, attn_norm=False)
    # C1-like: 1-D, pos_dim 1, 4 heads x (16+1), 'minus' residual exercises the sign
    enc_case("enc_galerkin_c1", 2, 300, d_model=64, pos_dim=1, n_head=4, dim_feedforward=128,
             attention_type="galerkin", layer_norm=False, attn_norm=True, residual_type="minus")
    # C1 as shipped: 1 head x (96+1)
    enc_case("enc_galerkin_c1_h1", 1, 200, d_model=96, pos_dim=1, n_head=1, dim_feedforward=192,
             attention_type="galerkin", layer_norm=False, attn_norm=True, residual_type="plus")

    # ---------------- spectral convs --------------------------------------------------
    def sc2_case(name, B, n, cin, cout, modes, flat=False, activation="silu"):
        torch.manual_seed(7)
        conv = L.SpectralConv2d(cin, cout, modes, dropout=0.0, activation=activation)
        perturb(conv, g, 0.05)
        x = rn(B, n * n, cin) if flat else rn(B, n, n, cin)
        meta = dict(kind="spectral_conv2d", B=B, n=n, in_dim=cin, out_dim=cout, modes=modes,
                    activation=activation, flat=flat)
        record(name, conv, dict(x=x), lambda m, x: m(x), meta, ctl)

    sc2_case("sconv2d_odd", 2, 33, 8, 16, 12)
    sc2_case("sconv2d_even_flat", 2, 28, 20, 12, 6, flat=True)
    sc2_case("sconv2d_relu", 1, 21, 8, 16, 5, activation="relu")

    def sc1_case(name, B, n, cin, cout, modes):
        torch.manual_seed(7)
        conv = L.SpectralConv1d(cin, cout, modes, dropout=0.0)
        perturb(conv, g, 0.05)
        x = rn(B, n, cin)
        meta = dict(kind="spectral_conv1d", B=B, n=n, in_dim=cin, out_dim=cout, modes=modes)
        record(name, conv, dict(x=x), lambda m, x: m(x), meta, ctl)

    sc1_case("sconv1d", 2, 256, 24, 16, 16)
    sc1_case("sconv1d_odd", 1, 101, 8, 8, 7)

    # ---------------- regressors -------------------------------------------------------
    torch.manual_seed(3)
    reg = M.SpectralRegressor(in_dim=24, n_hidden=16, freq_dim=16, out_dim=1, modes=6,
                              num_spectral_layers=2, spacial_dim=2, spacial_fc=True,
                              activation="silu", dropout=0.0)
    perturb(reg, g, 0.05)
    x, grid = rn(2, 25, 25, 24), torch.rand(2, 25, 25, 2, generator=g)
    record("spectral_regressor2d", reg, dict(x=x), lambda m, x, grid: m(x, grid=grid),
           dict(kind="spectral_regressor", in_dim=24, n_hidden=16, freq_dim=16, out_dim=1, modes=6,
                num_spectral_layers=2, spacial_dim=2, spacial_fc=True, activation="silu",
                last_activation=True), ctl, const_inputs=dict(grid=grid))

    torch.manual_seed(3)
    preg = M.PointwiseRegressor(in_dim=32, n_hidden=32, out_dim=1, num_layers=1, spacial_fc=True,
                                spacial_dim=2, dropout=0.0, activation="silu")
    perturb(preg, g, 0.05)
    x, grid = rn(2, 12, 12, 32), torch.rand(2, 12, 12, 2, generator=g)
    record("pointwise_regressor", preg, dict(x=x), lambda m, x, grid: m(x, grid=grid),
           dict(kind="pointwise_regressor", in_dim=32, n_hidden=32, out_dim=1, num_layers=1,
                spacial_fc=True, spacial_dim=2, activation="silu"), ctl, const_inputs=dict(grid=grid))

    # ---------------- whole models ------------------------------------------------------
    import yaml
    with open("/root/reference/config.yml") as f:
        cfgs = yaml.full_load(f)

    # Darcy (ex2) at reduced size: fine 57, coarse 15
    n_f, n_c = 57, 15
    down, up = FT.DarcyDataset.get_scaler_sizes(n_f, n_c)
    cfg = dict(cfgs["ex2_darcy"])
    cfg.update(n_hidden=32, n_head=2, dim_feedforward=64, num_encoder_layers=2, freq_dim=16,
               fourier_modes=6, downscaler_size=[float(v) for v in down],
               upscaler_size=[list(map(int, s)) for s in up], norm_eps=1e-7,
               dropout=0.0, downscaler_dropout=0.0, upscaler_dropout=0.0, ffn_dropout=0.0,
               encoder_dropout=0.0, decoder_dropout=0.0)
    torch.manual_seed(11)
    mcfg = dict(cfg)
    mcfg["downscaler_size"] = tuple(cfg["downscaler_size"])
    mcfg["upscaler_size"] = tuple(tuple(s) for s in cfg["upscaler_size"])
    model = M.FourierTransformer2D(**mcfg)
    perturb(model, g, 0.02)
    node, pos, grid = rn(2, n_f, n_f, 1), torch.rand(2, n_c * n_c, 2, generator=g), \
        torch.rand(2, n_f, n_f, 2, generator=g)
    masks = [(torch.rand(2, 2, 18, 18, generator=g) >= 0.5).float() * 2.0 for _ in range(2)]
    record("model_darcy_small", model, dict(node=node),
           lambda m, node, pos, grid: m(node, None, pos, grid)["preds"],
           dict(kind="fourier_transformer_2d", config=cfg, n_f=n_f, n_c=n_c), ctl, masks=masks,
           const_inputs=dict(pos=pos, grid=grid))

    # Darcy inverse (ex3) at reduced size: pointwise decoder, no upscaling
    n_f, n_c = 45, 12
    down, _ = FT.DarcyDataset.get_scaler_sizes(n_f, n_c, scale_factor=False)
    cfg = dict(cfgs["ex3_darcy_inv"])
    cfg.update(n_hidden=48, n_head=4, dim_feedforward=96, num_encoder_layers=2,
               downscaler_size=[list(map(int, s)) for s in down], upscaler_size=[[n_c, n_c], [n_c, n_c]],
               norm_eps=1e-7, dropout=0.0, downscaler_dropout=0.0, upscaler_dropout=0.0,
               ffn_dropout=0.0, encoder_dropout=0.0, decoder_dropout=0.0)
    torch.manual_seed(12)
    mcfg = dict(cfg)
    mcfg["downscaler_size"] = tuple(tuple(s) for s in cfg["downscaler_size"])
    mcfg["upscaler_size"] = tuple(tuple(s) for s in cfg["upscaler_size"])
    model = M.FourierTransformer2D(**mcfg)
    perturb(model, g, 0.02)
    node, pos, grid = rn(2, n_f, n_f, 1), torch.rand(2, n_c * n_c, 2, generator=g), \
        torch.rand(2, n_c, n_c, 2, generator=g)
    record("model_darcy_inv_small", model, dict(node=node),
           lambda m, node, pos, grid: m(node, None, pos, grid)["preds"],
           dict(kind="fourier_transformer_2d", config=cfg, n_f=n_f, n_c=n_c), ctl,
           const_inputs=dict(pos=pos, grid=grid))

    # Burgers (ex1) at reduced size, galerkin
    cfg = dict(cfgs["ex1_burgers"])
    cfg.update(attention_type="galerkin", n_hidden=32, n_head=2, dim_feedforward=64,
               num_encoder_layers=2, freq_dim=16, fourier_modes=8)
    torch.manual_seed(13)
    model = M.SimpleTransformer(**cfg)
    perturb(model, g, 0.02)
    node, pos = rn(2, 256, 1), torch.linspace(0, 1, 256)[None, :, None].repeat(2, 1, 1)
    record("model_burgers_small", model, dict(node=node),
           lambda m, node, pos: m(node, None, pos)["preds"],
           dict(kind="simple_transformer", config=cfg), ctl, const_inputs=dict(pos=pos))

    # Navier-Stokes lite (ex4) at reduced size
    cfg = dict(node_feats=10 + 2, pos_dim=2, n_targets=1, n_hidden=24, num_encoder_layers=2,
               n_head=1, dim_feedforward=48, attention_type="galerkin", layer_norm=True,
               attn_norm=False, xavier_init=0.01, diagonal_weight=0.01, encoder_dropout=0.0,
               ffn_dropout=0.0, dropout=0.0, decoder_dropout=0.0, decoder_type="ifft2",
               freq_dim=12, num_regressor_layers=2, fourier_modes=5, spacial_dim=2,
               spacial_fc=False, regressor_activation="silu", debug=False)
    torch.manual_seed(14)
    model = M.FourierTransformer2DLite(**cfg)
    perturb(model, g, 0.02)
    ng = 20
    node, pos, grid = rn(2, ng, ng, 10), torch.rand(2, ng * ng, 2, generator=g), \
        torch.rand(2, ng, ng, 2, generator=g)
    record("model_ns_lite_small", model, dict(node=node),
           lambda m, node, pos, grid: m(node, None, pos, grid)["preds"],
           dict(kind="fourier_transformer_2d_lite", config=cfg), ctl,
           const_inputs=dict(pos=pos, grid=grid))

    # key inventory for the full-size Darcy model (boundary check: state_dict keys + shapes)
    cfg = dict(cfgs["ex2_darcy"])
    down, up = FT.DarcyDataset.get_scaler_sizes(141, 43)
    cfg.update(downscaler_size=down, upscaler_size=up, norm_eps=1e-7)
    torch.manual_seed(1127802)
    model = M.FourierTransformer2D(**cfg)
    inv = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "darcy141_state_dict_keys.json"), "w") as f:
        json.dump(dict(n_params=sum(p.numel() for p in model.parameters()), keys=inv,
                       downscaler_size=[float(v) for v in down],
                       upscaler_size=[list(map(int, s)) for s in up]), f, indent=0)
    print("darcy141 params:", sum(p.numel() for p in model.parameters()), "keys:", len(inv))


if __name__ == "__main__":
    main()
