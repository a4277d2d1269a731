"""Module-level parity (-m gpu): our nn.Modules (HIP path, through the C ABI) against the golden
vectors recorded from the reference modules -- outputs, input gradients and every parameter
gradient, at the BASELINE.json tolerance (1e-5 relative L2, fp32)."""
import pytest
import torch

from _util import Golden, TOL, all_golden, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def GT(gpu_device):
    import galerkin_transformer as gt
    from galerkin_transformer import _hip
    _hip.lib()
    return gt


def build_module(gt, g: Golden):
    m, kind = g.meta, g.meta["kind"]
    if kind == "encoder_layer":
        kw = {k: v for k, v in m.items() if k not in ("kind", "B", "n", "base", "nopos")}
        return gt.SimpleTransformerEncoderLayer(dropout=0.0, ffn_dropout=0.0, **kw)
    if kind == "spectral_conv2d":
        return gt.SpectralConv2d(m["in_dim"], m["out_dim"], m["modes"], dropout=m.get("dropout", 0.0),
                                 activation=m["activation"], return_freq=bool(m.get("return_freq")))
    if kind == "spectral_conv1d":
        return gt.SpectralConv1d(m["in_dim"], m["out_dim"], m["modes"], dropout=0.0,
                                 return_freq=bool(m.get("return_freq")))
    if kind == "spectral_regressor":
        kw = {k: v for k, v in m.items() if k != "kind"}
        return gt.SpectralRegressor(dropout=0.0, **kw)
    if kind == "pointwise_regressor":
        kw = {k: v for k, v in m.items() if k != "kind"}
        return gt.PointwiseRegressor(dropout=0.0, **kw)
    cfg = dict(m["config"])
    for k in ("downscaler_size", "upscaler_size"):
        if cfg.get(k) is not None:
            cfg[k] = tuple(tuple(s) if isinstance(s, list) else s for s in cfg[k])
    if kind == "fourier_transformer_2d":
        return gt.FourierTransformer2D(**cfg)
    if kind == "simple_transformer":
        return gt.SimpleTransformer(**cfg)
    if kind == "fourier_transformer_2d_lite":
        return gt.FourierTransformer2DLite(**cfg)
    raise KeyError(kind)


def run_module(mod, g: Golden, ins):
    kind = g.meta["kind"]
    if kind == "encoder_layer":
        return mod(ins["x"], ins.get("pos"))          # corner_enc_galerkin_nopos: no coordinates, no fc
    if kind in ("spectral_conv2d", "spectral_conv1d"):
        if g.meta.get("return_freq"):                 # (out, out_ft) in the layout make_golden_corners.py recorded
            out, ft = mod(ins["x"])
            assert ft.dtype == torch.complex64 and not ft.requires_grad
            return torch.cat([out.flatten(), ft.real.flatten(), ft.imag.flatten()])
        return mod(ins["x"])
    if kind in ("spectral_regressor", "pointwise_regressor"):
        return mod(ins["x"], grid=ins["grid"])
    if kind == "simple_transformer":
        return mod(ins["node"], None, ins["pos"])["preds"]
    return mod(ins["node"], None, ins["pos"], ins["grid"])["preds"]


@pytest.mark.parametrize("name", all_golden())
def test_module_matches_reference_golden(GT, gpu_device, name):
    g = Golden(name)
    dev = gpu_device
    torch.manual_seed(0)
    mod = build_module(GT, g)
    missing = mod.load_state_dict(g.sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    mod = mod.to(dev).train()       # train(): nn.Dropout p=0 everywhere, so only the attention mask matters
    if g.masks:
        GT.set_attention_dropout("replay")
        GT.push_attention_masks([m.to(dev) for m in g.masks])
    else:
        GT.set_attention_dropout("off")
    from galerkin_transformer import ops
    real_dropout = ops.dropout
    try:
        ins = {k: v.to(dev) for k, v in g.inputs.items()}
        for k in g.din:
            ins[k].requires_grad_(True)
        if "dropmask" in ins:       # corner_sconv2d_dropmask: the Bernoulli mask of the FFT-branch dropout, replayed
            ops.dropout = lambda x, p, training=True: x * ins["dropmask"].reshape(x.shape)
        out = run_module(mod, g, ins)
        assert out.shape == g.out.shape
        e_out = rel_l2(out, g.out)
        out.backward(g.cot.to(dev))
        torch.cuda.synchronize()
    finally:
        ops.dropout = real_dropout
        GT.set_attention_dropout("reference")
    errs = {"out": e_out}
    for k in g.din:
        errs["d" + k] = rel_l2(ins[k].grad, g.din[k])
    params = dict(mod.named_parameters())
    for k, ref in g.dparam.items():
        assert params[k].grad is not None, k
        errs["dW:" + k] = rel_l2(params[k].grad, ref)
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, f"{name}: {bad}  (worst {max(errs.values()):.2e})"


def test_reference_dropout_mode_statistics(GT, gpu_device):
    """Default mode reproduces the reference's always-on p=0.5 attention dropout: ~half of the
    attention matrix is zero, survivors are doubled, and two forwards differ."""
    g = Golden("enc_galerkin_c2")
    mod = build_module(GT, g)
    mod.load_state_dict(g.sd)
    mod = mod.to(gpu_device).eval()
    mod.attn_weight = True
    x, pos = g.inputs["x"].to(gpu_device), g.inputs["pos"].to(gpu_device)
    GT.set_attention_dropout("off")
    _, w0 = mod(x, pos)
    GT.set_attention_dropout("reference")
    y1, w1 = mod(x, pos)
    y2, w2 = mod(x, pos)
    torch.cuda.synchronize()
    kept = (w1 != 0)
    assert abs(kept.float().mean().item() - 0.5) < 0.05
    assert torch.allclose(w1[kept], 2 * w0[kept], rtol=1e-5, atol=1e-7)
    assert not torch.equal(w1, w2) and rel_l2(y1, y2) > 1e-4


def test_training_dropout_matches_mask_replay(GT, gpu_device):
    """Training mode with every dropout active (ex2 values): extract the stateless-RNG masks for the
    known (seed, salt) pairs and check forward + all gradients against a plain torch fp64 restatement
    that applies those masks explicitly."""
    import torch.nn.functional as F
    from galerkin_transformer import _hip, ops
    from oracle import galerkin_oracle as O
    dev = gpu_device
    g = Golden("enc_galerkin_c2")
    kw = {k: v for k, v in g.meta.items() if k not in ("kind", "B", "n", "base")}
    p1, pf = 0.05, 0.1
    mod = GT.SimpleTransformerEncoderLayer(dropout=p1, ffn_dropout=pf, **kw)
    mod.load_state_dict(g.sd)
    mod = mod.to(dev).train()
    x0, pos, cot = g.inputs["x"].to(dev), g.inputs["pos"].to(dev), g.cot.to(dev)
    B, n, d = x0.shape
    h, f = kw["n_head"], kw["dim_feedforward"]
    Dr = d // h + pos.shape[-1]
    DP = (Dr + 3) // 4 * 4
    _hip.set_seed(4242, dev)
    _hip._salt[0] = 1000
    GT.set_attention_dropout("reference")
    x = x0.clone().requires_grad_(True)
    y = mod(x, pos)
    y.backward(cot)
    torch.cuda.synchronize()

    def mask(shape, p, salt):
        ones = torch.ones(shape, device=dev)
        return _hip.dropout_apply(ones, _hip.dropout_desc(p, salt, dev)).double()

    m_attn = mask((B, h, DP, DP), 0.5, 1000)[..., :Dr, :Dr]
    m1, mh, m2 = mask((B, n, d), p1, 1001), mask((B, n, f), pf, 1004), mask((B, n, d), p1, 1005)
    sd = {k: v.detach().double().requires_grad_(True) for k, v in mod.state_dict().items()}
    x64 = x0.double().requires_grad_(True)
    att, _ = O.simple_attention(O._sub(sd, "attn."), x64, pos.double(), n_head=h, attention_type="galerkin",
                                norm=True, eps=kw["norm_eps"], attn_drop=m_attn)
    x1 = x64 + att * m1
    hid = torch.relu(F.linear(x1, sd["ff.lr1.weight"], sd["ff.lr1.bias"])) * mh
    ref = x1 + F.linear(hid, sd["ff.lr2.weight"], sd["ff.lr2.bias"]) * m2
    names = list(sd)
    grads = torch.autograd.grad(ref, [x64] + [sd[k] for k in names], cot.double())
    errs = {"out": rel_l2(y, ref), "dx": rel_l2(x.grad, grads[0])}
    params = dict(mod.named_parameters())
    for k, gr in zip(names, grads[1:]):
        errs["dW:" + k] = rel_l2(params[k].grad, gr)
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad


def test_ns_rollout_matches_reference_golden(GT, gpu_device):
    """ex4 training objective (reference libs/ns_lite.py:205-238): 10-step autoregressive rollout of the Lite model,
    loss + regulariser summed over the steps, ONE backward -- loss, all predictions and every parameter gradient
    against the fixture recorded from the reference (tests/golden/make_golden_host.py)."""
    import json
    import os
    import numpy as np
    from _util import GOLDEN
    from galerkin_transformer.ns_lite import rollout_loss
    z = np.load(os.path.join(GOLDEN, "host_ns_rollout.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    t = lambda k: torch.from_numpy(np.array(z[k])).to(gpu_device)
    model = GT.FourierTransformer2DLite(**meta["config"])
    model.load_state_dict({k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith("sd/")}, strict=True)
    model = model.to(gpu_device).train()
    loss_func = GT.WeightedL2Loss2d(regularizer=True, h=meta["h"], gamma=meta["gamma"])
    GT.set_attention_dropout("off")
    try:
        total, reg_total, preds = rollout_loss(model, loss_func, t("in/x"), t("in/pos"), t("in/grid"), t("in/u"),
                                               t("in/gradu"))
        total.backward()
        torch.cuda.synchronize()
    finally:
        GT.set_attention_dropout("reference")
    assert abs(float(total) - float(z["loss_total"])) < 1e-5 * abs(float(z["loss_total"]))
    assert abs(reg_total - float(z["reg_total"])) < 1e-5 * abs(float(z["reg_total"]))
    errs = {"preds": rel_l2(torch.cat(preds, -1), torch.from_numpy(np.array(z["preds"])))}
    for k, p in model.named_parameters():
        errs["dW:" + k] = rel_l2(p.grad, torch.from_numpy(np.array(z["dparam/" + k])))
    bad = {k: v for k, v in errs.items() if not v < 3 * TOL}          # ten chained forwards: 3e-5
    assert not bad, f"{bad} (worst {max(errs.values()):.2e})"
