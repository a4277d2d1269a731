"""Pin the CPU oracle (oracle/galerkin_oracle.py) against the golden vectors recorded from the
real reference modules (tests/golden/make_golden.py).  CPU only."""
import pytest
import torch

from _util import Golden, all_golden, rel_l2
from oracle import galerkin_oracle as O

ORACLE_TOL = 2e-6     # fp32 round-off between two orderings of the same math


def _enc_kwargs(meta):
    keys = ("n_head", "attention_type", "layer_norm", "attn_norm", "norm_eps", "residual_type", "activation_type")
    kw = {k: meta[k] for k in keys if k in meta and meta[k] is not None}
    return kw


def run_oracle(g: Golden, sd, inputs):
    m, kind = g.meta, g.meta["kind"]
    drops = g.masks if g.masks else None
    def cat_freq(out, ft):                       # the layout make_golden_corners.py records (out, out_ft) in
        ft = ft.detach()
        return torch.cat([out.flatten(), ft.real.flatten(), ft.imag.flatten()])

    if kind == "encoder_layer":
        return O.encoder_layer(sd, inputs["x"], inputs.get("pos"),
                               attn_drop=drops[0] if drops else None, **_enc_kwargs(m))
    if kind == "spectral_conv2d":
        y = O.spectral_conv2d(sd, inputs["x"], modes=m["modes"], activation=m["activation"],
                              return_freq=bool(m.get("return_freq")), spec_mask=inputs.get("dropmask"))
        return cat_freq(*y) if m.get("return_freq") else y
    if kind == "spectral_conv1d":
        y = O.spectral_conv1d(sd, inputs["x"], modes=m["modes"], return_freq=bool(m.get("return_freq")))
        return cat_freq(*y) if m.get("return_freq") else y
    if kind == "spectral_regressor":
        return O.spectral_regressor(sd, inputs["x"], inputs["grid"], modes=m["modes"],
                                    num_spectral_layers=m["num_spectral_layers"],
                                    spacial_dim=m["spacial_dim"], spacial_fc=m["spacial_fc"],
                                    activation=m["activation"], last_activation=m["last_activation"])
    if kind == "pointwise_regressor":
        return O.pointwise_regressor(sd, inputs["x"], inputs["grid"], num_layers=m["num_layers"],
                                     spacial_fc=m["spacial_fc"], activation=m["activation"])
    if kind == "fourier_transformer_2d":
        return O.fourier_transformer_2d(sd, m["config"], inputs["node"], inputs["pos"], inputs["grid"],
                                        attn_drops=drops)
    if kind == "simple_transformer":
        return O.simple_transformer_1d(sd, m["config"], inputs["node"], inputs["pos"], attn_drops=drops)
    if kind == "fourier_transformer_2d_lite":
        return O.fourier_transformer_2d_lite(sd, m["config"], inputs["node"], inputs["pos"],
                                             inputs["grid"], attn_drops=drops)
    raise KeyError(kind)


@pytest.mark.parametrize("name", all_golden())
def test_oracle_matches_reference_golden(name):
    g = Golden(name)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in g.sd.items()}
    grad_in = [k for k in g.din]
    inputs = {k: (v.clone().requires_grad_(True) if k in grad_in else v) for k, v in g.inputs.items()}
    out = run_oracle(g, sd, inputs)
    assert out.shape == g.out.shape
    assert rel_l2(out, g.out) < ORACLE_TOL
    leaves = [inputs[k] for k in grad_in] + [sd[k] for k in g.dparam]
    grads = torch.autograd.grad(out, leaves, g.cot)
    for k, gr in zip(grad_in, grads):
        assert rel_l2(gr, g.din[k]) < 5 * ORACLE_TOL, f"d{k}"
    for k, gr in zip(g.dparam, grads[len(grad_in):]):
        assert rel_l2(gr, g.dparam[k]) < 5 * ORACLE_TOL, f"dparam {k}"


def test_oracle_fp64_envelope():
    """fp32 oracle vs the same oracle in fp64: the numerical envelope the HIP path is judged in."""
    g = Golden("enc_galerkin_c2")
    o32 = run_oracle(g, g.sd, g.inputs)
    sd64 = {k: v.double() for k, v in g.sd.items()}
    in64 = {k: v.double() for k, v in g.inputs.items()}
    o64 = run_oracle(g, sd64, in64)
    assert rel_l2(o32, o64) < 1e-6
