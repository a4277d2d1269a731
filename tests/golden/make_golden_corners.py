#!/usr/bin/env python3
"""Golden fixtures for the API corners of the hot-path modules, recorded from the REAL reference (build container only;
same machinery as make_golden.py, separate generator so the existing fixtures stay byte-identical):

  corner_enc_galerkin_nopos   SimpleTransformerEncoderLayer.forward(x, pos=None): no coordinate columns, no `fc`
                              (reference layers.py:869-874, 894-897)
  corner_enc_galerkin_gelu    the same layer with coordinates and FeedForward(activation='gelu')   (layers.py:968-969)
  corner_sconv2d_freq         SpectralConv2d(return_freq=True) -> (out, out_ft)   (layers.py:1190-1197)
  corner_sconv1d_freq         SpectralConv1d(return_freq=True)                    (layers.py:1102-1106)
  corner_sconv2d_dropmask     SpectralConv2d with dropout on the FFT branch input only (layers.py:1173), the Bernoulli
                              mask replaced by a recorded one on both sides

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_corners.py
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch

from make_golden import AttnDropCtl, import_reference, perturb, record


class MaskMul(torch.nn.Module):
    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask


def main():
    L, M, FT = import_reference()
    ctl = AttnDropCtl()
    g = torch.Generator().manual_seed(20260926)

    def rn(*shape):
        return torch.randn(*shape, generator=g)

    # encoder layer without coordinates
    torch.manual_seed(5)
    kw = dict(d_model=64, pos_dim=2, n_head=4, dim_feedforward=128, attention_type="galerkin", layer_norm=False,
              attn_norm=True, norm_eps=1e-7)
    layer = M.SimpleTransformerEncoderLayer(dropout=0.0, ffn_dropout=0.0, **kw)
    perturb(layer, g)
    x = rn(2, 140, 64)
    mask = (torch.rand(2, 4, 16, 16, generator=g) >= 0.5).float() * 2.0
    meta = dict(kind="encoder_layer", B=2, n=140, nopos=True, **kw)
    record("corner_enc_galerkin_nopos", layer, dict(x=x), lambda m, x: m(x, None), meta, ctl, masks=[mask])

    # encoder layer whose FeedForward runs the erf GELU (activation_type='gelu', reference layers.py:968-969): appended
    # draws from its own generator so the fixtures above stay byte-identical
    g2 = torch.Generator().manual_seed(20260927)
    torch.manual_seed(6)
    kwg = dict(d_model=64, pos_dim=2, n_head=4, dim_feedforward=128, attention_type="galerkin", layer_norm=False,
               attn_norm=True, norm_eps=1e-7, activation_type="gelu")
    layer = M.SimpleTransformerEncoderLayer(dropout=0.0, ffn_dropout=0.0, **kwg)
    perturb(layer, g2)
    xg = torch.randn(2, 140, 64, generator=g2)
    pg = torch.rand(2, 140, 2, generator=g2)
    maskg = (torch.rand(2, 4, 18, 18, generator=g2) >= 0.5).float() * 2.0
    record("corner_enc_galerkin_gelu", layer, dict(x=xg), lambda m, x, pos: m(x, pos),
           dict(kind="encoder_layer", B=2, n=140, **kwg), ctl, masks=[maskg], const_inputs=dict(pos=pg))

    def cat_freq(out, ft):
        ft = ft.detach()
        return torch.cat([out.flatten(), ft.real.flatten(), ft.imag.flatten()])

    torch.manual_seed(7)
    conv = L.SpectralConv2d(8, 16, 5, dropout=0.0, activation="silu", return_freq=True)
    perturb(conv, g, 0.05)
    x = rn(2, 21, 21, 8)
    record("corner_sconv2d_freq", conv, dict(x=x), lambda m, x: cat_freq(*m(x)),
           dict(kind="spectral_conv2d", B=2, n=21, in_dim=8, out_dim=16, modes=5, activation="silu", flat=False,
                return_freq=True), ctl)

    torch.manual_seed(7)
    conv1 = L.SpectralConv1d(8, 8, 7, dropout=0.0, return_freq=True)
    perturb(conv1, g, 0.05)
    x = rn(2, 101, 8)
    record("corner_sconv1d_freq", conv1, dict(x=x), lambda m, x: cat_freq(*m(x)),
           dict(kind="spectral_conv1d", B=2, n=101, in_dim=8, out_dim=8, modes=7, return_freq=True), ctl)

    torch.manual_seed(7)
    conv = L.SpectralConv2d(8, 16, 6, dropout=0.25, activation="silu")
    perturb(conv, g, 0.05)
    x = rn(2, 24, 24, 8)
    dmask = (torch.rand(2, 24, 24, 8, generator=g) >= 0.25).float() / 0.75
    conv.dropout = MaskMul(dmask)
    record("corner_sconv2d_dropmask", conv, dict(x=x), lambda m, x, dropmask: m(x),
           dict(kind="spectral_conv2d", B=2, n=24, in_dim=8, out_dim=16, modes=6, activation="silu", flat=False,
                dropout=0.25), ctl, const_inputs=dict(dropmask=dmask))


if __name__ == "__main__":
    main()
