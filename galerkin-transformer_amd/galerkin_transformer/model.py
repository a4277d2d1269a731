"""Model assembly with the reference's module API (reference: libs/model.py): same class names,
constructor keywords / config keys, forward signatures, return dicts and state_dict keys, so that
released checkpoints load and ``examples/ex{1,2,3}*.py`` run unchanged.  The encoder layers and
the spectral / pointwise decoders run on the HIP operators of ``ops`` / ``spectral``.
"""
from __future__ import annotations

import copy
import os
import sys
from collections import defaultdict

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from . import ops
from .layers import (FeedForward, Identity, Interp2dEncoder, Interp2dUpsample, PositionalEncoding,
                     SimpleAttention, SpectralConv1d, SpectralConv2d, _act_module, _act_name, default)

current_path = os.path.dirname(os.path.abspath(__file__))
SRC_ROOT = os.path.dirname(current_path)

ADDITIONAL_ATTR = ['normalizer', 'raw_laplacian', 'return_latent', 'residual_type', 'norm_type',
                   'norm_eps', 'boundary_condition', 'upscaler_size', 'downscaler_size', 'spacial_dim',
                   'spacial_fc', 'regressor_activation', 'attn_activation', 'downscaler_activation',
                   'upscaler_activation', 'encoder_dropout', 'decoder_dropout', 'ffn_dropout']


class SimpleTransformerEncoderLayer(nn.Module):
    """x <- x +/- drop(attn(x)); [LN]; x <- x + drop(ff(x)); [LN]      (reference model.py:33-140)."""

    def __init__(self, d_model=96, pos_dim=1, n_head=2, dim_feedforward=512, attention_type='fourier',
                 pos_emb=False, layer_norm=True, attn_norm=None, norm_type='layer', norm_eps=None,
                 batch_norm=False, attn_weight=False, xavier_init: float = 1e-2,
                 diagonal_weight: float = 1e-2, symmetric_init=False, residual_type='add',
                 activation_type='relu', dropout=0.1, ffn_dropout=None, debug=False):
        super().__init__()
        dropout = default(dropout, 0.05)
        if attention_type in ['linear', 'softmax']:
            dropout = 0.1
        ffn_dropout = default(ffn_dropout, dropout)
        norm_eps = default(norm_eps, 1e-5)
        attn_norm = default(attn_norm, not layer_norm)
        if (not layer_norm) and (not attn_norm):
            attn_norm = True
        norm_type = default(norm_type, 'layer')
        self.attn = SimpleAttention(n_head=n_head, d_model=d_model, attention_type=attention_type,
                                    diagonal_weight=diagonal_weight, xavier_init=xavier_init,
                                    symmetric_init=symmetric_init, pos_dim=pos_dim, norm=attn_norm,
                                    norm_type=norm_type, eps=norm_eps, dropout=dropout)
        self.d_model = d_model
        self.n_head = n_head
        self.pos_dim = pos_dim
        self.add_layer_norm = layer_norm
        self.norm_eps = norm_eps
        if layer_norm:
            self.layer_norm1 = nn.LayerNorm(d_model, eps=norm_eps)
            self.layer_norm2 = nn.LayerNorm(d_model, eps=norm_eps)
        dim_feedforward = default(dim_feedforward, 2 * d_model)
        self.ff = FeedForward(in_dim=d_model, dim_feedforward=dim_feedforward, batch_norm=batch_norm,
                              activation=activation_type, dropout=ffn_dropout)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.residual_type = residual_type
        self.add_pos_emb = pos_emb
        if self.add_pos_emb:
            self.pos_emb = PositionalEncoding(d_model)
        self.debug = debug
        self.attn_weight = attn_weight
        self.__name__ = attention_type.capitalize() + 'TransformerEncoderLayer'

    def forward(self, x, pos=None, weight=None):
        if weight is not None:
            raise NotImplementedError("weighted attention is outside the HIP hot path")
        if self.add_pos_emb:
            x = self.pos_emb(x)
        sign = 1.0 if (self.residual_type in ['add', 'plus'] or self.residual_type is None) else -1.0
        p1 = self.dropout1.p if self.training else 0.0
        p2 = self.dropout2.p if self.training else 0.0
        x, attn_weight = self.attn.fused_forward(x, pos, residual=x, sign=sign, p_out=p1,
                                                 need_weights=bool(self.attn_weight))
        if self.add_layer_norm:
            x = ops.layer_norm(x, self.layer_norm1.weight, self.layer_norm1.bias, self.norm_eps)
        x = self.ff.fused_forward(x, residual=x, p_out=p2)
        if self.add_layer_norm:
            x = ops.layer_norm(x, self.layer_norm2.weight, self.layer_norm2.bias, self.norm_eps)
        return (x, attn_weight) if self.attn_weight else x


# the reference's README / eval notebooks still use the pre-rename class name
FourierTransformerEncoderLayer = SimpleTransformerEncoderLayer


class PointwiseRegressor(nn.Module):
    """[fc(cat[x, grid])] -> num_layers x (Linear + act + dropout) -> Linear   (model.py:472-529)."""

    def __init__(self, in_dim, n_hidden, out_dim, num_layers: int = 2, spacial_fc: bool = False,
                 spacial_dim=1, dropout=0.1, activation='silu', return_latent=False, debug=False):
        super().__init__()
        dropout = default(dropout, 0.1)
        self.spacial_fc = spacial_fc
        activ = nn.SiLU() if activation == 'silu' else nn.ReLU()
        if self.spacial_fc:
            self.fc = nn.Linear(in_dim + spacial_dim, n_hidden)
        self.ff = nn.ModuleList([nn.Sequential(nn.Linear(n_hidden, n_hidden), activ)
                                 for _ in range(max(num_layers, 1))])
        self.dropout = nn.Dropout(dropout)
        self.out = nn.Linear(n_hidden, out_dim)
        self.return_latent = return_latent
        self.debug = debug

    def forward(self, x, grid=None):
        if self.spacial_fc:
            x = ops.linear(x, self.fc.weight, self.fc.bias, extra=grid)
        p = self.dropout.p if self.training else 0.0
        for layer in self.ff:
            x = ops.linear(x, layer[0].weight, layer[0].bias, act=_act_name(layer[1]), p_drop=p)
        x = ops.linear(x, self.out.weight, self.out.bias)
        return (x, None) if self.return_latent else x


class SpectralRegressor(nn.Module):
    """[fc(cat[x, grid])] -> N x SpectralConv -> Linear . act . Linear [-> un-normalise] (model.py:532-637)."""

    def __init__(self, in_dim, n_hidden, freq_dim, out_dim, modes: int, num_spectral_layers: int = 2,
                 n_grid=None, dim_feedforward=None, spacial_fc=False, spacial_dim=2, return_freq=False,
                 return_latent=False, normalizer=None, activation='silu', last_activation=True,
                 dropout=0.1, debug=False):
        super().__init__()
        if spacial_dim == 2:
            conv = SpectralConv2d
        elif spacial_dim == 1:
            conv = SpectralConv1d
        else:
            raise NotImplementedError("3D not implemented.")
        activation = default(activation, 'silu')
        self.activation = _act_module(activation)
        dropout = default(dropout, 0.1)
        self.spacial_fc = spacial_fc
        if self.spacial_fc:
            self.fc = nn.Linear(in_dim + spacial_dim, n_hidden)
        dims = [n_hidden] + [freq_dim] * num_spectral_layers
        self.spectral_conv = nn.ModuleList(
            [conv(in_dim=dims[j], out_dim=dims[j + 1], n_grid=n_grid, modes=modes, dropout=dropout,
                  activation=activation, return_freq=return_freq, debug=debug)
             for j in range(num_spectral_layers)])
        if not last_activation:
            self.spectral_conv[-1].activation = Identity()
        self.n_grid = n_grid
        self.dim_feedforward = default(dim_feedforward, 2 * spacial_dim * freq_dim)
        self.regressor = nn.Sequential(nn.Linear(freq_dim, self.dim_feedforward), self.activation,
                                       nn.Linear(self.dim_feedforward, out_dim))
        self.normalizer = normalizer
        self.return_freq = return_freq
        self.return_latent = return_latent
        self.debug = debug

    def forward(self, x, edge=None, pos=None, grid=None, upsample_to=None, x_nhwc=False, in_factor=None):
        """``upsample_to=(Ho, Wo)``: x is the (B, C, H1, W1) feature map -- (B, H1, W1, C) with ``x_nhwc`` -- BEFORE the
        scaler's final bilinear resize; the resize is then commuted behind ``fc`` (ops.upsample_fc).  ``in_factor``: see
        Interp2dUpsample.forward_features(want_factor=True)."""
        x_latent = []
        if upsample_to is not None:
            assert self.spacial_fc
            x = ops.upsample_fc(x, upsample_to, self.fc.weight, self.fc.bias, grid, x_nhwc=x_nhwc,
                                in_factor=in_factor if in_factor is not None and in_factor.numel() else None)
        elif self.spacial_fc:
            x = ops.linear(x, self.fc.weight, self.fc.bias, extra=grid)
        # every intermediate of the stack has exactly one consumer (unless the latents are handed out): the SiLU backward of
        # a layer rides on the kernel that forms its result's gradient (ops.silu_gate_scope)
        with ops.silu_gate_scope(not self.return_latent):
            for layer in self.spectral_conv:
                x = layer(x)
                if self.return_latent:
                    x_latent.append(x.contiguous())
            x = ops.mlp_head(x, self.regressor[0].weight, self.regressor[0].bias, self.regressor[2].weight,
                             self.regressor[2].bias, act=_act_name(self.activation))
        if self.normalizer:
            x = self.normalizer.inverse_transform(x)
        if self.return_freq or self.return_latent:
            return x, dict(preds_freq=[], preds_latent=x_latent)
        return x


class _ToChannelsLast(torch.autograd.Function):
    """(B,C,H,W) -> (B,H,W,C), both dense.  The backward hands the CNN a dense NCHW gradient: the
    MIOpen / interpolate backward kernels are only exercised by PyTorch with that layout (a
    permuted-stride gradient from the token side faulted in the torch CNN backward on gfx950)."""

    @staticmethod
    def forward(ctx, x):
        return x.permute(0, 2, 3, 1).contiguous()

    @staticmethod
    def backward(ctx, g):
        return g.permute(0, 3, 1, 2).contiguous()


class _ToChannelsFirst(torch.autograd.Function):
    """(B,H,W,C) -> (B,C,H,W), both dense; dense NHWC gradient back to the token side."""

    @staticmethod
    def forward(ctx, x):
        return x.permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def backward(ctx, g):
        return g.permute(0, 2, 3, 1).contiguous()


class DownScaler(nn.Module):
    """(B, n, n, in) -> (B, n_s, n_s, out) with conv + bilinear interpolation (model.py:640-687)."""

    def __init__(self, in_dim, out_dim, dropout=0.1, padding=5, downsample_mode='conv',
                 activation_type='silu', interp_size=None, debug=False):
        super().__init__()
        if downsample_mode == 'interp':
            self.downsample = Interp2dEncoder(in_dim=in_dim, out_dim=out_dim, interp_size=interp_size,
                                              activation_type=activation_type, dropout=dropout, debug=debug)
        else:
            raise NotImplementedError("downsample mode not implemented.")
        self.in_dim = in_dim
        self.out_dim = out_dim

    def forward(self, x):
        n_grid, bsz = x.size(1), x.size(0)
        x = x.reshape(bsz, n_grid, n_grid, self.in_dim)
        # one input channel: channels-last and channels-first are the same bytes
        x = x.reshape(bsz, 1, n_grid, n_grid) if self.in_dim == 1 else _ToChannelsFirst.apply(x)
        return self.downsample(x, out_nhwc=True)


class UpScaler(nn.Module):
    """(B, n_s, n_s, in) -> (B, n, n, out): interp -> conv -> interp (model.py:690-749)."""

    def __init__(self, in_dim: int, out_dim: int, hidden_dim=None, padding=2, output_padding=0, dropout=0.1,
                 upsample_mode='conv', activation_type='silu', interp_mode='bilinear', interp_size=None,
                 debug=False):
        super().__init__()
        if upsample_mode == 'interp':
            self.upsample = Interp2dUpsample(in_dim=in_dim, out_dim=out_dim, interp_mode=interp_mode,
                                             interp_size=interp_size, dropout=dropout,
                                             activation_type=activation_type, debug=debug)
        else:
            raise NotImplementedError("upsample mode not implemented.")
        self.in_dim = in_dim
        self.out_dim = out_dim

    def forward(self, x):
        return self.upsample(x, in_nhwc=True, out_nhwc=True)


class _ConfiguredModel(nn.Module):
    """Config-dict plumbing shared by the three model classes: every config key (and every name in
    ADDITIONAL_ATTR) becomes an attribute, missing keys read as None (model.py:1071-1074)."""

    _hip_attention = ('fourier', 'integral', 'local', 'galerkin')

    def _read_config(self, kwargs):
        self.config = defaultdict(lambda: None, **kwargs)
        for key in list(self.config.keys()) + ADDITIONAL_ATTR:
            setattr(self, key, self.config[key])
        self.dim_feedforward = default(self.dim_feedforward, 2 * self.n_hidden)
        self.dropout = default(self.dropout, 0.05)
        self.dpo = nn.Dropout(self.dropout)
        if self.decoder_type == 'attention':
            self.num_encoder_layers += 1

    def _stack_encoders(self, **extra):
        if self.attention_type not in self._hip_attention:
            raise NotImplementedError(f"attention_type={self.attention_type!r}: only the galerkin / fourier "
                                      "encoders are on the HIP hot path")
        layer = SimpleTransformerEncoderLayer(
            d_model=self.n_hidden, n_head=self.n_head, attention_type=self.attention_type,
            dim_feedforward=self.dim_feedforward, layer_norm=self.layer_norm, attn_norm=self.attn_norm,
            pos_dim=self.pos_dim, xavier_init=self.xavier_init, diagonal_weight=self.diagonal_weight,
            dropout=self.encoder_dropout, ffn_dropout=self.ffn_dropout, debug=self.debug, **extra)
        # all layers start as deep copies of one initialised layer (model.py:896-897, 1153-1154)
        self.encoder_layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(self.num_encoder_layers)])

    @staticmethod
    def _initialize_layer(layer, gain=1e-2):
        for param in layer.parameters():
            if param.ndim > 1:
                xavier_uniform_(param, gain=gain)
            else:
                constant_(param, 0)

    def print_config(self):
        for a in self.config.keys():
            if not a.startswith('__'):
                print(f"{a}: \t", getattr(self, a))

    def _drop(self, x):
        return ops.dropout(x, self.dpo.p, self.training)


class SimpleTransformer(_ConfiguredModel):
    """1-D model of ex1 (Burgers): Linear -> N encoder layers -> spectral/pointwise decoder (model.py:752)."""

    def __init__(self, **kwargs):
        super().__init__()
        self._read_config(kwargs)
        self.spacial_dim = default(self.spacial_dim, self.pos_dim)
        self.spacial_fc = default(self.spacial_fc, False)
        if self.num_feat_layers and self.num_feat_layers > 0 and self.feat_extract_type in ('gcn', 'gat'):
            raise NotImplementedError("graph feature extractors are outside the HIP hot path "
                                      "(num_feat_layers is 0 in every config)")
        self.feat_extract = Identity(in_features=self.node_feats, out_features=self.n_hidden)
        self._stack_encoders(norm_type=self.norm_type, batch_norm=self.batch_norm,
                             symmetric_init=self.symmetric_init, attn_weight=self.return_attn_weight,
                             residual_type=self.residual_type, activation_type=self.attn_activation)
        if self.n_freq_targets and self.n_freq_targets > 0:
            raise NotImplementedError("frequency-target regressors are outside the HIP hot path")
        if self.decoder_type == 'pointwise':
            self.regressor = PointwiseRegressor(in_dim=self.n_hidden, n_hidden=self.n_hidden,
                                                out_dim=self.n_targets, spacial_fc=self.spacial_fc,
                                                spacial_dim=self.spacial_dim,
                                                activation=self.regressor_activation,
                                                dropout=self.decoder_dropout, debug=self.debug)
            self._initialize_layer(self.regressor)
        elif self.decoder_type == 'ifft':
            self.regressor = SpectralRegressor(in_dim=self.n_hidden, n_hidden=self.n_hidden,
                                               freq_dim=self.freq_dim, out_dim=self.n_targets,
                                               num_spectral_layers=self.num_regressor_layers,
                                               modes=self.fourier_modes, spacial_dim=self.spacial_dim,
                                               spacial_fc=self.spacial_fc, dim_feedforward=self.freq_dim,
                                               activation=self.regressor_activation,
                                               dropout=self.decoder_dropout)
        else:
            raise NotImplementedError("Decoder type not implemented")
        self.config = dict(self.config)
        self.__name__ = self.attention_type.capitalize() + 'Transformer'

    def forward(self, node, edge, pos, grid=None, weight=None):
        x_latent, attn_weights = [], []
        x = self.feat_extract(node, edge)
        if self.spacial_residual or self.return_latent:
            res = x.contiguous()
            x_latent.append(res)
        for encoder in self.encoder_layers:
            if self.return_attn_weight:
                x, w = encoder(x, pos, weight)
                attn_weights.append(w)
            else:
                x = encoder(x, pos, weight)
            if self.return_latent:
                x_latent.append(x.contiguous())
        if self.spacial_residual:
            x = res + x
        x = self._drop(x)
        x = self.regressor(x, grid=grid)
        return dict(preds=x, preds_freq=None, preds_latent=x_latent, attn_weights=attn_weights)

    def get_encoder(self):
        return self.encoder_layers


class FourierTransformer2D(_ConfiguredModel):
    """2-D model of ex2/ex3 (Darcy): CNN downscaler -> N encoder layers -> upscaler -> decoder (model.py:945)."""

    def __init__(self, **kwargs):
        super().__init__()
        self._read_config(kwargs)
        if self.num_feat_layers and self.num_feat_layers > 0 and self.feat_extract_type in ('gcn', 'gat'):
            raise NotImplementedError("graph feature extractors are outside the HIP hot path "
                                      "(num_feat_layers is 0 in every config)")
        self.feat_extract = Identity()
        if self.downscaler_size:
            self.downscaler = DownScaler(in_dim=self.node_feats, out_dim=self.n_hidden,
                                         downsample_mode=self.downsample_mode,
                                         interp_size=self.downscaler_size, dropout=self.downscaler_dropout,
                                         activation_type=self.downscaler_activation)
        else:
            self.downscaler = Identity(in_features=self.node_feats + self.spacial_dim,
                                       out_features=self.n_hidden)
        if self.upscaler_size:
            self.upscaler = UpScaler(in_dim=self.n_hidden, out_dim=self.n_hidden,
                                     upsample_mode=self.upsample_mode, interp_size=self.upscaler_size,
                                     dropout=self.upscaler_dropout, activation_type=self.upscaler_activation)
        else:
            self.upscaler = Identity()
        self._stack_encoders(batch_norm=self.batch_norm, symmetric_init=self.symmetric_init,
                             attn_weight=self.return_attn_weight, norm_eps=self.norm_eps)
        if self.decoder_type == 'pointwise':
            self.regressor = PointwiseRegressor(in_dim=self.n_hidden, n_hidden=self.n_hidden,
                                                out_dim=self.n_targets,
                                                num_layers=self.num_regressor_layers,
                                                spacial_fc=self.spacial_fc, spacial_dim=self.spacial_dim,
                                                activation=self.regressor_activation,
                                                dropout=self.decoder_dropout,
                                                return_latent=self.return_latent, debug=self.debug)
        elif self.decoder_type == 'ifft2':
            self.regressor = SpectralRegressor(in_dim=self.n_hidden, n_hidden=self.freq_dim,
                                               freq_dim=self.freq_dim, out_dim=self.n_targets,
                                               num_spectral_layers=self.num_regressor_layers,
                                               modes=self.fourier_modes, spacial_dim=self.spacial_dim,
                                               spacial_fc=self.spacial_fc,
                                               activation=self.regressor_activation,
                                               last_activation=self.last_activation,
                                               dropout=self.decoder_dropout,
                                               return_latent=self.return_latent, debug=self.debug)
        else:
            raise NotImplementedError("Decoder type not implemented")
        self.config = dict(self.config)
        self.__name__ = self.attention_type.capitalize() + 'Transformer2D'

    def forward(self, node, edge, pos, grid, weight=None, boundary_value=None):
        bsz = node.size(0)
        n_s = int(pos.size(1) ** 0.5)
        x_latent, attn_weights = [], []
        if not self.downscaler_size:
            node = torch.cat([node, pos.contiguous().view(bsz, n_s, n_s, -1)], dim=-1)
        x = self.downscaler(node)
        x = x.reshape(bsz, -1, self.n_hidden)
        x = self._drop(x)
        for encoder in self.encoder_layers:
            if self.return_attn_weight:
                x, w = encoder(x, pos, weight)
                attn_weights.append(w)
            else:
                x = encoder(x, pos, weight)
            if self.return_latent:
                x_latent.append(x.contiguous())
        x = x.view(bsz, n_s, n_s, self.n_hidden)
        up = getattr(self.upscaler, 'upsample', None)
        if (isinstance(up, Interp2dUpsample) and up.interp_mode == 'bilinear' and not self.return_latent
                and isinstance(self.regressor, SpectralRegressor) and self.regressor.spacial_fc
                and grid is not None and not (self.training and self.dpo.p > 0)):
            # nothing sits between the upscaler's last resize and the regressor's fc: run fc at the coarse
            # resolution and interpolate its freq_dim channels instead of the n_hidden ones
            mid = up.features_nhwc()
            feat, fac = up.forward_features(x, in_nhwc=True, out_nhwc=mid, want_factor=True)
            x = self.regressor(feat, grid=grid, upsample_to=tuple(up.interp_size[1]), x_nhwc=mid, in_factor=fac)
        else:
            x = self.upscaler(x)
            if self.return_latent:
                x_latent.append(x.contiguous())
            x = self._drop(x)
            if self.return_latent:
                x, xr_latent = self.regressor(x, grid=grid)
                x_latent.append(xr_latent)
            else:
                x = self.regressor(x, grid=grid)
        if self.normalizer:
            x = self.normalizer.inverse_transform(x)
        if self.boundary_condition == 'dirichlet':
            x = F.pad(x[:, 1:-1, 1:-1].contiguous(), (0, 0, 1, 1, 1, 1), "constant", 0)
            if boundary_value is not None:
                assert x.size() == boundary_value.size()
                x = x + boundary_value
        return dict(preds=x, preds_latent=x_latent, attn_weights=attn_weights)

    # the normalizer is a plain object holding tensors: move it with the module (model.py:1026-1042)
    def cuda(self, device=None):
        self = super().cuda(device)
        if self.normalizer:
            self.normalizer = self.normalizer.cuda(device)
        return self

    def cpu(self):
        self = super().cpu()
        if self.normalizer:
            self.normalizer = self.normalizer.cpu()
        return self

    def to(self, *args, **kwargs):
        self = super().to(*args, **kwargs)
        if self.normalizer:
            self.normalizer = self.normalizer.to(*args, **kwargs)
        return self


FourierTransformer = SimpleTransformer      # pre-rename alias used by the reference's eval notebooks


class FourierTransformer2DLite(_ConfiguredModel):
    """Navier-Stokes (ex4) model: Linear(cat[node, pos]) -> N encoder layers -> spectral decoder
    (model.py:1186-1283)."""

    def __init__(self, **kwargs):
        super().__init__()
        self._read_config(kwargs)
        self.spacial_dim = default(self.spacial_dim, self.pos_dim)
        self.spacial_fc = default(self.spacial_fc, False)
        self.feat_extract = Identity(in_features=self.node_feats, out_features=self.n_hidden)
        self._stack_encoders(norm_type=self.norm_type)
        self.regressor = SpectralRegressor(in_dim=self.n_hidden, n_hidden=self.n_hidden,
                                           freq_dim=self.freq_dim, out_dim=self.n_targets,
                                           num_spectral_layers=self.num_regressor_layers,
                                           modes=self.fourier_modes, spacial_dim=self.spacial_dim,
                                           spacial_fc=self.spacial_fc, dim_feedforward=self.freq_dim,
                                           activation=self.regressor_activation,
                                           dropout=self.decoder_dropout)
        self.config = dict(self.config)

    def forward(self, node, edge, pos, grid=None):
        bsz, input_dim, n_grid = node.size(0), node.size(-1), grid.size(1)
        node = torch.cat([node.reshape(bsz, -1, input_dim), pos], dim=-1)
        x = self.feat_extract(node, edge)
        for encoder in self.encoder_layers:
            x = encoder(x, pos)
        x = self._drop(x)
        x = x.view(bsz, n_grid, n_grid, -1)
        x = self.regressor(x, grid=grid)
        return dict(preds=x, preds_freq=None, preds_latent=None, attn_weights=None)
