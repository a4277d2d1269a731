"""Layer modules with the reference's names, constructor arguments, forward signatures and
state_dict keys (reference: libs/layers.py), whose hot-path arithmetic runs in libgt_hip.so.

Hot path (HIP):  SimpleAttention (:764), FeedForward (:954), SpectralConv1d (:1040),
SpectralConv2d (:1109).  Not on the hot path (plain PyTorch-ROCm / MIOpen, as SURVEY.md section 8
scopes them): the conv + bilinear-interpolation scalers (:88-150, :431-512, :624-670).
"""
from __future__ import annotations

import copy
import math
import os

import numpy as np

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_normal_, xavier_uniform_
from torch.nn.parameter import Parameter

from . import ops, spectral
from .ops import get_attention_dropout, push_attention_masks, set_attention_dropout  # noqa: F401


def default(value, d):
    return d if value is None else value


_LINEAR_FAMILY = ("linear", "galerkin", "global")
_HIP_ATTENTION = ("galerkin", "fourier", "integral", "local")


def _act_module(name, fallback="silu"):
    return nn.SiLU() if default(name, fallback) == "silu" else nn.ReLU()


def _act_name(module) -> str:
    if isinstance(module, nn.SiLU):
        return "silu"
    if isinstance(module, nn.ReLU):
        return "relu"
    if isinstance(module, nn.GELU) and getattr(module, "approximate", "none") == "none":
        return "gelu"
    if isinstance(module, (nn.Identity, Identity)):
        return "none"
    raise NotImplementedError(f"activation {type(module).__name__} has no HIP path")


# --------------------------------------------------------------------------------------- small helpers
class Identity(nn.Module):
    """Pass-through, or a Linear when both feature sizes are given (layers.py:21-41)."""

    def __init__(self, in_features=None, out_features=None, *args, **kwargs):
        super().__init__()
        if in_features is not None and out_features is not None:
            self.id = nn.Linear(in_features, out_features)
        else:
            self.id = nn.Identity()

    def forward(self, x, edge=None, grid=None):
        if isinstance(self.id, nn.Linear):
            return ops.linear(x, self.id.weight, self.id.bias)
        return x


class PositionalEncoding(nn.Module):
    """Sinusoidal table added to (B, n, d) inputs (layers.py:60-85); unused by the shipped configs."""

    def __init__(self, d_model, dropout=0.1, max_len=2 ** 13):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        position = torch.arange(max_len, dtype=torch.float).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(2 ** 13) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(position * freq)
        pe[:, 1::2] = torch.cos(position * freq)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        return ops.dropout(x + self.pe[:, :x.size(1), :], self.dropout.p, self.training)


# --------------------------------------------------------------------------------------- CNN scalers (torch/MIOpen)
class Conv2dResBlock(nn.Module):
    """conv (no bias) -> dropout [-> act -> conv -> dropout] -> (+ shortcut) -> act (layers.py:88-150)."""

    def __init__(self, in_dim, out_dim, kernel_size=3, padding=1, dilation=1, dropout=0.1, stride=1,
                 bias=False, residual=False, basic_block=False, activation_type="silu"):
        super().__init__()
        self.activation = _act_module(activation_type)
        self.add_res = residual
        self.conv = nn.Sequential(
            nn.Conv2d(in_dim, out_dim, kernel_size=kernel_size, padding=padding, dilation=dilation,
                      stride=stride, bias=bias),
            nn.Dropout(dropout))
        self.basic_block = basic_block
        if basic_block:
            self.conv1 = nn.Sequential(
                self.activation,
                nn.Conv2d(out_dim, out_dim, kernel_size=kernel_size, padding=padding, bias=bias),
                nn.Dropout(dropout))
        self.apply_shortcut = in_dim != out_dim
        if residual:
            self.res = _Shortcut2d(in_dim, out_dim) if self.apply_shortcut else Identity()

    def forward(self, x):
        # nn.Dropout members are kept for the module tree / state_dict; the masks come from the
        # library's stateless RNG (ops.dropout) so the whole step shares one graph-safe seed.
        h = self.res(x) if self.add_res else None
        act = _act_name(self.activation)
        if self.basic_block:      # activation(dropout(conv(x))) is one elementwise pass (ops.drop_act)
            x = self.conv1[1](ops.drop_act(self.conv[0](x), self.conv[1].p, act, self.training))
            if not self.add_res:
                return ops.drop_act(x, self.conv1[2].p, act, self.training)
            x = ops.dropout(x, self.conv1[2].p, self.training)
        elif not self.add_res:
            return ops.drop_act(self.conv[0](x), self.conv[1].p, act, self.training)
        else:
            x = ops.dropout(self.conv[0](x), self.conv[1].p, self.training)
        return self.activation(x + h)

    def plain(self) -> bool:
        """conv -> dropout -> activation and nothing else (lets a caller fuse its own dropout/activation on)."""
        return not self.basic_block and not self.add_res


class _Shortcut2d(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.shortcut = nn.Linear(in_features, out_features)

    def forward(self, x, edge=None, grid=None):
        return self.shortcut(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


Shortcut2d = _Shortcut2d


def _resize(x, size, act_module=None, in_nhwc=False, out_nhwc=False):
    """activation(F.interpolate(x, ..., mode='bilinear', align_corners=True)) on the HIP resize kernel;
    a ReLU is fused into the kernel, any other activation is applied on its output."""
    name = "none" if act_module is None else _act_name(act_module)
    if isinstance(size, (tuple, list)) and isinstance(size[0], float):
        raise NotImplementedError("per-axis scale factors")
    y = ops.bilinear_resize(x, size if isinstance(size, float) else tuple(size), in_nhwc, out_nhwc,
                            act="relu" if name == "relu" else None)
    return y if name in ("relu", "none") else act_module(y)


class Interp2dEncoder(nn.Module):
    """conv0 -> resize -> act -> conv1 -> conv2 -> conv3 -> cat -> resize -> act (layers.py:431-512)."""

    def __init__(self, in_dim, out_dim, kernel_size=3, stride=1, padding=1, dilation=1, interp_size=None,
                 residual=False, activation_type="silu", dropout=0.1, debug=False):
        super().__init__()
        c0 = out_dim // 3
        c1 = out_dim // 3
        c2 = int(out_dim - c0 - c1)
        pad1 = max(padding // 2, 1)
        pad2 = max(padding // 4, 1)
        activation_type = default(activation_type, "silu")
        self.interp_size = interp_size
        self.is_scale_factor = isinstance(interp_size[0], float) and isinstance(interp_size[1], float)
        common = dict(kernel_size=kernel_size, residual=residual, dropout=dropout,
                      activation_type=activation_type)
        self.conv0 = Conv2dResBlock(in_dim, out_dim, padding=padding, **common)
        self.conv1 = Conv2dResBlock(out_dim, c0, padding=pad1, stride=stride, **common)
        self.conv2 = Conv2dResBlock(c0, c1, dilation=dilation, padding=pad2, **common)
        self.conv3 = Conv2dResBlock(c1, c2, **common)
        self.activation = _act_module(activation_type)
        self.add_res = residual
        self.debug = debug

    def _fused_act(self, blocks):
        """'relu' / 'silu' when the encoder's own activation and every given block's are that one type, else None
        (Interp2dEncoder builds them all from one activation_type, layers.py:446-482)."""
        for name, cls in (("relu", nn.ReLU), ("silu", nn.SiLU)):
            if isinstance(self.activation, cls) and all(isinstance(c.activation, cls) for c in blocks):
                return name
        return None

    def _conv0_fusable(self, x) -> bool:
        """conv0 is the plain 3x3 / padding-1 / bias-free block on a <= 4-channel input that needs no
        gradient, with ReLU (or SiLU, round 6) on both sides of the resize: the case gt_conv3x3_resize_* implements."""
        c = self.conv0
        cv = c.conv[0]
        size = self.interp_size[0]
        size_ok = isinstance(size, float) or (isinstance(size, (tuple, list)) and not isinstance(size[0], float))
        return (size_ok and self._fused_act((c,)) is not None
                and not c.add_res and not c.basic_block and cv.kernel_size == (3, 3) and cv.padding == (1, 1)
                and cv.stride == (1, 1) and cv.dilation == (1, 1) and cv.groups == 1 and cv.bias is None
                and cv.in_channels <= 4 and x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad))

    def _chain_ok(self, x, out_nhwc) -> bool:
        """conv1 / conv2 / conv3 as ops.scaler_conv_chain: channels-last output wanted, plain ReLU (or SiLU) blocks without
        residual, equal dropout, widths [c, c, <= padded c] that the segment layout of the last resize describes."""
        cs = (self.conv1, self.conv2, self.conv3)
        act = self._fused_act(cs)
        if not (out_nhwc and x.is_cuda and not self.add_res and act is not None
                and all(c.plain() for c in cs)
                and len({c.conv[1].p for c in cs}) == 1):
            return False
        s0 = self.interp_size[0]
        if isinstance(s0, float):
            h1, w1 = int(math.floor(x.shape[2] * s0)), int(math.floor(x.shape[3] * s0))
        elif isinstance(s0, (tuple, list)) and not isinstance(s0[0], float):
            h1, w1 = int(s0[0]), int(s0[1])
        else:
            return False
        if x.shape[0] * h1 * w1 < 1024 or w1 < 3 or h1 < 3:       # (16 384 until round 5: C2 at B <= 2 fell back to the library)
            return False
        convs = [c.conv[0] for c in cs]
        widths = [c.out_channels for c in convs]
        cp = (max(widths) + 15) // 16 * 16
        size = self.interp_size[1]
        if isinstance(size, float):
            h2, w2 = int(math.floor(h1 * size)), int(math.floor(w1 * size))
        elif isinstance(size, (tuple, list)) and not isinstance(size[0], float):
            h2, w2 = int(size[0]), int(size[1])
        else:
            return False
        # what the segment resize kernels take (gt_resize.hip: check_seg, taps_fit): an even segment width (out_dim = 112 /
        # 160 give 37 / 53: those run the conv1 / conv2 / conv3 + cat path below), and in the backward at most RS_MAXT = 6
        # output rows / columns per input cell, i.e. an up-sampling factor below ~2
        def taps_fit(ni, no):
            return no <= 6 if ni <= 1 else 2.0 * (no - 1) / (ni - 1) + 2.0 <= 6.0
        return (widths[0] == widths[1] and widths[0] % 2 == 0 and 0 < widths[2] <= cp and sum(widths) % 4 == 0
                and h2 > 0 and w2 > 0 and taps_fit(h1, h2) and taps_fit(w1, w2) and ops.scaler_chain_ok(convs, act))

    def forward(self, x, out_nhwc=False):
        """x (B, C, H, W).  ``out_nhwc`` returns (B, H', W', C') with the layout change fused into the
        last resize (what DownScaler feeds the encoder)."""
        chain = self._chain_ok(x, out_nhwc)
        if self._conv0_fusable(x):
            # conv0 -> dropout -> act -> resize -> act in one pass; the out_dim-channel fine map never exists
            x = ops.conv3x3_resize(x, self.conv0.conv[0].weight, self.interp_size[0], self.conv0.conv[1].p,
                                   self.training, out_nhwc=chain, act=self._fused_act((self.conv0,)))
        else:
            x = _resize(self.conv0(x), self.interp_size[0], self.activation, out_nhwc=chain)
        if chain:
            # channels-last from here on: the three narrow convolutions (+ dropout + ReLU) write the column segments of
            # ONE padded buffer (implicit GEMMs, ops.scaler_conv_chain), the last resize reads the real channels out of it
            cs = (self.conv1, self.conv2, self.conv3)
            seg = cs[0].conv[0].out_channels
            n_out = sum(c.conv[0].out_channels for c in cs)
            act = self._fused_act(cs)
            buf = ops.scaler_conv_chain(x, *(c.conv[0].weight for c in cs), p_drop=cs[0].conv[1].p, training=self.training,
                                        grad_masked=True, act=act)
            if act == "silu":       # + the factor (dropout scale x silu') the last resize hands the gradient back through
                buf, fac = buf
                return ops.bilinear_resize_seg(buf, n_out, self.interp_size[1], seg, buf.shape[-1] // 3, act="silu",
                                               in_factor=fac)
            return ops.bilinear_resize_seg(buf, n_out, self.interp_size[1], seg, buf.shape[-1] // 3, act="relu",
                                           relu_input=True)
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        out = torch.cat([x1, x2, x3], dim=1)
        if self.add_res:
            out = out + x
        return _resize(out, self.interp_size[1], self.activation, out_nhwc=out_nhwc)


_fuse_act2 = [os.environ.get("GT_CONV_ACT2", "1") != "0"]      # A/B switch: the up-scaler's two SiLUs on the convolution's epilogue


class Interp2dUpsample(nn.Module):
    """resize -> conv block -> dropout -> act -> resize (layers.py:624-670)."""

    def __init__(self, in_dim, out_dim, kernel_size=3, padding=1, residual=False, conv_block=True,
                 interp_mode="bilinear", interp_size=None, activation_type="silu", dropout=0.1, debug=False):
        super().__init__()
        self.activation = _act_module(activation_type)
        self.dropout = nn.Dropout(dropout)
        if conv_block:
            self.conv = nn.Sequential(
                Conv2dResBlock(in_dim, out_dim, kernel_size=kernel_size, padding=padding, residual=residual,
                               dropout=dropout, activation_type=default(activation_type, "silu")),
                self.dropout, self.activation)
        self.conv_block = conv_block
        self.interp_size = interp_size
        self.interp_mode = interp_mode
        self.debug = debug

    def features_nhwc(self) -> bool:
        """True when the conv block runs channels-last on the implicit-GEMM convolution (ops.conv3x3_nhwc): the
        intermediate feature map is then (B, H1, W1, C) and no layout change happens anywhere in the scaler."""
        return (self.conv_block and self.interp_mode == "bilinear" and self.conv[0].plain()
                and ops.conv3x3_nhwc_ok(self.conv[0].conv[0]))

    def forward_features(self, x, in_nhwc=False, out_nhwc=False, want_factor=False):
        """Everything but the final resize, at the intermediate size: (B, C, H1, W1) channels-first, or (B, H1, W1, C)
        with ``out_nhwc`` (only when ``features_nhwc()``).
        ``want_factor``: returns (features, fac-or-None).  With a tensor in the second place the two SiLUs behind the
        convolution ran on its epilogue (ops.conv3x3_nhwc(act2=True): dropouts off, channels-last) and the caller MUST be
        the only consumer of the features and multiply their gradient by ``fac`` (ops.upsample_fc(in_factor=fac))."""
        if want_factor:
            return self._features(x, in_nhwc, out_nhwc, True)
        return self._features(x, in_nhwc, out_nhwc, False)[0]

    def _features(self, x, in_nhwc, out_nhwc, may_fuse):
        if self.interp_mode != "bilinear":
            raise NotImplementedError(f"interp_mode={self.interp_mode!r}: only bilinear has a HIP path")
        if out_nhwc and not self.features_nhwc():
            raise ValueError("Interp2dUpsample: channels-last features need the implicit-GEMM conv block")
        x = _resize(x, self.interp_size[0], None, in_nhwc=in_nhwc, out_nhwc=out_nhwc)
        if self.conv_block:
            blk = self.conv[0]
            if blk.plain():       # conv -> drop -> act -> drop -> act: the four elementwise stages in one pass
                no_drop = not self.training or (blk.conv[1].p == 0 and self.dropout.p == 0)
                if (may_fuse and out_nhwc and no_drop and _fuse_act2[0] and ops.conv3x3_nhwc_implicit(x)
                        and _act_name(blk.activation) == "silu" and _act_name(self.activation) == "silu"):
                    return ops.conv3x3_nhwc(x, blk.conv[0].weight, act2=True)       # ... or on the product's epilogue
                y = ops.conv3x3_nhwc(x, blk.conv[0].weight) if out_nhwc else blk.conv[0](x)
                x = ops.drop_act(y, blk.conv[1].p, _act_name(blk.activation), self.training,
                                 self.dropout.p, _act_name(self.activation))
            else:
                x = ops.drop_act(blk(x), self.dropout.p, _act_name(self.activation), self.training)
        return x, None

    def forward(self, x, in_nhwc=False, out_nhwc=False):
        """x (B, C, H, W), or (B, H, W, C) with ``in_nhwc``; the layout changes ride on the resizes."""
        mid = self.features_nhwc()
        return _resize(self.forward_features(x, in_nhwc, mid), self.interp_size[1], None, in_nhwc=mid,
                       out_nhwc=out_nhwc)


# --------------------------------------------------------------------------------------- attention (HIP)
class SimpleAttention(nn.Module):
    """Softmax-free attention with per-head LayerNorm and coordinate concatenation.

    Same parameters / state_dict keys as the reference (layers.py:793-828): ``linears.{0,1,2}``,
    ``norm_K.{i}``, ``norm_V.{i}`` (galerkin) or ``norm_Q.{i}`` (fourier), ``fc``.  The HIP path covers
    self-attention (query is key is value) of the 'galerkin' and 'fourier' ('integral', 'local') types
    with norm_type='layer'; other variants of the reference are baselines outside the hot path."""

    def __init__(self, n_head, d_model, pos_dim: int = 1, attention_type="fourier", dropout=0.1,
                 xavier_init=1e-4, diagonal_weight=1e-2, symmetric_init=False, norm=False,
                 norm_type="layer", eps=1e-5, debug=False):
        super().__init__()
        assert d_model % n_head == 0
        self.attention_type = attention_type
        self.d_k = d_model // n_head
        self.n_head = n_head
        self.pos_dim = pos_dim
        self.linears = nn.ModuleList([nn.Linear(d_model, d_model) for _ in range(3)])
        self.xavier_init = xavier_init
        self.diagonal_weight = diagonal_weight
        self.symmetric_init = symmetric_init
        if self.xavier_init > 0:
            self._reset_parameters()
        self.add_norm = norm
        self.norm_type = norm_type
        self.eps = eps
        if norm:
            self._get_norm(eps=eps)
        if pos_dim > 0:
            self.fc = nn.Linear(d_model + n_head * pos_dim, d_model)
        self.attn_weight = None
        self.dropout = nn.Dropout(dropout)
        self.debug = debug

    # -- init contract of the reference (layers.py:901-913)
    def _reset_parameters(self):
        for param in self.linears.parameters():
            if param.ndim > 1:
                xavier_uniform_(param, gain=self.xavier_init)
                if self.diagonal_weight > 0.0:
                    param.data += self.diagonal_weight * torch.eye(param.size(-1), dtype=torch.float)
                if self.symmetric_init:
                    param.data += param.data.T
            else:
                constant_(param, 0)

    def _get_norm(self, eps):
        if self.norm_type != "layer":
            raise NotImplementedError("only norm_type='layer' has a HIP path (the reference's 'instance' "
                                      "branch is unused by its configs)")
        mk = lambda: nn.ModuleList([nn.LayerNorm(self.d_k, eps=eps) for _ in range(self.n_head)])
        self.norm_K = mk()
        if self.attention_type in _LINEAR_FAMILY:
            self.norm_V = mk()
        else:
            self.norm_Q = mk()

    def _norm_pair(self):
        if self.attention_type in _LINEAR_FAMILY:
            return self.norm_K, self.norm_V, 0b110
        return self.norm_Q, self.norm_K, 0b011

    def _pack_groups(self):
        """The parameter lists _packed() concatenates, in concatenation order (optim.FlatClipAdam(model=...) lays each
        group out contiguously, and ops.packed_params then returns views instead of copies)."""
        groups = [[l.weight for l in self.linears]]
        if all(l.bias is not None for l in self.linears):
            groups.append([l.bias for l in self.linears])
        if self.add_norm:
            first, second, _ = self._norm_pair()
            groups.append([m.weight for m in first] + [m.weight for m in second])
            groups.append([m.bias for m in first] + [m.bias for m in second])
        return groups

    def _packed(self):
        wqkv = ops.packed_params([l.weight for l in self.linears])
        bqkv = ops.packed_params([l.bias for l in self.linears])
        gamma = beta = None
        mask = 0
        if self.add_norm:
            first, second, mask = self._norm_pair()
            gamma = ops.packed_params([m.weight for m in first] + [m.weight for m in second])
            beta = ops.packed_params([m.bias for m in first] + [m.bias for m in second])
            gamma = gamma.view(2, self.n_head, self.d_k)
            beta = beta.view(2, self.n_head, self.d_k)
        return wqkv, bqkv, gamma, beta, mask

    def fused_forward(self, x, pos=None, residual=None, sign=1.0, p_out=0.0, need_weights=True):
        """res + sign*dropout(attention(x)); the encoder layer's entry point.  ``need_weights=False`` lets the
        Fourier type run fused (no n x n matrix in HBM; the returned weight is None)."""
        if self.attention_type not in _HIP_ATTENTION:
            raise NotImplementedError(f"attention_type={self.attention_type!r} is outside the HIP hot path "
                                      "(galerkin / fourier only)")
        use_pos = pos is not None and self.pos_dim > 0
        if use_pos:
            assert pos.size(-1) == self.pos_dim
            wfc, bfc = self.fc.weight, self.fc.bias
        else:
            # reference layers.py:869-874, 894-897: without coordinates the heads are not widened and `fc` is skipped --
            # the same operator with zero coordinate columns and the identity in fc's place
            pos = None
            d = self.n_head * self.d_k
            key = (x.device, d)
            if getattr(self, "_eye", (None, None))[0] != key:
                self._eye = (key, torch.eye(d, dtype=torch.float32, device=x.device))
            wfc, bfc = self._eye[1], None
        wqkv, bqkv, gamma, beta, mask = self._packed()
        kind = "galerkin" if self.attention_type == "galerkin" else "fourier"
        out, w = ops.simple_attention(x, pos, wqkv, bqkv, gamma, beta, wfc, bfc,
                                      kind=kind, n_head=self.n_head, norm_mask=mask, eps=self.eps,
                                      res=residual, sign=sign, p_out=p_out, need_weights=need_weights)
        self.attn_weight = w
        return out, w

    def forward(self, query, key, value, pos=None, mask=None, weight=None):
        if mask is not None:
            if self.attention_type in _LINEAR_FAMILY:
                raise RuntimeError("linear attention does not support casual mask.")
            raise NotImplementedError("attention masks are outside the HIP hot path")
        if weight is not None:
            raise NotImplementedError("weighted attention is outside the HIP hot path")
        if not (query is key and key is value):
            raise NotImplementedError("the HIP path implements self-attention (query is key is value)")
        return self.fused_forward(query, pos)


class FeedForward(nn.Module):
    """lr1 -> act -> dropout -> lr2 (layers.py:954-987); one fused HIP operator."""

    def __init__(self, in_dim=256, dim_feedforward: int = 1024, out_dim=None, batch_norm=False,
                 activation="relu", dropout=0.1):
        super().__init__()
        out_dim = default(out_dim, in_dim)
        self.lr1 = nn.Linear(in_dim, dim_feedforward)
        if activation == "silu":
            self.activation = nn.SiLU()
        elif activation == "gelu":
            self.activation = nn.GELU()
        else:
            self.activation = nn.ReLU()
        self.batch_norm = batch_norm
        if batch_norm:
            self.bn = nn.BatchNorm1d(dim_feedforward)
        self.lr2 = nn.Linear(dim_feedforward, out_dim)
        self.dropout = nn.Dropout(dropout)

    def fused_forward(self, x, residual=None, p_out=0.0):
        if self.batch_norm:
            raise NotImplementedError("batch_norm=True is outside the HIP hot path (False in every config)")
        p_h = self.dropout.p if self.training else 0.0
        if isinstance(self.activation, nn.GELU):
            # activation='gelu' (reference layers.py:968; no config selects it): the two GEMMs with the erf GELU and its
            # dropout as one elementwise pass between them -- the GEMM epilogues carry relu / silu only
            hid = ops.drop_act(ops.linear(x, self.lr1.weight, self.lr1.bias), 0.0, _act_name(self.activation),
                               training=True, p2=p_h)
            y = ops.linear(hid, self.lr2.weight, self.lr2.bias, p_drop=p_out)
            return y if residual is None else residual + y
        return ops.feed_forward(x, self.lr1.weight, self.lr1.bias, self.lr2.weight, self.lr2.bias,
                                res=residual, act=_act_name(self.activation), p_h=p_h, p_out=p_out)

    def forward(self, x):
        return self.fused_forward(x)


# --------------------------------------------------------------------------------------- spectral convs (HIP)
class SpectralConv1d(nn.Module):
    def __init__(self, in_dim, out_dim, modes: int, n_grid=None, dropout=0.1, return_freq=False,
                 activation="silu", debug=False):
        super().__init__()
        self.linear = nn.Linear(in_dim, out_dim)
        self.modes = modes
        self.activation = _act_module(activation)
        self.n_grid = n_grid
        self.fourier_weight = Parameter(torch.empty(in_dim, out_dim, modes, 2))
        xavier_normal_(self.fourier_weight, gain=1 / (in_dim * out_dim))
        self.dropout = nn.Dropout(dropout)
        self.return_freq = return_freq
        self.debug = debug

    def forward(self, x):
        act = _act_name(self.activation)
        args = (self.linear.weight, self.linear.bias, self.fourier_weight, self.modes)
        if self.training and self.dropout.p > 0:
            # the reference drops the input of the FFT branch only (layers.py:1083-1084: res = linear(x); x = dropout(x)).
            # By linearity  spec(xs) + lin(x) = [spec(xs) + lin(xs)] + lin(x - xs): the fused operator on the dropped input
            # plus one pointwise Linear on the difference (p is 0 in every shipped config)
            xs = ops.dropout(x, self.dropout.p, True)
            pre = spectral.spectral_conv1d(xs, *args, "none", return_freq=self.return_freq)
            pre, ft = pre if self.return_freq else (pre, None)
            y = self.activation(pre + ops.linear(x - xs, self.linear.weight, None))
            return (y, ft) if self.return_freq else y
        return spectral.spectral_conv1d(x, *args, act, return_freq=self.return_freq)


class SpectralConv2d(nn.Module):
    def __init__(self, in_dim, out_dim, modes: int, n_grid=None, dropout=0.1, norm="ortho",
                 activation="silu", return_freq=False, debug=False):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.linear = nn.Linear(in_dim, out_dim)
        self.modes = modes
        self.activation = _act_module(activation)
        self.n_grid = n_grid
        self.fourier_weight = nn.ParameterList(
            [Parameter(torch.empty(in_dim, out_dim, modes, modes, 2)) for _ in range(2)])
        for param in self.fourier_weight:
            xavier_normal_(param, gain=1 / (in_dim * out_dim) * np.sqrt(in_dim + out_dim))
        self.dropout = nn.Dropout(dropout)
        self.norm = norm
        self.return_freq = return_freq
        self.debug = debug

    def forward(self, x):
        if x.ndim == 4:
            n = x.size(1)
            assert x.size(1) == x.size(2)
        elif x.ndim == 3:
            n = int(x.size(1) ** 0.5)
        else:
            raise ValueError("Dimension not implemented")
        if self.norm != "ortho":
            raise NotImplementedError("only norm='ortho' (the reference default) has a HIP path")
        B, flat = x.size(0), x.ndim == 3
        h = x.reshape(B, n, n, self.in_dim)
        args = (self.linear.weight, self.linear.bias, self.fourier_weight[0], self.fourier_weight[1], self.modes)
        ft = None
        if self.training and self.dropout.p > 0:
            # dropout on the input of the FFT branch only (layers.py:1172-1173); see SpectralConv1d.forward
            hs = ops.dropout(h, self.dropout.p, True)
            pre = spectral.spectral_conv2d(hs, *args, "none", return_freq=self.return_freq)
            pre, ft = pre if self.return_freq else (pre, None)
            y = self.activation(pre + ops.linear(h - hs, self.linear.weight, None))
        else:
            y = spectral.spectral_conv2d(h, *args, _act_name(self.activation), return_freq=self.return_freq)
            y, ft = y if self.return_freq else (y, None)
        y = y.reshape(B, n * n, self.out_dim) if flat else y
        return (y, ft) if self.return_freq else y
