"""CPU oracle for the Galerkin/Fourier encoder + spectral decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this file:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / reported baseline.

This is a clean-room, functional restatement (plain torch CPU ops, any float
dtype) of the algorithm the reference implements with ``nn.Module`` classes.
Every function works on a flat ``state_dict``-style mapping ``{key: tensor}``
whose keys are the reference's own parameter names, so a reference checkpoint
(or one of our modules' ``state_dict()``) can be fed in unchanged.

Reference locations restated here (all under /root/reference/libs):
  * per-head LayerNorm, pos concat, head split/merge ... layers.py:829-899
  * galerkin  Q (K^T V / n)                           ... layers.py:708-734
  * fourier   (Q K^T / sqrt(d) / n) V                 ... layers.py:672-705
  * pointwise FFN                                      ... layers.py:954-987
  * encoder layer residual wiring                      ... model.py:104-140
  * SpectralConv1d / SpectralConv2d                    ... layers.py:1077-1106, 1153-1197
  * SpectralRegressor / PointwiseRegressor             ... model.py:603-637, 507-529
  * interp down/up scalers (conv + bilinear)           ... layers.py:483-512, 658-670
  * FourierTransformer2D / SimpleTransformer / 2DLite  ... model.py:953-1017, 760-807, 1197-1226

Parity pin: the reference ships no tests or golden vectors.  The pin is
``tests/golden/*.npz``: outputs and gradients of the *reference modules
themselves*, generated in the build container by ``tests/golden/make_golden.py``
(which imports /root/reference) and checked against this file by
``tests/test_oracle_golden.py``.

Dropout: every ``nn.Dropout`` of the reference is treated as identity (p=0 /
eval) except the attention-matrix dropout, which the reference applies with
p=0.5 in train *and* eval (layers.py:700-701, 730-731).  That one is controlled
by ``attn_drop``: ``None`` = identity, a tensor = multiplicative mask replay
(values 0 or 2), ``"random"`` = fresh Bernoulli(0.5) mask like the reference.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Mapping, Optional, Sequence, Union

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
AttnDrop = Union[None, str, Tensor]


# --------------------------------------------------------------------------- mask-replay audit
# A parity run that replays the device's ReLU decisions (relu_mask / relu_masks below) must not be able to hide a wrong
# mask behind the replay: with a list installed here, every replayed gate appends a record of how the replayed decisions
# compare with this function's OWN `pre > 0` -- {"where", "n", "flipped", "max_rel"}: elements, elements whose replayed
# decision differs, and the largest |pre| / rms(pre) among those.  Decisions may differ only where the pre-activation lies
# within rounding of the kink; tests/_util.py::check_replay_audit holds the records to that.
_replay_audit: List[Optional[list]] = [None]


def set_replay_audit(sink: Optional[list]) -> None:
    _replay_audit[0] = sink


def _audit(where: str, pre: Tensor, dec: Tensor) -> None:
    sink = _replay_audit[0]
    if sink is None:
        return
    with torch.no_grad():
        flip = (pre > 0) != dec.to(torch.bool)
        nf = int(flip.sum())
        rms = float(pre.double().pow(2).mean().sqrt())
        mx = float(pre[flip].abs().max()) / max(rms, 1e-300) if nf else 0.0
        sink.append({"where": where, "n": pre.numel(), "flipped": nf, "max_rel": mx})


# --------------------------------------------------------------------------- helpers
def _sub(sd: Mapping[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    """Sub-dict of ``sd`` with ``prefix`` stripped."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def _act(name: Optional[str], default: str = "silu") -> Callable[[Tensor], Tensor]:
    name = default if name is None else name
    if name == "silu":
        return F.silu
    if name == "gelu":
        return F.gelu
    if name == "identity":
        return lambda t: t
    return F.relu


def _apply_attn_drop(m: Tensor, attn_drop: AttnDrop) -> Tensor:
    if attn_drop is None:
        return m
    if isinstance(attn_drop, str):
        assert attn_drop == "random"
        keep = (torch.rand_like(m) >= 0.5).to(m.dtype)
        return m * keep * 2.0
    return m * attn_drop.to(m.dtype)


# --------------------------------------------------------------------------- attention
def head_layernorm(t: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tensor:
    """t: (B,h,n,dk); gamma/beta: (h,dk).  Separate affine LayerNorm per head over dk
    (layers.py:846-851 builds one nn.LayerNorm(d_k) per head)."""
    mu = t.mean(dim=-1, keepdim=True)
    var = ((t - mu) ** 2).mean(dim=-1, keepdim=True)
    th = (t - mu) / torch.sqrt(var + eps)
    return th * gamma[None, :, None, :] + beta[None, :, None, :]


def _stack_norm(sd: Mapping[str, Tensor], name: str, n_head: int):
    g = torch.stack([sd[f"{name}.{i}.weight"] for i in range(n_head)])
    b = torch.stack([sd[f"{name}.{i}.bias"] for i in range(n_head)])
    return g, b


def simple_attention(sd: Mapping[str, Tensor], x: Tensor, pos: Optional[Tensor], *,
                     n_head: int, attention_type: str = "galerkin", norm: bool = True,
                     eps: float = 1e-5, attn_drop: AttnDrop = None):
    """Self-attention block without softmax.  Returns (out, attn_matrix).

    sd keys: linears.{0,1,2}.{weight,bias}, norm_K/norm_V (galerkin) or
    norm_K/norm_Q (fourier) .{i}.{weight,bias}, fc.{weight,bias}."""
    B, n, d = x.shape
    dk = d // n_head
    q, k, v = (F.linear(x, sd[f"linears.{i}.weight"], sd[f"linears.{i}.bias"])
               .reshape(B, n, n_head, dk).permute(0, 2, 1, 3) for i in range(3))
    linear_family = attention_type in ("galerkin", "linear", "global")
    if norm:
        gk, bk = _stack_norm(sd, "norm_K", n_head)
        k = head_layernorm(k, gk, bk, eps)
        if linear_family:
            gv, bv = _stack_norm(sd, "norm_V", n_head)
            v = head_layernorm(v, gv, bv, eps)
        else:
            gq, bq = _stack_norm(sd, "norm_Q", n_head)
            q = head_layernorm(q, gq, bq, eps)
    use_pos = pos is not None and pos.shape[-1] > 0
    if use_pos:
        pp = pos[:, None].expand(B, n_head, n, pos.shape[-1]).to(x.dtype)
        q, k, v = (torch.cat([pp, t], dim=-1) for t in (q, k, v))
    if linear_family:
        assert attention_type == "galerkin", "only the softmax-free branch is on the hot path"
        m = torch.einsum("bhnd,bhne->bhde", k, v) / n
        m = _apply_attn_drop(m, attn_drop)
        o = torch.einsum("bhnd,bhde->bhne", q, m)
    else:
        assert attention_type in ("fourier", "integral", "local")
        dkp = q.shape[-1]
        m = torch.einsum("bhnd,bhmd->bhnm", q, k) / math.sqrt(dkp) / n
        m = _apply_attn_drop(m, attn_drop)
        o = torch.einsum("bhnm,bhmd->bhnd", m, v)
    o = o.permute(0, 2, 1, 3).reshape(B, n, -1)
    if use_pos:
        o = F.linear(o, sd["fc.weight"], sd["fc.bias"])
    return o, m


def feed_forward(sd: Mapping[str, Tensor], x: Tensor, activation: str = "relu",
                 relu_mask: Optional[Tensor] = None) -> Tensor:
    """layers.py:979-987.  relu_mask (0/1, shape of the hidden activation): the ReLU decisions are taken from the mask
    instead of the sign of the pre-activation -- ReLU mask replay for parity runs at sizes where some pre-activation of
    the ~1e7 lies within fp32 rounding of the kink, so that its derivative is undefined at fp32 resolution (any two
    fp32 implementations, the reference on two machines included, may disagree there; each such element moves the
    parameter gradients by ~1e-4 relative)."""
    pre = F.linear(x, sd["lr1.weight"], sd["lr1.bias"])
    if relu_mask is not None:
        _audit("ff", pre, relu_mask)
    h = pre * relu_mask.to(pre.dtype) if relu_mask is not None else _act(activation, "relu")(pre)
    return F.linear(h, sd["lr2.weight"], sd["lr2.bias"])


def encoder_layer(sd: Mapping[str, Tensor], x: Tensor, pos: Optional[Tensor], *,
                  n_head: int, attention_type: str = "galerkin", layer_norm: bool = False,
                  attn_norm: Optional[bool] = None, norm_eps: float = 1e-5,
                  residual_type: Optional[str] = "add", activation_type: str = "relu",
                  attn_drop: AttnDrop = None, return_attn: bool = False, relu_mask: Optional[Tensor] = None):
    """One encoder layer (model.py:104-140) with all nn.Dropout = identity."""
    if attn_norm is None:
        attn_norm = not layer_norm
    if (not layer_norm) and (not attn_norm):
        attn_norm = True
    att, m = simple_attention(_sub(sd, "attn."), x, pos, n_head=n_head,
                              attention_type=attention_type, norm=attn_norm,
                              eps=norm_eps, attn_drop=attn_drop)
    if residual_type in ("add", "plus") or residual_type is None:
        x = x + att
    else:
        x = x - att
    d = x.shape[-1]
    if layer_norm:
        x = F.layer_norm(x, (d,), sd["layer_norm1.weight"], sd["layer_norm1.bias"], norm_eps)
    x = x + feed_forward(_sub(sd, "ff."), x, activation_type,
                         relu_mask=None if relu_mask is None else relu_mask.reshape(x.shape[0], x.shape[1], -1))
    if layer_norm:
        x = F.layer_norm(x, (d,), sd["layer_norm2.weight"], sd["layer_norm2.bias"], norm_eps)
    return (x, m) if return_attn else x


# --------------------------------------------------------------------------- spectral convs
def _cmul(a_re, a_im, w, eq):
    """Complex channel mix with real-pair weights w[..., 2] (layers.py:1066-1075, 1143-1151)."""
    w_re, w_im = w[..., 0], w[..., 1]
    o_re = torch.einsum(eq, a_re, w_re) - torch.einsum(eq, a_im, w_im)
    o_im = torch.einsum(eq, a_im, w_re) + torch.einsum(eq, a_re, w_im)
    return o_re, o_im


def spectral_conv2d(sd: Mapping[str, Tensor], x: Tensor, *, modes: int,
                    activation: Optional[str] = "silu", norm: str = "ortho", return_freq: bool = False,
                    spec_mask: Optional[Tensor] = None):
    """x: (B,n,n,Cin) or (B,n*n,Cin) -> same leading shape with Cout.  spec_mask: multiplicative dropout mask on the
    input of the FFT branch only (layers.py:1172-1173: res = linear(x); x = dropout(x)); return_freq: also the
    zero-padded half spectrum (B,Cout,n,n//2+1) complex (layers.py:1194-1197)."""
    B = x.shape[0]
    flat = x.dim() == 3
    n = int(round(math.sqrt(x.shape[1]))) if flat else x.shape[1]
    cin = x.shape[-1]
    x = x.reshape(B, n, n, cin)
    res = F.linear(x, sd["linear.weight"], sd["linear.bias"])
    cout = res.shape[-1]
    if spec_mask is not None:
        x = x * spec_mask.to(x.dtype).reshape(x.shape)
    xf = torch.fft.rfft2(x.permute(0, 3, 1, 2), s=(n, n), norm=norm)
    m = modes
    of_re = x.new_zeros(B, cout, n, n // 2 + 1)
    of_im = x.new_zeros(B, cout, n, n // 2 + 1)
    lo_re, lo_im = _cmul(xf.real[:, :, :m, :m], xf.imag[:, :, :m, :m],
                         sd["fourier_weight.0"], "bixy,ioxy->boxy")
    hi_re, hi_im = _cmul(xf.real[:, :, -m:, :m], xf.imag[:, :, -m:, :m],
                         sd["fourier_weight.1"], "bixy,ioxy->boxy")
    # same write order as the reference: low block first, high block second
    of_re[:, :, :m, :m] = lo_re
    of_im[:, :, :m, :m] = lo_im
    of_re[:, :, -m:, :m] = hi_re
    of_im[:, :, -m:, :m] = hi_im
    oft = torch.complex(of_re, of_im)
    y = torch.fft.irfft2(oft, s=(n, n), norm=norm)
    y = _act(activation)(y.permute(0, 2, 3, 1) + res)
    y = y.reshape(B, n * n, cout) if flat else y
    return (y, oft) if return_freq else y


def spectral_conv1d(sd: Mapping[str, Tensor], x: Tensor, *, modes: int,
                    activation: Optional[str] = "silu", return_freq: bool = False):
    """x: (B,n,Cin) -> (B,n,Cout) (and the padded half spectrum (B,Cout,n//2+1) with return_freq, layers.py:1102-1106)."""
    B, n, _ = x.shape
    res = F.linear(x, sd["linear.weight"], sd["linear.bias"])
    cout = res.shape[-1]
    xf = torch.fft.rfft(x.permute(0, 2, 1), n=n, norm="ortho")
    o_re, o_im = _cmul(xf.real[:, :, :modes], xf.imag[:, :, :modes],
                       sd["fourier_weight"], "bix,iox->box")
    of_re = x.new_zeros(B, cout, n // 2 + 1)
    of_im = x.new_zeros(B, cout, n // 2 + 1)
    of_re[:, :, :modes] = o_re
    of_im[:, :, :modes] = o_im
    oft = torch.complex(of_re, of_im)
    y = torch.fft.irfft(oft, n=n, norm="ortho")
    y = _act(activation)(y.permute(0, 2, 1) + res)
    return (y, oft) if return_freq else y


def spectral_regressor(sd: Mapping[str, Tensor], x: Tensor, grid: Optional[Tensor], *,
                       modes: int, num_spectral_layers: int = 2, spacial_dim: int = 2,
                       spacial_fc: bool = False, activation: Optional[str] = "silu",
                       last_activation: bool = True) -> Tensor:
    if spacial_fc:
        x = F.linear(torch.cat([x, grid.to(x.dtype)], dim=-1), sd["fc.weight"], sd["fc.bias"])
    conv = spectral_conv2d if spacial_dim == 2 else spectral_conv1d
    for j in range(num_spectral_layers):
        a = activation
        if j == num_spectral_layers - 1 and not last_activation:
            a = "identity"
        x = conv(_sub(sd, f"spectral_conv.{j}."), x, modes=modes, activation=a)
    x = F.linear(x, sd["regressor.0.weight"], sd["regressor.0.bias"])
    x = _act(activation)(x)
    return F.linear(x, sd["regressor.2.weight"], sd["regressor.2.bias"])


def pointwise_regressor(sd: Mapping[str, Tensor], x: Tensor, grid: Optional[Tensor], *,
                        num_layers: int = 2, spacial_fc: bool = False,
                        activation: Optional[str] = "silu") -> Tensor:
    if spacial_fc:
        x = F.linear(torch.cat([x, grid.to(x.dtype)], dim=-1), sd["fc.weight"], sd["fc.bias"])
    act = F.silu if activation == "silu" else F.relu      # model.py:490: None -> ReLU here
    for j in range(num_layers):
        x = act(F.linear(x, sd[f"ff.{j}.0.weight"], sd[f"ff.{j}.0.bias"]))
    return F.linear(x, sd["out.weight"], sd["out.bias"])


# --------------------------------------------------------------------------- CNN scalers (not hot path)
def _interp(x: Tensor, size) -> Tensor:
    if isinstance(size, float):
        return F.interpolate(x, scale_factor=size, mode="bilinear",
                             recompute_scale_factor=True, align_corners=True)
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)


def interp_downscaler(sd: Mapping[str, Tensor], node: Tensor, *, interp_size,
                      activation: Optional[str] = "silu", relu_masks: Optional[Mapping] = None) -> Tensor:
    """node: (B,n,n,Cin) -> (B,nc,nc,Cout).  conv0 -> interp -> act -> conv1,2,3 -> cat -> interp -> act
    (layers.py:483-512).  relu_masks (ReLU down-scaler only, mask replay for parity runs at sizes where some pre-activation
    lies within rounding of the kink): {"conv0": (B,C,n,n) with 1 / 0 = the decision to replay and anything else = take
    this function's own, "chain": three (B,nc',nc',c_i) 0/1 masks of conv1..3}.  The two ReLUs behind the interpolations act
    on non-negative values and have no kink to replay."""
    act = _act(activation)
    x = node.permute(0, 3, 1, 2)

    def gated(pre, m, where="scaler"):
        if m is None:
            return act(pre)
        m = m.to(pre.device)
        own = (pre > 0)
        dec = torch.where(m == 1, torch.ones_like(own), torch.where(m == 0, torch.zeros_like(own), own))
        _audit(where, pre, dec)
        return pre * dec.to(pre.dtype)

    rm = relu_masks or {}
    chain = rm.get("chain") or [None, None, None]
    chain = [None if m is None else m.permute(0, 3, 1, 2) for m in chain]
    x = gated(F.conv2d(x, sd["downsample.conv0.conv.0.weight"], padding=1), rm.get("conv0"), "scaler.conv0")
    x = act(_interp(x, interp_size[0]))
    x1 = gated(F.conv2d(x, sd["downsample.conv1.conv.0.weight"], padding=1), chain[0], "scaler.conv1")
    x2 = gated(F.conv2d(x1, sd["downsample.conv2.conv.0.weight"], padding=1), chain[1], "scaler.conv2")
    x3 = gated(F.conv2d(x2, sd["downsample.conv3.conv.0.weight"], padding=1), chain[2], "scaler.conv3")
    out = torch.cat([x1, x2, x3], dim=1)
    out = act(_interp(out, interp_size[1]))
    return out.permute(0, 2, 3, 1)


def interp_upscaler(sd: Mapping[str, Tensor], x: Tensor, *, interp_size,
                    activation: Optional[str] = "silu") -> Tensor:
    """x: (B,nc,nc,C) -> (B,nf,nf,C).  interp -> conv -> act -> act -> interp (layers.py:640-670)."""
    act = _act(activation)
    x = x.permute(0, 3, 1, 2)
    x = _interp(x, interp_size[0])
    x = act(act(F.conv2d(x, sd["upsample.conv.0.conv.0.weight"], padding=1)))
    x = _interp(x, interp_size[1])
    return x.permute(0, 2, 3, 1)


# --------------------------------------------------------------------------- whole models
def _enc_kwargs(cfg: Mapping) -> dict:
    layer_norm = bool(cfg.get("layer_norm"))
    eps = cfg.get("norm_eps")
    return dict(n_head=cfg["n_head"], attention_type=cfg["attention_type"],
                layer_norm=layer_norm, attn_norm=cfg.get("attn_norm"),
                norm_eps=1e-5 if eps is None else eps)


def fourier_transformer_2d(sd: Mapping[str, Tensor], cfg: Mapping, node: Tensor, pos: Tensor,
                           grid: Tensor, *, attn_drops: Optional[Sequence[AttnDrop]] = None,
                           normalizer=None, relu_masks: Optional[Sequence[Tensor]] = None,
                           ffn_activation: Optional[str] = None, scaler_masks: Optional[Mapping] = None) -> Tensor:
    """FourierTransformer2D.forward (model.py:953-1017) -> preds (B,n,n,n_targets).  ffn_activation: the FeedForward
    activation of the encoder layers when a probe has swapped it (the reference builds them with 'relu', model.py:1127-1141)."""
    B = node.shape[0]
    ns = int(round(math.sqrt(pos.shape[1])))
    nh = cfg["n_hidden"]
    if cfg.get("downscaler_size"):
        x = interp_downscaler(_sub(sd, "downscaler."), node, interp_size=cfg["downscaler_size"],
                              activation=cfg.get("downscaler_activation"), relu_masks=scaler_masks)
    else:
        x = torch.cat([node, pos.reshape(B, ns, ns, -1)], dim=-1)
        x = F.linear(x, sd["downscaler.id.weight"], sd["downscaler.id.bias"])
    x = x.reshape(B, -1, nh)
    ek = _enc_kwargs(cfg)
    if ffn_activation is not None:
        ek["activation_type"] = ffn_activation
    for li in range(cfg["num_encoder_layers"]):
        ad = None if attn_drops is None else attn_drops[li]
        x = encoder_layer(_sub(sd, f"encoder_layers.{li}."), x, pos, attn_drop=ad,
                          relu_mask=None if relu_masks is None else relu_masks[li], **ek)
    x = x.reshape(B, ns, ns, nh)
    if cfg.get("upscaler_size"):
        x = interp_upscaler(_sub(sd, "upscaler."), x, interp_size=cfg["upscaler_size"],
                            activation=cfg.get("upscaler_activation"))
    rs = _sub(sd, "regressor.")
    if cfg["decoder_type"] == "ifft2":
        x = spectral_regressor(rs, x, grid, modes=cfg["fourier_modes"],
                               num_spectral_layers=cfg["num_regressor_layers"],
                               spacial_dim=cfg["spacial_dim"], spacial_fc=cfg["spacial_fc"],
                               activation=cfg.get("regressor_activation"),
                               last_activation=bool(cfg.get("last_activation")))
    else:
        x = pointwise_regressor(rs, x, grid, num_layers=cfg["num_regressor_layers"],
                                spacial_fc=cfg["spacial_fc"],
                                activation=cfg.get("regressor_activation"))
    if normalizer is not None:
        # model.py:1005-1006; the regressor itself is built without a normalizer (model.py:1157-1172)
        x = x * (normalizer["std"] + normalizer["eps"]) + normalizer["mean"]
    if cfg.get("boundary_condition") == "dirichlet":
        x = F.pad(x[:, 1:-1, 1:-1], (0, 0, 1, 1, 1, 1))
    return x


def simple_transformer_1d(sd: Mapping[str, Tensor], cfg: Mapping, node: Tensor, pos: Tensor, *,
                          attn_drops: Optional[Sequence[AttnDrop]] = None) -> Tensor:
    """SimpleTransformer.forward (model.py:760-807) -> preds (B,n,n_targets)."""
    x = F.linear(node, sd["feat_extract.id.weight"], sd["feat_extract.id.bias"])
    ek = _enc_kwargs(cfg)
    ek["residual_type"] = cfg.get("residual_type")
    ek["activation_type"] = cfg.get("attn_activation") or "relu"
    for li in range(cfg["num_encoder_layers"]):
        ad = None if attn_drops is None else attn_drops[li]
        x = encoder_layer(_sub(sd, f"encoder_layers.{li}."), x, pos, attn_drop=ad, **ek)
    return spectral_regressor(_sub(sd, "regressor."), x, None, modes=cfg["fourier_modes"],
                              num_spectral_layers=cfg["num_regressor_layers"],
                              spacial_dim=cfg.get("spacial_dim") or cfg["pos_dim"],
                              spacial_fc=bool(cfg.get("spacial_fc")),
                              activation=cfg.get("regressor_activation"))


def fourier_transformer_2d_lite(sd: Mapping[str, Tensor], cfg: Mapping, node: Tensor, pos: Tensor,
                                grid: Tensor, *,
                                attn_drops: Optional[Sequence[AttnDrop]] = None,
                                relu_masks: Optional[Sequence[Tensor]] = None) -> Tensor:
    """FourierTransformer2DLite.forward (model.py:1197-1226)."""
    B = node.shape[0]
    ng = grid.shape[1]
    x = torch.cat([node.reshape(B, -1, node.shape[-1]), pos], dim=-1)
    x = F.linear(x, sd["feat_extract.id.weight"], sd["feat_extract.id.bias"])
    ek = _enc_kwargs(cfg)
    for li in range(cfg["num_encoder_layers"]):
        ad = None if attn_drops is None else attn_drops[li]
        x = encoder_layer(_sub(sd, f"encoder_layers.{li}."), x, pos, attn_drop=ad,
                          relu_mask=None if relu_masks is None else relu_masks[li], **ek)
    x = x.reshape(B, ng, ng, -1)
    return spectral_regressor(_sub(sd, "regressor."), x, grid, modes=cfg["fourier_modes"],
                              num_spectral_layers=cfg["num_regressor_layers"],
                              spacial_dim=cfg.get("spacial_dim") or cfg["pos_dim"],
                              spacial_fc=bool(cfg.get("spacial_fc")),
                              activation=cfg.get("regressor_activation"))


# --------------------------------------------------------------------------- utilities for the checker
def rel_l2(a: Tensor, b: Tensor) -> float:
    """||a-b|| / ||b|| in float64."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


def grads_of(fn: Callable[..., Tensor], sd: Mapping[str, Tensor], inputs: Sequence[Tensor],
             cot: Tensor):
    """Run fn(sd, *inputs) with autograd on leaf copies; return (out, dinputs, dparams)."""
    sd_l = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ins = [t.detach().clone().requires_grad_(True) for t in inputs]
    out = fn(sd_l, *ins)
    leaves = ins + [v for v in sd_l.values() if v.requires_grad]
    gs = torch.autograd.grad(out, leaves, cot.to(out.dtype), allow_unused=True)
    dins = list(gs[:len(ins)])
    names = [k for k, v in sd_l.items() if v.requires_grad]
    dps = {k: g for k, g in zip(names, gs[len(ins):]) if g is not None}
    return out.detach(), dins, dps


def model_train_step_cpu(sd: Dict[str, Tensor], cfg: Mapping, node, pos, grid, target,
                         adam_state: dict, lr: float = 1e-3, clip: float = 0.99, attn_drops="random"):
    """One fwd + MSE + bwd + clip_grad_norm_ + Adam step in plain torch (cpu_baseline leg; the reference's step is
    utils_ft.py:676-681 around model.py:953-1017).  sd tensors must be leaf tensors with requires_grad=True; updated in
    place.  attn_drops: "random" (the reference's always-on attention dropout, fresh masks), None (identity), or one
    multiplicative mask per encoder layer (mask replay) -- the two deterministic forms serve the trajectory test."""
    params = [v for v in sd.values() if v.requires_grad]
    for p in params:
        p.grad = None
    if isinstance(attn_drops, str):
        attn_drops = [attn_drops] * cfg["num_encoder_layers"]
    dt = params[0].dtype
    pred = fourier_transformer_2d(sd, cfg, node.to(dt), pos.to(dt), grid.to(dt), attn_drops=attn_drops)
    loss = ((pred - target.to(dt)) ** 2).mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, clip)
    if "opt" not in adam_state:
        adam_state["opt"] = torch.optim.Adam(params, lr=lr)
    adam_state["opt"].step()
    return float(loss.detach())
