"""Autograd operators of the hot path, built on the C ABI (``_hip``).

Each ``torch.autograd.Function`` here enqueues hand-written HIP kernels for forward and
backward; torch only owns the tensors and the autograd graph.  Reference call sites
(under /root/reference/libs) are cited per operator.
"""
from __future__ import annotations

import math
import os

import torch
from torch.autograd import Function

from . import _hip as H

# ----------------------------------------------------------------------------------- dropout bookkeeping
# The reference applies F.dropout(p_attn) with the *default* p=0.5, training=True to the attention
# matrix in train and eval alike (layers.py:700-701, 730-731).  Modes:
#   "reference": stateless-RNG Bernoulli(0.5) mask, x2 rescale (default, reference-faithful)
#   "off"      : identity (exact-math parity runs)
#   "replay"   : multiply by explicit masks queued with push_attention_masks() (mask-replay parity)
_attn_mode = "reference"
_attn_masks = []


def set_attention_dropout(mode: str):
    global _attn_mode
    if mode not in ("reference", "off", "replay"):
        raise ValueError(mode)
    _attn_mode = mode
    _attn_masks.clear()


def get_attention_dropout() -> str:
    return _attn_mode


def push_attention_masks(masks):
    """Queue explicit multiplicative masks (values 0 or 2), consumed one per attention call."""
    _attn_masks.extend(masks)


# ReLU mask capture for parity runs: with a list installed, every FeedForward forward (ReLU, no hidden dropout) appends
# the boolean mask hidden > 0 it will differentiate with.  At bench sizes (~1e7 pre-activations per layer) some lie within
# fp32 rounding of the kink, where the derivative is decided by the last bit of the accumulation; the checker replays
# these decisions in its float64 model instead of comparing coin flips.
_relu_mask_sink = [None]


def set_relu_mask_sink(sink):
    """sink: a list to append the FeedForward ReLU masks to (in call order), or None to switch the capture off."""
    _relu_mask_sink[0] = sink


_scaler_mask_sink = [None]


def set_scaler_mask_sink(sink):
    """sink: a list that every scaler_conv_chain forward appends its three ReLU masks to ([B, H, W, width_i] booleans:
    output > 0, i.e. kept by the dropout and positive), or None.  Parity runs replay them in the checker, like
    set_relu_mask_sink's."""
    _scaler_mask_sink[0] = sink


# ----------------------------------------------------------------------------------- masked twins of data gradients
# A block of the encoder layer ends in  out = res + dropout(y)  (model.py:125, 132); its backward needs the incoming gradient
# twice: as it is (residual branch) and under that dropout's mask (three contractions).  The masked copy used to be one
# elementwise gt_dropout_apply pass per block and backward (12 per step at six layers).  Now the block that CONSUMES `out`
# produces it: its forward finds the mask parameters of its input registered here (keyed by the tensor's address), and the
# product of its backward that writes d(out) writes the masked copy too (gt_gemm_desc.c_masked).  Everything is advisory: a
# consumer that finds no twin for exactly its (tensor address, p, salt) -- another op in between, a gradient autograd
# accumulated from two consumers, a hook that replaced it -- runs the elementwise pass as before.
_fold_masks = [os.environ.get("GT_FOLD_MASKS", "1") != "0"]
_ffn_bwd_fused = [os.environ.get("GT_FFN_BWD_FUSED", "1") != "0"]      # data half of the FeedForward backward in one launch (gt_ffn_bwd)
_mask_hints = {}                # out.data_ptr() -> (p, salt, numel): "the gradient w.r.t. this tensor is wanted under this mask too"
_masked_twins = {}              # dx.data_ptr()  -> (masked copy, p, salt)


_fold_seq = [0]                 # forward blocks that use this registry, in call order


def _hint_output_mask(out, p: float, salt: int):
    """Called at the END of a block's forward.  A hint is only honoured by the very next block's forward (an address that
    the allocator hands out again later must not resurrect it), and any forward activity drops leftover twins of an earlier
    backward."""
    _masked_twins.clear()
    _mask_hints.clear()
    if _fold_masks[0] and p > 0:
        _mask_hints[out.data_ptr()] = (float(p), int(salt), out.numel(), _fold_seq[0])


def _wanted_mask(x):
    """Called at the START of a block's forward: (p, salt) under which the producer of x wants d(x) once more, or None."""
    _fold_seq[0] += 1
    e = _mask_hints.pop(x.data_ptr(), None) if _fold_masks[0] else None
    return (e[0], e[1]) if e is not None and e[2] == x.numel() and e[3] == _fold_seq[0] - 1 else None


def _offer_twin(dx, dxm, p: float, salt: int):
    if len(_masked_twins) > 64:
        _masked_twins.clear()
    _masked_twins[dx.data_ptr()] = (dxm, float(p), int(salt))


def _take_twin(g, p: float, salt: int):
    e = _masked_twins.pop(g.data_ptr(), None) if _fold_masks[0] else None
    if e is not None and e[1] == float(p) and e[2] == int(salt) and e[0].numel() == g.numel() and e[0].device == g.device:
        return e[0].view(g.shape)
    return None


_qkvnorm_fused = [True]         # QKV projection + head norm in one launch when the library supports the shape
_next_salt = H.next_salt        # call-site salt counter (rewound by _hip.set_seed / utils.get_seed)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------- elementwise dropout
class DropoutFn(Function):
    """Stateless-RNG dropout (gt_dropout_apply): one device-resident seed drives every mask of a
    training step, so a captured HIP graph replays with fresh masks and backward regenerates the
    forward mask from (seed, salt) instead of storing it."""

    @staticmethod
    def forward(ctx, x, p: float):
        xc = _c(x)
        ctx.cfg = (p, _next_salt(1))
        return H.dropout_apply(xc, H.dropout_desc(p, ctx.cfg[1], x.device))

    @staticmethod
    def backward(ctx, g):
        p, salt = ctx.cfg
        return H.dropout_apply(_c(g), H.dropout_desc(p, salt, g.device)), None


def dropout(x, p: float, training: bool = True):
    """Drop-in for nn.Dropout.forward on device tensors."""
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, float(p))


class DropActFn(Function):
    """y = act2(drop2(act1(drop1(x)))) in one elementwise pass; backward recomputes from x (gt_dropact_*)."""

    @staticmethod
    def forward(ctx, x, p1: float, act1: int, p2: float, act2: int):
        xc = _c(x)
        ctx.cfg = (p1, _next_salt(1) if p1 > 0 else 0, act1, p2, _next_salt(1) if p2 > 0 else 0, act2)
        ctx.save_for_backward(xc)
        dev = x.device
        return H.dropact_fwd(xc, H.dropout_desc(p1, ctx.cfg[1], dev), act1, H.dropout_desc(p2, ctx.cfg[4], dev), act2)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        p1, s1, act1, p2, s2, act2 = ctx.cfg
        dev = g.device
        gx = H.dropact_bwd(xc, _c(g), H.dropout_desc(p1, s1, dev), act1, H.dropout_desc(p2, s2, dev), act2)
        return gx, None, None, None, None


def drop_act(x, p1: float, act1: str, training: bool = True, p2: float = 0.0, act2: str = "none"):
    """act2(dropout(act1(dropout(x, p1)), p2)) -- the dropout -> activation tails of the conv blocks, fused."""
    if not training:
        p1 = p2 = 0.0
    return DropActFn.apply(x, float(p1), H.ACT_CODE[act1], float(p2), H.ACT_CODE[act2])


# ----------------------------------------------------------------------------------- bilinear resize
class ResizeFn(Function):
    """act(F.interpolate(x, size, mode='bilinear', align_corners=True)) with the layout change of the
    scaler boundaries fused in (layers.py:483-512, 658-670; model.py:675-687, 740-749)."""

    @staticmethod
    def forward(ctx, x, size, in_nhwc: bool, out_nhwc: bool, act: int):
        xc = _c(x)
        y = H.bilinear2d_fwd(xc, size, in_nhwc, out_nhwc, act)
        in_size = (xc.shape[1], xc.shape[2]) if in_nhwc else (xc.shape[2], xc.shape[3])
        ctx.cfg = (in_size, in_nhwc, out_nhwc, act)
        if act == H.ACT_RELU:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        in_size, in_nhwc, out_nhwc, act = ctx.cfg
        y = ctx.saved_tensors[0] if act == H.ACT_RELU else None
        return H.bilinear2d_bwd(_c(g), y, in_size, in_nhwc, out_nhwc, act), None, None, None, None


def bilinear_resize(x, size, in_nhwc: bool = False, out_nhwc: bool = False, act: str = None):
    """``size``: (Ho, Wo), or a float scale factor with the reference's recompute_scale_factor=True rule
    (output = floor(input * scale), then the align_corners scale is recomputed from the sizes)."""
    hi, wi = (x.shape[1], x.shape[2]) if in_nhwc else (x.shape[2], x.shape[3])
    if isinstance(size, float):
        size = (int(math.floor(hi * size)), int(math.floor(wi * size)))
    return ResizeFn.apply(x, (int(size[0]), int(size[1])), bool(in_nhwc), bool(out_nhwc), H.ACT_CODE[act])


_crb_bits = [os.environ.get("GT_CRB_BITS", "1") != "0"]


class Conv3x3ResizeFn(Function):
    """act(resize(act(dropout(conv3x3(x))))) in one pass, act = ReLU or SiLU -- the head of Interp2dEncoder
    (layers.py:483-495) when the input carries few channels and needs no gradient.  See gt_conv3x3_resize_fwd."""

    @staticmethod
    def forward(ctx, x, weight, size, p_drop: float, out_nhwc: bool = False, act: int = H.ACT_RELU):
        xc, wc = _c(x), _c(weight)
        salt = _next_salt(1)                     # the salt the stand-alone dropout would have drawn
        drop = H.dropout_desc(p_drop, salt, x.device) if p_drop > 0 else None
        # channels-last ReLU: the forward records its four ReLU / dropout decisions per (output pixel, channel) (4 bits each,
        # 1/8 of y), and the backward takes them from there instead of re-evaluating the convolution (GT_CRB_BITS=0: off).
        # SiLU: the backward re-evaluates the convolution and both activations from x; nothing but x and w is kept.
        want = bool(out_nhwc and _crb_bits[0] and ctx.needs_input_grad[1] and act == H.ACT_RELU)   # (nothing to record for an inference pass)
        r = H.conv3x3_resize_fwd(xc, wc, size, drop, out_nhwc, want_bits=want, act=act)
        y, bits = r if want else (r, None)
        ctx.save_for_backward(xc, wc, y if act == H.ACT_RELU else None, bits)
        ctx.cfg = (p_drop, salt, out_nhwc, act)
        return y

    @staticmethod
    def backward(ctx, g):
        xc, wc, y, bits = ctx.saved_tensors
        p_drop, salt, out_nhwc, act = ctx.cfg
        if ctx.needs_input_grad[0]:
            raise RuntimeError("conv3x3_resize: the fused path has no input gradient")
        drop = H.dropout_desc(p_drop, salt, g.device) if p_drop > 0 else None
        return None, H.conv3x3_resize_bwd(_c(g), y, xc, wc, drop, out_nhwc, bits=bits, act=act), None, None, None, None


def conv3x3_resize(x, weight, size, p_drop: float = 0.0, training: bool = True, out_nhwc: bool = False, act: str = "relu"):
    """out_nhwc: return (B, Ho, Wo, Cout) channels-last (same values, same dropout mask).  act: 'relu' or 'silu' (both
    activations of the block: Interp2dEncoder builds them from one activation_type)."""
    hi, wi = x.shape[2], x.shape[3]
    if isinstance(size, float):
        size = (int(math.floor(hi * size)), int(math.floor(wi * size)))
    if act not in ("relu", "silu"):
        raise NotImplementedError(f"conv3x3_resize: activation {act!r}")
    return Conv3x3ResizeFn.apply(x, weight, (int(size[0]), int(size[1])), float(p_drop) if training else 0.0,
                                 bool(out_nhwc), H.ACT_CODE[act])


class ResizeSegFn(Function):
    """act(F.interpolate(cat[x1, x2, x3], size, bilinear, align_corners=True)) on the padded three-segment buffer of
    scaler_conv_chain: (B, Hi, Wi, 3 segp) -> dense channels-last (B, Ho, Wo, C) (layers.py:508-512)."""

    @staticmethod
    def forward(ctx, x, Cc: int, size, seg: int, segp: int, act: int, relu_input: bool, in_factor=None):
        xc = _c(x)
        dact = None
        if act == H.ACT_SILU:                    # + silu'(resized value), the factor of the backward
            y, dact = H.bilinear2d_seg_fwd(xc, Cc, size, seg, segp, act, want_dact=True)
        else:
            y = H.bilinear2d_seg_fwd(xc, Cc, size, seg, segp, act)
        ctx.cfg = ((xc.shape[1], xc.shape[2]), seg, segp, act, relu_input)
        ctx.save_for_backward(y if act == H.ACT_RELU else dact, xc if relu_input else None, in_factor)
        return y

    @staticmethod
    def backward(ctx, g):
        in_size, seg, segp, act, relu_input = ctx.cfg
        y, xin, fac = ctx.saved_tensors
        dx = H.bilinear2d_seg_bwd(_c(g), y, in_size, seg, segp, act, x_gate=fac if fac is not None else xin,
                                  gate_mul=fac is not None)
        return dx, None, None, None, None, None, None, None


def bilinear_resize_seg(x, n_channels: int, size, seg: int, segp: int, act: str = None, relu_input: bool = False,
                        in_factor=None):
    """relu_input: x is the output of a ReLU (scaler_conv_chain's buffer) -- the gradient returned for x is already zeroed
    where x <= 0, which is what its producer's backward would do first (ScalerConvChainFn(grad_masked=True) skips it).
    in_factor: the same for any other activation -- a buffer shaped like x (scaler_conv_chain's second output: dropout
    scale x activation derivative) that the returned gradient is multiplied with."""
    hi, wi = x.shape[1], x.shape[2]
    if isinstance(size, float):
        size = (int(math.floor(hi * size)), int(math.floor(wi * size)))
    if relu_input and in_factor is not None:
        raise ValueError("bilinear_resize_seg: relu_input and in_factor are alternatives")
    return ResizeSegFn.apply(x, int(n_channels), (int(size[0]), int(size[1])), int(seg), int(segp), H.ACT_CODE[act],
                             bool(relu_input), in_factor)


class UpsampleFcFn(Function):
    """fc(cat[upsample(x), grid]) without ever materialising upsample(x).

    Reference: Interp2dUpsample's final F.interpolate (layers.py:658-670) followed by SpectralRegressor /
    PointwiseRegressor ``fc(torch.cat([x, grid], -1))`` (model.py:615-617, 507-512).  A pointwise Linear
    commutes with bilinear interpolation (the four weights sum to one), so

        fc(cat[up(x), grid]) = up(x W_x^T) + grid W_g^T + b

    x: (B, K, Hi, Wi) channels-first, or (B, Hi, Wi, K) with ``x_nhwc`` (what the scaler's conv block produces on
    its NCHW / implicit-GEMM path); weight (N, K+p); grid (B, Ho, Wo, p).  Returns (B, Ho, Wo, N).  Only valid when
    nothing (dropout) sits between the resize and the Linear -- the caller checks."""

    @staticmethod
    def forward(ctx, x, size, weight, bias, grid, x_nhwc, in_factor=None):
        H.need_f32_cuda(x, weight, bias, grid, in_factor)
        if in_factor is not None and not x_nhwc:
            raise NotImplementedError("ops.upsample_fc: in_factor needs channels-last features")
        if x_nhwc:
            B, Hi, Wi, K = x.shape
        else:
            B, K, Hi, Wi = x.shape
        N, p = weight.shape[0], grid.shape[-1]
        assert weight.shape[1] == K + p
        Ho, Wo = size
        xc, w, gc = _c(x), _c(weight), _c(grid)
        dev, HW = x.device, Hi * Wi
        z = torch.empty(B, Hi, Wi, N, dtype=torch.float32, device=dev)
        if x_nhwc:
            H.gemm(xc, w, z, B * HW, N, K, lda=K, ldb=K + p, ldc=N)
        else:
            H.gemm(xc, w, z, HW, N, K, layout_a=1, lda=HW, ldb=K + p, ldc=N, batch=(B, 1), a_bs=(K * HW, 0),
                   c_bs=(HW * N, 0))
        out = H.bilinear2d_fwd(z, (Ho, Wo), True, True, H.ACT_NONE, bias=bias, rp_a=gc.reshape(B, Ho, Wo, p),
                               rp_b=w[:, K:], rp_ldb=K + p)
        ctx.save_for_backward(xc, w, gc, None if in_factor is None else _c(in_factor))
        ctx.cfg = (B, K, Hi, Wi, Ho, Wo, N, p, bias is not None, x_nhwc)
        return out

    @staticmethod
    def backward(ctx, g):
        xc, w, gc, fac = ctx.saved_tensors
        B, K, Hi, Wi, Ho, Wo, N, p, has_b, x_nhwc = ctx.cfg
        dev, HW, To = g.device, Hi * Wi, B * Ho * Wo
        gg = _c(g)
        f32 = dict(dtype=torch.float32, device=dev)
        dz = H.bilinear2d_bwd(gg, None, (Hi, Wi), True, True, H.ACT_NONE)              # (B, Hi, Wi, N)
        dw = torch.empty(N, K + p, **f32)
        db = torch.empty(N, **f32) if has_b else None
        # d W_g = g^T grid  (+ d bias = column sums of g as a by-product)
        H.gemm(gg, gc, dw[:, K:], N, p, To, layout_a=1, layout_b=1, lda=N, ldb=p, ldc=K + p, split_k=0,
               a_colsum=db)
        dx = None
        if x_nhwc:
            # d W_x^T [K, N] = x^T dz over all B*HW pixels: the tall-skinny reduction (gt_tsmm.hip)
            dwxt = torch.empty(K, N, **f32)
            H.gemm(xc, dz, dwxt, K, N, B * HW, layout_a=1, layout_b=1, lda=K, ldb=N, ldc=N, split_k=0)
            dw[:, :K].copy_(dwxt.t())
            if ctx.needs_input_grad[0]:
                dx = torch.empty(B, Hi, Wi, K, **f32)
                if fac is None:
                    H.gemm(dz, w, dx, B * HW, K, N, layout_b=1, lda=N, ldb=K + p, ldc=K)
                else:       # in_factor: the producer's activation derivative rides on this product's epilogue
                    H.gemm(dz, w, dx, B * HW, K, N, layout_b=1, lda=N, ldb=K + p, ldc=K, aux_op=H.AUX_MUL,
                           aux=fac.reshape(B * HW, K), ldaux=K)
            return dx, None, dw, db, None, None, None
        # d W_x = sum_b dz_b^T x_b^T : one [N, K] slab per batch entry, reduced in a fixed order
        slabs = torch.empty(B, N, K, **f32)
        H.gemm(dz, xc, slabs, N, K, HW, layout_a=1, layout_b=0, lda=N, ldb=HW, ldc=K, batch=(B, 1),
               a_bs=(HW * N, 0), b_bs=(K * HW, 0), c_bs=(N * K, 0), split_k=0)
        dwx = torch.empty(N, K, **f32)
        H.slab_reduce(slabs, B, N * K, N * K, dwx)
        dw[:, :K].copy_(dwx)
        if ctx.needs_input_grad[0]:
            # dx_b [K, HW] = W_x^T dz_b^T
            dx = torch.empty(B, K, Hi, Wi, **f32)
            H.gemm(w, dz, dx, K, HW, N, layout_a=1, layout_b=0, lda=K + p, ldb=N, ldc=HW, batch=(B, 1),
                   b_bs=(HW * N, 0), c_bs=(K * HW, 0))
        return dx, None, dw, db, None, None, None


def upsample_fc(x, size, weight, bias, grid, x_nhwc: bool = False, in_factor=None):
    """in_factor (same shape as x, channels-last only): x is the output of conv3x3_nhwc(act2=True), whose backward expects
    the gradient already multiplied by this factor -- the data-gradient product here does it on its epilogue."""
    if grid.requires_grad:
        raise NotImplementedError("ops.upsample_fc: `grid` gets no gradient")
    return UpsampleFcFn.apply(x, (int(size[0]), int(size[1])), weight, bias, grid, bool(x_nhwc), in_factor)


# ----------------------------------------------------------------------------------- 3x3 convolution, channels-last
def conv3x3_nhwc_ok(conv: torch.nn.Conv2d) -> bool:
    """True when ``conv`` is a scaler-block convolution the implicit-GEMM path covers: 3x3, stride 1, zero padding 1,
    no bias / groups / dilation, both channel counts multiples of 16 and >= 96 (whole tiles of the split-operand ring
    kernel in the forward AND the data-gradient product), and a split-operand arithmetic mode."""
    return (isinstance(conv, torch.nn.Conv2d) and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and conv.bias is None and conv.padding_mode == "zeros"
            and conv.in_channels % 16 == 0 and conv.out_channels % 16 == 0
            and conv.in_channels >= 96 and conv.out_channels >= 96 and H.get_precision() != "f32"
            and _conv_implicit[0])


_plain_tiles = [os.environ.get("GT_PLAIN_TILES", "1") != "0"]          # K', V' head tiles without the LayerNorm affine
_dkv_ln_fused = [os.environ.get("GT_DKV_LN", "1") != "0"]               # gt_galerkin_dkv_ln vs gt_galerkin_dkv + gt_headnorm_bwd
_conv_implicit = [os.environ.get("GT_CONV_IMPLICIT", "1") != "0"]        # A/B switches (tools / tests)
# weight gradient of the implicit convolution: "hip" (default) = the nine-tap pixel contraction on the ring kernel,
# "miopen" = the library's channels-last wrw kernel on the same buffers.  Same box, back to back at B = 128 with the
# two-stream backward (tools/gpu_ab.sh): 33.31 / 33.46 vs 33.31 / 33.33 ms/step -- a tie; the library kernel is not
# the default because its solver choice is not ours: at B = 4 MIOpen has been seen to answer this layout with
# naive_conv_ab_nonpacked_wrw_nhwc (34 ms per call, profiles/README.md).
_conv_wgrad = [os.environ.get("GT_CONV_WGRAD", "hip") != "miopen"]
_conv_wgrad_planes = [os.environ.get("GT_CONV_WGRAD_PLANES", "1") != "0"]   # gt_conv3x3_wgrad_nhwc where it applies (A/B switch)



class _PackedParamsFn(Function):
    """cat(params, dim 0) of parameters that lie back to back in memory: the result is a VIEW over them (no copy), and
    the gradient is handed back as slices of the packed gradient (no copies either)."""

    @staticmethod
    def forward(ctx, *params):
        ctx.rows = [p.shape[0] for p in params]
        p0 = params[0]
        rows = sum(ctx.rows)
        out = p0.detach().as_strided((rows,) + tuple(p0.shape[1:]), p0.stride())
        return out

    @staticmethod
    def backward(ctx, g):
        outs, r0 = [], 0
        for r in ctx.rows:
            outs.append(g[r0:r0 + r])
            r0 += r
        return tuple(outs)


def packed_params(params):
    """torch.cat(params, 0) -- as a zero-copy view when the tensors follow each other in memory (FlatClipAdam(model=...)
    lays SimpleAttention's groups out that way), else as the copy it always was.  1-D parameters of equal length stack to
    [len(params) * n] (callers view it)."""
    ok = all(p.is_contiguous() and p.dtype == params[0].dtype and p.shape[1:] == params[0].shape[1:] for p in params)
    if ok:
        nxt = params[0].data_ptr()
        for p in params:
            if p.data_ptr() != nxt:
                ok = False
                break
            nxt += p.numel() * p.element_size()
    if ok and params[0].is_cuda:
        st = params[0].untyped_storage()
        last = params[-1]
        ok = last.data_ptr() + last.numel() * last.element_size() <= st.data_ptr() + st.nbytes()
    if ok and params[0].is_cuda:
        return _PackedParamsFn.apply(*params)
    return torch.cat(list(params), dim=0)


_gather_cache = {}


def _gathered(w, key, fn):
    """fn(w) for a pure re-layout `fn` (flips, permutes, zero padding, reshapes) as ONE index_select (+ one multiply by a 0 / 1
    mask when fn pads): the index table is worked out once per (fn, shape) by running fn on the element numbers.  The filter
    re-layouts of the scaler convolutions run every step (the weights change every step): flip + permute + zeros + copy +
    contiguous per filter were 26 ATen launches per step (tools/aten_in_step.py), now 8."""
    k = (key, tuple(w.shape), str(w.device))
    ent = _gather_cache.get(k)
    if ent is None:
        src = torch.arange(1, w.numel() + 1, dtype=torch.float64).reshape(w.shape)
        r = fn(src)
        flat = r.reshape(-1)
        idx = (flat.to(torch.int64) - 1).clamp_min(0).to(w.device)
        mask = (flat > 0).to(torch.float32).to(w.device) if bool((flat == 0).any()) else None
        ent = (idx, mask, tuple(r.shape))
        _gather_cache[k] = ent
    idx, mask, shape = ent
    out = w.detach().reshape(-1).index_select(0, idx)
    if mask is not None:
        out = out * mask
    return out.view(shape)


def _conv_k_order(w):
    """[N, 9 taps, C] filter -> [N, 9 C] in the contraction order of the implicit-GEMM kernel (gt_hip.h: cv_*): channel
    blocks of CB outermost, the nine taps of a block adjacent."""
    N, _, Cc = w.shape
    cb = 32 if Cc % 32 == 0 else 16
    return w.reshape(N, 9, Cc // cb, cb).permute(0, 2, 1, 3).reshape(N, 9 * Cc).contiguous()


class Conv3x3NhwcFn(Function):
    """y = conv2d(x, weight, padding=1) on channels-last activations, as implicit GEMMs on the split-operand ring kernel.

    Replaces ``nn.Conv2d(C, C', 3, padding=1, bias=False)`` of the scaler blocks (layers.py:98-100 inside
    Interp2dUpsample, layers.py:624-670) where the channel counts fill the 128 x 128 tiles (the up-scaler's
    n_hidden -> n_hidden convolution).  Forward: [pixels, 9 C] x [C', 9 C]^T with the nine shifted views of x read in
    place (gt_hip.h: cv_*).  Data gradient: the same product on gy with the taps reversed and the channel roles
    swapped.  Weight gradient: nine pixel-contracted products, one per tap, cut into K chunks on one launch of the ring
    kernel (images at least 16 pixels wide); GT_CONV_WGRAD=miopen (and narrower images) use the library's channels-last
    wrw kernel on the same buffers instead -- the two tie at B = 128."""

    @staticmethod
    def forward(ctx, x, weight, act2: bool = False):
        H.need_f32_cuda(x, weight)
        B, Hh, Ww, Cin = x.shape
        Cout = weight.shape[0]
        xc = _c(x)
        wf = _gathered(weight, "conv_fwd", lambda t: _conv_k_order(t.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)))   # [Cout][tap][Cin] -> k order
        y = torch.empty(B, Hh, Ww, Cout, dtype=torch.float32, device=x.device)
        ctx.prec = H.get_precision()             # the backward products run in the arithmetic of the forward
        ctx.save_for_backward(xc, weight)
        if not act2:
            H.gemm(xc, wf, y, B * Hh * Ww, Cout, 9 * Cin, lda=Cin, ldb=9 * Cin, ldc=Cout, conv=(Hh, Ww, Cin),
                   precision=ctx.prec)
            return y
        # act2: y = silu(silu(conv)) on the product's epilogue (GT_ACT_SILU2) + the factor silu' * silu'(silu), which the
        # CONSUMER's backward multiplies the gradient with (ops.upsample_fc(in_factor=fac)): what arrives here is then
        # already the gradient of the convolution's own output, and the backward below is the plain one
        need = any(ctx.needs_input_grad)         # (grad mode itself is off inside Function.forward)
        fac = torch.empty_like(y) if need else None
        H.gemm(xc, wf, y, B * Hh * Ww, Cout, 9 * Cin, lda=Cin, ldb=9 * Cin, ldc=Cout, conv=(Hh, Ww, Cin),
               act=H.ACT_SILU2, pre=fac, ldpre=Cout, precision=ctx.prec)
        if fac is None:
            fac = y.new_empty(0)
        ctx.mark_non_differentiable(fac)
        ctx.set_materialize_grads(False)         # (autograd filled a zero gradient the size of `fac` per step: 388 MB at C2)
        return y, fac

    @staticmethod
    def backward(ctx, gy, _gfac=None):
        if gy is None:
            return None, None, None
        xc, weight = ctx.saved_tensors
        B, Hh, Ww, Cin = xc.shape
        Cout = weight.shape[0]
        g = _c(gy)
        dx = dw = None
        dev = g.device
        if ctx.needs_input_grad[0]:
            # dx[pix][ci] = sum_tap sum_co gy[pix - shift(tap)][co] W[co][ci][tap]: tap' = 8 - tap has the opposite shift
            wd = _gathered(weight, "conv_dgrad", lambda t: _conv_k_order(t.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9, Cout)))     # [Cin][tap'][Cout]
            dx = torch.empty(B, Hh, Ww, Cin, dtype=torch.float32, device=g.device)
            with H.side_branch(dev, B * Hh * Ww):        # the data gradient next to the weight gradient below
                try:
                    H.gemm(g, wd, dx, B * Hh * Ww, Cin, 9 * Cout, lda=Cout, ldb=9 * Cout, ldc=Cin, conv=(Hh, Ww, Cout),
                           precision=ctx.prec)
                except H.GtNotSupported:         # gt_hip.h: on GT_ENOTSUP the caller takes the library convolution
                    dx = torch.ops.aten.convolution_backward(
                        g.permute(0, 3, 1, 2), xc.permute(0, 3, 1, 2), weight, None, [1, 1], [1, 1], [1, 1], False,
                        [0, 0], 1, [True, False, False])[0].permute(0, 2, 3, 1).contiguous()
        if ctx.needs_input_grad[1]:
            # dw[co][ci][tap] = sum_pix gy[pix][co] x[pix + shift(tap)][ci]
            dw = None
            hip_wgrad = Ww >= 16 and _conv_wgrad[0]
            if hip_wgrad and ctx.prec in H.SPLIT_EXACT and _conv_wgrad_planes[0]:
                # operands split once per block into LDS planes, nine taps co-resident (gt_convw.hip): images up to 80 wide
                try:
                    dw = H.conv3x3_wgrad_nhwc(g.reshape(-1, Cout), Cout, xc.reshape(-1, Cin), Cin, B, Hh, Ww, Cin, Cout,
                                              precision=ctx.prec)
                except H.GtNotSupported:
                    dw = None
            if hip_wgrad and dw is None:       # nine [Cout, Cin] products over the pixels, K chunks x taps on one launch
                                # + a fixed-order reduce (gt_hip.h: cv_wgrad)
                dw9 = torch.empty(9, Cout, Cin, dtype=torch.float32, device=g.device)
                try:
                    H.gemm(g, xc, dw9, Cout, Cin, B * Hh * Ww, layout_a=1, layout_b=1, lda=Cout, ldb=Cin, ldc=Cin,
                           batch=(9, 1), c_bs=(Cout * Cin, 0), split_k=0, conv=(Hh, Ww, Cin), conv_wgrad=True,
                           precision=ctx.prec)
                    dw = dw9.view(3, 3, Cout, Cin).permute(2, 3, 0, 1).contiguous()
                except H.GtNotSupported:
                    hip_wgrad = False
            if dw is None:      # the library's channels-last wrw kernel on the same buffers
                dw = torch.ops.aten.convolution_backward(
                    g.permute(0, 3, 1, 2), xc.permute(0, 3, 1, 2), weight.contiguous(memory_format=torch.channels_last),
                    None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1].contiguous()
        H.join_side(dev)
        return dx, dw, None


def conv3x3_nhwc_implicit(x) -> bool:
    """True when conv3x3_nhwc(x, .) runs the implicit GEMM (and so can carry act2)."""
    return x.shape[0] * x.shape[1] * x.shape[2] >= 96 and H.get_precision() != "f32"


def conv3x3_nhwc(x, weight, act2: bool = False):
    """x (B, H, W, Cin) channels-last, weight (Cout, Cin, 3, 3) -> (B, H, W, Cout); see Conv3x3NhwcFn.  Less than one
    tile of pixels (< 96) goes to the library convolution on the same buffers.
    act2: returns (silu(silu(conv)), fac) -- the caller hands `fac` to the ONLY consumer of the first result, whose backward
    must multiply the gradient by it (ops.upsample_fc(in_factor=fac)); implicit-GEMM path only (conv3x3_nhwc_implicit)."""
    if not conv3x3_nhwc_implicit(x):
        if act2:
            raise H.GtNotSupported("conv3x3_nhwc(act2=True) outside the implicit-GEMM path")
        # (the implicit GEMM exists on the split-operand engine only: gt_hip.h says GT_ENOTSUP -> the library convolution)
        return torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), weight, padding=1).permute(0, 2, 3, 1).contiguous()
    if act2:
        return Conv3x3NhwcFn.apply(x, weight, True)
    return Conv3x3NhwcFn.apply(x, weight)


# ----------------------------------------------------------------------------------- down-scaler convolution chain
_scaler_chain = [os.environ.get("GT_SCALER_CHAIN", "1") != "0"]          # A/B switch (tools / tests)


def scaler_chain_ok(convs, act_name: str) -> bool:
    """True when the three narrow 3x3 convolutions of Interp2dEncoder (layers.py:431-512: conv1, conv2, conv3, each
    conv -> dropout -> ReLU, outputs concatenated) can run as ``scaler_conv_chain``: plain 3x3 / stride 1 / zero padding 1 /
    bias-free, chained channel counts, the first input a multiple of 16 channels, ReLU (it commutes with the dropout
    scale, so both ride on the product's epilogue) or SiLU (round 6: the epilogue applies the dropout in front of it,
    GT_ACT_DROP_SILU), the split-operand arithmetic."""
    if not (_scaler_chain[0] and act_name in ("relu", "silu") and H.get_precision() in H.SPLIT_EXACT and len(convs) == 3):
        return False
    for c in convs:
        if not (isinstance(c, torch.nn.Conv2d) and tuple(c.kernel_size) == (3, 3) and tuple(c.stride) == (1, 1)
                and tuple(c.padding) == (1, 1) and tuple(c.dilation) == (1, 1) and c.groups == 1 and c.bias is None
                and c.padding_mode == "zeros"):
            return False
    c1, c2, c3 = convs
    return (c1.in_channels % 16 == 0 and c2.in_channels == c1.out_channels and c3.in_channels == c2.out_channels
            and 32 <= max(c.out_channels for c in convs) <= 64)


def _pad_filter(w, co_p, ci_p):
    """[co, ci, 3, 3] -> zero-padded [co_p, 9, ci_p] (tap-major) for _conv_k_order."""
    co, ci = w.shape[:2]
    out = torch.zeros(co_p, 9, ci_p, dtype=w.dtype, device=w.device)
    out[:co, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
    return out


class ScalerConvChainFn(Function):
    """cat[x1, x2, x3] with x_i = act(dropout(conv3x3(x_{i-1}, w_i))), act = ReLU or SiLU, channels-last, as three implicit
    GEMMs that write straight into ONE buffer [B, H, W, 3 CP] (CP = the widest output rounded up to 16: each convolution
    owns a 16-byte aligned CP-column segment, its padding columns come out of the product as exact zeros because the padded
    filter rows are zero, and the next convolution reads its input segment in place -- no cat, no copies, no layout change).
    Reference: Interp2dEncoder.forward, layers.py:497-507 (conv1 -> conv2 -> conv3 -> torch.cat) with Conv2dResBlock
    (layers.py:88-150: conv -> dropout -> activation).  Dropout and activation sit in the epilogue of the product.
    ReLU commutes with the non-negative dropout scale and the backward needs only the activated outputs: y > 0 <=> kept
    and pre-activation > 0.  SiLU (Interp2dEncoder's default, ex3's down-scaler; round 6) does not: the epilogue applies
    the dropout in FRONT of it (GT_ACT_DROP_SILU) and leaves the factor of the backward, keepscale * silu'(u), in a second
    buffer `fac` of the same layout (returned, non-differentiable: bilinear_resize_seg(in_factor=fac) multiplies the
    gradient with it on the way out).  Backward: per convolution (last to first) the data gradient is an implicit GEMM on
    the tap-reversed filter whose epilogue multiplies by the previous segment's mask / factor and adds that segment's own
    masked gradient (into a buffer of the backward: the incoming gradient is read-only), and the weight gradient runs next
    to it on the side stream."""

    @staticmethod
    def forward(ctx, x0, w1, w2, w3, p_drop: float, grad_masked: bool = False, act: int = H.ACT_RELU):
        H.need_f32_cuda(x0, w1, w2, w3)
        B, Hh, Ww, C0 = x0.shape
        ws = (w1, w2, w3)
        CP = (max(w.shape[0] for w in ws) + 15) // 16 * 16
        T = B * Hh * Ww
        dev = x0.device
        x0c = _c(x0)
        silu = act == H.ACT_SILU
        cat = torch.empty(T, 3 * CP, dtype=torch.float32, device=dev)
        fac = torch.empty(T, 3 * CP, dtype=torch.float32, device=dev) if silu else None
        salt = _next_salt(3)
        cin = (C0, CP, CP)
        ctx.prec = H.get_precision()             # bf16x3 or f16x2 (scaler_chain_ok); the backward runs in the same arithmetic
        for i, w in enumerate(ws):
            wf = _gathered(w, ("chain_fwd", CP, cin[i]), lambda t, ci_=cin[i]: _conv_k_order(_pad_filter(t, CP, ci_)))   # [CP, 9 cin] in k order
            A = x0c.reshape(T, C0) if i == 0 else cat[:, (i - 1) * CP:i * CP]
            H.gemm(A, wf, cat[:, i * CP:(i + 1) * CP], T, CP, 9 * cin[i], lda=(C0 if i == 0 else 3 * CP), ldb=9 * cin[i],
                   ldc=3 * CP, conv=(Hh, Ww, cin[i]), act=H.ACT_DROP_SILU if silu else H.ACT_RELU,
                   pre=fac[:, i * CP:(i + 1) * CP] if silu else None, ldpre=3 * CP,
                   drop=H.dropout_desc(p_drop, salt + i, dev) if p_drop > 0 else None, precision=ctx.prec)
        if _scaler_mask_sink[0] is not None:
            # ReLU: kept and positive.  SiLU: kept (the factor keepscale * silu'(u) vanishes only where the dropout struck)
            c4 = (fac if silu else cat).view(B, Hh, Ww, 3 * CP)
            _scaler_mask_sink[0].append([(c4[..., i * CP:i * CP + w.shape[0]] != 0) if silu else
                                         (c4[..., i * CP:i * CP + w.shape[0]] > 0) for i, w in enumerate(ws)])
        ctx.save_for_backward(x0c, w1, w2, w3, cat, fac)
        ctx.cfg = (p_drop, CP, grad_masked, silu)
        out = cat.view(B, Hh, Ww, 3 * CP)
        if not silu:
            return out
        fac4 = fac.view(B, Hh, Ww, 3 * CP)
        ctx.mark_non_differentiable(fac4)
        ctx.set_materialize_grads(False)         # (no zero gradient the size of `fac` from autograd)
        return out, fac4

    @staticmethod
    def backward(ctx, g, _gfac=None):
        if g is None:
            return (None,) * 7
        x0c, w1, w2, w3, cat, fac = ctx.saved_tensors
        p_drop, CP, grad_masked, silu = ctx.cfg
        prec = ctx.prec
        B, Hh, Ww, C0 = x0c.shape
        T = B * Hh * Ww
        dev = g.device
        ws = (w1, w2, w3)
        cin = (C0, CP, CP)
        # ReLU: the dropout scale rides on the products' alpha; SiLU: it is part of `fac`
        scale = 1.0 if silu else 1.0 / (1.0 - p_drop)
        # masked gradient of the three activated outputs in one pass
        # (grad_masked: the consumer -- bilinear_resize_seg(relu_input=True / in_factor=fac) -- has already applied it).
        # The incoming gradient is never written: the completed gradients of segments 0 and 1 (their own masked gradient
        # + the data gradient of the convolution behind them) go to a buffer of this function, so a hook / retain_grad on
        # the chain's output, or a second consumer summed by autograd, sees and produces what it should.
        g2 = _c(g).reshape(T, 3 * CP)
        gsrc = g2 if grad_masked else (g2 * fac if silu else H.act_bwd(g2, cat, H.ACT_RELU))
        acc = torch.empty(T, 2 * CP, dtype=torch.float32, device=dev)
        dws = [None, None, None]
        dx0 = None
        for i in (2, 1, 0):
            seg, ldseg = (gsrc[:, 2 * CP:], 3 * CP) if i == 2 else (acc[:, i * CP:(i + 1) * CP], 2 * CP)
            xin, ldx = (x0c.reshape(T, C0), C0) if i == 0 else (cat[:, (i - 1) * CP:i * CP], 3 * CP)
            if ctx.needs_input_grad[1 + i]:
                with H.side_branch(dev, T):
                    dws[i] = _scaler_wgrad(seg, ldseg, xin, ldx, ws[i], B, Hh, Ww, CP, cin[i], scale, prec)
            # dx[pix][ci] = scale * sum_tap sum_co dpre[pix - shift(tap)][co] W[co][ci][tap]: tap' = 8 - tap
            if i > 0 or ctx.needs_input_grad[0]:
                wd = _gathered(ws[i], ("chain_dgrad", CP, cin[i]),
                               lambda t, ci_=cin[i]: _conv_k_order(_pad_filter(t.flip(2, 3).transpose(0, 1), ci_, CP)))   # [cin, 9 CP] in k order
                if i == 0:
                    dx0 = torch.empty(T, C0, dtype=torch.float32, device=dev)
                    H.gemm(seg, wd, dx0, T, C0, 9 * CP, lda=ldseg, ldb=9 * CP, ldc=C0, conv=(Hh, Ww, CP), alpha=scale,
                           precision=prec)
                else:       # + the segment's own masked gradient (res), through its ReLU / dropout mask or its SiLU factor (aux)
                    aux = (fac if silu else cat)[:, (i - 1) * CP:i * CP]
                    H.gemm(seg, wd, acc[:, (i - 1) * CP:i * CP], T, CP, 9 * CP, lda=ldseg, ldb=9 * CP, ldc=2 * CP,
                           conv=(Hh, Ww, CP), alpha=scale, aux_op=H.AUX_MUL if silu else H.AUX_GT0, aux=aux, ldaux=3 * CP,
                           res=gsrc[:, (i - 1) * CP:i * CP], ldr=3 * CP, precision=prec)
            H.join_side(dev)        # the next weight gradient reads the segment this data gradient has just completed
        return (None if dx0 is None else dx0.view(B, Hh, Ww, C0)), dws[0], dws[1], dws[2], None, None, None


_scaler_wgrad_hip = [os.environ.get("GT_SCALER_WGRAD", "hip") != "miopen"]     # A/B switch (tools / tests)


def _scaler_wgrad(dseg, ldg, xin, ldx, w, B, Hh, Ww, CP, cin, scale, prec=None):
    """dW[co][ci][tap] = scale * sum_pix dseg[pix][co] xin[pix + shift(tap)][ci] for one narrow convolution: both operands are
    activations (column segments read in place through their row pitches), one side is <= 48 wide, nine taps share them:
    gt_conv3x3_wgrad_nhwc (gt_convw.hip) -- operands split once per block into LDS planes, all nine taps co-resident.
    Shapes outside that kernel (or GT_SCALER_WGRAD=miopen, the round-3 path) use the library's channels-last fp32
    weight-gradient kernel on dense copies of the two segments."""
    co, ci = w.shape[0], w.shape[1]
    if _scaler_wgrad_hip[0]:
        try:
            return H.conv3x3_wgrad_nhwc(dseg, ldg, xin, ldx, B, Hh, Ww, cin, CP, alpha=scale, precision=prec)[:co, :ci].contiguous()
        except H.GtNotSupported:
            pass
    gd = dseg.contiguous().view(B, Hh, Ww, CP).permute(0, 3, 1, 2)                      # dense channels-last
    xd = xin.contiguous().view(B, Hh, Ww, cin).permute(0, 3, 1, 2)
    wp = torch.empty(CP, cin, 3, 3, dtype=torch.float32, device=w.device).contiguous(memory_format=torch.channels_last)
    dw = torch.ops.aten.convolution_backward(gd, xd, wp, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                             [False, True, False])[1]
    return (dw[:co, :ci] * scale).contiguous()


def scaler_conv_chain(x0, w1, w2, w3, p_drop: float = 0.0, training: bool = True, grad_masked: bool = False,
                      act: str = "relu"):
    """x0 (B, H, W, C0) channels-last -> (B, H, W, 3 CP): see ScalerConvChainFn; column segment i holds x_{i+1} in its
    first w_i.shape[0] columns, zeros behind them.  grad_masked: the ONLY consumer of the result is
    bilinear_resize_seg(relu_input=True) (act='silu': in_factor=fac), whose backward hands the gradient over already
    multiplied by [result > 0] (by fac) -- the masking pass of the backward is then skipped; the gradient itself is only
    read.  act='silu' returns (buffer, fac)."""
    if act not in ("relu", "silu"):
        raise NotImplementedError(f"scaler_conv_chain: activation {act!r}")
    return ScalerConvChainFn.apply(x0, w1, w2, w3, float(p_drop) if training else 0.0, bool(grad_masked), H.ACT_CODE[act])


# ----------------------------------------------------------------------------------- Linear
class LinearFn(Function):
    """y = res + out_scale * dropout(act(x W^T + b + extra W_e^T)).

    ``extra`` (optional, [.., p] with small p) folds ``torch.cat([x, extra], -1)`` followed by a Linear
    over the concatenation into one GEMM + rank-p epilogue (model.py:615-617, 507-512); W then has
    in_features = K + p.  Replaces nn.Linear (+ activation + nn.Dropout) at layers.py:964-987,
    model.py:615-629.  Returns y (and keeps the pre-activation for SiLU backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, extra, act: int, p_drop: float):
        H.need_f32_cuda(x, weight, bias, extra)
        K = x.shape[-1]
        N = weight.shape[0]
        pe = 0 if extra is None else extra.shape[-1]
        assert weight.shape[1] == K + pe
        x2 = _c(x).reshape(-1, K)
        T = x2.shape[0]
        w = _c(weight)
        y = torch.empty(T, N, dtype=torch.float32, device=x.device)
        pre = torch.empty(T, N, dtype=torch.float32, device=x.device) if act == H.ACT_SILU else None
        e2 = None if extra is None else _c(extra).reshape(T, pe)
        salt = _next_salt()
        drop = H.dropout_desc(p_drop, salt, x.device) if p_drop > 0 else None
        H.gemm(x2, w, y, T, N, K, lda=K, ldb=K + pe, ldc=N, bias=bias, act=act,
               rp=pe, rp_a=e2, rp_lda=pe, rp_b=(w[:, K:] if pe else None), rp_ldb=K + pe,
               pre=pre, ldpre=N, drop=drop)
        ctx.save_for_backward(x2, w, e2, pre if act == H.ACT_SILU else (y if act == H.ACT_RELU else None))
        ctx.cfg = (act, p_drop, salt, K, N, pe, bias is not None, x.shape)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x2, w, e2, saved = ctx.saved_tensors
        act, p_drop, salt, K, N, pe, has_bias, xshape = ctx.cfg
        dev = gy.device
        T = x2.shape[0]
        g = _c(gy).reshape(T, N)
        # dpre = g * dropmask * act'(pre)
        if act == H.ACT_NONE and p_drop == 0:
            gp = g
        else:
            gp = torch.empty_like(g)
            if act == H.ACT_SILU:
                gp = H.act_bwd(g, saved, H.ACT_SILU)
                if p_drop > 0:
                    gp = H.dropout_apply(gp, H.dropout_desc(p_drop, salt, dev))
            elif act == H.ACT_RELU:
                # y = relu(pre)*mask*s : y>0 <=> kept and pre>0
                gp = H.act_bwd(g, saved, H.ACT_RELU)
                if p_drop > 0:
                    gp = gp * (1.0 / (1.0 - p_drop))
            else:
                gp = H.dropout_apply(g, H.dropout_desc(p_drop, salt, dev))
        dx = de = dw = db = None
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:     # weight gradient on the side stream, next to the data gradient (H.side_branch)
            dw = torch.empty(N, K + pe, dtype=torch.float32, device=dev)
            if want_db:         # the bias gradient rides on the weight-gradient GEMM (row sums of its A)
                db = torch.empty(N, dtype=torch.float32, device=dev)
            with H.side_branch(dev, T):
                H.gemm(gp, x2, dw, N, K, T, layout_a=1, layout_b=1, lda=N, ldb=K, ldc=K + pe, split_k=0,
                       a_colsum=db)
                if pe:
                    H.gemm(gp, e2, dw[:, K:], N, pe, T, layout_a=1, layout_b=1, lda=N, ldb=pe, ldc=K + pe,
                           split_k=0)
        elif want_db:
            db = H.colsum(gp, T, N, N)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(T, K, dtype=torch.float32, device=dev)
            H.gemm(gp, w, dx, T, K, N, layout_b=1, lda=N, ldb=K + pe, ldc=K)
            dx = dx.reshape(xshape)
        H.join_side(dev)
        return dx, dw, db, de, None, None


def linear(x, weight, bias=None, extra=None, act: str = None, p_drop: float = 0.0):
    if extra is not None and extra.requires_grad:
        raise NotImplementedError("ops.linear: `extra` (the concatenated coordinates) gets no gradient")
    return LinearFn.apply(x, weight, bias, extra, H.ACT_CODE[act], float(p_drop))


# ----------------------------------------------------------------------------------- SiLU gates on the producer of a gradient
# Inside a `silu_gate_scope` (SpectralRegressor's layer loop: every intermediate has exactly ONE consumer there) a Function
# whose result is silu(pre) OFFERS its pre-activation under the result's address; the Function that consumes the result
# TAKES it, and its backward multiplies the gradient it forms by silu'(pre) on the store of the kernel that forms it
# (gt_dft_synthesis_gated / gt_mlp_head_bwd_gated).  The producer's backward then receives the gradient of its
# PRE-activation (ctx.g_gated) and skips its own gt_act_bwd pass: 2 x 172 us per step at the headline shape.
_gate_fold = [os.environ.get("GT_FOLD_GATES", "1") != "0"]       # A/B switch (tools / tests)
_gate_depth = [0]
_silu_gates = {}                 # data_ptr of an activated result -> (ctx of its Function, pre-activation)


class silu_gate_scope:
    def __init__(self, enabled: bool = True):
        self.on = bool(enabled) and _gate_fold[0]

    def __enter__(self):
        if self.on:
            _gate_depth[0] += 1
        return self

    def __exit__(self, *exc):
        if self.on:
            _gate_depth[0] -= 1
            if _gate_depth[0] == 0:
                _silu_gates.clear()          # offers nobody took: their producers run gt_act_bwd as ever
        return False


def _offer_gate(ctx, out, pre):
    ctx.g_gated = False
    if _gate_depth[0] > 0 and pre is not None:
        _silu_gates[out.data_ptr()] = (ctx, pre)


def _take_gate(x):
    """The pre-activation whose SiLU produced ``x`` (the caller's backward MUST multiply d(x) by silu' of it), or None."""
    if _gate_depth[0] <= 0:
        return None
    ent = _silu_gates.pop(x.data_ptr(), None)
    if ent is None or ent[1].numel() != x.numel():
        return None
    ent[0].g_gated = True
    return ent[1]


class MlpHeadFn(Function):
    """y = W2 act(W1 x + b1) + b2 with a narrow output (n_out <= 4) and hidden width <= 128: the tail of
    SpectralRegressor / PointwiseRegressor (model.py:575-580, 625-629).  The [T, hidden] activation never
    reaches HBM in forward (row-dot epilogue); backward recomputes the pre-activation inside the GEMM that
    produces dL/dh and gets dW2 as a by-product of the same launch."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act: int):
        H.need_f32_cuda(x, w1, b1, w2, b2)
        K, N, no = x.shape[-1], w1.shape[0], w2.shape[0]
        x2 = _c(x).reshape(-1, K)
        T = x2.shape[0]
        w1c, w2c = _c(w1), _c(w2)
        out = torch.empty(T, no, dtype=torch.float32, device=x.device)
        ctx.prec = H.get_precision()              # the backward runs in the arithmetic of the forward
        if H.mlp_head_supported(K, N, no):       # dedicated one-pass kernel (gt_mlp_head_fwd)
            H.mlp_head_fwd(x2, w1c, b1, w2c, b2, act, out, precision=ctx.prec)
        else:
            H.gemm(x2, w1c, None, T, N, K, lda=K, ldb=K, ldc=N, bias=b1, act=act, ep_mode=H.EP_ROWDOT, w2=w2c,
                   b2=b2, out2=out)
        gate = _take_gate(x2) if ctx.needs_input_grad[0] else None
        ctx.save_for_backward(x2, w1c, b1, w2c, gate)
        ctx.cfg = (act, K, N, no, b1 is not None, b2 is not None, x.shape)
        return out.reshape(*x.shape[:-1], no)

    @staticmethod
    def backward(ctx, gy):
        x2, w1c, b1, w2c, gate = ctx.saved_tensors
        act, K, N, no, hb1, hb2, xshape = ctx.cfg
        dev, T = gy.device, x2.shape[0]
        g = _c(gy).reshape(T, no)
        f32 = dict(dtype=torch.float32, device=dev)
        if H.mlp_head_supported(K, N, no):       # everything in one pass over x (gt_mlp_head_bwd)
            dx = torch.empty(T, K, **f32) if ctx.needs_input_grad[0] else None
            dw1, dw2 = torch.empty(N, K, **f32), torch.empty(no, N, **f32)
            db1 = torch.empty(N, **f32) if hb1 else None
            db2 = torch.empty(no, **f32) if hb2 else None
            H.mlp_head_bwd(x2, w1c, b1, w2c, act, g, dx, dw1, db1, dw2, db2, precision=ctx.prec,
                           dx_gate=None if gate is None else gate.reshape(T, K))
            return (dx.reshape(xshape) if dx is not None else None), dw1, db1, dw2, db2, None
        dh, dw2 = torch.empty(T, N, **f32), torch.empty(no, N, **f32)
        H.gemm(x2, w1c, dh, T, N, K, lda=K, ldb=K, ldc=N, bias=b1, act=act, ep_mode=H.EP_MLP_BWD, w2=w2c, g2=g,
               dw2=dw2)
        db2 = H.colsum(g, T, no, no) if hb2 else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(T, K, **f32)
            H.gemm(dh, w1c, dx, T, K, N, layout_b=1, lda=N, ldb=K, ldc=K)
            if gate is not None:
                dx = H.act_bwd(dx, gate.reshape(T, K), H.ACT_SILU)
            dx = dx.reshape(xshape)
        dw1 = torch.empty(N, K, **f32)
        db1 = torch.empty(N, **f32) if hb1 else None
        H.gemm(dh, x2, dw1, N, K, T, layout_a=1, layout_b=1, lda=N, ldb=K, ldc=K, split_k=0, a_colsum=db1)
        return dx, dw1, db1, dw2, db2, None


def mlp_head(x, w1, b1, w2, b2, act: str = "silu"):
    """Fused two-layer pointwise head when it fits the kernel (hidden <= 128, n_out <= 4), else two linears."""
    if w1.shape[0] <= 128 and w2.shape[0] <= 4 and w2.shape[1] == w1.shape[0]:
        return MlpHeadFn.apply(x, w1, b1, w2, b2, H.ACT_CODE[act])
    return linear(linear(x, w1, b1, act=act), w2, b2)


# ----------------------------------------------------------------------------------- row LayerNorm
class LayerNormFn(Function):
    """nn.LayerNorm(d_model) of the encoder layer (model.py:84-85, 128-135)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps: float):
        xc = _c(x)
        y, stats = H.layernorm_fwd(xc, _c(weight), _c(bias), eps)
        ctx.save_for_backward(xc, weight, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, weight, stats = ctx.saved_tensors
        dx, dg, db = H.layernorm_bwd(_c(gy), xc, _c(weight), stats)
        return dx, dg, db, None


def layer_norm(x, weight, bias, eps):
    return LayerNormFn.apply(x, weight, bias, float(eps))


def _check_res_is_x(res, x, what: str):
    """The fused backward passes return d(res) folded into d(x): only valid when the residual input IS x."""
    if res is not None and res is not x and not (res.data_ptr() == x.data_ptr() and res.shape == x.shape
                                                 and res.stride() == x.stride()):
        raise ValueError(f"ops.{what}: `res` must be the input tensor itself (or None)")


# ----------------------------------------------------------------------------------- FFN
class FeedForwardFn(Function):
    """out = res + dropout2(lr2(dropout_h(act(lr1(x)))))   (layers.py:979-987 + model.py:131-132).

    res=None gives the bare FeedForward.forward."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res, act: int, p_h: float, p_out: float):
        H.need_f32_cuda(x, w1, b1, w2, b2, res)
        if act not in (H.ACT_RELU, H.ACT_SILU):
            raise NotImplementedError("FeedForward HIP path implements relu and silu")
        d, f, dout = x.shape[-1], w1.shape[0], w2.shape[0]
        xc = _c(x).reshape(-1, d)
        T = xc.shape[0]
        dev = x.device
        w1c, w2c = _c(w1), _c(w2)
        salt = _next_salt()
        hid = torch.empty(T, f, dtype=torch.float32, device=dev)
        pre = torch.empty(T, f, dtype=torch.float32, device=dev) if act == H.ACT_SILU else None
        out = torch.empty(T, dout, dtype=torch.float32, device=dev)
        rc = None if res is None else _c(res).reshape(T, dout)
        d_h = H.dropout_desc(p_h, salt, dev) if p_h > 0 else None
        d_o = H.dropout_desc(p_out, salt + 1, dev) if p_out > 0 else None
        bits = None
        if dout == d and H.ffn_fwd_supported(T, d, f, act):
            # both products in ONE launch, the hidden tile of 64 token rows kept in LDS between them (gt_ffn.hip): hid is
            # written for the backward (dW2) and never read back here; the ReLU / dropout decisions go along as one bit per
            # value for the fused data half of the backward
            bits = H.ffn_fwd(xc, w1c, b1, w2c, b2, rc, d_h, d_o, act, hid, out,
                             want_bits=bool(act == H.ACT_RELU and _ffn_bwd_fused[0] and any(ctx.needs_input_grad[:5])))
        else:
            H.gemm(xc, w1c, hid, T, f, d, lda=d, ldb=d, ldc=f, bias=b1, act=act, pre=pre, ldpre=f, drop=d_h, weight_b=True)
            H.gemm(hid, w2c, out, T, dout, f, lda=f, ldb=f, ldc=dout, bias=b2, drop=d_o, res=rc, ldr=dout, weight_b=True)
        if _relu_mask_sink[0] is not None and act == H.ACT_RELU and p_h == 0:
            _relu_mask_sink[0].append(hid > 0)
        ctx.save_for_backward(xc, w1c, w2c, hid, pre, bits)
        ctx.cfg = (act, p_h, p_out, salt, d, f, dout, b1 is not None, b2 is not None, res is not None,
                   x.shape)
        ctx.in_mask = _wanted_mask(xc)            # the producer of x wants d(x) under its own output mask too
        _hint_output_mask(out, p_out, salt + 1)
        return out.reshape(*x.shape[:-1], dout)

    @staticmethod
    def backward(ctx, gy):
        xc, w1c, w2c, hid, pre, bits = ctx.saved_tensors
        act, p_h, p_out, salt, d, f, dout, hb1, hb2, has_res, xshape = ctx.cfg
        dev = gy.device
        T = xc.shape[0]
        g = _c(gy).reshape(T, dout)
        # the masked gradient g*mask_out feeds three contractions: one elementwise pass is cheaper than
        # regenerating the mask in each GEMM's operand loader (measured: 257 -> 135 us on the gh GEMM at B=64)
        gm = g
        if p_out > 0:           # the masked copy the consumer of our output wrote next to d(out), else the elementwise pass
            gm = _take_twin(g, p_out, salt + 1)
            if gm is None:
                gm = H.dropout_apply(g, H.dropout_desc(p_out, salt + 1, dev))
        # gh = (gm W2) * mask_h * act'(pre)
        gh = torch.empty(T, f, dtype=torch.float32, device=dev)
        # bias gradients ride on the weight-gradient GEMMs (row sums of their A operand); the weight gradients run on
        # the side stream next to the data-gradient GEMMs (H.side_branch)
        dw2 = torch.empty(dout, f, dtype=torch.float32, device=dev)
        db2 = torch.empty(dout, dtype=torch.float32, device=dev) if hb2 else None
        with H.side_branch(dev, T):
            H.gemm(gm, hid, dw2, dout, f, T, layout_a=1, layout_b=1, lda=dout, ldb=f, ldc=f, split_k=0,
                   a_colsum=db2)
        dx = torch.empty(T, d, dtype=torch.float32, device=dev)
        same = has_res and dout == d
        dxm, want = None, ctx.in_mask
        if want is not None:
            dxm = torch.empty_like(dx)
        fused_bwd = bits is not None and act == H.ACT_RELU
        if fused_bwd:
            # gh = (gm W2) through the forward's decision bits, dx = g + gh W1 and its masked twin: ONE launch, the hidden
            # activation is not read (gt_ffn_bwd); dW1 = gh^T x follows on the side stream
            H.ffn_bwd(gm, w2c, w1c, bits, 1.0 / (1.0 - p_h), g if same else None, gh, dx, dxm,
                      H.dropout_desc(want[0], want[1], dev) if want else None)
        elif act == H.ACT_RELU:
            H.gemm(gm, w2c, gh, T, f, dout, layout_b=1, lda=dout, ldb=f, ldc=f,
                   aux_op=H.AUX_GT0, aux=hid, ldaux=f, aux_scale=1.0 / (1.0 - p_h), weight_b=True)
        else:
            H.gemm(gm, w2c, gh, T, f, dout, layout_b=1, lda=dout, ldb=f, ldc=f,
                   aux_op=H.AUX_DSILU, aux=pre, ldaux=f,
                   drop=H.dropout_desc(p_h, salt, dev) if p_h > 0 else None, weight_b=True)
        dw1 = torch.empty(f, d, dtype=torch.float32, device=dev)
        db1 = torch.empty(f, dtype=torch.float32, device=dev) if hb1 else None
        with H.side_branch(dev, T):
            H.gemm(gh, xc, dw1, f, d, T, layout_a=1, layout_b=1, lda=f, ldb=d, ldc=d, split_k=0, a_colsum=db1)
        if not fused_bwd:
            H.gemm(gh, w1c, dx, T, d, f, layout_b=1, lda=f, ldb=d, ldc=d, res=g if same else None, ldr=d, weight_b=True,
                   c_masked=dxm, ldc_masked=d, c_mask=H.dropout_desc(want[0], want[1], dev) if want else None)
        if dxm is not None:
            _offer_twin(dx, dxm, *want)
        H.join_side(dev)
        dx = dx.reshape(xshape)
        # the residual input is x itself: its gradient g is already folded into dx (res epilogue above), so the
        # `res` slot contributes nothing (None) -- no zero fill, no extra add in autograd
        dres = None if (not has_res or same) else gy
        return dx, dw1, db1, dw2, db2, dres, None, None, None


def feed_forward(x, w1, b1, w2, b2, res=None, act="relu", p_h=0.0, p_out=0.0):
    """res must be x itself (or None): the fused backward folds d(res) into d(x)."""
    _check_res_is_x(res, x, "feed_forward")
    return FeedForwardFn.apply(x, w1, b1, w2, b2, res, H.ACT_CODE[act], float(p_h), float(p_out))


# ----------------------------------------------------------------------------------- attention
class SimpleAttentionFn(Function):
    """out = res + sign * dropout1( fc( merge_heads( attention(Q', K', V') ) ) ).

    galerkin: per-head LN on K,V; M = mask .* (K'^T V')/n; heads: Q' M      (layers.py:708-734)
    fourier : per-head LN on Q,K; S = mask .* (Q' K'^T)/sqrt(d_k')/n; heads: S V' (layers.py:672-705)
    with X' = [pos, X] per head (layers.py:869-874) and fc over the merged heads (layers.py:894-897).
    Also returns the attention matrix (``attn_weight``), detached."""

    @staticmethod
    def forward(ctx, x, pos, wqkv, bqkv, gamma, beta, wfc, bfc, res, cfg, mask):
        (kind, h, norm_mask, eps, sign, p_attn, p_out, need_w) = cfg
        H.need_f32_cuda(x, pos, wqkv, bqkv, gamma, beta, wfc, bfc, res, mask)
        B, n, d = x.shape
        dk = d // h
        p = 0 if pos is None else pos.shape[-1]
        Dr, DP = dk + p, H.round4(dk + p)
        T = B * n
        dev = x.device
        xc = _c(x).reshape(T, d)
        posc = None if pos is None else _c(pos).reshape(T, p)
        wq, wf = _c(wqkv), _c(wfc)
        salt = _next_salt(4)
        qkv = out3 = stats = None
        # "plain" head tiles: when every consumer of K', V' is one of the fused Galerkin kernels, the tiles keep the
        # normalised values WITHOUT the LayerNorm affine (the consumers apply gamma / beta), the backward takes xh from the
        # tiles, and the raw projection has no reader left: it is neither written nor allocated (gt_hip.h: hn_plain)
        plain = (_plain_tiles[0] and kind == "galerkin" and _dkv_ln_fused[0] and H.galerkin_dkv_ln_supported(dk, p, norm_mask)
                 and H.galerkin_ktv_supported(dk, p))
        if _qkvnorm_fused[0] and dk in (16, 32, 48, 64) and bqkv is not None and H.get_precision() in H.SPLIT_EXACT:
            # head norm on the projection's epilogue (GT_EP_HEADNORM): one pass less over [T, 3d], one launch less
            out3 = torch.empty(3, T, h, DP, dtype=torch.float32, device=dev)
            stats = torch.empty(2, T, h, 2, dtype=torch.float32, device=dev)
            if not plain:
                qkv = torch.empty(T, 3 * d, dtype=torch.float32, device=dev)
            try:
                # the raw projection is kept for the LayerNorm backward only: the normalised streams' blocks of qkv
                H.gemm(xc, wq, qkv, T, 3 * d, d, lda=d, ldb=d, ldc=3 * d, bias=bqkv, weight_b=True,
                       hn=dict(gamma=gamma, beta=beta, pos=posc, out=out3, stats=stats, h=h, dk=dk, p=p,
                               norm_mask=norm_mask, eps=eps, skip_raw=7 if plain else (~norm_mask) & 7, plain=plain))
            except H.GtNotSupported:                          # shapes / alignment the fused kernel does not take
                out3 = None
        if out3 is None:
            plain = False
            qkv = torch.empty(T, 3 * d, dtype=torch.float32, device=dev)
            H.gemm(xc, wq, qkv, T, 3 * d, d, lda=d, ldb=d, ldc=3 * d, bias=bqkv, weight_b=True)
            out3, stats = H.headnorm_fwd(qkv, posc, gamma, beta, T, h, dk, p, norm_mask, eps)
        Qp, Kp, Vp = out3[0], out3[1], out3[2]
        hD = h * DP
        out = torch.empty(T, d, dtype=torch.float32, device=dev)
        rc = None if res is None else _c(res).reshape(T, d)
        d_attn = H.dropout_desc(p_attn, salt, dev) if (p_attn > 0 and mask is None) else None
        d_out = H.dropout_desc(p_out, salt + 1, dev) if p_out > 0 else None
        if kind == "galerkin":
            slabs = H.galerkin_ktv(Kp, Vp, B, n, h, dk, p, gamma=gamma if plain else None,
                                   beta=beta if plain else None)   # streaming MFMA kernel, token chunks
            if slabs is None:                                       # head sizes it does not cover
                slabs = torch.empty(1, B, h, DP, DP, dtype=torch.float32, device=dev)
                H.gemm(Kp, Vp, slabs, DP, DP, n, layout_a=1, layout_b=1, lda=hD, ldb=hD, ldc=DP, batch=(B, h),
                       a_bs=(n * hD, DP), b_bs=(n * hD, DP), c_bs=(h * DP * DP, DP * DP), split_k=0)
            Mt, P, Pv = H.galerkin_finalize_fwd(slabs, slabs.shape[0], B * h * DP * DP, B, h, DP, Dr, d, n, mask,
                                                d_attn, wf, value_rows_of=p)
            H.gemm(Qp, P, out, n, d, hD, layout_b=1, lda=hD, ldb=d, ldc=d, batch=(B, 1), a_bs=(n * hD, 0),
                   b_bs=(hD * d, 0), c_bs=(n * d, 0), bias=bfc, drop=d_out, res=rc, ldr=d, r_bs=(n * d, 0),
                   out_scale=sign)
            ctx.save_for_backward(xc, wq, gamma, wf, qkv, stats, out3, Mt, P, mask, beta if plain else None, Pv)
            ctx.plain = plain
            attn_w = Mt[:, :, :Dr, :Dr]
        else:
            scale = 1.0 / math.sqrt(Dr) / n
            wpad = torch.zeros(d, h, DP, dtype=torch.float32, device=dev)
            wpad[:, :, :Dr] = wf.reshape(d, h, Dr)
            wpad = wpad.reshape(d, hD)
            flash = (not need_w) and DP in H.FOURIER_DP
            # fp16 arithmetic: the p = 0.5 score mask is drawn per 4 x 4 block (one hash per block: gt_hip.h), by the fused
            # kernels and by the materialising path alike; decided once per forward, the backward follows it
            blk16 = bool(H.fourier16_active() and d_attn is not None and H.fourier16_block_mask(d_attn))
            ctx.f16_block = blk16
            if flash:
                # fused (Q'K'^T * scale .* mask) V': the n x n matrix never reaches HBM
                if H.fourier16_active():
                    # two-term fp16 kernels: the head tiles are split once into fragment-ordered images, kept for the backward
                    imgs = H.fourier16_presplit((Qp, Kp, Vp), B, n, h, DP)
                    att = H.fourier16_attn(imgs[0], None, imgs[1], imgs[2], B, n, h, DP, scale, mask, d_attn,
                                           False, block16=blk16).reshape(T, hD)
                    ctx.f16_imgs = imgs
                else:
                    att = H.fourier_attn(Qp, None, Kp, Vp, B, n, h, DP, scale, mask, d_attn, False).reshape(T, hD)
                    ctx.f16_imgs = None
                S = None
            else:
                S = torch.empty(B, h, n, n, dtype=torch.float32, device=dev)
                H.gemm(Qp, Kp, S, n, n, DP, lda=hD, ldb=hD, ldc=n, batch=(B, h), a_bs=(n * hD, DP),
                       b_bs=(n * hD, DP), c_bs=(h * n * n, n * n), alpha=scale, drop=None if blk16 else d_attn,
                       aux_op=H.AUX_MUL if mask is not None else H.AUX_NONE, aux=mask, ldaux=n,
                       aux_bs=(h * n * n, n * n))
                if blk16:
                    H.dropout_block16(S, B * h, n, d_attn)
                att = torch.empty(T, hD, dtype=torch.float32, device=dev)
                H.gemm(S, Vp, att, n, DP, n, layout_b=1, lda=n, ldb=hD, ldc=hD, batch=(B, h),
                       a_bs=(h * n * n, n * n), b_bs=(n * hD, DP), c_bs=(n * hD, DP))
            H.gemm(att, wpad, out, T, d, hD, lda=hD, ldb=hD, ldc=d, bias=bfc, drop=d_out, res=rc, ldr=d,
                   out_scale=sign)
            ctx.save_for_backward(xc, wq, gamma, wpad, qkv, stats, out3, S, att, mask)
            attn_w = S if S is not None else torch.empty(0, device=dev)
        ctx.cfg = cfg
        ctx.in_mask = _wanted_mask(xc)
        _hint_output_mask(out, p_out, salt + 1)
        ctx.dims = (B, n, d, h, dk, p, Dr, DP, salt, bqkv is not None, bfc is not None, res is not None,
                    x.shape)
        attn_w = attn_w.detach()
        ctx.mark_non_differentiable(attn_w)
        return out.reshape(x.shape), attn_w

    @staticmethod
    def backward(ctx, gy, _gw):
        (kind, h, norm_mask, eps, sign, p_attn, p_out, need_w) = ctx.cfg
        B, n, d, h, dk, p, Dr, DP, salt, hbq, hbf, has_res, xshape = ctx.dims
        T, hD = B * n, h * DP
        dev = gy.device
        g = _c(gy).reshape(T, d)
        g_in = g                                             # unmasked: what flows to the residual branch
        if p_out > 0:                                        # mask once, not in every consumer's operand loader
            gmk = _take_twin(g, p_out, salt + 1)             # ... and not at all when the consumer of `out` wrote the copy
            g = gmk if gmk is not None else H.dropout_apply(g, H.dropout_desc(p_out, salt + 1, dev))
        d_out = None
        fused_ln = False
        dO3 = None
        dbfc = None
        if kind == "galerkin":
            dbfc = torch.empty(d, dtype=torch.float32, device=dev) if hbf else None
            xc, wq, gamma, wf, qkv, stats, out3, Mt, P, mask, beta_plain, Pv = ctx.saved_tensors
            d_attn = H.dropout_desc(p_attn, salt, dev) if (p_attn > 0 and mask is None) else None
            Qp, Kp, Vp = out3[0], out3[1], out3[2]
            # dP^T[b] = (sign*g*mask1)^T[b] Q'[b]        [B, d, h*DP]
            dPt = torch.empty(B, d, hD, dtype=torch.float32, device=dev)
            with H.side_branch(dev, T): # the token-contracted product next to the token-row product below
                H.gemm(g, Qp, dPt, d, hD, n, layout_a=1, layout_b=1, lda=d, ldb=hD, ldc=hD, batch=(B, 1),
                       a_bs=(n * d, 0), b_bs=(n * hD, 0), c_bs=(d * hD, 0), split_k=0, a_drop=d_out,
                       a_drop_sign=sign, a_drop_ld=d, a_drop_bstride=n * d, alpha=(sign if d_out is None else 1.0),
                       a_colsum=dbfc)       # + d(fc bias) = column sums of the masked, signed g
            # dQ'[b] = (sign*g*mask1)[b] P[b]^T
            fused_ln = ctx.plain or (_dkv_ln_fused[0] and H.galerkin_dkv_ln_supported(dk, p, norm_mask))
            if fused_ln:
                # only the value columns of dQ' reach d_qkv (the coordinates take no gradient): contract with those rows
                # of P and write the Q block of d_qkv directly -- one 128-wide tile column instead of h*DP = 144, no
                # dQ' round trip
                dqkv = torch.empty(T, 3 * d, dtype=torch.float32, device=dev)
                H.gemm(g, Pv, dqkv, n, d, d, lda=d, ldb=d, ldc=3 * d, batch=(B, 1), a_bs=(n * d, 0),
                       b_bs=(d * d, 0), c_bs=(n * 3 * d, 0), a_drop=d_out, a_drop_sign=sign, a_drop_ld=d,
                       a_drop_bstride=n * d, alpha=(sign if d_out is None else 1.0))
            else:
                dO3 = torch.empty(3, T, h, DP, dtype=torch.float32, device=dev)
                H.gemm(g, P, dO3[0], n, hD, d, lda=d, ldb=d, ldc=hD, batch=(B, 1), a_bs=(n * d, 0),
                       b_bs=(hD * d, 0), c_bs=(n * hD, 0), a_drop=d_out, a_drop_sign=sign, a_drop_ld=d,
                       a_drop_bstride=n * d, alpha=(sign if d_out is None else 1.0))
            H.join_side(dev)
            dM, dWs = H.galerkin_finalize_bwd(dPt, Mt, mask, d_attn, wf, B, h, DP, Dr, d, n)
            dwfc = torch.empty(d, h * Dr, dtype=torch.float32, device=dev)
            H.slab_reduce(dWs, B, d * h * Dr, d * h * Dr, dwfc)
            # dK' = V' dM^T ; dV' = K' dM          per (b, head)
            if fused_ln:       # ... with the head LayerNorm backward behind them: dK', dV' stay in registers
                dqkv, dgamma, dbeta = H.galerkin_dkv_ln(Kp, Vp, dM, None, qkv, gamma, stats, B, n, h, dk, p, d_qkv=dqkv,
                                                        beta=beta_plain)
            elif DP in H.FOURIER_DP:                           # one streaming pass (gt_galerkin_dkv)
                H.galerkin_dkv(Kp, Vp, dM, dO3[1], dO3[2], B, n, h, DP)
            else:
                H.gemm(Vp, dM, dO3[1], n, DP, DP, lda=hD, ldb=DP, ldc=hD, batch=(B, h), a_bs=(n * hD, DP),
                       b_bs=(h * DP * DP, DP * DP), c_bs=(n * hD, DP))
                H.gemm(Kp, dM, dO3[2], n, DP, DP, layout_b=1, lda=hD, ldb=DP, ldc=hD, batch=(B, h),
                       a_bs=(n * hD, DP), b_bs=(h * DP * DP, DP * DP), c_bs=(n * hD, DP))
        else:
            xc, wq, gamma, wpad, qkv, stats, out3, S, att, mask = ctx.saved_tensors
            dO3 = torch.empty(3, T, h, DP, dtype=torch.float32, device=dev)
            d_attn = H.dropout_desc(p_attn, salt, dev) if (p_attn > 0 and mask is None) else None
            Qp, Kp, Vp = out3[0], out3[1], out3[2]
            scale = 1.0 / math.sqrt(Dr) / n
            asc = sign if d_out is None else 1.0
            dwpad = torch.empty(d, hD, dtype=torch.float32, device=dev)
            dbfc = torch.empty(d, dtype=torch.float32, device=dev) if hbf else None
            H.gemm(g, att, dwpad, d, hD, T, layout_a=1, layout_b=1, lda=d, ldb=hD, ldc=hD, split_k=0,
                   a_drop=d_out, a_drop_sign=sign, a_drop_ld=d, alpha=asc, a_colsum=dbfc)
            dwfc = dwpad.reshape(d, h, DP)[:, :, :Dr].reshape(d, h * Dr)
            datt = torch.empty(T, hD, dtype=torch.float32, device=dev)
            H.gemm(g, wpad, datt, T, hD, d, layout_b=1, lda=d, ldb=hD, ldc=hD, a_drop=d_out,
                   a_drop_sign=sign, a_drop_ld=d, alpha=asc)
            if S is None:
                # fused passes: dQ' = (dO V'^T .* m) K' ;  dV' = (S .* m)^T dO, dK' = (dO V'^T .* m)^T Q'
                datt3 = datt.reshape(T, h, DP)
                imgs = getattr(ctx, "f16_imgs", None)
                if imgs is not None:
                    iq, ik, iv = imgs
                    (ido,) = H.fourier16_presplit((datt3,), B, n, h, DP)
                    H.fourier16_attn(ido, None, iv, ik, B, n, h, DP, scale, mask, d_attn, False, O1=dO3[0],
                                     block16=ctx.f16_block)
                    H.fourier16_attn(ik, iv, iq, ido, B, n, h, DP, scale, mask, d_attn, True, O1=dO3[2], O2=dO3[1],
                                     block16=ctx.f16_block)
                else:
                    H.fourier_attn(datt3, None, Vp, Kp, B, n, h, DP, scale, mask, d_attn, False, O1=dO3[0])
                    H.fourier_attn(Kp, Vp, Qp, datt3, B, n, h, DP, scale, mask, d_attn, True, O1=dO3[2], O2=dO3[1])
            else:
                dS = torch.empty(B, h, n, n, dtype=torch.float32, device=dev)
                blk16 = getattr(ctx, "f16_block", False)
                H.gemm(datt, Vp, dS, n, n, DP, lda=hD, ldb=hD, ldc=n, batch=(B, h), a_bs=(n * hD, DP),
                       b_bs=(n * hD, DP), c_bs=(h * n * n, n * n), alpha=scale, drop=None if blk16 else d_attn,
                       aux_op=H.AUX_MUL if mask is not None else H.AUX_NONE, aux=mask, ldaux=n,
                       aux_bs=(h * n * n, n * n))
                if blk16:
                    H.dropout_block16(dS, B * h, n, d_attn)
                # dV' = S^T datt ; dQ' = dS K' ; dK' = dS^T Q'
                H.gemm(S, datt, dO3[2], n, DP, n, layout_a=1, layout_b=1, lda=n, ldb=hD, ldc=hD, batch=(B, h),
                       a_bs=(h * n * n, n * n), b_bs=(n * hD, DP), c_bs=(n * hD, DP))
                H.gemm(dS, Kp, dO3[0], n, DP, n, layout_b=1, lda=n, ldb=hD, ldc=hD, batch=(B, h),
                       a_bs=(h * n * n, n * n), b_bs=(n * hD, DP), c_bs=(n * hD, DP))
                H.gemm(dS, Qp, dO3[1], n, DP, n, layout_a=1, layout_b=1, lda=n, ldb=hD, ldc=hD, batch=(B, h),
                       a_bs=(h * n * n, n * n), b_bs=(n * hD, DP), c_bs=(n * hD, DP))
        if not (kind == "galerkin" and fused_ln):
            dqkv, dgamma, dbeta = H.headnorm_bwd(dO3, qkv, gamma, stats, T, h, dk, p, norm_mask)
        dwqkv = torch.empty(3 * d, d, dtype=torch.float32, device=dev)
        dbqkv = torch.empty(3 * d, dtype=torch.float32, device=dev) if hbq else None
        dx = torch.empty(T, d, dtype=torch.float32, device=dev)
        with H.side_branch(dev, T):     # weight gradient next to the data gradient
            H.gemm(dqkv, xc, dwqkv, 3 * d, d, T, layout_a=1, layout_b=1, lda=3 * d, ldb=d, ldc=d, split_k=0,
                   a_colsum=dbqkv)
        dxm, want = None, ctx.in_mask
        if want is not None:
            dxm = torch.empty_like(dx)
        H.gemm(dqkv, wq, dx, T, d, 3 * d, layout_b=1, lda=3 * d, ldb=d, ldc=d, res=g_in if has_res else None,
               ldr=d, weight_b=True, c_masked=dxm, ldc_masked=d,
               c_mask=H.dropout_desc(want[0], want[1], dev) if want else None)
        if dxm is not None:
            _offer_twin(dx, dxm, *want)
        H.join_side(dev)
        dres = None                                          # folded into dx (res is x): contributes nothing
        if not norm_mask:
            dgamma = dbeta = None
        return (dx.reshape(xshape), None, dwqkv, dbqkv, dgamma, dbeta, dwfc, dbfc, dres, None, None)


def simple_attention(x, pos, wqkv, bqkv, gamma, beta, wfc, bfc, *, kind: str, n_head: int, norm_mask: int,
                     eps: float, res=None, sign: float = 1.0, p_out: float = 0.0, need_weights: bool = True):
    """Self-attention block; ``res`` must be ``x`` (or None).  Returns (out, attn_weight).  For the Fourier
    type ``need_weights=False`` selects the fused kernel that never materialises the n x n matrix
    (attn_weight is then None); the Galerkin matrix is small and always returned."""
    _check_res_is_x(res, x, "simple_attention")
    mode = _attn_mode
    mask, p_attn = None, 0.0
    if mode == "reference":
        p_attn = 0.5
    elif mode == "replay":
        if not _attn_masks:
            raise RuntimeError("attention dropout mode 'replay' but no mask queued")
        m = _attn_masks.pop(0).to(device=x.device, dtype=torch.float32)
        if kind == "galerkin":
            Dr = m.shape[-1]
            DP = H.round4(Dr)
            mask = torch.zeros(*m.shape[:2], DP, DP, dtype=torch.float32, device=x.device)
            mask[..., :Dr, :Dr] = m
        else:
            mask = _c(m)
    cfg = (kind, int(n_head), int(norm_mask), float(eps), float(sign), float(p_attn), float(p_out),
           bool(need_weights))
    out, w = SimpleAttentionFn.apply(x, pos, wqkv, bqkv, gamma, beta, wfc, bfc, res, cfg, mask)
    return out, (w if w.numel() else None)
