"""ctypes binding of libgt_hip.so (the C ABI declared in include/gt_hip.h).

This module is the only place that touches the shared library.  It fails loudly when the
library is missing or cannot be loaded: there is no CPU / eager fallback for the hot path.
PyTorch is used for device memory, streams and autograd plumbing only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib")
_LIB_NAME = os.environ.get("GT_HIP_LIB", "libgt_hip.so")     # GT_HIP_LIB=libgt_hip_emu.so for the debug twin

ABI_VERSION = 21          # GT_ABI_VERSION of include/gt_hip.h this binding was written against
ACT_NONE, ACT_RELU, ACT_SILU, ACT_GELU = 0, 1, 2, 3
ACT_DROP_SILU = 4          # gt_gemm only: dropout in front of the SiLU, `pre` = keepscale * silu' (gt_hip.h)
ACT_SILU2 = 5              # gt_gemm only: silu(silu(v)), `pre` = silu'(v) * silu'(silu(v)) (gt_hip.h)
AUX_NONE, AUX_GT0, AUX_DSILU, AUX_MUL = 0, 1, 2, 3
EP_NORMAL, EP_ROWDOT, EP_MLP_BWD, EP_HEADNORM = 0, 1, 2, 3
PREC_F32, PREC_BF16X3, PREC_BF16X2, PREC_BF16 = 0, 1, 2, 3
PREC_F16X2 = 4
PREC_CODE = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16x2": PREC_BF16X2, "bf16": PREC_BF16, "f16x2": PREC_F16X2}
SPLIT_EXACT = ("bf16x3", "f16x2")      # the fp32-class split-operand modes: same kernel selection, same fused paths
ACT_CODE = {None: ACT_NONE, "none": ACT_NONE, "identity": ACT_NONE, "relu": ACT_RELU, "silu": ACT_SILU,
            "gelu": ACT_GELU}      # gelu: drop_act only (the GEMM epilogues reject it)


class GtDropout(C.Structure):
    _fields_ = [("p", C.c_float), ("salt", C.c_uint32), ("seed", C.c_void_p)]


class GtResizeAffine(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("rp", C.c_int32), ("rp_a", C.c_void_p), ("rp_lda", C.c_int64),
                ("rp_b", C.c_void_p), ("rp_ldb", C.c_int64)]


class GtGemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("layout_a", C.c_int32), ("layout_b", C.c_int32),
        ("batch0", C.c_int32), ("batch1", C.c_int32), ("split_k", C.c_int32),
        ("A", C.c_void_p), ("lda", C.c_int64), ("a_bs0", C.c_int64), ("a_bs1", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64), ("b_bs0", C.c_int64), ("b_bs1", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_bs0", C.c_int64), ("c_bs1", C.c_int64),
        ("a_drop", GtDropout), ("a_drop_sign", C.c_float),
        ("a_drop_ld", C.c_int64), ("a_drop_bstride", C.c_int64), ("a_colsum", C.c_void_p),
        ("alpha", C.c_float), ("bias", C.c_void_p),
        ("rp", C.c_int32), ("rp_a", C.c_void_p), ("rp_lda", C.c_int64), ("rp_a_bs0", C.c_int64),
        ("rp_b", C.c_void_p), ("rp_ldb", C.c_int64),
        ("add", C.c_void_p), ("ldadd", C.c_int64), ("add_bs0", C.c_int64), ("add_bs1", C.c_int64),
        ("pre", C.c_void_p), ("ldpre", C.c_int64),
        ("act", C.c_int32), ("aux_op", C.c_int32), ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("aux_bs0", C.c_int64), ("aux_bs1", C.c_int64), ("aux_scale", C.c_float),
        ("drop", GtDropout),
        ("res", C.c_void_p), ("ldr", C.c_int64), ("r_bs0", C.c_int64), ("r_bs1", C.c_int64),
        ("out_scale", C.c_float),
        ("ep_mode", C.c_int32), ("n_out", C.c_int32), ("w2", C.c_void_p), ("ldw2", C.c_int64),
        ("b2", C.c_void_p), ("out2", C.c_void_p), ("g2", C.c_void_p), ("dw2", C.c_void_p),
        ("K2", C.c_int32), ("A2", C.c_void_p), ("lda2", C.c_int64), ("a2_bs0", C.c_int64), ("a2_bs1", C.c_int64),
        ("B2", C.c_void_p), ("ldb2", C.c_int64), ("b2_bs0", C.c_int64), ("b2_bs1", C.c_int64),
        ("hn_gamma", C.c_void_p), ("hn_beta", C.c_void_p), ("hn_pos", C.c_void_p), ("hn_out", C.c_void_p),
        ("hn_stats", C.c_void_p), ("hn_h", C.c_int32), ("hn_dk", C.c_int32), ("hn_p", C.c_int32),
        ("hn_norm_mask", C.c_int32), ("hn_eps", C.c_float),
        ("precision", C.c_int32),
        ("cv_h", C.c_int32), ("cv_w", C.c_int32), ("cv_c", C.c_int32), ("cv_wgrad", C.c_int32),
        ("hn_skip_raw_mask", C.c_int32), ("hn_plain", C.c_int32),
        ("b_packed", C.c_void_p),
        ("c_masked", C.c_void_p), ("ldc_masked", C.c_int64), ("c_mask", GtDropout),
    ]


_lib = None

_PROTOS = {
    "gt_abi_version": (C.c_int, []),
    "gt_target_arch": (C.c_char_p, []),
    "gt_seed_advance": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "gt_dropout_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(GtDropout), C.c_void_p]),
    "gt_gemm_desc_init": (None, [C.POINTER(GtGemmDesc)]),
    "gt_gemm_ws_bytes": (C.c_int64, [C.POINTER(GtGemmDesc)]),
    "gt_gemm": (C.c_int, [C.POINTER(GtGemmDesc), C.c_void_p, C.c_int64, C.c_void_p]),
    "gt_gemm_packed_b_bytes": (C.c_int64, [C.POINTER(GtGemmDesc)]),
    "gt_gemm_pack_b_many": (C.c_int, [C.POINTER(GtGemmDesc), C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "gt_gemm_plan": (C.c_int, [C.POINTER(GtGemmDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                               C.POINTER(C.c_int32)]),
    "gt_gemm_kernel_name": (C.c_int, [C.POINTER(GtGemmDesc), C.c_char_p, C.c_int32]),
    "gt_colsum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(GtDropout), C.c_float,
                            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gt_act_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "gt_slab_reduce": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p,
                                 C.c_void_p]),
    "gt_headnorm_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 5 + [C.c_float, C.c_void_p, C.c_void_p,
                                                                     C.c_void_p]),
    "gt_headnorm_bwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 5 + [C.c_void_p] * 4 + [C.c_int64,
                                                                                        C.c_void_p]),
    "gt_headnorm_bwd_ws_bytes": (C.c_int64, [C.c_int32] * 3),
    "gt_galerkin_ktv_slabs": (C.c_int32, [C.c_int32, C.c_int32]),
    "gt_galerkin_ktv": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p, C.c_int32, C.c_void_p]),
    "gt_galerkin_ktv_affine": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 5 + [C.c_void_p, C.c_int32, C.c_void_p]),
    "gt_galerkin_dkv": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 4 + [C.c_void_p]),
    "gt_galerkin_dkv_ln_ws_bytes": (C.c_int64, [C.c_int32] * 3),
    "gt_galerkin_dkv_ln_plain": (C.c_int, [C.c_void_p] * 8 + [C.c_int32] * 5 + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p]),
    "gt_galerkin_dkv_ln": (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 5 + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p]),
    "gt_fourier_attn": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_float, C.c_void_p, C.POINTER(GtDropout),
                                                                C.c_int32, C.c_void_p]),
    "gt_fourier16_image_bytes": (C.c_int64, [C.c_int32] * 4),
    "gt_fourier16_presplit": (C.c_int, [C.c_void_p] * 8 + [C.c_int32] * 4 + [C.c_void_p]),
    "gt_fourier16_attn": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_float, C.c_void_p, C.POINTER(GtDropout),
                                                                  C.c_int32, C.c_int32, C.c_void_p]),
    "gt_dropout_block16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(GtDropout), C.c_void_p]),
    "gt_dropact_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(GtDropout), C.c_int32, C.POINTER(GtDropout),
                                 C.c_int32, C.c_void_p]),
    "gt_dropact_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(GtDropout), C.c_int32,
                                 C.POINTER(GtDropout), C.c_int32, C.c_void_p]),
    "gt_mlp_head_fwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 4 +
                        [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gt_mlp_head_bwd_ws_bytes": (C.c_int64, [C.c_int64]),
    "gt_mlp_head_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 3 +
                        [C.c_int32, C.c_int32] + [C.c_void_p] * 7 + [C.c_int64, C.c_void_p]),
    "gt_mlp_head_bwd_gated": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 3 +
                              [C.c_int32, C.c_int32] + [C.c_void_p] * 8 + [C.c_int64, C.c_void_p]),
    "gt_dft_synthesis_gated": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                                           C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gt_dft_analysis": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p]),
    "gt_dft_synthesis": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                                     C.c_int32, C.c_void_p, C.c_void_p]),
    "gt_galerkin_finalize_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64] + [C.c_int32] * 6 +
                                 [C.c_void_p, C.POINTER(GtDropout), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_void_p]),
    "gt_galerkin_finalize_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GtDropout),
                                           C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p,
                                                                            C.c_void_p]),
    "gt_layernorm_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                                      C.c_void_p, C.c_void_p]),
    "gt_layernorm_bwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int32, C.c_int32] + [C.c_void_p] * 4 +
                         [C.c_int64, C.c_void_p]),
    "gt_layernorm_bwd_ws_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "gt_modemix_fwd": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_int64, C.c_int64] +
                       [C.c_int32] * 3 + [C.c_void_p, C.c_void_p]),
    "gt_modemix_bwd_ws_bytes": (C.c_int64, [C.c_int32] * 4),
    "gt_modemix_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_int64, C.c_int64] +
                       [C.c_int32] * 3 + [C.c_void_p] * 3 + [C.c_int64, C.c_void_p]),
    "gt_bilinear2d_fwd": (C.c_int, [C.c_void_p] * 2 + [C.c_int32] * 9 + [C.c_void_p]),
    "gt_bilinear2d_fwd_affine": (C.c_int, [C.c_void_p] * 2 + [C.c_int32] * 9 + [C.POINTER(GtResizeAffine),
                                                                                C.c_void_p]),
    "gt_conv3x3_resize_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 7 + [C.POINTER(GtDropout), C.c_int32,
                                                                          C.c_void_p]),
    "gt_conv3x3_resize_bwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 7 + [C.POINTER(GtDropout), C.c_int32,
                                                                          C.c_void_p, C.c_void_p, C.c_int64,
                                                                          C.c_void_p]),
    "gt_conv3x3_resize_bwd_ws_bytes": (C.c_int64, [C.c_int32] * 5),
    "gt_conv3x3_resize_fwd_nhwc": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 7 + [C.POINTER(GtDropout), C.c_int32,
                                                                               C.c_void_p, C.c_void_p]),
    "gt_conv3x3_resize_bwd_nhwc": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 7 + [C.POINTER(GtDropout), C.c_int32,
                                                                               C.c_void_p, C.c_void_p, C.c_void_p,
                                                                               C.c_int64, C.c_void_p]),
    "gt_conv3x3_resize_bits_bytes": (C.c_int64, [C.c_int32] * 4),
    "gt_debug_conv0_mask": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gt_conv3x3_wgrad_nhwc": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int32] * 5 +
                              [C.c_float, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "gt_conv3x3_wgrad_nhwc_ws_bytes": (C.c_int64, [C.c_int32] * 5),
    "gt_bilinear2d_seg_fwd": (C.c_int, [C.c_void_p] * 2 + [C.c_int32] * 9 + [C.c_void_p, C.c_void_p]),
    "gt_bilinear2d_seg_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 9 + [C.c_void_p, C.c_int32, C.c_void_p]),
    "gt_bilinear2d_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 9 + [C.c_void_p]),
    "gt_ffn_fwd_ws_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "gt_ffn_fwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 5 +
                   [C.POINTER(GtDropout), C.POINTER(GtDropout), C.c_int32] + [C.c_void_p] * 6 + [C.c_int64, C.c_void_p]),
    "gt_ffn_bits_bytes": (C.c_int64, [C.c_int64]),
    "gt_ffn_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 3 + [C.c_float] +
                   [C.c_void_p] * 4 + [C.POINTER(GtDropout)] + [C.c_void_p] * 3 + [C.c_int64, C.c_void_p]),
    "gt_grad_sqnorm_ws_bytes": (C.c_int64, []),
    "gt_grad_sqnorm": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gt_adam_clip_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_void_p] +
                          [C.c_float] * 4 + [C.c_void_p, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


def lib_path() -> str:
    return os.path.join(_LIB_DIR, _LIB_NAME)


def lib():
    """Load (once) and return the shared library.  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP hot path is not built (run "
                f"`python galerkin-transformer_amd/build.py` or __graft_entry__.build()). "
                f"There is no CPU fallback for these operators.")
        try:
            handle = C.CDLL(path)      # torch is already imported: its libamdhip64.so.7 is reused
        except OSError as e:
            raise RuntimeError(f"cannot load {path}: {e}") from e
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.gt_abi_version() != ABI_VERSION:
            raise RuntimeError("libgt_hip ABI version mismatch")
        _lib = handle
    return _lib


# ----------------------------------------------------------------------------------- per-launch timing
_prof = None


class Profile:
    """``with Profile() as p:`` brackets every C-ABI launch with HIP events on the launch stream
    (torch's current stream is the stream handed to the library).  Used by bench.py for the
    roofline leg; adds a few microseconds of host work per launch, so never inside timed regions."""

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _prof
        _prof = self
        return self

    def __exit__(self, *exc):
        global _prof
        _prof = None

    def table(self, by_shape=False):
        """key -> dict(calls, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for key, flops, nbytes, e0, e1, _, shape in self.records:
            if by_shape and shape is not None:
                key = f"{key} {tuple(shape)}"
            r = out.setdefault(key, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            r["calls"] += 1
            r["ms"] += e0.elapsed_time(e1)
            r["flops"] += flops
            r["bytes"] += nbytes
        return out


_DEBUG_SYNC = bool(os.environ.get("GT_DEBUG_SYNC"))


def _timed(key, flops, nbytes, fn, replay=None, shape=None):
    if _DEBUG_SYNC:        # print-before-launch + sync-after: the last line names a faulting launch
        print(f"[gt] {key} {shape}", flush=True)
        rc = fn()
        torch.cuda.synchronize()
        return rc
    if _prof is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    _prof.records.append((key, flops, nbytes, e0, e1, replay, shape))
    return rc


class GtError(RuntimeError):
    pass


class GtNotSupported(GtError, NotImplementedError):
    """GT_ENOTSUP: the library has no kernel for this combination (no silent fallback exists)."""


_ERR = {-1: "GT_EINVAL (bad shape/flags)", -2: "GT_EALIGN (misaligned pointer/ld)",
        -3: "GT_EWS (scratch too small)", -4: "GT_ENOTSUP (not implemented)"}


def check(rc: int, what: str):
    if rc == -4:
        raise GtNotSupported(f"{what}: {_ERR[-4]}")
    if rc != 0:
        raise GtError(f"{what} failed: {_ERR.get(rc, 'hipError ' + str(rc))}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def need_f32_cuda(*ts: torch.Tensor):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("galerkin_transformer HIP operators need tensors on a ROCm device "
                               "(got a CPU tensor); there is no CPU fallback for the hot path")
        if t.dtype != torch.float32:
            raise TypeError(f"galerkin_transformer HIP operators are fp32 (got {t.dtype})")


# ----------------------------------------------------------------------------------- GEMM arithmetic mode
# "f16x2" (default since round 4): the packed-B kernels (a weight against >= 16384 token rows, the implicit convolutions)
#           split every operand into two fp16 terms under a running power-of-two scale and run three products on the f16
#           MFMA pipe; every other split-operand launch is bf16x3.  fp32-class results: the whole 1e-5 suite passes in it;
# "bf16x3": fp32 operands split exactly into three bf16 terms, six plane products on the bf16 MFMA pipe,
#           fp32 accumulation -- fp32-class results (the 1e-5 parity gate holds) at 16/6 of the fp32 matrix rate;
# "f32":    v_mfma_f32_16x16x4_f32, bit-for-bit an fp32 FMA chain;
# "bf16x2" / "bf16": two terms / plain bf16 operands -- throughput modes with their own looser gates.
# See gt_gemm_desc.precision in include/gt_hip.h.  GT_PRECISION in the environment sets the initial mode.
_precision = [PREC_CODE[os.environ.get("GT_PRECISION", "f16x2")]]


def set_precision(mode: str) -> str:
    """Select the arithmetic of every gt_gemm contraction issued from now on; returns the previous mode."""
    if mode not in PREC_CODE:
        raise ValueError(f"precision must be one of {sorted(PREC_CODE)}, got {mode!r}")
    old = get_precision()
    _precision[0] = PREC_CODE[mode]
    return old


def get_precision() -> str:
    return {v: k for k, v in PREC_CODE.items()}[_precision[0]]


# Per-launch-class arithmetic (diagnostics: tools/parity_bisect.py finds which class of contraction carries an error).
# Classes of a gt_gemm launch:  "hn" QKV projection with the head-norm epilogue * "conv" implicit 3x3 convolution
# (forward / data gradient) * "convw" its weight gradient * "wgrad" token-contracted weight gradients (both operands
# x-contiguous, unbatched) * "batched" per-sample products (Q'P, dP^T, the dQ block) * "tok" every other token-row
# product * "head" the fused regression head (gt_mlp_head_*).  A class named here overrides both the module mode and a launch's own `precision=` argument.
# GT_PREC_CLASS="wgrad=f32,tok=f32" sets it from the environment.
_prec_class = {}


def set_precision_classes(classes: Optional[dict]):
    """classes: {"wgrad": "f32", ...} or None / {} to switch the override off; returns the previous mapping."""
    old = dict(_prec_class)
    _prec_class.clear()
    for k, v in (classes or {}).items():
        if k not in ("hn", "conv", "convw", "wgrad", "batched", "tok", "head", "fourier") or v not in PREC_CODE:
            raise ValueError(f"set_precision_classes: bad entry {k}={v}")
        _prec_class[k] = v
    return old


if os.environ.get("GT_PREC_CLASS"):
    set_precision_classes(dict(kv.split("=") for kv in os.environ["GT_PREC_CLASS"].split(",") if kv))


def gemm_class(layout_a: int, layout_b: int, batch, hn, conv, conv_wgrad: bool) -> str:
    if hn is not None:
        return "hn"
    if conv is not None:
        return "convw" if conv_wgrad else "conv"
    if tuple(batch) != (1, 1):
        return "batched"
    return "wgrad" if (layout_a == 1 and layout_b == 1) else "tok"


# ----------------------------------------------------------------------------------- scratch / rng state
_ws_cache = {}            # (device, stream) -> [buffer, pinned]; insertion order = recency (re-inserted on use)
_ws_retired = []          # outgrown / evicted PINNED buffers stay alive: a captured HIP graph has their pointer baked in
_WS_MIN = 64 << 20
_WS_MAX_UNPINNED = 6      # scratch of at most this many non-capture streams is kept (least recently used go first)


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Scratch buffer per (device, stream): consumers of one stream are stream-ordered, so reuse is safe; two
    streams never share a buffer (a graph capture runs on its own stream, so captured work gets its own buffer
    from the capture's memory pool).  A buffer handed out during a capture is pinned for the life of the process (the
    graph replays into it); a pinned buffer that has to grow is retired, not freed.  Buffers of ordinary streams are
    kept for the _WS_MAX_UNPINNED most recently used streams only, so short-lived streams do not pin 64 MB each and a
    recycled stream handle cannot inherit a buffer that a live graph owns."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, dev, torch.cuda.current_stream(dev).cuda_stream)
    capturing = torch.cuda.is_current_stream_capturing()
    ent = _ws_cache.pop(key, None)
    if ent is not None and ent[1] and not capturing:
        # the handle of a capture stream came back as an ordinary stream: its buffer belongs to the graph
        _ws_retired.append(ent[0])
        ent = None
    if ent is None or ent[0].numel() < nbytes:
        size = max(_WS_MIN, int(nbytes * 1.25))
        if ent is not None and ent[1]:
            _ws_retired.append(ent[0])
        ent = [torch.empty(size, dtype=torch.uint8, device=device), capturing]
    ent[1] = ent[1] or capturing
    _ws_cache[key] = ent                                   # most recently used last
    unpinned = [k for k, v in _ws_cache.items() if not v[1]]
    for k in unpinned[:max(0, len(unpinned) - _WS_MAX_UNPINNED)]:
        del _ws_cache[k]
    return ent[0]


# ----------------------------------------------------------------------------------- second stream
# A weight gradient (K = tokens, few output tiles, VALU-bound on the operand split) and the data gradient of the same
# layer (memory-phase bound) are independent and want different resources: the backward passes put the weight gradient
# on a side stream between a fork and a join, so the two kernels share the chip.  Inside a graph capture the fork/join
# become parallel branches of the graph.  GT_DUAL_STREAM=0 (or a hip-event profile in progress) keeps one stream.
_dual_stream = [os.environ.get("GT_DUAL_STREAM", "1") != "0"]
_side_streams = {}
_side_pending = set()
SIDE_MIN_ROWS = 32768


def _side_stream(dev: int) -> "torch.cuda.Stream":
    st = _side_streams.get(dev)
    if st is None:
        st = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return st


class side_branch:
    """``with side_branch(device): launch(...)``: the body runs on the device's side stream, after everything already
    queued on the current stream.  ``join_side(device)`` makes the current stream wait for it; every user joins before
    its outputs leave the function."""

    def __init__(self, device: torch.device, rows: int = 1 << 30):
        """rows: token rows of the forked product.  An eager step pays two event record / wait pairs per fork; below
        SIDE_MIN_ROWS that costs more than the overlap returns (B = 4 eager: 623 -> 491 samples/s when every product
        forked), so small products fork only inside a graph capture, where the fork is free."""
        self.dev = device.index if device.index is not None else torch.cuda.current_device()
        self.ctx = None
        self.rows = rows

    def __enter__(self):
        if (_dual_stream[0] and _prof is None and not _DEBUG_SYNC
                and (self.rows >= SIDE_MIN_ROWS or torch.cuda.is_current_stream_capturing())):
            side = _side_stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            _side_pending.add(self.dev)
        return False


def join_side(device: torch.device):
    dev = device.index if device.index is not None else torch.cuda.current_device()
    if dev in _side_pending:
        torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
        _side_pending.discard(dev)


_seed_cache = {}


def seed_state(device: torch.device) -> torch.Tensor:
    """Device-resident uint64 dropout seed (as int64 tensor of one element)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    s = _seed_cache.get(key)
    if s is None:
        s = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
        _seed_cache[key] = s
    return s


# call-site salts: every operator call that draws masks takes the next value(s), so two sites never share a mask.
# set_seed() rewinds the counter: a seeded run reproduces whatever ran earlier in the process.  A captured HIP
# graph keeps the salts it was captured with and varies its masks through the device-resident seed instead.
_salt = [1]


def next_salt(k: int = 4) -> int:
    s = _salt[0]
    _salt[0] = (s + k) & 0x7FFFFFFF
    return s


def set_seed(seed: int, device: Optional[torch.device] = None, rewind_salts: bool = True):
    """Set the device-resident dropout seed.  rewind_salts=False keeps the call-site counter running (use it to
    reseed between steps of one traced / captured program whose salts must stay aligned)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    seed_state(device).fill_(seed & 0x7FFFFFFFFFFFFFFF)
    if rewind_salts:
        _salt[0] = 1


def advance_seed(device: Optional[torch.device] = None, inc: int = 1):
    """Bump the dropout seed (captured into graphs: every replay draws fresh masks)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    s = seed_state(device)
    check(lib().gt_seed_advance(s.data_ptr(), inc, stream_ptr()), "gt_seed_advance")


def dropout_desc(p: float, salt: int, device: torch.device) -> GtDropout:
    d = GtDropout()
    d.p = float(p) if p else 0.0
    d.salt = int(salt) & 0xFFFFFFFF
    d.seed = seed_state(device).data_ptr() if d.p > 0 else None
    return d


# ----------------------------------------------------------------------------------- weights packed ahead
class _WeightPacks:
    """One pack launch per step for the model's weights (dispatch diet): a token product whose B operand is flagged as a
    model weight (`gemm(..., weight_b=True)`: a parameter at a stable address) registers (B, layout, N, K) on first sight;
    `refresh()` -- called by the training step before its forward, i.e. after the optimizer last changed the weights --
    packs every registered weight in ONE launch (gt_gemm_pack_b_many) into persistent buffers, and until `invalidate()`
    the products take them through gt_gemm_desc.b_packed instead of packing per call.  Outside refresh() / invalidate()
    brackets nothing changes (per-call packs).  GT_PACK_ALL=0 switches it off."""

    def __init__(self):
        self.enabled = os.environ.get("GT_PACK_ALL", "1") != "0"
        self.entries = {}          # key -> [desc copy, buffer or None, B (kept alive: its address is the key), last round seen]
        self.ready = set()
        self.round = 0
        # buffers a captured HIP graph has baked in (raw pointers in its pack launch and its products): never freed while the
        # process lives -- a graph replayed after its entries were evicted would read and write freed memory
        self.pinned = set()
        self._keep = []

    def use(self, d, B):
        """b_packed for this product if it was packed in the current bracket.  A product is packed ahead only once it has
        been seen in two different brackets with the same operand address (a weight that is re-materialised every step --
        a torch.cat of parameters, a padded copy -- never qualifies and is forgotten)."""
        if not self.enabled or self.round == 0 or d.precision != PREC_F16X2:     # round 0: nobody brackets steps with refresh()
            return
        # M is part of the key: whether gt_gemm(d) is ONE packed-B launch (and so whether b_packed may be set at all) depends
        # on it -- the packed kernels want M >= 16384 and M >= 8 N, and a wide product is cut in two launches once its
        # aligned part fills the chip (gt_gemm_packed_b_bytes is asked per (weight, M))
        key = (B.data_ptr(), d.layout_b, d.ldb, d.N, d.K, d.M, B.device.index)
        ent = self.entries.get(key)
        if ent is None:
            need = int(lib().gt_gemm_packed_b_bytes(C.byref(d)))
            if need > 0 and len(self.entries) < 512:
                self.entries[key] = [GtGemmDesc.from_buffer_copy(d), None, B, self.round, need]
            return
        if ent[1] is None and ent[3] != self.round and not torch.cuda.is_current_stream_capturing():
            ent[1] = torch.empty(ent[4], dtype=torch.uint8, device=B.device)      # second sight: packed from the next bracket on
        ent[3] = self.round
        if key in self.ready:
            d.b_packed = ent[1].data_ptr()
            if torch.cuda.is_current_stream_capturing():
                self.pinned.add(key)

    def refresh(self):
        self.ready = set()
        if not self.enabled:
            return
        # forget what the last bracket did not use (transient operands); pack what has a buffer
        for k in [k for k, e in self.entries.items() if e[3] < self.round - 1 and k not in self.pinned]:
            del self.entries[k]
        self.round += 1
        items = [(k, e) for k, e in self.entries.items() if e[1] is not None]
        for i0 in range(0, len(items), 64):
            chunk = items[i0:i0 + 64]
            descs = (GtGemmDesc * len(chunk))(*[e[0] for _, e in chunk])
            outs = (C.c_void_p * len(chunk))(*[e[1].data_ptr() for _, e in chunk])
            rc = _timed("gt_gemm_pack_b_many", 0.0, 0.0,
                        lambda: lib().gt_gemm_pack_b_many(descs, outs, len(chunk), stream_ptr()), shape=(len(chunk),))
            if rc == -4:               # GT_ENOTSUP (an entry no longer takes the packed path): per-call packs this bracket
                self.ready = set()
                return
            check(rc, "gt_gemm_pack_b_many")
            self.ready.update(k for k, _ in chunk)

    def invalidate(self):
        self.ready = set()

    def clear(self):
        self._keep.extend(self.entries[k][1] for k in self.pinned if k in self.entries)    # (a captured graph may still replay)
        self.pinned = set()
        self.entries.clear()
        self.ready = set()


weight_packs = _WeightPacks()


# ----------------------------------------------------------------------------------- GEMM
def gemm(A: torch.Tensor, B: torch.Tensor, Cout: torch.Tensor, M: int, N: int, K: int, *,
         layout_a: int = 0, layout_b: int = 0, lda: int, ldb: int, ldc: int,
         batch: Tuple[int, int] = (1, 1), a_bs=(0, 0), b_bs=(0, 0), c_bs=(0, 0), split_k: int = 1,
         alpha: float = 1.0, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
         a_drop: Optional[GtDropout] = None, a_drop_sign: float = 1.0, a_drop_ld: int = 0,
         a_drop_bstride: int = 0, a_colsum: Optional[torch.Tensor] = None,
         rp: int = 0, rp_a: Optional[torch.Tensor] = None, rp_lda: int = 0, rp_a_bs0: int = 0,
         rp_b: Optional[torch.Tensor] = None, rp_ldb: int = 0,
         add: Optional[torch.Tensor] = None, ldadd: int = 0, add_bs=(0, 0),
         pre: Optional[torch.Tensor] = None, ldpre: int = 0,
         aux_op: int = AUX_NONE, aux: Optional[torch.Tensor] = None, ldaux: int = 0, aux_bs=(0, 0),
         aux_scale: float = 1.0, drop: Optional[GtDropout] = None,
         res: Optional[torch.Tensor] = None, ldr: int = 0, r_bs=(0, 0), out_scale: float = 1.0,
         ep_mode: int = 0, w2: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None,
         out2: Optional[torch.Tensor] = None, g2: Optional[torch.Tensor] = None,
         dw2: Optional[torch.Tensor] = None,
         K2: int = 0, A2: Optional[torch.Tensor] = None, lda2: int = 0, a2_bs=(0, 0),
         B2: Optional[torch.Tensor] = None, ldb2: int = 0, b2_bs=(0, 0), hn: Optional[dict] = None,
         precision: Optional[str] = None, conv: Optional[Tuple[int, int, int]] = None, conv_wgrad: bool = False,
         weight_b: bool = False, c_masked: Optional[torch.Tensor] = None, ldc_masked: int = 0,
         c_mask: Optional[GtDropout] = None):
    """Thin wrapper over gt_gemm (see include/gt_hip.h for the semantics).  weight_b: B is a model weight at a stable address
    (eligible for the once-per-step pack, weight_packs).  precision=None uses the module mode
    (set_precision).  conv=(H, W, C): A is a channels-last [B, H, W, C] image and the product is the implicit 3x3
    convolution (K = 9*C; gt_hip.h: cv_*); with conv_wgrad, B is that image and the nine batch entries are the taps of the
    weight gradient.  Cout may be None for a GT_EP_HEADNORM launch that stores no raw projection (hn["skip_raw"] == 7)."""
    need_f32_cuda(A, B, Cout, bias, rp_a, rp_b, add, pre, aux, res, a_colsum)
    L = lib()
    d = GtGemmDesc()
    L.gt_gemm_desc_init(C.byref(d))
    d.M, d.N, d.K = M, N, K
    d.layout_a, d.layout_b = layout_a, layout_b
    d.batch0, d.batch1 = batch
    d.split_k = split_k
    d.precision = _precision[0] if precision is None else PREC_CODE[precision]
    if _prec_class:
        pc = _prec_class.get(gemm_class(layout_a, layout_b, batch, hn, conv, conv_wgrad))
        if pc is not None:
            d.precision = PREC_CODE[pc]
    d.A, d.lda, d.a_bs0, d.a_bs1 = A.data_ptr(), lda, a_bs[0], a_bs[1]
    d.B, d.ldb, d.b_bs0, d.b_bs1 = B.data_ptr(), ldb, b_bs[0], b_bs[1]
    d.C, d.ldc, d.c_bs0, d.c_bs1 = ptr(Cout), ldc, c_bs[0], c_bs[1]
    if a_drop is not None and a_drop.p > 0:
        d.a_drop = a_drop
        d.a_drop_ld, d.a_drop_bstride = a_drop_ld, a_drop_bstride
    d.a_drop_sign = a_drop_sign
    d.a_colsum = ptr(a_colsum)
    d.alpha = alpha
    d.bias = ptr(bias)
    if rp:
        d.rp, d.rp_a, d.rp_lda, d.rp_a_bs0 = rp, rp_a.data_ptr(), rp_lda, rp_a_bs0
        d.rp_b, d.rp_ldb = rp_b.data_ptr(), rp_ldb
    if add is not None:
        d.add, d.ldadd, d.add_bs0, d.add_bs1 = add.data_ptr(), ldadd, add_bs[0], add_bs[1]
    if pre is not None:
        d.pre, d.ldpre = pre.data_ptr(), ldpre
    d.act = act
    if aux_op:
        d.aux_op, d.aux, d.ldaux = aux_op, aux.data_ptr(), ldaux
        d.aux_bs0, d.aux_bs1, d.aux_scale = aux_bs[0], aux_bs[1], aux_scale
    if drop is not None and drop.p > 0:
        d.drop = drop
    if res is not None:
        d.res, d.ldr, d.r_bs0, d.r_bs1 = res.data_ptr(), ldr, r_bs[0], r_bs[1]
    d.out_scale = out_scale
    if ep_mode:
        need_f32_cuda(w2, b2, out2, g2, dw2)
        d.ep_mode, d.n_out, d.w2, d.ldw2 = ep_mode, w2.shape[0], w2.data_ptr(), w2.stride(0)
        d.b2, d.out2, d.g2, d.dw2 = ptr(b2), ptr(out2), ptr(g2), ptr(dw2)
    if K2:
        need_f32_cuda(A2, B2)
        d.K2, d.A2, d.lda2, d.a2_bs0, d.a2_bs1 = K2, A2.data_ptr(), lda2, a2_bs[0], a2_bs[1]
        d.B2, d.ldb2, d.b2_bs0, d.b2_bs1 = B2.data_ptr(), ldb2, b2_bs[0], b2_bs[1]
    if hn is not None:          # GT_EP_HEADNORM: gamma, beta, pos, out, stats, h, dk, p, norm_mask, eps
        need_f32_cuda(hn.get("gamma"), hn.get("beta"), hn.get("pos"), hn["out"], hn.get("stats"))
        d.ep_mode = EP_HEADNORM
        d.hn_gamma, d.hn_beta, d.hn_pos = ptr(hn.get("gamma")), ptr(hn.get("beta")), ptr(hn.get("pos"))
        d.hn_out, d.hn_stats = hn["out"].data_ptr(), ptr(hn.get("stats"))
        d.hn_h, d.hn_dk, d.hn_p, d.hn_norm_mask, d.hn_eps = hn["h"], hn["dk"], hn["p"], hn["norm_mask"], hn["eps"]
        d.hn_skip_raw_mask = hn.get("skip_raw", 0)
        d.hn_plain = int(bool(hn.get("plain", False)))
    if conv is not None:
        d.cv_h, d.cv_w, d.cv_c = conv
        d.cv_wgrad = int(conv_wgrad)
    if c_masked is not None:    # the result once more under a second dropout mask (gt_hip.h: c_masked)
        need_f32_cuda(c_masked)
        d.c_masked, d.ldc_masked = c_masked.data_ptr(), ldc_masked
        if c_mask is not None and c_mask.p > 0:
            d.c_mask = c_mask
    if weight_b:
        weight_packs.use(d, B)
    need = L.gt_gemm_ws_bytes(C.byref(d))
    wsp, wsn = None, 0
    if need > 0:
        ws = workspace(A.device, need)
        wsp, wsn = ws.data_ptr(), ws.numel()
    if _prof is None and not _DEBUG_SYNC:
        check(L.gt_gemm(C.byref(d), wsp, wsn, stream_ptr()), "gt_gemm")
    else:
        nb = batch[0] * batch[1]
        bm, bn, sp = C.c_int32(), C.c_int32(), C.c_int32()
        L.gt_gemm_plan(C.byref(d), C.byref(bm), C.byref(bn), C.byref(sp))
        nm = C.create_string_buffer(160)
        L.gt_gemm_kernel_name(C.byref(d), nm, 160)
        key = (nm.value.decode().replace("(gt::GemmP)", "").replace("(gt::TsmmP)", "").replace("void ", "")
               .replace("gt::", ""))
        key += "+splitk" if (sp.value > 1 and "tsmm" not in key) else ""
        flops = 2.0 * M * N * (K + K2) * nb
        if conv_wgrad:
            nb = 1.0 + 8.0 * (M * N) / (M * K + K * N + M * N)     # both operands are read once for the nine taps
        nbytes = 4.0 * nb * (M * (conv[2] if conv and not conv_wgrad else K) + K * N + M * N * (1 + (res is not None) + (aux is not None) + (c_masked is not None) +
                                                      (add is not None) + (pre is not None)))
        keep = (A, B, Cout, bias, rp_a, rp_b, add, pre, aux, res, d)
        st = stream_ptr()
        call = lambda: L.gt_gemm(C.byref(d), wsp, wsn, st)
        check(_timed(key, flops, nbytes, call, replay=(call, keep), shape=(M, N, K, nb)), "gt_gemm")
    return Cout


_ffn_fused = [os.environ.get("GT_FFN_FUSED", "1") != "0"]


def ffn_fwd_supported(T: int, d: int, f: int, act: int) -> bool:
    """Shapes of the fused FeedForward forward (gt_ffn_fwd): the two-term fp16 arithmetic only."""
    return (_ffn_fused[0] and _precision[0] == PREC_F16X2 and not _prec_class and d == 128 and f == 256 and T >= 16384
            and act in (ACT_RELU, ACT_NONE))


def _ffn_packs(T, specs, dev_tensor):
    """b_packed of the two products of a fused FeedForward launch when the once-per-step pack holds them, else (None, None).
    specs: ((weight, layout_b, ldb, N, K), ...) -- the descriptors H.gemm would build for them."""
    L = lib()
    packs = []
    for (Bw, lb, ldb, N, K) in specs:
        dsc = GtGemmDesc()
        L.gt_gemm_desc_init(C.byref(dsc))
        dsc.M, dsc.N, dsc.K, dsc.lda, dsc.ldb, dsc.ldc, dsc.layout_b = T, N, K, K, ldb, N, lb
        dsc.A, dsc.B, dsc.C = dev_tensor.data_ptr(), Bw.data_ptr(), dev_tensor.data_ptr()
        dsc.precision = PREC_F16X2
        weight_packs.use(dsc, Bw)
        packs.append(dsc.b_packed)
    return (packs[0], packs[1]) if (packs[0] and packs[1]) else (None, None)


def ffn_fwd(x2: torch.Tensor, w1: torch.Tensor, b1, w2: torch.Tensor, b2, res, drop_h, drop_o, act: int,
            hid: torch.Tensor, out: torch.Tensor, want_bits: bool = False):
    """hid = drop_h(act(x2 W1^T + b1)), out = res + drop_o(hid W2^T + b2) in one launch (gt_hip.h: gt_ffn_fwd).  The packed
    weights come from the once-per-step pack when the two products are registered there (weight_packs), else the call packs.
    want_bits: also returns the ReLU / dropout decision bits of the hidden tile for ffn_bwd (an opaque uint8 tensor)."""
    need_f32_cuda(x2, w1, b1, w2, b2, res, hid, out)
    L = lib()
    T, d = x2.shape
    f = w1.shape[0]
    p1, p2 = _ffn_packs(T, ((w1, 0, d, f, d), (w2, 0, f, d, f)), x2)
    wsp, wsn = None, 0
    if p1 is None:
        ws = workspace(x2.device, L.gt_ffn_fwd_ws_bytes(T, d, f))
        wsp, wsn = ws.data_ptr(), ws.numel()
    bits = torch.empty(L.gt_ffn_bits_bytes(T), dtype=torch.uint8, device=x2.device) if want_bits else None
    dh = C.byref(drop_h) if (drop_h is not None and drop_h.p > 0) else None
    do = C.byref(drop_o) if (drop_o is not None and drop_o.p > 0) else None
    st = stream_ptr()
    call = lambda: L.gt_ffn_fwd(x2.data_ptr(), T, d, f, w1.data_ptr(), ptr(b1), w2.data_ptr(), ptr(b2), ptr(res), dh, do, act,
                                hid.data_ptr(), out.data_ptr(), ptr(bits), p1, p2, wsp, wsn, st)
    # algorithmic bytes: x, hid, out (+ the residual when it is another tensor than x) + the decision bits
    own_res = res is not None and res.data_ptr() != x2.data_ptr()
    nbytes = 4.0 * T * (d + f + d + (d if own_res else 0)) + (bits.numel() if bits is not None else 0)
    keep = (x2, w1, b1, w2, b2, res, drop_h, drop_o, hid, out, bits)
    check(_timed("gt_ffn_fwd", 4.0 * T * d * f, nbytes, call, replay=(call, keep), shape=(T, d, f)), "gt_ffn_fwd")
    return bits


def ffn_bwd(gm: torch.Tensor, w2: torch.Tensor, w1: torch.Tensor, bits: torch.Tensor, hid_scale: float, res,
            gh: torch.Tensor, dx: torch.Tensor, dx_masked=None, mask2=None):
    """gh = (gm W2) .* bits * hid_scale, dx = res + gh W1 (+ dx_masked = dx under mask2) in one launch (gt_hip.h: gt_ffn_bwd)."""
    need_f32_cuda(gm, w2, w1, res, gh, dx, dx_masked)
    L = lib()
    T, d = gm.shape
    f = w1.shape[0]
    p2, p1 = _ffn_packs(T, ((w2, 1, f, f, d), (w1, 1, d, d, f)), gm)
    wsp, wsn = None, 0
    if p2 is None:
        ws = workspace(gm.device, L.gt_ffn_fwd_ws_bytes(T, d, f))
        wsp, wsn = ws.data_ptr(), ws.numel()
    m2 = C.byref(mask2) if (mask2 is not None and mask2.p > 0) else None
    st = stream_ptr()
    call = lambda: L.gt_ffn_bwd(gm.data_ptr(), T, d, f, w2.data_ptr(), w1.data_ptr(), bits.data_ptr(), float(hid_scale), ptr(res),
                                gh.data_ptr(), dx.data_ptr(), ptr(dx_masked), m2, p2, p1, wsp, wsn, st)
    nbytes = 4.0 * T * (d + f + d + (d if res is not None else 0) + (d if dx_masked is not None else 0)) + bits.numel()
    keep = (gm, w2, w1, bits, res, gh, dx, dx_masked, mask2)
    check(_timed("gt_ffn_bwd", 4.0 * T * d * f, nbytes, call, replay=(call, keep), shape=(T, d, f)), "gt_ffn_bwd")


def gemm_kernel_name(A, B, M, N, K, *, layout_a=0, layout_b=0, lda, ldb, ldc, split_k=1, precision=None) -> str:
    """Symbol of the kernel gt_gemm launches for a plain (epilogue-free, unbatched) product."""
    d = GtGemmDesc()
    lib().gt_gemm_desc_init(C.byref(d))
    d.M, d.N, d.K, d.layout_a, d.layout_b, d.split_k = M, N, K, layout_a, layout_b, split_k
    d.precision = _precision[0] if precision is None else PREC_CODE[precision]
    d.A, d.lda, d.B, d.ldb, d.ldc = A.data_ptr(), lda, B.data_ptr(), ldb, ldc
    nm = C.create_string_buffer(160)
    check(lib().gt_gemm_kernel_name(C.byref(d), nm, 160), "gt_gemm_kernel_name")
    return nm.value.decode()


def gemm_plan(M, N, K, batch=(1, 1), split_k=1):
    d = GtGemmDesc()
    lib().gt_gemm_desc_init(C.byref(d))
    d.M, d.N, d.K = M, N, K
    d.batch0, d.batch1 = batch
    d.split_k = split_k
    bm, bn, sp = C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().gt_gemm_plan(C.byref(d), C.byref(bm), C.byref(bn), C.byref(sp)), "gt_gemm_plan")
    return bm.value, bn.value, sp.value


# ----------------------------------------------------------------------------------- small wrappers
def colsum(A: torch.Tensor, M: int, N: int, lda: int, a_drop: Optional[GtDropout] = None,
           sign: float = 1.0) -> torch.Tensor:
    need_f32_cuda(A)
    out = torch.empty(N, dtype=torch.float32, device=A.device)
    ws = workspace(A.device, 1024 * max(N, 64) * 4)       # >= the bounded number of partial rows
    dp = C.byref(a_drop) if (a_drop is not None and a_drop.p > 0) else None
    check(_timed("gt_colsum", 0, 0, lambda: lib().gt_colsum(A.data_ptr(), lda, M, N, dp, sign, out.data_ptr(), ws.data_ptr(), ws.numel(),
                          stream_ptr())), "gt_colsum")
    return out


def slab_reduce(slabs: torch.Tensor, n_slabs: int, stride: int, n: int, out: torch.Tensor,
                alpha: float = 1.0):
    need_f32_cuda(slabs, out)
    check(_timed("gt_slab_reduce", 0, 0, lambda: lib().gt_slab_reduce(slabs.data_ptr(), stride, n_slabs, n, alpha, out.data_ptr(), stream_ptr())), "gt_slab_reduce")
    return out


def act_bwd(dout: torch.Tensor, pre: torch.Tensor, act: int) -> torch.Tensor:
    need_f32_cuda(dout, pre)
    dout = dout.contiguous()
    out = torch.empty_like(pre)
    check(_timed("gt_act_bwd", 0, 0, lambda: lib().gt_act_bwd(dout.data_ptr(), pre.data_ptr(), out.data_ptr(), pre.numel(), act, stream_ptr())), "gt_act_bwd")
    return out


def dropout_apply(x: torch.Tensor, d: GtDropout) -> torch.Tensor:
    need_f32_cuda(x)
    out = torch.empty_like(x)
    check(_timed("gt_dropout_apply", 0, 0, lambda: lib().gt_dropout_apply(x.data_ptr(), out.data_ptr(), x.numel(), C.byref(d), stream_ptr())), "gt_dropout_apply")
    return out


def _dref(d):
    return C.byref(d) if (d is not None and d.p > 0) else None


def dropact_fwd(x: torch.Tensor, d1, act1: int, d2, act2: int) -> torch.Tensor:
    """act2(drop2(act1(drop1(x)))) elementwise (gt_dropact_fwd)."""
    need_f32_cuda(x)
    y = torch.empty_like(x)
    n = x.numel()
    check(_timed("gt_dropact_fwd", 0, 8.0 * n, lambda: lib().gt_dropact_fwd(x.data_ptr(), y.data_ptr(), n, _dref(d1), act1,
                                                                         _dref(d2), act2, stream_ptr()), shape=(n,)),
          "gt_dropact_fwd")
    return y


def dropact_bwd(x: torch.Tensor, gy: torch.Tensor, d1, act1: int, d2, act2: int) -> torch.Tensor:
    need_f32_cuda(x, gy)
    gx = torch.empty_like(x)
    n = x.numel()
    check(_timed("gt_dropact_bwd", 0, 12.0 * n, lambda: lib().gt_dropact_bwd(x.data_ptr(), gy.data_ptr(), gx.data_ptr(), n,
                                                                          _dref(d1), act1, _dref(d2), act2, stream_ptr()),
                 shape=(n,)), "gt_dropact_bwd")
    return gx


def round4(v: int) -> int:
    return (v + 3) & ~3


def headnorm_fwd(qkv: torch.Tensor, pos: Optional[torch.Tensor], gamma: Optional[torch.Tensor],
                 beta: Optional[torch.Tensor], T: int, h: int, dk: int, p: int, norm_mask: int, eps: float):
    """qkv [T,3*h*dk] -> out [3,T,h,DP], stats [2,T,h,2]."""
    need_f32_cuda(qkv, pos, gamma, beta)
    DP = round4(dk + p)
    out = torch.empty(3, T, h, DP, dtype=torch.float32, device=qkv.device)
    stats = torch.empty(2, T, h, 2, dtype=torch.float32, device=qkv.device)
    check(_timed("gt_headnorm_fwd", 0, 0, lambda: lib().gt_headnorm_fwd(qkv.data_ptr(), ptr(pos), ptr(gamma), ptr(beta), T, h, dk, p, norm_mask,
                                eps, out.data_ptr(), stats.data_ptr(), stream_ptr())), "gt_headnorm_fwd")
    return out, stats


def headnorm_bwd(d_out: torch.Tensor, qkv: torch.Tensor, gamma: Optional[torch.Tensor], stats: torch.Tensor,
                 T: int, h: int, dk: int, p: int, norm_mask: int):
    need_f32_cuda(d_out, qkv, gamma, stats)
    dev = qkv.device
    d_qkv = torch.empty(T, 3 * h * dk, dtype=torch.float32, device=dev)
    dgamma = torch.empty(2, h, dk, dtype=torch.float32, device=dev)
    dbeta = torch.empty(2, h, dk, dtype=torch.float32, device=dev)
    need = lib().gt_headnorm_bwd_ws_bytes(T, h, dk)
    ws = workspace(dev, need)
    check(_timed("gt_headnorm_bwd", 0, 0, lambda: lib().gt_headnorm_bwd(d_out.data_ptr(), qkv.data_ptr(), ptr(gamma), stats.data_ptr(), T, h, dk, p,
                                norm_mask, d_qkv.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                ws.data_ptr(), ws.numel(), stream_ptr())), "gt_headnorm_bwd")
    return d_qkv, dgamma, dbeta


def galerkin_finalize_fwd(slabs: torch.Tensor, n_slabs: int, slab_stride: int, B: int, h: int, DP: int,
                          Dr: int, d: int, n_tokens: int, mask: Optional[torch.Tensor],
                          drop: Optional[GtDropout], Wfc: torch.Tensor, value_rows_of: Optional[int] = None):
    """value_rows_of = p (coordinate columns): also return Pv [B, h (Dr - p), d], the value rows of P."""
    need_f32_cuda(slabs, mask, Wfc)
    dev = slabs.device
    Mt = torch.empty(B, h, DP, DP, dtype=torch.float32, device=dev)
    P = torch.empty(B, h * DP, d, dtype=torch.float32, device=dev)
    Pv = torch.empty(B, h * (Dr - value_rows_of), d, dtype=torch.float32, device=dev) if value_rows_of is not None else None
    dp = C.byref(drop) if (drop is not None and drop.p > 0) else None
    check(_timed("gt_galerkin_finalize_fwd", 0, 0, lambda: lib().gt_galerkin_finalize_fwd(slabs.data_ptr(), n_slabs, slab_stride, B, h, DP, Dr, d, n_tokens,
                                         ptr(mask), dp, Wfc.data_ptr(), Mt.data_ptr(), P.data_ptr(), ptr(Pv),
                                         value_rows_of or 0, stream_ptr())), "gt_galerkin_finalize_fwd")
    return (Mt, P) if value_rows_of is None else (Mt, P, Pv)


def galerkin_finalize_bwd(dPt: torch.Tensor, Mt: torch.Tensor, mask: Optional[torch.Tensor],
                          drop: Optional[GtDropout], Wfc: torch.Tensor, B: int, h: int, DP: int, Dr: int,
                          d: int, n_tokens: int):
    need_f32_cuda(dPt, Mt, mask, Wfc)
    dev = dPt.device
    dM = torch.empty(B, h, DP, DP, dtype=torch.float32, device=dev)
    dW_slabs = torch.empty(B, d, h * Dr, dtype=torch.float32, device=dev)
    dp = C.byref(drop) if (drop is not None and drop.p > 0) else None
    check(_timed("gt_galerkin_finalize_bwd", 0, 0, lambda: lib().gt_galerkin_finalize_bwd(dPt.data_ptr(), Mt.data_ptr(), ptr(mask), dp, Wfc.data_ptr(), B, h,
                                         DP, Dr, d, n_tokens, dM.data_ptr(), dW_slabs.data_ptr(),
                                         stream_ptr())), "gt_galerkin_finalize_bwd")
    return dM, dW_slabs


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    need_f32_cuda(x, gamma, beta)
    d = x.shape[-1]
    T = x.numel() // d
    y = torch.empty_like(x)
    stats = torch.empty(T, 2, dtype=torch.float32, device=x.device)
    check(_timed("gt_layernorm_fwd", 0, 0, lambda: lib().gt_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), T, d, eps, y.data_ptr(),
                                 stats.data_ptr(), stream_ptr())), "gt_layernorm_fwd")
    return y, stats


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, stats: torch.Tensor):
    need_f32_cuda(dy, x, gamma, stats)
    d = x.shape[-1]
    T = x.numel() // d
    dx = torch.empty_like(x)
    dg = torch.empty(d, dtype=torch.float32, device=x.device)
    db = torch.empty(d, dtype=torch.float32, device=x.device)
    ws = workspace(x.device, lib().gt_layernorm_bwd_ws_bytes(T, d))
    check(_timed("gt_layernorm_bwd", 0, 0, lambda: lib().gt_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), stats.data_ptr(), T, d,
                                 dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                 stream_ptr())), "gt_layernorm_bwd")
    return dx, dg, db


def modemix_fwd(X: torch.Tensor, W: torch.Tensor, Y: torch.Tensor, B: int, Q: int, Cin: int, Cout: int,
                q_total: int, q_off: int):
    """X [B,2,q_total,Cin], W [Cin,Cout,Q,2], Y [B,2,q_total,Cout] (written for q in [q_off,q_off+Q))."""
    need_f32_cuda(X, W, Y)
    check(_timed("gt_modemix_fwd", 0, 0, lambda: lib().gt_modemix_fwd(X.data_ptr(), W.data_ptr(), B, Q, Cin, Cout, 2 * q_total * Cin,
                               2 * q_total * Cout, q_total, q_total, q_off, Y.data_ptr(), stream_ptr())), "gt_modemix_fwd")
    return Y


def modemix_bwd(X: torch.Tensor, W: torch.Tensor, dY: torch.Tensor, dX: torch.Tensor, dW: torch.Tensor,
                B: int, Q: int, Cin: int, Cout: int, q_total: int, q_off: int):
    need_f32_cuda(X, W, dY, dX, dW)
    need = lib().gt_modemix_bwd_ws_bytes(B, Q, Cin, Cout)
    ws = workspace(X.device, need) if need > 0 else None
    check(_timed("gt_modemix_bwd", 0, 0, lambda: lib().gt_modemix_bwd(X.data_ptr(), W.data_ptr(), dY.data_ptr(), B, Q, Cin, Cout, 2 * q_total * Cin,
                               2 * q_total * Cout, q_total, q_total, q_off, dX.data_ptr(), dW.data_ptr(),
                               ptr(ws), ws.numel() if ws is not None else 0, stream_ptr())), "gt_modemix_bwd")
    return dX, dW


def _shape4(t: torch.Tensor, nhwc: bool):
    if nhwc:
        B, H, W, Cc = t.shape
    else:
        B, Cc, H, W = t.shape
    return B, Cc, H, W


def bilinear2d_fwd(x: torch.Tensor, size, in_nhwc: bool, out_nhwc: bool, act: int = ACT_NONE,
                   bias: Optional[torch.Tensor] = None, rp_a: Optional[torch.Tensor] = None,
                   rp_b: Optional[torch.Tensor] = None, rp_ldb: int = 0) -> torch.Tensor:
    """x dense [B,C,Hi,Wi] (or [B,Hi,Wi,C] when in_nhwc) -> dense [B,C,Ho,Wo] (or [B,Ho,Wo,C]).
    bias [C] / rp_a [B,Ho,Wo,p] / rp_b ([C,p] view with row stride rp_ldb): the affine epilogue of
    gt_bilinear2d_fwd_affine."""
    need_f32_cuda(x, bias, rp_a, rp_b)
    B, Cc, Hi, Wi = _shape4(x, in_nhwc)
    Ho, Wo = int(size[0]), int(size[1])
    y = torch.empty((B, Ho, Wo, Cc) if out_nhwc else (B, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
    nb = 4.0 * B * Cc * (Hi * Wi + Ho * Wo)
    aff = GtResizeAffine()
    aff.bias = ptr(bias)
    if rp_a is not None:
        aff.rp, aff.rp_a, aff.rp_lda = rp_a.shape[-1], rp_a.data_ptr(), rp_a.shape[-1]
        aff.rp_b, aff.rp_ldb = rp_b.data_ptr(), rp_ldb
    check(_timed("gt_bilinear2d_fwd", 0, nb, lambda: lib().gt_bilinear2d_fwd_affine(
        x.data_ptr(), y.data_ptr(), B, Cc, Hi, Wi, Ho, Wo, int(in_nhwc), int(out_nhwc), act, C.byref(aff),
        stream_ptr()),
        shape=(B, Cc, Hi, Ho, int(in_nhwc), int(out_nhwc))), "gt_bilinear2d_fwd")
    return y


def bilinear2d_bwd(g: torch.Tensor, y_saved: Optional[torch.Tensor], in_size, in_nhwc: bool, out_nhwc: bool,
                   act: int = ACT_NONE) -> torch.Tensor:
    need_f32_cuda(g, y_saved)
    B, Cc, Ho, Wo = _shape4(g, out_nhwc)
    Hi, Wi = int(in_size[0]), int(in_size[1])
    dx = torch.empty((B, Hi, Wi, Cc) if in_nhwc else (B, Cc, Hi, Wi), dtype=torch.float32, device=g.device)
    nb = 4.0 * B * Cc * (Hi * Wi + Ho * Wo * (2 if y_saved is not None else 1))
    check(_timed("gt_bilinear2d_bwd", 0, nb, lambda: lib().gt_bilinear2d_bwd(
        g.data_ptr(), ptr(y_saved), dx.data_ptr(), B, Cc, Hi, Wi, Ho, Wo, int(in_nhwc), int(out_nhwc), act,
        stream_ptr()), shape=(B, Cc, Hi, Ho, int(in_nhwc), int(out_nhwc))), "gt_bilinear2d_bwd")
    return dx


def conv3x3_resize_fwd(x: torch.Tensor, w: torch.Tensor, size, drop: Optional[GtDropout], out_nhwc: bool = False,
                       want_bits: bool = False, act: int = ACT_RELU):
    """relu(resize(relu(dropout(conv3x3(x, w, padding=1))))) -- x [B,Cin,H,W], w [Cout,Cin,3,3] -> [B,Cout,Ho,Wo], or
    channels-last [B,Ho,Wo,Cout] with ``out_nhwc``.  want_bits (channels-last, Cout % 16 == 0): also returns the byte buffer
    of the forward's decisions (gt_hip.h: relu_bits) for conv3x3_resize_bwd -> (y, bits); bits is None when not recorded."""
    need_f32_cuda(x, w)
    B, Cin, Hh, Ww = x.shape
    Cout, Ho, Wo = w.shape[0], int(size[0]), int(size[1])
    y = torch.empty((B, Ho, Wo, Cout) if out_nhwc else (B, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    dp = C.byref(drop) if (drop is not None and drop.p > 0) else None
    bits = None
    if want_bits and out_nhwc and act == ACT_RELU:           # (the SiLU form re-evaluates in its backward: nothing to record)
        nb = lib().gt_conv3x3_resize_bits_bytes(B, Cout, Ho, Wo)
        if nb > 0:
            bits = torch.empty(nb, dtype=torch.uint8, device=x.device)
    if out_nhwc:
        call = lambda: lib().gt_conv3x3_resize_fwd_nhwc(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, Cin, Cout, Hh, Ww, Ho,
                                                        Wo, dp, act, ptr(bits), stream_ptr())
    else:
        call = lambda: lib().gt_conv3x3_resize_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, Cin, Cout, Hh, Ww, Ho, Wo,
                                                   dp, act, stream_ptr())
    check(_timed("gt_conv3x3_resize_fwd", 2.0 * 36 * Cin * y.numel(), 4.0 * (x.numel() + y.numel()) + (bits.numel() if bits is not None else 0),
                 call, shape=(B, Cin, Cout, Hh, Ho, act)), "gt_conv3x3_resize_fwd")
    return (y, bits) if want_bits else y


def conv3x3_wgrad_nhwc(gy: torch.Tensor, ldg: int, x: torch.Tensor, ldx: int, B: int, Hh: int, Ww: int, Cin: int,
                       Cout: int, alpha: float = 1.0, precision: Optional[str] = None) -> torch.Tensor:
    """dw [Cout, Cin, 3, 3] of a narrow channels-last 3x3 convolution (gt_hip.h: gt_conv3x3_wgrad_nhwc).  gy / x: 2-D views
    [B*H*W, >= Cout / Cin] whose row pitch is ldg / ldx (column segments of wider buffers are read in place).  Raises
    GtNotSupported for shapes the kernel does not take (the caller owns the fallback)."""
    need_f32_cuda(gy, x)
    L = lib()
    prec = _precision[0] if precision is None else PREC_CODE[precision]
    need = L.gt_conv3x3_wgrad_nhwc_ws_bytes(B, Hh, Ww, Cin, Cout)
    if need <= 0 or prec not in (PREC_BF16X3, PREC_F16X2):
        raise GtNotSupported("gt_conv3x3_wgrad_nhwc: " + _ERR[-4])
    ws = workspace(gy.device, need)
    dw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=gy.device)
    check(_timed("gt_conv3x3_wgrad_nhwc", 18.0 * B * Hh * Ww * Cin * Cout, 4.0 * B * Hh * Ww * (Cin + Cout),
                 lambda: L.gt_conv3x3_wgrad_nhwc(gy.data_ptr(), ldg, x.data_ptr(), ldx, dw.data_ptr(), B, Hh, Ww, Cin, Cout,
                                                 float(alpha), prec, ws.data_ptr(), ws.numel(), stream_ptr()),
                 shape=(B, Hh, Ww, Cin, Cout)), "gt_conv3x3_wgrad_nhwc")
    return dw


def debug_conv0_mask(mask: Optional[torch.Tensor]):
    """Register (or, with None, remove) the byte buffer [B, Cout, H, W] in which gt_conv3x3_resize_fwd records its ReLU
    decisions (gt_hip.h: gt_debug_conv0_mask) -- parity tests replay them in the float64 checker."""
    if mask is not None and (mask.dtype != torch.uint8 or not mask.is_cuda or not mask.is_contiguous()):
        raise TypeError("debug_conv0_mask: a contiguous uint8 device tensor")
    check(lib().gt_debug_conv0_mask(ptr(mask), stream_ptr()), "gt_debug_conv0_mask")


def conv3x3_resize_bwd(g: torch.Tensor, y: Optional[torch.Tensor], x: torch.Tensor, w: torch.Tensor,
                       drop: Optional[GtDropout], out_nhwc: bool = False, bits: Optional[torch.Tensor] = None,
                       act: int = ACT_RELU) -> torch.Tensor:
    """out_nhwc: g and y are channels-last [B,Ho,Wo,Cout] (what conv3x3_resize_fwd(out_nhwc=True) returned).  bits: the
    decision buffer the forward recorded (want_bits): the backward then re-evaluates nothing and does not read y.
    act = ACT_SILU: both activations are re-evaluated from x, y is not read (may be None)."""
    need_f32_cuda(g, y, x, w)
    B, Cin, Hh, Ww = x.shape
    Cout = w.shape[0]
    Ho, Wo = (g.shape[1], g.shape[2]) if out_nhwc else (g.shape[2], g.shape[3])
    dw = torch.empty_like(w)
    ws = workspace(x.device, lib().gt_conv3x3_resize_bwd_ws_bytes(B, Cin, Cout, Hh, Ww))
    dp = C.byref(drop) if (drop is not None and drop.p > 0) else None
    if out_nhwc:
        call = lambda: lib().gt_conv3x3_resize_bwd_nhwc(g.data_ptr(), ptr(y), x.data_ptr(), w.data_ptr(), B, Cin, Cout,
                                                        Hh, Ww, Ho, Wo, dp, act, ptr(bits), dw.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), stream_ptr())
    else:
        call = lambda: lib().gt_conv3x3_resize_bwd(g.data_ptr(), ptr(y), x.data_ptr(), w.data_ptr(), B, Cin, Cout, Hh,
                                                   Ww, Ho, Wo, dp, act, dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   stream_ptr())
    reads_y = bits is None and act == ACT_RELU
    nbytes = 4.0 * (x.numel() + g.numel() * (2 if reads_y else 1)) + (bits.numel() if bits is not None else 0)
    check(_timed("gt_conv3x3_resize_bwd", 0, nbytes, call, shape=(B, Cin, Cout, Hh, Ho, act)), "gt_conv3x3_resize_bwd")
    return dw


def bilinear2d_seg_fwd(x: torch.Tensor, Cc: int, size, seg: int, segp: int, act: int = ACT_NONE, want_dact: bool = False):
    """x [B,Hi,Wi,3*segp] (the padded three-segment buffer of ops.scaler_conv_chain) -> dense [B,Ho,Wo,Cc].
    act = ACT_SILU with want_dact: returns (y, dact), dact = silu'(resized value) -- what bilinear2d_seg_bwd takes as y_saved."""
    need_f32_cuda(x)
    B, Hi, Wi, cp3 = x.shape
    assert cp3 == 3 * segp
    Ho, Wo = int(size[0]), int(size[1])
    y = torch.empty(B, Ho, Wo, Cc, dtype=torch.float32, device=x.device)
    dact = torch.empty_like(y) if (want_dact and act == ACT_SILU) else None
    check(_timed("gt_bilinear2d_seg_fwd", 0, 4.0 * B * Cc * (Hi * Wi + Ho * Wo * (2 if dact is not None else 1)),
                 lambda: lib().gt_bilinear2d_seg_fwd(x.data_ptr(), y.data_ptr(), B, Cc, Hi, Wi, Ho, Wo, act, seg, segp,
                                                     ptr(dact), stream_ptr()), shape=(B, Cc, Hi, Ho, act)),
          "gt_bilinear2d_seg_fwd")
    return (y, dact) if want_dact else y


def bilinear2d_seg_bwd(g: torch.Tensor, y_saved: Optional[torch.Tensor], in_size, seg: int, segp: int,
                       act: int = ACT_NONE, x_gate: Optional[torch.Tensor] = None, gate_mul: bool = False) -> torch.Tensor:
    """x_gate: the forward input (padded layout) when it came out of a ReLU -- dx is zeroed where it is not positive; with
    gate_mul a buffer of factors in the same layout (dx *= x_gate).  act = ACT_SILU: y_saved is the forward's dact."""
    need_f32_cuda(g, y_saved, x_gate)
    B, Ho, Wo, Cc = g.shape
    Hi, Wi = int(in_size[0]), int(in_size[1])
    dx = torch.empty(B, Hi, Wi, 3 * segp, dtype=torch.float32, device=g.device)
    check(_timed("gt_bilinear2d_seg_bwd", 0, 4.0 * B * Cc * (Hi * Wi + 2 * Ho * Wo),
                 lambda: lib().gt_bilinear2d_seg_bwd(g.data_ptr(), ptr(y_saved), dx.data_ptr(), B, Cc, Hi, Wi, Ho, Wo,
                                                     act, seg, segp, ptr(x_gate), int(bool(gate_mul)), stream_ptr()),
                 shape=(B, Cc, Hi, Ho, act)),
          "gt_bilinear2d_seg_bwd")
    return dx


def galerkin_ktv_supported(dk: int, p: int) -> bool:
    return not (dk % 16 or dk > 96 or dk // 16 == 5 or p > 2)


def galerkin_ktv(Kp: torch.Tensor, Vp: torch.Tensor, B: int, n: int, h: int, dk: int, p: int, gamma=None, beta=None):
    """K'^T V' partial slabs [n_slabs, B, h, DP, DP] from head tiles [B*n, h, DP]; None if the streaming kernel
    does not cover this head size (caller falls back to the batched GEMM).  gamma, beta [2, h, dk]: the tiles are
    "plain" (normalised values without the LayerNorm affine, gt_hip.h: hn_plain) and the affine is applied on load."""
    need_f32_cuda(Kp, Vp, gamma, beta)
    if not galerkin_ktv_supported(dk, p):
        return None
    DP = round4(dk + p)
    ns = lib().gt_galerkin_ktv_slabs(B, n)
    slabs = torch.empty(ns, B, h, DP, DP, dtype=torch.float32, device=Kp.device)
    check(_timed("gt_galerkin_ktv", 2.0 * B * h * n * DP * DP, 8.0 * B * n * h * DP,
                 lambda: lib().gt_galerkin_ktv_affine(Kp.data_ptr(), Vp.data_ptr(), ptr(gamma), ptr(beta), B, n, h, dk, p,
                                                      slabs.data_ptr(), ns, stream_ptr()),
                 shape=(B, n, h, dk, p)), "gt_galerkin_ktv")
    return slabs


# kernels that exist and pass their CPU lane-model checks but have not been measured on hardware yet are opt-in:


def galerkin_dkv(Kp, Vp, dM, dKp, dVp, B: int, n: int, h: int, DP: int):
    """dK' = V' dM^T, dV' = K' dM per (batch, head) in one streaming pass (gt_galerkin_dkv)."""
    need_f32_cuda(Kp, Vp, dM, dKp, dVp)
    check(_timed("gt_galerkin_dkv", 4.0 * B * h * n * DP * DP, 16.0 * B * n * h * DP,
                 lambda: lib().gt_galerkin_dkv(Kp.data_ptr(), Vp.data_ptr(), dM.data_ptr(), dKp.data_ptr(), dVp.data_ptr(),
                                               B, n, h, DP, stream_ptr()), shape=(B, n, h, DP)), "gt_galerkin_dkv")


def galerkin_dkv_ln_supported(dk: int, p: int, norm_mask: int) -> bool:
    """Shapes of gt_galerkin_dkv_ln: K and V normalised, head tile of 20 / 36 / 52 floats."""
    return norm_mask == 0b110 and dk % 4 == 0 and round4(dk + p) in FOURIER_DP


def galerkin_dkv_ln(Kp, Vp, dM, dQp, qkv, gamma, stats, B: int, n: int, h: int, dk: int, p: int, d_qkv=None, beta=None):
    """dK' = V' dM^T, dV' = K' dM with the per-head LayerNorm backward applied on the way out, plus the Q block: returns
    (d_qkv [B*n, 3 h dk], dgamma, dbeta [2, h, dk]) -- what galerkin_dkv + headnorm_bwd return, in one streaming pass.
    dQp=None with a caller-provided d_qkv: the Q block is already in place (written by the caller's dQ product).
    beta given: "plain" head tiles (the normalised values without the affine); qkv is then unused and may be None."""
    need_f32_cuda(Kp, Vp, dM, dQp, qkv, gamma, stats, d_qkv, beta)
    dev = Kp.device
    T = B * n
    if d_qkv is None:
        d_qkv = torch.empty(T, 3 * h * dk, dtype=torch.float32, device=dev)
    dgamma = torch.empty(2, h, dk, dtype=torch.float32, device=dev)
    dbeta = torch.empty(2, h, dk, dtype=torch.float32, device=dev)
    ws = workspace(dev, lib().gt_galerkin_dkv_ln_ws_bytes(B, h, dk))
    DP = round4(dk + p)
    check(_timed("gt_galerkin_dkv_ln", 4.0 * B * h * n * DP * DP, 4.0 * T * h * (3 * DP + 5 * dk),
                 lambda: lib().gt_galerkin_dkv_ln_plain(Kp.data_ptr(), Vp.data_ptr(), dM.data_ptr(), ptr(dQp), ptr(qkv),
                                                  gamma.data_ptr(), ptr(beta), stats.data_ptr(), B, n, h, dk, p, d_qkv.data_ptr(),
                                                  dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  stream_ptr()), shape=(B, n, h, DP)), "gt_galerkin_dkv_ln")
    return d_qkv, dgamma, dbeta


def mlp_head_supported(K: int, N: int, n_out: int) -> bool:
    """Shapes of the dedicated gt_mlp_head_* kernels (others: gt_gemm's fused-head epilogues)."""
    return K == 32 and N == 128 and n_out == 1


def head_precision(precision: Optional[str] = None) -> int:
    """Arithmetic code of a gt_mlp_head_* launch: the "head" class override, else `precision`, else the module mode."""
    if "head" in _prec_class:
        return PREC_CODE[_prec_class["head"]]
    return _precision[0] if precision is None else PREC_CODE[precision]


def mlp_head_fwd(x2, w1, b1, w2, b2, act: int, out, precision: Optional[str] = None):
    """precision f16x2: the two-term fp16 kernels; anything else: the fp32-MFMA kernels (gt_hip.h)."""
    need_f32_cuda(x2, w1, b1, w2, b2, out)
    T, K = x2.shape
    N, no = w1.shape[0], w2.shape[0]
    prec = head_precision(precision)
    check(_timed("gt_mlp_head_fwd", 2.0 * T * N * (K + no), 4.0 * T * (K + no),
                 lambda: lib().gt_mlp_head_fwd(x2.data_ptr(), T, K, N, no, w1.data_ptr(), ptr(b1), w2.data_ptr(), ptr(b2),
                                               act, prec, out.data_ptr(), stream_ptr()), shape=(T, K, N, no)),
          "gt_mlp_head_fwd")
    return out


def mlp_head_bwd(x2, w1, b1, w2, act: int, g, dx, dw1, db1, dw2, db2, precision: Optional[str] = None, dx_gate=None):
    """dx_gate [T, K]: dx *= silu'(dx_gate) on the kernel's store (gt_mlp_head_bwd_gated)."""
    need_f32_cuda(x2, w1, b1, w2, g, dx, dw1, db1, dw2, db2, dx_gate)
    T, K = x2.shape
    N, no = w1.shape[0], w2.shape[0]
    prec = head_precision(precision)
    need = lib().gt_mlp_head_bwd_ws_bytes(T)
    ws = workspace(x2.device, need)
    check(_timed("gt_mlp_head_bwd", 2.0 * T * N * (3 * K + 2 * no), 4.0 * T * ((3 if dx_gate is not None else 2) * K + no),
                 lambda: lib().gt_mlp_head_bwd_gated(x2.data_ptr(), T, K, N, no, w1.data_ptr(), ptr(b1), w2.data_ptr(), act,
                                                     prec, g.data_ptr(), ptr(dx), ptr(dx_gate), dw1.data_ptr(), ptr(db1),
                                                     ptr(dw2), ptr(db2), ws.data_ptr(), ws.numel(), stream_ptr()),
                 shape=(T, K, N, no)), "gt_mlp_head_bwd")


def dft_supported(n: int, P: int, C_: int, Co: int) -> bool:
    """Shapes gt_dft_analysis / gt_dft_synthesis implement (anything else goes through gt_gemm)."""
    return C_ == 32 and Co == 32 and P <= 32 and P % 4 == 0 and n <= 224


def dft_analysis(F, X, Y, nb: int, n: int, P: int, C_: int):
    """Y[b] (P x C) = F^T X[b] for nb grid lines (gt_dft_analysis)."""
    need_f32_cuda(F, X, Y)
    check(_timed("gt_dft_analysis", 2.0 * nb * n * P * C_, 4.0 * nb * (n + P) * C_,
                 lambda: lib().gt_dft_analysis(F.data_ptr(), X.data_ptr(), Y.data_ptr(), nb, n, P, C_, stream_ptr()),
                 shape=(nb, n, P, C_)), "gt_dft_analysis")
    return Y


def dft_synthesis(F, Z, Y, nb: int, n: int, P: int, Co: int, X2, W2, C2: int, bias=None, act: int = 0, pre=None,
                  out_gate=None):
    """Y[b] (n x Co) = act(F Z[b] + X2[b] W2 + bias) for nb grid lines (gt_dft_synthesis); out_gate [nb, n, Co]:
    Y *= silu'(out_gate) on the store (gt_dft_synthesis_gated)."""
    need_f32_cuda(F, Z, Y, X2, W2, bias, pre, out_gate)
    check(_timed("gt_dft_synthesis", 2.0 * nb * n * (P + C2) * Co,
                 4.0 * nb * (n * (C2 + Co * (2 if pre is not None or out_gate is not None else 1)) + P * Co),
                 lambda: lib().gt_dft_synthesis_gated(F.data_ptr(), Z.data_ptr(), Y.data_ptr(), nb, n, P, Co, X2.data_ptr(),
                                                      W2.data_ptr(), C2, ptr(bias), act, ptr(pre), ptr(out_gate),
                                                      stream_ptr()),
                 shape=(nb, n, P, Co, C2)), "gt_dft_synthesis")
    return Y


FOURIER_DP = (20, 36, 52)


def fourier16_active(precision: Optional[str] = None) -> bool:
    """True when the Fourier-type attention runs on the two-term fp16 kernels (gt_fourier16.hip): the default arithmetic;
    `f32` keeps the bit-exact fp32-MFMA kernel, the bf16 modes have no Fourier kernel of their own and use it too."""
    if "fourier" in _prec_class:
        return PREC_CODE[_prec_class["fourier"]] == PREC_F16X2
    return (_precision[0] if precision is None else PREC_CODE[precision]) == PREC_F16X2


def fourier16_presplit(tensors, B: int, n: int, h: int, DP: int):
    """Head tiles [B*n, h, DP] (up to four) -> their image blocks (uint8 tensors), one launch."""
    tensors = list(tensors)
    assert 1 <= len(tensors) <= 4
    need_f32_cuda(*tensors)
    nb = int(lib().gt_fourier16_image_bytes(B, n, h, DP))
    if nb <= 0:
        raise GtNotSupported(f"gt_fourier16: head tile width {DP}")
    imgs = [torch.empty(nb, dtype=torch.uint8, device=tensors[0].device) for _ in tensors]
    xs = [t.data_ptr() for t in tensors] + [None] * (4 - len(tensors))
    is_ = [t.data_ptr() for t in imgs] + [None] * (4 - len(tensors))
    check(_timed("gt_fourier16_presplit", 0.0, len(tensors) * (4.0 * B * n * h * DP + nb),
                 lambda: lib().gt_fourier16_presplit(*xs, *is_, B, n, h, DP, stream_ptr()),
                 shape=(B, n, h, DP, len(tensors))), "gt_fourier16_presplit")
    return imgs


def fourier16_block_mask(drop) -> bool:
    """The attention-score mask of the fp16 Fourier path is drawn per 4 x 4 block when p = 0.5 (the reference's always-on
    F.dropout default, layers.py:700-701) -- see gt_hip.h: gt_dropout_block16; any other p keeps one hash per element."""
    return drop is not None and drop.p == 0.5 and os.environ.get("GT_F16_BLOCK_MASK", "1") != "0"


def dropout_block16(S: torch.Tensor, BH: int, n: int, drop) -> torch.Tensor:
    """In place: S[bh, q, k] *= keep(q, k) / (1 - p) with the block mask of gt_fourier16_attn(block16 = 1)."""
    need_f32_cuda(S)
    assert S.is_contiguous() and S.numel() == BH * n * n
    check(_timed("gt_dropout_block16", 0.0, 8.0 * S.numel(),
                 lambda: lib().gt_dropout_block16(S.data_ptr(), BH, n, C.byref(drop), stream_ptr())), "gt_dropout_block16")
    return S


def fourier16_attn(F1, F2, T1, T2, B: int, n: int, h: int, DP: int, scale: float, mask, drop, owner_is_key: bool,
                   O1=None, O2=None, block16=None):
    """One pass of gt_fourier16_attn; F1, F2, T1, T2 are image blocks of fourier16_presplit.  Returns O1 (and O2).
    block16: draw the dropout mask per 4 x 4 block (default: fourier16_block_mask(drop))."""
    if block16 is None:
        block16 = mask is None and fourier16_block_mask(drop)
    dev = T1.device
    if O1 is None:
        O1 = torch.empty(B * n, h, DP, dtype=torch.float32, device=dev)
    if O2 is None and F2 is not None:
        O2 = torch.empty_like(O1)
    need_f32_cuda(mask, O1, O2)
    dp = C.byref(drop) if (drop is not None and drop.p > 0) else None
    nprod = 2 if F2 is not None else 1
    check(_timed("gt_fourier16_attn", 4.0 * nprod * B * h * n * n * DP, 4.0 * (3 + nprod) * B * n * h * DP,
                 lambda: lib().gt_fourier16_attn(F1.data_ptr(), ptr(F2), T1.data_ptr(), T2.data_ptr(), O1.data_ptr(),
                                                 ptr(O2), B, n, h, DP, scale, ptr(mask), dp, int(bool(block16)),
                                                 int(owner_is_key), stream_ptr()), shape=(B, n, h, DP, nprod)),
          "gt_fourier16_attn")
    return (O1, O2) if F2 is not None else O1


def fourier_attn(F1, F2, T1, T2, B: int, n: int, h: int, DP: int, scale: float, mask, drop, owner_is_key: bool,
                 O1=None, O2=None):
    """One pass of gt_fourier_attn on head tiles [B*n, h, DP]; returns O1 (and O2 when F2 is given).  O1 / O2 may
    be preallocated dense [B*n, h, DP] tensors (e.g. slices of the head-tile gradient buffer)."""
    need_f32_cuda(F1, F2, T1, T2, mask, O1, O2)
    if O1 is None:
        O1 = torch.empty(B * n, h, DP, dtype=torch.float32, device=F1.device)
    if O2 is None and F2 is not None:
        O2 = torch.empty_like(O1)
    dp = C.byref(drop) if (drop is not None and drop.p > 0) else None
    nprod = 2 if F2 is not None else 1
    check(_timed("gt_fourier_attn", 4.0 * nprod * B * h * n * n * DP, 4.0 * (3 + nprod) * B * n * h * DP,
                 lambda: lib().gt_fourier_attn(F1.data_ptr(), ptr(F2), T1.data_ptr(), T2.data_ptr(), O1.data_ptr(),
                                               ptr(O2), B, n, h, DP, scale, ptr(mask), dp, int(owner_is_key),
                                               stream_ptr()), shape=(B, n, h, DP, nprod)), "gt_fourier_attn")
    return (O1, O2) if F2 is not None else O1
