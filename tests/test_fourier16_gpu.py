"""gt_fourier16_* (two-term fp16 Fourier attention, csrc/gt_fourier16.hip) against float64 restatements of
layers.py:672-705 and its backward, through the C ABI: the three uses (forward, d/dQ', the dual d/dK' + d/dV' pass), the three
mask sources (none, an explicit [B,h,n,n] mask, the stateless dropout hash -- held to the fp32-MFMA kernel, which draws the
same mask), all three head-tile widths, ragged n, and operand magnitudes far from 1."""
import math

import pytest
import torch

from _util import rel_l2

pytestmark = pytest.mark.gpu

TOL = 3e-6          # fp32-class: the fp32-MFMA kernel measures 2e-7 ... 6e-7 on the same cases


@pytest.fixture(scope="module")
def H(gpu_device):
    from galerkin_transformer import _hip
    _hip.lib()
    return _hip


def _tiles(B, n, h, DP, seed, scales=(1.0, 1.0, 1.0, 1.0), ramp=0.0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for s in scales:
        t = torch.randn(B * n, h, DP, generator=g)
        t[..., :2] = torch.rand(B * n, h, 2, generator=g)          # coordinate columns
        if DP % 16 == 4:
            t[..., -2:] = 0.0                                       # pad columns of the head tile
        if ramp:
            t = t * torch.pow(10.0, ramp * (torch.rand(B * n, 1, 1, generator=g) - 0.5))
        out.append(t * s)
    return out


def _ref(Q, K, V, dO, B, n, h, DP, scale, mask):
    q, k, v, do = (t.double().reshape(B, n, h, DP).permute(0, 2, 1, 3) for t in (Q, K, V, dO))
    S = scale * q @ k.transpose(-1, -2)
    dS = scale * do @ v.transpose(-1, -2)
    if mask is not None:
        S, dS = S * mask.double(), dS * mask.double()
    back = lambda t: t.permute(0, 2, 1, 3).reshape(B * n, h, DP)
    return back(S @ v), back(dS @ k), back(dS.transpose(-1, -2) @ q), back(S.transpose(-1, -2) @ do)


def _run16(H, Q, K, V, dO, B, n, h, DP, scale, mask, drop, block16=False):
    iq, ik, iv, ido = H.fourier16_presplit((Q, K, V, dO), B, n, h, DP)
    out = H.fourier16_attn(iq, None, ik, iv, B, n, h, DP, scale, mask, drop, False, block16=block16)
    dq = H.fourier16_attn(ido, None, iv, ik, B, n, h, DP, scale, mask, drop, False, block16=block16)
    dv, dk = H.fourier16_attn(ik, iv, iq, ido, B, n, h, DP, scale, mask, drop, True, block16=block16)
    torch.cuda.synchronize()
    return out, dq, dk, dv


def _run32(H, Q, K, V, dO, B, n, h, DP, scale, mask, drop):
    out = H.fourier_attn(Q, None, K, V, B, n, h, DP, scale, mask, drop, False)
    dq = H.fourier_attn(dO, None, V, K, B, n, h, DP, scale, mask, drop, False)
    dv, dk = H.fourier_attn(K, V, Q, dO, B, n, h, DP, scale, mask, drop, True)
    torch.cuda.synchronize()
    return out, dq, dk, dv


@pytest.mark.parametrize("B,n,h,DP", [(2, 200, 4, 36), (1, 77, 4, 20), (2, 131, 2, 52), (1, 1000, 2, 36), (3, 32, 1, 36),
                                      (1, 5, 2, 36)])
@pytest.mark.parametrize("mode", ["plain", "mask"])
def test_fourier16_matches_float64(H, gpu_device, B, n, h, DP, mode):
    dev = gpu_device
    Q, K, V, dO = (t.to(dev) for t in _tiles(B, n, h, DP, seed=n + DP))
    scale = 1.0 / math.sqrt(DP - 2) / n
    mask = None
    if mode == "mask":
        mask = (2.0 * (torch.rand(B, h, n, n, generator=torch.Generator().manual_seed(5)) < 0.5).float()).to(dev)
    ref = _ref(Q, K, V, dO, B, n, h, DP, scale, mask)
    got = _run16(H, Q, K, V, dO, B, n, h, DP, scale, mask, None)
    f32 = _run32(H, Q, K, V, dO, B, n, h, DP, scale, mask, None)
    for name, g, f, r in zip(("out", "dQ", "dK", "dV"), got, f32, ref):
        e16, e32 = rel_l2(g, r), rel_l2(f, r)
        assert e16 < TOL, (name, e16, e32)
        if DP % 16 == 4:                                          # pad columns of the outputs are exact zeros
            assert float(g[..., -2:].abs().max()) == 0.0


@pytest.mark.parametrize("B,n,h,DP", [(2, 200, 4, 36), (1, 77, 4, 20), (2, 131, 2, 52)])
@pytest.mark.parametrize("p", [0.5, 0.3])
def test_fourier16_dropout_draws_the_mask_of_the_fp32_kernel(H, gpu_device, B, n, h, DP, p):
    """Same seed and salt => the same stateless mask in both kernels (p = 0.5 runs the top-bit shortcut, 0.3 the full hash)."""
    dev = gpu_device
    Q, K, V, dO = (t.to(dev) for t in _tiles(B, n, h, DP, seed=7 * n))
    scale = 1.0 / math.sqrt(DP - 2) / n
    H.set_seed(991, dev)
    drop = H.dropout_desc(p, 11, dev)
    got = _run16(H, Q, K, V, dO, B, n, h, DP, scale, None, drop, block16=False)
    f32 = _run32(H, Q, K, V, dO, B, n, h, DP, scale, None, drop)
    for name, g, f in zip(("out", "dQ", "dK", "dV"), got, f32):
        assert rel_l2(g, f) < TOL, (name, rel_l2(g, f))


@pytest.mark.parametrize("B,n,h,DP", [(2, 200, 4, 36), (1, 77, 4, 20), (2, 131, 2, 52), (1, 5, 2, 36)])
def test_fourier16_block_mask(H, gpu_device, B, n, h, DP):
    """block16: the p = 0.5 mask per 4 x 4 block of the score matrix.  gt_dropout_block16 materialises it (applied to a matrix of
    ones); the three fused passes are held to float64 with exactly that mask; its statistics: half the entries kept, the 16
    bits of a block uncorrelated."""
    dev = gpu_device
    Q, K, V, dO = (t.to(dev) for t in _tiles(B, n, h, DP, seed=11 * n))
    scale = 1.0 / math.sqrt(DP - 2) / n
    H.set_seed(77, dev)
    drop = H.dropout_desc(0.5, 21, dev)
    mask = H.dropout_block16(torch.ones(B, h, n, n, device=dev), B * h, n, drop)
    assert set(mask.unique().tolist()) <= {0.0, 2.0}
    ref = _ref(Q, K, V, dO, B, n, h, DP, scale, mask)
    got = _run16(H, Q, K, V, dO, B, n, h, DP, scale, None, drop, block16=True)
    for name, g, r in zip(("out", "dQ", "dK", "dV"), got, ref):
        assert rel_l2(g, r) < TOL, (name, rel_l2(g, r))
    if n >= 128:
        keep = (mask > 0).float()
        assert abs(float(keep.mean()) - 0.5) < 4.0 / math.sqrt(keep.numel())
        n4 = n // 4 * 4
        blocks = keep[..., :n4, :n4].reshape(B * h, n4 // 4, 4, n4 // 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 16)
        c = torch.corrcoef(blocks.T.double())
        off = c - torch.eye(16, device=dev, dtype=torch.float64)
        assert float(off.abs().max()) < 6.0 / math.sqrt(blocks.shape[0])


@pytest.mark.parametrize("scales", [(1e-6, 1e5, 1e-3, 1e8), (1e4, 1e4, 1e4, 1e4), (1e-12, 1e-12, 1e-12, 1e-12)])
def test_fourier16_operand_magnitudes(H, gpu_device, scales):
    """Per-tile exponents + the running accumulator exponent: tensors far from unit scale, token rows spread over three
    decades inside every tile."""
    dev = gpu_device
    B, n, h, DP = 1, 333, 2, 36
    Q, K, V, dO = (t.to(dev) for t in _tiles(B, n, h, DP, seed=3, scales=scales, ramp=3.0))
    scale = 1.0 / n
    ref = _ref(Q, K, V, dO, B, n, h, DP, scale, None)
    got = _run16(H, Q, K, V, dO, B, n, h, DP, scale, None, None)
    for name, g, r in zip(("out", "dQ", "dK", "dV"), got, ref):
        assert torch.isfinite(g).all()
        assert rel_l2(g, r) < TOL, (name, rel_l2(g, r))


def test_fourier16_time_vs_fp32_kernel(H, gpu_device):
    """Informational: C3's layer shape (B = 8), both kernels, all three passes; printed, and the fp16 one must not be
    slower."""
    dev = gpu_device
    B, n, h, DP = 8, 3721, 4, 36
    Q, K, V, dO = (t.to(dev) for t in _tiles(B, n, h, DP, seed=1))
    scale = 1.0 / math.sqrt(34) / n
    H.set_seed(5, dev)
    drop = H.dropout_desc(0.5, 3, dev)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5

    timed(lambda: _run32(H, Q, K, V, dO, B, n, h, DP, scale, None, drop))          # clocks up
    t16 = timed(lambda: _run16(H, Q, K, V, dO, B, n, h, DP, scale, None, drop, block16=True))
    t32 = timed(lambda: _run32(H, Q, K, V, dO, B, n, h, DP, scale, None, drop))
    imgs = H.fourier16_presplit((Q, K, V, dO), B, n, h, DP)
    tf = timed(lambda: H.fourier16_attn(imgs[0], None, imgs[1], imgs[2], B, n, h, DP, scale, None, drop, False))
    td = timed(lambda: H.fourier16_attn(imgs[1], imgs[2], imgs[0], imgs[3], B, n, h, DP, scale, None, drop, True))
    tfe = timed(lambda: H.fourier16_attn(imgs[0], None, imgs[1], imgs[2], B, n, h, DP, scale, None, drop, False, block16=False))
    tde = timed(lambda: H.fourier16_attn(imgs[1], imgs[2], imgs[0], imgs[3], B, n, h, DP, scale, None, drop, True, block16=False))
    print(f"\nper-element hash (p = 0.5, top-bit shortcut): fwd {tfe:.3f}, dual {tde:.3f}")
    tp = timed(lambda: H.fourier16_presplit((Q, K, V, dO), B, n, h, DP))
    tfp = timed(lambda: H.fourier16_attn(imgs[0], None, imgs[1], imgs[2], B, n, h, DP, scale, None, None, False))
    tdp = timed(lambda: H.fourier16_attn(imgs[1], imgs[2], imgs[0], imgs[3], B, n, h, DP, scale, None, None, True))
    d3 = H.dropout_desc(0.3, 3, dev)
    tf3 = timed(lambda: H.fourier16_attn(imgs[0], None, imgs[1], imgs[2], B, n, h, DP, scale, None, d3, False))
    print(f"\nwithout dropout: fwd {tfp:.3f}, dual {tdp:.3f}; p = 0.3 (full hash + compare): fwd {tf3:.3f}")
    print(f"\nfourier B={B} n={n}: f16x2 three passes {t16:.3f} ms (fwd {tf:.3f}, dual {td:.3f}, presplit x4 {tp:.3f}) vs fp32 MFMA {t32:.3f} ms")
    assert t16 < t32
