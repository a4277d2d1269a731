"""CPU checks of MFMA operand / result index maps with the lane-accurate numpy model in tools/mfma_sim.py.
Each test restates a kernel's per-lane loads, k-step enumeration and stores exactly as the HIP source does and
compares the assembled result with a plain matmul -- the arithmetic of the layout is pinned before (and
independently of) any GPU run."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import mfma_sim as S  # noqa: E402

J, KQ = S.X, S.KQ


def test_mfma_model_selftest():
    S.selftest()


@pytest.mark.parametrize("G,n", [(1, 21), (2, 37), (3, 16)])
def test_galerkin_dkv_lane_map(G, n):
    """galerkin_dkv_kernel<G> (gt_ops.hip): dK' = V' dM^T, dV' = K' dM for one (batch, head)."""
    DP, NS, NMT = 16 * G + 4, 4 * G + 1, G + 1
    rng = np.random.default_rng(G)
    K, V, dM = rng.standard_normal((n, DP)), rng.standard_normal((n, DP)), rng.standard_normal((DP, DP))
    dK, dV = np.full((n, DP), np.nan), np.full((n, DP), np.nan)

    def kidx(s):                                   # contraction index of k-step s, per lane
        return 4 * (KQ + 4 * (s >> 2)) + (s & 3) if s < 4 * G else 16 * G + KQ

    a1 = np.zeros((NMT, NS, 64)); a2 = np.zeros((NMT, NS, 64))
    for mt in range(NMT):
        col = 16 * mt + J
        cc = np.minimum(col, DP - 1)
        live = (col < DP).astype(float)
        for s in range(NS):
            k = kidx(s)
            a1[mt, s] = live * dM[cc, k]
            a2[mt, s] = live * dM[k, cc]
    for tile in range((n + 15) // 16):
        t = 16 * tile + J
        tc = np.minimum(t, n - 1)
        kk = [K[tc[:, None], (4 * (KQ + 4 * g))[:, None] + np.arange(4)] for g in range(G)] + [K[tc, 16 * G:16 * G + 4]]
        vv = [V[tc[:, None], (4 * (KQ + 4 * g))[:, None] + np.arange(4)] for g in range(G)] + [V[tc, 16 * G:16 * G + 4]]
        acc1 = [np.zeros((64, 4)) for _ in range(NMT)]
        acc2 = [np.zeros((64, 4)) for _ in range(NMT)]
        for s in range(NS):
            if s < 4 * G:
                bv, bk = vv[s >> 2][:, s & 3], kk[s >> 2][:, s & 3]
            else:
                bv, bk = vv[G][S.LANES, KQ], kk[G][S.LANES, KQ]
            for mt in range(NMT):
                acc1[mt] = S.mfma(a1[mt, s], bv, acc1[mt])
                acc2[mt] = S.mfma(a2[mt, s], bk, acc2[mt])
        for l in range(64):
            if t[l] < n:
                for mt in range(NMT):
                    col = 16 * mt + 4 * KQ[l]
                    if col < DP:
                        dK[t[l], col:col + 4] = acc1[mt][l]
                        dV[t[l], col:col + 4] = acc2[mt][l]
    assert np.allclose(dK, V @ dM.T, atol=1e-10)
    assert np.allclose(dV, K @ dM, atol=1e-10)


@pytest.mark.parametrize("h,dk,pp,mask", [(4, 32, 2, 0b110), (2, 64, 1, 0b011), (8, 16, 2, 0b110), (4, 32, 0, 0b000)])
def test_headnorm_epilogue_lane_map(h, dk, pp, mask):
    """headnorm_scatter (gt_gemm.hip, GT_EP_HEADNORM): the epilogue lane (li, kq) of a 64x64 wave tile holds rows
    mw0 + MT*(4kq + r) + s and the 4 columns nb .. nb+3; head statistics are xor-shuffles inside dk/4-lane groups."""
    MT, NT, eps = 4, 4, 1e-7
    N, DP = 3 * h * dk, (dk + pp + 3) & ~3
    M = 64 * 2
    rng = np.random.default_rng(h * dk)
    V = rng.standard_normal((M, N))
    gamma, beta = rng.standard_normal((3, h, dk)), rng.standard_normal((3, h, dk))
    pos = rng.standard_normal((M, max(pp, 1)))
    out = np.full((3, M, h, DP), np.nan)
    stats = np.full((3, M, h, 2), np.nan)
    li, kq = S.X, S.KQ
    G = dk // 4
    for mw0 in range(0, M, 16 * MT):                  # wave tiles: 64 rows x 64 columns
        for n0 in range(0, N, 16 * NT):
            nb = n0 + NT * li                          # per lane
            for s in range(MT):
                for r in range(4):
                    m = mw0 + MT * (4 * kq + r) + s    # per lane
                    v = V[m[:, None], nb[:, None] + np.arange(4)]            # [64, 4]
                    stream, head, dim = nb // (h * dk), (nb // dk) % h, nb % dk
                    normed = (mask >> stream) & 1
                    ni = np.array([bin(mask & ((1 << st) - 1)).count("1") for st in stream])
                    y = v.copy()
                    tot = v.sum(1)
                    o = G >> 1
                    while o > 0:
                        tot = tot + tot[S.LANES ^ o]
                        o >>= 1
                    mu = tot / dk
                    c = v - mu[:, None]
                    ss = (c * c).sum(1)
                    o = G >> 1
                    while o > 0:
                        ss = ss + ss[S.LANES ^ o]
                        o >>= 1
                    rstd = 1.0 / np.sqrt(ss / dk + eps)
                    for l in range(64):
                        if normed[l]:
                            gm = gamma[ni[l], head[l], dim[l]:dim[l] + 4]
                            bt = beta[ni[l], head[l], dim[l]:dim[l] + 4]
                            y[l] = c[l] * rstd[l] * gm + bt
                            if dim[l] == 0:
                                stats[ni[l], m[l], head[l]] = (mu[l], rstd[l])
                        out[stream[l], m[l], head[l], pp + dim[l]:pp + dim[l] + 4] = y[l]
                        if dim[l] == 0:
                            out[stream[l], m[l], head[l], :pp] = pos[m[l], :pp]
                        if dim[l] == dk - 4:
                            out[stream[l], m[l], head[l], pp + dk:] = 0.0
    # reference: per-head LayerNorm of the packed projection, coordinates in front, zero pad
    X = V.reshape(M, 3, h, dk)
    ref = np.zeros((3, M, h, DP))
    nidx = 0
    for st in range(3):
        x = X[:, st]
        if (mask >> st) & 1:
            mu = x.mean(-1, keepdims=True)
            var = ((x - mu) ** 2).mean(-1, keepdims=True)
            x = (x - mu) / np.sqrt(var + eps) * gamma[nidx] + beta[nidx]
            assert np.allclose(stats[nidx, :, :, 0], mu[..., 0]) and np.allclose(stats[nidx, :, :, 1], 1 / np.sqrt(var[..., 0] + eps))
            nidx += 1
        ref[st, :, :, pp:pp + dk] = x
        ref[st, :, :, :pp] = pos[:, None, :pp]
    assert not np.isnan(out).any()
    assert np.allclose(out, ref, atol=1e-10)
