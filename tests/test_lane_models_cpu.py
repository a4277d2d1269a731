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


@pytest.mark.parametrize("MP,N,K", [(1, 2, 37), (1, 32, 64), (4, 32, 23)])
def test_tsmm_lane_map(MP, N, K):
    """tsmm_kernel (gt_tsmm.hip): one float2 per lane feeds an even-column and an odd-column tile."""
    M = 32 * MP
    rng = np.random.default_rng(MP + N)
    A, B = rng.standard_normal((K, M)), rng.standard_normal((K, N))
    i, kq = S.X, S.KQ
    acc = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(2 * MP)]
    cs = np.zeros((MP, 2, 64))
    for k in range(0, K, 4):
        ok = k + kq < K
        row = np.minimum(k + kq, K - 1)
        b2 = np.zeros((64, 2))
        for f in range(2):
            col = 2 * i + f
            b2[:, f] = np.where(ok & (2 * i < N), B[row, np.minimum(col, N - 1)], 0.0)
        for mp in range(MP):
            a2 = np.stack([np.where(ok, A[row, 32 * mp + 2 * i + e], 0.0) for e in range(2)], 1)
            cs[mp] += a2.T
            for e in range(2):
                for f in range(2):
                    acc[2 * mp + e][f] = S.mfma(a2[:, e], b2[:, f], acc[2 * mp + e][f])
    C = np.full((M, N), np.nan)
    colsum = np.zeros(M)
    for l in range(64):
        for mp in range(MP):
            for e in range(2):
                if kq[l] == 0:
                    colsum[32 * mp + 2 * i[l] + e] = sum(cs[mp, e, i[l] + 16 * q] for q in range(4))
                for f in range(2):
                    for r in range(4):
                        m, n = 32 * mp + 2 * (4 * kq[l] + r) + e, 2 * i[l] + f
                        if n < N:
                            C[m, n] = acc[2 * mp + e][f][l, r]
    assert np.allclose(C, A.T @ B, atol=1e-10)
    assert np.allclose(colsum, A.sum(0), atol=1e-10)


@pytest.mark.parametrize("T", [16, 29])
def test_head_backward_lane_map(T):
    """head_bwd_kernel (gt_head.hip), one 16-row tile of one wave pair: H^T = W1 X^T with k = 8kq + s, the
    register-s trick for dX^T = W1^T dh^T, the LDS-staged dh for dW1 += dh^T X, and the pair's dX exchange."""
    HK, HN = 32, 128
    rng = np.random.default_rng(T)
    X, W1 = rng.standard_normal((T, HK)), rng.standard_normal((HN, HK)) * 0.3
    b1, w2, g = rng.standard_normal(HN), rng.standard_normal(HN), rng.standard_normal(T)
    j, kq = S.X, S.KQ
    dX = np.full((T, HK), np.nan)
    dW1, dw2, db1 = np.zeros((HN, HK)), np.zeros(HN), np.zeros(HN)
    sig = lambda v: 1 / (1 + np.exp(-v))
    for m0 in range(0, T, 16):
        row = np.minimum(m0 + j, T - 1)
        gv = np.where(m0 + j < T, g[row], 0.0)
        xs = X[row[:, None], (8 * kq)[:, None] + np.arange(8)]                  # the lane's two float4
        partial = {}
        for half in range(2):                                                    # the two waves of the pair
            hb = 64 * half
            acc = []
            for mt in range(4):
                a = np.zeros((64, 4))
                for s in range(8):
                    a = S.mfma(W1[hb + 16 * mt + j, 8 * kq + s], xs[:, s], a)
                acc.append(a)
            dh_lds = np.zeros((16, 64))
            for mt in range(4):
                hid = hb + 16 * mt + 4 * kq[:, None] + np.arange(4)             # [64, 4]
                hpre = acc[mt] + b1[hid]
                sg = sig(hpre)
                a_, da = hpre * sg, sg * (1 + hpre * (1 - sg))
                d = gv[:, None] * w2[hid] * da
                for l in range(64):
                    dw2[hid[l]] += gv[l] * a_[l]
                    db1[hid[l]] += d[l]
                    dh_lds[j[l], 16 * mt + 4 * kq[l]:16 * mt + 4 * kq[l] + 4] = d[l]
                acc[mt] = d
            accX = [np.zeros((64, 4)), np.zeros((64, 4))]
            for mt in range(4):
                for s in range(4):
                    for t in range(2):
                        accX[t] = S.mfma(W1[hb + 16 * mt + 4 * kq + s, 16 * t + j], acc[mt][:, s], accX[t])
            partial[half] = accX
            accW = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(4)]
            for mt in range(4):
                for s in range(4):
                    a = dh_lds[4 * kq + s, 16 * mt + j]
                    for t in range(2):
                        xr = X[np.minimum(m0 + 4 * kq + s, T - 1), 16 * t + j]
                        accW[mt][t] = S.mfma(a, xr, accW[mt][t])
            for l in range(64):
                for mt in range(4):
                    for t in range(2):
                        for r in range(4):
                            dW1[hb + 16 * mt + 4 * kq[l] + r, 16 * t + j[l]] += accW[mt][t][l, r]
        for half in range(2):                                                    # each wave stores its own t = half
            tot = partial[half][half] + partial[half ^ 1][half]
            for l in range(64):
                if m0 + j[l] < T:
                    dX[m0 + j[l], 16 * half + 4 * kq[l]:16 * half + 4 * kq[l] + 4] = tot[l]
    # the row lanes hold duplicates of dw2 / db1 contributions per kq group: the kernel sums over the 16 j lanes of
    # ONE kq group per hidden value; here every lane added its own row's term exactly once, which is the same sum
    H = X @ W1.T + b1
    sg = sig(H)
    dh = g[:, None] * w2[None, :] * (sg * (1 + H * (1 - sg)))
    assert np.allclose(dX, dh @ W1, atol=1e-9)
    assert np.allclose(dW1, dh.T @ X, atol=1e-9)
    assert np.allclose(db1, dh.sum(0), atol=1e-9)
    assert np.allclose(dw2, (g[:, None] * H * sg).sum(0), atol=1e-9)


@pytest.mark.parametrize("KS,n", [(9, 50), (5, 64)])
def test_fourier_core_lane_map(KS, n):
    """fourier_core_kernel (gt_fourier.hip), one wave, one 64-row stream tile: S^T tile = T1 F1^T, then the second
    product reads register s of the score tile as its k-step s (stream row 16mt + 4kq + s), last 4 columns on
    the VALU with a kq reduction."""
    DP, NF, XC = 4 * KS, (4 * KS - 4) // 16, 4 * KS - 4
    rng = np.random.default_rng(KS)
    F1 = rng.standard_normal((32, DP))                 # 32 owner rows of the wave
    T1, T2 = rng.standard_normal((n, DP)), rng.standard_normal((n, DP))
    j, kq = S.X, S.KQ
    f1 = [[F1[16 * nt + j, 4 * s + kq] for s in range(KS)] for nt in range(2)]
    acc1 = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(NF)]
    ax = np.zeros((2, 64, 4))
    for s0 in range(0, n, 64):
        t1 = np.zeros((64, DP)); t2 = np.zeros((64, DP))
        rows = min(64, n - s0)
        t1[:rows], t2[:rows] = T1[s0:s0 + rows], T2[s0:s0 + rows]
        sa = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(4)]
        for s in range(KS):
            for mt in range(4):
                a1 = t1[16 * mt + j, 4 * s + kq]
                for nt in range(2):
                    sa[mt][nt] = S.mfma(a1, f1[nt][s], sa[mt][nt])
        for mt in range(4):
            for s in range(4):
                row = 16 * mt + 4 * kq + s
                for dt in range(NF):
                    a = t2[row, 16 * dt + j]
                    for nt in range(2):
                        acc1[dt][nt] = S.mfma(a, sa[mt][nt][:, s], acc1[dt][nt])
                x1 = t2[row, XC:XC + 4]
                for nt in range(2):
                    ax[nt] += sa[mt][nt][:, s][:, None] * x1
    O = np.full((32, DP), np.nan)
    for l in range(64):
        for nt in range(2):
            ow = 16 * nt + j[l]
            for dt in range(NF):
                O[ow, 16 * dt + 4 * kq[l]:16 * dt + 4 * kq[l] + 4] = acc1[dt][nt][l]
            if kq[l] == 0:
                O[ow, XC:XC + 4] = sum(ax[nt][j[l] + 16 * q] for q in range(4))
    assert np.allclose(O, (F1 @ T1.T) @ T2, atol=1e-9)


@pytest.mark.parametrize("n,P", [(141, 24), (30, 8)])
def test_dft_analysis_lane_map(n, P):
    """dft_analysis_kernel (gt_dft.hip): the line slab lands in LDS with granule q of row r holding logical granule
    q ^ (((r >> 1) & 1) << 2); wave (mt, nt) computes one 16 x 16 tile of Y = F^T X with two accumulators."""
    C = 32
    KMAX = 16 if n <= 64 else (36 if n <= 144 else 56)
    rng = np.random.default_rng(n)
    F, X = rng.standard_normal((n, P)), rng.standard_normal((n, C))
    nrow = 4 * KMAX + 16
    lds = np.zeros((nrow, C))
    for r in range(n):                                   # direct-load image: linear granule e = 8r + q
        for q in range(8):
            g = q ^ (((r >> 1) & 1) << 2)
            lds[r, 4 * q:4 * q + 4] = X[r, 4 * g:4 * g + 4]
    flat = lds.reshape(-1)
    j, kq = S.X, S.KQ
    Y = np.full((P, C), np.nan)
    for wave in range(4):
        mt, nt = wave >> 1, wave & 1
        pr = np.minimum(16 * mt + j, P - 1)
        fa = [np.where(4 * s + kq < n, F[np.minimum(4 * s + kq, n - 1), pr], 0.0) for s in range(KMAX)]
        slot = (4 * nt + (j >> 2)) ^ (((kq >> 1) & 1) << 2)
        boff = kq * C + slot * 4 + (j & 3)
        acc = [np.zeros((64, 4)), np.zeros((64, 4))]
        for s in range(KMAX):
            acc[s & 1] = S.mfma(fa[s], flat[boff + 4 * s * C], acc[s & 1])
        for l in range(64):
            for r in range(4):
                prr = min(16 * mt + 4 * kq[l] + r, P - 1)
                Y[prr, 16 * nt + j[l]] = acc[0][l, r] + acc[1][l, r]
    assert np.allclose(Y, F.T @ X, atol=1e-9)
