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


@pytest.mark.parametrize("LB,M,N", [(0, 96, 128), (1, 70, 64)])
def test_gemm_ws_lane_map(LB, M, N):
    """gemm_ws_kernel (gt_gemm.hip, staged): swizzled [row][128] LDS images, k = 16 g8 + 4 kq + c on both operands,
    row/column assignment of gemm_epilogue<1, 2> (rows mw0 + 4kq + r, columns nb + t with nb = n0 + 32 wn + 2 li)."""
    K = 128
    rng = np.random.default_rng(LB + M)
    A = rng.standard_normal((M, K))
    Bm = rng.standard_normal((N, K)) if LB == 0 else rng.standard_normal((K, N))    # B(k, n)
    Bkn = Bm.T if LB == 0 else Bm
    li, kq = S.X, S.KQ
    C = np.full((M, N), np.nan)
    for slice_ in range(N // 64):
        n0 = slice_ * 64
        sB = np.zeros(64 * 128)
        if LB == 0:
            for e in range(64 * 32):
                n, g = e >> 5, (e & 31) ^ ((e >> 5) & 7)
                sB[4 * e:4 * e + 4] = Bm[n0 + n, 4 * g:4 * g + 4]
        else:
            for e in range(64 * 128):
                k, n = e // 64, e % 64
                sB[(n * 32 + ((k >> 2) ^ (n & 7))) * 4 + (k & 3)] = Bm[k, n0 + n]
        for tile in range((M + 31) // 32):
            m0 = tile * 32
            sA = np.zeros(32 * 128)
            for e in range(1024):                         # direct-load image, zero16 beyond M
                r, g = e >> 5, (e & 31) ^ ((e >> 5) & 7)
                if m0 + r < M:
                    sA[4 * e:4 * e + 4] = A[m0 + r, 4 * g:4 * g + 4]
            for wave in range(4):
                wm, wn = wave >> 1, wave & 1
                arow = wm * 16 + li
                bcol = wn * 32 + 2 * li
                acc = [np.zeros((64, 4)), np.zeros((64, 4))]
                for g8 in range(8):
                    G = 4 * g8 + kq
                    ia = (arow * 32 + (G ^ (arow & 7))) * 4
                    ib0 = (bcol * 32 + (G ^ (bcol & 7))) * 4
                    ib1 = ((bcol + 1) * 32 + (G ^ ((bcol + 1) & 7))) * 4
                    for c in range(4):
                        acc[0] = S.mfma(sA[ia + c], sB[ib0 + c], acc[0])
                        acc[1] = S.mfma(sA[ia + c], sB[ib1 + c], acc[1])
                mw0, nb = m0 + wm * 16, n0 + wn * 32 + 2 * li
                for l in range(64):
                    for r in range(4):
                        m = mw0 + 1 * (4 * kq[l] + r) + 0
                        if m < M:
                            for t in range(2):
                                C[m, nb[l] + t] = acc[t][l, r]
    assert np.allclose(C, A @ Bkn, atol=1e-9)


def test_gemm_ws_grid_decomposition():
    """Every (slice, 32-row tile) is visited exactly once, and the slice blocks that share a y group sit on one XCD."""
    for N, M in ((384, 236672), (256, 4096), (64, 2048 + 5)):
        slices = N // 64
        per_xcd = max(1, 64 // slices)
        blocks, ygroups = 8 * per_xcd * slices, 8 * per_xcd
        mtiles = (M + 31) // 32
        seen = {}
        for L in range(blocks):
            xcd, q = L & 7, L >> 3
            slice_, ygrp = q % slices, (q // slices) * 8 + xcd
            assert ygrp < ygroups and ygrp % 8 == xcd
            for tile in range(ygrp, mtiles, ygroups):
                assert (slice_, tile) not in seen
                seen[(slice_, tile)] = L
        assert len(seen) == slices * mtiles
