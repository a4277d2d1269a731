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


@pytest.mark.parametrize("G,n,p", [(2, 37, 2), (1, 21, 2), (2, 40, 1), (3, 18, 2)])
def test_galerkin_dkv_ln_plain_lane_map(G, n, p):
    """galerkin_dkv_ln_kernel<G, PLAIN = true> (gt_ops.hip) for one (batch, head): output columns in value order (a row
    permutation of the dM fragments), gamma folded into the fragment rows, beta dM as the accumulators' start value, the
    LayerNorm backward on the lanes that hold the products (row means over the four kq lanes of a token), d(gamma) /
    d(beta) folded over the token lanes -- against the plain numpy statement of the same backward."""
    DP, NS, NMT = 16 * G + 4, 4 * G + 1, G + 1
    dk = DP - 4 if p else DP                         # round4(dk + p) == DP with dk % 4 == 0
    rng = np.random.default_rng(10 * G + p)
    xh = [rng.standard_normal((n, dk)) for _ in range(2)]            # normalised K and V values
    pos = rng.standard_normal((n, p))
    tile = [np.zeros((n, DP)) for _ in range(2)]                     # plain head tiles [pos | xh | pad]
    for s_ in range(2):
        tile[s_][:, :p] = pos
        tile[s_][:, p:p + dk] = xh[s_]
    gam, bet = 1 + 0.3 * rng.standard_normal((2, dk)), 0.2 * rng.standard_normal((2, dk))
    rstd = 0.5 + rng.random((2, n))
    dM = rng.standard_normal((DP, DP))

    # ---- reference
    aff = [t.copy() for t in tile]
    for s_ in range(2):
        aff[s_][:, p:p + dk] = xh[s_] * gam[s_] + bet[s_]
    gyK, gyV = (aff[1] @ dM.T)[:, p:p + dk], (aff[0] @ dM)[:, p:p + dk]
    ref_dx, ref_dg, ref_db = [], [], []
    for s_, gy in enumerate((gyK, gyV)):
        gg = gy * gam[s_]
        m1, m2 = gg.mean(1, keepdims=True), (gg * xh[s_]).mean(1, keepdims=True)
        ref_dx.append(rstd[s_][:, None] * (gg - m1 - xh[s_] * m2))
        ref_dg.append((gy * xh[s_]).sum(0))
        ref_db.append(gy.sum(0))

    # ---- lane model
    def kidx(s):
        return 4 * (KQ + 4 * (s >> 2)) + (s & 3) if s < 4 * G else 16 * G + KQ

    a1 = np.zeros((NMT, NS, 64)); a2 = np.zeros((NMT, NS, 64))
    for mt in range(NMT):
        cp = 16 * mt + J
        col = np.where(cp < dk, cp + p, np.where(cp < dk + p, cp - dk, cp))
        cc = np.minimum(col, DP - 1)
        live = (cp < DP).astype(float)
        for s in range(NS):
            k = kidx(s)
            isval = (k >= p) & (k < p + dk)
            sv = np.where(isval, gam[1][np.clip(k - p, 0, dk - 1)], 1.0)
            sk = np.where(isval, gam[0][np.clip(k - p, 0, dk - 1)], 1.0)
            a1[mt, s] = live * sv * dM[cc, k]
            a2[mt, s] = live * sk * dM[k, cc]
    cst = np.zeros((2, 4, NMT, 4))
    for kq2 in range(4):
        for mt in range(NMT):
            for c in range(4):
                cp = 16 * mt + 4 * kq2 + c
                if cp >= DP:
                    continue
                col = cp + p if cp < dk else (cp - dk if cp < dk + p else cp)
                cst[0, kq2, mt, c] = sum(bet[1][v] * dM[col, p + v] for v in range(dk))
                cst[1, kq2, mt, c] = sum(bet[0][v] * dM[p + v, col] for v in range(dk))
    ok = [(16 * mt + 4 * KQ) < dk for mt in range(NMT)]
    dx = [np.full((n, dk), np.nan) for _ in range(2)]
    dg = [[np.zeros((64, 4)) for _ in range(NMT)] for _ in range(2)]
    db = [[np.zeros((64, 4)) for _ in range(NMT)] for _ in range(2)]
    for tl in range((n + 15) // 16):
        t = 16 * tl + J
        tc = np.minimum(t, n - 1)
        kk = [tile[0][tc[:, None], (4 * (KQ + 4 * g))[:, None] + np.arange(4)] for g in range(G)] + [tile[0][tc, 16 * G:16 * G + 4]]
        vv = [tile[1][tc[:, None], (4 * (KQ + 4 * g))[:, None] + np.arange(4)] for g in range(G)] + [tile[1][tc, 16 * G:16 * G + 4]]
        acc = [[cst[w][KQ, mt].copy() for mt in range(NMT)] for w in range(2)]
        for s in range(NS):
            if s < 4 * G:
                bv, bk = vv[s >> 2][:, s & 3], kk[s >> 2][:, s & 3]
            else:
                bv, bk = vv[G][S.LANES, KQ], kk[G][S.LANES, KQ]
            for mt in range(NMT):
                acc[0][mt] = S.mfma(a1[mt, s], bv, acc[0][mt])
                acc[1][mt] = S.mfma(a2[mt, s], bk, acc[1][mt])
        live_t = t < n
        for s_ in range(2):
            x = [None] * NMT
            for mt in range(NMT):                  # tile_load4(row + p + 4 kq + 16 mt): the value-ordered float4 of the tile row
                cols = np.minimum(p + 16 * mt + 4 * KQ[:, None] + np.arange(4), DP - 1)
                x[mt] = np.where(ok[mt][:, None], tile[s_][tc[:, None], cols], 0.0)
            gm = [np.where(ok[mt][:, None], gam[s_][np.minimum(16 * mt + 4 * KQ[:, None] + np.arange(4), dk - 1)], 0.0) for mt in range(NMT)]
            gg = [np.where(ok[mt][:, None], acc[s_][mt] * gm[mt], 0.0) for mt in range(NMT)]
            s1 = sum(g_.sum(1) for g_ in gg)
            s2 = sum((g_ * x_).sum(1) for g_, x_ in zip(gg, x))
            for sh in (16, 32):                     # the two cross-lane adds over the kq lanes of a token
                s1 = s1 + s1[S.LANES ^ sh]
                s2 = s2 + s2[S.LANES ^ sh]
            m1, m2 = s1 / dk, s2 / dk
            for l in range(64):
                if not live_t[l]:
                    continue
                for mt in range(NMT):
                    if not ok[mt][l]:
                        continue
                    v0 = 16 * mt + 4 * KQ[l]
                    dx[s_][t[l], v0:v0 + 4] = rstd[s_][t[l]] * (gg[mt][l] - m1[l] - x[mt][l] * m2[l])
                    dg[s_][mt][l] += acc[s_][mt][l] * x[mt][l]
                    db[s_][mt][l] += acc[s_][mt][l]
    for s_ in range(2):
        assert np.allclose(dx[s_], ref_dx[s_], atol=1e-9)
        got_dg, got_db = np.zeros(dk), np.zeros(dk)
        for mt in range(NMT):
            for l in range(64):
                v0 = 16 * mt + 4 * KQ[l]
                if v0 < dk:
                    got_dg[v0:v0 + 4] += dg[s_][mt][l]
                    got_db[v0:v0 + 4] += db[s_][mt][l]
        assert np.allclose(got_dg, ref_dg[s_], atol=1e-9) and np.allclose(got_db, ref_db[s_], atol=1e-9)


def _mfma32(first, second, acc):
    """Lane model of v_mfma_f32_32x32x16_bf16 (values kept in float64): D[32][32] += X[32][16] Y[16][32] with
    lane l supplying row l & 31 of X / column l & 31 of Y for k = 8 (l >> 5) .. + 7, and holding
    D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31] in accumulator register r."""
    lanes = np.arange(64)
    lr, lh = lanes & 31, lanes >> 5
    X = np.zeros((32, 16)); Y = np.zeros((16, 32))
    for e in range(8):
        X[lr, 8 * lh + e] = first[:, e]
        Y[8 * lh + e, lr] = second[:, e]
    D = X @ Y
    out = acc.copy()
    for r in range(16):
        out[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * lh, lr]
    return out


@pytest.mark.parametrize("conv", [False, True])
def test_packed_b_kernel_lane_map(conv):
    """gemm_x3p_kernel (gt_gemm_x3.hip), one 128 x 128 block tile: x3_pack_b_kernel's fragment order, the wave's B
    fragment addresses, the A rows of a stage (plain, or the implicit 3x3 convolution: channel block -> tap -> channel
    stage order, tap-valid bits, neighbour-row offsets) and the accumulator -> (row, column) map of the epilogue,
    against A B^T / conv2d restated in numpy.  Values stay float64: the bf16 split is exact and tested on the GPU."""
    rng = np.random.default_rng(int(conv))
    if conv:
        Bn, Hh, Ww, Cc, N = 1, 9, 13, 32, 128           # 117 pixels: one partial tile; 32 channels: CB = 32, two stages per tap
        M, K = Bn * Hh * Ww, 9 * Cc
        img = rng.standard_normal((Bn, Hh, Ww, Cc))
        A = img.reshape(M, Cc)
        cb = 32 if Cc % 32 == 0 else 16
        w = rng.standard_normal((N, Cc, 3, 3))
        Bmat = w.transpose(0, 2, 3, 1).reshape(N, 9, Cc // cb, cb).transpose(0, 2, 1, 3).reshape(N, K)    # ops._conv_k_order
    else:
        M, N, K = 100, 72, 40                            # partial M tile, N that does not fill the tile, partial last stage
        A = rng.standard_normal((M, K))
        Bmat = rng.standard_normal((N, K))
    NT, KS = ((N + 127) // 128) * 4, (K + 15) // 16
    lanes = np.arange(64)
    lr, lh = lanes & 31, lanes >> 5
    # x3_pack_b_kernel (one plane)
    packed = np.zeros((NT, KS, 64, 8))
    for nt in range(NT):
        for ks in range(KS):
            n = nt * 32 + lr
            for e in range(8):
                k = ks * 16 + 8 * lh + e
                okb = (n < N) & (k < K)
                packed[nt, ks, :, e] = np.where(okb, Bmat[np.minimum(n, N - 1), np.minimum(k, K - 1)], 0.0)
    C = np.full((M, N), np.nan)
    m0 = n0 = 0
    for wave in range(4):
        wm, wn = wave >> 1, wave & 1
        acc = [[np.zeros((64, 16)) for _ in range(2)] for _ in range(2)]
        tap, c0 = 0, 0
        for ks in range(KS):
            am = []
            for i in range(2):
                row = wm * 64 + 32 * i + lr
                m = m0 + row
                v = np.zeros((64, 8))
                if conv:
                    pix = np.minimum(m, M - 1) % (Hh * Ww)
                    y, x = pix // Ww, pix % Ww
                    dy, dx = tap // 3 - 1, tap % 3 - 1
                    valid = (m < M) & (y + dy >= 0) & (y + dy < Hh) & (x + dx >= 0) & (x + dx < Ww)
                    src = np.clip(m + dy * Ww + dx, 0, M - 1)
                    for e in range(8):
                        v[:, e] = np.where(valid, A[src, c0 + 8 * lh + e], 0.0)
                else:
                    for e in range(8):
                        k = ks * 16 + 8 * lh + e
                        v[:, e] = np.where((m < M) & (k < K), A[np.minimum(m, M - 1), np.minimum(k, K - 1)], 0.0)
                am.append(v)
            if conv:                                   # the issue state machine: channel block first, taps second
                c0 += 16
                if c0 % cb == 0:
                    c0 -= cb
                    tap += 1
                    if tap == 9:
                        tap, c0 = 0, c0 + cb
            nt0 = (n0 + wn * 64) >> 5
            for j in range(2):
                bn = packed[nt0 + j, ks]
                for i in range(2):
                    acc[i][j] = _mfma32(bn, am[i], acc[i][j])
        for i in range(2):                             # x3_epilogue: register 4 g + t of accumulator (i, j)
            for j in range(2):
                for g in range(4):
                    for t in range(4):
                        mrow = m0 + wm * 64 + 32 * i + lr
                        ncol = n0 + wn * 64 + 32 * j + 8 * g + 4 * lh + t
                        okc = (mrow < M) & (ncol < N)
                        C[mrow[okc], ncol[okc]] = acc[i][j][okc, 4 * g + t]
    if conv:
        pad = np.pad(img, ((0, 0), (1, 1), (1, 1), (0, 0)))
        ref = np.zeros((Bn, Hh, Ww, N))
        for ky in range(3):
            for kx in range(3):
                ref += pad[:, ky:ky + Hh, kx:kx + Ww, :] @ w[:, :, ky, kx].T
        ref = ref.reshape(M, N)
    else:
        ref = A @ Bmat.T
    assert not np.isnan(C).any() and np.allclose(C, ref, atol=1e-9)


# ------------------------------------------------------------------------------------------- round 3, second half
def _dpp(v, ctrl, row_mask=0xF):
    """One v_mov_b32_dpp with old = 0 and bound_ctrl on a 64-lane vector (gfx9 DPP controls used by wave_sum_lane63 in
    gt_common.h): quad_perm (ctrl < 0x100: two bits per lane of a quad), row_shr:n (0x110 + n), row_bcast:15 (0x142),
    row_bcast:31 (0x143).  Lanes of a masked row and lanes without a source receive 0."""
    out = np.zeros(64)
    for lane in range(64):
        row, idx = lane >> 4, lane & 15
        if not (row_mask >> row) & 1:
            continue
        if ctrl < 0x100:
            src = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3)
        elif 0x111 <= ctrl <= 0x11F:
            n = ctrl - 0x110
            src = lane - n if idx >= n else None
        elif ctrl == 0x142:
            src = 16 * row - 1 if row > 0 else None              # lane 15 of the previous row
        elif ctrl == 0x143:
            src = 31 if row >= 2 else None                       # lane 31 into rows 2 and 3
        else:
            raise AssertionError(hex(ctrl))
        out[lane] = v[src] if src is not None else 0.0
    return out


def test_wave_sum_lane63_dpp_model():
    """wave_sum_lane63 (gt_common.h): after the six DPP adds lane 63 holds the sum of all 64 lanes (the other lanes hold
    partial sums) -- the one-lane store of conv_resize_bwd_kernel's 72 accumulators reads lane 63."""
    rng = np.random.default_rng(63)
    for _ in range(4):
        v = rng.standard_normal(64)
        total = v.sum()
        for ctrl, mask in ((0xB1, 0xF), (0x4E, 0xF), (0x114, 0xF), (0x118, 0xF), (0x142, 0xA), (0x143, 0xC)):
            v = v + _dpp(v, ctrl, mask)
        assert abs(v[63] - total) < 1e-12 * max(1.0, abs(total))


@pytest.mark.parametrize("W4", [9, 10, 13, 17, 18, 20, 21])
def test_headnorm_tile_store_walk(W4):
    """x3_epilogue_hn (gt_gemm_x3.hip): the staged 32 x W4 tile of 16-byte granules is stored 64 granules per instruction;
    lane l owns granules e = l + 64 it and walks (row, column) incrementally -- the walk equals divmod(e, W4), covers every
    granule of the 32 rows exactly once in nit = ceil(32 W4 / 64) iterations (+ the chunk-of-three overrun, all past row 31)."""
    nit = (32 * W4 + 63) >> 6
    dr, dc = 64 // W4, 64 - (64 // W4) * W4
    seen = np.zeros((32, W4), dtype=int)
    for lane in range(64):
        r, c = lane // W4, lane - (lane // W4) * W4
        for it in range(0, nit, 3):
            for u in range(3):
                e = lane + 64 * (it + u)
                assert (r, c) == divmod(e, W4)
                if r < 32:
                    seen[r, c] += 1
                r, c = r + dr, c + dc
                if c >= W4:
                    r, c = r + 1, c - W4
    assert (seen == 1).all()


@pytest.mark.parametrize("MI", [1, 2])
def test_fast_epilogue_batches_cover_wave_tile(MI):
    """x3_epilogue_fast (gt_gemm_x3.hip): a wave's (32 MI) x 64 tile leaves in 2 MI batches of four 256-byte row segments
    per lane group; batch b reads staging rows 16 (b & 1) + 4 k + rsub of row tile b >> 1 and writes output row
    mtile0 + 16 b + 4 k + rsub, columns 4 c4 .. + 3 -- every element exactly once, from the accumulator the MFMA left it in."""
    acc = np.arange(MI * 2 * 64 * 16, dtype=np.int64).reshape(MI, 2, 64, 16)     # [i][j][lane][register]
    out = -np.ones((32 * MI, 64), dtype=np.int64)
    for i in range(MI):
        stg = np.zeros((32, 68), dtype=np.int64)
        for lane in range(64):
            lr, lh = lane & 31, lane >> 5
            for j in range(2):
                for g in range(4):
                    stg[lr, 32 * j + 8 * g + 4 * lh: 32 * j + 8 * g + 4 * lh + 4] = acc[i, j, lane, 4 * g: 4 * g + 4]
        for b in (2 * i, 2 * i + 1):
            for lane in range(64):
                c4, rsub = lane & 15, lane >> 4
                for k in range(4):
                    row = 16 * b + 4 * k + rsub
                    assert (out[row, 4 * c4: 4 * c4 + 4] == -1).all()
                    out[row, 4 * c4: 4 * c4 + 4] = stg[16 * (b & 1) + 4 * k + rsub, 4 * c4: 4 * c4 + 4]
    # 32x32 MFMA result map with the N-side tile as the A operand (x3_epilogue's header): lane (lr, lh) holds output row
    # 32 i + lr and, in register 4 g + t of accumulator (i, j), column 32 j + 8 g + 4 lh + t
    for i in range(MI):
        for j in range(2):
            for lane in range(64):
                lr, lh = lane & 31, lane >> 5
                for g in range(4):
                    for t in range(4):
                        assert out[32 * i + lr, 32 * j + 8 * g + 4 * lh + t] == acc[i, j, lane, 4 * g + t]


@pytest.mark.parametrize("R", [3, 4, 5, 6])
@pytest.mark.parametrize("nk", [1, 2, 3, 4, 7, 8, 9, 27])
def test_packed_b_request_order_and_counted_waits(R, nk):
    """K loop of gemm_x3p_kernel (gt_gemm_x3.hip) as a request queue that retires in order: requests B(0) A(0) .. A(R-2) |
    B(1) A(R-1) | B(2) A(R) | ... (A = 2 loads, B = 6, the same in every iteration: zero stages / a repeated last B stage
    past the end of K).  At each wait `vmcnt(N)` everything but the N youngest loads has landed: the stage about to be
    consumed (A(kt) in its ring slot, B(kt) in its register set) must be among it, the slot an A request overwrites must
    already have been consumed, and a register set must not be re-requested before its MFMAs were issued."""
    queue = []                                   # issue order: (kind, stage)
    landed_before = {}                           # wait index -> set of landed requests

    def issue_a(s):
        queue.extend([("A", s)] * 2)

    def load_b(s, which):
        queue.extend([("B", s, which)] * 6)

    def wait(n):
        return set(queue[:len(queue) - n]) if n else set(queue)

    klast = nk - 1
    consumed_a, slot_of = set(), {}
    set_holds = {0: None, 1: None}

    def request_a(s):
        slot = s % R
        prev = slot_of.get(slot)
        assert prev is None or prev in consumed_a, f"slot {slot} of stage {prev} overwritten by stage {s} before use"
        slot_of[slot] = s
        issue_a(s)

    def request_b(s, which):
        held = set_holds[which]
        assert held is None or ("done", held) in done, f"B set {which} re-requested before stage {held} was consumed"
        set_holds[which] = s
        load_b(s, which)

    done = set()

    def consume(kt, which, n_wait):
        have = wait(n_wait)
        assert ("A", kt) in have and ("B", min(kt, klast), which) in have or set_holds[which] != min(kt, klast) and False, \
            f"stage {kt} consumed before it landed (vmcnt({n_wait}))"
        assert set_holds[which] == kt
        consumed_a.add(kt)
        done.add(("done", kt))

    request_b(0, 0)
    for s in range(R - 1):
        request_a(s)
    # iteration 0
    have = wait(2 * (R - 2))
    assert ("A", 0) in have and ("B", 0, 0) in have
    request_b(min(1, klast), 1)
    request_a(R - 1)
    consumed_a.add(0)
    done.add(("done", 0))
    kt = 1
    while kt + 1 < nk:
        consume(kt, 1, 2)                        # X3P_WAIT_AB(2, bn1)
        # set 0 held stage kt - 1 (consumed): request B(kt + 1) into it, then A(kt + R - 1) into the slot of stage kt - 1
        set_holds[0] = None if ("done", set_holds[0]) in done else set_holds[0]
        request_b(kt + 1, 0)
        request_a(kt + R - 1)
        consume(kt + 1, 0, 2)                    # X3P_WAIT_AB(2, bn0)
        set_holds[1] = None if ("done", set_holds[1]) in done else set_holds[1]
        request_b(min(kt + 2, klast), 1)
        request_a(kt + R)
        kt += 2
    if kt < nk:
        consume(kt, 1, 2)
    assert consumed_a >= set(range(nk))


def _mfma_16x16x32(a, b, d):
    """Lane model of v_mfma_f32_16x16x32_f16 (values kept in float64): D[16][16] += A[16][32] B[32][16];
    lane l = 16 kq + x holds a[l][e] = A[x][8 kq + e], b[l][e] = B[8 kq + e][x] (e < 8), d[l][r] = D[4 kq + r][x]."""
    x, kq = S.X, S.KQ
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for e in range(8):
        A[x, 8 * kq + e] = a[:, e]
        B[8 * kq + e, x] = b[:, e]
    D = A @ B
    out = d.copy()
    for r in range(4):
        out[:, r] += D[4 * kq + r, x]
    return out


@pytest.mark.parametrize("T", [32, 45])
def test_head_backward16_lane_map(T):
    """head_bwd16_kernel (gt_head.hip, GT_PREC_F16X2 instance), one 32-row trip of one wave pair, index maps and the algebra of
    the power-of-two scales (the fp16 rounding itself is exercised on the GPU): (1) H^T = W1 X^T, the lane's 8-float slice of a
    row is the B operand as loaded; (2) dX^T = W1^T dh^T with the k-step enumeration hidden = 32 t + 16 (e >> 2) + 4 kq + (e & 3)
    on both operands; (3) dW1 += dh^T X over the 32 rows, dh read transposed from the wave's tile scaled by 2^e_j, x scaled by
    2^(c - e_j): the row scales cancel, the result carries 2^c; rows without gradient get the factor 0."""
    HK, HN = 32, 128
    rng = np.random.default_rng(100 + T)
    X, W1 = rng.standard_normal((T, HK)), rng.standard_normal((HN, HK)) * 0.3
    X *= np.exp2(rng.integers(-6, 7, size=(T, 1)))                 # rows of very different magnitude
    b1, w2 = rng.standard_normal(HN), rng.standard_normal(HN)
    g = rng.standard_normal(T) * np.exp2(rng.integers(-20, 1, size=T))
    g[3] = 0.0                                                     # a row without gradient
    j, kq = S.X, S.KQ
    sig = lambda v: 1 / (1 + np.exp(-v))
    expo = lambda m: np.where(m > 0, 14 - np.floor(np.log2(np.where(m > 0, m, 1.0))), 0.0)   # puts m into [2^14, 2^15)
    eW = float(expo(np.abs(W1).max()))
    dX = np.full((T, HK), np.nan)
    dW1 = np.zeros((HN, HK))
    for m0 in range(0, T, 32):
        for half in range(2):                                      # the two waves of the pair: hidden [64 half, +64)
            hb = 64 * half
            dh_lds = np.zeros((32, 64))
            edh = np.zeros((2, 64)); rowmag = np.zeros((2, 64)); live = np.zeros((2, 64), bool)
            ownX = {}
            for u in range(2):
                row = np.minimum(m0 + 16 * u + j, T - 1)
                gv = np.where(m0 + 16 * u + j < T, g[row], 0.0)
                xs = X[row[:, None], (8 * kq)[:, None] + np.arange(8)]
                xm = np.abs(X[row]).max(1)                          # amax over the row's four kq lanes
                ex = expo(xm)
                d = np.zeros((64, 4, 4))
                for mt in range(4):
                    ft = hb // 16 + mt
                    wfrag = W1[(16 * ft + j)[:, None], (8 * kq)[:, None] + np.arange(8)] * 2.0 ** eW
                    acc = _mfma_16x16x32(wfrag, xs * np.exp2(ex)[:, None], np.zeros((64, 4)))
                    hid = hb + 16 * mt + 4 * kq[:, None] + np.arange(4)
                    hpre = acc * np.exp2(-ex)[:, None] * 2.0 ** -eW + b1[hid]
                    sg = sig(hpre)
                    d[:, mt] = gv[:, None] * w2[hid] * (sg * (1 + hpre * (1 - sg)))
                dm = np.abs(d).reshape(64, 16).max(1)
                dm = np.array([dm[(j == j[l])].max() for l in range(64)])      # over the row's four kq lanes
                ed = expo(dm)
                live[u], edh[u] = dm > 0, ed
                rowmag[u] = np.where(dm > 0, xm * np.exp2(-ed), 0.0)
                ds = d * np.exp2(ed)[:, None, None]
                for l in range(64):
                    for mt in range(4):
                        dh_lds[16 * u + j[l], 16 * mt + 4 * kq[l]:16 * mt + 4 * kq[l] + 4] = ds[l, mt]
                accX = [np.zeros((64, 4)), np.zeros((64, 4))]
                for t in range(2):
                    bfrag = np.concatenate([ds[:, 2 * t], ds[:, 2 * t + 1]], axis=1)          # e < 4: tile 2t, e >= 4: tile 2t + 1
                    for ti in range(2):
                        e = np.arange(8)
                        hidden = 32 * (2 * half + t) + 16 * (e >> 2)[None, :] + 4 * kq[:, None] + (e & 3)[None, :]
                        afrag = W1[hidden, (16 * ti + j)[:, None]] * 2.0 ** eW
                        accX[ti] = _mfma_16x16x32(afrag, bfrag, accX[ti])
                ownX[u] = [a * np.exp2(-ed)[:, None] * 2.0 ** -eW for a in accX]
            partial = ownX
            if half == 0:
                first = partial
            # (3) with this wave's tile
            c = 13 - np.floor(np.log2(rowmag.max())) if rowmag.max() > 0 else 0.0
            fr = np.zeros(32)
            for u in range(2):
                for l in range(64):
                    if kq[l] == 0:
                        fr[16 * u + j[l]] = 2.0 ** (c - edh[u, l]) if live[u, l] else 0.0
            accW = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(4)]
            rows8 = np.minimum(m0 + 8 * kq[:, None] + np.arange(8), T - 1)
            for mt in range(4):
                afrag = dh_lds[(8 * kq)[:, None] + np.arange(8), (16 * mt + j)[:, None]]
                for ti in range(2):
                    bfrag = X[rows8, (16 * ti + j)[:, None]] * fr[(8 * kq)[:, None] + np.arange(8)]
                    accW[mt][ti] = _mfma_16x16x32(afrag, bfrag, accW[mt][ti])
            for l in range(64):
                for mt in range(4):
                    for ti in range(2):
                        for r in range(4):
                            dW1[hb + 16 * mt + 4 * kq[l] + r, 16 * ti + j[l]] += accW[mt][ti][l, r] * 2.0 ** -c
            if half == 1:                                          # the pair's exchange: wave `half` stores in-features [16 half, +16)
                for u in range(2):
                    for hh, mine, other in ((0, first, partial), (1, partial, first)):
                        tot = mine[u][hh] + other[u][hh]
                        for l in range(64):
                            if m0 + 16 * u + j[l] < T:
                                dX[m0 + 16 * u + j[l], 16 * hh + 4 * kq[l]:16 * hh + 4 * kq[l] + 4] = tot[l]
    H = X @ W1.T + b1
    sg = sig(H)
    dh = g[:, None] * w2[None, :] * (sg * (1 + H * (1 - sg)))
    assert np.allclose(dX, dh @ W1, rtol=1e-9, atol=1e-12 * np.abs(dh @ W1).max())
    ref = dh.T @ X
    assert np.allclose(dW1, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("tiles,n_split", [(1, 256), (2, 256), (2, 171), (3, 171), (3, 8), (2, 5)])
def test_x3w_chunk_major_block_order(tiles, n_split):
    """gemm_x3w_kernel with x3w_map (gt_gemm_x3.hip): a 1-D grid of 8 ceil(n_split / 8) tiles blocks, block id % 8 = XCD;
    XCD x owns the K chunks [x spx, (x + 1) spx) and the tiles of one chunk are consecutive blocks of that XCD.  Every
    (tile, chunk) pair is worked exactly once, surplus blocks leave, and a chunk's tiles share an XCD back to back."""
    spx = (n_split + 7) >> 3
    grid = 8 * spx * tiles
    seen = {}
    for bid in range(grid):
        s = bid >> 3
        by, tile = (bid & 7) * spx + s // tiles, s % tiles
        if by >= n_split:
            continue
        assert (tile, by) not in seen
        seen[(tile, by)] = bid
    assert set(seen) == {(t, c) for t in range(tiles) for c in range(n_split)}
    for c in range(n_split):
        ids = [seen[(t, c)] for t in range(tiles)]
        assert len({i & 7 for i in ids}) == 1                                  # one XCD
        assert [i >> 3 for i in ids] == list(range(ids[0] >> 3, (ids[0] >> 3) + tiles))   # consecutive slots of it


@pytest.mark.parametrize("Cout,npx", [(128, 70), (64, 33), (16, 5)])
def test_conv0_decision_bits_layout(Cout, npx):
    """relu_bits of the fused conv0 + resize (gt_resize.hip): the forward thread of (pixel e, channel block bc of 16) writes
    ONE 64-bit word [b][bc][e] with nibble j = channel 16 bc + j; the backward thread of (pixel e, channel group c0 of 8) reads
    the same word, takes its low or high half by c0 & 8, and tests bit 4 j + t for channel c0 + j, source pixel t."""
    rng = np.random.default_rng(Cout + npx)
    B = 2
    dec = rng.integers(0, 16, size=(B, npx, Cout))                 # the forward's four decisions per (pixel, channel)
    words = np.zeros((B, Cout // 16, npx), dtype=np.uint64)
    for b in range(B):
        for bc in range(Cout // 16):
            for e in range(npx):
                nib = 0
                for j4 in range(0, 16, 4):                          # the forward packs four channels at a time
                    n16 = 0
                    for jj in range(4):
                        n16 |= int(dec[b, e, 16 * bc + j4 + jj]) << (4 * jj)
                    nib |= n16 << (4 * j4)
                words[b, bc, e] = nib
    for b in range(B):
        for c0 in range(0, Cout, 8):
            for e in range(npx):
                w64 = int(words[b, c0 >> 4, e])
                d32 = (w64 >> 32) if (c0 & 8) else (w64 & 0xFFFFFFFF)
                for j in range(8):
                    for t in range(4):
                        assert bool(d32 & (1 << (4 * j + t))) == bool((int(dec[b, e, c0 + j]) >> t) & 1)


@pytest.mark.parametrize("K,with_acs", [(64, True), (45, False)])
def test_x3w_kernel_lane_map(K, with_acs):
    """gemm_x3w_kernel (gt_gemm_x3.hip), one 128 x 128 output tile of one K chunk: the staging roles (waves 0-1 the A tile,
    2-3 the B tile; a thread owns rows 4 r4 .. + 3 and the eight tokens of k-group kg), the plane layout [k-group][row] of
    16-byte units (eight tokens of one row), the fragment a lane reads for MFMA k-step ks (unit 2 ks + lh of its row), the
    operand order of the MFMA (N-side tile first), the sign alternation of odd rows on both sides and its undo, the slab
    store map, and the bias-gradient column sums.  Values stay float64 (the fp16 split is exercised on the GPU)."""
    rng = np.random.default_rng(K)
    A, B = rng.standard_normal((K, 128)), rng.standard_normal((K, 128))
    lanes = np.arange(64)
    lr, lh = lanes & 31, lanes >> 5
    acc = {(w, i, jj): np.zeros((64, 16)) for w in range(4) for i in range(2) for jj in range(2)}
    asum = np.zeros(128)
    for k0 in range(0, K, 32):
        planes = {"A": np.zeros((4, 128, 8)), "B": np.zeros((4, 128, 8))}     # [k-group][row][token in group]
        for tid in range(256):
            wave = tid >> 6
            op = "B" if wave >= 2 else "A"
            src = B if wave >= 2 else A
            st = tid & 127
            r4, kg = st & 31, st >> 5
            v = np.zeros((8, 4))
            for e in range(8):
                k = k0 + 8 * kg + e
                if k < K:
                    v[e] = src[k, 4 * r4:4 * r4 + 4]
            for c in range(4):
                if op == "A" and with_acs:
                    asum[4 * r4 + c] += v[:, c].sum()
                sign = -1.0 if (c & 1) else 1.0                                  # odd rows enter negated (row 4 r4 + c)
                planes[op][kg, 4 * r4 + c] = sign * v[:, c]
        for w in range(4):
            wm, wn = w >> 1, w & 1
            for ks in range(2):
                kq = 2 * ks + lh
                am = [planes["A"][kq, wm * 64 + 32 * i + lr] for i in range(2)]     # [64 lanes][8 tokens]
                bn = [planes["B"][kq, wn * 64 + 32 * jj + lr] for jj in range(2)]
                for i in range(2):
                    for jj in range(2):
                        acc[(w, i, jj)] = _mfma32(bn[jj], am[i], acc[(w, i, jj)])
    C = np.full((128, 128), np.nan)
    for w in range(4):
        wm, wn = w >> 1, w & 1
        for i in range(2):
            m = wm * 64 + 32 * i + lr
            us = np.where(lr & 1, -1.0, 1.0)                                      # x3_alt_sign(lr): the row's sign
            for jj in range(2):
                for g_ in range(4):
                    n = wn * 64 + 32 * jj + 8 * g_ + 4 * lh
                    a = acc[(w, i, jj)]
                    vals = np.stack([a[:, 4 * g_] * us, -a[:, 4 * g_ + 1] * us, a[:, 4 * g_ + 2] * us, -a[:, 4 * g_ + 3] * us], 1)
                    for l in range(64):
                        C[m[l], n[l]:n[l] + 4] = vals[l]
    assert np.allclose(C, A.T @ B, atol=1e-9)
    if with_acs:
        assert np.allclose(asum, A.sum(0), atol=1e-9)


# ------------------------------------------------------------------------------------------------ gt_fourier16.hip
def _f16_geom(DP):
    NM = DP // 32; TC = DP - 32 * NM; TG = (TC + 7) // 8; ND = (DP + 15) // 16
    MAIN_G, TAIL_G = 128 * NM, 32 * TG
    return dict(NM=NM, TG=TG, NS=NM + 1, ND=ND, MAIN_G=MAIN_G, TAIL_G=TAIL_G, RM_G=MAIN_G + TAIL_G, TR_G=64 * ND)


def _f16_presplit(X, tile, DP):
    """fourier16_presplit_kernel: the image of one 32-row tile as granules of 8 values, (exponent, norm bound).  The fp16
    planes are kept as float64 holding fp16-representable values."""
    g = _f16_geom(DP)
    NM, TG, MAIN_G, TAIL_G, RM_G, TR_G = g["NM"], g["TG"], g["MAIN_G"], g["TAIL_G"], g["RM_G"], g["TR_G"]
    s = np.zeros((32, DP))
    rows = X[32 * tile:32 * tile + 32]
    s[:len(rows)] = rows
    amax = np.abs(s).max()
    ex = 0 if amax == 0 else 13 - int(np.floor(np.log2(amax)))           # amax 2^ex in [2^13, 2^14)
    sc = np.exp2(ex)
    nrm = np.sqrt(((s * sc) ** 2).sum(1).max()) * 1.0001
    ln = max(int(np.floor(np.log2(nrm))) + 1, -40) if nrm > 0 else -40   # nrm < 2^ln
    img = np.zeros((2 * RM_G + 2 * TR_G, 8))
    for gi in range(RM_G + TR_G):
        v = np.zeros(8)
        if gi < MAIN_G:
            row, q = gi // (4 * NM), gi % (4 * NM)
            v = s[row, 8 * q:8 * q + 8].copy()
            d0, d1 = gi, MAIN_G + gi
        elif gi < RM_G:
            gt_ = gi - MAIN_G
            row, q = gt_ // TG, gt_ % TG
            c0 = 32 * NM + 8 * q
            for e in range(8):
                v[e] = s[row, c0 + e] if c0 + e < DP else 0.0
            d0, d1 = 2 * MAIN_G + gt_, 2 * MAIN_G + TAIL_G + gt_
        else:
            gr = gi - RM_G
            kq, m, dt = gr & 3, (gr >> 2) & 15, gr >> 6
            col = 16 * dt + m
            sg = -1.0 if (col & 1) else 1.0
            for e in range(8):
                row = 16 * (e >> 2) + 4 * kq + (e & 3)
                v[e] = sg * s[row, col] if col < DP else 0.0
            d0, d1 = 2 * RM_G + gr, 2 * RM_G + TR_G + gr
        h0 = (v * sc).astype(np.float16).astype(np.float64)
        h1 = (v * sc - h0).astype(np.float16).astype(np.float64)
        img[d0], img[d1] = h0, h1
    return img, ex, ln


def _fmix_top(x):
    """fmix32 without its last xorshift (bits 16..31 do not depend on it), uint32 arithmetic."""
    x = np.asarray(x, dtype=np.uint64) & 0xffffffff
    x ^= x >> 16; x = (x * 0x85ebca6b) & 0xffffffff
    x ^= x >> 13; x = (x * 0xc2b2ae35) & 0xffffffff
    return x


def _block16_keep(zn4, nq4, q, k, key):
    """gt_hip.h gt_dropout_block16: element (q, k) keeps iff bit 16 + 4 (q & 3) + (k & 3) of the hash of its 4 x 4 block."""
    blk = ((zn4 + (q >> 2)) * nq4 + (k >> 2)) & 0xffffffff
    x = _fmix_top((blk * 0x9e3779b1 + key) & 0xffffffff)
    return (int(x) >> (16 + 4 * (q & 3) + (k & 3))) & 1


@pytest.mark.parametrize("DP,n,mode", [(20, 75, "plain"), (36, 75, "plain"), (52, 40, "plain"), (36, 75, "block_key"),
                                       (20, 44, "block_query")])
def test_fourier16_lane_map(DP, n, mode):
    """fourier16_kernel<DP, false, PLAIN / DROPB> (gt_fourier16.hip), every owner wave of one (batch, head): the images the
    pre-split writes, the fragment offsets the kernel reads them at, the D-layout -> B-layout register hand-over between the
    two products (contraction slot 8 kq + 4 mt + r = stream row 16 mt + 4 kq + r), the running exponent, the chain sign
    (-1)^(dim + owner), and the per-block dropout bookkeeping (hw / hstep / bpos) against the definition in gt_hip.h.
    O[o][d] = scale sum_s keep(s, o) T2[s][d] (T1[s] . F1[o]); two-term fp16 planes, fp32 accumulation."""
    g = _f16_geom(DP)
    NM, TG, NS, ND, MAIN_G, TAIL_G, RM_G, TR_G = (g[k] for k in ("NM", "TG", "NS", "ND", "MAIN_G", "TAIL_G", "RM_G", "TR_G"))
    rng = np.random.default_rng(DP + n)
    mag = np.exp2(rng.integers(-5, 6, size=(3, n, 1)))                    # rows of very different magnitude
    F1, T1, T2 = (rng.standard_normal((n, DP)) * mag[i] for i in range(3))
    ntile = (n + 31) // 32
    imgs = {nm: [_f16_presplit(Xt, t, DP) for t in range(ntile)] for nm, Xt in (("F1", F1), ("T1", T1), ("T2", T2))}
    j, kq = J, KQ
    scale, key, zn4, nq4 = 1.0 / n, 0x1234abcd, 5 * ((n + 3) >> 2), (n + 3) >> 2
    owner_is_key = mode == "block_key"
    keep = np.ones((n, n))                                                # [query][key]
    if mode != "plain":
        for q in range(n):
            for k in range(n):
                keep[q, k] = _block16_keep(zn4, nq4, q, k, key)

    def mma3(ah, al, bh, bl, acc):
        for a, b in ((al, bh), (ah, bl), (ah, bh)):
            acc = _mfma_16x16x32(a, b, acc).astype(np.float32).astype(np.float64)
        return acc

    O = np.full((n, DP), np.nan)
    for ot in range(ntile):
        o0 = 32 * ot
        fimg, ef1, lf1 = imgs["F1"][ot]
        f1h, f1l = {}, {}
        for nt in range(2):
            for s in range(NS):
                ok = (kq < TG) if s == NM else np.ones(64, bool)
                g0 = (16 * nt + j) * 4 * NM + 4 * s + kq if s < NM else 2 * MAIN_G + (16 * nt + j) * TG + kq
                g1 = g0 + (MAIN_G if s < NM else TAIL_G)
                f1h[nt, s] = np.where(ok[:, None], fimg[np.minimum(g0, len(fimg) - 1)], 0.0)
                f1l[nt, s] = np.where(ok[:, None], fimg[np.minimum(g1, len(fimg) - 1)], 0.0)
        hw = {}
        for nt in range(2):
            own4 = (o0 + 16 * nt + j) >> 2
            blk = (zn4 + kq) * nq4 + own4 if owner_is_key else (zn4 + own4) * nq4 + kq
            hw[nt] = (blk * 0x9e3779b1 + key) & 0xffffffff
        hstep = (nq4 if owner_is_key else 1) * 0x9e3779b1
        bpos = [16 + (4 * r + (j & 3) if owner_is_key else 4 * (j & 3) + r) for r in range(4)]
        osign = np.where(j & 1, -1.0, 1.0)
        acc1 = {(dt, nt): np.zeros((64, 4)) for dt in range(ND) for nt in range(2)}
        EA = 1 << 20
        for t in range(ntile):
            (i1, e1, l1), (i2, e2, l2) = imgs["T1"][t], imgs["T2"][t]
            cap = 15 - l1 - lf1 + e1 + ef1 + e2
            if cap < EA:
                if t > 0:
                    for kk in acc1:
                        acc1[kk] = acc1[kk] * np.exp2(cap - EA)
                EA = cap
            sigA = osign * np.exp2(EA - e1 - ef1 - e2)
            km = np.ones((2, 8, 64))
            if mode != "plain":
                for nt in range(2):
                    hk = hw[nt]
                    for mt in range(2):
                        xb = _fmix_top(hk)
                        hk = (hk + 4 * hstep) & 0xffffffff
                        for r in range(4):
                            km[nt, 4 * mt + r] = (xb >> np.asarray(bpos[r], dtype=np.uint64)) & 1
                    hw[nt] = hk
            sa = {}
            for mt in range(2):
                for nt in range(2):
                    sa[mt, nt] = np.zeros((64, 4))
                for s in range(NS):
                    if s < NM:
                        a0, pl = (16 * mt + j) * 4 * NM + 4 * s + kq, MAIN_G
                        ok = np.ones(64, bool)
                    else:
                        a0, pl = 2 * MAIN_G + (16 * mt + j) * TG + kq, TAIL_G
                        ok = kq < TG
                    ah = np.where(ok[:, None], i1[np.minimum(a0, len(i1) - 1)], 0.0)
                    al = np.where(ok[:, None], i1[np.minimum(a0 + pl, len(i1) - 1)], 0.0)
                    for nt in range(2):
                        sa[mt, nt] = mma3(ah, al, f1h[nt, s], f1l[nt, s], sa[mt, nt])
            ph, pl_ = {}, {}
            for nt in range(2):
                v = np.zeros((64, 8))
                for mt in range(2):
                    for r in range(4):
                        v[:, 4 * mt + r] = sa[mt, nt][:, r] * sigA * km[nt, 4 * mt + r]
                assert np.abs(v).max() < 65504.0                            # the Cauchy-Schwarz bound behind `cap`
                ph[nt] = v.astype(np.float32).astype(np.float16).astype(np.float64)
                pl_[nt] = (v.astype(np.float32) - ph[nt].astype(np.float32)).astype(np.float16).astype(np.float64)
            for dt in range(ND):
                a0 = 2 * RM_G + j * 4 + kq + 64 * dt
                for nt in range(2):
                    acc1[dt, nt] = mma3(i2[a0], i2[a0 + TR_G], ph[nt], pl_[nt], acc1[dt, nt])
        fs = scale * (2.0 if mode != "plain" else 1.0)
        for nt in range(2):
            ow = o0 + 16 * nt + j
            for dt in range(ND):
                dim = 16 * dt + 4 * kq
                for l in range(64):
                    if ow[l] < n and dim[l] < DP:
                        for r in range(4):
                            sg = -fs if ((r + j[l]) & 1) else fs
                            O[ow[l], dim[l] + r] = acc1[dt, nt][l, r] * sg * np.exp2(-EA)
    Sc = T1 @ F1.T                                                          # [stream][owner]
    # keep is [query][key]; the score matrix here is [stream][owner]: stream = query when the owner is the key
    kso = keep if owner_is_key else keep.T
    ref = scale * (2.0 if mode != "plain" else 1.0) * (Sc * kso).T @ T2
    assert not np.isnan(O).any()
    err = np.linalg.norm(O - ref) / np.linalg.norm(ref)
    assert err < 2e-6, err
