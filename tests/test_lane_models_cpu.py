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
