"""Lane-accurate numpy model of v_mfma_f32_16x16x4_f32, for checking a kernel's operand / result index maps
on the CPU before it ever runs on a GPU (hipcc checks the syntax, this checks the arithmetic of the layout).

    D[i][j] += sum_k A[i][k] * B[k][j],  i, j < 16, k < 4
    lane l = 16*kq + x  (x = l & 15, kq = l >> 4):
        a[l] = A[x][kq]      b[l] = B[kq][x]      d[l][r] = D[4*kq + r][x],  r < 4
"""
import numpy as np

LANES = np.arange(64)
X = LANES & 15          # "i" for the A operand, "j" for B and for the D column
KQ = LANES >> 4


def mfma(a, b, d):
    """a, b: [64] per-lane scalars; d: [64, 4] per-lane accumulators.  Returns the new d."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[X, KQ] = a
    B[KQ, X] = b
    D = A @ B                                   # [16, 16]
    out = d.copy()
    for r in range(4):
        out[:, r] += D[4 * KQ + r, X]
    return out


def selftest():
    rng = np.random.default_rng(0)
    A, B = rng.standard_normal((16, 4)), rng.standard_normal((4, 16))
    d = mfma(A[X, KQ], B[KQ, X], np.zeros((64, 4)))
    D = A @ B
    for l in range(64):
        for r in range(4):
            assert abs(d[l, r] - D[4 * (l >> 4) + r, l & 15]) < 1e-12


if __name__ == "__main__":
    selftest()
    print("mfma_sim ok")
