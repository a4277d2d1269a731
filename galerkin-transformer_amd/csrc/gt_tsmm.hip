// Tall-skinny weight-gradient products  C (M x N) = alpha * A^T B,  A [K, M], B [K, N] row-major with K in the
// millions and M <= 128, N <= 32: the decoder's pointwise layers at every fine-grid point (model.py:615-629
// SpectralRegressor.fc / regressor, layers.py:1128 SpectralConv2d.linear -- their nn.Linear weight gradients
// under autograd).  The operands are read exactly once and the result is tiny, so this is an HBM stream; on
// the tiled GEMM engine the K slices of such a problem keep only one small stage per block in flight
// (measured 0.7-1.4 TB/s).  Here every wave owns a contiguous range of rows and feeds the MFMA operands
// straight from global memory: lane (i, kq) loads the float2 A[k + kq][2i .. 2i+1] of a 32-column group (one
// 128-byte row per 16 lanes), which is the A operand of TWO 16x16x4 tiles (even / odd columns); likewise B.
// Several 4-row steps are requested before the first MFMA, 3-4 blocks per CU keep > 64 KB per CU in flight.
// Fixed-order reduction: 4 waves -> LDS -> one slab per block -> gt_slab_reduce.  Optional by-product: the
// column sums of A (the bias gradient), from the same registers.
#include "gt_common.h"
#include <algorithm>

namespace gt {

struct TsmmP {
    const float* A; const float* B; float* slabs; float a_sign;
    int64_t lda, ldb;
    int M, N, K, rows_per_wave, want_colsum;
};

template <int MP, int U>      // MP = M / 32 column groups of A, U = 4-row steps in flight
__global__ __launch_bounds__(256, MP >= 3 ? 2 : 3) void tsmm_kernel(const TsmmP p) {
    __shared__ float red[MP * 32 * 32 + MP * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t k0 = gw * p.rows_per_wave;
    const int64_t k1 = std::min<int64_t>(k0 + p.rows_per_wave, p.K);
    const bool bcol = 2 * i < p.N;                       // this lane's B column pair exists

    f32x4 acc[2 * MP][2];
    f32x2 cs[MP];
#pragma unroll
    for (int t = 0; t < 2 * MP; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mp = 0; mp < MP; ++mp) cs[mp] = f32x2{0.f, 0.f};

    const float* ap = p.A + (k0 + kq) * p.lda + 2 * i;
    const float* bp = p.B + (k0 + kq) * p.ldb + 2 * i;
    int64_t k = k0;
    for (; k + 4 * U <= k1; k += 4 * U) {                // full groups: no guards
        f32x2 a[U][MP], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int mp = 0; mp < MP; ++mp) a[u][mp] = *reinterpret_cast<const f32x2*>(ap + 4 * u * p.lda + 32 * mp);
            b[u] = bcol ? *reinterpret_cast<const f32x2*>(bp + 4 * u * p.ldb) : f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mp = 0; mp < MP; ++mp) {
                cs[mp] += a[u][mp];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[2 * mp + e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mp][e], b[u][f], acc[2 * mp + e][f], 0, 0, 0);
            }
        ap += 4 * U * p.lda;
        bp += 4 * U * p.ldb;
    }
    for (; k < k1; k += 4) {                             // tail: rows beyond the range contribute zeros
        const bool ok = k + kq < k1;
        f32x2 a[MP], b;
#pragma unroll
        for (int mp = 0; mp < MP; ++mp) a[mp] = ok ? *reinterpret_cast<const f32x2*>(ap + 32 * mp) : f32x2{0.f, 0.f};
        b = (ok && bcol) ? *reinterpret_cast<const f32x2*>(bp) : f32x2{0.f, 0.f};
#pragma unroll
        for (int mp = 0; mp < MP; ++mp) {
            cs[mp] += a[mp];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int f = 0; f < 2; ++f)
                    acc[2 * mp + e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mp][e], b[f], acc[2 * mp + e][f], 0, 0, 0);
        }
        ap += 4 * p.lda;
        bp += 4 * p.ldb;
    }
    // column sums: combine the 4 row lanes
#pragma unroll
    for (int mp = 0; mp < MP; ++mp)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float v = cs[mp][e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            cs[mp][e] = v;
        }
    // block reduction in wave order (deterministic): tile (2mp+e, f) register r of lane (j, kq) is
    // C[m = 32mp + 2(4kq + r) + e][n = 2j + f]
    const int MN = p.M * p.N;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mp = 0; mp < MP; ++mp)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = 32 * mp + 2 * (4 * kq + r) + e, n = 2 * i + f;
                            if (n < p.N) {
                                float* q = &red[m * p.N + n];
                                *q = (w == 0) ? acc[2 * mp + e][f][r] : *q + acc[2 * mp + e][f][r];
                            }
                        }
                    if (kq == 0) {
                        float* q = &red[MN + 32 * mp + 2 * i + e];
                        *q = (w == 0) ? cs[mp][e] : *q + cs[mp][e];
                    }
                }
        }
        __syncthreads();
    }
    const int tot = MN + (p.want_colsum ? p.M : 0);
    float* out = p.slabs + (int64_t)blockIdx.x * (MN + p.M);
    for (int e = tid; e < tot; e += 256) out[e] = red[e];
}

// C[m][n] = alpha * sum over slabs (fixed order: 8 interleaved slab lanes, then the 8 partials in order);
// elements >= M*N of a slab are the column sums of A.
__global__ __launch_bounds__(256) void tsmm_reduce_kernel(const float* __restrict__ slabs, int nslab, int M, int N,
                                                         int64_t ldc, float alpha, float* __restrict__ C,
                                                         float cs_scale, float* __restrict__ colsum) {
    __shared__ float part[8][32];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int MN = M * N, stride = MN + M, tot = MN + (colsum ? M : 0);
    const int e = blockIdx.x * 32 + el;
    float a0 = 0.f, a1 = 0.f;
    if (e < tot) {
        int s = sl;
        for (; s + 8 < nslab; s += 16) {
            a0 += slabs[(int64_t)s * stride + e];
            a1 += slabs[(int64_t)(s + 8) * stride + e];
        }
        if (s < nslab) a0 += slabs[(int64_t)s * stride + e];
    }
    part[sl][el] = a0 + a1;
    __syncthreads();
    if (sl == 0 && e < tot) {
        float v = part[0][el];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += part[k][el];
        if (e < MN) C[(int64_t)(e / N) * ldc + e % N] = alpha * v;
        else colsum[e - MN] = cs_scale * v;
    }
}

// rows per wave: a multiple of 16 that spreads K over ~4 blocks per CU
static void tsmm_geometry(int K, int* blocks, int* rows_per_wave) {
    const int64_t waves = 4096;
    int64_t rpw = (K + waves - 1) / waves;
    rpw = std::max<int64_t>(64, (rpw + 15) / 16 * 16);
    *rows_per_wave = (int)rpw;
    *blocks = (int)((K + 4 * rpw - 1) / (4 * rpw));
}

bool tsmm_eligible(const gt_gemm_desc* d) {
    if (d->layout_a != 1 || d->layout_b != 1 || d->batch0 * d->batch1 != 1 || d->split_k != 0) return false;
    if (d->K < 65536 || d->M % 32 || d->M > 128 || d->N > 32 || (d->N & 1)) return false;
    if (d->a_drop.p > 0.f || d->bias || d->rp || d->add || d->pre || d->act || d->aux_op || d->drop.p > 0.f ||
        d->res || d->out_scale != 1.f || d->ep_mode != GT_EP_NORMAL || d->K2 > 0 || d->c_masked)
        return false;
    if ((d->lda & 1) || (d->ldb & 1) || d->ldc < d->N) return false;
    if ((reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->B)) & 7) return false;
    return true;
}

int64_t tsmm_ws_bytes(const gt_gemm_desc* d) {
    int blocks, rpw;
    tsmm_geometry(d->K, &blocks, &rpw);
    return (int64_t)blocks * ((int64_t)d->M * d->N + d->M) * (int64_t)sizeof(float);
}

const char* tsmm_kernel_name(const gt_gemm_desc* d) {
    switch (d->M / 32) {
        case 1: return "void gt::tsmm_kernel<1, 8>(gt::TsmmP)";
        case 2: return "void gt::tsmm_kernel<2, 4>(gt::TsmmP)";
        case 3: return "void gt::tsmm_kernel<3, 2>(gt::TsmmP)";
        default: return "void gt::tsmm_kernel<4, 2>(gt::TsmmP)";
    }
}

int tsmm_run(const gt_gemm_desc* d, void* ws, int64_t ws_bytes, void* stream) {
    if (!ws || ws_bytes < tsmm_ws_bytes(d)) return GT_EWS;
    int blocks, rpw;
    tsmm_geometry(d->K, &blocks, &rpw);
    TsmmP p{d->A, d->B, reinterpret_cast<float*>(ws), d->a_drop_sign, d->lda, d->ldb, d->M, d->N, d->K, rpw,
            d->a_colsum ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    switch (d->M / 32) {
        case 1: hipLaunchKernelGGL((tsmm_kernel<1, 8>), dim3(blocks), dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((tsmm_kernel<2, 4>), dim3(blocks), dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((tsmm_kernel<3, 2>), dim3(blocks), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((tsmm_kernel<4, 2>), dim3(blocks), dim3(256), 0, st, p); break;
    }
    GT_LAUNCH_CHECK();
    const int tot = d->M * d->N + (d->a_colsum ? d->M : 0);
    hipLaunchKernelGGL(tsmm_reduce_kernel, dim3((tot + 31) / 32), dim3(256), 0, st, reinterpret_cast<const float*>(ws),
                       blocks, d->M, d->N, d->ldc, d->alpha, d->C, d->a_drop_sign, d->a_colsum);
    GT_LAUNCH_CHECK();
    return 0;
}

}  // namespace gt
