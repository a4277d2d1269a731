// Fused pointwise regression head  y = w2 . act(W1 x + b1) + b2  at every grid point: the tail of
// SpectralRegressor (model.py:575-580, 625-629: Linear(32 -> 128), activation, Linear(128 -> 1)), forward and
// the COMPLETE backward in one pass over x each -- the [T, 128] hidden activation and its gradient never touch
// HBM (at batch 128 they are 1.3 GB each; the unfused backward wrote dL/dh once and read it twice).
//
// Every product is a 16x16x4 fp32 MFMA on a 16-row tile, fed without cross-lane shuffles:
//
//   (1) H^T (hidden x rows) = W1 X^T          A = W1 rows from LDS, B = the row's 8-float slice [8kq, 8kq+8) of x
//                                              straight from the two float4 the lane loaded (the contraction
//                                              index is enumerated as k = 8kq + s on both operands)
//       D layout: lane (row j, kq) holds hidden 16mt + 4kq + r  ->  bias / activation / w2 per register
//   forward:   out[row] = sum over the lane's hidden values, then over the 4 kq lanes
//   backward:  dh^T = g[row] * w2[hidden] * act'(h)      kept in the same registers
//   (2) dX^T (in x rows)  = W1^T dh^T         B operand = register s of the same lane (hidden 16mt+4kq+s)
//   (3) dW1 (hidden x in) += dh^T X           the one layout change: dh goes through a wave-private LDS tile
//                                              [16 rows][hidden], A = dh^T[hidden][row 4kq+s], B = x[row 4kq+s][in]
//       dw2 += g * act(h),  db1 += dh,  db2 += g   per-lane register sums
// and a fixed-order reduction at the end (lanes -> waves -> one slab per block -> head_reduce_kernel).
// The backward splits the hidden axis over a PAIR of waves (64 each; the full 128 would need > 256 registers
// per lane for the dW1 accumulators): the pair exchanges its two partial dX^T tiles through LDS once per tile.
#include "gt_common.h"
#include <algorithm>

namespace gt {

constexpr int HK = 32, HN = 128;     // input features, hidden width
constexpr int W1P = 36;              // LDS pitch of W1 [128][32]: b128 reads of 8 floats at 8kq, conflict-free
constexpr int DHP = 68;              // LDS pitch of the wave-private dh tile [16 rows][64 hidden]
constexpr int HSLAB = HN * HK + 2 * HN + 1;   // dW1 | dw2 | db1 | db2

struct HeadP {
    const float* X; const float* W1; const float* b1; const float* w2; const float* b2; const float* g;
    float* out; float* dX; float* slabs;
    int64_t T;
    int n_mtiles;
};

template <int ACT>
__device__ __forceinline__ void head_act(float h, float& a, float& da) {
    if (ACT == GT_ACT_SILU) silu_both(h, a, da);
    else if (ACT == GT_ACT_RELU) { a = fmaxf(h, 0.f); da = h > 0.f ? 1.f : 0.f; }
    else { a = h; da = 1.f; }
}

__device__ __forceinline__ void head_stage_weights(const HeadP& p, float* sW1, float* sB1, float* sW2) {
    const int tid = threadIdx.x;
    for (int e = tid; e < HN * HK; e += 256) sW1[(e >> 5) * W1P + (e & 31)] = p.W1[e];
    if (tid < HN) {
        sB1[tid] = p.b1 ? p.b1[tid] : 0.f;
        sW2[tid] = p.w2[tid];
    }
    __syncthreads();
}

// H^T tiles of NMT hidden groups starting at hidden hb: acc[mt][r] = sum_k W1[hb + 16mt + 4kq + r][k] x[row j][k]
template <int NMT>
__device__ __forceinline__ void head_gemm1(const float* sW1, int hb, int j, int kq, const float (&xs)[8],
                                           f32x4 (&acc)[NMT]) {
#pragma unroll
    for (int mt = 0; mt < NMT; mt += 2) {
        f32x4 w[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            w[u][0] = *reinterpret_cast<const f32x4*>(&sW1[(hb + 16 * (mt + u) + j) * W1P + 8 * kq]);
            w[u][1] = *reinterpret_cast<const f32x4*>(&sW1[(hb + 16 * (mt + u) + j) * W1P + 8 * kq + 4]);
            acc[mt + u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc[mt + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][s >> 2][s & 3], xs[s], acc[mt + u], 0, 0, 0);
    }
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void head_fwd_kernel(const HeadP p) {
    __shared__ __attribute__((aligned(16))) float sW1[HN * W1P];
    __shared__ __attribute__((aligned(16))) float sB1[HN], sW2[HN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    head_stage_weights(p, sW1, sB1, sW2);
    const float b2v = p.b2 ? p.b2[0] : 0.f;
    const int stride = gridDim.x * 4;
    int mtile = blockIdx.x * 4 + wave;
    f32x4 xa = {0.f, 0.f, 0.f, 0.f}, xb = {0.f, 0.f, 0.f, 0.f};
    if (mtile < p.n_mtiles) {       // lane (row j, kq) takes the floats [8kq, 8kq+8) of its row
        const int64_t row = std::min<int64_t>((int64_t)mtile * 16 + j, p.T - 1);
        xa = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq);
        xb = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq + 4);
    }
    for (; mtile < p.n_mtiles; mtile += stride) {
        const int64_t m0 = (int64_t)mtile * 16;
        const float xs[8] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
        if (mtile + stride < p.n_mtiles) {
            const int64_t row = std::min<int64_t>((int64_t)(mtile + stride) * 16 + j, p.T - 1);
            xa = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq);
            xb = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq + 4);
        }
        f32x4 acc[8];
        head_gemm1<8>(sW1, 0, j, kq, xs, acc);
        float part = 0.f;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&sB1[16 * mt + 4 * kq]);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&sW2[16 * mt + 4 * kq]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a, da;
                head_act<ACT>(acc[mt][r] + bv[r], a, da);
                part = fmaf(a, wv[r], part);
            }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (kq == 0 && m0 + j < p.T) p.out[m0 + j] = part + b2v;
    }
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void head_bwd_kernel(const HeadP p) {
    __shared__ __attribute__((aligned(16))) float sW1[HN * W1P];
    __shared__ __attribute__((aligned(16))) float sB1[HN], sW2[HN];
    __shared__ __attribute__((aligned(16))) float sScr[4 * 16 * DHP + 2 * 4 * 2 * 256];   // dh tiles | dX exchange x2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int pair = wave >> 1, half = wave & 1, hb = 64 * half;
    head_stage_weights(p, sW1, sB1, sW2);
    float* dh = sScr + wave * 16 * DHP;
    float* exch = sScr + 4 * 16 * DHP;

    f32x4 accW[4][2], sumW2[4], sumB1[4];
    float gsum = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        accW[mt][0] = accW[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        sumW2[mt] = sumB1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int stride = gridDim.x * 2;
    const int first = blockIdx.x * 2;                                  // pair 0's first tile: sets the trip count
    const int iters = first < p.n_mtiles ? (p.n_mtiles - 1 - first) / stride + 1 : 0;
    int mtile = first + pair;
    f32x4 xa = {0.f, 0.f, 0.f, 0.f}, xb = {0.f, 0.f, 0.f, 0.f};
    float gv = 0.f;
    if (mtile < p.n_mtiles) {
        const int64_t nrow = (int64_t)mtile * 16 + j, row = std::min<int64_t>(nrow, p.T - 1);
        xa = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq);
        xb = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq + 4);
        gv = (nrow < p.T) ? p.g[row] : 0.f;
    }
    for (int it = 0; it < iters; ++it, mtile += stride) {
        const bool active = mtile < p.n_mtiles;                        // wave-uniform; barriers stay unconditional
        const int64_t m0 = (int64_t)mtile * 16;
        const float xs[8] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
        const float g = active ? gv : 0.f;
        if (mtile + stride < p.n_mtiles) {
            const int64_t nrow = (int64_t)(mtile + stride) * 16 + j, row = std::min<int64_t>(nrow, p.T - 1);
            xa = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq);
            xb = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq + 4);
            gv = (nrow < p.T) ? p.g[row] : 0.f;
        }
        float* ex = exch + (it & 1) * (4 * 2 * 256);
        f32x4 acc[4];
        f32x4 accX[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (active) {
            head_gemm1<4>(sW1, hb, j, kq, xs, acc);
            gsum += (kq == 0 && half == 0) ? g : 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&sB1[hb + 16 * mt + 4 * kq]);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&sW2[hb + 16 * mt + 4 * kq]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a, da;
                    head_act<ACT>(acc[mt][r] + bv[r], a, da);
                    const float d = g * wv[r] * da;
                    sumW2[mt][r] = fmaf(g, a, sumW2[mt][r]);
                    sumB1[mt][r] += d;
                    acc[mt][r] = d;
                }
                // stage dh[row j][16mt + 4kq .. +3] (this wave's 64 hidden) for the weight-gradient product
                *reinterpret_cast<f32x4*>(&dh[j * DHP + 16 * mt + 4 * kq]) = acc[mt];
            }
            // (2) partial dX^T tiles over this wave's hidden half; A = W1[hidden hb + 16mt + 4kq + s][16t + j]
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        accX[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            sW1[(hb + 16 * mt + 4 * kq + s) * W1P + 16 * t + j], acc[mt][s], accX[t], 0, 0, 0);
        }
        // hand the tile this wave does NOT store to its partner
        *reinterpret_cast<f32x4*>(&ex[(wave * 2 + (half ^ 1)) * 256 + lane * 4]) = accX[half ^ 1];
        __syncthreads();
        if (active) {
            if (p.dX && m0 + j < p.T) {                                // this wave stores in-features [16 half, +16)
                const f32x4 o = *reinterpret_cast<const f32x4*>(&ex[((wave ^ 1) * 2 + half) * 256 + lane * 4]);
                *reinterpret_cast<f32x4*>(p.dX + (m0 + j) * HK + 16 * half + 4 * kq) = accX[half] + o;
            }
            // (3) dW1 tiles (hidden hb + 16mt.., in 16t..) += dh^T[hidden][row 4kq + s] x[row 4kq + s][in 16t + j]
            float xr[2][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int64_t row = std::min<int64_t>(m0 + 4 * kq + s, p.T - 1);
#pragma unroll
                for (int t = 0; t < 2; ++t) xr[t][s] = p.X[row * HK + 16 * t + j];
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float a = dh[(4 * kq + s) * DHP + 16 * mt + j];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        accW[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xr[t][s], accW[mt][t], 0, 0, 0);
                }
        }
    }
    // per-lane sums over the 16 row lanes (dw2, db1, db2), then waves in order through LDS
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = sumW2[mt][r], b = sumB1[mt][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                b += __shfl_xor(b, o, 64);
            }
            sumW2[mt][r] = a;
            sumB1[mt][r] = b;
        }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) gsum += __shfl_xor(gsum, o, 64);
    __syncthreads();                                   // everybody is done with the scratch tiles
    float* red = sScr;                                 // [HSLAB]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            const bool init = w < 2;                   // waves 0 / 1 own the two hidden halves first
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {     // D: column j = in-feature 16t + j, rows = hidden
                        float* q = &red[(hb + 16 * mt + 4 * kq + r) * HK + 16 * t + j];
                        *q = init ? accW[mt][t][r] : *q + accW[mt][t][r];
                    }
                if (j == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* q2 = &red[HN * HK + hb + 16 * mt + 4 * kq + r];
                        float* q1 = &red[HN * HK + HN + hb + 16 * mt + 4 * kq + r];
                        *q2 = init ? sumW2[mt][r] : *q2 + sumW2[mt][r];
                        *q1 = init ? sumB1[mt][r] : *q1 + sumB1[mt][r];
                    }
                }
            }
            if (lane == 0 && half == 0) {
                float* q = &red[HN * HK + 2 * HN];
                *q = (w == 0) ? gsum : *q + gsum;
            }
        }
        __syncthreads();
    }
    float* out = p.slabs + (int64_t)blockIdx.x * HSLAB;
    for (int e = tid; e < HSLAB; e += 256) out[e] = red[e];
}

// dst = sum over slabs in a fixed order (8 interleaved slab lanes, then the 8 partials in order)
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ slabs, int nslab, float* dW1,
                                                         float* dw2, float* db1, float* db2) {
    __shared__ float part[8][32];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    float a0 = 0.f, a1 = 0.f;
    if (e < HSLAB) {
        int s = sl;
        for (; s + 8 < nslab; s += 16) {
            a0 += slabs[(int64_t)s * HSLAB + e];
            a1 += slabs[(int64_t)(s + 8) * HSLAB + e];
        }
        if (s < nslab) a0 += slabs[(int64_t)s * HSLAB + e];
    }
    part[sl][el] = a0 + a1;
    __syncthreads();
    if (sl != 0 || e >= HSLAB) return;
    float v = part[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) v += part[k][el];
    if (e < HN * HK) dW1[e] = v;
    else if (e < HN * HK + HN) { if (dw2) dw2[e - HN * HK] = v; }
    else if (e < HN * HK + 2 * HN) { if (db1) db1[e - HN * HK - HN] = v; }
    else if (db2) db2[0] = v;
}

static int head_blocks(int64_t T, int tiles_per_block) {
    const int64_t mt = (T + 15) / 16;
    return (int)std::min<int64_t>(512, (mt + tiles_per_block - 1) / tiles_per_block);
}

}  // namespace gt

using namespace gt;

static int head_check(const float* X, int64_t T, int32_t K, int32_t N, int32_t n_out, const float* W1, const float* w2,
                      int32_t act) {
    if (!X || !W1 || !w2 || T <= 0) return GT_EINVAL;
    if (act != GT_ACT_NONE && act != GT_ACT_RELU && act != GT_ACT_SILU) return GT_EINVAL;
    if (K != HK || N != HN || n_out != 1 || T > (int64_t)1 << 34) return GT_ENOTSUP;
    if (reinterpret_cast<uintptr_t>(X) & 15) return GT_EALIGN;
    return 0;
}

extern "C" int gt_mlp_head_fwd(const float* X, int64_t T, int32_t K, int32_t N, int32_t n_out, const float* W1,
                               const float* b1, const float* w2, const float* b2, int32_t act, float* out,
                               void* stream) {
    int rc = head_check(X, T, K, N, n_out, W1, w2, act);
    if (rc) return rc;
    if (!out) return GT_EINVAL;
    HeadP p{X, W1, b1, w2, b2, nullptr, out, nullptr, nullptr, T, (int)((T + 15) / 16)};
    dim3 grid((unsigned)head_blocks(T, 4));
    hipStream_t st = (hipStream_t)stream;
    if (act == GT_ACT_SILU) hipLaunchKernelGGL((head_fwd_kernel<GT_ACT_SILU>), grid, dim3(256), 0, st, p);
    else if (act == GT_ACT_RELU) hipLaunchKernelGGL((head_fwd_kernel<GT_ACT_RELU>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((head_fwd_kernel<GT_ACT_NONE>), grid, dim3(256), 0, st, p);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gt_mlp_head_bwd_ws_bytes(int64_t T) {
    return T > 0 ? (int64_t)head_blocks(T, 2) * HSLAB * (int64_t)sizeof(float) : 0;
}

extern "C" int gt_mlp_head_bwd(const float* X, int64_t T, int32_t K, int32_t N, int32_t n_out, const float* W1,
                               const float* b1, const float* w2, int32_t act, const float* g, float* dX, float* dW1,
                               float* db1, float* dw2, float* db2, void* ws, int64_t ws_bytes, void* stream) {
    int rc = head_check(X, T, K, N, n_out, W1, w2, act);
    if (rc) return rc;
    if (!g || !dW1) return GT_EINVAL;
    if (reinterpret_cast<uintptr_t>(dX) & 15) return GT_EALIGN;
    if (!ws || ws_bytes < gt_mlp_head_bwd_ws_bytes(T)) return GT_EWS;
    const int blocks = head_blocks(T, 2);
    HeadP p{X, W1, b1, w2, nullptr, g, nullptr, dX, reinterpret_cast<float*>(ws), T, (int)((T + 15) / 16)};
    hipStream_t st = (hipStream_t)stream;
    if (act == GT_ACT_SILU) hipLaunchKernelGGL((head_bwd_kernel<GT_ACT_SILU>), dim3(blocks), dim3(256), 0, st, p);
    else if (act == GT_ACT_RELU) hipLaunchKernelGGL((head_bwd_kernel<GT_ACT_RELU>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((head_bwd_kernel<GT_ACT_NONE>), dim3(blocks), dim3(256), 0, st, p);
    GT_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_reduce_kernel, dim3((HSLAB + 31) / 32), dim3(256), 0, st, reinterpret_cast<const float*>(ws),
                       blocks, dW1, dw2, db1, db2);
    GT_LAUNCH_CHECK();
    return 0;
}
