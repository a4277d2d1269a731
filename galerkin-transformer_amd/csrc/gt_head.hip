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
#include <atomic>
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
    const float* gate = nullptr;       // gt_mlp_head_bwd_gated: dX *= silu'(gate) (same layout as X)
};

// dX *= silu'(pre-activation of the layer that produced X): the SiLU backward of that layer on this kernel's store
__device__ __forceinline__ void head_gate4(const float* gp, f32x4& v) {
    const f32x4 gt = *reinterpret_cast<const f32x4*>(gp);
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] *= dsilu_f(gt[t]);
}

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
                f32x4 dxv = accX[half] + o;
                if (p.gate) head_gate4(p.gate + (m0 + j) * HK + 16 * half + 4 * kq, dxv);
                *reinterpret_cast<f32x4*>(p.dX + (m0 + j) * HK + 16 * half + 4 * kq) = dxv;
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

// =====================================================================================================================
// GT_PREC_F16X2 instances (round 4).  The fp32 kernels above are matrix-pipe bound on v_mfma_f32_16x16x4_f32 (forward 21 GF,
// backward 62 GF at ~80 TFLOP/s: 270 / 730 us for 0.36 / 0.67 GB of traffic).  Here every product runs as two fp16 terms
// per operand, three products, on v_mfma_f32_16x16x32_f16 (one instruction contracts 32 indices: the whole K = 32 of
// product (1), 32 hidden units of product (2), 32 rows of product (3)); fp16's range is handled with exact powers of two:
//   (1) H^T = W1 X^T        x: per-ROW exponent (row amax -> [2^14, 2^15)), W1: one exponent for the tensor (taken when the
//                           block stages its fragments); un-scaled by the FMA that adds the bias
//   (2) dX^T = W1^T dh^T    dh: per-row exponent e_j over the wave's 64 hidden units, split once
//   (3) dW1 += dh^T X       contraction over ROWS: the factor must not depend on the row, so row j's dh enters scaled by
//                           2^e_j (the values of (2), stored once, fp32, in the wave's LDS tile) and its x by 2^(c - e_j):
//                           the row scales cancel in every term, c is ONE running exponent per wave (lowered -- with the
//                           accumulators rescaled -- whenever a scaled |x| would reach 2^15, as in gemm_x3w_kernel).  A row
//                           whose terms are small against the wave's largest loses low bits of x in fp16's subnormal
//                           range, i.e. bits below 2^-38 of that largest term.
// Lane layouts are those of the fp32 kernels (lane (j, kq): row j, floats [8kq, 8kq + 8) of it = the B operand of (1) as
// loaded; D registers = hidden 16mt + 4kq + r), so bias / activation / w2 / the sums work unchanged per register.  The
// backward takes 32 rows per wave and trip (two 16-row sub-tiles through (1), (2); one 32-row contraction in (3)).
typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t hu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float h16_pow2(int e) {                       // 2^e, e clamped to the normal range
    e = e < -126 ? -126 : (e > 127 ? 127 : e);
    return __uint_as_float((uint32_t)(e + 127) << 23);
}
// exponent that puts amax (>= 0) into [2^14, 2^15); 0 for amax = 0 / subnormal
__device__ __forceinline__ int h16_exp_for(float amax) {
    const int ex = (int)(__float_as_uint(amax) >> 23);
    return ex == 0 ? 0 : (141 - ex > 126 ? 126 : 141 - ex);      // <= 126: 2^-e stays a normal number too
}
__device__ __forceinline__ void h16_split_pair(float a, float b, float s, uint32_t& hi, uint32_t& lo) {
    // (the four-instruction v_fma_mix* form of gt_common.h was SLOWER here: head_bwd16 485 -> 512 us in the step -- these
    // kernels are bound by the quarter-rate v_exp / v_rcp of SiLU, and opaque asm in their long VALU chains costs scheduling)
    const f32x2 r = f32x2{a, b} * s;
    const hf16x2 h0 = __builtin_convertvector(r, hf16x2);               // v_cvt_pk_f16_f32 (RNE)
    const hf16x2 h1 = __builtin_convertvector(r - __builtin_convertvector(h0, f32x2), hf16x2);
    hi = __builtin_bit_cast(uint32_t, h0);
    lo = __builtin_bit_cast(uint32_t, h1);
}
__device__ __forceinline__ void h16_split8(const float (&v)[8], float s, hf16x8& hi, hf16x8& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) h16_split_pair(v[2 * t], v[2 * t + 1], s, h[t], l[t]);
    hi = __builtin_bit_cast(hf16x8, hu32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(hf16x8, hu32x4{l[0], l[1], l[2], l[3]});
}
// acc += a b with a = ah + al, b = bh + bl (the al bl term is below the resolution of the others): small terms first
__device__ __forceinline__ f32x4 h16_mma3(hf16x8 ah, hf16x8 al, hf16x8 bh, hf16x8 bl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
}

// Stage W1 as fp16 fragments (one exponent for the tensor; returns it): sW1[plane][mt][lane] = the A operand of (1) for
// hidden tile mt, lane (j, kq): W1[16mt + j][8kq .. 8kq + 8); with WT also sWT[plane][t][ti][lane] = the A operand of (2)
// for hidden k-step t (32 units) and in-feature tile ti, lane (m, kq): W1[32t + 16(e >> 2) + 4kq + (e & 3)][16ti + m], e < 8.
template <bool WT>
__device__ __forceinline__ int head16_stage(const HeadP& p, hu32x4* sW1, hu32x4* sWT, float* sB1, float* sW2, float* sRed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[2][8], m = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = tid + 256 * u, mt = f >> 6, j = f & 15, kq = (f & 63) >> 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.W1 + (16 * mt + j) * HK + 8 * kq);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.W1 + (16 * mt + j) * HK + 8 * kq + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[u][e] = a[e];
            v[u][4 + e] = b[e];
            m = fmaxf(m, fmaxf(fabsf(a[e]), fabsf(b[e])));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) sRed[wave] = m;
    if (tid < HN) {
        sB1[tid] = p.b1 ? p.b1[tid] : 0.f;
        sW2[tid] = p.w2[tid];
    }
    __syncthreads();
    const int eW = h16_exp_for(fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3])));
    const float sw = h16_pow2(eW);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        hf16x8 hi, lo;
        h16_split8(v[u], sw, hi, lo);
        sW1[tid + 256 * u] = __builtin_bit_cast(hu32x4, hi);
        sW1[512 + tid + 256 * u] = __builtin_bit_cast(hu32x4, lo);
    }
    if (WT) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = tid + 256 * u, t = f >> 7, ti = (f >> 6) & 1, mm = f & 15, kq = (f & 63) >> 4;
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = p.W1[(32 * t + 16 * (e >> 2) + 4 * kq + (e & 3)) * HK + 16 * ti + mm];
            hf16x8 hi, lo;
            h16_split8(w, sw, hi, lo);
            sWT[f] = __builtin_bit_cast(hu32x4, hi);
            sWT[512 + f] = __builtin_bit_cast(hu32x4, lo);
        }
    }
    __syncthreads();
    return eW;
}

// the row's eight values -> fp16 terms with the row's own exponent (amax over the four kq lanes of the row); returns it
__device__ __forceinline__ int head16_split_row(const float (&xs)[8], hf16x8& xh, hf16x8& xl) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(xs[e]));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const int ex = h16_exp_for(m);
    h16_split8(xs, h16_pow2(ex), xh, xl);
    return ex;
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void head_fwd16_kernel(const HeadP p) {
    __shared__ hu32x4 sW1[2 * 512];
    __shared__ __attribute__((aligned(16))) float sB1[HN], sW2[HN];
    __shared__ float sRed[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int eW = head16_stage<false>(p, sW1, nullptr, sB1, sW2, sRed);
    const float b2v = p.b2 ? p.b2[0] : 0.f;
    const int stride = gridDim.x * 4;
    int mtile = blockIdx.x * 4 + wave;
    f32x4 xa = {0.f, 0.f, 0.f, 0.f}, xb = {0.f, 0.f, 0.f, 0.f};
    if (mtile < p.n_mtiles) {
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
        hf16x8 xh, xl;
        const int ex = head16_split_row(xs, xh, xl);
        const float us = h16_pow2(-ex) * h16_pow2(-eW);
        float part = 0.f;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const hf16x8 wh = __builtin_bit_cast(hf16x8, sW1[mt * 64 + lane]);
            const hf16x8 wl = __builtin_bit_cast(hf16x8, sW1[512 + mt * 64 + lane]);
            const f32x4 acc = h16_mma3(wh, wl, xh, xl, f32x4{0.f, 0.f, 0.f, 0.f});
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&sB1[16 * mt + 4 * kq]);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&sW2[16 * mt + 4 * kq]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a, da;
                head_act<ACT>(fmaf(acc[r], us, bv[r]), a, da);
                part = fmaf(a, wv[r], part);
            }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (kq == 0 && m0 + j < p.T) p.out[m0 + j] = part + b2v;
    }
}

#ifndef H16_ABL
#define H16_ABL 0         // timing ablations of head_bwd16_kernel (wrong results): 1 no (3) * 2 no (2) * 4 no loop barriers * 8 no shuffles * 16 no (1)
#endif
#define H16_SHFL(v, o) ((H16_ABL & 8) ? (v) : __shfl_xor((v), (o), 64))
constexpr int DH16P = 68;                            // floats per row of a wave's dh tile [32 rows][64 hidden]
constexpr int H16_LIMIT = 15, H16_TARGET = 13;       // running exponent of (3): scaled |x| kept below 2^15, reset to 2^13
// dynamic LDS of head_bwd16_kernel: W1 fragments (1) 16 KB | W1^T fragments (2) 16 KB | b1, w2 | dh tiles 4 x 8.5 KB |
// row factors 4 x 32 | dX exchange 8 KB   (76 KB: two blocks per CU)
constexpr int H16_SMEM = 4 * 512 * 16 + (2 * HN + 16) * 4 + 4 * 32 * DH16P * 4 + 4 * 32 * 4 + 4 * 2 * 256 * 4;

template <int ACT>
__global__ __launch_bounds__(256, 2) void head_bwd16_kernel(const HeadP p) {
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    hu32x4* sW1 = reinterpret_cast<hu32x4*>(hsm);
    hu32x4* sWT = sW1 + 1024;
    float* sB1 = reinterpret_cast<float*>(sWT + 1024);
    float* sW2 = sB1 + HN;
    float* sRed = sW2 + HN;
    float* sDh = sRed + 16;
    float* sF = sDh + 4 * 32 * DH16P;
    float* sEx = sF + 4 * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int pair = wave >> 1, half = wave & 1, hb = 64 * half;
    const int eW = head16_stage<true>(p, sW1, sWT, sB1, sW2, sRed);
    const float usW = h16_pow2(-eW);
    float* dh = sDh + wave * 32 * DH16P;
    float* fr = sF + wave * 32;

    f32x4 accW[4][2], sumW2[4], sumB1[4];
    float gsum = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        accW[mt][0] = accW[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        sumW2[mt] = sumB1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int cexp = 100;                                  // running exponent of (3): the first non-zero trip sets the real one
    const int n32 = (int)((p.T + 31) / 32);
    const int stride = gridDim.x * 2;
    const int first = blockIdx.x * 2;                // pair 0's first 32-row tile: sets the trip count (barriers inside)
    const int iters = first < n32 ? (n32 - 1 - first) / stride + 1 : 0;
    int tile = first + pair;
    f32x4 xa[2], xb[2];
    float gv[2];
    auto load = [&](int tl) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t nrow = (int64_t)tl * 32 + 16 * u + j, row = std::min<int64_t>(nrow, p.T - 1);
            xa[u] = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq);
            xb[u] = *reinterpret_cast<const f32x4*>(p.X + row * HK + 8 * kq + 4);
            gv[u] = (nrow < p.T) ? p.g[row] : 0.f;
        }
    };
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        xa[u] = xb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        gv[u] = 0.f;
    }
    if (tile < n32) load(tile);
    for (int it = 0; it < iters; ++it, tile += stride) {
        const bool active = tile < n32;              // wave-uniform; barriers stay unconditional
        const int64_t m0 = (int64_t)tile * 32;
        // the two sub-tiles' rows -> fp16 terms (per-row exponent); then their registers take the next trip's rows
        hf16x8 xh[2], xl[2];
        float us[2], xm[2], g[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float xs[8] = {xa[u][0], xa[u][1], xa[u][2], xa[u][3], xb[u][0], xb[u][1], xb[u][2], xb[u][3]};
            float m = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(xs[e]));
            m = fmaxf(m, H16_SHFL(m, 16));
            m = fmaxf(m, H16_SHFL(m, 32));
            const int ex = h16_exp_for(m);
            h16_split8(xs, h16_pow2(ex), xh[u], xl[u]);
            us[u] = h16_pow2(-ex) * usW;
            xm[u] = m;
            g[u] = active ? gv[u] : 0.f;
        }
        if (tile + stride < n32) load(tile + stride);
        f32x4 own[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float rowmag[2] = {0.f, 0.f};                // |x|max 2^-e_j of this lane's row, per sub-tile
        int edh[2] = {100000, 100000};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 accX[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (active) {
                gsum += (kq == 0 && half == 0) ? g[u] : 0.f;
                float d[4][4], dm = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int ft = (hb >> 4) + mt;   // hidden tile of the tensor
                    const hf16x8 wh = __builtin_bit_cast(hf16x8, sW1[ft * 64 + lane]);
                    const hf16x8 wl = __builtin_bit_cast(hf16x8, sW1[512 + ft * 64 + lane]);
                    const f32x4 acc = (H16_ABL & 16) ? f32x4{xm[u], g[u], xm[u], g[u]}
                                                     : h16_mma3(wh, wl, xh[u], xl[u], f32x4{0.f, 0.f, 0.f, 0.f});
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(&sB1[hb + 16 * mt + 4 * kq]);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(&sW2[hb + 16 * mt + 4 * kq]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a, da;
                        head_act<ACT>(fmaf(acc[r], us[u], bv[r]), a, da);
                        const float dv = g[u] * wv[r] * da;
                        sumW2[mt][r] = fmaf(g[u], a, sumW2[mt][r]);
                        sumB1[mt][r] += dv;
                        d[mt][r] = dv;
                        dm = fmaxf(dm, fabsf(dv));
                    }
                }
                // the row's exponent over this wave's 64 hidden units; the scaled values serve (2) from registers and
                // (3) from the LDS tile
                dm = fmaxf(dm, H16_SHFL(dm, 16));
                dm = fmaxf(dm, H16_SHFL(dm, 32));
                const int ed = h16_exp_for(dm);
                const float sd = h16_pow2(ed);
                // a row without gradient (padding, or every unit switched off) takes no part in (3): factor 0, and it
                // must not set the wave's exponent
                const bool live = (__float_as_uint(dm) >> 23) != 0;
                edh[u] = live ? ed : 100000;
                rowmag[u] = live ? xm[u] * h16_pow2(-ed) : 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[mt][r] *= sd;
                    *reinterpret_cast<f32x4*>(&dh[(16 * u + j) * DH16P + 16 * mt + 4 * kq]) =
                        f32x4{d[mt][0], d[mt][1], d[mt][2], d[mt][3]};
                }
                // (2) partial dX^T over this wave's hidden half: k-step t = hidden tiles 2t, 2t + 1 of the half
#pragma unroll
                for (int t = 0; t < ((H16_ABL & 2) ? 0 : 2); ++t) {
                    const float dv8[8] = {d[2 * t][0], d[2 * t][1], d[2 * t][2], d[2 * t][3],
                                          d[2 * t + 1][0], d[2 * t + 1][1], d[2 * t + 1][2], d[2 * t + 1][3]};
                    hf16x8 dhi, dlo;
                    h16_split8(dv8, 1.f, dhi, dlo);
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti) {
                        const int f = ((2 * half + t) * 2 + ti) * 64 + lane;
                        const hf16x8 wh = __builtin_bit_cast(hf16x8, sWT[f]);
                        const hf16x8 wl = __builtin_bit_cast(hf16x8, sWT[512 + f]);
                        accX[ti] = h16_mma3(wh, wl, dhi, dlo, accX[ti]);
                    }
                }
                const float ux = h16_pow2(-ed) * usW;
                accX[0] *= ux;
                accX[1] *= ux;
            }
            // hand the in-feature tile this wave does NOT store to its partner, keep the other
            *reinterpret_cast<f32x4*>(&sEx[(wave * 2 + u) * 256 + lane * 4]) = half ? accX[0] : accX[1];
            own[u] = half ? accX[1] : accX[0];
        }
        // x of product (3), transposed: rows 8kq + e of the tile, in-feature 16ti + j (the lines are in L1 / L2: this wave
        // and its partner have just read them); requested here, consumed behind the barrier
        float xr[2][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t row = std::min<int64_t>(m0 + 8 * kq + e, p.T - 1);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) xr[ti][e] = (active && !(H16_ABL & 1)) ? p.X[row * HK + 16 * ti + j] : 0.f;
        }
        if (!(H16_ABL & 4)) __syncthreads();
        if (active && p.dX) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (m0 + 16 * u + j < p.T) {         // this wave stores in-features [16 half, +16)
                    const f32x4 o = *reinterpret_cast<const f32x4*>(&sEx[((wave ^ 1) * 2 + u) * 256 + lane * 4]);
                    f32x4 dxv = own[u] + o;
                    if (p.gate) head_gate4(p.gate + (m0 + 16 * u + j) * HK + 16 * half + 4 * kq, dxv);
                    *reinterpret_cast<f32x4*>(p.dX + (m0 + 16 * u + j) * HK + 16 * half + 4 * kq) = dxv;
                }
        }
        if (active && !(H16_ABL & 1)) {
            // (3): one exponent for the wave.  mag = largest |x| 2^-e_j of the 32 rows
            float mag = fmaxf(rowmag[0], rowmag[1]);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mag = fmaxf(mag, H16_SHFL(mag, o));
            const int xe = (int)(__float_as_uint(mag) >> 23);
            if (xe != 0 && xe + cexp - 127 >= H16_LIMIT) {             // wave-uniform
                const int cn = H16_TARGET + 127 - xe, dlt = cn - cexp;
                const float f = dlt < -126 ? 0.f : h16_pow2(dlt);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    accW[mt][0] *= f;
                    accW[mt][1] *= f;
                }
                cexp = cn;
            }
            // row factors 2^(c - e_j): the kq = 0 lane of a row writes its two
            if (kq == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int fe = cexp - edh[u];
                    fr[16 * u + j] = fe < -126 ? 0.f : h16_pow2(fe);
                }
            }
            const f32x4 f0 = *reinterpret_cast<const f32x4*>(&fr[8 * kq]);
            const f32x4 f1 = *reinterpret_cast<const f32x4*>(&fr[8 * kq + 4]);
            const float fv[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
            hf16x8 bh[2], bl[2];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                float w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = xr[ti][e] * fv[e];
                h16_split8(w, 1.f, bh[ti], bl[ti]);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float a8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] = dh[(8 * kq + e) * DH16P + 16 * mt + j];
                hf16x8 ah, al;
                h16_split8(a8, 1.f, ah, al);
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) accW[mt][ti] = h16_mma3(ah, al, bh[ti], bl[ti], accW[mt][ti]);
            }
        }
        if (!(H16_ABL & 4)) __syncthreads();         // the partner has read the exchange tile
    }
    // un-scale the weight-gradient accumulators
    {
        const float uc = h16_pow2(-cexp);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            accW[mt][0] *= uc;
            accW[mt][1] *= uc;
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
    float* red = sDh;                                  // [HSLAB] (17 KB of the 34 KB of dh tiles)
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

// GT_PREC_F16X2 runs the two-term fp16 kernels; every other value of `precision` the fp32-MFMA ones (bit-exact fp32
// products: what GT_PREC_F32 promises, and what the bf16 modes have always used here).  GT_HEAD_F16=0: fp32 kernels always.
static bool head_use_f16(int32_t precision) {
    static const int on = [] { const char* e = getenv("GT_HEAD_F16"); return e ? atoi(e) : 1; }();
    return on && precision == GT_PREC_F16X2;
}

extern "C" int gt_mlp_head_fwd(const float* X, int64_t T, int32_t K, int32_t N, int32_t n_out, const float* W1,
                               const float* b1, const float* w2, const float* b2, int32_t act, int32_t precision,
                               float* out, void* stream) {
    int rc = head_check(X, T, K, N, n_out, W1, w2, act);
    if (rc) return rc;
    if (!out || precision < GT_PREC_F32 || precision > GT_PREC_F16X2) return GT_EINVAL;
    HeadP p{X, W1, b1, w2, b2, nullptr, out, nullptr, nullptr, T, (int)((T + 15) / 16)};
    dim3 grid((unsigned)head_blocks(T, 4));
    hipStream_t st = (hipStream_t)stream;
    if (head_use_f16(precision)) {
        if (act == GT_ACT_SILU) hipLaunchKernelGGL((head_fwd16_kernel<GT_ACT_SILU>), grid, dim3(256), 0, st, p);
        else if (act == GT_ACT_RELU) hipLaunchKernelGGL((head_fwd16_kernel<GT_ACT_RELU>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((head_fwd16_kernel<GT_ACT_NONE>), grid, dim3(256), 0, st, p);
        GT_LAUNCH_CHECK();
        return 0;
    }
    if (act == GT_ACT_SILU) hipLaunchKernelGGL((head_fwd_kernel<GT_ACT_SILU>), grid, dim3(256), 0, st, p);
    else if (act == GT_ACT_RELU) hipLaunchKernelGGL((head_fwd_kernel<GT_ACT_RELU>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((head_fwd_kernel<GT_ACT_NONE>), grid, dim3(256), 0, st, p);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gt_mlp_head_bwd_ws_bytes(int64_t T) {
    return T > 0 ? (int64_t)head_blocks(T, 2) * HSLAB * (int64_t)sizeof(float) : 0;
}

template <typename Kern>
static int head16_allow_lds(Kern k) {                 // 76 KB of dynamic LDS: opt in once per instance
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, H16_SMEM);
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int gt_mlp_head_bwd(const float* X, int64_t T, int32_t K, int32_t N, int32_t n_out, const float* W1,
                               const float* b1, const float* w2, int32_t act, int32_t precision, const float* g,
                               float* dX, float* dW1, float* db1, float* dw2, float* db2, void* ws, int64_t ws_bytes,
                               void* stream) {
    return gt_mlp_head_bwd_gated(X, T, K, N, n_out, W1, b1, w2, act, precision, g, dX, nullptr, dW1, db1, dw2, db2, ws,
                                 ws_bytes, stream);
}

extern "C" int gt_mlp_head_bwd_gated(const float* X, int64_t T, int32_t K, int32_t N, int32_t n_out, const float* W1,
                                     const float* b1, const float* w2, int32_t act, int32_t precision, const float* g,
                                     float* dX, const float* dx_gate, float* dW1, float* db1, float* dw2, float* db2,
                                     void* ws, int64_t ws_bytes, void* stream) {
    int rc = head_check(X, T, K, N, n_out, W1, w2, act);
    if (rc) return rc;
    if (dx_gate && !dX) return GT_EINVAL;
    if (reinterpret_cast<uintptr_t>(dx_gate) & 15) return GT_EALIGN;
    if (!g || !dW1 || precision < GT_PREC_F32 || precision > GT_PREC_F16X2) return GT_EINVAL;
    if (reinterpret_cast<uintptr_t>(dX) & 15) return GT_EALIGN;
    if (!ws || ws_bytes < gt_mlp_head_bwd_ws_bytes(T)) return GT_EWS;
    hipStream_t st = (hipStream_t)stream;
    if (head_use_f16(precision)) {                   // 32 rows per wave pair and trip, two pairs per block
        const int64_t n32 = (T + 31) / 32;
        const int blocks16 = (int)std::min<int64_t>(512, (n32 + 1) / 2);
        HeadP q{X, W1, b1, w2, nullptr, g, nullptr, dX, reinterpret_cast<float*>(ws), T, (int)((T + 15) / 16), dx_gate};
        // the opt-in is per DEVICE: one bit per device ordinal (a process driving several GPUs)
        static std::atomic<uint64_t> raised{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t dbit = 1ull << (dev & 63);
        if (!(raised.load(std::memory_order_acquire) & dbit)) {
            if (head16_allow_lds(head_bwd16_kernel<GT_ACT_NONE>) | head16_allow_lds(head_bwd16_kernel<GT_ACT_RELU>) |
                head16_allow_lds(head_bwd16_kernel<GT_ACT_SILU>))
                return GT_ENOTSUP;
            raised.fetch_or(dbit, std::memory_order_release);
        }
        if (act == GT_ACT_SILU)
            hipLaunchKernelGGL((head_bwd16_kernel<GT_ACT_SILU>), dim3(blocks16), dim3(256), H16_SMEM, st, q);
        else if (act == GT_ACT_RELU)
            hipLaunchKernelGGL((head_bwd16_kernel<GT_ACT_RELU>), dim3(blocks16), dim3(256), H16_SMEM, st, q);
        else hipLaunchKernelGGL((head_bwd16_kernel<GT_ACT_NONE>), dim3(blocks16), dim3(256), H16_SMEM, st, q);
        GT_LAUNCH_CHECK();
        hipLaunchKernelGGL(head_reduce_kernel, dim3((HSLAB + 31) / 32), dim3(256), 0, st, reinterpret_cast<const float*>(ws),
                           blocks16, dW1, dw2, db1, db2);
        GT_LAUNCH_CHECK();
        return 0;
    }
    const int blocks = head_blocks(T, 2);
    HeadP p{X, W1, b1, w2, nullptr, g, nullptr, dX, reinterpret_cast<float*>(ws), T, (int)((T + 15) / 16), dx_gate};
    if (act == GT_ACT_SILU) hipLaunchKernelGGL((head_bwd_kernel<GT_ACT_SILU>), dim3(blocks), dim3(256), 0, st, p);
    else if (act == GT_ACT_RELU) hipLaunchKernelGGL((head_bwd_kernel<GT_ACT_RELU>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((head_bwd_kernel<GT_ACT_NONE>), dim3(blocks), dim3(256), 0, st, p);
    GT_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_reduce_kernel, dim3((HSLAB + 31) / 32), dim3(256), 0, st, reinterpret_cast<const float*>(ws),
                       blocks, dW1, dw2, db1, db2);
    GT_LAUNCH_CHECK();
    return 0;
}
