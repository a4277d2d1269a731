// Split-operand bf16-MFMA GEMM kernel: fp32 operands, fp32 accumulation, fp32-equivalent results at the bf16
// matrix rate (v_mfma_f32_32x32x16_bf16 = 16x the flops per cycle of v_mfma_f32_16x16x4_f32).
//
// Every fp32 operand value a is split EXACTLY into up to three bf16 planes while it is staged into LDS,
//     h0 = bf16_rne(a),  h1 = bf16_rne(a - h0),  h2 = bf16_rne(a - h0 - h1)         (both subtractions are exact)
// so a = h0 + h1 + h2 up to 2^-24 |a| (three 8-bit significands cover the 24 bits of an fp32).  The product of two
// split operands is accumulated plane pair by plane pair into ONE fp32 accumulator, smallest terms first:
//     PLANES = 3 (GT_PREC_BF16X3):  a2 b0 + a1 b1 + a0 b2  (2^-16)  +  a1 b0 + a0 b1  (2^-8)  +  a0 b0
//                                   -- 6 MFMAs; dropped terms a1 b2, a2 b1, a2 b2 are <= 2^-23 |a||b|, i.e. the
//                                   rounding class of an fp32 FMA chain: this is the mode that meets the 1e-5 gate.
//     PLANES = 2 (GT_PREC_BF16X2):  a1 b0 + a0 b1 + a0 b0     -- 3 MFMAs, ~2^-16 relative (between bf16 and fp32)
//     PLANES = 1 (GT_PREC_BF16)  :  a0 b0                     -- 1 MFMA, operands rounded to bf16 (throughput mode)
// A bf16 x bf16 product is exact in fp32, so the only roundings are the accumulator's.
//
// Geometry: 256 threads = 2 x 2 waves, block tile 128 x 128, wave tile 64 x 64 = 2 x 2 MFMA 32x32 accumulators.
// One LDS stage = 16 k (one MFMA k-step), double-buffered; per operand and plane an image [128 rows][16 k] bf16
// with a 48-byte row pitch: the ds_write_b128 of a staging thread (its 8 consecutive k of one row) and the
// ds_read_b128 of an MFMA lane (row = lane & 31, k-half = lane >> 5) are both bank-conflict-free.
// The MFMA's "A" operand is the N-side (weight) tile and its "B" operand the M-side tile, so the 32x32 result
// registers of a lane are ONE output row m and four groups of 4 consecutive columns n -> 16-byte stores through the
// same fused epilogue as the fp32 kernels (ep_row).
// Loader: k-contiguous operands (L = 0) are read as two float4 per thread, x-contiguous ones (L = 1) as eight
// coalesced dword loads (64 consecutive rows per wave instruction); the dropout mask of the A prologue, the
// row-sum by-product (bias gradients), split-K, batching and the second accumulated product are those of gt_gemm.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "gt_gemm_core.h"

namespace gt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 16, X3_PITCH = 48;       // bytes per LDS row (16 bf16 + pad)
constexpr int X3_PLANE = X3_BM * X3_PITCH;                               // 6144 B

// two fp32 -> PLANES packed bf16 pairs (exact residual chain, see the header comment)
template <int PLANES>
__device__ __forceinline__ void split_pair(float a, float b, uint32_t (&out)[PLANES]) {
    f32x2 r = {a, b};
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl) {
        const bf16x2 h = __builtin_convertvector(r, bf16x2);            // v_cvt_pk_bf16_f32 (RNE)
        out[pl] = __builtin_bit_cast(uint32_t, h);
        if (pl + 1 < PLANES) r = r - __builtin_convertvector(h, f32x2);
    }
}

// 8 consecutive k (k0 .. k0+7) of operand row x:  L == 0: base[x*ld + k],  L == 1: base[k*ld + x].
// Branch-free: out-of-range elements are redirected to a device zero, so every lane issues the same loads and no
// s_waitcnt lands between the loads and the MFMAs of the stage being computed (a divergent loader makes hipcc drain
// vmcnt at the join, i.e. BEFORE the MFMAs it should overlap with).  `whole` (block-uniform): the stage lies inside
// [.., kend) and the operand is 16-byte aligned, so a k-contiguous row is two dwordx4 loads.
__device__ __attribute__((aligned(16))) float x3_zero[4] = {0.f, 0.f, 0.f, 0.f};

template <int L>
__device__ __forceinline__ void x3_load8(const float* __restrict__ base, int64_t ld, int x, int X, int k0, int kend,
                                         bool whole, float (&v)[8]) {
    const bool row_ok = x < X;
    if (L == 0) {
        const float* ptr = base + (int64_t)x * ld + k0;
        if (whole) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(row_ok ? ptr : x3_zero);
            const f32x4 b = *reinterpret_cast<const f32x4*>(row_ok ? ptr + 4 : x3_zero);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
            v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *((row_ok && k0 + j < kend) ? ptr + j : x3_zero);
        }
    } else {
        const float* ptr = base + (int64_t)k0 * ld + x;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *((row_ok && (whole || k0 + j < kend)) ? ptr + (int64_t)j * ld : x3_zero);
    }
}

// stateless dropout mask of the A prologue on the 8 staged values (same mask index as gload in gt_gemm_core.h)
template <int L>
__device__ __forceinline__ void x3_mask8(const DropDev& dd, uint32_t dkey, int64_t dld, int64_t dboff, int x, int k0,
                                         float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t di = dboff + (L == 0 ? (int64_t)x * dld + k0 + j : (int64_t)(k0 + j) * dld + x);
        v[j] *= drop_mul(dd, dkey, (uint32_t)di);
    }
}

template <int PLANES>
__device__ __forceinline__ void x3_store8(char* __restrict__ img, int row, int khalf, const float (&v)[8]) {
    uint32_t q[4][PLANES];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pair<PLANES>(v[2 * i], v[2 * i + 1], q[i]);
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
        *reinterpret_cast<u32x4*>(img + pl * X3_PLANE + row * X3_PITCH + khalf * 16) =
            u32x4{q[0][pl], q[1][pl], q[2][pl], q[3][pl]};
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef GT_ABL_X3_NOMFMA          // ablation build: keep the operands live, skip the matrix pipe
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Sign-alternating accumulation (GT_X3_ALT, default on).  Measured on gfx950 (tools/mfma_chain_probe.hip,
// profiles/r05_mfma_chain_probe.json): the bf16 MFMA does not round its sum to nearest -- addends whose low bits fall below
// the accumulator's guard bits are chopped toward -infinity.  Per instruction that is ~2^-9 ulp, but it has ONE direction:
// a chain of six plane products per stage ends ~0.1 of its rms error below the exact sum in EVERY output element, whatever
// the operand signs (K = 1152: mean signed error -2.7e-9 sum|a||b| against an rms of 2.4e-8; negate one operand and the
// mean becomes +2.8e-9; the fp32 MFMA chain: 1e-11).  A coherent offset like that survives every later reduction over
// tokens or pixels that the zero-mean part averages away: it was the 10x excess of the default arithmetic in the
// exact-math gradient parity of the whole model.  The kernels therefore negate the operand rows of
// odd index on both sides (the M-side row in registers, the N-side row at pack / split time; the packed-B kernel, which
// has no register left for a per-lane sign, alternates its M side per 32-row tile instead), so the chain of output
// (m, n) is accumulated with the sign (-1)^(m+n), and undo it on the accumulator before the epilogue: per-element
// accuracy is unchanged, the offset alternates in a checkerboard and cancels in any sum over rows or columns.
#ifndef GT_X3_ALT
#define GT_X3_ALT 1
#endif
// sign of operand row `parity & 1`
__device__ __forceinline__ float x3_alt_sign(int parity) { return (GT_X3_ALT && (parity & 1)) ? -1.f : 1.f; }
// accumulator register e of a lane = output (m = the lane's own row, n = .. + 8 (e >> 2) + 4 lh + (e & 3)): n's parity is e & 1
template <int NI, int NJ>
__device__ __forceinline__ void x3_alt_undo(f32x16 (&acc)[NI][NJ], float rsgn) {
#if GT_X3_ALT
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] *= (e & 1) ? -rsgn : rsgn;
#endif
}

// ---- two-term fp16 arithmetic (GT_PREC_F16X2) --------------------------------------------------------------------------
// fp32-class results from THREE products per stage instead of six: every operand value x is split as
//     h0 = f16_rne(x s),  h1 = f16_rne(x s - h0)          (s a power of two; 11 + 11 significand bits, both steps exact)
// and the products h1 g0 + h0 g1 + h0 g0 are accumulated in fp32 by v_mfma_f32_32x32x16_f16 (dropped: h1 g1 <= 2^-22).
// fp16 has five exponent bits, so the scale s must track the data; no tensor statistics are passed in for that:
//   * N side (the packed weight): x3_pack_b16_kernel takes the amax of each 32-column fragment tile when it packs it and
//     stores the tile's exponent behind the planes -- a wave-uniform factor of one accumulator column block;
//   * M side (activation rows, split in registers): a lane holds ONE row of its 32-row tile (and with the N-side tile as
//     the MFMA's first operand all sixteen accumulator registers of that lane belong to that row), so the scale is a
//     PER-ROW running exponent kept in the lane: before a stage's eight values are split the lane pair of the row takes
//     their amax; if amax 2^e would reach 2^15 the exponent is lowered to put it at 2^13 and the lane's accumulators are
//     multiplied by the same power of two (exact) -- the online-rescaling of a streaming softmax, applied to a dot
//     product.  Nothing can overflow (the check precedes the split), a row whose early stages are its largest simply
//     resolves the later ones relative to that maximum, like any fp32 accumulation does.
// The accumulators are un-scaled together with the GT_X3_ALT sign, before the epilogue.
#ifndef GT_X3H_CVT_SPLIT
#define GT_X3H_CVT_SPLIT 0      // 1: the round-4 split (multiply, v_cvt_pk, two v_cvt, subtract, v_cvt_pk) for A/B timing
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int X3H_E0 = 120;                       // start exponent: any non-zero first stage sets the real one
constexpr int X3H_TARGET = 13, X3H_LIMIT = 15;    // scaled row amax is put in [2^13, 2^14) and kept below 2^15

__device__ __forceinline__ float x3h_pow2(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }   // -126 <= e <= 127

// two fp32 times the scale -> two packed fp16 pairs (round 5: four v_fma_mix* instead of multiply + conversions, gt_common.h)
__device__ __forceinline__ void x3h_split_pair(float a, float b, float s, uint32_t (&out)[2]) {
#if GT_X3H_CVT_SPLIT
    const f32x2 r = f32x2{a, b} * s;
    const f16x2 h0 = __builtin_convertvector(r, f16x2);           // v_cvt_pk_f16_f32 (RNE)
    const f16x2 h1 = __builtin_convertvector(r - __builtin_convertvector(h0, f32x2), f16x2);
    out[0] = __builtin_bit_cast(uint32_t, h0);
    out[1] = __builtin_bit_cast(uint32_t, h1);
#else
    f16_mulsplit_pair(a, s, b, s, out[0], out[1]);
#endif
}

__device__ __forceinline__ f32x16 mfma32h(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// Epilogue shared by both kernels.  Result registers of the 32x32 MFMA with the N-side tile as its A operand: lane
// (lr = lane & 31, lh = lane >> 5) holds output row  mrow + 32 i  of accumulator (i, j) and the four 4-column groups
// ncol + 32 j + 8 g .. + 3  (ncol already includes 4 * lh).
// The 32x32 MFMA leaves a lane with ONE output row and 4-column groups 32 bytes apart, so stores (and the epilogue's
// res / aux / add loads) straight from the accumulator layout touch 32-byte pieces of 32 different rows per
// instruction: rocprofv3 WRITE_SIZE showed 1.4-1.6x the algorithmic bytes on every token GEMM (profiles/
// r02m_pmc_step_summary.txt).  The wave therefore transposes its 32 x 64 row tile through a private LDS tile first:
// afterwards 16 consecutive lanes hold one row's 64 columns and every global access of the fused epilogue is a full
// 256-byte row segment.  mtile0 / ntile0: first row / column of the wave's tile.  stg: 32 x X3_EP_SW floats.
#ifndef GT_X3_EP_FAST                              // 0: every tile through ep_row (the round-2 epilogue), for A/B timing
#define GT_X3_EP_FAST 1
#endif
constexpr int X3_EP_SW = 68;                     // staging row pitch in floats (64 + 4: conflict-free both ways)
constexpr int X3_EP_STG = 32 * X3_EP_SW;

// The epilogue's four bias values of a lane (columns ntile0 + 4 (lane & 15) ..): a kernel that fetches them in front of
// its K loop takes one memory round trip out of every block's epilogue.
__device__ __forceinline__ void x3_bias4(const GemmP& p, int ntile0, int lane, float (&b)[4]) {
    const int nb = ntile0 + 4 * (lane & 15);
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = (p.bias && nb + t < p.N) ? p.bias[nb + t] : 0.f;
}

// Batched form of the fused epilogue for whole, 16-byte aligned tiles (every token GEMM of the hot path).  ep_row handles
// one row segment at a time behind run-time switches: each segment's res / aux load was followed by its use, and the
// s_waitcnt vmcnt(0) in front of that use also waited for the STORES of the segments before it -- sixteen store round
// trips in a row per wave.  A block of the FFN launch spent 15.6 us of its 28.7 us in the epilogue writing 64 KB, and
// 55-60 % of the resident blocks of the chip were in that state at any time (tools/x3p_prof.py, profiles/r03z_*).
// Here a wave works in batches of four segments (16 rows x 256 B): the res / aux loads of batch b + 1 are issued before
// the stores of batch b, the values of a batch are computed together, and nothing ever waits for a store.
template <int MI>
__device__ __forceinline__ void x3_epilogue_fast(const GemmP& p, const f32x16 (&acc)[MI][2], int mtile0, int nb, int lane,
                                                 float* __restrict__ stg, float* __restrict__ C, int z, int b0, int b1,
                                                 const float (&biasv)[4], uint32_t dkey) {
    constexpr int NB = 2 * MI, HB = 4;
    const int lr = lane & 31, lh = lane >> 5, c4 = lane & 15, rsub = lane >> 4;
    const float* resb = p.res ? p.res + b0 * p.r_bs0 + b1 * p.r_bs1 + nb : nullptr;
    const float* auxb = p.aux_op ? p.aux + b0 * p.aux_bs0 + b1 * p.aux_bs1 + nb : nullptr;
    f32x4 rs[HB], ax[HB];
    auto loads = [&](int b) {
        f32x4 (&r)[HB] = rs;
        f32x4 (&a)[HB] = ax;
        if (resb) {
#pragma unroll
            for (int k = 0; k < HB; ++k)
                r[k] = *reinterpret_cast<const f32x4*>(resb + (int64_t)(mtile0 + 16 * b + 4 * k + rsub) * p.ldr);
        }
        if (auxb) {
#pragma unroll
            for (int k = 0; k < HB; ++k)
                a[k] = *reinterpret_cast<const f32x4*>(auxb + (int64_t)(mtile0 + 16 * b + 4 * k + rsub) * p.ldaux);
        }
    };
    loads(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = b >> 1;
        if ((b & 1) == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(stg + lr * X3_EP_SW + 32 * j + 8 * g + 4 * lh) =
                        f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            // wave-private tile, LDS operations of one wave execute in order: a compiler fence + counter wait is enough
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        f32x4 v[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k)
            v[k] = *reinterpret_cast<const f32x4*>(stg + (16 * (b & 1) + 4 * k + rsub) * X3_EP_SW + 4 * c4);
        if (b & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next tile is staged
        const int m0b = mtile0 + 16 * b + rsub;                       // row of segment k: m0b + 4 k
#pragma unroll
        for (int k = 0; k < HB; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) v[k][t] = p.alpha * v[k][t] + biasv[t];
        if (p.act == GT_ACT_DROP_SILU) {     // dropout in front of the SiLU; `pre` = keepscale * silu'(u) (gt_hip.h, ep_row)
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const uint32_t di = (uint32_t)(((int64_t)z * p.M + m0b + 4 * k) * p.drop_ld + p.n_off + nb);
                f32x4 df;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float ks = p.drop.thresh ? drop_mul(p.drop, dkey, di + t) : 1.f;
                    float a, da;
                    silu_both(v[k][t] * ks, a, da);
                    v[k][t] = a;
                    df[t] = ks * da;
                }
                if (p.pre) *reinterpret_cast<f32x4*>(p.pre + ((int64_t)z * p.M + m0b + 4 * k) * p.ldpre + nb) = df;
            }
        } else if (p.act == GT_ACT_SILU2) {     // silu(silu(v)); `pre` = silu'(v) silu'(silu(v)) (gt_hip.h, ep_row)
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                f32x4 df;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float a1, d1, a2, d2;
                    silu_both(v[k][t], a1, d1);
                    silu_both(a1, a2, d2);
                    v[k][t] = a2;
                    df[t] = d1 * d2;
                }
                if (p.pre) *reinterpret_cast<f32x4*>(p.pre + ((int64_t)z * p.M + m0b + 4 * k) * p.ldpre + nb) = df;
            }
        } else if (p.pre) {
#pragma unroll
            for (int k = 0; k < HB; ++k)
                *reinterpret_cast<f32x4*>(p.pre + ((int64_t)z * p.M + m0b + 4 * k) * p.ldpre + nb) = v[k];
        }
        if (p.act == GT_ACT_RELU) {
#pragma unroll
            for (int k = 0; k < HB; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) v[k][t] = fmaxf(v[k][t], 0.f);
        } else if (p.act == GT_ACT_SILU) {
#pragma unroll
            for (int k = 0; k < HB; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) v[k][t] = silu_f(v[k][t]);
        }
        if (auxb) {
            const f32x4 (&a)[HB] = ax;
            if (p.aux_op == GT_AUX_GT0) {
#pragma unroll
                for (int k = 0; k < HB; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[k][t] *= a[k][t] > 0.f ? p.aux_scale : 0.f;
            } else if (p.aux_op == GT_AUX_DSILU) {
#pragma unroll
                for (int k = 0; k < HB; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[k][t] *= dsilu_f(a[k][t]);
            } else {
#pragma unroll
                for (int k = 0; k < HB; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[k][t] *= a[k][t] * p.aux_scale;
            }
        }
        if (p.drop.thresh && p.act != GT_ACT_DROP_SILU) {
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const uint32_t di = (uint32_t)(((int64_t)z * p.M + m0b + 4 * k) * p.drop_ld + p.n_off + nb);
#pragma unroll
                for (int t = 0; t < 4; ++t) v[k][t] *= drop_mul(p.drop, dkey, di + t);
            }
        }
        if (resb) {
            const f32x4 (&r)[HB] = rs;
#pragma unroll
            for (int k = 0; k < HB; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) v[k][t] = r[k][t] + p.out_scale * v[k][t];
        } else {
#pragma unroll
            for (int k = 0; k < HB; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) v[k][t] *= p.out_scale;
        }
        if (b + 1 < NB) loads(b + 1);          // rs / ax are consumed: the next batch's loads go out in front of the stores
#pragma unroll
        for (int k = 0; k < HB; ++k) *reinterpret_cast<f32x4*>(C + (int64_t)(m0b + 4 * k) * p.ldc + nb) = v[k];
        if (p.c2) {                            // gt_gemm_desc.c_masked: the same rows under the second mask
            const uint32_t key2 = drop_key_dev(p.drop2);
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const uint32_t di = (uint32_t)(((int64_t)z * p.M + m0b + 4 * k) * p.drop_ld + p.n_off + nb);
                f32x4 w = v[k];
                if (p.drop2.thresh) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) w[t] *= drop_mul(p.drop2, key2, di + t);
                }
                *reinterpret_cast<f32x4*>(p.c2 + (int64_t)(m0b + 4 * k) * p.ldc2 + nb) = w;
            }
        }
    }
}

template <int MI>
__device__ __forceinline__ void x3_epilogue(const GemmP& p, const f32x16 (&acc)[MI][2], int mtile0, int ntile0, int lane,
                                            float* __restrict__ stg, int z, int b0, int b1, int sidx,
                                            const float* bias_pre = nullptr) {     // bias_pre: x3_bias4() of this lane
    const int64_t coff = b0 * p.c_bs0 + b1 * p.c_bs1 + (int64_t)sidx * p.c_split;
    float* __restrict__ C = p.C + coff;
    const uint32_t dkey = drop_key_dev(p.drop);
    const int lr = lane & 31, lh = lane >> 5;
    const int c4 = lane & 15, rsub = lane >> 4;          // read side: 16 lanes per row, 4 rows per instruction
    const int nb = ntile0 + 4 * c4;
    const bool col_ok = nb < p.N, full = nb + 4 <= p.N;
    float biasv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) biasv[t] = bias_pre ? bias_pre[t] : (p.bias && nb + t < p.N) ? p.bias[nb + t] : 0.f;
#if GT_X3_EP_FAST
    if (p.c_vec && !p.raw && !p.rp && !p.add && ntile0 + 64 <= p.N && mtile0 + 32 * MI <= p.M) {   // wave-uniform
        x3_epilogue_fast<MI>(p, acc, mtile0, nb, lane, stg, C, z, b0, b1, biasv, dkey);
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(stg + lr * X3_EP_SW + 32 * j + 8 * g + 4 * lh) =
                    f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        // wave-private tile, LDS operations of one wave execute in order: a compiler fence + counter wait is enough
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = 4 * it + rsub, m = mtile0 + 32 * i + r;
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(stg + r * X3_EP_SW + 4 * c4);
            if (m < p.M && col_ok) {
                float v[4] = {t4[0], t4[1], t4[2], t4[3]};
                ep_row<4>(p, v, biasv, C, m, nb, z, b0, b1, full, dkey);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next row tile overwrites the staging
    }
}

// GT_EP_HEADNORM epilogue (QKV projection + per-head LayerNorm + position columns, see gt_hip.h): same register map
// as x3_epilogue.  A head segment (DK columns) of output row m lies inside this wave's 64 columns and is shared by
// the lane pair (lane, lane ^ 32): each lane holds DK / 2 of its values, so the statistics are a local sum plus ONE
// cross-lane exchange.  Raw projection -> C (16-byte stores).  The head-tile rows ([pos | values | pad], DP floats per
// head, the wave's 64 / DK heads adjacent in memory) are first assembled in a wave-private LDS tile and then written
// as whole 16-byte aligned granules, a row at a time: with the coordinates in front the values sit at an 8-byte
// offset, and storing them straight from the accumulator layout (8-byte pieces of 32 different rows per instruction)
// cost 1.26 GB of HBM writes for 0.77 GB of data (rocprofv3 WRITE_SIZE, profiles/r02_pmc_step.json).
constexpr int X3_HN_STG = 32 * 88;               // floats of staging per wave: 32 rows x (4 heads x DP 20 + pad) max

template <int DK, int MI>
__device__ __forceinline__ void x3_epilogue_hn(const GemmP& p, const f32x16 (&acc)[MI][2], int mrow, int ncol, int lane,
                                               float* __restrict__ stg) {
    constexpr int NSEG = 64 / DK, GPS = DK / 8;              // segments per wave row; 4-column groups per lane per segment
    const int lr = lane & 31, lh = lane >> 5;
    const int nwave = ncol - 4 * lh;                          // first column of this wave's 64 (a multiple of 64)
    if (nwave >= p.N) return;                                 // wave-uniform: N is a multiple of 64 here
    // head slots (DK = 64 only): the head occupies the first DKR = hn_dkr (48) columns of its 64-column slot, the accumulators
    // of the 16 columns behind it are exact zeros (zero rows of the packed weight).  GPR: the lane's real 4-column groups.
    const int DKR = (DK == 64) ? p.hn_dkr : DK, GPR = DKR >> 3;
    const float inv = 1.f / (float)DKR;
    const int DP = p.hn_DP, W = NSEG * DP, sw = W + 4, W4 = W >> 2;
    const int stream = nwave / (p.hn_h * DK), head0 = (nwave / DK) % p.hn_h;
    const bool normed = (p.hn_mask >> stream) & 1;
    const bool store_raw = !((p.hn_skip_raw >> stream) & 1);   // the raw projection of this stream goes to C
    const int ni = __popc(p.hn_mask & ((1 << stream) - 1));
    // What the tile loop reads from memory (bias, the rows' coordinates) is fetched here, in front of the first store: a
    // load inside the loop is followed by its use, and the s_waitcnt vmcnt(0) in front of that use also waits for every
    // store issued before it (the serialisation x3_epilogue_fast removes from the plain epilogue).  gamma / beta stay in
    // the loop: the hot path writes plain tiles (hn_plain), and 64 more registers would spill.
    f32x4 bv[NSEG][GPS];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
        for (int q = 0; q < GPS; ++q)
            bv[sg][q] = (p.bias && q < GPR) ? *reinterpret_cast<const f32x4*>(p.bias + (nwave / DK + sg) * DKR + 8 * q + 4 * lh)
                                             : f32x4{0.f, 0.f, 0.f, 0.f};
    // granule walk of the tile store below: lane's first granule (row, 16-byte column) and the step of 64 granules
    const int g_r0 = lane / W4, g_c0 = lane - g_r0 * W4, g_dr = 64 / W4, g_dc = 64 - g_dr * W4;
    const int nit = (32 * W4 + 63) >> 6, rstride = p.hn_h * DP;
    float posv[MI][4];                                        // hn_p <= 4 coordinates of this lane's rows (lane half 0 writes them)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            posv[i][jj] = (lh == 0 && jj < p.hn_p && mrow + 32 * i < p.M) ? p.hn_pos[(int64_t)(mrow + 32 * i) * p.hn_p + jj] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mrow + 32 * i;
        const bool row_ok = m < p.M;
        float* srow = stg + lr * sw;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            const int head = head0 + sg;
            float v[GPS][4];
#pragma unroll
            for (int q = 0; q < GPS; ++q) {
                const int c = sg * DK + 8 * q;                // column offset of this group inside the wave's 64 (+ 4 lh)
                const int j = c >> 5, g = (c & 31) >> 3;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[q][t] = p.alpha * acc[i][j][4 * g + t] + bv[sg][q][t];
                if (row_ok && store_raw && q < GPR)          // (column of the caller's [M, 3 h DKR] projection)
                    *reinterpret_cast<f32x4*>(p.C + (int64_t)m * p.ldc + (nwave / DK + sg) * DKR + 8 * q + 4 * lh) =
                        f32x4{v[q][0], v[q][1], v[q][2], v[q][3]};
            }
            float mu = 0.f, rstd = 1.f;
            if (normed) {                                     // wave-uniform branch: the exchange below is convergent
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < GPS; ++q)
                    if (q < GPR) sum += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
                sum = xor32_sum(sum);
                mu = sum * inv;
                float ss = 0.f;
#pragma unroll
                for (int q = 0; q < GPS; ++q)
                    if (q < GPR) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) { const float c0 = v[q][t] - mu; ss = fmaf(c0, c0, ss); }
                    }
                ss = xor32_sum(ss);
                rstd = 1.f / sqrtf(ss * inv + p.hn_eps);
            }
            float* seg = srow + sg * DP;
#pragma unroll
            for (int q = 0; q < GPS; ++q) {
                if (q >= GPR) continue;
                const int dim = 8 * q + 4 * lh;
                float y[4] = {v[q][0], v[q][1], v[q][2], v[q][3]};
                if (normed) {
                    if (p.hn_plain) {                         // the product path of the Galerkin layers: no load in the loop
#pragma unroll
                        for (int t = 0; t < 4; ++t) y[t] = (y[t] - mu) * rstd;
                    } else {
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.hn_gamma + (ni * p.hn_h + head) * DKR + dim);
                        const f32x4 bt = *reinterpret_cast<const f32x4*>(p.hn_beta + (ni * p.hn_h + head) * DKR + dim);
#pragma unroll
                        for (int t = 0; t < 4; ++t) y[t] = (y[t] - mu) * rstd * gm[t] + bt[t];
                    }
                }
                float* dst = seg + p.hn_p + dim;
                if ((p.hn_p & 1) == 0) {                      // two 8-byte stores (the coordinates in front shift the head by
                    // hn_p floats: no 16-byte alignment).  Thirty-two rows at a pitch that is a multiple of four floats meet in
                    // eight banks: four scalar stores per group were the kernel's LDS bank conflicts (0.58 - 0.77 of its LDS cycles)
                    *reinterpret_cast<f32x2*>(dst) = f32x2{y[0], y[1]};
                    *reinterpret_cast<f32x2*>(dst + 2) = f32x2{y[2], y[3]};
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) dst[t] = y[t];
                }
            }
            if (lh == 0) {                                    // one lane of the pair: coordinates, padding, statistics
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (jj < p.hn_p) seg[jj] = posv[i][jj];
                for (int jj = p.hn_p + DKR; jj < DP; ++jj) seg[jj] = 0.f;
                if (normed && row_ok)
                    *reinterpret_cast<f32x2*>(p.hn_stats + (((int64_t)ni * p.M + m) * p.hn_h + head) * 2) = f32x2{mu, rstd};
            }
        }
        // the tile is wave-private and LDS operations of one wave execute in order: a compiler fence is enough
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the tile's rows go out as whole 16-byte granules, 64 per instruction (lane -> granule e = lane + 64 it of the
        // 32 x W4 tile, walked incrementally), three instructions' worth of LDS reads in front of their three stores:
        // the straightforward loop (a division and 64-bit address arithmetic per granule, every read waited for before
        // its store) was 45 instructions + an LDS round trip per granule, 18 times per wave
        const int mbase = mrow - lr + 32 * i;
        float* __restrict__ gtile = p.hn_out + (((int64_t)stream * p.M + mbase) * p.hn_h + head0) * DP;
        const int nrows = p.M - mbase < 32 ? p.M - mbase : 32;
        int r = g_r0, c4 = g_c0;
        for (int it = 0; it < nit; it += 3) {
            f32x4 val[3];
            int off[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                off[u] = r < nrows ? r * rstride + 4 * c4 : -1;
                if (r < 32) val[u] = *reinterpret_cast<const f32x4*>(stg + r * sw + 4 * c4);
                r += g_dr; c4 += g_dc;
                if (c4 >= W4) { c4 -= W4; ++r; }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (off[u] >= 0) *reinterpret_cast<f32x4*>(gtile + off[u]) = val[u];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next row tile overwrites the staging
    }
}

template <int LA, int LB, int PLANES>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const GemmP p) {
    constexpr int STAGE = 2 * PLANES * X3_PLANE;                          // A planes then B planes
    constexpr int SMEM = (2 * STAGE > 4 * X3_EP_STG * 4) ? 2 * STAGE : 4 * X3_EP_STG * 4;   // stages, later staging tiles
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    // XCD-aware tile order (same map as gemm_kernel): each XCD walks a contiguous range of tile ids
    int tile;
    {
        const int tiles = gridDim.x, q = tiles >> 3, r = tiles & 7;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * X3_BM, n0 = tn * X3_BN;
    const int z = blockIdx.z, b0 = z / p.batch1, b1 = z % p.batch1;
    const int kbeg = blockIdx.y * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);

    const float* A = p.A + b0 * p.a_bs0 + b1 * p.a_bs1;
    const float* Bm = p.B + b0 * p.b_bs0 + b1 * p.b_bs1;
    int64_t lda_c = p.lda, ldb_c = p.ldb;
    int kend_c = kend, avec_c = p.a_vec, bvec_c = p.b_vec;
    const uint32_t akey = drop_key_dev(p.a_drop);
    const int64_t adoff = (int64_t)z * p.a_drop_bstride;

    // staging role of this thread: one row of each operand tile, one k-half (8 consecutive k) per stage
    const int arow = (LA == 0) ? (tid >> 1) : (tid & 127), akh = (LA == 0) ? (tid & 1) : (tid >> 7);
    const int brow = (LB == 0) ? (tid >> 1) : (tid & 127), bkh = (LB == 0) ? (tid & 1) : (tid >> 7);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra[8], rb[8];
    float asum = 0.f;
    int ka = 0;                                   // first k of the values held in ra (for the mask index)
    const bool do_acs = (LA == 1) && p.acs != nullptr && tn == 0;
    // FAST: every element of the stage is in range along k and k-contiguous rows are 16-byte aligned -- straight-line
    // loads (row validity by pointer select), so the only vmcnt wait of an iteration sits in r2s, after the MFMAs.
    auto g2r = [&](int k0, auto fast) {
        constexpr bool FAST = decltype(fast)::value;
        const bool inside = FAST || (k0 + X3_BK <= kend_c);       // block-uniform
        ka = k0 + 8 * akh;
        x3_load8<LA>(A, lda_c, m0 + arow, p.M, ka, kend_c, inside && (FAST || LA == 1 || avec_c), ra);
        x3_load8<LB>(Bm, ldb_c, n0 + brow, p.N, k0 + 8 * bkh, kend_c, inside && (FAST || LB == 1 || bvec_c), rb);
    };
    auto r2s = [&](int buf) {                     // first use of the loaded registers: the vmcnt wait lands here
        if (p.a_drop.thresh) x3_mask8<LA>(p.a_drop, akey, p.a_drop_ld, adoff, m0 + arow, ka, ra);
        if (LA == 1 && do_acs) asum += ((ra[0] + ra[1]) + (ra[2] + ra[3])) + ((ra[4] + ra[5]) + (ra[6] + ra[7]));
        if (GT_X3_ALT) {                          // odd operand rows enter negated (tile origins are even)
            const float sa_ = x3_alt_sign(arow), sb_ = x3_alt_sign(brow);
#pragma unroll
            for (int j = 0; j < 8; ++j) { ra[j] *= sa_; rb[j] *= sb_; }
        }
        char* st = smem + buf * STAGE;
        x3_store8<PLANES>(st, arow, akh, ra);
        x3_store8<PLANES>(st + PLANES * X3_PLANE, brow, bkh, rb);
    };
    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE;
        const char* sb = sa + PLANES * X3_PLANE;
        bf16x8 am[2][PLANES], bn[2][PLANES];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                am[i][pl] = *reinterpret_cast<const bf16x8*>(sa + pl * X3_PLANE + (wm * 64 + 32 * i + lr) * X3_PITCH + lh * 16);
                bn[i][pl] = *reinterpret_cast<const bf16x8*>(sb + pl * X3_PLANE + (wn * 64 + 32 * i + lr) * X3_PITCH + lh * 16);
            }
        // plane pairs in increasing magnitude; the four accumulators interleave inside every pair
#pragma unroll
        for (int s = PLANES - 1; s >= 0; --s) {          // plane pairs (pa, pb) with pa + pb = s <= PLANES - 1
#pragma unroll
            for (int pa = 0; pa < PLANES; ++pa) {
                const int pb = s - pa;
                if (pb < 0 || pb >= PLANES) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(bn[j][pb], am[i][pa], acc[i][j]);
            }
        }
    };

    const int nk1 = (kend > kbeg) ? (kend - kbeg + X3_BK - 1) / X3_BK : 0;
    const int nk = nk1 + (p.K2 > 0 ? (p.K2 + X3_BK - 1) / X3_BK : 0);
    auto enter_seg2 = [&]() {
        A = p.A2 + b0 * p.a2_bs0 + b1 * p.a2_bs1;
        Bm = p.B2 + b0 * p.b2_bs0 + b1 * p.b2_bs1;
        lda_c = p.lda2; ldb_c = p.ldb2; kend_c = p.K2; avec_c = p.a2_vec; bvec_c = p.b2_vec;
    };
    if (nk > 0) {
        if (nk1 == 0) { enter_seg2(); g2r(0, std::false_type{}); }
        else g2r(kbeg, std::false_type{});
        r2s(0);
    }
    __syncthreads();
    // iterations whose NEXT stage is a whole one of the first product take the branch-free loop
    const bool fast_ok = (LA == 1 || p.a_vec) && (LB == 1 || p.b_vec);
    const int nfast = fast_ok ? max(0, (kend - kbeg) / X3_BK - 1) : 0;
    int kt = 0;
    for (; kt < nfast; ++kt) {
        g2r(kbeg + (kt + 1) * X3_BK, std::true_type{});
        compute(kt & 1);
        r2s((kt + 1) & 1);
        __syncthreads();
    }
    for (; kt + 1 < nk; ++kt) {                   // K tail, unaligned operands, the second product
        if (kt + 1 == nk1) enter_seg2();
        g2r(kt + 1 < nk1 ? kbeg + (kt + 1) * X3_BK : (kt + 1 - nk1) * X3_BK, std::false_type{});
        compute(kt & 1);
        r2s((kt + 1) & 1);
        __syncthreads();
    }
    if (nk > 0) compute((nk - 1) & 1);
    __syncthreads();

    if (LA == 1 && do_acs) {              // uniform per block; the stages are free after the loop's last barrier
        float* part = reinterpret_cast<float*>(smem);
        part[akh * X3_BM + arow] = asum;
        __syncthreads();
        if (tid < X3_BM && m0 + tid < p.M)
            p.acs[((int64_t)blockIdx.y * gridDim.z + z) * p.M + m0 + tid] = part[tid] + part[X3_BM + tid];
    }

    __syncthreads();                               // every wave is done with the stages: they become staging tiles
    x3_alt_undo(acc, x3_alt_sign(lr));
    x3_epilogue<2>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, reinterpret_cast<float*>(smem) + wave * X3_EP_STG, z, b0, b1,
                   (int)blockIdx.y);
}

// =================================================================================================================
// Ring variant for aligned operands: the fp32 tiles go global -> LDS by direct loads (no VGPR staging, no ds_write
// pass) into a ring of R 16-deep stages, R - 1 stages in flight while one is consumed, so an HBM round trip is covered
// by ~R - 1 stage times instead of one; the exact bf16 split happens in registers on the way from LDS into the MFMA
// operands (each wave converts the two A and two B row tiles it multiplies).  One s_barrier per stage publishes the
// landed stage and frees the slot the next request overwrites.
//
// LDS images of one stage (A then B, 8 KB each); a direct load writes wave-uniform base + lane * 16 B, so the images are
// lane-linear and every swizzle is applied to the SOURCE address:
//   k-contiguous operand (L == 0):  [128 rows][16 k]: the 16-byte granule g of row r sits at slot g ^ ((r >> 2) & 3);
//                                   an MFMA lane (row, k-half h) reads granules 2h and 2h + 1 (two conflict-free
//                                   ds_read_b128);
//   x-contiguous operand (L == 1):  [16 k][128 x] as in memory; a lane reads its row's 8 k as 8 ds_read_b32 (lanes of a
//                                   half-wave hit consecutive banks).
// Needs: 16-byte aligned operands and leading dimensions, K % 4 == 0 for k-contiguous operands, X % 4 == 0 for
// x-contiguous ones, no second product (the register-staged kernel above takes everything else).
typedef __attribute__((address_space(3))) void* x3_lds_ptr;
typedef const __attribute__((address_space(1))) void* x3_glb_ptr;
constexpr int X3R_OP = X3_BM * X3_BK * 4;        // 8192 B: one operand tile of one stage
constexpr int X3R_STAGE = 2 * X3R_OP;

template <int L>
__device__ __forceinline__ void x3r_issue(const float* __restrict__ base, int64_t ld, int x0, int X, int k0, int kend,
                                          char* img, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave * 2 + i;                // 1-KiB piece of the 8-KiB image
        const float* src;
        if (L == 0) {
            const int row = 16 * q + (lane >> 2), slot = lane & 3;
            const int g = slot ^ ((row >> 2) & 3);
            const int x = x0 + row, k = k0 + 4 * g;
            src = (x < X && k < kend) ? base + (int64_t)x * ld + k : x3_zero;
        } else {
            const int kr = 2 * q + (lane >> 5), xx = x0 + 4 * (lane & 31), k = k0 + kr;
            src = (k < kend && xx < X) ? base + (int64_t)k * ld + xx : x3_zero;
        }
        __builtin_amdgcn_global_load_lds((x3_glb_ptr)src, (x3_lds_ptr)(img + q * 1024), 16, 0, 0);
    }
}

// this lane's 8 consecutive k (k-half lh) of tile row `row` (0..127) from a stage image
template <int L>
__device__ __forceinline__ void x3r_frag(const char* __restrict__ img, int row, int lh, float (&v)[8]) {
    if (L == 0) {
        const int s = (row >> 2) & 3;
        const f32x4 a = *reinterpret_cast<const f32x4*>(img + row * 64 + (((2 * lh) ^ s) << 4));
        const f32x4 b = *reinterpret_cast<const f32x4*>(img + row * 64 + (((2 * lh + 1) ^ s) << 4));
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
        const float* f = reinterpret_cast<const float*>(img) + (8 * lh) * X3_BM + row;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[j * X3_BM];
    }
}

template <int PLANES>
__device__ __forceinline__ void x3r_split(const float (&v)[8], bf16x8 (&out)[PLANES]) {
    uint32_t q[4][PLANES];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pair<PLANES>(v[2 * i], v[2 * i + 1], q[i]);
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
        out[pl] = __builtin_bit_cast(bf16x8, u32x4{q[0][pl], q[1][pl], q[2][pl], q[3][pl]});
}

// Implicit 3x3 convolution (gt_hip.h: cv_*): the k-contiguous A image of a stage is the 16 channels [c0, c0 + 16) of
// tap (dy, dx) of the tile's 128 pixels -- a lane's granule comes from its pixel's neighbour row, or from x3_zero
// outside the picture.  A stage never straddles two taps (cv_C % 16 == 0).  `ok` = the lane's pixel's 9 tap-valid bits.
// Stage order: the nine taps of one 32-channel block (16 when cv_C % 32 != 0) before the next block -- a pixel's 128-byte
// line is then read by its nine taps within 18 consecutive stages and stays in L2; taps-outermost measured 3.07 GB of
// fabric reads per launch for 0.39 GB of activations (rocprofv3 FETCH_SIZE x 2, profiles/r02q_pmc_step.json).
__device__ __forceinline__ void x3r_issue_conv(const float* const (&rowp)[2], const int (&ok)[2], int tap, int c0, int W,
                                               int64_t ld, char* img, int wave, int lane) {
    const int64_t shift = (int64_t)((tap / 3 - 1) * W + (tap % 3 - 1)) * ld + c0;      // ld = pixel pitch (>= channels)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave * 2 + i;
        const int row = 16 * q + (lane >> 2), g = (lane & 3) ^ ((row >> 2) & 3);
        const float* src = ((ok[i] >> tap) & 1) ? rowp[i] + shift + 4 * g : x3_zero;
        __builtin_amdgcn_global_load_lds((x3_glb_ptr)src, (x3_lds_ptr)(img + q * 1024), 16, 0, 0);
    }
}

// Weight-gradient flavour (cv_wgrad): the x-contiguous B image of a stage is 16 consecutive pixels p of the tap-shifted
// activations, B(p, n) = X[p + (dy, dx)][n]; rows whose neighbour falls outside the picture (or p >= kend) read zero.
// (y, x) = the lane's two pixels of the current stage; W >= 16, so one stage wraps at most one image row.
__device__ __forceinline__ void x3r_issue_convw(const float* __restrict__ X, int64_t C, int n0, int N, int k0, int kend,
                                                const int (&py)[2], const int (&px)[2], int dy, int dx, int H, int W,
                                                char* img, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave * 2 + i;
        const int kr = 2 * q + (lane >> 5), xx = n0 + 4 * (lane & 31), k = k0 + kr;
        const bool ok = k < kend && xx < N && (unsigned)(py[i] + dy) < (unsigned)H && (unsigned)(px[i] + dx) < (unsigned)W;
        const float* src = ok ? X + ((int64_t)k + dy * W + dx) * C + xx : x3_zero;
        __builtin_amdgcn_global_load_lds((x3_glb_ptr)src, (x3_lds_ptr)(img + q * 1024), 16, 0, 0);
    }
}

// HN = head width of the fused head-norm epilogue (0 = general epilogue)
// CV = 0 plain GEMM, 1 implicit 3x3 convolution on A (forward / data gradient), 2 on B, one tap per block (weight gradient)
template <int LA, int LB, int PLANES, int R, int HN = 0, int CV = 0>
__global__ __launch_bounds__(256, (R <= 3 ? 3 : 2)) void gemm_x3r_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) char smem[R * X3R_STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    int tile, by, z;
    if (CV == 2) {
        // 1-D grid: the nine taps of one (K chunk, tile) sit next to each other on ONE XCD, so the dY and X rows they
        // share come out of that XCD's L2 instead of HBM nine times
        const int tiles = p.tiles_m * p.tiles_n;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int ct = (j / 9) * 8 + xcd;
        if (ct >= p.n_split * tiles) return;
        z = j % 9; by = ct / tiles; tile = ct % tiles;
    } else {
        const int tiles = gridDim.x, q = tiles >> 3, r = tiles & 7;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
        by = blockIdx.y; z = blockIdx.z;
    }
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * X3_BM, n0 = tn * X3_BN;
    const int b0 = z / p.batch1, b1 = z % p.batch1;
    const int kbeg = by * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const float* A = p.A + b0 * p.a_bs0 + b1 * p.a_bs1;
    const float* Bm = p.B + b0 * p.b_bs0 + b1 * p.b_bs1;
    const uint32_t akey = drop_key_dev(p.a_drop);
    const int64_t adoff = (int64_t)z * p.a_drop_bstride;
    const bool do_acs = (LA == 1) && p.acs != nullptr && tn == 0 && wn == 0 && CV != 2;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float asum[2] = {0.f, 0.f};
    const float rsgn = x3_alt_sign(lr);

    const int nk = (kend > kbeg) ? (kend - kbeg + X3_BK - 1) / X3_BK : 0;
    // convolution: this lane's two pixels of the A image (fixed over the stages), and the running (tap, channel) of
    // the next stage to request (stages are requested in order)
    const float* cv_row[2] = {A, A};
    int cv_ok[2] = {0, 0}, cv_tap = 0, cv_c0 = 0;
    const int cv_cb = (p.cv_C & 31) ? 16 : 32;
    int cw_y[2] = {0, 0}, cw_x[2] = {0, 0};        // CV == 2: (y, x) of this lane's two k-rows (pixels) of the next stage
    if (CV == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pix = (kbeg + 2 * (wave * 2 + i) + (lane >> 5)) % (p.cv_H * p.cv_W);
            cw_y[i] = pix / p.cv_W;
            cw_x[i] = pix - cw_y[i] * p.cv_W;
        }
    }
    if (CV == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 16 * (wave * 2 + i) + (lane >> 2);
            if (m < p.M) {
                const int pix = m % (p.cv_H * p.cv_W), y = pix / p.cv_W, x = pix - y * p.cv_W;
                int ok = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    ok |= (((unsigned)(y + t / 3 - 1) < (unsigned)p.cv_H) && ((unsigned)(x + t % 3 - 1) < (unsigned)p.cv_W)) << t;
                cv_ok[i] = ok;
                cv_row[i] = A + (int64_t)m * p.lda;
            }
        }
    }
    auto issue = [&](int s) {                      // stage s -> slot s % R : 4 load instructions per wave
        char* st = smem + (s % R) * X3R_STAGE;
        const int k0 = kbeg + s * X3_BK;
        if (CV == 1) {
            x3r_issue_conv(cv_row, cv_ok, cv_tap, cv_c0, p.cv_W, p.lda, st, wave, lane);
            cv_c0 += X3_BK;                        // channel block first, taps second, channel blocks last (gt_hip.h)
            if ((cv_c0 & (cv_cb - 1)) == 0) {
                cv_c0 -= cv_cb;
                if (++cv_tap == 9) { cv_tap = 0; cv_c0 += cv_cb; }
            }
        } else {
            x3r_issue<LA>(A, p.lda, m0, p.M, k0, kend, st, wave, lane);
        }
        if (CV == 2) {
            x3r_issue_convw(p.B, p.ldb, n0, p.N, k0, kend, cw_y, cw_x, z / 3 - 1, z % 3 - 1, p.cv_H, p.cv_W,
                            st + X3R_OP, wave, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i) {          // the next stage's pixels: 16 further along the row-major image
                cw_x[i] += X3_BK;
                if (cw_x[i] >= p.cv_W) {
                    cw_x[i] -= p.cv_W;
                    if (++cw_y[i] == p.cv_H) cw_y[i] = 0;
                }
            }
        } else {
            x3r_issue<LB>(Bm, p.ldb, n0, p.N, k0, kend, st + X3R_OP, wave, lane);
        }
    };
#pragma unroll
    for (int s = 0; s < R - 1; ++s)
        if (s < nk) issue(s);

    for (int kt = 0; kt < nk; ++kt) {
        // stages kt+1 .. min(nk-1, kt+R-2) were requested after stage kt: VMEM retires in order, so "at most
        // 4 * newer loads outstanding" means stage kt has landed for this wave; the barrier extends that to the block
        // and also says everybody is done reading slot (kt-1) % R, which the next request overwrites
        const int newer = min(nk - 1, kt + R - 2) - kt;
        if (R > 3 && newer >= 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + R - 1 < nk) issue(kt + R - 1);

        const char* sa = smem + (kt % R) * X3R_STAGE;
        const char* sb = sa + X3R_OP;
        const int kbase = kbeg + kt * X3_BK + 8 * lh;
        bf16x8 am[2][PLANES], bn[2][PLANES];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v[8];
            const int row = wm * 64 + 32 * i + lr;
            x3r_frag<LA>(sa, row, lh, v);
            if (p.a_drop.thresh) x3_mask8<LA>(p.a_drop, akey, p.a_drop_ld, adoff, m0 + row, kbase, v);
            if (LA == 1 && do_acs) asum[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            if (GT_X3_ALT) {                      // row parity = lane parity on both sides (see x3_alt_undo)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] *= rsgn;
            }
#ifdef GT_ABL_X3_NOSPLIT_A       // ablation builds (tools/ablate_x3.sh): timing only, results are wrong
            for (int pl = 0; pl < PLANES; ++pl) am[i][pl] = __builtin_bit_cast(bf16x8, u32x4{__float_as_uint(v[0]), __float_as_uint(v[2]), __float_as_uint(v[4]), __float_as_uint(v[6])});
#else
            x3r_split<PLANES>(v, am[i]);
#endif
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[8];
            x3r_frag<LB>(sb, wn * 64 + 32 * j + lr, lh, v);
            if (GT_X3_ALT) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] *= rsgn;
            }
#ifdef GT_ABL_X3_NOSPLIT_B
            for (int pl = 0; pl < PLANES; ++pl) bn[j][pl] = __builtin_bit_cast(bf16x8, u32x4{__float_as_uint(v[1]), __float_as_uint(v[3]), __float_as_uint(v[5]), __float_as_uint(v[7])});
#else
            x3r_split<PLANES>(v, bn[j]);
#endif
        }
#pragma unroll
        for (int s = PLANES - 1; s >= 0; --s) {          // plane pairs (pa, pb) with pa + pb = s <= PLANES - 1
#pragma unroll
            for (int pa = 0; pa < PLANES; ++pa) {
                const int pb = s - pa;
                if (pb < 0 || pb >= PLANES) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(bn[j][pb], am[i][pa], acc[i][j]);
            }
        }
    }
#ifdef GT_ABL_X3_NOSTORE
    if (acc[0][0][0] != 12345.678f) return;
#endif

    if (LA == 1 && do_acs) {        // row sums of the (masked) A operand: combine the two k-halves of a row
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float t = asum[i] + __shfl_xor(asum[i], 32, 64);
            const int m = m0 + wm * 64 + 32 * i + lr;
            if (lh == 0 && m < p.M) p.acs[((int64_t)by * gridDim.z + z) * p.M + m] = t;
        }
    }
    x3_alt_undo(acc, rsgn);
    if (HN > 0) {
        __syncthreads();                           // every wave is done with the ring: its first slots become staging
        x3_epilogue_hn<(HN > 0 ? HN : 32), 2>(p, acc, m0 + wm * 64 + lr, n0 + wn * 64 + 4 * lh, lane,
                                               reinterpret_cast<float*>(smem) + wave * X3_HN_STG);
    } else {
        __syncthreads();                           // every wave is done with the ring: its first slots become staging
        x3_epilogue<2>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, reinterpret_cast<float*>(smem) + wave * X3_EP_STG, z, b0,
                       b1, by);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Packed-B variant: when B is small next to A (a weight matrix against many token rows), every block would split the
// same B values again -- the in-situ ablation put the B split alone at 2.5 of 37.5 ms/step (tools/ablate_x3.sh).
// x3_pack_b_kernel splits B ONCE into its three bf16 planes, stored in MFMA fragment order
//     Bp[plane][n-tile of 32][k-stage of 16][lane 0..63] = 8 bf16  (k = 16 ks + 8 (lane >> 5) + e, n = 32 nt + (lane & 31)),
// zero beyond N and K, n-tiles padded to whole 128-column block tiles.  A wave then reads a B fragment of a stage as ONE
// coalesced 1-KB global load straight into the registers the MFMA takes it from (the planes stay L2-resident: 3 x 2 bytes
// per weight), B needs no LDS, and the ring holds A only (8 KB per stage, depth 4).  The fragments of stage kt + 1 are
// requested at the top of iteration kt into a second register set; the counted waits (vector-memory loads retire in
// order) are written out next to the loop.
#ifndef GT_X3P_RING                                // A-ring depth of the packed-B kernel (stages of 8 KB)
#define GT_X3P_RING 4
#endif
constexpr int X3P_R = GT_X3P_RING;
static_assert(X3P_R >= 3 && X3P_R <= 6, "the counted waits below are written out for ring depths 3..6");

// Row of the weight behind tile row n (pad48: the tile rows are 64-column head slots holding 48-wide heads, the 16 rows behind
// a head are zero -- gt_gemm.hip: hn_slots; N counts slot rows then), -1: a zero row.
__device__ __forceinline__ int x3_pack_row(int n, int N, int pad48) {
    if (n >= N) return -1;
    if (!pad48) return n;
    const int j = n & 63;
    return j < 48 ? (n >> 6) * 48 + j : -1;
}

__global__ __launch_bounds__(256) void x3_pack_b_kernel(const float* __restrict__ B, int layout_b, int64_t ldb, int N, int K,
                                                        int NT, int KS, u32x4* __restrict__ out, int pad48) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NT * KS * 64) return;
    const int lane = idx & 63, t = idx >> 6, ks = t % KS, nt = t / KS;
    const int n = nt * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    const int nr = x3_pack_row(n, N, pad48);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        v[e] = (nr >= 0 && k < K) ? (layout_b == 0 ? B[(int64_t)nr * ldb + k] : B[(int64_t)k * ldb + nr]) : 0.f;
        v[e] *= x3_alt_sign(n);                    // GT_X3_ALT: odd rows of the N-side operand enter negated
    }
    uint32_t q[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pair<3>(v[2 * i], v[2 * i + 1], q[i]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        out[(((int64_t)pl * NT + nt) * KS + ks) * 64 + lane] = u32x4{q[0][pl], q[1][pl], q[2][pl], q[3][pl]};
}

// GT_PREC_F16X2: the two fp16 planes of B in the same fragment order, one block per 32-column tile: pass 1 takes the tile's
// amax (its exponent e: amax 2^e in [2^13, 2^14)), pass 2 splits the scaled values.  The exponents follow the planes as NT ints.
__device__ __forceinline__ void x3_pack_b16_tile(const float* __restrict__ B, int layout_b, int64_t ldb, int N, int K, int NT,
                                                 int KS, u32x4* __restrict__ out, int nt, int pad48 = 0) {
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = nt * 32 + (lane & 31);
    const int nr = x3_pack_row(n, N, pad48);
    auto load8 = [&](int ks, float (&v)[8]) {        // this lane's eight k of stage ks (one row n, k contiguous or strided)
        const int k0 = ks * 16 + 8 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            v[e] = (nr >= 0 && k < K) ? (layout_b == 0 ? B[(int64_t)nr * ldb + k] : B[(int64_t)k * ldb + nr]) : 0.f;
        }
    };
    // sixteen waves, a stage each per trip; a wave's stages stay in registers between the two passes when there are at most
    // four of them (K <= 1024), so the weight is read once
    constexpr int KEEP = 4;
    float keep[KEEP][8];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int ks = w + 16 * i;
        if (ks < KS) {
            load8(ks, keep[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(keep[i][e]));
        }
    }
    for (int ks = w + 16 * KEEP; ks < KS; ks += 16) {
        float v[8];
        load8(ks, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) red[w] = amax;
    __syncthreads();
    amax = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, red[i]);
    const int ex = (int)((__float_as_uint(amax) >> 23) & 0xff);
    const int e = ex == 0 ? 0 : X3H_TARGET + 127 - ex;            // amax 2^e in [2^13, 2^14); an all-zero tile keeps 1
    const float sc = x3h_pow2(e) * x3_alt_sign(n);                // GT_X3_ALT: odd rows of the N-side operand enter negated
    auto store = [&](int ks, const float (&v)[8]) {
        uint32_t q[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) x3h_split_pair(v[2 * i], v[2 * i + 1], sc, q[i]);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            out[(((int64_t)pl * NT + nt) * KS + ks) * 64 + lane] = u32x4{q[0][pl], q[1][pl], q[2][pl], q[3][pl]};
    };
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int ks = w + 16 * i;
        if (ks < KS) store(ks, keep[i]);
    }
    for (int ks = w + 16 * KEEP; ks < KS; ks += 16) {
        float v[8];
        load8(ks, v);
        store(ks, v);
    }
    if (tid == 0) reinterpret_cast<int*>(out + (int64_t)2 * NT * KS * 64)[nt] = e;
}
__global__ __launch_bounds__(1024) void x3_pack_b16_kernel(const float* __restrict__ B, int layout_b, int64_t ldb, int N, int K,
                                                           int NT, int KS, u32x4* __restrict__ out, int pad48) {
    x3_pack_b16_tile(B, layout_b, ldb, N, K, NT, KS, out, blockIdx.x, pad48);
}
// All the step's weights in ONE launch (round 5, dispatch diet: a step packed 45 weights in 45 launches of 5 - 8 us): block b
// works tile b - start[e] of entry e.
constexpr int X3_PACK_MANY = 64;
struct PackManyP {
    const float* B[X3_PACK_MANY];
    u32x4* out[X3_PACK_MANY];
    int64_t ldb[X3_PACK_MANY];
    int layout_b[X3_PACK_MANY], N[X3_PACK_MANY], K[X3_PACK_MANY], NT[X3_PACK_MANY], KS[X3_PACK_MANY], start[X3_PACK_MANY + 1];
    unsigned char pad48[X3_PACK_MANY];
    int n;
};
__global__ __launch_bounds__(1024) void x3_pack_b16_many_kernel(const PackManyP p) {
    int e = 0;
    while (e + 1 < p.n && (int)blockIdx.x >= p.start[e + 1]) ++e;
    x3_pack_b16_tile(p.B[e], p.layout_b[e], p.ldb[e], p.N[e], p.K[e], p.NT[e], p.KS[e], p.out[e], (int)blockIdx.x - p.start[e],
                     p.pad48[e]);
}

#ifndef GT_X3P_BLOCKS                              // resident blocks per CU the general instances are compiled for
#define GT_X3P_BLOCKS 3
#endif
#ifdef GT_X3P_PROF                                 // tools/x3p_prof.py: wall-clock stamps (100 MHz) of every block's phases
__device__ unsigned long long x3p_prof[8 * 8192];
extern "C" int gt_debug_x3p_prof(void* dst, long long bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(x3p_prof), (size_t)bytes);
}
#define X3P_STAMP(i) const unsigned long long ts##i = __builtin_amdgcn_s_memrealtime()
#else
#define X3P_STAMP(i)
#endif
// BN = 128: 2 x 2 waves of 64 x 64;  BN = 64 (narrow outputs: the 42 / 44-channel convolutions of the down-scaler, padded to
// 48): 4 x 1 waves of 32 x 64 -- a wave then splits ONE 32-row tile of A per stage for its twelve MFMAs, the same split-to-
// matrix ratio as the wide tile, and a 48-column product wastes a quarter of the tile instead of five eighths.
template <int LA, int HN, int CV, int BN = 128>     // CV: 0 plain, 1 implicit 3x3 convolution on A
__global__ __launch_bounds__(256, ((HN > 0 || LA == 1 || CV == 1) ? 3 : GT_X3P_BLOCKS)) void gemm_x3p_kernel(const GemmP p) {
    constexpr int MI = BN == 64 ? 1 : 2;           // 32-row tiles of A per wave
    static_assert(BN == 128 || (BN == 64 && HN == 0 && LA == 0), "the narrow tile serves plain / convolution launches");
    constexpr int R = X3P_R, PLANES = 3;
    constexpr int STG = (HN > 0 ? X3_HN_STG : X3_EP_STG) * 4 * 4;       // bytes of epilogue staging, four waves
    constexpr int SMEM = R * X3R_OP > STG ? R * X3R_OP : STG;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    X3P_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = BN == 64 ? wave : wave >> 1, wn = BN == 64 ? 0 : wave & 1;
    const int wrow = BN == 64 ? wm * 32 : wm * 64; // first tile row of this wave
    const int lr = lane & 31, lh = lane >> 5;
    int tile;
    {
        const int tiles = gridDim.x, q = tiles >> 3, r = tiles & 7;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * X3_BM, n0 = tn * BN;
    const int kend = p.K;
    const float* A = p.A;
    const uint32_t akey = drop_key_dev(p.a_drop);

    float bias4[4] = {0.f, 0.f, 0.f, 0.f};         // the plain epilogue's bias, fetched under the K loop
    if (HN == 0) x3_bias4(p, n0 + wn * 64, lane, bias4);
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (kend + X3_BK - 1) / X3_BK;
    // GT_X3_ALT in this kernel (no register to spare for a per-lane sign): the M-side sign alternates per 32-ROW TILE -- a
    // compile-time constant of the unrolled tile loop for the 64-row wave tile (a source modifier of the split's first
    // instructions), the wave's parity (a scalar) for the 32-row one -- the N-side per column (x3_pack_b_kernel)
    const float tsgn = x3_alt_sign(BN == 64 ? __builtin_amdgcn_readfirstlane(wm) : 0);
    const float* cv_row[2] = {A, A};
    int cv_ok[2] = {0, 0}, cv_none[2] = {0, 0}, cv_tap = 0, cv_c0 = 0;      // cv_none: a stage past the end of K reads zeros
    const int cv_cb = (p.cv_C & 31) ? 16 : 32;
    if (CV == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 16 * (wave * 2 + i) + (lane >> 2);
            if (m < p.M) {
                const int pix = m % (p.cv_H * p.cv_W), y = pix / p.cv_W, x = pix - y * p.cv_W;
                int ok = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    ok |= (((unsigned)(y + t / 3 - 1) < (unsigned)p.cv_H) && ((unsigned)(x + t % 3 - 1) < (unsigned)p.cv_W)) << t;
                cv_ok[i] = ok;
                cv_row[i] = A + (int64_t)m * p.lda;
            }
        }
    }
    auto issue = [&](int s) {                      // A stage s -> slot s % R : 2 load instructions per wave
        char* st = smem + (s % R) * X3R_OP;
        if (CV == 1) {
            x3r_issue_conv(cv_row, s < nk ? cv_ok : cv_none, cv_tap, cv_c0, p.cv_W, p.lda, st, wave, lane);
            cv_c0 += X3_BK;                        // channel block first, taps second, channel blocks last (gt_hip.h)
            if ((cv_c0 & (cv_cb - 1)) == 0) {
                cv_c0 -= cv_cb;
                if (++cv_tap == 9) { cv_tap = 0; cv_c0 += cv_cb; }
            }
        } else {
            x3r_issue<LA>(A, p.lda, m0, p.M, s * X3_BK, kend, st, wave, lane);
        }
    };
    // this wave's two 32-column B fragments of stage ks: plane pl, fragment j at bbase + pl * bplane + (j * KS + ks) KiB
    // (wave-uniform address in SGPRs + the lane's 16-byte slot).  The loads are inline asm on purpose: hipcc's own
    // scoreboard answers a register load inside this loop with s_waitcnt vmcnt(0) before the MFMAs, which also drains
    // the A stage requested a moment earlier (measured in the ISA); here the wait is the counted one below.
    const int wn_u = __builtin_amdgcn_readfirstlane(wn);
    const char* bbase = reinterpret_cast<const char*>(p.Bp) + (int64_t)((n0 + wn_u * 64) >> 5) * p.bp_KS * 1024;
    const int64_t bplane = (int64_t)p.bp_NT * p.bp_KS * 1024;
    const uint32_t voff = lane * 16;
    // Two register sets for B, one stage apart: B(kt + 1) is requested at the TOP of iteration kt, before the A stage of that
    // iteration, and is consumed one iteration later.  Vector-memory loads retire in order, so a wait for B also waits for
    // every A stage requested before it: with ONE set the loads of B(kt + 1) can only go out behind the MFMAs of B(kt), and
    // their latency (plus that of the A stage requested one iteration earlier) stands in front of every stage's MFMAs --
    // load, matrix and store time of a launch add up instead of overlapping (tools/ablate_x3.sh: 37 + 15 + 30 = 82 us).
    bf16x8 bn0[2][PLANES], bn1[2][PLANES];
    auto loadb = [&](int ks, bf16x8 (&bn)[2][PLANES]) {
#ifdef GT_ABL_X3_NOLOADB         // ablation: the B fragments are fetched for the first two stages only (timing; wrong results)
        if (ks > 1) return;
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
#ifdef GT_ABL_X3_NOSPLIT_B       // ablation: every stage reads the same (cache-resident) fragment
                const char* sp = bbase + pl * bplane + (int64_t)j * p.bp_KS * 1024;
#else
                const char* sp = bbase + pl * bplane + ((int64_t)j * p.bp_KS + ks) * 1024;
#endif
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bn[j][pl]) : "v"(voff), "s"(sp));
            }
    };
    // one 32-row tile of A at a time: its fragment is split and goes through its twelve MFMAs before the next one is read
    // (the three planes of ONE tile are live, not of both: the second B set has to fit under 168 registers)
    auto stage = [&](int kt, bf16x8 (&bn)[2][PLANES]) {
        const char* sa = smem + (kt % R) * X3R_OP;
        const int kbase = kt * X3_BK + 8 * lh;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float v[8];
            bf16x8 am[PLANES];
            const int row = wrow + 32 * i + lr;
            x3r_frag<LA>(sa, row, lh, v);
            if (p.a_drop.thresh) x3_mask8<LA>(p.a_drop, akey, p.a_drop_ld, 0, m0 + row, kbase, v);
            if (GT_X3_ALT && (MI == 1 || (i & 1))) {   // M-side: the sign alternates per 32-row tile (see below)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = MI == 1 ? v[q] * tsgn : -v[q];
            }
#ifdef GT_ABL_X3_NOSPLIT_A       // ablation builds (tools/ablate_x3.sh): timing only, results are wrong
            for (int pl = 0; pl < PLANES; ++pl) am[pl] = __builtin_bit_cast(bf16x8, u32x4{__float_as_uint(v[0]), __float_as_uint(v[2]), __float_as_uint(v[4]), __float_as_uint(v[6])});
#else
            x3r_split<PLANES>(v, am);
#endif
#pragma unroll
            for (int s = PLANES - 1; s >= 0; --s) {      // plane pairs (pa, pb) with pa + pb = s, smallest terms first
#pragma unroll
                for (int pa = 0; pa < PLANES; ++pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb >= PLANES) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(bn[j][pb], am[pa], acc[i][j]);
                }
            }
        }
    };
    // Request order of a wave:  B(0) A(0) .. A(R-2) | B(1) A(R-1) | B(2) A(R) | ...   (A = 2 loads, B = 6).
    // Top of iteration kt >= 1: A(kt) and B(kt) must have landed; the only request behind B(kt) is A(kt+R-2): vmcnt(2).
    // The barrier makes every wave's pieces of A(kt) visible and frees slot (kt-1) % R for the request of A(kt+R-1).
    // Tying the set to the statement keeps its MFMAs behind the wait.  Every iteration issues the same requests -- past
    // the end of K the A loader reads the zero line into a free slot and B re-reads its last stage -- so the counts hold
    // to the last stage and no load sits under a branch: a conditional asm load makes hipcc allocate fresh registers for
    // it and COPY them into the set at the join, before the data has arrived (seen in the ISA of a first version).
#define X3P_WAIT_AB_(N, bn)                                                                                            \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier"                                                                \
                 : "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[0][2]), "+v"(bn[1][0]), "+v"(bn[1][1]), "+v"(bn[1][2])      \
                 :                                                                                                     \
                 : "memory")
#ifdef GT_X3P_PROF                                 // time spent in the waits of the K loop (wave 0)
    unsigned long long tw_sum = 0;
#define X3P_WAIT_AB(N, bn)                                                                                             \
    do {                                                                                                               \
        const unsigned long long w0_ = __builtin_amdgcn_s_memrealtime();                                              \
        X3P_WAIT_AB_(N, bn);                                                                                           \
        tw_sum += __builtin_amdgcn_s_memrealtime() - w0_;                                                              \
    } while (0)
#else
#define X3P_WAIT_AB(N, bn) X3P_WAIT_AB_(N, bn)
#endif
    const int klast = nk - 1;
    loadb(0, bn0);
#pragma unroll
    for (int s = 0; s < R - 1; ++s) issue(s);
    if (R == 3) X3P_WAIT_AB(2, bn0);               // iteration 0: behind A(0) are the R - 2 other stages of the prologue
    else if (R == 4) X3P_WAIT_AB(4, bn0);
    else if (R == 5) X3P_WAIT_AB(6, bn0);
    else X3P_WAIT_AB(8, bn0);
    X3P_STAMP(1);
    loadb(klast < 1 ? klast : 1, bn1);
    issue(R - 1);
    stage(0, bn0);
    int kt = 1;
    for (; kt + 1 < nk; kt += 2) {
        X3P_WAIT_AB(2, bn1);
        loadb(kt + 1, bn0);
        issue(kt + R - 1);
        stage(kt, bn1);
        X3P_WAIT_AB(2, bn0);
        loadb(kt + 2 < klast ? kt + 2 : klast, bn1);
        issue(kt + R);
        stage(kt + 1, bn0);
    }
    if (kt < nk) {                                 // odd stage out (nk even): nothing left to request
        X3P_WAIT_AB(2, bn1);
        stage(kt, bn1);
    }
    X3P_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero stages requested past the end of K: the ring becomes staging
#undef X3P_WAIT_AB
#undef X3P_WAIT_AB_
#ifdef GT_ABL_X3_NOSTORE
    if (acc[0][0][0] != 12345.678f) return;
#endif

#if GT_X3_ALT
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float sg = MI == 1 ? tsgn : ((i & 1) ? -1.f : 1.f);
                acc[i][j][e] *= (e & 1) ? -sg : sg;
            }
#endif
    __syncthreads();                               // every wave is done with the ring: its first slots become staging
    if constexpr (HN > 0)
        x3_epilogue_hn<(HN > 0 ? HN : 32), 2>(p, acc, m0 + wm * 64 + lr, n0 + wn * 64 + 4 * lh, lane,
                                               reinterpret_cast<float*>(smem) + wave * X3_HN_STG);
    else
        x3_epilogue<MI>(p, acc, m0 + wrow, n0 + wn * 64, lane, reinterpret_cast<float*>(smem) + wave * X3_EP_STG, 0, 0, 0, 0,
                        bias4);
#ifdef GT_X3P_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the block's stores have left
    X3P_STAMP(3);
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long* o = x3p_prof + 8 * blockIdx.x;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID: cu / sh / se
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // XCC_ID
        o[6] = tile;
        o[7] = tw_sum;
    }
#endif
}

// The GT_PREC_F16X2 twin of gemm_x3p_kernel (two fp16 planes, three products, see above).  It is a SEPARATE body on purpose:
// folding both arithmetics into one templated body changed hipcc's code for the bf16 instances enough to make the head-norm
// launch return different bits from run to run (16 rows of one head, one launch in four; tools: 40 repeated launches) while
// the instruction stream around its counted waits looked the same -- the bf16 kernel above is therefore textually the one
// that has passed every suite, and this one is gated by its own repeat-launch test.  F16 is always 1 here.
template <int LA, int HN, int CV, int BN, int F16>     // CV: 0 plain, 1 implicit 3x3 convolution on A
__device__ __forceinline__ void x3p_body(const GemmP& p) {
    constexpr int MI = BN == 64 ? 1 : 2;           // 32-row tiles of A per wave
    static_assert(BN == 128 || (BN == 64 && HN == 0 && LA == 0), "the narrow tile serves plain / convolution launches");
    constexpr int R = X3P_R, PLANES = F16 ? 2 : 3;
    constexpr int STG = (HN > 0 ? X3_HN_STG : X3_EP_STG) * 4 * 4;       // bytes of epilogue staging, four waves
    constexpr int SMEM = R * X3R_OP > STG ? R * X3R_OP : STG;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    X3P_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = BN == 64 ? wave : wave >> 1, wn = BN == 64 ? 0 : wave & 1;
    const int wrow = BN == 64 ? wm * 32 : wm * 64; // first tile row of this wave
    const int lr = lane & 31, lh = lane >> 5;
    int tile;
    {
        const int tiles = gridDim.x, q = tiles >> 3, r = tiles & 7;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * X3_BM, n0 = tn * BN;
    const int kend = p.K;
    const float* A = p.A;
    const uint32_t akey = drop_key_dev(p.a_drop);

    float bias4[4] = {0.f, 0.f, 0.f, 0.f};         // the plain epilogue's bias, fetched under the K loop
    if (HN == 0) x3_bias4(p, n0 + wn * 64, lane, bias4);
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (kend + X3_BK - 1) / X3_BK;
    int ea[MI];                                    // F16: running exponent of this lane's row in tile i (see x3h_* above)
#pragma unroll
    for (int i = 0; i < MI; ++i) ea[i] = X3H_E0;
    // GT_X3_ALT in this kernel (no register to spare for a per-lane sign): the M-side sign alternates per 32-ROW TILE -- a
    // compile-time constant of the unrolled tile loop for the 64-row wave tile (a source modifier of the split's first
    // instructions), the wave's parity (a scalar) for the 32-row one -- the N-side per column (x3_pack_b_kernel)
    const float tsgn = x3_alt_sign(BN == 64 ? __builtin_amdgcn_readfirstlane(wm) : 0);
    const float* cv_row[2] = {A, A};
    int cv_ok[2] = {0, 0}, cv_none[2] = {0, 0}, cv_tap = 0, cv_c0 = 0;      // cv_none: a stage past the end of K reads zeros
    const int cv_cb = (p.cv_C & 31) ? 16 : 32;
    if (CV == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 16 * (wave * 2 + i) + (lane >> 2);
            if (m < p.M) {
                const int pix = m % (p.cv_H * p.cv_W), y = pix / p.cv_W, x = pix - y * p.cv_W;
                int ok = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    ok |= (((unsigned)(y + t / 3 - 1) < (unsigned)p.cv_H) && ((unsigned)(x + t % 3 - 1) < (unsigned)p.cv_W)) << t;
                cv_ok[i] = ok;
                cv_row[i] = A + (int64_t)m * p.lda;
            }
        }
    }
    auto issue = [&](int s) {                      // A stage s -> slot s % R : 2 load instructions per wave
        char* st = smem + (s % R) * X3R_OP;
        if (CV == 1) {
            x3r_issue_conv(cv_row, s < nk ? cv_ok : cv_none, cv_tap, cv_c0, p.cv_W, p.lda, st, wave, lane);
            cv_c0 += X3_BK;                        // channel block first, taps second, channel blocks last (gt_hip.h)
            if ((cv_c0 & (cv_cb - 1)) == 0) {
                cv_c0 -= cv_cb;
                if (++cv_tap == 9) { cv_tap = 0; cv_c0 += cv_cb; }
            }
        } else {
            x3r_issue<LA>(A, p.lda, m0, p.M, s * X3_BK, kend, st, wave, lane);
        }
    };
    // this wave's two 32-column B fragments of stage ks: plane pl, fragment j at bbase + pl * bplane + (j * KS + ks) KiB
    // (wave-uniform address in SGPRs + the lane's 16-byte slot).  The loads are inline asm on purpose: hipcc's own
    // scoreboard answers a register load inside this loop with s_waitcnt vmcnt(0) before the MFMAs, which also drains
    // the A stage requested a moment earlier (measured in the ISA); here the wait is the counted one below.
    const int wn_u = __builtin_amdgcn_readfirstlane(wn);
    const char* bbase = reinterpret_cast<const char*>(p.Bp) + (int64_t)((n0 + wn_u * 64) >> 5) * p.bp_KS * 1024;
    const int64_t bplane = (int64_t)p.bp_NT * p.bp_KS * 1024;
    const uint32_t voff = lane * 16;
    // Two register sets for B, one stage apart: B(kt + 1) is requested at the TOP of iteration kt, before the A stage of that
    // iteration, and is consumed one iteration later.  Vector-memory loads retire in order, so a wait for B also waits for
    // every A stage requested before it: with ONE set the loads of B(kt + 1) can only go out behind the MFMAs of B(kt), and
    // their latency (plus that of the A stage requested one iteration earlier) stands in front of every stage's MFMAs --
    // load, matrix and store time of a launch add up instead of overlapping (tools/ablate_x3.sh: 37 + 15 + 30 = 82 us).
    using frag_t = std::conditional_t<F16 != 0, f16x8, bf16x8>;    // the MFMA's own operand type: no conversion (= no copy of a
                                                                   // register the load has not filled yet) between load and use
    frag_t bn0[2][PLANES], bn1[2][PLANES];
    auto loadb = [&](int ks, frag_t (&bn)[2][PLANES]) {
#ifdef GT_ABL_X3_NOLOADB         // ablation: the B fragments are fetched for the first two stages only (timing; wrong results)
        if (ks > 1) return;
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
#ifdef GT_ABL_X3_NOSPLIT_B       // ablation: every stage reads the same (cache-resident) fragment
                const char* sp = bbase + pl * bplane + (int64_t)j * p.bp_KS * 1024;
#else
                const char* sp = bbase + pl * bplane + ((int64_t)j * p.bp_KS + ks) * 1024;
#endif
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bn[j][pl]) : "v"(voff), "s"(sp));
            }
    };
    // one 32-row tile of A at a time: its fragment is split and goes through its twelve MFMAs before the next one is read
    // (the three planes of ONE tile are live, not of both: the second B set has to fit under 168 registers)
    auto stage = [&](int kt, frag_t (&bn)[2][PLANES]) {
        const char* sa = smem + (kt % R) * X3R_OP;
        const int kbase = kt * X3_BK + 8 * lh;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float v[8];
            const int row = wrow + 32 * i + lr;
            x3r_frag<LA>(sa, row, lh, v);
            if (p.a_drop.thresh) x3_mask8<LA>(p.a_drop, akey, p.a_drop_ld, 0, m0 + row, kbase, v);
            if constexpr (F16) {
                // the row's amax of this stage (both k-halves); lower the row's exponent -- and rescale what the lane has
                // accumulated for it -- before anything could overflow
                // (a chain, so that it becomes four v_max3_f32 with |.| modifiers; the other k-half of the row sits in lane
                // l ^ 32: one v_permlane32_swap instead of a ds_bpermute round trip per tile and stage)
                float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));
                amax = fmaxf(fmaxf(amax, fabsf(v[3])), fabsf(v[4]));
                amax = fmaxf(fmaxf(amax, fabsf(v[5])), fabsf(v[6]));
                amax = xor32_max(fmaxf(amax, fabsf(v[7])));
                const int ex = (int)(__float_as_uint(amax) >> 23);
                const bool need = ex + ea[i] - 127 >= X3H_LIMIT;
                if (__any(need)) {                        // wave-uniform
                    const int enew = need ? X3H_TARGET + 127 - ex : ea[i];
                    const int d = enew - ea[i];           // <= 0
                    const float f = d < -126 ? 0.f : x3h_pow2(d);   // 2^-127 of a value is below fp32 resolution of the new ones
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] *= f;
                    ea[i] = enew;
                }
                const float sv = x3h_pow2(ea[i]) * (MI == 1 ? tsgn : ((GT_X3_ALT && (i & 1)) ? -1.f : 1.f));
                uint32_t q[4][2];
#pragma unroll
                for (int t = 0; t < 4; ++t) x3h_split_pair(v[2 * t], v[2 * t + 1], sv, q[t]);
                f16x8 am[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) am[pl] = __builtin_bit_cast(f16x8, u32x4{q[0][pl], q[1][pl], q[2][pl], q[3][pl]});
#pragma unroll
                for (int s = 1; s >= 0; --s)             // h1 g0 + h0 g1, then h0 g0
#pragma unroll
                    for (int pa = 0; pa < 2; ++pa) {
                        const int pb = s - pa;
                        if (pb < 0 || pb > 1) continue;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = mfma32h(bn[j][pb], am[pa], acc[i][j]);
                    }
                continue;
            }
            bf16x8 am[3];
            if (GT_X3_ALT && (MI == 1 || (i & 1))) {   // M-side: the sign alternates per 32-row tile (see below)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = MI == 1 ? v[q] * tsgn : -v[q];
            }
#ifdef GT_ABL_X3_NOSPLIT_A       // ablation builds (tools/ablate_x3.sh): timing only, results are wrong
            for (int pl = 0; pl < PLANES; ++pl) am[pl] = __builtin_bit_cast(bf16x8, u32x4{__float_as_uint(v[0]), __float_as_uint(v[2]), __float_as_uint(v[4]), __float_as_uint(v[6])});
#else
            x3r_split<3>(v, am);
#endif
            if constexpr (!F16) {
#pragma unroll
                for (int s = 2; s >= 0; --s) {             // plane pairs (pa, pb) with pa + pb = s, smallest terms first
#pragma unroll
                    for (int pa = 0; pa < 3; ++pa) {
                        const int pb = s - pa;
                        if (pb < 0 || pb >= 3) continue;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = mfma32(bn[j][pb], am[pa], acc[i][j]);
                    }
                }
            }
        }
    };
    // Request order of a wave:  B(0) A(0) .. A(R-2) | B(1) A(R-1) | B(2) A(R) | ...   (A = 2 loads, B = 6).
    // Top of iteration kt >= 1: A(kt) and B(kt) must have landed; the only request behind B(kt) is A(kt+R-2): vmcnt(2).
    // The barrier makes every wave's pieces of A(kt) visible and frees slot (kt-1) % R for the request of A(kt+R-1).
    // Tying the set to the statement keeps its MFMAs behind the wait.  Every iteration issues the same requests -- past
    // the end of K the A loader reads the zero line into a free slot and B re-reads its last stage -- so the counts hold
    // to the last stage and no load sits under a branch: a conditional asm load makes hipcc allocate fresh registers for
    // it and COPY them into the set at the join, before the data has arrived (seen in the ISA of a first version).
#define X3P_WAIT_AB_(N, bn)                                                                                            \
    do {                                                                                                               \
        if constexpr (PLANES == 3)                                                                                     \
            asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier"                                                        \
                         : "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[0][PLANES - 1]), "+v"(bn[1][0]), "+v"(bn[1][1]),    \
                           "+v"(bn[1][PLANES - 1])                                                                     \
                         :                                                                                             \
                         : "memory");                                                                                  \
        else                                                                                                           \
            asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier"                                                        \
                         : "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[1][0]), "+v"(bn[1][1])                              \
                         :                                                                                             \
                         : "memory");                                                                                  \
    } while (0)
#ifdef GT_X3P_PROF                                 // time spent in the waits of the K loop (wave 0)
    unsigned long long tw_sum = 0;
#define X3P_WAIT_AB(N, bn)                                                                                             \
    do {                                                                                                               \
        const unsigned long long w0_ = __builtin_amdgcn_s_memrealtime();                                              \
        X3P_WAIT_AB_(N, bn);                                                                                           \
        tw_sum += __builtin_amdgcn_s_memrealtime() - w0_;                                                              \
    } while (0)
#else
#define X3P_WAIT_AB(N, bn) X3P_WAIT_AB_(N, bn)
#endif
    const int klast = nk - 1;
    loadb(0, bn0);
#pragma unroll
    for (int s = 0; s < R - 1; ++s) issue(s);
    if (R == 3) X3P_WAIT_AB(2, bn0);               // iteration 0: behind A(0) are the R - 2 other stages of the prologue
    else if (R == 4) X3P_WAIT_AB(4, bn0);
    else if (R == 5) X3P_WAIT_AB(6, bn0);
    else X3P_WAIT_AB(8, bn0);
    X3P_STAMP(1);
    loadb(klast < 1 ? klast : 1, bn1);
    issue(R - 1);
    stage(0, bn0);
    int kt = 1;
    for (; kt + 1 < nk; kt += 2) {
        X3P_WAIT_AB(2, bn1);
        loadb(kt + 1, bn0);
        issue(kt + R - 1);
        stage(kt, bn1);
        X3P_WAIT_AB(2, bn0);
        loadb(kt + 2 < klast ? kt + 2 : klast, bn1);
        issue(kt + R);
        stage(kt + 1, bn0);
    }
    if (kt < nk) {                                 // odd stage out (nk even): nothing left to request
        X3P_WAIT_AB(2, bn1);
        stage(kt, bn1);
    }
    X3P_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero stages requested past the end of K: the ring becomes staging
#undef X3P_WAIT_AB
#undef X3P_WAIT_AB_
#ifdef GT_ABL_X3_NOSTORE
    if (acc[0][0][0] != 12345.678f) return;
#endif

    if constexpr (F16) {           // un-scale (row exponent of the lane, tile exponent of the packed columns) with the ALT sign
        const int* ebp = reinterpret_cast<const int*>(reinterpret_cast<const char*>(p.Bp) + 2 * bplane) + ((n0 + wn_u * 64) >> 5);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int et = -(ea[i] + ebp[j]), etc = et < -126 ? -126 : (et > 126 ? 126 : et);
                const float sg = x3h_pow2(etc) * (MI == 1 ? tsgn : ((GT_X3_ALT && (i & 1)) ? -1.f : 1.f));
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] *= (GT_X3_ALT && (e & 1)) ? -sg : sg;
                if (et != etc) {       // row amax x tile amax below ~2^-100: the rest of the power of two (no cliff)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = ldexpf(acc[i][j][e], et - etc);
                }
            }
    } else {
#if GT_X3_ALT
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float sg = MI == 1 ? tsgn : ((i & 1) ? -1.f : 1.f);
                    acc[i][j][e] *= (e & 1) ? -sg : sg;
                }
#endif
    }
    __syncthreads();                               // every wave is done with the ring: its first slots become staging
    if constexpr (HN > 0)
        x3_epilogue_hn<(HN > 0 ? HN : 32), 2>(p, acc, m0 + wm * 64 + lr, n0 + wn * 64 + 4 * lh, lane,
                                               reinterpret_cast<float*>(smem) + wave * X3_HN_STG);
    else
        x3_epilogue<MI>(p, acc, m0 + wrow, n0 + wn * 64, lane, reinterpret_cast<float*>(smem) + wave * X3_EP_STG, 0, 0, 0, 0,
                        bias4);
#ifdef GT_X3P_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the block's stores have left
    X3P_STAMP(3);
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long* o = x3p_prof + 8 * blockIdx.x;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID: cu / sh / se
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // XCC_ID
        o[6] = tile;
        o[7] = tw_sum;
    }
#endif
}

template <int LA, int HN, int CV, int BN = 128>        // the GT_PREC_F16X2 instances
__global__ __launch_bounds__(256, 3) void gemm_x3h_kernel(const GemmP p) {
    x3p_body<LA, HN, CV, BN, 1>(p);
}

// ---------------------------------------------------------------------------------------------------------------------
// Token-contracted weight gradients in GT_PREC_F16X2:  C[M][N] = sum_k A[k][M] B[k][N]  (both operands x-contiguous
// activations, K = tokens, M / N multiples of 128), split-K slabs like the ring kernel's.  gemm_x3r_kernel<1, 1> splits both
// operands again in every wave that multiplies them (each value is split twice per block, ~10 VALU instructions per MFMA:
// 121 / 181 us for 363 / 484 MB of operands); here a stage of 32 tokens is split ONCE, by the thread that fetched it, into
// two fp16 planes in LDS ([plane][k-group of 8 tokens][row]: a fragment is one aligned ds_read_b128, consecutive rows in
// consecutive 16-byte slots), and the four waves read their fragments from there: 24 MFMAs per wave and stage against ~180
// VALU instructions per thread.  The scale is one running exponent per operand and BLOCK: every stage the block takes the
// amax of the two tiles it is about to split (wave reduce + four floats through LDS, the barrier is there anyway), lowers the
// exponent -- rescaling its accumulators -- when the scaled amax would reach 2^15, and otherwise keeps it, so nothing can
// overflow and values are resolved to 2^-22 of the largest magnitude the block has seen (the weight gradient is a sum over
// all tokens: the tensor's scale is the relevant one).  GT_X3_ALT as everywhere: odd rows of both operands enter negated.
constexpr int X3W_KG = 4;                            // k-groups (8 tokens) per stage
constexpr int X3W_PLANE = X3W_KG * 128 * 16;         // bytes of one plane of one operand tile: 8 KB

// PF = stages of raw operand values a thread keeps in flight (registers).  With PF = 1 the next stage was requested after the
// current one had been split, i.e. its latency was covered by 24 MFMAs only (~0.3 us against >= 2 us under load): every
// stage paid most of a memory round trip, and the launch time did not move when the counters showed a quarter less traffic
// (profiles/r05_x3w_prefetch.json).  PF = 2: the request for stage s + 2 is issued when stage s has been split, a full
// stage of split + MFMA work earlier; 32 more registers (180: two blocks per CU instead of three, four stages per CU in
// flight instead of three).
#ifdef GT_X3W_PROF                                 // tools/x3w_prof.py: shader-clock cycles wave 0 of a block spends per phase
__device__ unsigned long long x3w_prof[8 * 4096];
extern "C" int gt_debug_x3w_prof(void* dst, long long bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(x3w_prof), (size_t)bytes);
}
#define X3W_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define X3W_T(i)
#endif

template <int PF>
__global__ __launch_bounds__(256, PF == 1 ? 3 : 2) void gemm_x3w_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * X3W_PLANE];      // A planes 0 / 1, B planes 0 / 1
    __shared__ float red[1][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    int tile, by;
    if (p.x3w_map) {
        // 1-D grid, 8 * ceil(n_split / 8) * tiles blocks.  Workgroups go to the eight XCDs round-robin: XCD x owns the K
        // chunks [x spx, (x + 1) spx), and the 2 - 3 output tiles of ONE chunk are consecutive workgroups of that XCD -- they
        // stream the same rows of the narrower operand at the same time, so its second (third) reader is served by the XCD's
        // L2.  With tile = blockIdx.x and chunk = blockIdx.y the tiles of a chunk sit on DIFFERENT XCDs and the counters show
        // the operand fetched once per tile (485 -> 364 MB at [128 x 256], 727 -> 498 MB at [384 x 128]).
        const int tiles = p.tiles_m * p.tiles_n, spx = (p.n_split + 7) >> 3;
        const int s = blockIdx.x >> 3;
        by = (blockIdx.x & 7) * spx + s / tiles;
        tile = s % tiles;
        if (by >= p.n_split) return;
    } else {
        const int tiles = gridDim.x, q = tiles >> 3, r = tiles & 7;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
        by = blockIdx.y;
    }
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int kbeg = by * p.k_chunk, kend = min(p.K, kbeg + p.k_chunk);
    const bool do_acs = p.acs != nullptr && tn == 0;

    // staging role: waves 0, 1 stage the A tile, waves 2, 3 the B tile; a thread owns rows 4 r4 .. 4 r4 + 3 of its tile and the
    // eight tokens of k-group kg: eight 16-byte loads (a wave instruction covers 512 contiguous bytes of two token rows), four
    // 8-token units to split and store
    const bool isB = wave >= 2;                        // wave-uniform
    const int st = tid & 127, r4 = st & 31, kg = st >> 5;
    const int wsw = (r4 >> 1) & 3;                     // plane-store swizzle of rows 4 r4 + c:  ((4 r4 + c) >> 3) & 3
    const int lrs = lr ^ ((lr >> 3) & 3);              // fragment-read swizzle of row .. + lr (the tile bases are multiples of 32)
    const float* Op = isB ? p.B + n0 + 4 * r4 : p.A + m0 + 4 * r4;
    const int64_t ldo = isB ? p.ldb : p.lda;
    // A whole stage (the usual case, block-uniform test): the address of a load is a wave-uniform row pointer (token k0 + e of
    // the operand: scalar registers, advanced by scalar adds) + a per-thread byte offset that never changes -- no vector
    // address arithmetic and no branch per load (the general form below cost ~10 VALU / SALU instructions per load, a
    // fifth of the split phase this kernel is bound by: tools/x3w_prof.py)
    const uint32_t voff = (uint32_t)((8 * kg * ldo + (isB ? n0 : m0) + 4 * r4) * (int64_t)sizeof(float));
    const char* rowbase = reinterpret_cast<const char*>(isB ? p.B : p.A);
    // round 5: partial edge tiles (M, N multiples of 32, e.g. ex3's 192 / 384 / 576): a thread whose four rows lie beyond the
    // operand's width stages zeros (its rows would be the NEXT token's values)
    const bool live = (isB ? n0 : m0) + 4 * r4 < (isB ? p.N : p.M);
    auto fetch = [&](f32x4 (&v)[8], int k0) __attribute__((always_inline)) {
        if (!live) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
        if (k0 + 32 <= kend) {
            const char* b = rowbase + (int64_t)k0 * ldo * (int64_t)sizeof(float);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = *reinterpret_cast<const f32x4*>(b + (int64_t)e * ldo * (int64_t)sizeof(float) + voff);
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + 8 * kg + e;
            v[e] = *reinterpret_cast<const f32x4*>(k < kend ? Op + (int64_t)k * ldo : x3_zero);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int ea = X3H_E0, eb = X3H_E0;                      // block-uniform running exponents of the two operands
    float asum[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef GT_X3W_PROF
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const unsigned long long tbeg = tlast;
#endif

    // one stage: the 32 tokens [k0, k0 + 32) whose values are in v; afterwards v holds the stage PF x 32 tokens further on
    auto stage = [&](f32x4 (&v)[8], int k0) __attribute__((always_inline)) {
        // amax of the stage (the values are in registers), per operand over its two waves
        float mx = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[e][0]), fabsf(v[e][1])), fmaxf(fabsf(v[e][2]), fabsf(v[e][3]))));
        // the exponents only move when some value would reach 2^LIMIT under the current one: a wave whose lanes are all
        // below that reports 0 ("in range") without the six cross-lane exchanges of a full reduction
        {
            const int xl = (int)(__float_as_uint(mx) >> 23);
            const bool over = xl + (isB ? eb : ea) - 127 >= X3H_LIMIT;
            if (__builtin_amdgcn_ballot_w64(over) != 0ull) {           // wave-uniform
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            } else {
                mx = 0.f;
            }
        }
        X3W_T(0);                                      // the stage's values have arrived (vmcnt) + amax
        if (lane == 0) red[0][wave] = mx;
        __syncthreads();                               // also: every wave is done reading the previous stage's planes
        X3W_T(1);
        const float ma = fmaxf(red[0][0], red[0][1]), mb = fmaxf(red[0][2], red[0][3]);
        const int xa = (int)(__float_as_uint(ma) >> 23), xb = (int)(__float_as_uint(mb) >> 23);
        int d = 0;
        if (xa + ea - 127 >= X3H_LIMIT) { d += X3H_TARGET + 127 - xa - ea; ea = X3H_TARGET + 127 - xa; }
        if (xb + eb - 127 >= X3H_LIMIT) { d += X3H_TARGET + 127 - xb - eb; eb = X3H_TARGET + 127 - xb; }
        if (d != 0) {                                  // block-uniform
            const float f = d < -126 ? 0.f : x3h_pow2(d);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] *= f;
        }
        const float sc = x3h_pow2(isB ? eb : ea);
        char* planes = smem + (isB ? 2 * X3W_PLANE : 0);
        if (do_acs && !isB) {                          // column sums of A (the bias gradient): the four rows at once, as
            const f32x4 s4 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));   // packed adds
#pragma unroll
            for (int c = 0; c < 4; ++c) asum[c] += s4[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {                  // row 4 r4 + c: its eight tokens -> one unit per plane
            const float sv = (GT_X3_ALT && (c & 1)) ? -sc : sc;          // odd rows enter negated
            uint32_t q[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) x3h_split_pair(v[2 * t][c], v[2 * t + 1][c], sv, q[t]);
            // slot swizzle: row R sits in slot R ^ ((R >> 3) & 3).  A thread owns the four rows 4 r4 + c (its global loads are
            // float4 over rows), so without it the eight lanes of a ds_write_b128 pass hit 16-byte slots 64 bytes apart -- two
            // bank groups, a four-way conflict on every plane store (SQ_LDS_BANK_CONFLICT: 0.59 of the kernel's LDS cycles);
            // with it those eight slots are distinct modulo 8, and so are the eight consecutive rows of a fragment read
            const int off = (kg * 128 + 4 * r4 + (c ^ wsw)) << 4;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                *reinterpret_cast<u32x4*>(planes + pl * X3W_PLANE + off) = u32x4{q[0][pl], q[1][pl], q[2][pl], q[3][pl]};
        }
        X3W_T(2);                                      // exponents, split, plane stores
        if (k0 + 32 * PF < kend) fetch(v, k0 + 32 * PF);   // the values of stage s + PF travel under PF stages of work
        __syncthreads();
        X3W_T(3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {               // two MFMA k-steps of 16 tokens
            f16x8 am[2][2], bn[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int kq = 2 * ks + lh;
                    am[i][pl] = *reinterpret_cast<const f16x8*>(smem + pl * X3W_PLANE + ((kq * 128 + wm * 64 + 32 * i + lrs) << 4));
                    bn[i][pl] = *reinterpret_cast<const f16x8*>(smem + (2 + pl) * X3W_PLANE + ((kq * 128 + wn * 64 + 32 * i + lrs) << 4));
                }
#pragma unroll
            for (int s = 1; s >= 0; --s)
#pragma unroll
                for (int pa = 0; pa < 2; ++pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb > 1) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32h(bn[j][pb], am[i][pa], acc[i][j]);
                }
        }
        X3W_T(4);                                      // fragment reads + MFMA issue
    };

    if (PF == 1) {
        f32x4 v[8];
        if (kbeg < kend) fetch(v, kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += 32) stage(v, k0);
    } else {
        f32x4 v0[8], v1[8];
        if (kbeg < kend) fetch(v0, kbeg);
        if (kbeg + 32 < kend) fetch(v1, kbeg + 32);
        for (int k0 = kbeg; k0 < kend; k0 += 64) {
            stage(v0, k0);
            if (k0 + 32 < kend) stage(v1, k0 + 32);    // block-uniform
        }
    }

    if (do_acs) {                                      // row sums of A: the four k-group threads of a row
        __syncthreads();
        float* part = reinterpret_cast<float*>(smem);
        if (!isB) {
#pragma unroll
            for (int c = 0; c < 4; ++c) part[kg * 128 + 4 * r4 + c] = asum[c];
        }
        __syncthreads();
        if (tid < 128 && m0 + tid < p.M)
            p.acs[(int64_t)by * p.M + m0 + tid] = (part[tid] + part[128 + tid]) + (part[256 + tid] + part[384 + tid]);
    }
#ifdef GT_X3W_PROF
    if (tid == 0 && blockIdx.x < 4096) {
        unsigned long long* o = x3w_prof + 8 * blockIdx.x;
        for (int i = 0; i < 5; ++i) o[i] = tacc[i];
        o[5] = __builtin_readcyclecounter() - tbeg;
        o[6] = (unsigned long long)((kend - kbeg + 31) / 32);
        o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // un-scale, undo the sign, store the slab tile: lane (lr, lh) holds row m = .. + 32 i + lr and columns .. + 32 j + 8 g + 4 lh + t
    const int et = -(ea + eb), etc = et < -126 ? -126 : (et > 126 ? 126 : et);
    const float us = x3h_pow2(etc) * x3_alt_sign(lr);
    if (et != etc) {                   // operands below ~2^-100 of unit scale: apply the rest of the power of two first
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = ldexpf(acc[i][j][e], et - etc);
    }
    float* C = p.C + (int64_t)by * p.c_split;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + 32 * i + lr;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + 32 * j + 8 * g + 4 * lh;
                if (m < p.M && n < p.N)
                    *reinterpret_cast<f32x4*>(C + (int64_t)m * p.ldc + n) =
                        f32x4{acc[i][j][4 * g] * us, -acc[i][j][4 * g + 1] * us, acc[i][j][4 * g + 2] * us, -acc[i][j][4 * g + 3] * us};
            }
    }
}

static int x3w_prefetch() {                          // GT_X3W_PF = 1: the single-stage prefetch, for A/B runs
    static const int pf = [] { const char* e = getenv("GT_X3W_PF"); return e && atoi(e) == 1 ? 1 : 2; }();
    return pf;
}

// the launches gemm_x3w_kernel takes: GT_PREC_F16X2, both operands x-contiguous and 16-byte aligned, M / N multiples of 32 (partial
// edge tiles stage zeros: round 5, ex3's 192 / 384 / 576-wide weights),
// a long token contraction cut into split-K slabs (raw epilogue), no batching / dropout / second product
bool x3w_ok(const gt_gemm_desc* d, int split) {
    static const int on = [] { const char* e = getenv("GT_X3W"); return e ? atoi(e) : 1; }();
    return on && d->precision == GT_PREC_F16X2 && d->layout_a == 1 && d->layout_b == 1 && split > 1 && d->K >= 16384 &&
           (d->M & 31) == 0 && (d->N & 31) == 0 && d->M >= 32 && d->N >= 32 && d->batch0 * d->batch1 == 1 && d->K2 == 0 && d->cv_c == 0 &&
           !(d->a_drop.p > 0.f) && ((reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->B)) & 15) == 0 &&
           (d->lda & 3) == 0 && (d->ldb & 3) == 0;
}

// operands the ring kernel's direct loads can take (see its header comment)
static bool x3r_ok(const GemmP& p, int layout_a, int layout_b) {
    if (p.K2 > 0 || !p.a_vec || !p.b_vec) return false;
    if ((layout_a == 0 || layout_b == 0) && (p.K & 3)) return false;
    if (layout_a == 1 && (p.M & 3)) return false;
    if (layout_b == 1 && (p.N & 3)) return false;
    return true;
}

bool x3_shape_ok(const gt_gemm_desc* d) {
    // whole 128 x 128 tiles dominate (the padding of a partial edge tile is bounded by the sizes below)
    if (d->ep_mode != GT_EP_NORMAL && d->ep_mode != GT_EP_HEADNORM) return false;
    // narrow implicit convolutions (N >= 32) ride on the 64-wide tile of the packed-B kernel
    if (d->cv_c > 0 && !d->cv_wgrad && d->N >= 32 && d->N < 96 && d->M >= 1024 && d->K >= 16) return true;
    // round 6: narrow token products too (N < 96: the 64-column remainder of a width-split 192-wide product -- ex3's d_model --
    // and the d_model = 48 / 64 products of ex4 / ex1), under exactly the conditions that put them on the packed-B kernels
    // (x3_packed_ok), whose 128 x 64 tile instance takes N <= 64; they ran on the fp32 MFMA engine until now
    static const int narrow = [] { const char* e = getenv("GT_X3_NARROW"); return e ? atoi(e) : 1; }();
    if (narrow && d->cv_c == 0 && d->N >= 16 && d->N < 96 && d->M >= 16384 && d->M >= 8 * (int64_t)d->N && d->K >= 16 &&
        (d->precision == GT_PREC_BF16X3 || d->precision == GT_PREC_F16X2) &&
        (d->K & 3) == 0 && d->layout_a == 0 && d->batch0 * d->batch1 == 1 && d->K2 == 0 && !d->a_colsum && d->split_k == 1 &&
        d->ep_mode == GT_EP_NORMAL && (d->lda & 3) == 0 && (reinterpret_cast<uintptr_t>(d->A) & 15) == 0)
        return true;
    // ... and the token-contracted weight gradients of the narrow models (ex1: d_model 64, ffn 128): gemm_x3w_kernel takes
    // partial 128-blocks since round 5, a lone 32 / 64-wide block is the same code path (they ran on the fp32 engine)
    if (narrow && d->precision == GT_PREC_F16X2 && d->layout_a == 1 && d->layout_b == 1 && d->split_k == 0 && d->K >= 16384 &&
        d->M >= 32 && d->N >= 32 && (d->M & 31) == 0 && (d->N & 31) == 0 && x3w_ok(d, 2))
        return true;
    return d->M >= 96 && d->N >= 96 && d->K >= 16;
}

// the fused head-norm epilogue exists on the ring kernel, three planes, 16-byte aligned everything
bool x3_headnorm_ok(const GemmP& p, int layout_a, int layout_b, int planes) {
    if (layout_a || layout_b || planes != 3 || !x3r_ok(p, 0, 0)) return false;
    if (p.hn_dk != 16 && p.hn_dk != 32 && p.hn_dk != 64) return false;
    if (p.hn_dkr != p.hn_dk && !(p.hn_dk == 64 && p.hn_dkr == 48 && p.Bp)) return false;   // head slots: packed-B kernels only
    if (p.hn_p > 4) return false;                  // the epilogue keeps a row's coordinates in four registers
    if (!p.c_vec || (p.N & 63) || (64 / p.hn_dk) * p.hn_DP + 4 > 88) return false;      // staging row fits X3_HN_STG
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return al(p.bias) && al(p.hn_gamma) && al(p.hn_beta) && al(p.hn_out) && al(p.hn_stats);
}

// ring depth: 3 stages x 16 KB = 48 KB, three blocks per CU (default; measured 38.0 vs 38.35 ms/step at B128) or
// GT_X3_RING_DEPTH=4: 4 x 16 KB = 64 KB, two blocks per CU
static int x3_ring_depth() {
    static const int d = [] { const char* e = getenv("GT_X3_RING_DEPTH"); return (e && atoi(e) == 4) ? 4 : 3; }();
    return d;
}

template <int LA, int LB, int R>
static void x3_launch_ring(const GemmP& p, int planes, dim3 grid, hipStream_t st) {
    if (planes == 1) hipLaunchKernelGGL((gemm_x3r_kernel<LA, LB, 1, R>), grid, dim3(256), 0, st, p);
    else if (planes == 2) hipLaunchKernelGGL((gemm_x3r_kernel<LA, LB, 2, R>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_x3r_kernel<LA, LB, 3, R>), grid, dim3(256), 0, st, p);
}

template <int LA, int LB>
static void x3_launch_planes(const GemmP& p, int planes, bool ring, dim3 grid, hipStream_t st) {
    if (ring) {
        if (x3_ring_depth() == 3) x3_launch_ring<LA, LB, 3>(p, planes, grid, st);
        else x3_launch_ring<LA, LB, 4>(p, planes, grid, st);
        return;
    }
    if (planes == 1) hipLaunchKernelGGL((gemm_x3_kernel<LA, LB, 1>), grid, dim3(256), 0, st, p);
    else if (planes == 2) hipLaunchKernelGGL((gemm_x3_kernel<LA, LB, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_x3_kernel<LA, LB, 3>), grid, dim3(256), 0, st, p);
}

static bool x3_use_ring(const GemmP& p, int layout_a, int layout_b) {
    static const int force = [] { const char* e = getenv("GT_X3_RING"); return e ? atoi(e) : -1; }();   // tuning knob
    return force == 0 ? false : x3r_ok(p, layout_a, layout_b);
}

int x3_launch(const GemmP& p, int layout_a, int layout_b, int planes, unsigned tiles, unsigned split, unsigned batch,
              hipStream_t st) {
    if (planes < 1 || planes > 3) return GT_EINVAL;
    const dim3 grid(tiles, split, batch);
    const int lay = layout_a * 2 + layout_b;
    const bool ring = x3_use_ring(p, layout_a, layout_b);
    if (p.Bp) {                                    // packed-B kernel (x3_packed_ok said yes)
        const int hn = p.ep_mode == GT_EP_HEADNORM ? p.hn_dk : 0;
        if (hn && !x3_headnorm_ok(p, layout_a, 0, planes)) return GT_ENOTSUP;
        if (p.N <= 64 && !hn && layout_a == 0) {   // narrow output: 128 x 64 tiles (p.tiles_n counts 128-wide tiles: one)
            GemmP q = p;
            q.tiles_n = (p.N + 63) / 64;
            const dim3 gn((unsigned)(p.tiles_m * q.tiles_n));
            if (p.bp_f16) {
                if (p.cv_C > 0) hipLaunchKernelGGL((gemm_x3h_kernel<0, 0, 1, 64>), gn, dim3(256), 0, st, q);
                else hipLaunchKernelGGL((gemm_x3h_kernel<0, 0, 0, 64>), gn, dim3(256), 0, st, q);
            } else if (p.cv_C > 0) hipLaunchKernelGGL((gemm_x3p_kernel<0, 0, 1, 64>), gn, dim3(256), 0, st, q);
            else hipLaunchKernelGGL((gemm_x3p_kernel<0, 0, 0, 64>), gn, dim3(256), 0, st, q);
            GT_LAUNCH_CHECK();
            return 0;
        }
        if (p.bp_f16) {
            if (p.cv_C > 0) hipLaunchKernelGGL((gemm_x3h_kernel<0, 0, 1>), grid, dim3(256), 0, st, p);
            else if (hn == 16) hipLaunchKernelGGL((gemm_x3h_kernel<0, 16, 0>), grid, dim3(256), 0, st, p);
            else if (hn == 32) hipLaunchKernelGGL((gemm_x3h_kernel<0, 32, 0>), grid, dim3(256), 0, st, p);
            else if (hn == 64) hipLaunchKernelGGL((gemm_x3h_kernel<0, 64, 0>), grid, dim3(256), 0, st, p);
            else if (layout_a == 0) hipLaunchKernelGGL((gemm_x3h_kernel<0, 0, 0>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_x3h_kernel<1, 0, 0>), grid, dim3(256), 0, st, p);
            GT_LAUNCH_CHECK();
            return 0;
        }
        if (p.cv_C > 0) hipLaunchKernelGGL((gemm_x3p_kernel<0, 0, 1>), grid, dim3(256), 0, st, p);
        else if (hn == 16) hipLaunchKernelGGL((gemm_x3p_kernel<0, 16, 0>), grid, dim3(256), 0, st, p);
        else if (hn == 32) hipLaunchKernelGGL((gemm_x3p_kernel<0, 32, 0>), grid, dim3(256), 0, st, p);
        else if (hn == 64) hipLaunchKernelGGL((gemm_x3p_kernel<0, 64, 0>), grid, dim3(256), 0, st, p);
        else if (layout_a == 0) hipLaunchKernelGGL((gemm_x3p_kernel<0, 0, 0>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_x3p_kernel<1, 0, 0>), grid, dim3(256), 0, st, p);
        GT_LAUNCH_CHECK();
        return 0;
    }
    if (p.ep_mode == GT_EP_HEADNORM) {
        if (!x3_headnorm_ok(p, layout_a, layout_b, planes)) return GT_ENOTSUP;
        if (x3_ring_depth() == 3) {
            if (p.hn_dk == 16) hipLaunchKernelGGL((gemm_x3r_kernel<0, 0, 3, 3, 16>), grid, dim3(256), 0, st, p);
            else if (p.hn_dk == 32) hipLaunchKernelGGL((gemm_x3r_kernel<0, 0, 3, 3, 32>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_x3r_kernel<0, 0, 3, 3, 64>), grid, dim3(256), 0, st, p);
        } else {
            if (p.hn_dk == 16) hipLaunchKernelGGL((gemm_x3r_kernel<0, 0, 3, 4, 16>), grid, dim3(256), 0, st, p);
            else if (p.hn_dk == 32) hipLaunchKernelGGL((gemm_x3r_kernel<0, 0, 3, 4, 32>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_x3r_kernel<0, 0, 3, 4, 64>), grid, dim3(256), 0, st, p);
        }
        GT_LAUNCH_CHECK();
        return 0;
    }
    if (p.cv_C > 0 && p.cv_wgrad) {                // convolution weight gradient: nine taps x K chunks, 1-D grid
        if (lay != 3 || !x3r_ok(p, 1, 1) || batch != 9 || p.cv_W < X3_BK) return GT_ENOTSUP;
        const dim3 g1((unsigned)(72 * ((tiles * split + 7) / 8)));
        if (planes == 1) hipLaunchKernelGGL((gemm_x3r_kernel<1, 1, 1, 3, 0, 2>), g1, dim3(256), 0, st, p);
        else if (planes == 2) hipLaunchKernelGGL((gemm_x3r_kernel<1, 1, 2, 3, 0, 2>), g1, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_x3r_kernel<1, 1, 3, 3, 0, 2>), g1, dim3(256), 0, st, p);
        GT_LAUNCH_CHECK();
        return 0;
    }
    if (p.cv_C > 0) {                              // implicit convolution: ring kernel, depth 3
        if (lay != 0 || !x3r_ok(p, 0, 0) || (p.cv_C & 15) || split != 1 || batch != 1) return GT_ENOTSUP;
        if (planes == 1) hipLaunchKernelGGL((gemm_x3r_kernel<0, 0, 1, 3, 0, 1>), grid, dim3(256), 0, st, p);
        else if (planes == 2) hipLaunchKernelGGL((gemm_x3r_kernel<0, 0, 2, 3, 0, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_x3r_kernel<0, 0, 3, 3, 0, 1>), grid, dim3(256), 0, st, p);
        GT_LAUNCH_CHECK();
        return 0;
    }
    if (p.wg_f16) {                                // GT_PREC_F16X2 weight gradient (x3w_ok said yes)
        const int pf = x3w_prefetch();
        // map 0: tile-major grid; 1: chunk-major within an XCD (default since round 6: the same time as 2 at three tiles, and
        // the narrower operand is fetched once instead of once per tile -- 727 -> ~500 MB at [384 x 128]); 2: that for two tiles only
        static const int map = [] { const char* e = getenv("GT_X3W_MAP"); return e ? atoi(e) : 1; }();
        GemmP q = p;
        q.x3w_map = map == 1 || (map == 2 && tiles == 2);
        const dim3 g1 = q.x3w_map ? dim3(8u * ((split + 7) / 8) * tiles) : grid;
        if (pf == 1) hipLaunchKernelGGL(gemm_x3w_kernel<1>, g1, dim3(256), 0, st, q);
        else hipLaunchKernelGGL(gemm_x3w_kernel<2>, g1, dim3(256), 0, st, q);
        GT_LAUNCH_CHECK();
        return 0;
    }
    if (lay == 0) x3_launch_planes<0, 0>(p, planes, ring, grid, st);
    else if (lay == 1) x3_launch_planes<0, 1>(p, planes, ring, grid, st);
    else if (lay == 2) x3_launch_planes<1, 0>(p, planes, ring, grid, st);
    else x3_launch_planes<1, 1>(p, planes, ring, grid, st);
    GT_LAUNCH_CHECK();
    return 0;
}

// ---- packed-B path: host side ---------------------------------------------------------------------------------------
// B small next to A (a weight against >= 16384 token rows), three planes, one launch (no batching, split-K, second
// product or row-sum by-product), operands the direct loads can take.  GT_X3_PACKED=0 switches it off (A/B runs).
bool x3_packed_ok(const gt_gemm_desc* d, int planes, int split) {
    static const int on = [] { const char* e = getenv("GT_X3_PACKED"); return e ? atoi(e) : 1; }();
    if (!on || planes != 3 || split != 1 || d->batch0 * d->batch1 != 1 || d->K2 > 0 || d->a_colsum) return false;
    if (d->cv_c > 0 && d->cv_wgrad) return false;
    // below 16384 rows the extra (pack) launch is not paid back -- except for the implicit convolutions, whose K = 9 C makes
    // the product long enough (round 5: the down-scaler chain runs at every batch, no library convolution below B = 3)
    if (d->M < (d->cv_c > 0 ? 1024 : 16384) || d->M < 8 * (int64_t)d->N) return false;
    const bool a16 = (reinterpret_cast<uintptr_t>(d->A) & 15) == 0;
    if (d->cv_c > 0) return a16 && d->layout_a == 0 && (d->cv_c & 15) == 0 && (d->lda <= d->cv_c || (d->lda & 3) == 0);
    if (!a16 || (d->lda & 3)) return false;
    return d->layout_a == 0 ? (d->K & 3) == 0 : (d->M & 3) == 0;
}

static inline int x3p_nt(int N) { return ((N + X3_BN - 1) / X3_BN) * (X3_BN / 32); }
static inline int x3p_ks(int K) { return (K + X3_BK - 1) / X3_BK; }

int64_t x3_packed_bytes(const gt_gemm_desc* d) { return (int64_t)3 * x3p_nt(d->N) * x3p_ks(d->K) * 1024; }

// the descriptor (already in head slots, gt_gemm.hip: hn_slots) describes 48-wide heads in 64-column slots
static inline int x3_pad48(const gt_gemm_desc* d) {
    return d->ep_mode == GT_EP_HEADNORM && d->hn_dk == 48 && d->N == 3 * d->hn_h * 64;
}

int x3_pack_b(const gt_gemm_desc* d, GemmP& p, void* ws, int64_t ws_bytes, hipStream_t st) {
    if (d->b_packed) {                             // the caller packed this weight already (gt_gemm_pack_b_many)
        if (reinterpret_cast<uintptr_t>(d->b_packed) & 15) return GT_EALIGN;
        p.bp_f16 = d->precision == GT_PREC_F16X2;
        p.Bp = d->b_packed; p.bp_NT = x3p_nt(d->N); p.bp_KS = x3p_ks(d->K);
        return 0;
    }
    if (!ws || ws_bytes < x3_packed_bytes(d) || (reinterpret_cast<uintptr_t>(ws) & 15)) return GT_EWS;
    const int NT = x3p_nt(d->N), KS = x3p_ks(d->K);
    const int threads = NT * KS * 64;
    const int pad48 = x3_pad48(d);
    p.bp_f16 = d->precision == GT_PREC_F16X2;
    if (p.bp_f16)              // two planes + NT tile exponents: fits the three-plane buffer
        hipLaunchKernelGGL(x3_pack_b16_kernel, dim3(NT), dim3(1024), 0, st, d->B, d->layout_b, d->ldb, d->N, d->K, NT, KS,
                           reinterpret_cast<u32x4*>(ws), pad48);
    else
        hipLaunchKernelGGL(x3_pack_b_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, d->B, d->layout_b, d->ldb, d->N,
                           d->K, NT, KS, reinterpret_cast<u32x4*>(ws), pad48);
    GT_LAUNCH_CHECK();
    p.Bp = ws; p.bp_NT = NT; p.bp_KS = KS;
    return 0;
}

int x3_pack_b_many(const gt_gemm_desc* descs, void* const* outs, int n, hipStream_t st) {
    if (n <= 0 || n > X3_PACK_MANY) return GT_EINVAL;
    PackManyP q{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const gt_gemm_desc* d = &descs[i];
        if (d->precision != GT_PREC_F16X2 || !d->B || !outs[i] || (reinterpret_cast<uintptr_t>(outs[i]) & 15)) return GT_ENOTSUP;
        q.B[i] = d->B; q.out[i] = reinterpret_cast<u32x4*>(outs[i]); q.ldb[i] = d->ldb; q.layout_b[i] = d->layout_b;
        q.N[i] = d->N; q.K[i] = d->K; q.NT[i] = x3p_nt(d->N); q.KS[i] = x3p_ks(d->K); q.pad48[i] = (unsigned char)x3_pad48(d);
        q.start[i] = blocks;
        blocks += q.NT[i];
    }
    q.start[n] = blocks;
    q.n = n;
    hipLaunchKernelGGL(x3_pack_b16_many_kernel, dim3(blocks), dim3(1024), 0, st, q);
    GT_LAUNCH_CHECK();
    return 0;
}

const char* x3_kernel_name(const GemmP& p, int layout_a, int layout_b, int planes, int hn_dk) {
    static thread_local char buf[112];
    if (p.Bp) {
        snprintf(buf, sizeof(buf), "void gt::gemm_x3%c_kernel<%d, %d, %d, %d>(gt::GemmP)", p.bp_f16 ? 'h' : 'p', p.cv_C > 0 ? 0 : layout_a, hn_dk,
                 p.cv_C > 0 ? 1 : 0, (p.N <= 64 && !hn_dk && layout_a == 0) ? 64 : 128);
        return buf;
    }
    if (hn_dk > 0) {
        snprintf(buf, sizeof(buf), "void gt::gemm_x3r_kernel<0, 0, 3, %d, %d, 0>(gt::GemmP)", x3_ring_depth(), hn_dk);
        return buf;
    }
    if (p.wg_f16) {
        snprintf(buf, sizeof(buf), "void gt::gemm_x3w_kernel<%d>(gt::GemmP)", x3w_prefetch());
        return buf;
    }
    if (p.cv_C > 0 && p.cv_wgrad)
        snprintf(buf, sizeof(buf), "void gt::gemm_x3r_kernel<1, 1, %d, 3, 0, 2>(gt::GemmP)", planes);
    else if (p.cv_C > 0)
        snprintf(buf, sizeof(buf), "void gt::gemm_x3r_kernel<0, 0, %d, 3, 0, 1>(gt::GemmP)", planes);
    else if (x3_use_ring(p, layout_a, layout_b))
        snprintf(buf, sizeof(buf), "void gt::gemm_x3r_kernel<%d, %d, %d, %d, 0, 0>(gt::GemmP)", layout_a, layout_b, planes,
                 x3_ring_depth());
    else
        snprintf(buf, sizeof(buf), "void gt::gemm_x3_kernel<%d, %d, %d>(gt::GemmP)", layout_a, layout_b, planes);
    return buf;
}

}  // namespace gt
