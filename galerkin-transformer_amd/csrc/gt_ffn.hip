// Fused FeedForward forward (reference libs/layers.py:979-987 with the residual of libs/model.py:131-132) in the two-term
// fp16 arithmetic (GT_PREC_F16X2, gt_gemm_x3.hip):
//
//     hid[t][:] = dropout_h( relu( x[t] W1^T + b1 ) )            -> HBM (the backward reads it: ReLU decisions, dW2)
//     out[t][:] = res[t] + dropout_o( hid[t] W2^T + b2 )
//
// ONE launch instead of two packed-B products: a block owns 64 token rows, the [64 x f] hidden tile stays in LDS between
// the two contractions (as fp32, in the stage-image layout the split-operand kernels read their A operand from), so the
// second product has no global A loads at all -- its per-block chain is MFMA + epilogue -- and the hidden activation is
// written to HBM once and never read back in the forward (layer forward traffic 2.30 -> 1.94 GB at B = 128).
// The arithmetic is that of the two separate gemm_x3h launches, value for value: the same packed weight planes
// (gt_gemm_pack_b_many), the same per-row running exponent, the same products in the same order -- the fused path returns
// the bits of the unfused one (tests/test_kernels_gpu.py::test_ffn_fused_forward_equals_two_launches).
//
// Geometry (d = 128, f = 256): 256 threads = 4 waves.
//   phase 1  H[64 x 256] = x[64 x 128] W1^T : waves side by side over the hidden columns (wave w: columns 64 w ..), each
//            wave two 32-row tiles x two 32-column fragments = f32x16[2][2]; the whole x tile (8 stages of 64 rows x 16 k)
//            requested at once by direct global->LDS loads into the region of the later H image; B fragments of the packed
//            W1 three stages ahead in three register sets; counted vmcnt waits as in gemm_x3p_kernel.
//   epilogue 1  un-scale, + b1, ReLU, dropout -> the H image in LDS (16 stages x 4 KB + 16 B skew).
//   phase 2  out[64 x 128] = H W2^T : waves side by side over the output columns (64 x 32 each), sixteen stages straight out
//            of LDS; every B fragment of W2 fetched by one wave only.
//   epilogue 2  residual rows requested BEFORE the hidden tile's stores (vector memory retires in order: a wait for them
//            behind sixteen stores would wait for the stores), H -> HBM as whole 1-KB rows, then the out tile through a
//            wave-private transposing tile (aliasing the dead H image): + b2, dropout, + residual, 256-byte row segments.
// LDS: 65 792 B (x tile / H image / out staging, one after the other) -> two blocks per CU.
#include <cstdint>
#include <cstdlib>
#include <algorithm>

#include "gt_common.h"

namespace gt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* ffn_lds_ptr;
typedef const __attribute__((address_space(1))) void* ffn_glb_ptr;

#ifndef GT_X3_ALT
#define GT_X3_ALT 1
#endif
constexpr int FFN_BM = 64, FFN_BK = 16;
constexpr int FFN_STAGE = FFN_BM * FFN_BK * 4;             // 4096 B: one A stage (64 rows x 16 k, fp32)
constexpr int FFN_HSTRIDE = FFN_STAGE + 16;                // stage pitch of the H image (skewed: row reads across stages spread over the banks)
constexpr int FFN_E0 = 120, FFN_TARGET = 13, FFN_LIMIT = 15;   // = X3H_* of gt_gemm_x3.hip
constexpr int FFN_EP_SW = 36;                              // staging row pitch (floats) of a wave's 64 x 32 out tile: 32 + 4

__device__ __attribute__((aligned(16))) float ffn_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct FfnP {
    const float* x; const float* res; float* hid; float* out;
    const float* b1; const float* b2;
    const void* Bp1; const void* Bp2;                      // packed planes of W1 [f, d] and W2 [d, f] (x3_pack_b16 layout)
    int T, d, f;
    int KS1, NT1, KS2, NT2;
    DropDev dh, dout;
    int act;                                               // GT_ACT_RELU | GT_ACT_NONE
    // ReLU / dropout decisions of the hidden tile, one bit per value in the accumulator layout of phase 1: word
    // [(tile * 4 + wave) * 64 + lane][i], bit 16 j + 4 g + t  <->  hidden value (row 32 i + lr, column 64 wave + 32 j + 8 g + 4 lh + t).
    // The forward writes them (bits_out), the backward (bwd != 0) takes its mask from them (bits_in): both run this kernel's
    // geometry, so the word is lane-local on both sides.
    uint32_t* bits_out; const uint32_t* bits_in;
    int bwd;                                               // 1: gh = (gm W2) .* bits * e1_scale ;  dx = res + gh W1 ;  out2 = dx .* mask2
    float e1_scale;
    float* out2; DropDev d2;
};

__device__ __forceinline__ float ffn_pow2(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }
__device__ __forceinline__ float ffn_alt(int parity) { return (GT_X3_ALT && (parity & 1)) ? -1.f : 1.f; }

// this lane's 8 consecutive k (k-half lh) of tile row `row` from a stage image (gt_gemm_x3.hip: x3r_frag<0>)
__device__ __forceinline__ void ffn_frag(const char* __restrict__ img, int row, int lh, float (&v)[8]) {
    const int s = (row >> 2) & 3;
    const f32x4 a = *reinterpret_cast<const f32x4*>(img + row * 64 + (((2 * lh) ^ s) << 4));
    const f32x4 b = *reinterpret_cast<const f32x4*>(img + row * 64 + (((2 * lh + 1) ^ s) << 4));
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// One stage of one 32-row tile: the row's running exponent (lowered, with the accumulators, before anything could overflow),
// the two-term fp16 split of the lane's eight values, the three products against the stage's B fragments.  NJ column fragments.
template <int NJ>
__device__ __forceinline__ void ffn_tile_stage(const float (&v)[8], int& ea, float sgn, const f16x8 (&bn)[NJ][2],
                                               f32x16 (&acc)[NJ]) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));
    amax = fmaxf(fmaxf(amax, fabsf(v[3])), fabsf(v[4]));
    amax = fmaxf(fmaxf(amax, fabsf(v[5])), fabsf(v[6]));
    amax = xor32_max(fmaxf(amax, fabsf(v[7])));
    const int ex = (int)(__float_as_uint(amax) >> 23);
    const bool need = ex + ea - 127 >= FFN_LIMIT;
    if (__any(need)) {                                     // wave-uniform
        const int enew = need ? FFN_TARGET + 127 - ex : ea;
        const int dlt = enew - ea;                         // <= 0
        const float f = dlt < -126 ? 0.f : ffn_pow2(dlt);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] *= f;
        ea = enew;
    }
    const float sv = ffn_pow2(ea) * sgn;
    uint32_t q[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) f16_mulsplit_pair(v[2 * t], sv, v[2 * t + 1], sv, q[t][0], q[t][1]);
    f16x8 am[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) am[pl] = __builtin_bit_cast(f16x8, u32x4{q[0][pl], q[1][pl], q[2][pl], q[3][pl]});
#pragma unroll
    for (int s = 1; s >= 0; --s)                           // h1 g0 + h0 g1, then h0 g0
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const int pb = s - pa;
            if (pb < 0 || pb > 1) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bn[j][pb], am[pa], acc[j], 0, 0, 0);
        }
}

// un-scale one accumulator (row exponent of the lane, tile exponent of the packed columns) with the ALT sign
__device__ __forceinline__ void ffn_unscale(f32x16& a, int ea, int eb, float sgn) {
    const int et = -(ea + eb), etc = et < -126 ? -126 : (et > 126 ? 126 : et);
    const float sg = ffn_pow2(etc) * sgn;
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] *= (GT_X3_ALT && (e & 1)) ? -sg : sg;
    if (et != etc) {
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = ldexpf(a[e], et - etc);
    }
}

#define FFN_WAIT_B_(N, bn)                                                                                             \
    asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[1][0]), "+v"(bn[1][1])::"memory")
#define FFN_WAIT_AB_(N, bn)                                                                                            \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier"                                                                \
                 : "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[1][0]), "+v"(bn[1][1])::"memory")

// PS (pre-split, the default): epilogue 1 splits the hidden tile ONCE into two fp16 planes in LDS under one power-of-two
// exponent per token row (taken over all f columns: the four waves' column-chunk maxima meet in a 1-KB table), phase 2 reads
// finished MFMA fragments and has no operand split left -- in the !PS form every wave re-splits the whole tile for its 32
// output columns (1 280 of the ~4 100 VALU instructions of a lane; the launch is issue-bound).  The fp32 values stay in the
// phase-1 accumulator registers and go to HBM after phase 2 through a wave-private transposing tile.  !PS returns the bits
// of the two gt_gemm launches (running exponent per row and stage, as gemm_x3h); PS the same products under a different
// (never smaller) scale: fp32-class against fp64 like every other launch of the arithmetic, not bit-identical.
constexpr int FFN_HST_SW = 68;                                 // PS: row pitch (floats) of a wave's 64 x 64 hidden staging tile
constexpr int FFN_PS_TAB = 4 * 64 * FFN_HST_SW * 4;            // PS: byte offset of the [4][64] chunk-amax table
template <bool PS>
__global__ __launch_bounds__(256, 2) void ffn_fwd16_kernel(const FfnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const himg = smem;                                   // x tile (8 stages) in phase 1, then the H image: 16 stages x FFN_HSTRIDE
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    int tile;
    {                                                          // consecutive tiles on one XCD (gemm_x3p_kernel's map)
        const int tiles = gridDim.x, q = tiles >> 3, r = tiles & 7;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int m0 = tile * FFN_BM;
    const uint32_t voff = lane * 16;

    // ------------------------------------------------------------------------------------------------- phase 1
    // The block's whole x tile (64 rows x 128 k = 8 stages of 4 KB) is requested at once into the region the H image will
    // occupy later -- no ring, no slot reuse, ONE wait + barrier for all of it -- and the B fragments of W1 run three stages
    // ahead in three register sets: with a ring of three stages and B one stage ahead a block spent its K loop waiting
    // (eight stages of 0.16 us of MFMA work each behind a ~1 us round trip; 215 us per launch).
    const int wn_u = __builtin_amdgcn_readfirstlane(wave);     // this wave's 64 hidden columns: 64 wn_u ..
    const char* bbase = reinterpret_cast<const char*>(p.Bp1) + (int64_t)(wn_u * 2) * p.KS1 * 1024;
    const int64_t bplane = (int64_t)p.NT1 * p.KS1 * 1024;
    f16x8 bs0[2][2], bs1[2][2], bs2[2][2];
    auto loadb = [&](int ks, f16x8 (&bn)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const char* sp = bbase + pl * bplane + ((int64_t)j * p.KS1 + ks) * 1024;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bn[j][pl]) : "v"(voff), "s"(sp));
            }
    };
    // bias of the lane's hidden columns 64 wave + 32 j + 8 g + 4 lh .. + 3 (the oldest requests of the block)
    f32x4 b1v[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            b1v[j][g] = p.b1 ? *reinterpret_cast<const f32x4*>(p.b1 + 64 * wave + 32 * j + 8 * g + 4 * lh) : f32x4{0.f, 0.f, 0.f, 0.f};
    {                                                          // A stages 0 .. 7: one load instruction per wave and stage
        const int row = 16 * wave + (lane >> 2), slot = lane & 3;
        const int g = slot ^ ((row >> 2) & 3);
        const int xr = m0 + row;
        const float* src0 = xr < p.T ? p.x + (int64_t)xr * p.d + 4 * g : ffn_zero;
        const int step = xr < p.T ? FFN_BK : 0;
#pragma unroll
        for (int st = 0; st < 8; ++st)
            __builtin_amdgcn_global_load_lds((ffn_glb_ptr)(src0 + st * step), (ffn_lds_ptr)(himg + st * FFN_HSTRIDE + wave * 1024), 16, 0, 0);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int ea[2] = {FFN_E0, FFN_E0};
    auto stage = [&](int kt, f16x8 (&bn)[2][2]) {
        const char* sa = himg + kt * FFN_HSTRIDE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v[8];
            ffn_frag(sa, 32 * i + lr, lh, v);
            ffn_tile_stage<2>(v, ea[i], ffn_alt(i), bn, acc[i]);
        }
    };
    // Request order of a wave: bias (8) | A(0..7) (8) | B(0) B(1) B(2) (12) | B(3) | B(4) | ...; behind B(k) sit B(k+1), B(k+2).
    loadb(0, bs0); loadb(1, bs1); loadb(2, bs2);
    FFN_WAIT_AB_(8, bs0);                                      // every A stage and B(0) have landed; the barrier shares the tile
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(b1v[j][g]));   // (older than everything waited for: no compiler wait later)
    stage(0, bs0); loadb(3, bs0);
    FFN_WAIT_B_(8, bs1); stage(1, bs1); loadb(4, bs1);
    FFN_WAIT_B_(8, bs2); stage(2, bs2); loadb(5, bs2);
    FFN_WAIT_B_(8, bs0); stage(3, bs0); loadb(6, bs0);
    FFN_WAIT_B_(8, bs1); stage(4, bs1); loadb(7, bs1);
    FFN_WAIT_B_(8, bs2); stage(5, bs2);
    FFN_WAIT_B_(4, bs0); stage(6, bs0);
    FFN_WAIT_B_(0, bs1); stage(7, bs1);

    // the fragments of W2 for the first four stages of phase 2 go out now, under epilogue 1
    const char* bbase2 = reinterpret_cast<const char*>(p.Bp2) + (int64_t)wn_u * p.KS2 * 1024;
    const int64_t bplane2 = (int64_t)p.NT2 * p.KS2 * 1024;
    f16x8 cn0[1][2], cn1[1][2], cn2[1][2], cn3[1][2];
    auto loadb2 = [&](int ks, f16x8 (&bn)[1][2]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const char* sp = bbase2 + pl * bplane2 + (int64_t)ks * 1024;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bn[0][pl]) : "v"(voff), "s"(sp));
        }
    };
    loadb2(0, cn0); loadb2(1, cn1); loadb2(2, cn2); loadb2(3, cn3);
    asm volatile("s_barrier" ::: "memory");                    // everybody is done with the x tile: the region becomes the H image

    // ------------------------------------------------------------------------------------------------- epilogue 1
    int er[2] = {0, 0};                                        // PS: the rows' exponents (this lane's row of tile i)
    if constexpr (PS) {
        const int* ebp = reinterpret_cast<const int*>(reinterpret_cast<const char*>(p.Bp1) + 2 * bplane) + wn_u * 2;
        const uint32_t key = drop_key_dev(p.dh);
        float* tab = reinterpret_cast<float*>(smem + FFN_PS_TAB);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 32 * i + lr;
            const uint32_t di0 = (uint32_t)((int64_t)(m0 + row) * p.f + 64 * wave + 4 * lh);
            float amax = 0.f;
            const int64_t widx = ((int64_t)(tile * 4 + wave) * 64 + lane) * 2 + i;
            const uint32_t bin = p.bwd ? p.bits_in[widx] : 0u;
            uint32_t bout = 0u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ffn_unscale(acc[i][j], ea[i], ebp[j], ffn_alt(i));
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {          // the hidden value, in place in the accumulator register
                        float y;
                        if (p.bwd) {                       // d(hidden): through the forward's ReLU / dropout decision
                            y = ((bin >> (16 * j + 4 * g + t)) & 1u) ? acc[i][j][4 * g + t] * p.e1_scale : 0.f;
                        } else {
                            y = acc[i][j][4 * g + t] + b1v[j][g][t];
                            if (p.act == GT_ACT_RELU) y = fmaxf(y, 0.f);
                            if (p.dh.thresh) y *= drop_mul(p.dh, key, di0 + 32 * j + 8 * g + t);
                            bout |= (y > 0.f ? 1u : 0u) << (16 * j + 4 * g + t);
                        }
                        acc[i][j][4 * g + t] = y;
                        amax = fmaxf(amax, fabsf(y));
                    }
            }
            if (p.bits_out) p.bits_out[widx] = bout;
            amax = xor32_max(amax);                        // the row's 64 columns of this wave sit in the lane pair
            if (lh == 0) tab[wave * 64 + row] = amax;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // table complete; everybody is done with the x tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 32 * i + lr;
            const float a = fmaxf(fmaxf(tab[row], tab[64 + row]), fmaxf(tab[128 + row], tab[192 + row]));
            const int ex = (int)(__float_as_uint(a) >> 23);
            er[i] = ex == 0 ? 0 : min(FFN_E0, FFN_TARGET + 127 - ex);      // scaled row amax in [2^13, 2^14)
            const float sv = ffn_pow2(er[i]) * ffn_alt(i);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t h0, l0, h1, l1;
                    f16_mulsplit_pair(acc[i][j][4 * g], sv, acc[i][j][4 * g + 1], sv, h0, l0);
                    f16_mulsplit_pair(acc[i][j][4 * g + 2], sv, acc[i][j][4 * g + 3], sv, h1, l1);
                    // hidden columns 64 wave + 32 j + 8 g + 4 lh .. + 3 = k-half g & 1 of stage 4 wave + 2 j + g / 2, second half of
                    // the fragment's eight k for lh = 1:  [plane][stage][k-half][row] x 16 B -- rows in consecutive 16-byte slots:
                    // the 8-byte stores of a wave cover 512 contiguous bytes and a fragment read is conflict-free (with the
                    // k-halves of a row side by side, at a row pitch of 32 bytes, both hit every bank group several times)
                    const int st = 4 * wave + 2 * j + (g >> 1);
                    char* dst = smem + (((st * 2 + (g & 1)) * 64 + row) * 16 + lh * 8);
                    *reinterpret_cast<uint2*>(dst) = uint2{h0, h1};
                    *reinterpret_cast<uint2*>(dst + 16 * 64 * 32) = uint2{l0, l1};
                }
        }
    } else {
        const int* ebp = reinterpret_cast<const int*>(reinterpret_cast<const char*>(p.Bp1) + 2 * bplane) + wn_u * 2;
        const uint32_t key = drop_key_dev(p.dh);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 32 * i + lr, swz = (row >> 2) & 3;
            const uint32_t di0 = (uint32_t)((int64_t)(m0 + row) * p.f + 64 * wave + 4 * lh);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ffn_unscale(acc[i][j], ea[i], ebp[j], ffn_alt(i));
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 h;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float y = acc[i][j][4 * g + t] + b1v[j][g][t];
                        if (p.act == GT_ACT_RELU) y = fmaxf(y, 0.f);
                        if (p.dh.thresh) y *= drop_mul(p.dh, key, di0 + 32 * j + 8 * g + t);
                        h[t] = y;
                    }
                    // hidden column c = 64 wave + 32 j + 8 g + 4 lh: stage c / 16, granule (c % 16) / 4
                    const int st = 4 * wave + 2 * j + (g >> 1), gq = 2 * (g & 1) + lh;
                    *reinterpret_cast<f32x4*>(himg + st * FFN_HSTRIDE + row * 64 + ((gq ^ swz) << 4)) = h;
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the H image is complete

    // ------------------------------------------------------------------------------------------------- phase 2
    // waves side by side over the OUTPUT columns (wave w: columns 32 w .. 32 w + 31, both 32-row tiles): every B fragment of
    // W2 is fetched by exactly one wave of the block; four register sets, three stages ahead (a stage is 0.08 us of MFMA work)
#define FFN_WAIT_B2_(N, bn) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bn[0][0]), "+v"(bn[0][1])::"memory")
    f32x16 acc2[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][0][e] = 0.f;
    int ea2[2] = {FFN_E0, FFN_E0};
    auto stage2 = [&](int kt2, f16x8 (&bn)[1][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (PS) {                                // finished fragments: no split, no exponent bookkeeping
                const char* fr = smem + (((kt2 * 2 + lh) * 64 + 32 * i + lr) * 16);
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(fr);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(fr + 16 * 64 * 32);
                acc2[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bn[0][0], a1, acc2[i][0], 0, 0, 0);     // h1 g0
                acc2[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bn[0][1], a0, acc2[i][0], 0, 0, 0);     // h0 g1
                acc2[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bn[0][0], a0, acc2[i][0], 0, 0, 0);     // h0 g0
            } else {
                float v[8];
                ffn_frag(himg + kt2 * FFN_HSTRIDE, 32 * i + lr, lh, v);
                ffn_tile_stage<1>(v, ea2[i], ffn_alt(i), bn, acc2[i]);
            }
        }
    };
    // behind B(k) sit B(k+1) .. B(k+3): 6 loads
    FFN_WAIT_B2_(6, cn0); stage2(0, cn0); loadb2(4, cn0);
    FFN_WAIT_B2_(6, cn1); stage2(1, cn1); loadb2(5, cn1);
    FFN_WAIT_B2_(6, cn2); stage2(2, cn2); loadb2(6, cn2);
    FFN_WAIT_B2_(6, cn3); stage2(3, cn3); loadb2(7, cn3);
    FFN_WAIT_B2_(6, cn0); stage2(4, cn0); loadb2(8, cn0);
    FFN_WAIT_B2_(6, cn1); stage2(5, cn1); loadb2(9, cn1);
    FFN_WAIT_B2_(6, cn2); stage2(6, cn2); loadb2(10, cn2);
    FFN_WAIT_B2_(6, cn3); stage2(7, cn3); loadb2(11, cn3);
    FFN_WAIT_B2_(6, cn0); stage2(8, cn0); loadb2(12, cn0);
    FFN_WAIT_B2_(6, cn1); stage2(9, cn1); loadb2(13, cn1);
    FFN_WAIT_B2_(6, cn2); stage2(10, cn2); loadb2(14, cn2);
    FFN_WAIT_B2_(6, cn3); stage2(11, cn3); loadb2(15, cn3);
    FFN_WAIT_B2_(6, cn0); stage2(12, cn0);
    FFN_WAIT_B2_(4, cn1); stage2(13, cn1);
    FFN_WAIT_B2_(2, cn2); stage2(14, cn2);
    FFN_WAIT_B2_(0, cn3); stage2(15, cn3);
#undef FFN_WAIT_B2_
    {
        const int* ebp2 = reinterpret_cast<const int*>(reinterpret_cast<const char*>(p.Bp2) + 2 * bplane2) + wn_u;
#pragma unroll
        for (int i = 0; i < 2; ++i) ffn_unscale(acc2[i][0], PS ? er[i] : ea2[i], ebp2[0], ffn_alt(i));
    }

    // ------------------------------------------------------------------------------------------------- epilogue 2
    // read side of the wave's 64 x 32 out tile: 8 lanes per row (128 bytes), 8 rows per instruction, 8 instructions
    const int c8 = lane & 7, rsub = lane >> 3;
    const int ocol = 32 * wave + 4 * c8;
    f32x4 rs[8];
    if (p.res) {                                               // residual rows first: in front of the hidden tile's stores
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = m0 + 8 * it + rsub;
            rs[it] = m < p.T ? *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.d + ocol) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const f32x4 b2v = p.b2 ? *reinterpret_cast<const f32x4*>(p.b2 + ocol) : f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PS) {
        // hidden tile -> HBM out of the phase-1 accumulator registers: the wave's 64 x 64 tile through its private transposing
        // tile (over the dead planes), then 256-byte row segments, four rows per instruction
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everybody is done with the planes
        float* hs = reinterpret_cast<float*>(smem) + wave * (64 * FFN_HST_SW);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(hs + (32 * i + lr) * FFN_HST_SW + 32 * j + 8 * g + 4 * lh) =
                        f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int c4 = lane & 15, r4 = lane >> 4;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = 4 * it + r4;
            const f32x4 h = *reinterpret_cast<const f32x4*>(hs + row * FFN_HST_SW + 4 * c4);
            if (m0 + row < p.T) *reinterpret_cast<f32x4*>(p.hid + (int64_t)(m0 + row) * p.f + 64 * wave + 4 * c4) = h;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the tile is read out: the wave's region becomes the out staging
    } else {
    // hidden tile -> HBM: wave w stores rows 16 w .. 16 w + 15, a row = 1 KB = one instruction (lane l: columns 4 l ..)
    {
        const int st = lane >> 2, gq = lane & 3;
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr) {
            const int row = 16 * wave + rr;
            const f32x4 h = *reinterpret_cast<const f32x4*>(himg + st * FFN_HSTRIDE + row * 64 + ((gq ^ ((row >> 2) & 3)) << 4));
            if (m0 + row < p.T) *reinterpret_cast<f32x4*>(p.hid + (int64_t)(m0 + row) * p.f + 4 * lane) = h;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everybody is done with the H image: it becomes staging
    }
    {
        float* stg = reinterpret_cast<float*>(smem) + wave * (PS ? 64 * FFN_HST_SW : 64 * FFN_EP_SW);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(stg + (32 * i + lr) * FFN_EP_SW + 8 * g + 4 * lh) =
                    f32x4{acc2[i][0][4 * g], acc2[i][0][4 * g + 1], acc2[i][0][4 * g + 2], acc2[i][0][4 * g + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private tile, LDS operations of one wave execute in order
        const uint32_t key = drop_key_dev(p.dout);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = 8 * it + rsub, m = m0 + r;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * FFN_EP_SW + 4 * c8);
            const uint32_t di = (uint32_t)((int64_t)m * p.d + ocol);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float y = v[t] + b2v[t];
                if (p.dout.thresh) y *= drop_mul(p.dout, key, di + t);
                v[t] = p.res ? rs[it][t] + y : y;
            }
            if (m < p.T) *reinterpret_cast<f32x4*>(p.out + (int64_t)m * p.d + ocol) = v;
            if (p.out2) {                                  // the same rows under a second mask (gt_gemm_desc.c_masked's twin)
                const uint32_t key2 = drop_key_dev(p.d2);
                if (p.d2.thresh) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] *= drop_mul(p.d2, key2, di + t);
                }
                if (m < p.T) *reinterpret_cast<f32x4*>(p.out2 + (int64_t)m * p.d + ocol) = v;
            }
        }
    }
}

}  // namespace gt

using namespace gt;

extern "C" int64_t gt_ffn_fwd_ws_bytes(int64_t T, int32_t d, int32_t f) {
    gt_gemm_desc a, b;
    gt_gemm_desc_init(&a); gt_gemm_desc_init(&b);
    a.M = b.M = (int32_t)std::min<int64_t>(T, 1 << 30);
    a.N = f; a.K = d; a.lda = d; a.ldb = d; a.ldc = f; a.precision = GT_PREC_F16X2;
    b.N = d; b.K = f; b.lda = f; b.ldb = f; b.ldc = d; b.precision = GT_PREC_F16X2;
    a.A = a.B = b.A = b.B = reinterpret_cast<const float*>(uintptr_t(16));     // alignment is what the query looks at
    a.C = b.C = reinterpret_cast<float*>(uintptr_t(16));
    const int64_t pa = gt_gemm_packed_b_bytes(&a), pb = gt_gemm_packed_b_bytes(&b);
    return (pa > 0 && pb > 0) ? pa + pb : 0;
}

static int ffn_launch(FfnP& p, const gt_gemm_desc& a, const gt_gemm_desc& b, const void* p1, const void* p2, void* ws,
                      int64_t ws_bytes, void* stream) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if ((p1 == nullptr) != (p2 == nullptr)) return GT_EINVAL;
    const int64_t pa = gt_gemm_packed_b_bytes(&a), pb = gt_gemm_packed_b_bytes(&b);
    if (pa <= 0 || pb <= 0) return GT_ENOTSUP;
    if (!p1) {                                                 // pack both weights into the scratch (one launch)
        if (!ws || ws_bytes < pa + pb || !al(ws)) return GT_EWS;
        gt_gemm_desc two[2] = {a, b};
        void* outs[2] = {ws, reinterpret_cast<char*>(ws) + pa};
        if (int rc = gt_gemm_pack_b_many(two, outs, 2, stream)) return rc;
        p1 = outs[0]; p2 = outs[1];
    } else if (!al(p1) || !al(p2)) return GT_EALIGN;
    p.Bp1 = p1; p.Bp2 = p2;
    static const int presplit = [] { const char* e = getenv("GT_FFN_PRESPLIT"); return e ? atoi(e) : 1; }();
    if (!presplit && (p.bwd || p.bits_out)) return GT_ENOTSUP;       // the decision bits exist in the pre-split form only
    const size_t lds0 = (size_t)16 * FFN_HSTRIDE, lds1 = (size_t)FFN_PS_TAB + 1024;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fwd16_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fwd16_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const unsigned tiles = (unsigned)(((int64_t)p.T + FFN_BM - 1) / FFN_BM);
    hipStream_t st = (hipStream_t)stream;
    if (presplit) hipLaunchKernelGGL(ffn_fwd16_kernel<true>, dim3(tiles), dim3(256), lds1, st, p);
    else hipLaunchKernelGGL(ffn_fwd16_kernel<false>, dim3(tiles), dim3(256), lds0, st, p);
    GT_LAUNCH_CHECK();
    return 0;
}

static bool ffn_shape_ok(int64_t T, int32_t d, int32_t f) { return d == 128 && f == 256 && T >= 16384 && T <= (1 << 30); }

extern "C" int64_t gt_ffn_bits_bytes(int64_t T) { return T > 0 ? ((T + FFN_BM - 1) / FFN_BM) * 4 * 64 * 2 * (int64_t)sizeof(uint32_t) : 0; }

extern "C" int gt_ffn_fwd(const float* x, int64_t T, int32_t d, int32_t f, const float* W1, const float* b1,
                          const float* W2, const float* b2, const float* res, const gt_dropout* drop_h,
                          const gt_dropout* drop_o, int32_t act, float* hid, float* out, void* relu_bits,
                          const void* w1_packed, const void* w2_packed, void* ws, int64_t ws_bytes, void* stream) {
    if (!x || !W1 || !W2 || !hid || !out || T <= 0) return GT_EINVAL;
    if (!ffn_shape_ok(T, d, f) || (act != GT_ACT_RELU && act != GT_ACT_NONE)) return GT_ENOTSUP;
    if ((drop_h && drop_h->p > 0.f && !drop_h->seed) || (drop_o && drop_o->p > 0.f && !drop_o->seed)) return GT_EINVAL;
    if ((drop_h && (drop_h->p < 0.f || drop_h->p >= 1.f)) || (drop_o && (drop_o->p < 0.f || drop_o->p >= 1.f))) return GT_EINVAL;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al(x) || !al(W1) || !al(W2) || !al(hid) || !al(out) || !al(res) || !al(b1) || !al(b2) || !al(relu_bits)) return GT_EALIGN;
    gt_gemm_desc a, b;
    gt_gemm_desc_init(&a); gt_gemm_desc_init(&b);
    a.M = b.M = (int32_t)T;
    a.N = f; a.K = d; a.A = x; a.lda = d; a.B = W1; a.ldb = d; a.C = hid; a.ldc = f; a.precision = GT_PREC_F16X2;
    b.N = d; b.K = f; b.A = hid; b.lda = f; b.B = W2; b.ldb = f; b.C = out; b.ldc = d; b.precision = GT_PREC_F16X2;
    FfnP p{x, res, hid, out, b1, b2, nullptr, nullptr, (int)T, d, f,
           d / 16, ((f + 127) / 128) * 4, f / 16, ((d + 127) / 128) * 4, make_drop(drop_h), make_drop(drop_o), act,
           reinterpret_cast<uint32_t*>(relu_bits), nullptr, 0, 1.f, nullptr, make_drop(nullptr)};
    return ffn_launch(p, a, b, w1_packed, w2_packed, ws, ws_bytes, stream);
}

extern "C" int gt_ffn_bwd(const float* gm, int64_t T, int32_t d, int32_t f, const float* W2, const float* W1,
                          const void* relu_bits, float hid_scale, const float* res, float* gh, float* dx, float* dx_masked,
                          const gt_dropout* mask2, const void* w2_packed, const void* w1_packed, void* ws, int64_t ws_bytes,
                          void* stream) {
    if (!gm || !W1 || !W2 || !relu_bits || !gh || !dx || T <= 0) return GT_EINVAL;
    if (!ffn_shape_ok(T, d, f)) return GT_ENOTSUP;
    if (mask2 && (mask2->p < 0.f || mask2->p >= 1.f || (mask2->p > 0.f && !mask2->seed))) return GT_EINVAL;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al(gm) || !al(W1) || !al(W2) || !al(gh) || !al(dx) || !al(res) || !al(dx_masked) || !al(relu_bits)) return GT_EALIGN;
    // gh [T, f] = gm [T, d] W2  (B(k, n) = W2[k * f + n]: layout_b = 1);  dx [T, d] = gh W1  (B(k, n) = W1[k * d + n])
    gt_gemm_desc a, b;
    gt_gemm_desc_init(&a); gt_gemm_desc_init(&b);
    a.M = b.M = (int32_t)T;
    a.N = f; a.K = d; a.A = gm; a.lda = d; a.B = W2; a.ldb = f; a.layout_b = 1; a.C = gh; a.ldc = f; a.precision = GT_PREC_F16X2;
    b.N = d; b.K = f; b.A = gh; b.lda = f; b.B = W1; b.ldb = d; b.layout_b = 1; b.C = dx; b.ldc = d; b.precision = GT_PREC_F16X2;
    FfnP p{gm, res, gh, dx, nullptr, nullptr, nullptr, nullptr, (int)T, d, f,
           d / 16, ((f + 127) / 128) * 4, f / 16, ((d + 127) / 128) * 4, make_drop(nullptr), make_drop(nullptr), GT_ACT_NONE,
           nullptr, reinterpret_cast<const uint32_t*>(relu_bits), 1, hid_scale, dx_masked, make_drop(mask2)};
    return ffn_launch(p, a, b, w2_packed, w1_packed, ws, ws_bytes, stream);
}
