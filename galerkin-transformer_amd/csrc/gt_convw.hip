// Weight gradient of a narrow channels-last 3x3 convolution (the down-scaler's 128->42, 42->42, 42->44 convolutions,
// reference libs/layers.py:88-150 Conv2dResBlock inside Interp2dEncoder, layers.py:463-482):
//
//     dW[co][ci][dy+1][dx+1] = alpha * sum_{b,y,x} gy[b][y][x][co] * x[b][y+dy][x+dx][ci]            (zero padding)
//
// Both operands are activations, one side is <= 48 channels wide and all nine taps read the same two tensors: the
// 128 x 128 tile engine (gt_gemm_x3.hip: one tap per block, both operands split again in every block) fits none of that.
// Here a block owns a run of image rows of one image and a block of <= 48 input channels, and works a row at a time:
//
//  * Split once.  Each staged value is split exactly into its three bf16 planes (the arithmetic of gt_gemm_x3.hip: six plane
//    products per stage, fp32 accumulation) ONCE, by the thread that fetched it, and the planes live in LDS: a value of x
//    is then used by 9 taps x Cout/16 tiles, a value of gy by 9 taps x CI/16 tiles, straight out of LDS as MFMA operands.
//  * Pixel-interleaved k.  v_mfma_f32_16x16x32_bf16 takes 8 consecutive k per lane.  k is the pixel x of the current image
//    row, but lane group q's eight elements are the pixels  q, q + 10, q + 20, .., q + 70  (10 groups cover an 80-pixel
//    row), stored as one 16-byte unit [plane][q][channel].  A horizontal tap shift is then a WHOLE-UNIT shift: the gy
//    operand of tap dx for pixel group q is the unit q - dx, one aligned ds_read_b128 like every other fragment (two halo
//    units, q = -1 and q = 10, are staged for it); the vertical shift selects the x row (a ring of three rows).  Within a
//    group of 16 lanes the units of one q are 16 consecutive 16-byte slots: no bank conflict for any fragment read.
//  * One wave per dy.  Wave w accumulates the three taps (dy = w - 1, dx = -1, 0, 1) for all CI x Cout of the block: its x
//    fragments (row y + dy) serve three taps, the gy fragments serve CI/16 tiles.  3 x (CI/16) x (Cout/16) accumulator tiles
//    per wave.
//  * Rows y + 2 / y + 1 of the next iteration are fetched into registers before the MFMAs of row y and written (split) to LDS
//    behind them; with CI = 32 two blocks share a CU and one computes while the other stages.
//  * Sign-alternating accumulation (gt_gemm_x3.hip: GT_X3_ALT): channels at odd LDS positions enter negated on both sides,
//    the accumulators are un-flipped when the block writes its partial result.
//
// Partial results go to slabs [image, row chunk][tap][ci][co] and are summed in a fixed order by convw_reduce_kernel, which
// also transposes to the reference's [co][ci][3][3] and applies alpha: deterministic, no atomics.
#include <algorithm>
#include <cstdlib>

#include "gt_common.h"

namespace gt {

typedef __bf16 cw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cw_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t cw_u32x4 __attribute__((ext_vector_type(4)));

constexpr int CW_THREADS = 192;        // three waves: dy = -1, 0, 1
constexpr int CW_NQ = 10;              // pixel groups (units) per row and channel: 80 pixels
constexpr int CW_MAXW = 8 * CW_NQ;

struct ConvWP {
    const float* gy; int64_t ldg;      // [B*H*W][>= Cout] pixel pitch ldg
    const float* x; int64_t ldx;       // [B*H*W][>= Cin]
    float* slabs;                      // [B * chunks][9][Cin][Cout]
    int B, H, W, Cin, Cout, chunks, rows_per_chunk;
};

// channel c of a C-channel block sits at LDS position (c & 3) * (C / 4) + (c >> 2): the four channels of a staged float4
// land C/4 units apart, so the ds_write_b32 of a half-wave (eight channel groups x four dwords) cover all 32 banks
__host__ __device__ __forceinline__ int cw_chan_of_pos(int pos, int C) { return 4 * (pos % (C / 4)) + pos / (C / 4); }

__device__ __forceinline__ void cw_split3(float a, float b, uint32_t (&out)[3]) {       // two values -> three packed bf16 pairs
    f32x2 r = {a, b};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const cw_bf16x2 h = __builtin_convertvector(r, cw_bf16x2);
        out[pl] = __builtin_bit_cast(uint32_t, h);
        if (pl < 2) r = r - __builtin_convertvector(h, f32x2);
    }
}

// One staged row of one operand: NITEM float4 pairs per thread.  Item idx = (g, m, q), g fastest: dword m of unit q of the four
// channels 4g .. 4g+3, i.e. the pixels  q + QOFF + 20 m  and  + 10  (QOFF = -1 for gy: its units run q = -1 .. 10).  A wave's
// loads walk the contiguous channel groups of a pixel; its ds_write_b32 (positions c * C/4 + g) spread over the banks.
template <int C, int NQ, int QOFF>
struct CwStage {
    static constexpr int G4 = C / 4;
    static constexpr int ITEMS = 4 * G4 * NQ;
    static constexpr int NITEM = (ITEMS + CW_THREADS - 1) / CW_THREADS;
    f32x4 v[NITEM][2];

    // rowp: first pixel of the image row (channel offset applied), or nullptr for a row outside the image
    __device__ __forceinline__ void load(const float* __restrict__ rowp, int64_t ld, int W, int tid) {
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int idx = tid + it * CW_THREADS;
            const int g = idx % G4, m = (idx / G4) & 3, q = idx / (4 * G4);
            const int px0 = q + QOFF + 20 * m, px1 = px0 + 10;
            const bool ok = rowp != nullptr && idx < ITEMS;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v[it][0] = (ok && px0 >= 0 && px0 < W) ? *reinterpret_cast<const f32x4*>(rowp + (int64_t)px0 * ld + 4 * g) : z;
            v[it][1] = (ok && px1 < W) ? *reinterpret_cast<const f32x4*>(rowp + (int64_t)px1 * ld + 4 * g) : z;
        }
    }
    // planes: [3][NQ][C] units of 16 bytes
    __device__ __forceinline__ void store(char* __restrict__ planes, int tid) const {
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int idx = tid + it * CW_THREADS;
            if (idx >= ITEMS) continue;
            const int g = idx % G4, m = (idx / G4) & 3, q = idx / (4 * G4);
            const float sg = (g & 1) ? -1.f : 1.f;                 // position parity = g parity (C / 4 is even)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t h[3];
                cw_split3(sg * v[it][0][c], sg * v[it][1][c], h);
                const int pos = c * G4 + g;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<uint32_t*>(planes + (((pl * NQ + q) * C + pos) << 4) + 4 * m) = h[pl];
            }
        }
    }
};

template <int CIT, int COT>
__global__ __launch_bounds__(CW_THREADS, (CIT == 3 ? 1 : 2)) void convw_kernel(const ConvWP p) {
    constexpr int CI = 16 * CIT, CO = 16 * COT;
    constexpr int XSLOT = 3 * CW_NQ * CI * 16;            // bytes of one x row (three planes)
    constexpr int YBUF = 3 * (CW_NQ + 2) * CO * 16;       // the gy row with its two halo units
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                                       // [3 slots][3 planes][NQ][CI] units
    char* ys = smem + 3 * XSLOT;                           // [3 planes][NQ + 2][CO] units
    char* zs = ys + YBUF;                                  // 16 zero bytes: the operands of pixel groups past the row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int ci0 = blockIdx.y * CI;
    const int bc = blockIdx.x, b = bc / p.chunks, chunk = bc - b * p.chunks;
    const int y0 = chunk * p.rows_per_chunk, y1 = min(p.H, y0 + p.rows_per_chunk);
    const int dy = wave - 1;

    for (int i = tid; i < (3 * XSLOT + YBUF + 16) / 16; i += CW_THREADS)
        reinterpret_cast<cw_u32x4*>(smem)[i] = cw_u32x4{0u, 0u, 0u, 0u};

    auto xrow = [&](int y) -> const float* {
        return (y >= 0 && y < p.H) ? p.x + ((int64_t)(b * p.H + y) * p.W) * p.ldx + ci0 : nullptr;
    };
    auto grow = [&](int y) -> const float* {
        return (y >= 0 && y < p.H) ? p.gy + ((int64_t)(b * p.H + y) * p.W) * p.ldg : nullptr;
    };
    CwStage<CI, CW_NQ, 0> sx;
    CwStage<CO, CW_NQ + 2, -1> sy;
    __syncthreads();
    // prologue: x rows y0 - 1, y0, y0 + 1 and gy row y0
    for (int r = -1; r <= 1; ++r) {
        sx.load(xrow(y0 + r), p.ldx, p.W, tid);
        sx.store(xs + ((y0 + r + 3) % 3) * XSLOT, tid);
    }
    sy.load(grow(y0), p.ldg, p.W, tid);
    sy.store(ys, tid);
    __syncthreads();

    f32x4 acc[3][CIT][COT];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[d][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int y = y0; y < y1; ++y) {
        const bool more = y + 1 < y1;
        if (more) {                                        // block-uniform: next iteration's rows on their way
            sx.load(xrow(y + 2), p.ldx, p.W, tid);
            sy.load(grow(y + 1), p.ldg, p.W, tid);
        }
        const char* xr = xs + ((y + dy + 3) % 3) * XSLOT;
#pragma unroll 1
        for (int ks = 0; ks < 3; ++ks) {
            const int q = 4 * ks + kq;
            const bool valid = q < CW_NQ;
            cw_bf16x8 a[CIT][3];
#pragma unroll
            for (int i = 0; i < CIT; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const char* ap = valid ? xr + (((pl * CW_NQ + q) * CI + 16 * i + li) << 4) : zs;
                    a[i][pl] = *reinterpret_cast<const cw_bf16x8*>(ap);
                }
#pragma unroll
            for (int d = 0; d < 3; ++d) {                  // dx = d - 1: gy unit q - dx, stored at unit index q - dx + 1
#pragma unroll
                for (int j = 0; j < COT; ++j) {
                    cw_bf16x8 bq[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const char* bp = valid ? ys + (((pl * (CW_NQ + 2) + (q + 2 - d)) * CO + 16 * j + li) << 4) : zs;
                        bq[pl] = *reinterpret_cast<const cw_bf16x8*>(bp);
                    }
#pragma unroll
                    for (int s = 2; s >= 0; --s)           // plane pairs, smallest terms first (gt_gemm_x3.hip)
#pragma unroll
                        for (int pa = 0; pa < 3; ++pa) {
                            const int pb = s - pa;
                            if (pb < 0 || pb > 2) continue;
#pragma unroll
                            for (int i = 0; i < CIT; ++i)
                                acc[d][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][pa], bq[pb], acc[d][i][j], 0, 0, 0);
                        }
                }
            }
        }
        __syncthreads();                                   // every wave is done with row y's operands
        if (more) {
            sx.store(xs + ((y + 2) % 3) * XSLOT, tid);     // over row y - 1
            sy.store(ys, tid);
        }
        __syncthreads();
    }

    // partial result: accumulator register r of lane (li, kq) = (x position 16 i + 4 kq + r, gy position 16 j + li)
    float* slab = p.slabs + (int64_t)bc * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
            for (int j = 0; j < COT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pa = 16 * i + 4 * kq + r, pb = 16 * j + li;
                    const int ci = ci0 + cw_chan_of_pos(pa, CI), co = cw_chan_of_pos(pb, CO);
                    const float sg = ((r + li) & 1) ? -1.f : 1.f;      // position parities (16 i + 4 kq and 16 j are even)
                    if (ci < p.Cin && co < p.Cout)
                        slab[((int64_t)((dy + 1) * 3 + d) * p.Cin + ci) * p.Cout + co] = sg * acc[d][i][j][r];
                }
}

// dw[co][ci][tap] = alpha * sum_s slabs[s][tap][ci][co]
__global__ __launch_bounds__(256) void convw_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int Cin, int Cout,
                                                           float alpha, float* __restrict__ dw) {
    const int e = blockIdx.x * 256 + threadIdx.x, n = 9 * Cin * Cout;
    if (e >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = 0;
    for (; s + 3 < n_slabs; s += 4) {
        s0 += slabs[(int64_t)s * n + e];
        s1 += slabs[(int64_t)(s + 1) * n + e];
        s2 += slabs[(int64_t)(s + 2) * n + e];
        s3 += slabs[(int64_t)(s + 3) * n + e];
    }
    for (; s < n_slabs; ++s) s0 += slabs[(int64_t)s * n + e];
    const int co = e % Cout, ci = (e / Cout) % Cin, tap = e / (Cout * Cin);
    dw[((int64_t)co * Cin + ci) * 9 + tap] = alpha * ((s0 + s1) + (s2 + s3));
}

struct CwPlan { int cit, ciblocks, chunks, rows; size_t lds; };

static bool cw_plan(int B, int H, int W, int Cin, int Cout, CwPlan* pl) {
    if (B <= 0 || H <= 0 || W <= 0 || W > CW_MAXW || Cin <= 0 || Cout <= 0 || (Cin & 15) || Cout != 48) return false;
    // CI = 32: two blocks per CU (one stages while the other multiplies); 48-channel inputs run as one block of three tiles
    pl->cit = (Cin % 32 == 0) ? 2 : (Cin % 48 == 0) ? 3 : 1;
    pl->ciblocks = Cin / (16 * pl->cit);
    const int per_cu = pl->cit == 3 ? 1 : 2, want = 256 * per_cu;
    int chunks = std::max(1, (want + B * pl->ciblocks - 1) / (B * pl->ciblocks));
    chunks = std::min(chunks, std::max(1, H / 8));          // at least eight rows per block: the three-row prologue is paid once
    pl->rows = (H + chunks - 1) / chunks;
    pl->chunks = (H + pl->rows - 1) / pl->rows;
    pl->lds = (size_t)3 * 3 * CW_NQ * 16 * pl->cit * 16 + (size_t)3 * (CW_NQ + 2) * Cout * 16 + 16;
    return true;
}

}  // namespace gt

using namespace gt;

extern "C" int64_t gt_conv3x3_wgrad_nhwc_ws_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    CwPlan pl;
    if (!cw_plan(B, H, W, Cin, Cout, &pl)) return 0;
    return (int64_t)B * pl.chunks * 9 * Cin * Cout * (int64_t)sizeof(float);
}

extern "C" int gt_conv3x3_wgrad_nhwc(const float* gy, int64_t ldg, const float* x, int64_t ldx, float* dw, int32_t B,
                                     int32_t H, int32_t W, int32_t Cin, int32_t Cout, float alpha, void* ws,
                                     int64_t ws_bytes, void* stream) {
    if (!gy || !x || !dw || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldg < Cout || ldx < Cin) return GT_EINVAL;
    CwPlan pl;
    if (!cw_plan(B, H, W, Cin, Cout, &pl)) return GT_ENOTSUP;
    if (((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(x)) & 15) || (ldg & 3) || (ldx & 3)) return GT_EALIGN;
    if (!ws || ws_bytes < gt_conv3x3_wgrad_nhwc_ws_bytes(B, H, W, Cin, Cout)) return GT_EWS;
    if ((int64_t)B * pl.chunks > 65535LL * 32768) return GT_EINVAL;
    ConvWP p{gy, ldg, x, ldx, reinterpret_cast<float*>(ws), B, H, W, Cin, Cout, pl.chunks, pl.rows};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(B * pl.chunks), (unsigned)pl.ciblocks);
    // more than 64 KB of LDS per block: the limit is raised once per kernel instance
    static bool raised[4] = {false, false, false, false};
    auto launch = [&](auto kern, int idx) -> int {
        if (!raised[idx]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)pl.lds) != hipSuccess)
                return GT_ENOTSUP;
            raised[idx] = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(CW_THREADS), pl.lds, st, p);
        return 0;
    };
    int rc;
    if (pl.cit == 1) rc = launch(convw_kernel<1, 3>, 1);
    else if (pl.cit == 2) rc = launch(convw_kernel<2, 3>, 2);
    else rc = launch(convw_kernel<3, 3>, 3);
    if (rc) return rc;
    GT_LAUNCH_CHECK();
    const int n = 9 * Cin * Cout;
    hipLaunchKernelGGL(convw_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.slabs, B * pl.chunks, Cin, Cout, alpha,
                       dw);
    GT_LAUNCH_CHECK();
    return 0;
}
