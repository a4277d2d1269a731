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
//  * Two loader waves, one per operand (wave 0: x rows, wave 4: gy rows): while the three MFMA waves (1..3) work on row y they
//    fetch, split and write x row y + 2 (into the free slot of a ring of four) and gy row y + 1 (second buffer); one barrier per
//    row.  A workgroup's waves are dealt to the SIMDs cyclically, so waves 0 and 4 share the SIMD the three MFMA waves leave
//    free: their VALU work interleaves there (one loader wave alone sustained ~6.5 clk per instruction and, holding a whole
//    row pair in registers, spilled to scratch: 204 us for the 48-channel launch against 144 us of matrix time; staged by the
//    MFMA waves themselves 307 us; four loader waves sharing the MFMA waves' SIMDs 280 us).
//  * Sign-alternating accumulation (gt_gemm_x3.hip: GT_X3_ALT): channels at odd LDS positions enter negated on both sides,
//    the accumulators are un-flipped when the block writes its partial result.
//
//  * Two arithmetics (template parameter F16): the three bf16 planes / six products of GT_PREC_BF16X3, or the two fp16 planes /
//    three products of GT_PREC_F16X2 (gt_gemm_x3.hip) -- two planes are a third less loader work and LDS, which buys the wide
//    channel blocks (64 input channels), and half the matrix work.  Its power-of-two scale is one running exponent per operand
//    and BLOCK, kept by the loader: it takes the amax of every row it stages (wave reduce), lowers the exponent when the scaled
//    amax would reach 2^15, splits the row with it and publishes the row's exponent next to the planes; an MFMA wave compares
//    the exponent sum of the two rows it is about to multiply with the one its accumulators carry and rescales them when it
//    has dropped (monotone per wave, so rare).
//
// Partial results go to slabs [image, row chunk][tap][ci][co] and are summed in a fixed order by convw_reduce_kernel, which
// also transposes to the reference's [co][ci][3][3] and applies alpha: deterministic, no atomics.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gt_common.h"
#include <atomic>

namespace gt {

typedef __bf16 cw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cw_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t cw_u32x4 __attribute__((ext_vector_type(4)));
// pointers into device memory as such (inside the non-inlined role functions the compiler no longer sees where they came from
// and would fall back to flat loads / stores)
typedef __attribute__((address_space(1))) f32x4 cw_gf32x4;
typedef __attribute__((address_space(1))) float cw_gf32;

#ifndef GT_CW_ABL
#define GT_CW_ABL 0
#endif
constexpr int CW_LOADERS = 64;          // threads of one loader wave
constexpr int CW_THREADS = 320;         // wave 0: the x loader, waves 1..3: dy = -1, 0, 1 (MFMA), wave 4: the gy loader
constexpr int CW_NQ = 10;              // pixel groups (units) per row and channel: 80 pixels
constexpr int CW_MAXW = 8 * CW_NQ;

struct ConvWP {
    const float* gy; int64_t ldg;      // [B*H*W][>= Cout] pixel pitch ldg
    const float* x; int64_t ldx;       // [B*H*W][>= Cin]
    float* slabs;                      // [B * chunks * nseg][9][Cin][Cout]
    int B, H, W, Cin, Cout, chunks, rows_per_chunk;
    int nseg, seg_w;                   // rows wider than 80 pixels are worked in nseg segments of seg_w x-pixels (round 5)
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

typedef _Float16 cw_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cw_f16x2 __attribute__((ext_vector_type(2)));
constexpr int CWH_E0 = 120, CWH_TARGET = 13, CWH_LIMIT = 15;          // as X3H_* in gt_gemm_x3.hip
__device__ __forceinline__ float cw_pow2(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }
__device__ __forceinline__ void cw_split2h(float a, float b, uint32_t (&out)[3]) {      // two scaled values -> two packed fp16 pairs
    const f32x2 r = {a, b};
    const cw_f16x2 h0 = __builtin_convertvector(r, cw_f16x2);
    const cw_f16x2 h1 = __builtin_convertvector(r - __builtin_convertvector(h0, f32x2), cw_f16x2);
    out[0] = __builtin_bit_cast(uint32_t, h0);
    out[1] = __builtin_bit_cast(uint32_t, h1);
}

__device__ __attribute__((aligned(16))) float cw_zero[4] = {0.f, 0.f, 0.f, 0.f};

// One staged row of one operand: NITEM float4 quadruples per thread.  Item idx = (g, mm, q), g fastest: dwords 2 mm, 2 mm + 1 of
// unit q of the four channels 4g .. 4g+3, i.e. the pixels  q + QOFF + 40 mm + {0, 10, 20, 30}  (QOFF = -1 for gy: its units run
// q = -1 .. 10).  A wave's loads walk the contiguous channel groups of a pixel; its ds_write_b64 (unit positions c * C/4 + g)
// spread over the banks.  Everything that does not change from row to row (element offsets of the sixteen pixels, the LDS
// offset, the sign) is worked out once by init(); a pixel outside the row reads the zero line instead of sitting under a branch.
template <int C, int NQ, int QOFF, int NT, int PL>      // NT: threads that share the row; PL: planes (3 bf16 / 2 fp16)
struct CwStage {
    static constexpr int G4 = C / 4;
    static constexpr int ITEMS = 2 * G4 * NQ;
    static constexpr int NITEM = (ITEMS + NT - 1) / NT;
    f32x4 v[NITEM][4];
    int off0[NITEM];                                       // element offset of the item's first pixel from the row's first pixel
    int vm[NITEM];                                         // bit e: pixel e of the item exists
    int lds[NITEM];                                        // byte offset of (unit q, position g, dword pair mm) in plane 0, or -1
    int stride10;

    // x0, S: the block's segment [x0, x0 + S) of the row (S <= 80).  The x operand (QOFF = 0) stages exactly those pixels; the
    // gy operand (QOFF = -1) also its two neighbours x0 - 1 and x0 + S -- the halo units of the horizontal taps: real data
    // where the neighbour is inside the picture (another block's segment), zero outside.
    __device__ __forceinline__ void init(int64_t ld, int W, int tid, int x0, int S) {
        stride10 = (int)(10 * ld);
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int idx = tid + it * NT;
            const int g = idx % G4, mm = (idx / G4) & 1, q = idx / (2 * G4);
            const int px0 = q + QOFF + 40 * mm;
            off0[it] = (x0 + px0) * (int)ld + 4 * g;
            int m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ps = px0 + 10 * e;               // position in the segment
                m |= (idx < ITEMS && ps >= QOFF && ps < S - QOFF && x0 + ps >= 0 && x0 + ps < W) ? (1 << e) : 0;
            }
            vm[it] = m;
            lds[it] = idx < ITEMS ? ((q * C + g) << 4) + 8 * mm : -1;
        }
    }
    // rowp: first pixel of the image row (channel offset applied), or nullptr (wave-uniform) for a row outside the image
    __device__ __forceinline__ void load(const float* __restrict__ rowp) {
        if (rowp == nullptr) {
#pragma unroll
            for (int it = 0; it < NITEM; ++it)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[it][e] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
#pragma unroll
        for (int it = 0; it < NITEM; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* src = ((vm[it] >> e) & 1) ? rowp + (off0[it] + e * stride10) : cw_zero;
                v[it][e] = *(const cw_gf32x4*)src;
            }
    }
    // amax of the row in flight over this thread's items (the loader wave-reduces it)
    __device__ __forceinline__ float amax() const {
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < NITEM; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v[it][e][0]), fabsf(v[it][e][1])), fmaxf(fabsf(v[it][e][2]), fabsf(v[it][e][3]))));
        return m;
    }
    // planes: [PL][NQ][C] units of 16 bytes; scale: 1 (bf16 planes) or the block's power of two (fp16 planes)
    __device__ __forceinline__ void store(char* __restrict__ planes, float scale) const {
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            if (lds[it] < 0) continue;
            char* base = planes + lds[it];
            const float sg = (lds[it] & 16) ? -scale : scale;      // position parity = g parity (C / 4 is even)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t h0[3], h1[3];
                if (PL == 3) {
                    cw_split3(sg * v[it][0][c], sg * v[it][1][c], h0);
                    cw_split3(sg * v[it][2][c], sg * v[it][3][c], h1);
                } else {                                    // scale and split in four v_fma_mix*_f16 per pair (gt_common.h)
                    f16_mulsplit_pair(v[it][0][c], sg, v[it][1][c], sg, h0[0], h0[1]);
                    f16_mulsplit_pair(v[it][2][c], sg, v[it][3][c], sg, h1[0], h1[1]);
                }
#pragma unroll
                for (int pl = 0; pl < PL; ++pl)
                    *reinterpret_cast<uint2*>(base + ((pl * NQ * C + c * G4) << 4)) = uint2{h0[pl], h1[pl]};
            }
        }
    }
};

template <int CIT, int COT, int F16>
struct CwGeom {
    static constexpr int CI = 16 * CIT, CO = 16 * COT, PL = F16 ? 2 : 3;
    static constexpr int XSLOT = PL * CW_NQ * CI * 16;    // bytes of one x row (all planes)
    static constexpr int YBUF = PL * (CW_NQ + 2) * CO * 16;  // one gy row with its two halo units
    // LDS: [4 slots][planes][NQ][CI] units of x (row r in slot (r + 4) & 3), [2][planes][NQ + 2][CO] units of gy (row r in
    // buffer r & 1), 16 zero bytes, the exponents of the four x slots and the two gy buffers (F16)
    static constexpr int XS = 0, YS = 4 * XSLOT, ZS = YS + 2 * YBUF, ES = ZS + 16, BYTES = ES + 32;
};

extern __shared__ __attribute__((aligned(16))) char cw_smem[];

// The two roles are separate (non-inlined) functions so that each gets a register allocation of its own: inlined into one
// kernel body the loader's prefetched rows were parked in accumulator registers behind a wait on every single load (one
// memory latency per load: 2.4 ms for the 128 -> 48 launch).

// ---- the loader wave: the prologue rows, then during the MFMAs of row y it writes x row y + 2 and gy row y + 1; what it writes
// in one iteration was requested from memory an iteration earlier
// ROLE 0: the x rows (prologue rows y0 - 1 .. y0 + 1, then row y + 2 during the MFMAs of row y); ROLE 1: the gy rows (row y0,
// then row y + 1).  `base`: pixel (0, 0) of the block's image with the channel block offset applied.
template <int CIT, int COT, int F16, int ROLE>
__device__ __noinline__ void cw_loader(const float* base, int64_t ld, int H, int W, int y0, int y1, int lt, int x0, int S) {
    using G = CwGeom<CIT, COT, F16>;
    char* buf = cw_smem + (ROLE == 0 ? G::XS : G::YS);
    int* exps = reinterpret_cast<int*>(cw_smem + G::ES) + (ROLE == 0 ? 0 : 4);   // exponent each slot / buffer was split with
    auto rowp = [&](int y) __attribute__((always_inline)) -> const float* { return (y >= 0 && y < H) ? base + (int64_t)y * W * ld : nullptr; };
    CwStage<(ROLE == 0 ? G::CI : G::CO), (ROLE == 0 ? CW_NQ : CW_NQ + 2), (ROLE == 0 ? 0 : -1), CW_LOADERS, G::PL> st;
    st.init(ld, W, lt, x0, S);
    int e = CWH_E0;                                        // F16: the block's running exponent of this operand
    // split + write the row in flight: the exponent first drops if this row's amax asks for it
    // (always_inline: left to its heuristics hipcc OUTLINED this lambda in the three-plane instances -- the staged row then
    // lived in scratch memory between load() and store(): 4.4 ms instead of 1.5 for the 128 -> 128 launch)
    auto put = [&](int slot) __attribute__((always_inline)) {
        float scale = 1.f;
        if (F16) {
            float m = st.amax();
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const int mx = (int)(__float_as_uint(m) >> 23);
            if (mx + e - 127 >= CWH_LIMIT) e = CWH_TARGET + 127 - mx;
            scale = cw_pow2(e);
            if (lt == 0) exps[slot] = e;
        }
#if GT_CW_ABL != 1            // ablation build 1 (timing only): the loaders fetch but neither split nor write
        st.store(buf + slot * (ROLE == 0 ? G::XSLOT : G::YBUF), scale);
#else
        if (scale == 12345.f) st.store(buf + slot * (ROLE == 0 ? G::XSLOT : G::YBUF), scale);
#endif
    };
    if (ROLE == 0) {
        for (int r = -1; r <= 1; ++r) {
            st.load(rowp(y0 + r));
            put((y0 + r + 4) & 3);
        }
        if (y0 + 1 < y1) st.load(rowp(y0 + 2));           // what the first iteration writes
    } else {
        st.load(rowp(y0));
        put(y0 & 1);
        if (y0 + 1 < y1) st.load(rowp(y0 + 1));
    }
    __syncthreads();
    for (int y = y0; y < y1; ++y) {
        if (y + 1 < y1) {
            if (ROLE == 0) {
                put((y + 2 + 4) & 3);
                if (y + 2 < y1) st.load(rowp(y + 3));
            } else {
                put((y + 1) & 1);
                if (y + 2 < y1) st.load(rowp(y + 2));
            }
        }
        __syncthreads();
    }
}

// ---- an MFMA wave: the three taps (dy, dx = -1, 0, 1) for all CI x CO of the block
template <int CIT, int COT, int F16>
__device__ __noinline__ void cw_mfma(float* slab, int Cin, int Cout, int ci0, int co0, int y0, int y1, int dy, int lane) {
    using G = CwGeom<CIT, COT, F16>;
    constexpr int CI = G::CI, CO = G::CO, PL = G::PL;
    using frag_t = std::conditional_t<F16 != 0, cw_f16x8, cw_bf16x8>;
    const char* xs = cw_smem + G::XS;
    const char* ys = cw_smem + G::YS;
    const int* xexp = reinterpret_cast<const int*>(cw_smem + G::ES);
    const int* gexp = xexp + 4;
    const int li = lane & 15, kq = lane >> 4;
    f32x4 acc[3][CIT][COT];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[d][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int eacc = 2 * CWH_E0;                                 // F16: the exponent sum this wave's accumulators carry

    // One k-step = four pixel groups (the lane groups kq) of row y.  Ten groups per row: the third step has two; its lanes
    // kq >= 2 re-read group 9 and get zero x fragments instead (the ds_read addresses then are base + immediate throughout).
    // The gy fragments of tap column d + 1 are requested before the MFMAs of column d.
    auto kstep = [&](const char* xr, const char* yr, int q, auto last) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last)::value;
        const int qq = LAST ? min(q, CW_NQ - 1) : q;
        const char* abase = xr + ((qq * CI + li) << 4);
        const char* bbase = yr + (((qq + 2) * CO + li) << 4);
        frag_t a[CIT][PL], bq[2][COT][PL];
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
                a[i][pl] = *reinterpret_cast<const frag_t*>(abase + pl * (CW_NQ * CI * 16) + i * 256);
        auto loadb = [&](int d, frag_t (&dst)[COT][PL]) __attribute__((always_inline)) {     // dx = d - 1: gy unit q - dx, stored at unit index q + 2 - d
#pragma unroll
            for (int j = 0; j < COT; ++j)
#pragma unroll
                for (int pl = 0; pl < PL; ++pl)
                    dst[j][pl] = *reinterpret_cast<const frag_t*>(bbase + pl * ((CW_NQ + 2) * CO * 16) - d * (CO * 16) + j * 256);
        };
        loadb(0, bq[0]);
        if (LAST && q >= CW_NQ) {
#pragma unroll
            for (int i = 0; i < CIT; ++i)
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) a[i][pl] = __builtin_bit_cast(frag_t, cw_u32x4{0u, 0u, 0u, 0u});
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d < 2) loadb(d + 1, bq[(d + 1) & 1]);
#pragma unroll
            for (int s = PL - 1; s >= 0; --s)              // plane pairs, smallest terms first (gt_gemm_x3.hip)
#pragma unroll
                for (int pa = 0; pa < PL; ++pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb >= PL) continue;
#pragma unroll
                    for (int i = 0; i < CIT; ++i)
#pragma unroll
                        for (int j = 0; j < COT; ++j) {
                            if constexpr (F16)
                                acc[d][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][pa], bq[d & 1][j][pb], acc[d][i][j], 0, 0, 0);
                            else
                                acc[d][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][pa], bq[d & 1][j][pb], acc[d][i][j], 0, 0, 0);
                        }
                }
        }
    };
    __syncthreads();                                       // the loader's prologue rows are in place
    for (int y = y0; y < y1; ++y) {
        const int xslot = (y + dy + 4) & 3;
        const char* xr = xs + xslot * G::XSLOT;
        const char* yr = ys + (y & 1) * G::YBUF;
        if (F16) {                                         // the rows' exponents: rescale what has been accumulated when their sum dropped
            const int es = __builtin_amdgcn_readfirstlane(xexp[xslot] + gexp[y & 1]);
            if (es != eacc) {
                const int dd = es - eacc;
                const float f = dd < -126 ? 0.f : cw_pow2(dd);
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int i = 0; i < CIT; ++i)
#pragma unroll
                        for (int j = 0; j < COT; ++j) acc[d][i][j] *= f;
                eacc = es;
            }
        }
#if GT_CW_ABL != 2            // ablation build 2 (timing only): the MFMA waves only keep the barriers
        kstep(xr, yr, kq, std::false_type{});
        kstep(xr, yr, 4 + kq, std::false_type{});
        kstep(xr, yr, 8 + kq, std::true_type{});
#endif
        __syncthreads();                                   // row y is done with; the loader has published rows y + 2 / y + 1
    }

    // partial result: accumulator register r of lane (li, kq) = (x position 16 i + 4 kq + r, gy position 16 j + li)
    const int et = -eacc, etc = et < -126 ? -126 : (et > 126 ? 126 : et);
    const float us = F16 ? cw_pow2(etc) : 1.f;
    const int erest = F16 ? et - etc : 0;                  // non-zero only for operands ~2^-100 below unit scale
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
            for (int j = 0; j < COT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pa = 16 * i + 4 * kq + r, pb = 16 * j + li;
                    const int ci = ci0 + cw_chan_of_pos(pa, CI), co = co0 + cw_chan_of_pos(pb, CO);
                    const float sg = ((r + li) & 1) ? -us : us;        // position parities (16 i + 4 kq and 16 j are even)
                    if (ci < Cin && co < Cout)
                        ((cw_gf32*)slab)[((int64_t)((dy + 1) * 3 + d) * Cin + ci) * Cout + co] =
                            erest ? ldexpf(sg * acc[d][i][j][r], erest) : sg * acc[d][i][j][r];
                }
}

template <int CIT, int COT, int F16>
__global__ __launch_bounds__(CW_THREADS, 1) void convw_kernel(const ConvWP p) {
    using G = CwGeom<CIT, COT, F16>;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = blockIdx.y * G::CI, co0 = blockIdx.z * G::CO;
    const int bcs = blockIdx.x, bc = bcs / p.nseg, seg = bcs - bc * p.nseg, b = bc / p.chunks, chunk = bc - b * p.chunks;
    const int y0 = chunk * p.rows_per_chunk, y1 = min(p.H, y0 + p.rows_per_chunk);
    const int x0 = seg * p.seg_w, S = min(p.seg_w, p.W - x0);
    for (int i = tid; i < G::ES / 16; i += CW_THREADS)
        reinterpret_cast<cw_u32x4*>(cw_smem)[i] = cw_u32x4{0u, 0u, 0u, 0u};
    if (tid < 8) reinterpret_cast<int*>(cw_smem + G::ES)[tid] = CWH_E0;
    __syncthreads();
    if (wave == 0)
        cw_loader<CIT, COT, F16, 0>(p.x + (int64_t)b * p.H * p.W * p.ldx + ci0, p.ldx, p.H, p.W, y0, y1, tid & 63, x0, S);
    else if (wave == 4)
        cw_loader<CIT, COT, F16, 1>(p.gy + (int64_t)b * p.H * p.W * p.ldg + co0, p.ldg, p.H, p.W, y0, y1, tid & 63, x0, S);
    else
        cw_mfma<CIT, COT, F16>(p.slabs + (int64_t)bcs * 9 * p.Cin * p.Cout, p.Cin, p.Cout, ci0, co0, y0, y1, wave - 2, tid & 63);
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

struct CwPlan { int cit, cot, f16, ciblocks, coblocks, chunks, rows, nseg, seg_w; size_t lds; };

// Channel blocks of one thread block: outputs of 48 channels (the down-scaler's narrow convolutions, padded) in one piece,
// wide outputs (the up-scaler's 128 -> 128 convolution) in blocks of 64; inputs in the widest block that divides Cin and fits
// the LDS (two fp16 planes: up to 64 channels; three bf16 planes: 48 / 32).  One block per CU (ring of four x rows + two gy rows).
static bool cw_plan(int B, int H, int W, int Cin, int Cout, int f16, CwPlan* pl) {
    if (B <= 0 || H <= 0 || W <= 0 || W > 64 * CW_MAXW || Cin <= 0 || Cout <= 0 || (Cin & 15)) return false;
    pl->f16 = f16;
    pl->nseg = (W + CW_MAXW - 1) / CW_MAXW;                 // rows wider than 80 pixels: equal segments (C3: 113 / 114 -> 2 x 57)
    pl->seg_w = (W + pl->nseg - 1) / pl->nseg;
    if (Cout == 48) {
        pl->cot = 3;
        pl->cit = (f16 && Cin % 64 == 0) ? 4 : (Cin % 48 == 0) ? 3 : (Cin % 32 == 0) ? 2 : 1;
    } else if (Cout % 64 == 0 && Cin % 32 == 0) {
        pl->cot = 4;
        pl->cit = 2;                                        // 64 x 64 blocks need more accumulators than two waves per SIMD leave
        // fp16 planes: 64 input x 32 output channels balance the two loader waves (5 / 3 items a row against 2.5 / 6 for
        // 32 x 64): 1061 vs 1214 us for the up-scaler's 128 -> 128
        if (f16 && Cin % 64 == 0) { pl->cit = 4; pl->cot = 2; }
    } else {
        return false;
    }
    if (const char* e = getenv("GT_CW_CIT")) {              // tuning override (tools): input-channel tiles per block
        const int c = atoi(e);
        if (c >= 1 && c <= 4 && Cin % (16 * c) == 0 && (c < 4 || f16) && !(pl->cot == 4 && c != 2 && c != 4)) pl->cit = c;
    }
    pl->ciblocks = Cin / (16 * pl->cit);
    pl->coblocks = Cout / (16 * pl->cot);
    const int per_row = B * pl->ciblocks * pl->coblocks * pl->nseg;
    int chunks = std::max(1, (256 + per_row - 1) / per_row);
    chunks = std::min(chunks, std::max(1, H / 8));          // at least eight rows per block: the three-row prologue is paid once
    pl->rows = (H + chunks - 1) / chunks;
    pl->chunks = (H + pl->rows - 1) / pl->rows;
    const size_t planes = f16 ? 2 : 3;
    pl->lds = 4 * planes * CW_NQ * 16 * pl->cit * 16 + 2 * planes * (CW_NQ + 2) * 16 * pl->cot * 16 + 16 + 32;
    return true;
}

}  // namespace gt

using namespace gt;

extern "C" int64_t gt_conv3x3_wgrad_nhwc_ws_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    // the row chunks depend on the channel blocks and those on the arithmetic (Cout = 48, Cin % 64 == 0: four input tiles in
    // fp16, three or two in bf16): the size that serves BOTH plans (the fp16 launch wrote up to ~2x the slabs the
    // bf16 plan had advertised)
    CwPlan p0, p1;
    const bool ok0 = cw_plan(B, H, W, Cin, Cout, 0, &p0), ok1 = cw_plan(B, H, W, Cin, Cout, 1, &p1);
    if (!ok0 && !ok1) return 0;
    const int chunks = std::max(ok0 ? p0.chunks : 0, ok1 ? p1.chunks : 0), nseg = ok0 ? p0.nseg : p1.nseg;
    return (int64_t)B * chunks * nseg * 9 * Cin * Cout * (int64_t)sizeof(float);
}

extern "C" int gt_conv3x3_wgrad_nhwc(const float* gy, int64_t ldg, const float* x, int64_t ldx, float* dw, int32_t B,
                                     int32_t H, int32_t W, int32_t Cin, int32_t Cout, float alpha, int32_t precision,
                                     void* ws, int64_t ws_bytes, void* stream) {
    if (!gy || !x || !dw || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldg < Cout || ldx < Cin) return GT_EINVAL;
    if (precision != GT_PREC_BF16X3 && precision != GT_PREC_F16X2) return GT_ENOTSUP;
    CwPlan pl;
    if (!cw_plan(B, H, W, Cin, Cout, precision == GT_PREC_F16X2, &pl)) return GT_ENOTSUP;
    if (((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(x)) & 15) || (ldg & 3) || (ldx & 3)) return GT_EALIGN;
    const int64_t nslab = (int64_t)B * pl.chunks * pl.nseg;
    if (!ws || ws_bytes < nslab * 9 * Cin * Cout * (int64_t)sizeof(float)) return GT_EWS;   // the plan launched
    if (nslab > 0x7fffffffLL) return GT_EINVAL;
    ConvWP p{gy, ldg, x, ldx, reinterpret_cast<float*>(ws), B, H, W, Cin, Cout, pl.chunks, pl.rows, pl.nseg, pl.seg_w};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)nslab, (unsigned)pl.ciblocks, (unsigned)pl.coblocks);
    // more than 64 KB of LDS per block: the limit is raised once per kernel instance
    // (the attribute is per device: one bit per device ordinal, set once; concurrent host threads at worst set it twice)
    static std::atomic<uint64_t> raised[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t dbit = 1ull << (dev & 63);
    auto launch = [&](auto kern, int idx) -> int {
        if (!(raised[idx].load(std::memory_order_acquire) & dbit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)pl.lds) != hipSuccess)
                return GT_ENOTSUP;
            raised[idx].fetch_or(dbit, std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, grid, dim3(CW_THREADS), pl.lds, st, p);
        return 0;
    };
    int rc = GT_ENOTSUP;
    const int key = pl.f16 * 100 + pl.cit * 10 + pl.cot;
    switch (key) {
        case 13: rc = launch(convw_kernel<1, 3, 0>, 0); break;
        case 23: rc = launch(convw_kernel<2, 3, 0>, 1); break;
        case 33: rc = launch(convw_kernel<3, 3, 0>, 2); break;
        case 24: rc = launch(convw_kernel<2, 4, 0>, 3); break;
        case 113: rc = launch(convw_kernel<1, 3, 1>, 4); break;
        case 123: rc = launch(convw_kernel<2, 3, 1>, 5); break;
        case 133: rc = launch(convw_kernel<3, 3, 1>, 6); break;
        case 143: rc = launch(convw_kernel<4, 3, 1>, 7); break;
        case 124: rc = launch(convw_kernel<2, 4, 1>, 8); break;
        case 144: rc = launch(convw_kernel<4, 4, 1>, 9); break;
        case 142: rc = launch(convw_kernel<4, 2, 1>, 10); break;
        default: break;
    }
    if (rc) return rc;
    GT_LAUNCH_CHECK();
    const int n = 9 * Cin * Cout;
    hipLaunchKernelGGL(convw_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.slabs, (int)nslab, Cin, Cout, alpha,
                       dw);
    GT_LAUNCH_CHECK();
    return 0;
}
