// Fourier-type attention  out = ((Q' K'^T) * scale .* mask) V'  (layers.py:672-705) on the 16-bit matrix pipe with
// fp32-class results: the softmax-free flash kernel of gt_fourier.hip (no n x n matrix in HBM, one template for the
// forward, d/dQ' and the dual d/dK' + d/dV' pass) in the two-term fp16 arithmetic of DESIGN 4.1 -- every operand value
// x is h0 + h1 with h0 = f16(x s), h1 = f16(x s - h0), s a power of two, and a product is the three MFMAs h1 g0 + h0 g1 +
// h0 g0 on v_mfma_f32_16x16x32_f16 (16 x the rate of the fp32 MFMA the round-1 kernel runs on).
//
// Who splits what, once:
//   * the head tiles (Q', K', V', dO: [B, n, h, DP] fp32) are split by fourier16_presplit_kernel into IMAGES, one per
//     (batch, head, tile of 32 token rows): the two fp16 planes in exactly the order the MFMA lanes read them, in two
//     layouts -- "rm" (a lane's 8 consecutive COLUMNS of one row: operand of the first product, S^T = T1 F1^T) and "tr"
//     (a lane's 8 ROWS of one column, rows in the order the first product's result registers enumerate them: the A operand
//     T2^T of the second product, O^T += T2^T S) -- with one power-of-two exponent per tile and the tile's largest row
//     norm.  A stream tile then goes global -> LDS by direct loads (global_load_lds_dwordx4, 1 KiB per wave instruction,
//     the image IS the LDS image) and every fragment read is one contiguous, conflict-free ds_read_b128;
//   * the score tile is masked, scaled and split in registers between the two products: the MFMA D layout of the first
//     product (lane (j, kq): stream rows 16 mt + 4 kq + r of owner column j) is the B layout of the second one when its
//     contraction index enumerates the stream rows as 8 kq + 4 mt + r -- no cross-lane traffic (the trick of gt_fourier.hip).
//
// Scales.  Planes of tile t of tensor X hold X 2^ex[t] (amax in [2^13, 2^14)); lnrm[t] >= log2 of the tile's largest
// scaled row norm.  With D = the true dot product the first product returns 2^(e1[t] + ef) D and |.| <= 2^(l1[t] + lf)
// (Cauchy-Schwarz): the score is multiplied by sigma_t = 2^(E - e1[t] - ef - e2[t]) with a per-wave RUNNING exponent
// E = min over the tiles so far of (15 - l1 - lf + e1 + ef + e2), so that the scaled score is below 2^15 (nothing can
// overflow; no amax of the score tile is taken) and the second product accumulates 2^E (D .* mask) T2 whatever the tile's
// own exponents; the accumulators are multiplied by 2^(E' - E) when E drops (the online rescaling of DESIGN 4.1).
// The second product's chains (K = n) run sign-alternated -- (-1)^(dim + owner), DESIGN 2: the 16-bit MFMA chops its
// addends toward -inf -- the dim sign is baked into the "tr" image, the owner sign rides on sigma.
#include "gt_common.h"
#include <algorithm>
#include <cstdlib>

namespace gt {

typedef _Float16 ff16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ff16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t fu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* f16_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* f16_glb_ptr_t;

template <int DP>
struct F16G {
    static constexpr int NM = DP / 32;                 // full 32-column k-steps of the first product
    static constexpr int TC = DP - 32 * NM;            // columns of the last (partial) k-step
    static constexpr int TG = (TC + 7) / 8;            // its 16-byte granules per row (the other lane groups read zeros)
    static constexpr int NS = NM + 1;
    static constexpr int ND = (DP + 15) / 16;          // 16-row tiles of O^T
    static constexpr int MAIN_G = 128 * NM;            // granules per plane: 32 rows x 4 NM
    static constexpr int TAIL_G = 32 * TG;
    static constexpr int RM_G = MAIN_G + TAIL_G, TR_G = 64 * ND;
    static constexpr int RM_BYTES = 2 * RM_G * 16, TR_BYTES = 2 * TR_G * 16, IMG = RM_BYTES + TR_BYTES;
    // rm image: [plane 0 main][plane 1 main][plane 0 tail][plane 1 tail];  tr image: [plane 0][plane 1]
    static_assert(DP % 4 == 0 && TC > 0 && RM_BYTES % 1024 == 0 && TR_BYTES % 1024 == 0, "image chunks are 1 KiB");
};

__device__ __forceinline__ float f16_pow2(int e) {                        // 2^e, e clamped to the normal range
    e = e < -126 ? -126 : (e > 127 ? 127 : e);
    return __uint_as_float((uint32_t)(e + 127) << 23);
}
__device__ __forceinline__ float f16_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// two fp32 (already scaled) -> packed fp16 pairs h0, h1
__device__ __forceinline__ void f16_split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2 r = f32x2{a, b};
    const ff16x2 h0 = __builtin_convertvector(r, ff16x2);                  // v_cvt_pk_f16_f32 (RNE)
    const ff16x2 h1 = __builtin_convertvector(r - __builtin_convertvector(h0, f32x2), ff16x2);
    hi = __builtin_bit_cast(uint32_t, h0);
    lo = __builtin_bit_cast(uint32_t, h1);
}
// acc += a b, a = ah + al, b = bh + bl: small terms first
__device__ __forceinline__ f32x4 f16_mma3(ff16x8 ah, ff16x8 al, ff16x8 bh, ff16x8 bl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------ pre-split
struct F16PreP {
    const float* X[4];
    uint8_t* I[4];
    int n, h, ntile;
    int64_t hdr;             // bytes of the exponent header in front of the images
};

// one wave per (tile of 32 rows, batch x head, tensor)
template <int DP>
__global__ __launch_bounds__(64) void fourier16_presplit_kernel(const F16PreP p) {
    using G = F16G<DP>;
    constexpr int KS = DP / 4, NM = G::NM, TG = G::TG;
    __shared__ __attribute__((aligned(16))) float s[32 * DP];
    const int lane = threadIdx.x, tile = blockIdx.x, bh = blockIdx.y, tz = blockIdx.z;
    const float* X = p.X[tz];
    uint8_t* I = p.I[tz];
    const int b = bh / p.h, head = bh % p.h;
    const int64_t hD = (int64_t)p.h * DP;
    const int64_t base = ((int64_t)b * p.n) * hD + (int64_t)head * DP;
    float amax = 0.f;
    for (int e = lane; e < 32 * KS; e += 64) {
        const int r = e / KS, c = e % KS, row = 32 * tile + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < p.n) v = *reinterpret_cast<const f32x4*>(X + base + (int64_t)row * hD + 4 * c);
        *reinterpret_cast<f32x4*>(&s[r * DP + 4 * c]) = v;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    amax = f16_wave_max(amax);
    __syncthreads();
    const int be = (int)(__float_as_uint(amax) >> 23);
    int ex = be == 0 ? 0 : 140 - be;                   // amax 2^ex in [2^13, 2^14)
    ex = ex > 126 ? 126 : ex;
    const float sc = f16_pow2(ex);
    float nr = 0.f;
    if (lane < 32)
        for (int c = 0; c < DP; ++c) {
            const float v = s[lane * DP + c] * sc;
            nr += v * v;
        }
    nr = f16_wave_max(nr);
    const float nrm = sqrtf(nr) * 1.0001f;
    int ln = (int)(__float_as_uint(nrm) >> 23) - 126;  // nrm < 2^ln
    ln = ln < -40 ? -40 : ln;
    if (lane == 0) {
        int32_t* hx = reinterpret_cast<int32_t*>(I) + 2 * ((int64_t)bh * p.ntile + tile);
        hx[0] = ex;
        hx[1] = ln;
    }
    fu32x4* img = reinterpret_cast<fu32x4*>(I + p.hdr + ((int64_t)bh * p.ntile + tile) * G::IMG);
    for (int g = lane; g < G::RM_G + G::TR_G; g += 64) {
        float v[8];
        int d0, d1;
        if (g < G::MAIN_G) {
            if constexpr (NM > 0) {
                const int row = g / (4 * NM), q = g % (4 * NM);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = s[row * DP + 8 * q + e];
            }
            d0 = g;
            d1 = G::MAIN_G + g;
        } else if (g < G::RM_G) {
            const int gt_ = g - G::MAIN_G, row = gt_ / TG, q = gt_ % TG, c0 = 32 * NM + 8 * q;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c0 + e < DP) ? s[row * DP + c0 + e] : 0.f;
            d0 = 2 * G::MAIN_G + gt_;
            d1 = 2 * G::MAIN_G + G::TAIL_G + gt_;
        } else {
            const int gr = g - G::RM_G, kq = gr & 3, m = (gr >> 2) & 15, dt = gr >> 6, col = 16 * dt + m;
            const float sg = (col & 1) ? -1.f : 1.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = 16 * (e >> 2) + 4 * kq + (e & 3);
                v[e] = (col < DP) ? sg * s[row * DP + col] : 0.f;
            }
            d0 = G::RM_BYTES / 16 + gr;
            d1 = G::RM_BYTES / 16 + G::TR_G + gr;
        }
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) f16_split_pair(v[2 * e] * sc, v[2 * e + 1] * sc, hi[e], lo[e]);
        img[d0] = fu32x4{hi[0], hi[1], hi[2], hi[3]};
        img[d1] = fu32x4{lo[0], lo[1], lo[2], lo[3]};
    }
}

// ------------------------------------------------------------------------------------------------------ core
struct F16P {
    const uint8_t* F1; const uint8_t* F2; const uint8_t* T1; const uint8_t* T2;      // image blocks (header + images)
    float* O1; float* O2;
    const float* mask;           // explicit multiplicative mask [B,h,n,n] (query-major) or null
    DropDev drop;
    int n, h, ntile, nblk, total;
    int64_t hdr;
    float scale;
    int owner_is_key;
};
// DROPH: per-element hash, threshold without low 16 bits.  DROPB: p = 0.5 drawn per 4 x 4 block of the score matrix -- ONE hash
// per block (query >> 2, key >> 2), element (q, k) keeps iff bit 16 + 4 (q & 3) + (k & 3) of it is set (gt_hip.h:
// gt_dropout_block16): a lane's four result registers of a 16 x 16 score tile are four keys (or four queries) of one block,
// so the hash is evaluated once per register QUAD in both orientations -- the score phase is bound by instruction issue
// (profiles/r06c_fourier16_sq_counters.txt) and the per-element hash was 128 of its 200 instructions per tile.
enum { F16_PLAIN = 0, F16_DROP = 1, F16_DROPH = 2, F16_MASK = 3, F16_DROPB = 4 };

// NW = waves per block (4 or 8): a block owns 32 NW owner rows and all of them share one stream tile in LDS.  The stream images
// go L2 -> LDS by direct loads and that path saturates near 6.4 TB/s for the chip (MI355X_MICROARCH.md: ldsdma-fill); with
// 128 owners per tile the C3 passes asked for 4.4 - 5.3 TB/s of it (1.24 GB per launch) and did not get faster without the
// dropout hash, with 256 owners for half of that.
template <int DP, bool DUAL, int MODE, int NW>
__global__ __launch_bounds__(64 * NW, DUAL ? 2 : (DP > 36 ? 3 : 4)) void fourier16_kernel(const F16P p) {
    using G = F16G<DP>;
    constexpr int NM = G::NM, TG = G::TG, NS = G::NS, ND = G::ND;
    constexpr int STAGE = DUAL ? 2 * G::IMG : G::IMG;            // one image of each stream tensor / T1's rm + T2's tr
    constexpr int N1 = (DUAL ? G::IMG : G::RM_BYTES) / 1024, NCH = STAGE / 1024;
    constexpr int OFF_T1RM = 0, OFF_T1TR = G::RM_BYTES;
    constexpr int OFF_T2RM = G::IMG, OFF_T2TR = DUAL ? G::IMG + G::RM_BYTES : G::RM_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t smem[2][STAGE];
    __shared__ __attribute__((aligned(16))) uint32_t zero_g[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    // workgroups go round-robin to the 8 XCDs: give each XCD a contiguous range of (batch, head, owner block), so the
    // ~30 blocks that stream the same images share one L2
    const int wg = blockIdx.x, per = p.total >> 3, rem = p.total & 7, xcd = wg & 7;
    const int logical = xcd * per + min(xcd, rem) + (wg >> 3);
    const int bh = logical / p.nblk, xblk = logical - bh * p.nblk;
    const int b = bh / p.h, head = bh - b * p.h;
    const int otile = xblk * NW + wave;                          // this wave's 32 owner rows
    const bool olive = otile < p.ntile;
    const int ot = olive ? otile : p.ntile - 1;
    const int o0 = otile * 32;
    const int64_t hD = (int64_t)p.h * DP;
    const int64_t base = ((int64_t)b * p.n) * hD + (int64_t)head * DP;
    const uint32_t zn = ((uint32_t)b * (uint32_t)p.h + (uint32_t)head) * (uint32_t)p.n;
    const int64_t tile0 = (int64_t)bh * p.ntile;
    if (tid < 4) zero_g[tid] = 0u;

    auto issue = [&](int t, int buf) {
        const uint8_t* s1 = p.T1 + p.hdr + (tile0 + t) * G::IMG;
        const uint8_t* s2 = p.T2 + p.hdr + (tile0 + t) * G::IMG + (DUAL ? 0 : G::RM_BYTES);
#pragma unroll
        for (int i = 0; i < (NCH + NW - 1) / NW; ++i) {
            const int q = wave + NW * i;
            if (q < NCH) {
                const uint8_t* src = (q < N1 ? s1 + q * 1024 : s2 + (q - N1) * 1024) + lane * 16;
                __builtin_amdgcn_global_load_lds((f16_glb_ptr_t)src, (f16_lds_ptr_t)(&smem[buf][q * 1024]), 16, 0, 0);
            }
        }
    };
    issue(0, 0);

    // owner fragments (B operand of the first product): lane (j, kq) holds columns 8 kq .. 8 kq + 7 of k-step s of owner
    // row 16 nt + j, straight from the owner tensor's rm image
    ff16x8 f1h[2][NS], f1l[2][NS], f2h[DUAL ? 2 : 1][DUAL ? NS : 1], f2l[DUAL ? 2 : 1][DUAL ? NS : 1];
    {
        const fu32x4* i1 = reinterpret_cast<const fu32x4*>(p.F1 + p.hdr + (tile0 + ot) * G::IMG);
        const fu32x4* i2 = DUAL ? reinterpret_cast<const fu32x4*>(p.F2 + p.hdr + (tile0 + ot) * G::IMG) : i1;
        const fu32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bool ok = olive && (s < NM || kq < TG);
                const int g0 = s < NM ? (16 * nt + j) * 4 * NM + 4 * s + kq : 2 * G::MAIN_G + (16 * nt + j) * TG + kq;
                const int g1 = g0 + (s < NM ? G::MAIN_G : G::TAIL_G);
                f1h[nt][s] = __builtin_bit_cast(ff16x8, ok ? i1[g0] : z);
                f1l[nt][s] = __builtin_bit_cast(ff16x8, ok ? i1[g1] : z);
                if (DUAL) {
                    f2h[nt][s] = __builtin_bit_cast(ff16x8, ok ? i2[g0] : z);
                    f2l[nt][s] = __builtin_bit_cast(ff16x8, ok ? i2[g1] : z);
                }
            }
    }
    const int32_t* x1 = reinterpret_cast<const int32_t*>(p.T1) + 2 * tile0;
    const int32_t* x2 = reinterpret_cast<const int32_t*>(p.T2) + 2 * tile0;
    const int32_t* xf1 = reinterpret_cast<const int32_t*>(p.F1) + 2 * (tile0 + ot);
    const int32_t* xf2 = DUAL ? reinterpret_cast<const int32_t*>(p.F2) + 2 * (tile0 + ot) : xf1;
    const int ef1 = __builtin_amdgcn_readfirstlane(xf1[0]), lf1 = __builtin_amdgcn_readfirstlane(xf1[1]);
    const int ef2 = __builtin_amdgcn_readfirstlane(xf2[0]), lf2 = __builtin_amdgcn_readfirstlane(xf2[1]);

    // dropout hash carriers: hw[nt] = idx*G + key of (first stream row of this lane in the tile, owner nt)
    constexpr uint32_t GOLD = 0x9e3779b1u;
    uint32_t hw[2] = {0u, 0u}, hstep = 0u;
    if (MODE == F16_DROP || MODE == F16_DROPH) {
        const uint32_t key = drop_key_dev(p.drop);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const uint32_t ow = (uint32_t)(o0 + 16 * nt + j), st = 4u * (uint32_t)kq;
            const uint32_t idx = p.owner_is_key ? (zn + st) * (uint32_t)p.n + ow : (zn + ow) * (uint32_t)p.n + st;
            hw[nt] = idx * GOLD + key;
        }
        hstep = p.owner_is_key ? (uint32_t)p.n * GOLD : GOLD;      // idx step per stream row, times G
    }
    uint32_t bpos[4] = {0u, 0u, 0u, 0u};
    if (MODE == F16_DROPB) {
        // block index ((b h + head) nq4 + (query >> 2)) nq4 + (key >> 2); the lane's stream rows 16 mt + 4 kq + r of tile t
        // are rows 4 (8 t + 4 mt + kq) + r: one block per (nt, mt)
        const uint32_t key = drop_key_dev(p.drop), nq4 = (uint32_t)((p.n + 3) >> 2);
        const uint32_t zn4 = ((uint32_t)b * (uint32_t)p.h + (uint32_t)head) * nq4;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const uint32_t own4 = (uint32_t)(o0 + 16 * nt + j) >> 2;
            const uint32_t blk = p.owner_is_key ? (zn4 + (uint32_t)kq) * nq4 + own4 : (zn4 + own4) * nq4 + (uint32_t)kq;
            hw[nt] = blk * GOLD + key;
        }
        hstep = p.owner_is_key ? nq4 * GOLD : GOLD;               // step per block along the stream axis, times G
#pragma unroll
        for (int r = 0; r < 4; ++r) bpos[r] = 16u + (p.owner_is_key ? 4u * r + (uint32_t)(j & 3) : 4u * (uint32_t)(j & 3) + r);
    }
    const float osign = (j & 1) ? -1.f : 1.f;                     // owner half of the chain sign (-1)^(dim + owner)

    f32x4 acc1[ND][2], acc2[DUAL ? ND : 1][2];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            acc1[dt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (DUAL) acc2[dt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    int EA = 1 << 20, EB = 1 << 20;                               // running exponents of the two accumulator sets

    // per-lane fragment offsets inside a stage (bytes)
    const int offm = (j * 4 * NM + kq) * 16;                       // main k-step 0, row tile 0, plane 0
    const bool tail_ok = kq < TG;
    const int offt = (2 * G::MAIN_G + j * TG + kq) * 16;           // tail, row tile 0, plane 0
    const int offr = (j * 4 + kq) * 16;                            // tr image, dim tile 0, plane 0

    // the exponents of tile t + 1 are requested in front of its image (vector-memory loads return in order: a load
    // behind the image requests would make its first use wait for the whole image)
    int ne1 = x1[0], nl1 = x1[1], ne2 = x2[0], nl2 = x2[1];
    for (int t = 0; t < p.ntile; ++t) {
        // tile t has landed for this wave (vmcnt) and for everybody (barrier); everybody is also done with tile t-1,
        // whose buffer the next request overwrites
        // (lgkmcnt too: the zero_g LDS store in front of the loop and this wave's LDS reads of tile t - 1 are formally
        // ordered before the other waves' accesses only once the LDS counter has drained -- ADVICE r5)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int e1 = __builtin_amdgcn_readfirstlane(ne1), l1 = __builtin_amdgcn_readfirstlane(nl1);
        const int e2 = __builtin_amdgcn_readfirstlane(ne2), l2 = __builtin_amdgcn_readfirstlane(nl2);
        if (t + 1 < p.ntile) {
            ne1 = x1[2 * t + 2]; nl1 = x1[2 * t + 3]; ne2 = x2[2 * t + 2]; nl2 = x2[2 * t + 3];
            issue(t + 1, (t + 1) & 1);
        }
        const uint8_t* st = smem[t & 1];
        const uint8_t* zp = reinterpret_cast<const uint8_t*>(zero_g);

        // running exponents (wave-uniform integer arithmetic) and the score multipliers
        float sigA, sigB = 0.f;
        {
            const int cap = 15 - l1 - lf1 + e1 + ef1 + e2;
            if (cap < EA) {
                if (t > 0) {
                    const int d = cap - EA;
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc1[dt][nt][r] = ldexpf(acc1[dt][nt][r], d);
                }
                EA = cap;
            }
            sigA = osign * f16_pow2(EA - e1 - ef1 - e2);
        }
        if (DUAL) {
            const int cap = 15 - l2 - lf2 + e2 + ef2 + e1;
            if (cap < EB) {
                if (t > 0) {
                    const int d = cap - EB;
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc2[dt][nt][r] = ldexpf(acc2[dt][nt][r], d);
                }
                EB = cap;
            }
            sigB = osign * f16_pow2(EB - e2 - ef2 - e1);
        }
        // keep masks of this tile's score elements (dropout modes), in front of the first product in program order: they do
        // not depend on it, and the sched_group_barrier sequence behind the product interleaves them with its MFMAs
        // (a wave's own VALU under its own matrix instructions; the score phase is issue-bound)
        uint32_t km[2][8];
        constexpr bool HASHED = MODE == F16_DROP || MODE == F16_DROPH || MODE == F16_DROPB;
        if (HASHED) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                uint32_t hk = hw[nt];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (MODE == F16_DROPB) {
                        // bits 16..31 of the finaliser do not depend on its last step x ^= x >> 16
                        uint32_t xb = hk;
                        xb ^= xb >> 16; xb *= 0x85ebca6bu; xb ^= xb >> 13; xb *= 0xc2b2ae35u;
                        hk += 4u * hstep;
#pragma unroll
                        for (int r = 0; r < 4; ++r) km[nt][4 * mt + r] = (uint32_t)__builtin_amdgcn_sbfe((int32_t)xb, bpos[r], 1u);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            uint32_t x = hk;
                            x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u;
                            if (MODE == F16_DROPH) {
                                // the finaliser's last step x ^= x >> 16 leaves the top 16 bits alone and the threshold
                                // has no bits below them: keep <=> top bit
                                km[nt][4 * mt + r] = (uint32_t)((int32_t)x >> 31);
                            } else {
                                x ^= x >> 16;
                                km[nt][4 * mt + r] = x >= p.drop.thresh ? 0xffffffffu : 0u;
                            }
                            hk += hstep;
                        }
                        hk += 12u * hstep;
                    }
                }
                hw[nt] = hk;                                      // advanced by 32 stream rows (8 blocks)
            }
        }

        // first product: score tiles (stream rows x owner columns), 2 row tiles x 2 column tiles per wave
        f32x4 sa[2][2], sb[DUAL ? 2 : 1][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                sa[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (DUAL) sb[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint8_t* a0;
                int pl;
                if (s < NM) {
                    a0 = st + offm + (mt * 16 * 4 * NM + 4 * s) * 16;
                    pl = G::MAIN_G * 16;
                } else {
                    a0 = st + offt + mt * 16 * TG * 16;
                    pl = G::TAIL_G * 16;
                }
                const uint8_t* pa = (s < NM || tail_ok) ? a0 + OFF_T1RM : zp;
                const ff16x8 ah = *reinterpret_cast<const ff16x8*>(pa);
                const ff16x8 al = *reinterpret_cast<const ff16x8*>((s < NM || tail_ok) ? pa + pl : zp);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) sa[mt][nt] = f16_mma3(ah, al, f1h[nt][s], f1l[nt][s], sa[mt][nt]);
                if (DUAL) {
                    const uint8_t* pb = (s < NM || tail_ok) ? a0 + OFF_T2RM : zp;
                    const ff16x8 bh = *reinterpret_cast<const ff16x8*>(pb);
                    const ff16x8 bl = *reinterpret_cast<const ff16x8*>((s < NM || tail_ok) ? pb + pl : zp);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) sb[mt][nt] = f16_mma3(bh, bl, f2h[nt][s], f2l[nt][s], sb[mt][nt]);
                }
            }
        }
        if (HASHED) {
            constexpr int NMF = 12 * NS * (DUAL ? 2 : 1), NVA = MODE == F16_DROPB ? 40 : 16 * 7;
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, (NVA + NMF - 1) / NMF, 0);    // its share of the mask VALU
            }
        }
        // The score phase reads the MFMA results from inline asm (f16_mulsplit_pair), and hipcc's hazard recogniser does not
        // look into inline asm: without this the first v_fma_mix* after the last MFMA read stale registers (measured: 2.9e-5
        // instead of 3e-7).  An 8-pass MFMA result needs 11 wait states before a VALU access; all score registers pass
        // through this statement, so every MFMA above has issued before it and every reader below comes after it.
        if (DUAL)
            asm volatile("s_nop 7\n\ts_nop 3" : "+v"(sa[0][0]), "+v"(sa[0][1]), "+v"(sa[1][0]), "+v"(sa[1][1]), "+v"(sb[0][0]),
                         "+v"(sb[0][1]), "+v"(sb[DUAL ? 1 : 0][0]), "+v"(sb[DUAL ? 1 : 0][1]));
        else
            asm volatile("s_nop 7\n\ts_nop 3" : "+v"(sa[0][0]), "+v"(sa[0][1]), "+v"(sa[1][0]), "+v"(sa[1][1]));
        // mask, scale, split: element (stream row 32 t + 16 mt + 4 kq + r, owner column o0 + 16 nt + j).  Rows / columns
        // beyond n hold exact zeros (zero image rows, zeroed owner fragments), whatever the mask says.
        ff16x8 ph[2], pl_[2], qh[DUAL ? 2 : 1], ql[DUAL ? 2 : 1];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float va[8], vb[8], wa[8], wb[8];                       // score values and their multipliers
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float ma = sigA, mb = sigB;
                    if (HASHED) {
                        ma = __uint_as_float(__float_as_uint(sigA) & km[nt][4 * mt + r]);
                        if (DUAL) mb = __uint_as_float(__float_as_uint(sigB) & km[nt][4 * mt + r]);
                    } else if (MODE == F16_MASK) {
                        const int sr = min(32 * t + 16 * mt + 4 * kq + r, p.n - 1), ow = min(o0 + 16 * nt + j, p.n - 1);
                        const int qi = p.owner_is_key ? sr : ow, ki = p.owner_is_key ? ow : sr;
                        const float m = p.mask[((int64_t)(b * p.h + head) * p.n + qi) * p.n + ki];
                        ma = m * sigA;
                        if (DUAL) mb = m * sigB;
                    }
                    va[4 * mt + r] = sa[mt][nt][r];
                    wa[4 * mt + r] = ma;
                    if (DUAL) {
                        vb[4 * mt + r] = sb[mt][nt][r];
                        wb[4 * mt + r] = mb;
                    }
                }
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) f16_mulsplit_pair(va[2 * e], wa[2 * e], va[2 * e + 1], wa[2 * e + 1], hi[e], lo[e]);
            ph[nt] = __builtin_bit_cast(ff16x8, fu32x4{hi[0], hi[1], hi[2], hi[3]});
            pl_[nt] = __builtin_bit_cast(ff16x8, fu32x4{lo[0], lo[1], lo[2], lo[3]});
            if (DUAL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) f16_mulsplit_pair(vb[2 * e], wb[2 * e], vb[2 * e + 1], wb[2 * e + 1], hi[e], lo[e]);
                qh[nt] = __builtin_bit_cast(ff16x8, fu32x4{hi[0], hi[1], hi[2], hi[3]});
                ql[nt] = __builtin_bit_cast(ff16x8, fu32x4{lo[0], lo[1], lo[2], lo[3]});
            }
        }
        // second product: O^T (dims x owners) += T^T (dims x stream) S (stream x owners); contraction slot 8 kq + 4 mt + r
        // = stream row 16 mt + 4 kq + r of the tile = register r of score tile mt in this lane
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            const uint8_t* a0 = st + offr + dt * 64 * 16;
            const ff16x8 ah = *reinterpret_cast<const ff16x8*>(a0 + OFF_T2TR);
            const ff16x8 al = *reinterpret_cast<const ff16x8*>(a0 + OFF_T2TR + G::TR_G * 16);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc1[dt][nt] = f16_mma3(ah, al, ph[nt], pl_[nt], acc1[dt][nt]);
            if (DUAL) {
                const ff16x8 bh = *reinterpret_cast<const ff16x8*>(a0 + OFF_T1TR);
                const ff16x8 bl = *reinterpret_cast<const ff16x8*>(a0 + OFF_T1TR + G::TR_G * 16);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc2[dt][nt] = f16_mma3(bh, bl, qh[nt], ql[nt], acc2[dt][nt]);
            }
        }
    }
    // O^T tile (dt, nt): rows = dims 16 dt + 4 kq + r, column = owner o0 + 16 nt + j  ->  O[owner][dim .. dim + 3];
    // the chain sign (-1)^(dim + owner) = (-1)^(r + j) comes off with the scales
    const float fs = p.scale * ((MODE == F16_DROP || MODE == F16_DROPH || MODE == F16_DROPB) ? p.drop.scale : 1.f);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int ow = o0 + 16 * nt + j;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            const int dim = 16 * dt + 4 * kq;
            if (olive && ow < p.n && dim < DP) {
                f32x4 v1, v2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sg = ((r + j) & 1) ? -fs : fs;
                    v1[r] = ldexpf(acc1[dt][nt][r] * sg, -EA);
                    if (DUAL) v2[r] = ldexpf(acc2[dt][nt][r] * sg, -EB);
                }
                *reinterpret_cast<f32x4*>(p.O1 + base + (int64_t)ow * hD + dim) = v1;
                if (DUAL) *reinterpret_cast<f32x4*>(p.O2 + base + (int64_t)ow * hD + dim) = v2;
            }
        }
    }
}

template <int DP, int NW>
static void fourier16_launch_nw(const F16P& p, bool dual, bool block16, hipStream_t st) {
    int mode = F16_PLAIN;
    if (p.mask) mode = F16_MASK;
    else if (p.drop.thresh) mode = block16 ? F16_DROPB : ((p.drop.thresh & 0xffffu) ? F16_DROP : F16_DROPH);
    const dim3 grid((unsigned)p.total);
#define GT_F16(D, M) hipLaunchKernelGGL((fourier16_kernel<DP, D, M, NW>), grid, dim3(64 * NW), 0, st, p)
    if (dual) {
        if (mode == F16_DROPB) GT_F16(true, F16_DROPB);
        else if (mode == F16_DROPH) GT_F16(true, F16_DROPH);
        else if (mode == F16_DROP) GT_F16(true, F16_DROP);
        else if (mode == F16_MASK) GT_F16(true, F16_MASK);
        else GT_F16(true, F16_PLAIN);
    } else {
        if (mode == F16_DROPB) GT_F16(false, F16_DROPB);
        else if (mode == F16_DROPH) GT_F16(false, F16_DROPH);
        else if (mode == F16_DROP) GT_F16(false, F16_DROP);
        else if (mode == F16_MASK) GT_F16(false, F16_MASK);
        else GT_F16(false, F16_PLAIN);
    }
#undef GT_F16
}

// waves per block: 8 when that still leaves at least ~2 blocks per CU's worth of work (halves the image traffic), else 4
// Measured at C3's layer shape (B = 8, n = 3721; tests/test_fourier16_gpu.py -k time): forward 0.247 ms with 8 waves per block
// vs 0.271 with 4; the dual pass (2 waves per SIMD) 0.586 vs 0.546 -- it keeps 4.
static int f16_pick_nw(int64_t BH, int ntile, bool dual) {
    if (const char* e = getenv("GT_F16_NW")) { const int v = atoi(e); if (v == 4 || v == 8) return v; }
    return (!dual && ntile >= 16 && BH * ((ntile + 7) / 8) >= 384) ? 8 : 4;
}

template <int DP>
static void fourier16_launch(F16P& p, bool dual, bool block16, int64_t BH, hipStream_t st) {
    const int nw = f16_pick_nw(BH, p.ntile, dual);
    p.nblk = ceil_div(p.ntile, nw);
    p.total = (int)(BH * p.nblk);
    if (nw == 8) fourier16_launch_nw<DP, 8>(p, dual, block16, st);
    else fourier16_launch_nw<DP, 4>(p, dual, block16, st);
}

static int64_t f16_hdr_bytes(int64_t BH, int ntile) { return (BH * ntile * 8 + 1023) / 1024 * 1024; }
static int f16_img_bytes(int DP) {
    switch (DP) {
        case 20: return F16G<20>::IMG;
        case 36: return F16G<36>::IMG;
        case 52: return F16G<52>::IMG;
        default: return 0;
    }
}

}  // namespace gt

using namespace gt;

extern "C" int64_t gt_fourier16_image_bytes(int32_t B, int32_t n, int32_t h, int32_t DP) {
    const int img = f16_img_bytes(DP);
    if (!img || B <= 0 || n <= 0 || h <= 0) return 0;
    const int ntile = ceil_div(n, 32);
    return f16_hdr_bytes((int64_t)B * h, ntile) + (int64_t)B * h * ntile * img;
}

// S[bh][q][k] *= keep(q, k) / (1 - p) with the 4 x 4-block mask of F16_DROPB: what the materialising path (attention weights
// requested) applies to the score matrix a gt_gemm wrote, so that it draws the mask the fused kernels draw.
__global__ __launch_bounds__(256) void dropout_block16_kernel(float* S, int64_t BH, int n, DropDev d) {
    const uint32_t nq4 = (uint32_t)((n + 3) >> 2);
    const int64_t total = BH * (int64_t)n * nq4;
    const uint32_t key = drop_key_dev(d);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const uint32_t k4 = (uint32_t)(i % nq4);
        const int64_t row = i / nq4;                               // bh * n + q
        const uint32_t q = (uint32_t)(row % n), bh = (uint32_t)(row / n);
        uint32_t x = ((bh * nq4 + (q >> 2)) * nq4 + k4) * 0x9e3779b1u + key;
        x = fmix32(x);
        float* s = S + row * (int64_t)n + 4 * (int64_t)k4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * k4 + c < (uint32_t)n) s[c] = ((x >> (16u + 4u * (q & 3u) + c)) & 1u) ? s[c] * d.scale : 0.f;
    }
}

extern "C" int gt_dropout_block16(float* S, int64_t BH, int32_t n, const gt_dropout* drop, void* stream) {
    if (!S || BH <= 0 || n <= 0 || !drop || drop->p != 0.5f || !drop->seed) return GT_EINVAL;
    const int64_t total = BH * (int64_t)n * ((n + 3) >> 2);
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(dropout_block16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, S, BH, n, make_drop(drop));
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_fourier16_presplit(const float* X0, const float* X1, const float* X2, const float* X3, void* I0, void* I1,
                                     void* I2, void* I3, int32_t B, int32_t n, int32_t h, int32_t DP, void* stream) {
    if (B <= 0 || n <= 0 || h <= 0 || !X0 || !I0) return GT_EINVAL;
    if (!f16_img_bytes(DP)) return GT_ENOTSUP;
    if ((int64_t)B * h > 65535) return GT_EINVAL;
    F16PreP p{};
    const float* X[4] = {X0, X1, X2, X3};
    void* I[4] = {I0, I1, I2, I3};
    int nt = 0;
    for (int i = 0; i < 4; ++i) {
        if (!X[i]) break;
        if (!I[i]) return GT_EINVAL;
        if ((reinterpret_cast<uintptr_t>(X[i]) | reinterpret_cast<uintptr_t>(I[i])) & 15) return GT_EALIGN;
        p.X[nt] = X[i];
        p.I[nt] = static_cast<uint8_t*>(I[i]);
        ++nt;
    }
    p.n = n; p.h = h; p.ntile = ceil_div(n, 32);
    p.hdr = f16_hdr_bytes((int64_t)B * h, p.ntile);
    const dim3 grid((unsigned)p.ntile, (unsigned)(B * h), (unsigned)nt);
    hipStream_t st = (hipStream_t)stream;
    switch (DP) {
        case 20: hipLaunchKernelGGL(fourier16_presplit_kernel<20>, grid, dim3(64), 0, st, p); break;
        case 36: hipLaunchKernelGGL(fourier16_presplit_kernel<36>, grid, dim3(64), 0, st, p); break;
        default: hipLaunchKernelGGL(fourier16_presplit_kernel<52>, grid, dim3(64), 0, st, p); break;
    }
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_fourier16_attn(const void* F1, const void* F2, const void* T1, const void* T2, float* O1, float* O2,
                                 int32_t B, int32_t n, int32_t h, int32_t DP, float scale, const float* mask,
                                 const gt_dropout* drop, int32_t block16, int32_t owner_is_key, void* stream) {
    if (!F1 || !T1 || !T2 || !O1 || B <= 0 || n <= 0 || h <= 0) return GT_EINVAL;
    if (!f16_img_bytes(DP)) return GT_ENOTSUP;
    if (block16 && drop && drop->p > 0.f && drop->p != 0.5f) return GT_EINVAL;
    const bool dual = F2 != nullptr;
    if (dual && !O2) return GT_EINVAL;
    if (drop && drop->p > 0.f && !drop->seed) return GT_EINVAL;
    const uintptr_t al = reinterpret_cast<uintptr_t>(F1) | reinterpret_cast<uintptr_t>(F2) | reinterpret_cast<uintptr_t>(T1) |
                         reinterpret_cast<uintptr_t>(T2) | reinterpret_cast<uintptr_t>(O1) | reinterpret_cast<uintptr_t>(O2);
    if (al & 15) return GT_EALIGN;
    F16P p{};
    p.F1 = static_cast<const uint8_t*>(F1); p.F2 = static_cast<const uint8_t*>(F2);
    p.T1 = static_cast<const uint8_t*>(T1); p.T2 = static_cast<const uint8_t*>(T2);
    p.O1 = O1; p.O2 = O2; p.mask = mask; p.drop = make_drop(mask ? nullptr : drop);
    p.n = n; p.h = h; p.ntile = ceil_div(n, 32);
    if ((int64_t)B * h * ceil_div(p.ntile, 4) > 0x7fffffff) return GT_EINVAL;
    p.hdr = f16_hdr_bytes((int64_t)B * h, p.ntile);
    p.scale = scale; p.owner_is_key = owner_is_key;
    hipStream_t st = (hipStream_t)stream;
    switch (DP) {
        case 20: fourier16_launch<20>(p, dual, block16 != 0, (int64_t)B * h, st); break;
        case 36: fourier16_launch<36>(p, dual, block16 != 0, (int64_t)B * h, st); break;
        default: fourier16_launch<52>(p, dual, block16 != 0, (int64_t)B * h, st); break;
    }
    GT_LAUNCH_CHECK();
    return 0;
}
