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

// 8 consecutive k (k0 .. k0+7) of operand row x:  L == 0: base[x*ld + k],  L == 1: base[k*ld + x]
template <int L>
__device__ __forceinline__ void x3_load8(const float* __restrict__ base, int64_t ld, int x, int X, int k0, int kend,
                                         int vec, const DropDev& dd, uint32_t dkey, int64_t dld, int64_t dboff,
                                         float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (x >= X || k0 >= kend) return;
    if (L == 0) {
        const float* ptr = base + (int64_t)x * ld + k0;
        if (vec && k0 + 7 < kend) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ptr);
            const f32x4 b = *reinterpret_cast<const f32x4*>(ptr + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
            v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < kend) v[j] = ptr[j];
        }
        if (dd.thresh) {
            const uint32_t di = (uint32_t)(dboff + (int64_t)x * dld + k0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= drop_mul(dd, dkey, di + j);
        }
    } else {
        const float* ptr = base + (int64_t)k0 * ld + x;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < kend) v[j] = ptr[(int64_t)j * ld];
        if (dd.thresh) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] *= drop_mul(dd, dkey, (uint32_t)(dboff + (int64_t)(k0 + j) * dld + x));
        }
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
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int LA, int LB, int PLANES>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const GemmP p) {
    constexpr int STAGE = 2 * PLANES * X3_PLANE;                          // A planes then B planes
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

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
    const DropDev nodrop{0u, 0u, 1.f, nullptr};

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
    const bool do_acs = (LA == 1) && p.acs != nullptr && tn == 0;
    auto g2r = [&](int k0) {
        x3_load8<LA>(A, lda_c, m0 + arow, p.M, k0 + 8 * akh, kend_c, avec_c, p.a_drop, akey, p.a_drop_ld, adoff, ra);
        x3_load8<LB>(Bm, ldb_c, n0 + brow, p.N, k0 + 8 * bkh, kend_c, bvec_c, nodrop, 0u, 0, 0, rb);
        if (LA == 1 && do_acs) asum += ((ra[0] + ra[1]) + (ra[2] + ra[3])) + ((ra[4] + ra[5]) + (ra[6] + ra[7]));
    };
    auto r2s = [&](int buf) {
        char* st = smem + buf * STAGE;
        x3_store8<PLANES>(st, arow, akh, ra);
        x3_store8<PLANES>(st + PLANES * X3_PLANE, brow, bkh, rb);
    };

    const int nk1 = (kend > kbeg) ? (kend - kbeg + X3_BK - 1) / X3_BK : 0;
    const int nk = nk1 + (p.K2 > 0 ? (p.K2 + X3_BK - 1) / X3_BK : 0);
    auto enter_seg2 = [&]() {
        A = p.A2 + b0 * p.a2_bs0 + b1 * p.a2_bs1;
        Bm = p.B2 + b0 * p.b2_bs0 + b1 * p.b2_bs1;
        lda_c = p.lda2; ldb_c = p.ldb2; kend_c = p.K2; avec_c = p.a2_vec; bvec_c = p.b2_vec;
    };
    if (nk > 0) {
        if (nk1 == 0) { enter_seg2(); g2r(0); }
        else g2r(kbeg);
        r2s(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            if (kt + 1 == nk1) enter_seg2();
            g2r(kt + 1 < nk1 ? kbeg + (kt + 1) * X3_BK : (kt + 1 - nk1) * X3_BK);
        }
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
        for (int s = 2 * (PLANES - 1); s >= 0; --s) {
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
        if (kt + 1 < nk) r2s(buf ^ 1);
        __syncthreads();
    }

    if (LA == 1 && do_acs) {              // uniform per block; the stages are free after the loop's last barrier
        float* part = reinterpret_cast<float*>(smem);
        part[akh * X3_BM + arow] = asum;
        __syncthreads();
        if (tid < X3_BM && m0 + tid < p.M)
            p.acs[((int64_t)blockIdx.y * gridDim.z + z) * p.M + m0 + tid] = part[tid] + part[X3_BM + tid];
    }

    // ------------------------------- epilogue: lane = one row m per accumulator row tile -------------------
    const int64_t coff = b0 * p.c_bs0 + b1 * p.c_bs1 + (int64_t)blockIdx.y * p.c_split;
    float* __restrict__ C = p.C + coff;
    const uint32_t dkey = drop_key_dev(p.drop);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nb = n0 + wn * 64 + 32 * j + 8 * g + 4 * lh;
            if (nb >= p.N) continue;
            const bool full = nb + 4 <= p.N;
            float biasv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) biasv[t] = (p.bias && nb + t < p.N) ? p.bias[nb + t] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = m0 + wm * 64 + 32 * i + lr;
                if (m >= p.M) continue;
                float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                ep_row<4>(p, v, biasv, C, m, nb, z, b0, b1, full, dkey);
            }
        }
    }
}

bool x3_shape_ok(const gt_gemm_desc* d) {
    // whole 128 x 128 tiles dominate (the padding of a partial edge tile is bounded by the sizes below)
    return d->ep_mode == GT_EP_NORMAL && d->M >= 96 && d->N >= 96 && d->K >= 16;
}

template <int LA, int LB>
static void x3_launch_planes(const GemmP& p, int planes, dim3 grid, hipStream_t st) {
    if (planes == 1) hipLaunchKernelGGL((gemm_x3_kernel<LA, LB, 1>), grid, dim3(256), 0, st, p);
    else if (planes == 2) hipLaunchKernelGGL((gemm_x3_kernel<LA, LB, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_x3_kernel<LA, LB, 3>), grid, dim3(256), 0, st, p);
}

int x3_launch(const GemmP& p, int layout_a, int layout_b, int planes, unsigned tiles, unsigned split, unsigned batch,
              hipStream_t st) {
    if (planes < 1 || planes > 3) return GT_EINVAL;
    const dim3 grid(tiles, split, batch);
    const int lay = layout_a * 2 + layout_b;
    if (lay == 0) x3_launch_planes<0, 0>(p, planes, grid, st);
    else if (lay == 1) x3_launch_planes<0, 1>(p, planes, grid, st);
    else if (lay == 2) x3_launch_planes<1, 0>(p, planes, grid, st);
    else x3_launch_planes<1, 1>(p, planes, grid, st);
    GT_LAUNCH_CHECK();
    return 0;
}

const char* x3_kernel_name(int layout_a, int layout_b, int planes) {
    static thread_local char buf[96];
    snprintf(buf, sizeof(buf), "void gt::gemm_x3_kernel<%d, %d, %d>(gt::GemmP)", layout_a, layout_b, planes);
    return buf;
}

}  // namespace gt
