// Batched fp32 GEMM on the CDNA4 matrix pipe (v_mfma_f32_16x16x4_f32) with fused
// prologue (stateless dropout mask on A) and epilogue (bias, rank-p update, activation,
// aux-multiply, dropout, residual).  One kernel template serves every contraction of the
// encoder layer and the spectral decoder, forward and backward (see include/gt_hip.h).
//
// Tiling (gfx950): block = WM x WN waves of 64 lanes; each wave owns a (16*MT) x (16*NT)
// output tile as MT x NT MFMA 16x16 accumulators.  K is consumed in steps of 16 through a
// double-buffered LDS stage.  LDS images are k-major ([16][BM], [16][BN]) so that a lane's
// MT (NT) operands for one k are contiguous: one ds_read_b128 feeds 4 MFMAs.  Because the
// lane->row map of the MFMA is free, lane i takes rows {MT*i .. MT*i+MT-1}: the accumulator
// of a lane then holds NT consecutive output columns -> 16-byte global stores.
// An XOR swizzle on the column index (bits 3-4, keyed by k>>2) makes both the transposing
// ds_write_b32 of k-contiguous operands and the ds_read_b128 of the MFMA loop conflict-free.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gt_gemm_core.h"

namespace gt {

// HEAD: instance with the fused two-layer-head epilogues instead of the general one (kept out of the
// general instances: its register footprint would cost them occupancy)
template <int LA, int LB, int MT, int NT, int WM, int WN, int BK, int HEAD = 0>
// (4 waves per SIMD requested for the 16-deep general instances: keeps the register allocation at <= 128)
__global__ __launch_bounds__(WM* WN * 64, ((HEAD == 0 && BK == 16) ? 4 : 1)) void gemm_kernel(const GemmP p) {
    constexpr int BM = WM * 16 * MT, BN = WN * 16 * NT, T = WM * WN * 64;
    constexpr int FA = BM * BK / 4, FB = BN * BK / 4;          // float4 per stage
    constexpr int NVA = (FA + T - 1) / T, NVB = (FB + T - 1) / T;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * BM + 2 * BK * BN];
    float* sA = smem;
    float* sB = smem + 2 * BK * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, kq = lane >> 4;
    // XCD-aware tile order: blocks are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), so give each
    // XCD a contiguous range of tile ids -- the N-tiles that share an A row panel then hit the same L2
    // instead of fetching the panel once per XCD (bijective for any tile count; speed only).
    int tile;
    {
        const int tiles = gridDim.x, q = tiles >> 3, r = tiles & 7;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z, b0 = z / p.batch1, b1 = z % p.batch1;
    const int kbeg = blockIdx.y * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);

    // operands of the K segment being loaded (uniform values; switched once when a second product follows)
    const float* A = p.A + b0 * p.a_bs0 + b1 * p.a_bs1;
    const float* Bm = p.B + b0 * p.b_bs0 + b1 * p.b_bs1;
    int64_t lda_c = p.lda, ldb_c = p.ldb;
    int kend_c = min(p.K, (int)blockIdx.y * p.k_chunk + p.k_chunk), avec_c = p.a_vec, bvec_c = p.b_vec;
    const uint32_t akey = drop_key_dev(p.a_drop);
    const int64_t adoff = (int64_t)z * p.a_drop_bstride;
    const DropDev nodrop{0u, 0u, 1.f, nullptr};

    f32x4 acc[MT][NT];
#pragma unroll
    for (int s = 0; s < MT; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ra[NVA], rb[NVB];
    // LA == 1: a thread's float4 always covers the same 4 rows m of the tile (T is a multiple of BM/4),
    // so the row sums of A (= bias gradients in the weight-gradient use) accumulate in registers for free
    f32x4 asum = {0.f, 0.f, 0.f, 0.f};
    const bool do_acs = (LA == 1) && p.acs != nullptr && tn == 0;
    auto g2r = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int idx = tid + i * T;
            if (FA % T == 0 || idx < FA) {
                ra[i] = gload<LA, BM, BK>(A, lda_c, m0, p.M, k0, kend_c, idx, avec_c, p.a_drop, akey,
                                          p.a_drop_ld, adoff);
                if (LA == 1 && do_acs) asum += ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int idx = tid + i * T;
            if (FB % T == 0 || idx < FB)
                rb[i] = gload<LB, BN, BK>(Bm, ldb_c, n0, p.N, k0, kend_c, idx, bvec_c, nodrop, 0u, 0, 0);
        }
    };
    auto r2s = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int idx = tid + i * T;
            if (FA % T == 0 || idx < FA) sstore<LA, BM, BK>(sA + buf * BK * BM, idx, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int idx = tid + i * T;
            if (FB % T == 0 || idx < FB) sstore<LB, BN, BK>(sB + buf * BK * BN, idx, rb[i]);
        }
    };

    const int nk1 = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    const int nk = nk1 + (p.K2 > 0 ? (p.K2 + BK - 1) / BK : 0);
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
#ifndef GT_ABL_NOLOAD
        if (kt + 1 < nk) {
            if (kt + 1 == nk1) enter_seg2();
            g2r(kt + 1 < nk1 ? kbeg + (kt + 1) * BK : (kt + 1 - nk1) * BK);
        }
#endif
        const float* __restrict__ cA = sA + buf * BK * BM;
        const float* __restrict__ cB = sB + buf * BK * BN;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int k = 4 * ks + kq;
            const int swa = ((ks & 3) << 3) & (BM - 1), swb = ((ks & 3) << 3) & (BN - 1);
            float a[MT], b[NT];
            lds_frag<MT>(cA + k * BM + ((wm * 16 * MT + MT * li) ^ swa), a);
            lds_frag<NT>(cB + k * BN + ((wn * 16 * NT + NT * li) ^ swb), b);
#pragma unroll
            for (int s = 0; s < MT; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[s][t] = mfma16(a[s], b[t], acc[s][t]);
        }
        if (kt + 1 < nk) r2s(buf ^ 1);
        __syncthreads();
    }

    if (LA == 1 && do_acs) {        // uniform per block; smem is free after the loop's last barrier
        constexpr int Q = BM / 4, ROWS = T / Q;
        static_assert(T % Q == 0 && ROWS * BM <= 2 * BK * BM, "row-sum scratch must fit the A stage");
        *reinterpret_cast<f32x4*>(&smem[(tid / Q) * BM + 4 * (tid % Q)]) = asum;
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) t += smem[r * BM + tid];
            p.acs[((int64_t)blockIdx.y * gridDim.z + z) * p.M + m0 + tid] = t;
        }
    }

    if (HEAD > 0) {
        head_epilogue<MT, NT, WM, WN, (HEAD > 0 ? HEAD : 1)>(p, acc, smem, m0, wm, wn, li, kq, tm);
        return;
    }
    // ------------------------------- epilogue -------------------------------------------------
    gemm_epilogue<MT, NT>(p, acc, m0 + wm * 16 * MT, n0 + wn * 16 * NT + NT * li, z, b0, b1,
                                      (int)blockIdx.y, kq);
}

// =================================================================================================
// Streamed kernel (aligned operands, 128-wide N tiles): persistent blocks, direct global->LDS loads.
//
// The v1 kernel above runs every tile as load -> MFMA -> store with all co-resident blocks in the same
// phase, so HBM and the matrix pipe take turns.  Here a block walks a list of (tile, K-slice) items as ONE
// stream of 32-deep K stages: stage g+1 is requested with `global_load_lds_dwordx4` (no VGPR staging, no
// ds_write pass) right after the barrier that publishes stage g, and the stream does not stop at a tile
// boundary -- the first stage of the next tile is in flight while the current tile's last MFMAs and its
// epilogue run, and the epilogue's stores drain under the next tile's MFMAs.
//
// LDS images (per stage, A then B), chosen so that a direct load (wave-uniform base + lane*16 B) lands
// conflict-free for the MFMA fragment reads:
//   k-contiguous operand (L==0):  [rows][32]  with the 16-byte granule g of row r stored at slot
//                                 g ^ ((r>>2)&7)           -> lane reads one float per (row, k)
//   x-contiguous operand (L==1):  [32][BX]    with column granules XOR-ed by ((k>>2)&3)<<1 (same image
//                                 as v1)                   -> lane reads MT/NT consecutive floats
// The swizzles are applied on the SOURCE address (the LDS side of a direct load is linear).
// Out-of-range rows / K-tail granules read a 16-byte device zero instead.
__device__ __attribute__((aligned(16))) float gt_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct SItem {
    int m0, n0, z, b0, b1, sidx, kbeg, kend, nk;
    const float* A;
    const float* B;
    int64_t adoff;
};

// second launch-bound argument = waves per SIMD the register allocation must leave room for: the LDS
// footprint admits 2 (128-row tiles) or 3 (64-row tiles) blocks per CU
template <int LA, int LB, int MT>
__global__ __launch_bounds__(256, (MT == 4 ? 2 : 3)) void gemm_stream_kernel(const GemmP p) {
    constexpr int NT = 4, WN = 2, BK = 32, T = 256;
    constexpr int BM = 32 * MT, BN = 128;
    constexpr int SA = BM * BK, SB = BN * BK, STAGE = SA + SB;
    constexpr int NIA = BM / 32, NIB = BN / 32;               // 1-KiB load instructions per wave per stage
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, kq = lane >> 4;
    const uint32_t akey = drop_key_dev(p.a_drop);

    const int per_xcd = (p.n_work + 7) >> 3;
    const int xcd = blockIdx.x & 7, slots = gridDim.x >> 3;
    int jpos = blockIdx.x >> 3;

    auto decode = [&](int j, SItem& it) -> bool {
        const int w = xcd * per_xcd + j;
        if (j >= per_xcd || w >= p.n_work) return false;
        const int tiles = p.tiles_m * p.tiles_n;
        const int t = w % tiles, r = w / tiles;
        it.sidx = r % p.n_split;
        it.z = r / p.n_split;
        it.m0 = (t / p.tiles_n) * BM;
        it.n0 = (t % p.tiles_n) * BN;
        it.b0 = it.z / p.batch1;
        it.b1 = it.z % p.batch1;
        it.kbeg = it.sidx * p.k_chunk;
        it.kend = min(p.K, it.kbeg + p.k_chunk);
        it.nk = (it.kend - it.kbeg + BK - 1) / BK;
        it.A = p.A + it.b0 * p.a_bs0 + it.b1 * p.a_bs1;
        it.B = p.B + it.b0 * p.b_bs0 + it.b1 * p.b_bs1;
        it.adoff = (int64_t)it.z * p.a_drop_bstride;
        return true;
    };

    // request one 32-deep stage (A and B tiles of item `it` at k0) into buffer `buf`
    auto issue = [&](const SItem& it, int k0, int buf) {
        float* sa = smem + buf * STAGE;
        float* sb = sa + SA;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int q = wave * NIA + i;                      // 1-KiB chunk of the A image
            const float* src;
            if (LA == 0) {
                const int row = 8 * q + (lane >> 3), slot = lane & 7;
                const int g = slot ^ ((row >> 2) & 7);
                const int m = it.m0 + row, k = k0 + 4 * g;
                src = (m < p.M && k < it.kend) ? it.A + (int64_t)m * p.lda + k : gt_zero16;
            } else {
                constexpr int GPR = BM / 4;                    // granules per k-row
                const int e = q * 64 + lane, kr = e / GPR, pc = e % GPR;
                const int g = pc ^ ((((kr >> 2) & 3) << 1) & (GPR - 1));
                const int k = k0 + kr, m = it.m0 + 4 * g;
                src = (k < it.kend && m < p.M) ? it.A + (int64_t)k * p.lda + m : gt_zero16;
            }
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(sa + q * 256), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const int q = wave * NIB + i;
            const float* src;
            if (LB == 0) {
                const int row = 8 * q + (lane >> 3), slot = lane & 7;
                const int g = slot ^ ((row >> 2) & 7);
                const int n = it.n0 + row, k = k0 + 4 * g;
                src = (n < p.N && k < it.kend) ? it.B + (int64_t)n * p.ldb + k : gt_zero16;
            } else {
                constexpr int GPR = BN / 4;
                const int e = q * 64 + lane, kr = e / GPR, pc = e % GPR;
                const int g = pc ^ ((((kr >> 2) & 3) << 1) & (GPR - 1));
                const int k = k0 + kr, n = it.n0 + 4 * g;
                src = (k < it.kend && n < p.N) ? it.B + (int64_t)k * p.ldb + n : gt_zero16;
            }
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(sb + q * 256), 16, 0, 0);
        }
    };

    SItem cur, nxt;
    bool light_wait = false;
    bool have = decode(jpos, cur);
    int g = 0;                                                 // running stage counter -> LDS buffer g&1
    if (have) issue(cur, cur.kbeg, 0);

    while (have) {
        jpos += slots;
        const bool have_next = decode(jpos, nxt);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int s = 0; s < MT; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float asum[MT];
#pragma unroll
        for (int s = 0; s < MT; ++s) asum[s] = 0.f;
        const bool do_acs = (LA == 1) && p.acs != nullptr && cur.n0 == 0 && wn == 0;

        for (int kt = 0; kt < cur.nk; ++kt, ++g) {
            // stage g has landed for this wave (vmcnt) and for everybody (barrier); everybody is also done
            // reading the other buffer, which the next request overwrites.
            // Right after the epilogue of a FULL tile this wave has issued >= 4*MT output stores AFTER the
            // stage-g request; VMEM operations retire in order on gfx9-class counters, so "at most 4*MT
            // outstanding" already implies the (older) stage loads have landed -- the stores may drain under
            // the next MFMAs instead of stalling the first stage of every tile.
            if (light_wait) {
                if (MT == 4) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                light_wait = false;
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
#ifndef GT_ABL_NOLOAD
            if (kt + 1 < cur.nk) issue(cur, cur.kbeg + (kt + 1) * BK, (g + 1) & 1);
            else if (have_next) issue(nxt, nxt.kbeg, (g + 1) & 1);
#endif
            const float* __restrict__ cA = smem + (g & 1) * STAGE;
            const float* __restrict__ cB = cA + SA;
            const int kbase = cur.kbeg + kt * BK;
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                const int k = 4 * ks + kq;
                float a[MT], b[NT];
                if (LA == 0) {
#pragma unroll
                    for (int s = 0; s < MT; ++s) {
                        const int row = wm * 16 * MT + MT * li + s;
                        a[s] = cA[row * 32 + 4 * (ks ^ ((row >> 2) & 7)) + kq];
                    }
                } else {
                    lds_frag<MT>(cA + k * BM + ((wm * 16 * MT + MT * li) ^ (((ks & 3) << 3) & (BM - 1))), a);
                }
                if (LB == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int row = wn * 16 * NT + NT * li + t;
                        b[t] = cB[row * 32 + 4 * (ks ^ ((row >> 2) & 7)) + kq];
                    }
                } else {
                    lds_frag<NT>(cB + k * BN + ((wn * 16 * NT + NT * li) ^ (((ks & 3) << 3) & (BN - 1))), b);
                }
                if (p.a_drop.thresh) {                          // dropout mask on A, regenerated from its index
#pragma unroll
                    for (int s = 0; s < MT; ++s) {
                        const int m = cur.m0 + wm * 16 * MT + MT * li + s, kk = kbase + k;
                        const int64_t di = cur.adoff + (LA == 0 ? (int64_t)m * p.a_drop_ld + kk
                                                                : (int64_t)kk * p.a_drop_ld + m);
                        a[s] *= drop_mul(p.a_drop, akey, (uint32_t)di);
                    }
                }
                if (LA == 1 && do_acs) {
#pragma unroll
                    for (int s = 0; s < MT; ++s) asum[s] += a[s];
                }
#pragma unroll
                for (int s = 0; s < MT; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[s][t] = mfma16(a[s], b[t], acc[s][t]);
            }
        }

        if (LA == 1 && do_acs) {        // row sums of A: combine the 4 k-lanes; lanes kq == 0 hold MT rows each
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                float v = asum[s];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int m = cur.m0 + wm * 16 * MT + MT * li + s;
                if (kq == 0 && m < p.M)
                    p.acs[((int64_t)cur.sidx * p.n_batch + cur.z) * p.M + m] = v;
            }
        }
        gemm_epilogue<MT, NT>(p, acc, cur.m0 + wm * 16 * MT, cur.n0 + wn * 16 * NT + NT * li, cur.z, cur.b0, cur.b1,
                              cur.sidx, kq);
        // every row and column of this wave's sub-tile was in range => exactly 4*MT (or more, with `pre`)
        // store instructions were issued by the epilogue
        light_wait = p.light_wait && (cur.m0 + BM <= p.M) && (cur.n0 + BN <= p.N) && cur.nk > 0;
        cur = nxt;
        have = have_next;
    }
}

// out_z[m][n..n+3] = alpha * sum_s slab[s][z][m][n..n+3]   (N % 4 == 0, 16-byte aligned everything)
// Block = 32 consecutive float4 outputs x 8 slab lanes; slab lane j sums slabs j, j+8, ... in order,
// then the 8 partials are combined in a fixed order through LDS (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float* __restrict__ slabs, int nsplit,
                                                             int64_t slab_stride, int M, int N4, int batch1,
                                                             float alpha, float* __restrict__ C, int64_t ldc,
                                                             int64_t c_bs0, int64_t c_bs1, int64_t total4) {
    __shared__ f32x4 part[8][32];
    const int ox = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + ox;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < total4) {
        const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + i;
        const int64_t st4 = slab_stride / 4;
#pragma unroll 4
        for (int k = sl; k < nsplit; k += 8) s += src[k * st4];
    }
    part[sl][ox] = s;
    __syncthreads();
    if (sl == 0 && i < total4) {
#pragma unroll
        for (int k = 1; k < 8; ++k) s += part[k][ox];
        const int n4 = (int)(i % N4);
        const int64_t r = i / N4;
        const int m = (int)(r % M);
        const int z = (int)(r / M);
        *reinterpret_cast<f32x4*>(C + (z / batch1) * c_bs0 + (z % batch1) * c_bs1 + (int64_t)m * ldc + 4 * n4) =
            alpha * s;
    }
}

// out_z[m][n] = alpha * sum_s slab[s][z][m][n]
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int nsplit, int64_t slab_stride,
                                     int M, int N, int batch1, float alpha, float* __restrict__ C,
                                     int64_t ldc, int64_t c_bs0, int64_t c_bs1, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += slabs[k * slab_stride + i];
        const int n = (int)(i % N);
        const int64_t r = i / N;
        const int m = (int)(r % M);
        const int z = (int)(r / M);
        C[(z / batch1) * c_bs0 + (z % batch1) * c_bs1 + (int64_t)m * ldc + n] = alpha * s;
    }
}

struct Cfg { int mt, nt, wm, wn; };
static const Cfg kCfgs[] = {{4, 4, 2, 2}, {2, 4, 2, 2}, {2, 2, 2, 2}, {2, 2, 4, 1}, {2, 1, 4, 1}};
constexpr int kNumCfg = 5;

template <int LA, int LB, int BK>
static void launch_cfg(int cfg, dim3 grid, hipStream_t st, const GemmP& p) {
    switch (cfg) {
        case 0: hipLaunchKernelGGL((gemm_kernel<LA, LB, 4, 4, 2, 2, BK>), grid, dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<LA, LB, 2, 4, 2, 2, BK>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<LA, LB, 2, 2, 2, 2, BK>), grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((gemm_kernel<LA, LB, 2, 2, 4, 1, BK>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_kernel<LA, LB, 2, 1, 4, 1, BK>), grid, dim3(256), 0, st, p); break;
    }
}

// fused two-layer head: activations [tokens, K] times nn.Linear weight [N, K] only (LA = LB = 0)
template <int NO>
static void launch_head_no(int cfg, dim3 grid, hipStream_t st, const GemmP& p) {
    switch (cfg) {
        case 0: hipLaunchKernelGGL((gemm_kernel<0, 0, 4, 4, 2, 2, 16, NO>), grid, dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<0, 0, 2, 4, 2, 2, 16, NO>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<0, 0, 2, 2, 2, 2, 16, NO>), grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((gemm_kernel<0, 0, 2, 2, 4, 1, 16, NO>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_kernel<0, 0, 2, 1, 4, 1, 16, NO>), grid, dim3(256), 0, st, p); break;
    }
}
static void launch_head(int cfg, dim3 grid, hipStream_t st, const GemmP& p) {
    if (p.n_out == 1) launch_head_no<1>(cfg, grid, st, p);
    else launch_head_no<4>(cfg, grid, st, p);
}

static int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;      // MI355X
        n = v;
    }
    return n;
}

// Persistent launch: the grid never exceeds what is resident at once (a block runs until its share of the
// work list is empty, so a block waiting for a slot would serialise behind the others).
template <int LA, int LB, int MT>
static void launch_stream(hipStream_t st, const GemmP& p) {
    static const int per_cu = [] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_stream_kernel<LA, LB, MT>, 256, 0) != hipSuccess ||
            n < 1)
            n = 1;
        return std::min(n, 4);
    }();
    int nblk = std::min(p.n_work, num_cus() * per_cu);
    if (const char* e = getenv("GT_GEMM_BLOCKS")) nblk = std::min(p.n_work, std::max(1, atoi(e)));
    nblk = ((nblk + 7) / 8) * 8;
    GemmP q = p;
    q.light_wait = 1;
    if (const char* e = getenv("GT_GEMM_LIGHTWAIT")) q.light_wait = atoi(e) != 0;
    if (getenv("GT_GEMM_DEBUG"))
        fprintf(stderr, "[gt_gemm] stream<%d,%d,%d> M=%d N=%d K=%d work=%d per_cu=%d grid=%d\n", LA, LB, MT, p.M, p.N,
                p.K, p.n_work, per_cu, nblk);
    hipLaunchKernelGGL((gemm_stream_kernel<LA, LB, MT>), dim3(nblk), dim3(256), 0, st, q);
}

struct Plan { int cfg, bm, bn, bk, tiles_m, tiles_n, split, k_chunk, stream, x3; };

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool m4(int64_t v) { return (v & 3) == 0; }

static bool has_epilogue(const gt_gemm_desc* d) {
    return d->c_masked || d->bias || d->rp || d->add || d->pre || d->act || d->aux_op || d->drop.p > 0.f || d->res ||
           d->out_scale != 1.f || d->ep_mode != GT_EP_NORMAL || d->K2 > 0;
}

// number of bf16 planes the split-operand kernel (gt_gemm_x3.hip) would use for d, 0 = fp32 MFMA kernels
static int x3_planes(const gt_gemm_desc* d) {
    // GT_PREC_F16X2: the packed-B kernels run the two-term fp16 arithmetic, every other split-operand launch bf16x3
    const int planes = (d->precision == GT_PREC_BF16X3 || d->precision == GT_PREC_F16X2) ? 3
                       : d->precision == GT_PREC_BF16X2 ? 2 : d->precision == GT_PREC_BF16 ? 1 : 0;
    if (!planes || !x3_shape_ok(d) || (d->a_colsum && d->layout_a != 1)) return 0;
    return planes;
}

static int make_plan(const gt_gemm_desc* d, Plan* pl) {
    if (d->M <= 0 || d->N <= 0 || d->K < 0 || d->batch0 <= 0 || d->batch1 <= 0) return GT_EINVAL;
    if (d->precision < GT_PREC_F32 || d->precision > GT_PREC_F16X2) return GT_EINVAL;
    if (d->act < GT_ACT_NONE || (d->act > GT_ACT_SILU && d->act != GT_ACT_DROP_SILU && d->act != GT_ACT_SILU2))
        return GT_EINVAL;                          // GT_ACT_GELU: elementwise entry points only
    if (d->act == GT_ACT_SILU2 && d->drop.p > 0.f) return GT_EINVAL;
    const int64_t batch = (int64_t)d->batch0 * d->batch1;
    if (batch > 65535) return GT_EINVAL;
    pl->x3 = x3_planes(d);
    // Tile choice by a small cost model (calibrated on MI355X with tools/gemm_probe.py): the blocks that
    // share a CU share its matrix pipes, so time ~ ceil(tiles / CUs) * tile area / efficiency of the
    // configuration (MFMAs per LDS read / per barrier).  Padding waste shows up through the tile count.
    static const double kEff[kNumCfg] = {1.00, 0.95, 0.80, 0.70, 0.45};
    int c = 0;
    double best = 0.0;
    for (int i = 0; i < kNumCfg; ++i) {
        const int bm = kCfgs[i].wm * 16 * kCfgs[i].mt, bn = kCfgs[i].wn * 16 * kCfgs[i].nt;
        const bool head_ep = d->ep_mode == GT_EP_ROWDOT || d->ep_mode == GT_EP_MLP_BWD;
        if (head_ep && bn < d->N) continue;                         // the fused head needs whole rows per block
        // the 128x128 head instance needs > 256 registers (one block per CU): the 64x128 one runs two
        if (head_ep && i == 0 && d->N <= 128) continue;
        const double tiles = (double)ceil_div(d->M, bm) * ceil_div(d->N, bn) * (double)batch;
        // under-filled grids: with split-K available the K-slices fill the chip (time ~ total padded work),
        // otherwise every block has a CU to itself (time ~ one tile)
        const bool can_split = d->split_k != 1 && d->K >= 512 && !has_epilogue(d);
        const double units = tiles >= 256.0 ? std::ceil(tiles / 256.0) : (can_split ? tiles / 256.0 : 1.0);
        const double cost = units * bm * bn / kEff[i];
        if (best == 0.0 || cost < best) { best = cost; c = i; }
    }
    if (const char* e = getenv("GT_GEMM_CFG")) {      // tuning/debug override (tools/gemm_bench.py)
        const int f = atoi(e);
        if (f >= 0 && f < kNumCfg) c = f;
    }
    if (pl->x3) c = 0;                                // the split-operand kernel has one geometry: 128 x 128 tiles
    pl->cfg = c;
    // BK=32 pays for the long-K reductions with row-contiguous operands (weight gradients); the
    // short-K token GEMMs are prologue/epilogue-bound and run better with the smaller stage
    pl->bk = (d->layout_a == 1 && d->layout_b == 1 && d->K >= 512) ? 32 : 16;
    if (const char* e = getenv("GT_GEMM_BK")) pl->bk = (atoi(e) == 16) ? 16 : 32;
    // streamed kernel: 128-wide N tiles, 16-byte aligned operands, whole granules at every edge
    // (measured: it wins from K = 256 up; at K = 128 a tile is only 4 stages long and the v1 kernel's
    // higher occupancy hides the per-tile epilogue better)
    pl->stream = (c <= 1) && d->K >= 256 && (d->K & 3) == 0 && al16(d->A) && al16(d->B) && m4(d->lda) &&
                 m4(d->ldb) && m4(d->a_bs0) && m4(d->a_bs1) && m4(d->b_bs0) && m4(d->b_bs1) &&
                 (d->layout_a == 0 || (d->M & 3) == 0) && (d->layout_b == 0 || (d->N & 3) == 0);
    if (const char* e = getenv("GT_GEMM_STREAM")) pl->stream = pl->stream && atoi(e) != 0;
    if (d->ep_mode != GT_EP_NORMAL) { pl->stream = 0; pl->bk = 16; }
    if (d->K2 > 0) pl->stream = 0;
    if (pl->stream) pl->bk = 32;
    if (pl->x3) { pl->stream = 0; pl->bk = 16; }
    pl->bm = kCfgs[c].wm * 16 * kCfgs[c].mt;
    pl->bn = kCfgs[c].wn * 16 * kCfgs[c].nt;
    pl->tiles_m = ceil_div(d->M, pl->bm);
    pl->tiles_n = ceil_div(d->N, pl->bn);
    const int64_t blocks = (int64_t)pl->tiles_m * pl->tiles_n * batch;
    int split = d->split_k;
    if (split == 0) {
        split = 1;
        // aim at ~2 resident blocks per CU (two waves per SIMD hide each other's barriers and loads)
        int target = 512;
        if (const char* e = getenv("GT_GEMM_TARGET")) target = std::max(1, atoi(e));
        if (!has_epilogue(d) && blocks < 384 && d->K >= 512) {
            split = (int)std::min<int64_t>((target + blocks / 2) / blocks, d->K / (4 * pl->bk));
            if (split < 1) split = 1;
            // token-contracted products on the split engine hand whole groups of K chunks to the eight XCDs (gemm_x3w_kernel's
            // chunk-major grid): a multiple of eight chunks keeps every XCD at or under its 64 resident blocks -- 171 chunks x
            // 3 tiles put 66 on seven of them and the launch paid a second round (178 us against 117)
            if (pl->x3 && d->layout_a == 1 && d->layout_b == 1 && split >= 16) split &= ~7;
        }
    }
    if (d->cv_c > 0 && d->cv_wgrad)                   // nine taps per chunk: ~3 resident blocks per CU, whole XCD groups
        split = (int)std::max<int64_t>(1, std::min<int64_t>((768 + blocks - 1) / blocks, d->K / (4 * pl->bk)));
    if (split > 1 && has_epilogue(d)) return GT_ENOTSUP;
    if (split > 1024) split = 1024;
    int chunk = ceil_div(std::max(d->K, 1), split);
    chunk = ((chunk + pl->bk - 1) / pl->bk) * pl->bk;
    split = std::max(1, ceil_div(std::max(d->K, 1), chunk));
    pl->split = split;
    pl->k_chunk = chunk;
    return 0;
}


}  // namespace gt

using namespace gt;

// GT_EP_HEADNORM with 48-wide heads (ex3: d_model 192, 4 heads): the fused epilogue needs every head inside one wave's 64
// output columns, so the product runs with each head in a 64-column SLOT -- N' = 3 h 64 tile columns, the 16 columns behind a
// head are zero rows of the PACKED weight (the pack kernels map slot rows to weight rows; the weight itself is read in place),
// and the epilogue normalises / stores the 48 real columns.  The caller's descriptor keeps N = 3 h 48; every entry point
// plans with this copy.  (A third more MFMA work in a launch that is bound by its memory traffic; no extra HBM byte.)
static const gt_gemm_desc* hn_slots(const gt_gemm_desc* d, gt_gemm_desc* tmp) {
    if (!d || d->ep_mode != GT_EP_HEADNORM || d->hn_dk != 48 || d->hn_h <= 0 || d->N != 3 * d->hn_h * 48) return d;
    *tmp = *d;
    tmp->N = 3 * d->hn_h * 64;
    return tmp;
}

extern "C" void gt_gemm_desc_init(gt_gemm_desc* d) {
    memset(d, 0, sizeof(*d));
    d->alpha = 1.f;
    d->out_scale = 1.f;
    d->a_drop_sign = 1.f;
    d->aux_scale = 1.f;
    d->batch0 = d->batch1 = 1;
    d->split_k = 1;
}

extern "C" int gt_gemm_plan(const gt_gemm_desc* d, int32_t* bm, int32_t* bn, int32_t* split) {
    Plan pl;
    gt_gemm_desc hs;
    d = hn_slots(d, &hs);
    int rc = make_plan(d, &pl);
    if (rc) return rc;
    if (bm) *bm = pl.bm;
    if (bn) *bn = pl.bn;
    if (split) *split = pl.split;
    return 0;
}

static int64_t slab_bytes(const gt_gemm_desc* d, const Plan& pl) {
    if (pl.split <= 1) return 0;
    return (int64_t)pl.split * d->batch0 * d->batch1 * d->M * d->N * (int64_t)sizeof(float);
}
static int64_t acs_parts(const gt_gemm_desc* d, const Plan& pl) {
    return d->a_colsum ? (int64_t)pl.split * d->batch0 * d->batch1 : 0;
}

// Symbol of the kernel instance gt_gemm would launch for `d` (as rocprofv3 prints it), for matching the
// bench's roofline line against a kernel trace.
extern "C" int gt_gemm_kernel_name(const gt_gemm_desc* d, char* buf, int32_t n) {
    Plan pl;
    if (!d || !buf || n <= 0) return GT_EINVAL;
    gt_gemm_desc hs;
    d = hn_slots(d, &hs);
    if (!getenv("GT_GEMM_NO_TSMM") && tsmm_eligible(d)) {
        snprintf(buf, n, "%s", tsmm_kernel_name(d));
        return 0;
    }
    int rc = make_plan(d, &pl);
    if (rc) return rc;
    const Cfg& c = kCfgs[pl.cfg];
    if (pl.x3) {
        GemmP q;
        memset(&q, 0, sizeof(q));
        q.M = d->M; q.N = d->N; q.K = d->K; q.K2 = d->K2; q.cv_C = d->cv_c; q.cv_wgrad = d->cv_wgrad != 0;
        if (x3_packed_ok(d, pl.x3, pl.split)) { q.Bp = d->B; q.bp_f16 = d->precision == GT_PREC_F16X2; }
        q.wg_f16 = x3w_ok(d, pl.split);
        q.a_vec = al16(d->A) && m4(d->lda) && m4(d->a_bs0) && m4(d->a_bs1);
        q.b_vec = al16(d->B) && m4(d->ldb) && m4(d->b_bs0) && m4(d->b_bs1);
        snprintf(buf, n, "%s", x3_kernel_name(q, d->layout_a, d->layout_b, pl.x3,
                                              d->ep_mode == GT_EP_HEADNORM ? hn_slot_width(d->hn_dk) : 0));
    }
    else if (pl.stream)
        snprintf(buf, n, "void gt::gemm_stream_kernel<%d, %d, %d>(gt::GemmP)", d->layout_a, d->layout_b, c.mt);
    else
        snprintf(buf, n, "void gt::gemm_kernel<%d, %d, %d, %d, %d, %d, %d, %d>(gt::GemmP)", d->layout_a, d->layout_b,
                 c.mt, c.nt, c.wm, c.wn, pl.bk,
                 d->ep_mode == GT_EP_NORMAL ? 0 : (d->n_out == 1 ? 1 : 4));
    return 0;
}

static bool width_split(const gt_gemm_desc* d, gt_gemm_desc* a, gt_gemm_desc* b);
static int64_t ws_bytes_one(const gt_gemm_desc* d);

extern "C" int64_t gt_gemm_ws_bytes(const gt_gemm_desc* d) {
    gt_gemm_desc a, b, hs;
    d = hn_slots(d, &hs);
    if (d && width_split(d, &a, &b)) return std::max(ws_bytes_one(&a), ws_bytes_one(&b));     // (b_packed is inherited by both)
    return ws_bytes_one(d);
}

static int64_t ws_bytes_one(const gt_gemm_desc* d) {
    Plan pl;
    if (d && !getenv("GT_GEMM_NO_TSMM") && tsmm_eligible(d)) return tsmm_ws_bytes(d);
    if (make_plan(d, &pl)) return 0;
    if (pl.x3 && x3_packed_ok(d, pl.x3, pl.split)) return d->b_packed ? 0 : x3_packed_bytes(d);
    const int64_t parts = acs_parts(d, pl);
    if (d->ep_mode == GT_EP_MLP_BWD)
        return (int64_t)pl.tiles_m * kCfgs[pl.cfg].wm * d->n_out * d->N * (int64_t)sizeof(float);
    return slab_bytes(d, pl) + (parts > 0 ? parts * d->M * (int64_t)sizeof(float) : 0);
}

// One kernel launch (+ split-K reduce) for the column range [n_off, n_off + d->N) of a problem whose
// full width is drop_ld (d already points at that column range).
static int gemm_one(const gt_gemm_desc* d, int drop_ld, int n_off, void* ws, int64_t ws_bytes, void* stream) {
    if (!d || !d->A || !d->B) return GT_EINVAL;
    if (!d->C && d->ep_mode != GT_EP_ROWDOT && !(d->ep_mode == GT_EP_HEADNORM && (d->hn_skip_raw_mask & 7) == 7))
        return GT_EINVAL;
    if (d->ep_mode == GT_EP_HEADNORM) {
        if (d->layout_a || d->layout_b || d->batch0 * d->batch1 != 1 || d->K2 > 0 || d->split_k > 1) return GT_ENOTSUP;
        if (d->hn_h <= 0 || d->hn_p < 0 || d->N != 3 * d->hn_h * hn_slot_width(d->hn_dk) || (d->hn_norm_mask & ~7)) return GT_EINVAL;
        if (!d->hn_out || (d->hn_p > 0 && !d->hn_pos)) return GT_EINVAL;
        if (d->hn_norm_mask && (!d->hn_gamma || !d->hn_beta || !d->hn_stats)) return GT_EINVAL;
        if (d->rp || d->add || d->pre || d->act || d->aux_op || d->drop.p > 0.f || d->res || d->out_scale != 1.f ||
            d->a_drop.p > 0.f || d->a_colsum)
            return GT_ENOTSUP;
    } else if (d->ep_mode != GT_EP_NORMAL) {
        if (d->ep_mode != GT_EP_ROWDOT && d->ep_mode != GT_EP_MLP_BWD) return GT_EINVAL;
        if (d->N > 128 || d->batch0 * d->batch1 != 1 || d->n_out < 1 || d->n_out > 4 || !d->w2) return GT_ENOTSUP;
        if (d->ep_mode == GT_EP_ROWDOT && !d->out2) return GT_EINVAL;
        if (d->ep_mode == GT_EP_MLP_BWD && (!d->g2 || !d->dw2)) return GT_EINVAL;
        if (d->a_colsum || d->a_drop.p > 0.f) return GT_ENOTSUP;
    }
    if ((d->layout_a | d->layout_b) & ~1) return GT_EINVAL;
    if (d->cv_c != 0) {                               // implicit 3x3 convolution on A (gt_hip.h: cv_*)
        if (d->cv_c < 0 || d->cv_h <= 0 || d->cv_w <= 0) return GT_EINVAL;
        if (d->cv_wgrad) {
            if (d->K % ((int64_t)d->cv_h * d->cv_w) != 0 || d->N != d->cv_c || d->batch0 != 9 || d->batch1 != 1)
                return GT_EINVAL;
            if (d->layout_a != 1 || d->layout_b != 1 || d->split_k != 0 || d->cv_w < 16 || has_epilogue(d) ||
                d->a_drop.p > 0.f || d->a_colsum)
                return GT_ENOTSUP;
        } else {
            if (d->K != 9 * d->cv_c || d->M % ((int64_t)d->cv_h * d->cv_w) != 0) return GT_EINVAL;
            if (d->layout_a || d->layout_b || (d->cv_c & 15) || d->batch0 * d->batch1 != 1 || d->split_k != 1 ||
                d->a_drop.p > 0.f || d->a_colsum || d->K2 > 0 || d->ep_mode != GT_EP_NORMAL)
                return GT_ENOTSUP;
        }
    }
    if (d->C && !getenv("GT_GEMM_NO_TSMM") && tsmm_eligible(d)) return d->b_packed ? GT_EINVAL : tsmm_run(d, ws, ws_bytes, stream);
    if (d->rp < 0 || d->rp > 8) return GT_EINVAL;
    if ((d->a_drop.p > 0.f && !d->a_drop.seed) || (d->drop.p > 0.f && !d->drop.seed)) return GT_EINVAL;
    if (d->a_drop.p >= 1.f || d->drop.p >= 1.f || d->a_drop.p < 0.f || d->drop.p < 0.f) return GT_EINVAL;
    Plan pl;
    int rc = make_plan(d, &pl);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t batch = (int64_t)d->batch0 * d->batch1;

    GemmP p;
    memset(&p, 0, sizeof(p));
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.batch1 = d->batch1; p.k_chunk = pl.k_chunk;
    p.n_split = pl.split; p.n_batch = (int)batch;
    const int64_t n_work64 = (int64_t)pl.tiles_m * pl.tiles_n * pl.split * batch;
    if (n_work64 > (1 << 30)) return GT_EINVAL;
    p.n_work = (int)n_work64;
    p.A = d->A; p.lda = d->lda; p.a_bs0 = d->a_bs0; p.a_bs1 = d->a_bs1;
    p.B = d->B; p.ldb = d->ldb; p.b_bs0 = d->b_bs0; p.b_bs1 = d->b_bs1;
    p.a_vec = al16(d->A) && m4(d->lda) && m4(d->a_bs0) && m4(d->a_bs1);
    p.b_vec = al16(d->B) && m4(d->ldb) && m4(d->b_bs0) && m4(d->b_bs1);
    p.a_drop = make_drop(&d->a_drop, d->a_drop_sign);
    p.a_drop_ld = d->a_drop_ld; p.a_drop_bstride = d->a_drop_bstride;
    if (d->cv_c > 0) {
        if (!pl.x3 || (!d->cv_wgrad && pl.split != 1)) return GT_ENOTSUP;
        p.cv_H = d->cv_h; p.cv_W = d->cv_w; p.cv_C = d->cv_c; p.cv_wgrad = d->cv_wgrad != 0;
        // lda (forward / data gradient) / ldb (weight gradient) = the pixel pitch of the image: >= cv_c, so a convolution can
        // read a channel slice of a wider channels-last buffer in place (values below cv_c mean "dense")
        if (d->cv_wgrad) {
            p.ldb = d->ldb > d->cv_c ? d->ldb : d->cv_c; p.b_bs0 = p.b_bs1 = 0;
            p.b_vec = al16(d->B) && m4(p.ldb);
        } else {
            p.lda = d->lda > d->cv_c ? d->lda : d->cv_c;
            p.a_vec = al16(d->A) && m4(p.lda);
        }
    }

    if (d->K2 > 0) {
        if (!d->A2 || !d->B2 || d->a_drop.p > 0.f || d->a_colsum || pl.split != 1) return GT_ENOTSUP;
        p.K2 = d->K2; p.A2 = d->A2; p.lda2 = d->lda2; p.a2_bs0 = d->a2_bs0; p.a2_bs1 = d->a2_bs1;
        p.B2 = d->B2; p.ldb2 = d->ldb2; p.b2_bs0 = d->b2_bs0; p.b2_bs1 = d->b2_bs1;
        p.a2_vec = al16(d->A2) && m4(d->lda2) && m4(d->a2_bs0) && m4(d->a2_bs1);
        p.b2_vec = al16(d->B2) && m4(d->ldb2) && m4(d->b2_bs0) && m4(d->b2_bs1);
    }
    const int64_t mn = (int64_t)d->M * d->N;
    float* dw2_partial = nullptr;
    const int dw2_slabs = pl.tiles_m * kCfgs[pl.cfg].wm;
    if (d->ep_mode == GT_EP_HEADNORM) {
        if (!pl.x3 || pl.split != 1) return GT_ENOTSUP;
        p.ep_mode = d->ep_mode;
        p.hn_gamma = d->hn_gamma; p.hn_beta = d->hn_beta; p.hn_pos = d->hn_pos; p.hn_out = d->hn_out;
        p.hn_stats = d->hn_stats; p.hn_h = d->hn_h; p.hn_dk = hn_slot_width(d->hn_dk); p.hn_dkr = d->hn_dk; p.hn_p = d->hn_p;
        p.hn_DP = (d->hn_dk + d->hn_p + 3) & ~3; p.hn_mask = d->hn_norm_mask; p.hn_eps = d->hn_eps;
        p.hn_skip_raw = d->hn_skip_raw_mask & 7; p.hn_plain = d->hn_plain != 0;
    } else if (d->ep_mode != GT_EP_NORMAL) {
        if (pl.tiles_n != 1 || pl.split != 1) return GT_ENOTSUP;
        p.ep_mode = d->ep_mode; p.n_out = d->n_out; p.w2 = d->w2; p.ldw2 = d->ldw2; p.b2 = d->b2;
        p.out2 = d->out2; p.g2 = d->g2;
        if (d->ep_mode == GT_EP_MLP_BWD) {
            const int64_t need = (int64_t)dw2_slabs * d->n_out * d->N * (int64_t)sizeof(float);
            if (!ws || ws_bytes < need) return GT_EWS;
            dw2_partial = reinterpret_cast<float*>(ws);
            p.dw2_partial = dw2_partial;
        }
    }
    const int64_t parts = acs_parts(d, pl);
    float* acs_partial = nullptr;
    if (d->a_colsum) {
        if (d->layout_a != 1) return GT_ENOTSUP;
        // without a mask the sign of a_drop_sign is not applied by the loader: fold it into the reduce
        const bool need_scale = !(d->a_drop.p > 0.f) && d->a_drop_sign != 1.f;
        if (parts > 1 || need_scale) {
            const int64_t off = slab_bytes(d, pl);
            if (!ws || ws_bytes < off + parts * d->M * (int64_t)sizeof(float)) return GT_EWS;
            acs_partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + off);
            p.acs = acs_partial;
        } else {
            p.acs = d->a_colsum;
        }
    }
    if (pl.split > 1) {
        const int64_t need = (int64_t)pl.split * batch * mn * (int64_t)sizeof(float);
        if (!ws || ws_bytes < need) return GT_EWS;
        p.C = reinterpret_cast<float*>(ws);
        p.ldc = d->N; p.c_bs0 = (int64_t)d->batch1 * mn; p.c_bs1 = mn; p.c_split = batch * mn;
        p.c_vec = al16(ws) && m4(d->N) && m4(mn);
        p.raw = 1;
        p.alpha = 1.f; p.out_scale = 1.f;
    } else {
        p.C = d->C; p.ldc = d->ldc; p.c_bs0 = d->c_bs0; p.c_bs1 = d->c_bs1; p.c_split = 0;
        p.c_vec = al16(d->C) && m4(d->ldc) && m4(d->c_bs0) && m4(d->c_bs1);
        if (d->res) p.c_vec = p.c_vec && al16(d->res) && m4(d->ldr) && m4(d->r_bs0) && m4(d->r_bs1);
        if (d->add) p.c_vec = p.c_vec && al16(d->add) && m4(d->ldadd) && m4(d->add_bs0) && m4(d->add_bs1);
        if (d->aux_op) p.c_vec = p.c_vec && al16(d->aux) && m4(d->ldaux) && m4(d->aux_bs0) && m4(d->aux_bs1);
        if (d->pre) p.c_vec = p.c_vec && al16(d->pre) && m4(d->ldpre) && m4((int64_t)d->M * d->ldpre);
        p.alpha = d->alpha; p.bias = d->bias;
        p.rp = d->rp; p.rp_a = d->rp_a; p.rp_lda = d->rp_lda; p.rp_a_bs0 = d->rp_a_bs0;
        p.rp_b = d->rp_b; p.rp_ldb = d->rp_ldb;
        if (p.rp && (!p.rp_a || !p.rp_b)) return GT_EINVAL;
        p.add = d->add; p.ldadd = d->ldadd; p.add_bs0 = d->add_bs0; p.add_bs1 = d->add_bs1;
        p.pre = d->pre; p.ldpre = d->ldpre;
        p.act = d->act; p.aux_op = d->aux_op; p.aux = d->aux; p.ldaux = d->ldaux;
        p.aux_bs0 = d->aux_bs0; p.aux_bs1 = d->aux_bs1; p.aux_scale = d->aux_scale;
        if (p.aux_op && !p.aux) return GT_EINVAL;
        p.drop = make_drop(&d->drop);
        p.drop_ld = drop_ld; p.n_off = n_off;
        p.res = d->res; p.ldr = d->ldr; p.r_bs0 = d->r_bs0; p.r_bs1 = d->r_bs1;
        p.out_scale = d->out_scale;
        if (d->c_masked) {
            if (batch != 1 || d->ep_mode != GT_EP_NORMAL || d->K2 > 0) return GT_ENOTSUP;
            if (d->c_mask.p < 0.f || d->c_mask.p >= 1.f || (d->c_mask.p > 0.f && !d->c_mask.seed)) return GT_EINVAL;
            p.c2 = d->c_masked; p.ldc2 = d->ldc_masked; p.drop2 = make_drop(&d->c_mask);
            p.c_vec = p.c_vec && al16(d->c_masked) && m4(d->ldc_masked);
        }
    }
    if (d->c_masked && pl.split > 1) return GT_ENOTSUP;

    dim3 grid((unsigned)(pl.tiles_m * pl.tiles_n), (unsigned)pl.split, (unsigned)batch);
    const int lay = d->layout_a * 2 + d->layout_b;
    if (d->b_packed && !(pl.x3 && x3_packed_ok(d, pl.x3, pl.split))) return GT_EINVAL;   // not a packed-B launch: see gt_hip.h
    if (pl.x3) {
        if (x3_packed_ok(d, pl.x3, pl.split)) {
            int rcp = x3_pack_b(d, p, ws, ws_bytes, st);
            if (rcp) return rcp;
        }
        p.wg_f16 = x3w_ok(d, pl.split);
        int rcx = x3_launch(p, d->layout_a, d->layout_b, pl.x3, (unsigned)(pl.tiles_m * pl.tiles_n), (unsigned)pl.split,
                            (unsigned)batch, st);
        if (rcx) return rcx;
    } else if (d->ep_mode != GT_EP_NORMAL) {
        if (lay != 0) return GT_ENOTSUP;
        launch_head(pl.cfg, grid, st, p);
    } else if (pl.stream) {
        if (pl.cfg == 0) {
            if (lay == 0) launch_stream<0, 0, 4>(st, p);
            else if (lay == 1) launch_stream<0, 1, 4>(st, p);
            else if (lay == 2) launch_stream<1, 0, 4>(st, p);
            else launch_stream<1, 1, 4>(st, p);
        } else {
            if (lay == 0) launch_stream<0, 0, 2>(st, p);
            else if (lay == 1) launch_stream<0, 1, 2>(st, p);
            else if (lay == 2) launch_stream<1, 0, 2>(st, p);
            else launch_stream<1, 1, 2>(st, p);
        }
    } else if (pl.bk == 32) {
        if (lay == 0) launch_cfg<0, 0, 32>(pl.cfg, grid, st, p);
        else if (lay == 1) launch_cfg<0, 1, 32>(pl.cfg, grid, st, p);
        else if (lay == 2) launch_cfg<1, 0, 32>(pl.cfg, grid, st, p);
        else launch_cfg<1, 1, 32>(pl.cfg, grid, st, p);
    } else {
        if (lay == 0) launch_cfg<0, 0, 16>(pl.cfg, grid, st, p);
        else if (lay == 1) launch_cfg<0, 1, 16>(pl.cfg, grid, st, p);
        else if (lay == 2) launch_cfg<1, 0, 16>(pl.cfg, grid, st, p);
        else launch_cfg<1, 1, 16>(pl.cfg, grid, st, p);
    }
    GT_LAUNCH_CHECK();
    if (dw2_partial) {
        const int64_t n2 = (int64_t)d->n_out * d->N;
        int rc3 = gt_slab_reduce(dw2_partial, n2, dw2_slabs, n2, 1.f, d->dw2, stream);
        if (rc3) return rc3;
    }
    if (acs_partial) {
        const float sc = (d->a_drop.p > 0.f) ? 1.f : d->a_drop_sign;
        int rc2 = gt_slab_reduce(acs_partial, d->M, (int)parts, d->M, sc, d->a_colsum, stream);
        if (rc2) return rc2;
    }

    if (pl.split > 1) {
        const int64_t total = batch * mn;
        const bool v4 = m4(d->N) && m4(d->ldc) && m4(d->c_bs0) && m4(d->c_bs1) && al16(d->C) && al16(ws);
        if (v4) {
            const int64_t total4 = total / 4;
            const int blocks4 = (int)((total4 + 31) / 32);
            hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(blocks4), dim3(256), 0, st,
                               reinterpret_cast<const float*>(ws), pl.split, batch * mn, d->M, d->N / 4,
                               d->batch1, d->alpha, d->C, d->ldc, d->c_bs0, d->c_bs1, total4);
            GT_LAUNCH_CHECK();
            return 0;
        }
        const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st,
                           reinterpret_cast<const float*>(ws), pl.split, batch * mn, d->M, d->N,
                           d->batch1, d->alpha, d->C, d->ldc, d->c_bs0, d->c_bs1, total);
        GT_LAUNCH_CHECK();
    }
    return 0;
}

// Splits d into the 128-aligned column range `a` and the remainder `b` when that pays (see gt_gemm); false otherwise.
static bool width_split(const gt_gemm_desc* d, gt_gemm_desc* a, gt_gemm_desc* b) {
    const int rem = d->N % 128;
    Plan pl;
    if (!(d->N > 128 && rem != 0 && rem <= 64 && d->ep_mode == GT_EP_NORMAL && d->cv_c == 0 && make_plan(d, &pl) == 0 &&
          pl.bn == 128))
        return false;
    // only when the aligned part alone fills the chip: two half-empty launches would serialise instead.  One launch
    // per tile (split == 1): >= 512 tiles (measured);  K-split products (token-contracted, no epilogue): the K
    // slices of the aligned part fill it
    const int64_t main_tiles = (int64_t)pl.tiles_m * (d->N / 128) * d->batch0 * d->batch1;
    if (pl.split == 1 ? main_tiles < 512 : (main_tiles * pl.split < 256 || d->K2 > 0)) return false;
    const int n_main = d->N - rem;
    *a = *d;
    *b = *d;
    a->N = n_main;
    b->N = rem;
    b->B = d->B + (d->layout_b == 0 ? (int64_t)n_main * d->ldb : (int64_t)n_main);
    b->C = d->C + n_main;
    if (d->K2 > 0) b->B2 = d->B2 + (d->layout_b == 0 ? (int64_t)n_main * d->ldb2 : (int64_t)n_main);
    if (d->bias) b->bias = d->bias + n_main;
    if (d->rp) b->rp_b = d->rp_b + (int64_t)n_main * d->rp_ldb;
    if (d->add) b->add = d->add + n_main;
    if (d->pre) b->pre = d->pre + n_main;
    if (d->aux) b->aux = d->aux + n_main;
    if (d->res) b->res = d->res + n_main;
    if (d->c_masked) b->c_masked = d->c_masked + n_main;
    if (pl.split == 1) a->split_k = b->split_k = 1;      // else: each part plans its own K slices (split_k as given)
    b->a_colsum = nullptr;                               // the row sums of A ride on the first launch only
    return true;
}

// Widths just above a multiple of 128 (the merged-head width h*(d_k+p) = 144 of the Darcy model) would
// waste most of a second 128-wide tile column: run the aligned part and the remainder as two launches,
// the remainder on a narrow-tile configuration.
// > 0: gt_gemm(d) is ONE launch of the packed-B kernels and this is the size of its packed weight (the buffer a caller that
// packs ahead -- gt_gemm_pack_b_many -- hands over in d->b_packed); 0: some other path, b_packed must stay NULL
static int64_t packed_bytes_one(const gt_gemm_desc* d) {
    Plan pl;
    if (!getenv("GT_GEMM_NO_TSMM") && tsmm_eligible(d)) return 0;
    if (make_plan(d, &pl) || !pl.x3 || !x3_packed_ok(d, pl.x3, pl.split)) return 0;
    return x3_packed_bytes(d);
}

// Round 6: a width-split product (N = 192 = 128 + 64, ex3's d_model) is TWO packed-B launches when both column ranges take
// the packed kernels; its packed weight is the two packs back to back (the aligned range first).
extern "C" int64_t gt_gemm_packed_b_bytes(const gt_gemm_desc* d) {
    if (!d) return 0;
    gt_gemm_desc a, b, hs;
    d = hn_slots(d, &hs);
    if (width_split(d, &a, &b)) {
        a.b_packed = b.b_packed = nullptr;
        const int64_t pa = packed_bytes_one(&a), pb = packed_bytes_one(&b);
        return (pa > 0 && pb > 0) ? pa + pb : 0;
    }
    return packed_bytes_one(d);
}

extern "C" int gt_gemm_pack_b_many(const gt_gemm_desc* descs, void* const* outs, int32_t n, void* stream) {
    if (!descs || !outs || n < 0 || n > 64) return GT_EINVAL;
    gt_gemm_desc parts[128];
    void* pouts[128];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (gt_gemm_packed_b_bytes(&descs[i]) <= 0) return GT_ENOTSUP;
        gt_gemm_desc a, b, hs;
        const gt_gemm_desc* di = hn_slots(&descs[i], &hs);
        if (di != &descs[i]) {
            parts[m] = *di; pouts[m++] = outs[i];
        } else if (width_split(&descs[i], &a, &b)) {
            a.b_packed = b.b_packed = nullptr;
            parts[m] = a; pouts[m++] = outs[i];
            parts[m] = b; pouts[m++] = reinterpret_cast<char*>(outs[i]) + packed_bytes_one(&a);
        } else {
            parts[m] = descs[i]; pouts[m++] = outs[i];
        }
    }
    for (int i0 = 0; i0 < m; i0 += 64) {
        int rc = x3_pack_b_many(parts + i0, pouts + i0, std::min(64, m - i0), (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int gt_gemm(const gt_gemm_desc* d, void* ws, int64_t ws_bytes, void* stream) {
    if (!d) return GT_EINVAL;
    gt_gemm_desc a, b, hs;
    d = hn_slots(d, &hs);
    if (width_split(d, &a, &b)) {
        // a weight packed ahead is the two column ranges' packs back to back (gt_gemm_packed_b_bytes): each range gets ITS
        // pack -- never the full-width pointer with the wrong tile geometry
        if (d->b_packed) {
            a.b_packed = b.b_packed = nullptr;
            const int64_t pa = packed_bytes_one(&a);
            if (pa <= 0 || packed_bytes_one(&b) <= 0) return GT_EINVAL;
            a.b_packed = d->b_packed;
            b.b_packed = reinterpret_cast<const char*>(d->b_packed) + pa;
        }
        int rc = gemm_one(&a, d->N, 0, ws, ws_bytes, stream);
        if (rc) return rc;
        return gemm_one(&b, d->N, a.N, ws, ws_bytes, stream);      // same stream: the scratch is free again
    }
    return gemm_one(d, d->N, 0, ws, ws_bytes, stream);
}
