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
#include "gt_common.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace gt {

struct GemmP {
    int M, N, K;
    int tiles_m, tiles_n, batch1, k_chunk, n_split, n_batch, n_work;
    const float* A; int64_t lda, a_bs0, a_bs1;
    const float* B; int64_t ldb, b_bs0, b_bs1;
    float* C; int64_t ldc, c_bs0, c_bs1, c_split;
    int a_vec, b_vec, c_vec, raw;
    DropDev a_drop; int64_t a_drop_ld, a_drop_bstride;
    float* acs;                 // per-(K-slice, batch) partial row sums of the (masked) A operand, or null
    float alpha; const float* bias;
    int rp; const float* rp_a; int64_t rp_lda, rp_a_bs0; const float* rp_b; int64_t rp_ldb;
    const float* add; int64_t ldadd, add_bs0, add_bs1;
    float* pre; int64_t ldpre;
    int act, aux_op; const float* aux; int64_t ldaux, aux_bs0, aux_bs1; float aux_scale;
    DropDev drop; int drop_ld, n_off;      // mask index = (z*M + m)*drop_ld + n_off + n
    const float* res; int64_t ldr, r_bs0, r_bs1;
    float out_scale;
    int ep_mode, n_out; const float* w2; int64_t ldw2; const float* b2; float* out2; const float* g2;
    float* dw2_partial;      // MLP_BWD: [tiles_m * WM][n_out][N]
    int K2; const float* A2; int64_t lda2, a2_bs0, a2_bs1; const float* B2; int64_t ldb2, b2_bs0, b2_bs1;
    int a2_vec, b2_vec;
    int light_wait;          // streamed kernel: counted vmcnt after full-tile epilogues (see kernel)
    // GT_EP_HEADNORM: head-norm forward fused behind the QKV projection
    const float* hn_gamma; const float* hn_beta; const float* hn_pos; float* hn_out; float* hn_stats;
    int hn_h, hn_dk, hn_p, hn_DP, hn_mask; float hn_eps;
};

// ---- global -> registers: 4 consecutive elements of the operand tile -----------------------
// L == 0: operand(x,k) = base[x*ld + k]  (k contiguous)  idx -> x = idx/(BK/4), k = 4*(idx%(BK/4))
// L == 1: operand(x,k) = base[k*ld + x]  (x contiguous)  idx -> k = idx/(BX/4), x = 4*(idx%(BX/4))
template <int L, int BX, int BK>
__device__ __forceinline__ f32x4 gload(const float* __restrict__ base, int64_t ld, int x0, int X,
                                       int k0, int kend, int idx, int vec, const DropDev& dd,
                                       uint32_t dkey, int64_t dld, int64_t dboff) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (L == 0) {
        constexpr int KQ = BK / 4;
        const int x = x0 + idx / KQ, k = k0 + ((idx % KQ) << 2);
        if (x < X && k < kend) {
            const float* ptr = base + (int64_t)x * ld + k;
            if (vec && k + 3 < kend) {
                v = *reinterpret_cast<const f32x4*>(ptr);
            } else {
                v[0] = ptr[0];
                if (k + 1 < kend) v[1] = ptr[1];
                if (k + 2 < kend) v[2] = ptr[2];
                if (k + 3 < kend) v[3] = ptr[3];
            }
            if (dd.thresh) {
                const uint32_t di = (uint32_t)(dboff + (int64_t)x * dld + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= drop_mul(dd, dkey, di + j);
            }
        }
    } else {
        constexpr int Q = BX / 4;
        const int k = k0 + idx / Q, x = x0 + ((idx % Q) << 2);
        if (k < kend && x < X) {
            const float* ptr = base + (int64_t)k * ld + x;
            if (vec && x + 3 < X) {
                v = *reinterpret_cast<const f32x4*>(ptr);
            } else {
                v[0] = ptr[0];
                if (x + 1 < X) v[1] = ptr[1];
                if (x + 2 < X) v[2] = ptr[2];
                if (x + 3 < X) v[3] = ptr[3];
            }
            if (dd.thresh) {
                const uint32_t di = (uint32_t)(dboff + (int64_t)k * dld + x);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= drop_mul(dd, dkey, di + j);
            }
        }
    }
    return v;
}

// ---- registers -> LDS image s[BK][BX], column swizzled by ((k>>2)&3)<<3 ---------------------
template <int L, int BX, int BK>
__device__ __forceinline__ void sstore(float* __restrict__ s, int idx, f32x4 v) {
    if (L == 0) {
        constexpr int KQ = BK / 4;
        const int x = idx / KQ, c = idx % KQ;
        const int col = x ^ (((c & 3) << 3) & (BX - 1));
        s[(4 * c + 0) * BX + col] = v[0];
        s[(4 * c + 1) * BX + col] = v[1];
        s[(4 * c + 2) * BX + col] = v[2];
        s[(4 * c + 3) * BX + col] = v[3];
    } else {
        constexpr int Q = BX / 4;
        const int k = idx / Q, x = (idx % Q) << 2;
        const int col = x ^ ((((k >> 2) & 3) << 3) & (BX - 1));
        *reinterpret_cast<f32x4*>(&s[k * BX + col]) = v;
    }
}

template <int NV>
__device__ __forceinline__ void lds_frag(const float* __restrict__ s, float (&f)[NV]) {
    if (NV == 4) {
        f32x4 t = *reinterpret_cast<const f32x4*>(s);
        f[0] = t[0]; f[1] = t[1]; f[2] = t[2]; f[3] = t[3];
    } else if (NV == 2) {
        f32x2 t = *reinterpret_cast<const f32x2*>(s);
        f[0] = t[0]; f[1] = t[1];
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) f[j] = s[j];
    }
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
#ifdef GT_ABL_NOMFMA      // ablation build (tools/ablate_gemm.sh): keep the operands live, skip the matrix pipe
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#endif
#ifdef GT_EMULATE_MFMA
    // Debug build: the same distributed-operand semantics with shuffles (documents the layout the
    // kernel assumes: A[row=lane&15][k=lane>>4], B[k=lane>>4][col=lane&15], D[row=4*(lane>>4)+r][col=lane&15]).
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float av = __shfl(a, ((lane >> 4) * 4 + r) + 16 * k, 64);
            float bv = __shfl(b, (lane & 15) + 16 * k, 64);
            c[r] = fmaf(av, bv, c[r]);
        }
    }
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// Fused epilogue shared by both kernels.  The calling lane holds, for each of the MT x NT 16x16
// accumulator tiles, rows  mw0 + MT*(4*kq + r) + s  (r = 0..3) and columns  nb + t.
// HN: additionally run the per-head LayerNorm of gt_headnorm_fwd on the (biased) row segments this lane group
// holds and scatter them into the head-tile layout (GT_EP_HEADNORM; NT == 4, dk/4 lanes per head segment).
template <int NT>
__device__ __forceinline__ void headnorm_scatter(const GemmP& p, const float (&v)[NT], int m, int nb) {
    const int dk = p.hn_dk, G = dk >> 2;
    const int stream = nb / (p.hn_h * dk), head = (nb / dk) % p.hn_h, dim = nb % dk;
    const bool normed = (p.hn_mask >> stream) & 1;
    float y[4] = {v[0], v[1 % NT], v[2 % NT], v[3 % NT]};
    if (normed) {
        const int ni = __popc(p.hn_mask & ((1 << stream) - 1));
        const float inv = 1.f / (float)dk;
        float sum = (y[0] + y[1]) + (y[2] + y[3]);
        for (int o = G >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mu = sum * inv;
        float c[4] = {y[0] - mu, y[1] - mu, y[2] - mu, y[3] - mu};
        float ss = (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
        for (int o = G >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float rstd = 1.f / sqrtf(ss * inv + p.hn_eps);
        const float* gm = p.hn_gamma + (ni * p.hn_h + head) * dk + dim;
        const float* bt = p.hn_beta + (ni * p.hn_h + head) * dk + dim;
#pragma unroll
        for (int t = 0; t < 4; ++t) y[t] = c[t] * rstd * gm[t] + bt[t];
        if (dim == 0)
            *reinterpret_cast<f32x2*>(p.hn_stats + (((int64_t)ni * p.M + m) * p.hn_h + head) * 2) = f32x2{mu, rstd};
    }
    float* row = p.hn_out + (((int64_t)stream * p.M + m) * p.hn_h + head) * p.hn_DP;
    float* dst = row + p.hn_p + dim;
    if ((p.hn_p & 3) == 0) *reinterpret_cast<f32x4*>(dst) = f32x4{y[0], y[1], y[2], y[3]};
    else if ((p.hn_p & 1) == 0) {
        *reinterpret_cast<f32x2*>(dst) = f32x2{y[0], y[1]};
        *reinterpret_cast<f32x2*>(dst + 2) = f32x2{y[2], y[3]};
    } else { dst[0] = y[0]; dst[1] = y[1]; dst[2] = y[2]; dst[3] = y[3]; }
    if (dim == 0)
        for (int jj = 0; jj < p.hn_p; ++jj) row[jj] = p.hn_pos[(int64_t)m * p.hn_p + jj];
    if (dim == dk - 4)
        for (int jj = p.hn_p + dk; jj < p.hn_DP; ++jj) row[jj] = 0.f;
}

template <int MT, int NT, bool HN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, const f32x4 (&acc)[MT][NT], int mw0, int nb,
                                              int z, int b0, int b1, int sidx, int kq) {
        if (nb >= p.N) return;
#ifdef GT_ABL_NOSTORE
    if (acc[0][0][0] != 12345.678f) return;
#endif
    const bool full = (nb + NT <= p.N);
    const int64_t coff = b0 * p.c_bs0 + b1 * p.c_bs1 + (int64_t)sidx * p.c_split;
    float* __restrict__ C = p.C + coff;

    float biasv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) biasv[t] = (p.bias && nb + t < p.N) ? p.bias[nb + t] : 0.f;
    const uint32_t dkey = drop_key_dev(p.drop);

#pragma unroll
    for (int s = 0; s < MT; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mw0 + MT * (4 * kq + r) + s;
            if (m >= p.M) continue;
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = acc[s][t][r];
            float* cp = C + (int64_t)m * p.ldc + nb;
            if (p.raw) {
                if (full && p.c_vec && NT == 4) {
                    *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (nb + t < p.N) cp[t] = v[t];
                }
                continue;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = p.alpha * v[t] + biasv[t];
            if (p.rp) {
                const float* ra_ = p.rp_a + b0 * p.rp_a_bs0 + (int64_t)m * p.rp_lda;
                for (int j = 0; j < p.rp; ++j) {
                    const float aj = ra_[j];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (nb + t < p.N) v[t] += aj * p.rp_b[(int64_t)(nb + t) * p.rp_ldb + j];
                }
            }
            const bool vec4 = full && p.c_vec && NT == 4;
            // tile-row accessors: one 16-byte access when the row segment is aligned, scalars otherwise
            auto ldrow = [&](const float* src, float (&o)[NT]) {
                if (vec4) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                    for (int t = 0; t < NT; ++t) o[t] = t4[t & 3];
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) o[t] = (nb + t < p.N) ? src[t] : 0.f;
                }
            };
            if (p.add) {
                float ad[NT];
                ldrow(p.add + b0 * p.add_bs0 + b1 * p.add_bs1 + (int64_t)m * p.ldadd + nb, ad);
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] += ad[t];
            }
            if (p.pre) {
                float* pp = p.pre + ((int64_t)z * p.M + m) * p.ldpre + nb;
                if (vec4) {
                    *reinterpret_cast<f32x4*>(pp) = f32x4{v[0], v[1 % NT], v[2 % NT], v[3 % NT]};
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (nb + t < p.N) pp[t] = v[t];
                }
            }
            if (p.act == GT_ACT_RELU) {
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] = fmaxf(v[t], 0.f);
            } else if (p.act == GT_ACT_SILU) {
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] = silu_f(v[t]);
            }
            if (p.aux_op) {
                float ax[NT];
                ldrow(p.aux + b0 * p.aux_bs0 + b1 * p.aux_bs1 + (int64_t)m * p.ldaux + nb, ax);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float a = ax[t];
                    v[t] *= (p.aux_op == GT_AUX_GT0)   ? (a > 0.f ? p.aux_scale : 0.f)
                            : (p.aux_op == GT_AUX_DSILU) ? dsilu_f(a)
                                                         : a * p.aux_scale;
                }
            }
            if (p.drop.thresh) {
                const uint32_t di = (uint32_t)(((int64_t)z * p.M + m) * p.drop_ld + p.n_off + nb);
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] *= drop_mul(p.drop, dkey, di + t);
            }
            if (p.res) {
                float rv[NT];
                ldrow(p.res + b0 * p.r_bs0 + b1 * p.r_bs1 + (int64_t)m * p.ldr + nb, rv);
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] = rv[t] + p.out_scale * v[t];
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] *= p.out_scale;
            }
            if (full && p.c_vec && NT == 4) {
                *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
            } else if (full && p.c_vec && NT == 2) {
                *reinterpret_cast<f32x2*>(cp) = f32x2{v[0], v[1]};
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (nb + t < p.N) cp[t] = v[t];
            }
            if (HN) headnorm_scatter<NT>(p, v, m, nb);     // v = alpha*acc + bias (no other epilogue field is set)
        }
    }
}

// Epilogues of the fused two-layer pointwise head (see gt_gemm_desc.ep_mode).  Called by every thread of
// the block after the K loop (smem is free then); N <= BN, so the block owns complete rows.
template <int MT, int NT, int WM, int WN, int NO>
__device__ __forceinline__ void head_epilogue(const GemmP& p, const f32x4 (&acc)[MT][NT], float* smem, int m0,
                                              int wm, int wn, int li, int kq, int tile_m) {
    constexpr int BM = WM * 16 * MT;
    const int nb = wn * 16 * NT + NT * li;
    const int tid = threadIdx.x;
    float biasv[NT], w2v[NO][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        biasv[t] = (p.bias && nb + t < p.N) ? p.bias[nb + t] : 0.f;
#pragma unroll
        for (int o = 0; o < NO; ++o) w2v[o][t] = (o < p.n_out && nb + t < p.N) ? p.w2[(int64_t)o * p.ldw2 + nb + t] : 0.f;
    }
    if (p.ep_mode == GT_EP_ROWDOT) {
        float* part = smem;                                    // [WN][BM][4]
#pragma unroll
        for (int s = 0; s < MT; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ml = wm * 16 * MT + MT * (4 * kq + r) + s;
                float d[NO];
#pragma unroll
                for (int o = 0; o < NO; ++o) d[o] = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float v = p.alpha * acc[s][t][r] + biasv[t];
                    v = (p.act == GT_ACT_RELU) ? fmaxf(v, 0.f) : (p.act == GT_ACT_SILU ? silu_f(v) : v);
#pragma unroll
                    for (int o = 0; o < NO; ++o) d[o] = fmaf(v, w2v[o][t], d[o]);
                }
#pragma unroll
                for (int o = 0; o < NO; ++o) {                 // sum over the 16 column lanes of this row
                    float x = d[o];
                    x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64);
                    x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
                    d[o] = x;
                }
                if (li == 0) {
#pragma unroll
                    for (int o = 0; o < NO; ++o) part[(wn * BM + ml) * 4 + o] = d[o];
                }
            }
        __syncthreads();
        for (int e = tid; e < BM * p.n_out; e += blockDim.x) {
            const int ml = e / p.n_out, o = e % p.n_out, m = m0 + ml;
            if (m < p.M) {
                float x = p.b2 ? p.b2[o] : 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) x += part[(w * BM + ml) * 4 + o];
                p.out2[(int64_t)m * p.n_out + o] = x;
            }
        }
    } else {                                                   // GT_EP_MLP_BWD
        float cs[NO][NT];
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int t = 0; t < NT; ++t) cs[o][t] = 0.f;
#pragma unroll
        for (int s = 0; s < MT; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * MT + MT * (4 * kq + r) + s;
                const bool ok = m < p.M;
                float g[NO];
#pragma unroll
                for (int o = 0; o < NO; ++o) g[o] = (ok && o < p.n_out) ? p.g2[(int64_t)m * p.n_out + o] : 0.f;
                float outv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float h = p.alpha * acc[s][t][r] + biasv[t];
                    float gw = 0.f;
#pragma unroll
                    for (int o = 0; o < NO; ++o) gw = fmaf(g[o], w2v[o][t], gw);
                    float a, da;
                    if (p.act == GT_ACT_SILU) silu_both(h, a, da);
                    else if (p.act == GT_ACT_RELU) { a = fmaxf(h, 0.f); da = h > 0.f ? 1.f : 0.f; }
                    else { a = h; da = 1.f; }
                    outv[t] = gw * da;
#pragma unroll
                    for (int o = 0; o < NO; ++o) cs[o][t] = fmaf(g[o], a, cs[o][t]);
                }
                if (ok && nb < p.N) {
                    float* cp = p.C + (int64_t)m * p.ldc + nb;
                    if (NT == 4 && p.c_vec && nb + 4 <= p.N) *reinterpret_cast<f32x4*>(cp) = f32x4{outv[0], outv[1 % NT], outv[2 % NT], outv[3 % NT]};
                    else {
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (nb + t < p.N) cp[t] = outv[t];
                    }
                }
            }
        // dw2 partial of this wave's 16*MT rows: combine the 4 row lanes (kq), lanes kq == 0 store
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float x = cs[o][t];
                x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
                if (kq == 0 && o < p.n_out && nb + t < p.N)
                    p.dw2_partial[(((int64_t)tile_m * WM + wm) * p.n_out + o) * p.N + nb + t] = x;
            }
    }
}

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
    gemm_epilogue<MT, NT, (HEAD < 0)>(p, acc, m0 + wm * 16 * MT, n0 + wn * 16 * NT + NT * li, z, b0, b1,
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

// =================================================================================================
// Weight-stationary kernel for the K = 128 token GEMMs (QKV, FFN1 and the FFN2 input gradient) -- STAGED behind
// GT_STAGED=wsgemm, not yet run on hardware.  The v1 kernel spends a K = 128 tile as load -> 8 short stages ->
// store with every co-resident block in the same phase (49-55 % of the fp32 MFMA peak).  Here a persistent
// block keeps its 64-column slice of B (64 x 128 floats, 32 KB) in LDS for its whole life and only streams
// 32-row A tiles (16 KB, double-buffered, direct global->LDS): one barrier and no B traffic per tile, the
// epilogue's stores drain under the next tile's MFMAs (counted vmcnt), the next A tile is already in flight.
//
// Both LDS images are [row][128] in 16-byte granules with granule g of row r stored at slot g ^ (r & 7); a lane
// (x = row or column, kq) reads the granules 4*g8 + kq, g8 = 0..7, as one ds_read_b128 each and uses component c
// as the operand of k-step (g8, c), i.e. k = 16 g8 + 4 kq + c on both operands: 8 MFMAs per 3 LDS reads.
// Row / column assignment follows gemm_epilogue's convention (MT = 1, NT = 2), so every fused epilogue works.
// XCD-aware persistent grid: the N/64 slice blocks that share an A row panel get the same block id mod 8.
constexpr int WS_BM = 32, WS_BN = 64, WS_K = 128;
template <int LB>
__global__ __launch_bounds__(256, 2) void gemm_ws_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) float sB[WS_BN * WS_K];
    __shared__ __attribute__((aligned(16))) float sA[2][WS_BM * WS_K];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // block -> (slice, y group): blocks L, L+8, L+16, ... of one XCD walk the slices of one y group first
    const int slices = p.N / WS_BN, per_xcd = (int)gridDim.x / (8 * slices);
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int slice = q % slices, ygrp = (q / slices) * 8 + xcd, ygroups = per_xcd * 8;
    const int n0 = slice * WS_BN;
    const int mtiles = (p.M + WS_BM - 1) / WS_BM;

    auto issue = [&](int tile, int buf) {                 // A tile: 16 chunks of 1 KiB, 4 per wave
        const int m0 = tile * WS_BM;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = wave * 4 + i, e = ch * 64 + lane;
            const int r = e >> 5, g = (e & 31) ^ (r & 7);
            const float* src = (m0 + r < p.M) ? p.A + (int64_t)(m0 + r) * p.lda + 4 * g : gt_zero16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(&sA[buf][ch * 256]), 16, 0, 0);
        }
    };
    int tile = ygrp;
    if (tile < mtiles) issue(tile, 0);
    // B slice, once: image [n][128], granule g of column n at slot g ^ (n & 7)
    if (LB == 0) {
        for (int e = tid; e < WS_BN * 32; e += 256) {
            const int n = e >> 5, g = (e & 31) ^ (n & 7);
            *reinterpret_cast<f32x4*>(&sB[e * 4]) = *reinterpret_cast<const f32x4*>(p.B + (int64_t)(n0 + n) * p.ldb + 4 * g);
        }
    } else {
        for (int e = tid; e < WS_BN * WS_K; e += 256) {
            const int k = e / WS_BN, n = e - k * WS_BN;    // coalesced along n
            sB[(n * 32 + ((k >> 2) ^ (n & 7))) * 4 + (k & 3)] = p.B[(int64_t)k * p.ldb + n0 + n];
        }
    }
    __syncthreads();                                       // the ds_writes of the B image are visible to every wave
    const int arow = wm * 16 + li;                         // A operand row of this lane (MT = 1)
    const int bcol = wn * 32 + 2 * li;                     // B operand columns bcol, bcol + 1 (NT = 2)
    int buf = 0, stores_prev = -1;
    for (; tile < mtiles; tile += ygroups, buf ^= 1) {
        // tile's A has landed for this wave (older than the previous tile's stores, which may still drain)
        if (stores_prev == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (tile + ygroups < mtiles) issue(tile + ygroups, buf ^ 1);
        const float* a_img = sA[buf];
        f32x4 acc[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) {
            const int G = 4 * g8 + kq;
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&a_img[(arow * 32 + (G ^ (arow & 7))) * 4]);
            const f32x4 b40 = *reinterpret_cast<const f32x4*>(&sB[(bcol * 32 + (G ^ (bcol & 7))) * 4]);
            const f32x4 b41 = *reinterpret_cast<const f32x4*>(&sB[((bcol + 1) * 32 + (G ^ ((bcol + 1) & 7))) * 4]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[0][0] = mfma16(a4[c], b40[c], acc[0][0]);
                acc[0][1] = mfma16(a4[c], b41[c], acc[0][1]);
            }
        }
        const int m0 = tile * WS_BM;
        gemm_epilogue<1, 2>(p, acc, m0 + wm * 16, n0 + wn * 32 + 2 * li, 0, 0, 0, 0, kq);
        // a full, aligned tile whose epilogue reads nothing but the bias issues exactly 4 float2 stores after the
        // prefetch (the bias loads have been consumed by then): allowing 4 outstanding operations is safe
        stores_prev = (m0 + WS_BM <= p.M && p.c_vec && !p.pre && !p.res && !p.add && !p.aux_op && !p.rp) ? 4 : 0;
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
// QKV projection + head norm (GT_EP_HEADNORM): 128-wide tile columns only (a head segment stays inside a wave)
static void launch_hn(int cfg, dim3 grid, hipStream_t st, const GemmP& p) {
    if (cfg == 0) hipLaunchKernelGGL((gemm_kernel<0, 0, 4, 4, 2, 2, 16, -1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_kernel<0, 0, 2, 4, 2, 2, 16, -1>), grid, dim3(256), 0, st, p);
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

struct Plan { int cfg, bm, bn, bk, tiles_m, tiles_n, split, k_chunk, stream, ws; };

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool m4(int64_t v) { return (v & 3) == 0; }

static bool has_epilogue(const gt_gemm_desc* d) {
    return d->bias || d->rp || d->add || d->pre || d->act || d->aux_op || d->drop.p > 0.f || d->res ||
           d->out_scale != 1.f || d->ep_mode != GT_EP_NORMAL || d->K2 > 0;
}

static bool staged(const char* what) {
    const char* e = getenv("GT_STAGED");
    return e && strstr(e, what) != nullptr;
}
static int ws_blocks(const gt_gemm_desc* d) {
    const int slices = d->N / WS_BN;
    return 8 * std::max(1, 64 / slices) * slices;
}
// weight-stationary K = 128 kernel (staged): token GEMMs with a k-contiguous A, whole 64-column slices
static bool ws_eligible(const gt_gemm_desc* d) {
    static const bool on = staged("wsgemm");
    if (!on || d->K != WS_K || d->layout_a != 0 || d->batch0 * d->batch1 != 1 || d->split_k > 1) return false;
    if (d->N % WS_BN || d->N / WS_BN > 64 || d->ep_mode != GT_EP_NORMAL || d->K2 > 0) return false;
    if (d->a_drop.p > 0.f || d->a_colsum) return false;
    if (!al16(d->A) || !m4(d->lda)) return false;
    if (d->layout_b == 0 && (!al16(d->B) || !m4(d->ldb))) return false;
    const int slices = d->N / WS_BN, ygroups = 8 * std::max(1, 64 / slices);
    return ceil_div(d->M, WS_BM) >= ygroups;
}

static int make_plan(const gt_gemm_desc* d, Plan* pl) {
    if (d->M <= 0 || d->N <= 0 || d->K < 0 || d->batch0 <= 0 || d->batch1 <= 0) return GT_EINVAL;
    const int64_t batch = (int64_t)d->batch0 * d->batch1;
    if (batch > 65535) return GT_EINVAL;
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
        if (d->ep_mode == GT_EP_HEADNORM && i > 1) continue;         // head segments must stay inside a wave tile
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
        }
    }
    if (split > 1 && has_epilogue(d)) return GT_ENOTSUP;
    if (split > 1024) split = 1024;
    int chunk = ceil_div(std::max(d->K, 1), split);
    chunk = ((chunk + pl->bk - 1) / pl->bk) * pl->bk;
    split = std::max(1, ceil_div(std::max(d->K, 1), chunk));
    pl->split = split;
    pl->k_chunk = chunk;
    pl->ws = ws_eligible(d) ? 1 : 0;
    if (pl->ws) { pl->split = 1; pl->stream = 0; pl->k_chunk = WS_K; }
    return 0;
}


}  // namespace gt

using namespace gt;

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
    if (!getenv("GT_GEMM_NO_TSMM") && tsmm_eligible(d)) {
        snprintf(buf, n, "%s", tsmm_kernel_name(d));
        return 0;
    }
    int rc = make_plan(d, &pl);
    if (rc) return rc;
    const Cfg& c = kCfgs[pl.cfg];
    if (pl.ws)
        snprintf(buf, n, "void gt::gemm_ws_kernel<%d>(gt::GemmP)", d->layout_b);
    else if (pl.stream)
        snprintf(buf, n, "void gt::gemm_stream_kernel<%d, %d, %d>(gt::GemmP)", d->layout_a, d->layout_b, c.mt);
    else
        snprintf(buf, n, "void gt::gemm_kernel<%d, %d, %d, %d, %d, %d, %d, %d>(gt::GemmP)", d->layout_a, d->layout_b,
                 c.mt, c.nt, c.wm, c.wn, pl.bk,
                 d->ep_mode == GT_EP_NORMAL ? 0 : (d->ep_mode == GT_EP_HEADNORM ? -1 : (d->n_out == 1 ? 1 : 4)));
    return 0;
}

extern "C" int64_t gt_gemm_ws_bytes(const gt_gemm_desc* d) {
    Plan pl;
    if (d && !getenv("GT_GEMM_NO_TSMM") && tsmm_eligible(d)) return tsmm_ws_bytes(d);
    if (make_plan(d, &pl)) return 0;
    const int64_t parts = acs_parts(d, pl);
    if (d->ep_mode == GT_EP_MLP_BWD)
        return (int64_t)pl.tiles_m * kCfgs[pl.cfg].wm * d->n_out * d->N * (int64_t)sizeof(float);
    return slab_bytes(d, pl) + (parts > 0 ? parts * d->M * (int64_t)sizeof(float) : 0);
}

// One kernel launch (+ split-K reduce) for the column range [n_off, n_off + d->N) of a problem whose
// full width is drop_ld (d already points at that column range).
static int gemm_one(const gt_gemm_desc* d, int drop_ld, int n_off, void* ws, int64_t ws_bytes, void* stream) {
    if (!d || !d->A || !d->B) return GT_EINVAL;
    if (!d->C && d->ep_mode != GT_EP_ROWDOT) return GT_EINVAL;
    if (d->ep_mode == GT_EP_HEADNORM) {
        if (d->layout_a || d->layout_b || d->batch0 * d->batch1 != 1 || d->K2 > 0) return GT_ENOTSUP;
        if (d->hn_dk != 16 && d->hn_dk != 32 && d->hn_dk != 64) return GT_ENOTSUP;
        if (d->hn_h <= 0 || d->hn_p < 0 || d->N != 3 * d->hn_h * d->hn_dk || (d->hn_norm_mask & ~7)) return GT_EINVAL;
        if (!d->hn_out || (d->hn_p > 0 && !d->hn_pos)) return GT_EINVAL;
        if (d->hn_norm_mask && (!d->hn_gamma || !d->hn_beta || !d->hn_stats)) return GT_EINVAL;
        if (d->rp || d->add || d->pre || d->act || d->aux_op || d->drop.p > 0.f || d->res || d->out_scale != 1.f ||
            d->a_drop.p > 0.f || d->a_colsum)
            return GT_ENOTSUP;
        if ((reinterpret_cast<uintptr_t>(d->hn_out) | reinterpret_cast<uintptr_t>(d->hn_stats)) & 15) return GT_EALIGN;
    } else if (d->ep_mode != GT_EP_NORMAL) {
        if (d->ep_mode != GT_EP_ROWDOT && d->ep_mode != GT_EP_MLP_BWD) return GT_EINVAL;
        if (d->N > 128 || d->batch0 * d->batch1 != 1 || d->n_out < 1 || d->n_out > 4 || !d->w2) return GT_ENOTSUP;
        if (d->ep_mode == GT_EP_ROWDOT && !d->out2) return GT_EINVAL;
        if (d->ep_mode == GT_EP_MLP_BWD && (!d->g2 || !d->dw2)) return GT_EINVAL;
        if (d->a_colsum || d->a_drop.p > 0.f) return GT_ENOTSUP;
    }
    if ((d->layout_a | d->layout_b) & ~1) return GT_EINVAL;
    if (d->C && !getenv("GT_GEMM_NO_TSMM") && tsmm_eligible(d)) return tsmm_run(d, ws, ws_bytes, stream);
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
        if (pl.split != 1 || pl.cfg > 1) return GT_ENOTSUP;
        p.ep_mode = d->ep_mode;
        p.hn_gamma = d->hn_gamma; p.hn_beta = d->hn_beta; p.hn_pos = d->hn_pos; p.hn_out = d->hn_out;
        p.hn_stats = d->hn_stats; p.hn_h = d->hn_h; p.hn_dk = d->hn_dk; p.hn_p = d->hn_p;
        p.hn_DP = (d->hn_dk + d->hn_p + 3) & ~3; p.hn_mask = d->hn_norm_mask; p.hn_eps = d->hn_eps;
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
    }

    dim3 grid((unsigned)(pl.tiles_m * pl.tiles_n), (unsigned)pl.split, (unsigned)batch);
    const int lay = d->layout_a * 2 + d->layout_b;
    if (pl.ws) {
        if (d->layout_b == 0) hipLaunchKernelGGL((gemm_ws_kernel<0>), dim3(ws_blocks(d)), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_ws_kernel<1>), dim3(ws_blocks(d)), dim3(256), 0, st, p);
    } else if (d->ep_mode == GT_EP_HEADNORM) {
        launch_hn(pl.cfg, grid, st, p);
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

// Widths just above a multiple of 128 (the merged-head width h*(d_k+p) = 144 of the Darcy model) would
// waste most of a second 128-wide tile column: run the aligned part and the remainder as two launches,
// the remainder on a narrow-tile configuration.
extern "C" int gt_gemm(const gt_gemm_desc* d, void* ws, int64_t ws_bytes, void* stream) {
    if (!d) return GT_EINVAL;
    const int rem = d->N % 128;
    Plan pl;
    // only when the aligned part alone fills the chip: two half-empty launches would serialise instead
    if (d->N > 128 && rem != 0 && rem <= 64 && make_plan(d, &pl) == 0 && pl.split == 1 && pl.bn == 128 &&
        (int64_t)pl.tiles_m * (d->N / 128) * d->batch0 * d->batch1 >= 512) {
        const int n_main = d->N - rem;
        gt_gemm_desc a = *d, b = *d;
        a.N = n_main;
        b.N = rem;
        b.B = d->B + (d->layout_b == 0 ? (int64_t)n_main * d->ldb : (int64_t)n_main);
        b.C = d->C + n_main;
        if (d->bias) b.bias = d->bias + n_main;
        if (d->rp) b.rp_b = d->rp_b + (int64_t)n_main * d->rp_ldb;
        if (d->add) b.add = d->add + n_main;
        if (d->pre) b.pre = d->pre + n_main;
        if (d->aux) b.aux = d->aux + n_main;
        if (d->res) b.res = d->res + n_main;
        a.split_k = b.split_k = 1;
        b.a_colsum = nullptr;
        int rc = gemm_one(&a, d->N, 0, ws, ws_bytes, stream);
        if (rc) return rc;
        return gemm_one(&b, d->N, n_main, ws, ws_bytes, stream);
    }
    return gemm_one(d, d->N, 0, ws, ws_bytes, stream);
}
