// Shared pieces of the GEMM engine (gt_gemm.hip: fp32-MFMA kernels; gt_gemm_x3.hip: split-operand bf16-MFMA
// kernel): the launch-parameter block, the operand loaders and the fused epilogue.  gfx950 only.
#pragma once
#include "gt_common.h"

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
    // GT_EP_HEADNORM (split-operand ring kernel): head-norm forward fused behind the QKV projection
    const float* hn_gamma; const float* hn_beta; const float* hn_pos; float* hn_out; float* hn_stats;
    int hn_h, hn_dk, hn_p, hn_DP, hn_mask, hn_skip_raw, hn_plain; float hn_eps;
    float* c2; int64_t ldc2; DropDev drop2;   // gt_gemm_desc.c_masked: the result once more under a second dropout mask
    int hn_dkr;              // real head width: == hn_dk, or 48 inside hn_dk = 64-column head SLOTS (gt_gemm.hip: hn_slots)
    int cv_H, cv_W, cv_C, cv_wgrad;  // implicit 3x3 convolution (gt_hip.h: cv_*), cv_C = 0: plain GEMM
    const void* Bp; int bp_NT, bp_KS, bp_f16; // packed-B kernel (gt_gemm_x3.hip): bf16 (fp16: bp_f16) planes of B in fragment order
    int wg_f16;                              // gemm_x3w_kernel: the GT_PREC_F16X2 token-contracted weight gradient
    int x3w_map;                             // gemm_x3w_kernel: 1 = 1-D grid, chunk-major inside an XCD (set by x3_launch)
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

// One row segment of the fused epilogue: v[0..NT) = accumulators of output row m, columns nb .. nb+NT-1 of batch
// entry z = (b0, b1); C already points at the batch entry (and split-K slab); biasv[t] = bias[nb + t] (0 past N).
template <int NT>
__device__ __forceinline__ void ep_row(const GemmP& p, float (&v)[NT], const float (&biasv)[NT], float* __restrict__ C,
                                       int m, int nb, int z, int b0, int b1, bool full, uint32_t dkey) {
    float* cp = C + (int64_t)m * p.ldc + nb;
    if (p.raw) {
        if (full && p.c_vec && NT == 4) {
            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (nb + t < p.N) cp[t] = v[t];
        }
        return;
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
    float dfac[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) dfac[t] = 0.f;
    if (p.act == GT_ACT_DROP_SILU) {
        // dropout in FRONT of the activation (gt_hip.h): u = keepscale * v, result silu(u); `pre` takes the factor the
        // backward multiplies the output gradient with, keepscale * silu'(u), in place of the pre-activation
        const uint32_t di = (uint32_t)(((int64_t)z * p.M + m) * p.drop_ld + p.n_off + nb);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float ks = p.drop.thresh ? drop_mul(p.drop, dkey, di + t) : 1.f;
            float a, da;
            silu_both(v[t] * ks, a, da);
            v[t] = a;
            dfac[t] = ks * da;
        }
    }
    if (p.act == GT_ACT_SILU2) {               // silu(silu(v)); `pre` takes the product of the two derivatives (gt_hip.h)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float a1, d1, a2, d2;
            silu_both(v[t], a1, d1);
            silu_both(a1, a2, d2);
            v[t] = a2;
            dfac[t] = d1 * d2;
        }
    }
    if (p.pre) {
        float* pp = p.pre + ((int64_t)z * p.M + m) * p.ldpre + nb;
        const bool df = p.act == GT_ACT_DROP_SILU || p.act == GT_ACT_SILU2;
        if (vec4) {
            *reinterpret_cast<f32x4*>(pp) = df ? f32x4{dfac[0], dfac[1 % NT], dfac[2 % NT], dfac[3 % NT]}
                                               : f32x4{v[0], v[1 % NT], v[2 % NT], v[3 % NT]};
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (nb + t < p.N) pp[t] = df ? dfac[t] : v[t];
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
    if (p.drop.thresh && p.act != GT_ACT_DROP_SILU) {
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
    if (p.c2) {                                // the same values under the second mask (gt_gemm_desc.c_masked)
        const uint32_t key2 = drop_key_dev(p.drop2);
        const uint32_t di = (uint32_t)(((int64_t)z * p.M + m) * p.drop_ld + p.n_off + nb);
        float* c2 = p.c2 + (int64_t)m * p.ldc2 + nb;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (nb + t < p.N) c2[t] = v[t] * (p.drop2.thresh ? drop_mul(p.drop2, key2, di + t) : 1.f);
    }
}

// Fused epilogue of the fp32-MFMA kernels.  The calling lane holds, for each of the MT x NT 16x16
// accumulator tiles, rows  mw0 + MT*(4*kq + r) + s  (r = 0..3) and columns  nb + t.
template <int MT, int NT>
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
            ep_row<NT>(p, v, biasv, C, m, nb, z, b0, b1, full, dkey);
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


// split-operand bf16 kernel (gt_gemm_x3.hip)
bool x3_shape_ok(const gt_gemm_desc* d);
bool x3_headnorm_ok(const GemmP& p, int layout_a, int layout_b, int planes);
int x3_launch(const GemmP& p, int layout_a, int layout_b, int planes, unsigned tiles, unsigned split, unsigned batch,
              hipStream_t st);
const char* x3_kernel_name(const GemmP& p, int layout_a, int layout_b, int planes, int hn_dk = 0);
// packed-B variant: eligibility for descriptor d (x3 planes already chosen), bytes of the plane buffer, and the pack
// launch (fills ws, sets p.Bp / bp_NT / bp_KS)
bool x3_packed_ok(const gt_gemm_desc* d, int planes, int split);
int64_t x3_packed_bytes(const gt_gemm_desc* d);
int x3_pack_b_many(const gt_gemm_desc* descs, void* const* outs, int n, hipStream_t st);
int x3_pack_b(const gt_gemm_desc* d, GemmP& p, void* ws, int64_t ws_bytes, hipStream_t st);
inline int hn_slot_width(int dk) { return dk == 48 ? 64 : dk; }     // head width -> columns of its slot in the N dimension
bool x3w_ok(const gt_gemm_desc* d, int split);

}  // namespace gt
