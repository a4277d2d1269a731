// Fused Fourier-type attention  out = ((Q' K'^T) * scale .* mask) V'   (layers.py:672-705) without ever
// writing the n x n score matrix to HBM: a softmax-free "flash" kernel -- no running maximum, the score
// tile is scaled, masked (stateless dropout or an explicit mask) and consumed by the second product in
// registers.  One template serves the three passes of training:
//
//   owner side  = the rows the block owns (resident MFMA B fragments F1 [, F2]), 128 per block, 32 per wave
//   stream side = the other token axis, walked in 64-row tiles staged in LDS (T1, T2)
//
//   Sa^T-tile = T1 F1^T  (stream x owner),  scaled and masked          [DUAL: Sb-tile = T2 F2^T likewise]
//   O1^T     += T2^T Sa                                                 [DUAL: O2^T += T1^T Sb]
//
//   forward          owner = queries : F1 = Q',  T1 = K', T2 = V'            -> O1 = attention output
//   d/dQ'            owner = queries : F1 = dO,  T1 = V', T2 = K'            -> O1 = dQ'
//   d/dV', d/dK'     owner = keys    : F1 = K', F2 = V', T1 = Q', T2 = dO    -> O1 = dV', O2 = dK'   (DUAL)
//
// Trick that avoids any cross-lane traffic between the two products: the MFMA D layout of a 16x16 score
// tile puts stream rows 4*(lane>>4)+r, r = 0..3, in the lane's 4 accumulator registers; the second product
// is free to enumerate its contraction index in any order, so its k-step s uses stream row 4*(lane>>4)+s --
// exactly register s of the same lane.  The A operand (T^T) is read from LDS with the matching row.
#include "gt_common.h"
#include <algorithm>

namespace gt {

struct FourierP {
    const float* F1; const float* F2; const float* T1; const float* T2;
    float* O1; float* O2;
    const float* mask;           // explicit multiplicative mask [B,h,n,n] (query-major) or null
    DropDev drop;
    int n, h;
    float scale;
    int owner_is_key;
};

constexpr int FA_TS = 64;        // stream rows per LDS tile
constexpr int FA_OW = 32;        // owner rows per wave

template <int KS, bool DUAL>     // KS = DP/4 contraction steps of the first product
__global__ __launch_bounds__(256) void fourier_core_kernel(const FourierP p) {
    constexpr int DP = 4 * KS, NDT = (DP + 15) / 16, LP = 16 * NDT + 4;   // LDS row pitch (2-way conflicts at most)
    __shared__ __attribute__((aligned(16))) float t1[FA_TS * LP];
    __shared__ __attribute__((aligned(16))) float t2[FA_TS * LP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z;
    const int o0 = blockIdx.x * (4 * FA_OW) + wave * FA_OW;
    const int64_t hD = (int64_t)p.h * DP;
    const int64_t base = ((int64_t)b * p.n) * hD + (int64_t)head * DP;
    const uint32_t key = drop_key_dev(p.drop);
    const uint32_t zn = ((uint32_t)b * (uint32_t)p.h + (uint32_t)head) * (uint32_t)p.n;

    // zero the LDS pad columns once (columns >= DP are never written again)
    for (int e = tid; e < FA_TS * LP; e += 256) { t1[e] = 0.f; t2[e] = 0.f; }

    // owner fragments: B operand of the first product, lane (j, kq) holds F[owner j][4s + kq]
    float f1[2][KS], f2[DUAL ? 2 : 1][DUAL ? KS : 1];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int o = o0 + 16 * nt + j;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f1[nt][s] = (o < p.n) ? p.F1[base + (int64_t)o * hD + 4 * s + kq] : 0.f;
            if (DUAL) f2[nt][s] = (o < p.n) ? p.F2[base + (int64_t)o * hD + 4 * s + kq] : 0.f;
        }
    }
    f32x4 acc1[NDT][2], acc2[DUAL ? NDT : 1][2];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            acc1[dt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (DUAL) acc2[dt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    __syncthreads();

    for (int s0 = 0; s0 < p.n; s0 += FA_TS) {
        // stage the stream tile: rows s0..s0+63 (zero beyond n), DP floats each, coalesced float4
        for (int e = tid; e < FA_TS * KS; e += 256) {
            const int r = e / KS, c = e % KS, st = s0 + r;
            f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = {0.f, 0.f, 0.f, 0.f};
            if (st < p.n) {
                v1 = *reinterpret_cast<const f32x4*>(p.T1 + base + (int64_t)st * hD + 4 * c);
                v2 = *reinterpret_cast<const f32x4*>(p.T2 + base + (int64_t)st * hD + 4 * c);
            }
            *reinterpret_cast<f32x4*>(&t1[r * LP + 4 * c]) = v1;
            *reinterpret_cast<f32x4*>(&t2[r * LP + 4 * c]) = v2;
        }
        __syncthreads();

        // first product: score tiles (stream rows x owner columns), 4 row tiles x 2 column tiles per wave
        f32x4 sa[4][2], sb[DUAL ? 4 : 1][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                sa[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (DUAL) sb[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float a1 = t1[(16 * mt + j) * LP + 4 * s + kq];
                float a2 = 0.f;
                if (DUAL) a2 = t2[(16 * mt + j) * LP + 4 * s + kq];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    sa[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, f1[nt][s], sa[mt][nt], 0, 0, 0);
                    if (DUAL) sb[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, f2[nt][s], sb[mt][nt], 0, 0, 0);
                }
            }
        // scale + mask: element (stream row s0 + 16mt + 4kq + r, owner column o0 + 16nt + j)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int st = s0 + 16 * mt + 4 * kq + r, ow = o0 + 16 * nt + j;
                    const int qi = p.owner_is_key ? st : ow, ki = p.owner_is_key ? ow : st;
                    float m = p.scale;
                    if (st < p.n && ow < p.n) {
                        const uint32_t idx = (zn + (uint32_t)qi) * (uint32_t)p.n + (uint32_t)ki;
                        if (p.mask) m *= p.mask[((int64_t)(b * p.h + head) * p.n + qi) * p.n + ki];
                        else if (p.drop.thresh) m *= drop_mul(p.drop, key, idx);
                    }
                    sa[mt][nt][r] *= m;
                    if (DUAL) sb[mt][nt][r] *= m;
                }
        // second product: O^T (dims x owners) += T^T (dims x stream) * S (stream x owners); k-step s of row
        // tile mt contracts stream row 16mt + 4kq + s = accumulator register s of this lane
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const float a1 = t2[(16 * mt + 4 * kq + s) * LP + 16 * dt + j];
                    float a2 = 0.f;
                    if (DUAL) a2 = t1[(16 * mt + 4 * kq + s) * LP + 16 * dt + j];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc1[dt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, sa[mt][nt][s], acc1[dt][nt], 0, 0, 0);
                        if (DUAL)
                            acc2[dt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, sb[mt][nt][s], acc2[dt][nt], 0, 0, 0);
                    }
                }
        __syncthreads();
    }
    // O^T tile (dt, nt): rows = dims 16dt + 4kq + r, column = owner o0 + 16nt + j  ->  O[owner][dim..dim+3]
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int ow = o0 + 16 * nt + j, dim = 16 * dt + 4 * kq;
            if (ow < p.n && dim < DP) {
                *reinterpret_cast<f32x4*>(p.O1 + base + (int64_t)ow * hD + dim) = acc1[dt][nt];
                if (DUAL) *reinterpret_cast<f32x4*>(p.O2 + base + (int64_t)ow * hD + dim) = acc2[dt][nt];
            }
        }
}

}  // namespace gt

using namespace gt;

extern "C" int gt_fourier_attn(const float* F1, const float* F2, const float* T1, const float* T2, float* O1,
                               float* O2, int32_t B, int32_t n, int32_t h, int32_t DP, float scale,
                               const float* mask, const gt_dropout* drop, int32_t owner_is_key, void* stream) {
    if (!F1 || !T1 || !T2 || !O1 || B <= 0 || n <= 0 || h <= 0 || DP <= 0) return GT_EINVAL;
    const bool dual = F2 != nullptr;
    if (dual && !O2) return GT_EINVAL;
    if (drop && drop->p > 0.f && !drop->seed) return GT_EINVAL;
    if (B > 65535 || h > 65535) return GT_EINVAL;
    const uintptr_t al = reinterpret_cast<uintptr_t>(T1) | reinterpret_cast<uintptr_t>(T2) |
                         reinterpret_cast<uintptr_t>(O1) | reinterpret_cast<uintptr_t>(O2);
    if (al & 15) return GT_EALIGN;
    FourierP p{F1, F2, T1, T2, O1, O2, mask, make_drop(mask ? nullptr : drop), n, h, scale, owner_is_key};
    dim3 grid((unsigned)ceil_div(n, 4 * FA_OW), (unsigned)h, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    switch (DP) {
        case 20:
            if (dual) hipLaunchKernelGGL((fourier_core_kernel<5, true>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((fourier_core_kernel<5, false>), grid, dim3(256), 0, st, p);
            break;
        case 36:
            if (dual) hipLaunchKernelGGL((fourier_core_kernel<9, true>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((fourier_core_kernel<9, false>), grid, dim3(256), 0, st, p);
            break;
        case 52:
            if (dual) hipLaunchKernelGGL((fourier_core_kernel<13, true>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((fourier_core_kernel<13, false>), grid, dim3(256), 0, st, p);
            break;
        default: return GT_ENOTSUP;
    }
    GT_LAUNCH_CHECK();
    return 0;
}
