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
enum { FA_PLAIN = 0, FA_DROP = 1, FA_MASK = 2 };

__device__ __attribute__((aligned(16))) float fa_zero16[4] = {0.f, 0.f, 0.f, 0.f};
typedef __attribute__((address_space(3))) void* fa_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* fa_glb_ptr_t;

// Head tiles are DP = 16*NF + 4 floats wide (d_k in {16, 32, 48} plus the position columns, padded to a
// float4): the NF full 16-column groups of the second product run on MFMA, the last 4 columns on packed
// VALU FMAs (a third/quarter-empty MFMA tile would cost 25 % of the matrix time at DP = 36).
//
// LDS: the stream tile image is [64][DP] with no padding (pitch DP = 4 mod 16 makes both fragment read
// patterns conflict-free beyond the inherent 2 passes of a 64-lane read) -- i.e. linear in float4 granules,
// so it is filled by direct global->LDS loads (no staging registers), double-buffered: tile i+1 is in flight
// while tile i is consumed, one barrier per tile.
//
// Dropout (MODE == FA_DROP): the mask of score element idx is fmix32(idx*G + key) >= thresh (gt_common.h);
// idx is affine in the stream row, so idx*G + key is carried by additions; the owner fragments are pre-scaled
// by scale/(1-p) and a dropped score is a select, not a multiply.
template <int KS, bool DUAL, int MODE>     // KS = DP/4 contraction steps of the first product
__global__ __launch_bounds__(256, DUAL ? (KS > 9 ? 1 : 2) : (KS > 9 ? 2 : 3)) void fourier_core_kernel(const FourierP p) {
    constexpr int DP = 4 * KS, NF = (DP - 4) / 16, XC = DP - 4, TILE = FA_TS * DP;
    static_assert(DP % 16 == 4, "head tile width must be 16*NF + 4");
    __shared__ __attribute__((aligned(16))) float smem[2][2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z;
    const int o0 = blockIdx.x * (4 * FA_OW) + wave * FA_OW;
    const int64_t hD = (int64_t)p.h * DP;
    const int64_t base = ((int64_t)b * p.n) * hD + (int64_t)head * DP;
    const uint32_t zn = ((uint32_t)b * (uint32_t)p.h + (uint32_t)head) * (uint32_t)p.n;
    const int ntile = (p.n + FA_TS - 1) / FA_TS;

    auto issue = [&](int t, int buf) {
        const int s0 = t * FA_TS;
#pragma unroll
        for (int i = 0; i < (KS + 3) / 4; ++i) {
            const int q = wave + 4 * i;                    // 1-KiB chunk (64 float4 granules) of the tile image
            if (q < KS) {
                const int e = q * 64 + lane, r = e / KS, c = e % KS;
                const bool ok = s0 + r < p.n;
                const int64_t off = base + (int64_t)(s0 + r) * hD + 4 * c;
                const float* s1 = ok ? p.T1 + off : fa_zero16;
                const float* s2 = ok ? p.T2 + off : fa_zero16;
                __builtin_amdgcn_global_load_lds((fa_glb_ptr_t)s1, (fa_lds_ptr_t)(&smem[buf][0][q * 256]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((fa_glb_ptr_t)s2, (fa_lds_ptr_t)(&smem[buf][1][q * 256]), 16, 0, 0);
            }
        }
    };
    issue(0, 0);

    // owner fragments: B operand of the first product, lane (j, kq) holds fs * F[owner j][4s + kq]
    const float fs = p.scale * (MODE == FA_DROP ? p.drop.scale : 1.f);
    float f1[2][KS], f2[DUAL ? 2 : 1][DUAL ? KS : 1];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int o = o0 + 16 * nt + j, oc = min(o, p.n - 1);
        const float live = (o < p.n) ? fs : 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f1[nt][s] = live * p.F1[base + (int64_t)oc * hD + 4 * s + kq];
            if (DUAL) f2[nt][s] = live * p.F2[base + (int64_t)oc * hD + 4 * s + kq];
        }
    }
    // dropout hash carriers: hw[nt] = idx*G + key of (first stream row of this lane in the tile, owner nt)
    constexpr uint32_t G = 0x9e3779b1u;
    uint32_t hw[2] = {0u, 0u}, hstep = 0u;
    if (MODE == FA_DROP) {
        const uint32_t key = drop_key_dev(p.drop);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const uint32_t ow = (uint32_t)(o0 + 16 * nt + j), st = 4u * (uint32_t)kq;
            const uint32_t idx = p.owner_is_key ? (zn + st) * (uint32_t)p.n + ow : (zn + ow) * (uint32_t)p.n + st;
            hw[nt] = idx * G + key;
        }
        hstep = p.owner_is_key ? (uint32_t)p.n * G : G;      // idx step per stream row, times G
    }

    f32x4 acc1[NF][2], acc2[DUAL ? NF : 1][2];
    f32x2 ax1[2][2], ax2[2][2];                              // last 4 columns: [nt][column pair], partial over kq
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int dt = 0; dt < NF; ++dt) {
            acc1[dt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (DUAL) acc2[dt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ax1[nt][0] = ax1[nt][1] = ax2[nt][0] = ax2[nt][1] = f32x2{0.f, 0.f};
    }

    for (int t = 0; t < ntile; ++t) {
        // tile t has landed for this wave (vmcnt) and for everybody (barrier); everybody is also done with
        // tile t-1, whose buffer the next request overwrites
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 1 < ntile) issue(t + 1, (t + 1) & 1);
        const float* t1 = smem[t & 1][0];
        const float* t2 = smem[t & 1][1];
        const int s0 = t * FA_TS;

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
                const float a1 = t1[(16 * mt + j) * DP + 4 * s + kq];
                float a2 = 0.f;
                if (DUAL) a2 = t2[(16 * mt + j) * DP + 4 * s + kq];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    sa[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, f1[nt][s], sa[mt][nt], 0, 0, 0);
                    if (DUAL) sb[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, f2[nt][s], sb[mt][nt], 0, 0, 0);
                }
            }
        // mask: element (stream row s0 + 16mt + 4kq + r, owner column o0 + 16nt + j).  Rows / columns beyond n
        // hold exact zeros (zero-filled tile rows, zeroed owner fragments), whatever the mask says.
        if (MODE == FA_DROP) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                uint32_t hk = hw[nt];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool keep = fmix32(hk) >= p.drop.thresh;
                        sa[mt][nt][r] = keep ? sa[mt][nt][r] : 0.f;
                        if (DUAL) sb[mt][nt][r] = keep ? sb[mt][nt][r] : 0.f;
                        hk += hstep;
                    }
                    hk += 12u * hstep;
                }
                hw[nt] = hk;                                  // advanced by 64 stream rows
            }
        } else if (MODE == FA_MASK) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int st = min(s0 + 16 * mt + 4 * kq + r, p.n - 1), ow = min(o0 + 16 * nt + j, p.n - 1);
                        const int qi = p.owner_is_key ? st : ow, ki = p.owner_is_key ? ow : st;
                        const float m = p.mask[((int64_t)(b * p.h + head) * p.n + qi) * p.n + ki];
                        sa[mt][nt][r] *= m;
                        if (DUAL) sb[mt][nt][r] *= m;
                    }
        }
        // second product: O^T (dims x owners) += T^T (dims x stream) * S (stream x owners); k-step s of row
        // tile mt contracts stream row 16mt + 4kq + s = accumulator register s of this lane
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = 16 * mt + 4 * kq + s;
#pragma unroll
                for (int dt = 0; dt < NF; ++dt) {
                    const float a1 = t2[row * DP + 16 * dt + j];
                    float a2 = 0.f;
                    if (DUAL) a2 = t1[row * DP + 16 * dt + j];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc1[dt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, sa[mt][nt][s], acc1[dt][nt], 0, 0, 0);
                        if (DUAL)
                            acc2[dt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, sb[mt][nt][s], acc2[dt][nt], 0, 0, 0);
                    }
                }
                // last 4 columns on the vector unit: this lane's stream row `row`, its owner columns
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(&t2[row * DP + XC]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float sv = sa[mt][nt][s];
                    ax1[nt][0] += f32x2{sv, sv} * f32x2{x1[0], x1[1]};
                    ax1[nt][1] += f32x2{sv, sv} * f32x2{x1[2], x1[3]};
                }
                if (DUAL) {
                    const f32x4 x2 = *reinterpret_cast<const f32x4*>(&t1[row * DP + XC]);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const float sv = sb[mt][nt][s];
                        ax2[nt][0] += f32x2{sv, sv} * f32x2{x2[0], x2[1]};
                        ax2[nt][1] += f32x2{sv, sv} * f32x2{x2[2], x2[3]};
                    }
                }
            }
        }
    }
    // O^T tile (dt, nt): rows = dims 16dt + 4kq + r, column = owner o0 + 16nt + j  ->  O[owner][dim..dim+3]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int ow = o0 + 16 * nt + j;
#pragma unroll
        for (int dt = 0; dt < NF; ++dt)
            if (ow < p.n) {
                const int dim = 16 * dt + 4 * kq;
                *reinterpret_cast<f32x4*>(p.O1 + base + (int64_t)ow * hD + dim) = acc1[dt][nt];
                if (DUAL) *reinterpret_cast<f32x4*>(p.O2 + base + (int64_t)ow * hD + dim) = acc2[dt][nt];
            }
        // last 4 columns: sum the four kq partials (lanes j, j+16, j+32, j+48), lane kq == 0 stores
        f32x4 v1 = {ax1[nt][0][0], ax1[nt][0][1], ax1[nt][1][0], ax1[nt][1][1]};
        f32x4 v2 = {ax2[nt][0][0], ax2[nt][0][1], ax2[nt][1][0], ax2[nt][1][1]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            v1[c] += __shfl_xor(v1[c], 16, 64);
            v1[c] += __shfl_xor(v1[c], 32, 64);
            if (DUAL) {
                v2[c] += __shfl_xor(v2[c], 16, 64);
                v2[c] += __shfl_xor(v2[c], 32, 64);
            }
        }
        if (kq == 0 && ow < p.n) {
            *reinterpret_cast<f32x4*>(p.O1 + base + (int64_t)ow * hD + XC) = v1;
            if (DUAL) *reinterpret_cast<f32x4*>(p.O2 + base + (int64_t)ow * hD + XC) = v2;
        }
    }
}

template <int KS>
static void fourier_launch(const FourierP& p, bool dual, dim3 grid, hipStream_t st) {
    const int mode = p.mask ? FA_MASK : (p.drop.thresh ? FA_DROP : FA_PLAIN);
#define GT_FA(D, M) hipLaunchKernelGGL((fourier_core_kernel<KS, D, M>), grid, dim3(256), 0, st, p)
    if (dual) {
        if (mode == FA_DROP) GT_FA(true, FA_DROP);
        else if (mode == FA_MASK) GT_FA(true, FA_MASK);
        else GT_FA(true, FA_PLAIN);
    } else {
        if (mode == FA_DROP) GT_FA(false, FA_DROP);
        else if (mode == FA_MASK) GT_FA(false, FA_MASK);
        else GT_FA(false, FA_PLAIN);
    }
#undef GT_FA
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
        case 20: fourier_launch<5>(p, dual, grid, st); break;
        case 36: fourier_launch<9>(p, dual, grid, st); break;
        case 52: fourier_launch<13>(p, dual, grid, st); break;
        default: return GT_ENOTSUP;
    }
    GT_LAUNCH_CHECK();
    return 0;
}
