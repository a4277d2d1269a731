// Truncated-DFT stages along the contiguous grid axis of SpectralConv2d (layers.py:1176 rfft2 / :1187 irfft2
// restricted to the kept modes, plus the residual nn.Linear of layers.py:1128,1196): the two skinny batched
// products of the S1..S4 pipeline (spectral.py) whose batch is one grid line,
//
//   analysis   Y[b] (P x 32)  = F^T (P x n) X[b] (n x 32)                                   b < nb = B*n lines
//   synthesis  Y[b] (n x 32)  = act( F (n x P) Z[b] (P x 32) + X2[b] (n x 32) W2 (32 x 32) + bias )
//
// with P = 2*modes <= 32 and 32 channels.  A line is 18-27 KB and needs only 0.3-0.6 MFLOP, so these are HBM
// streams; on the general GEMM engine every line was one under-filled 64x64 / 128x32 tile whose load latency
// nothing covered (1.3-1.8 TB/s).  Here a persistent block walks its lines with the line slab (and Z) brought
// in by direct global->LDS loads one or two lines ahead (no staging registers, one barrier per line), the
// item-independent operands (F, W2) live in registers, and the four waves split the output tiles.
//
// LDS images are linear in 16-byte granules (direct loads write wave-uniform base + lane*16), so the bank
// swizzles are applied on the SOURCE address: granule q of row r holds logical granule q ^ f(r).
//   analysis  (B operand, lane = (column j, k-row 4s+kq)):   f(r) = ((r >> 1) & 1) << 2
//   synthesis (A operand, lane = (row 16mt+i, k 4s+kq)):     f(r) = r & 7
// Edge rows are CLAMPED, not masked (duplicate identical stores): every wave issues the same number of
// stores per line, which the counted vmcnt waits rely on (vmcnt retires in order, stores included).
#include "gt_common.h"
#include <algorithm>

namespace gt {

__device__ __attribute__((aligned(16))) float dft_zero16[4] = {0.f, 0.f, 0.f, 0.f};
typedef __attribute__((address_space(3))) void* dft_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* dft_glb_ptr_t;

constexpr int DFT_C = 32;            // channels (freq_dim of every spectral config but the NS-lite one)

__device__ __forceinline__ void dft_wait_vm(int n) {   // s_waitcnt vmcnt(min(n, 63)) for a wave-uniform n
#define GT_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 63 ? n : 63) {
        GT_VMC(0) GT_VMC(1) GT_VMC(2) GT_VMC(3) GT_VMC(4) GT_VMC(5) GT_VMC(6) GT_VMC(7) GT_VMC(8) GT_VMC(9)
        GT_VMC(10) GT_VMC(11) GT_VMC(12) GT_VMC(13) GT_VMC(14) GT_VMC(15) GT_VMC(16) GT_VMC(17) GT_VMC(18)
        GT_VMC(19) GT_VMC(20) GT_VMC(21) GT_VMC(22) GT_VMC(23) GT_VMC(24) GT_VMC(25) GT_VMC(26) GT_VMC(27)
        GT_VMC(28) GT_VMC(29) GT_VMC(30) GT_VMC(31) GT_VMC(32) GT_VMC(33) GT_VMC(34) GT_VMC(35) GT_VMC(36)
        GT_VMC(37) GT_VMC(38) GT_VMC(39) GT_VMC(40) GT_VMC(41) GT_VMC(42) GT_VMC(43) GT_VMC(44) GT_VMC(45)
        GT_VMC(46) GT_VMC(47) GT_VMC(48) GT_VMC(49) GT_VMC(50) GT_VMC(51) GT_VMC(52) GT_VMC(53) GT_VMC(54)
        GT_VMC(55) GT_VMC(56) GT_VMC(57) GT_VMC(58) GT_VMC(59) GT_VMC(60) GT_VMC(61) GT_VMC(62) GT_VMC(63)
    }
#undef GT_VMC
}

struct DftAP {
    const float* F; const float* X; float* Y;
    int nb, n, P, nbuf;
};

// KMAX = contraction steps (4 rows each) the kernel always runs; rows >= n are zero on both sides.
template <int KMAX>
__global__ __launch_bounds__(256, 2) void dft_analysis_kernel(const DftAP p) {
    constexpr int NCH = (KMAX * 4 * DFT_C * 4 / 1024 + 3) / 4 * 4;    // 1-KiB chunks per line buffer
    constexpr int LPI = NCH / 4;                                        // direct loads per wave per line
    constexpr int BUF = NCH * 256;                                      // floats
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int mt = wave >> 1, nt = wave & 1;
    const int gran_valid = p.n * (DFT_C / 4);

    auto issue = [&](int item, int buf) {
        const float* src0 = p.X + (int64_t)item * p.n * DFT_C;
#pragma unroll
        for (int i = 0; i < LPI; ++i) {
            const int q = wave + 4 * i, e = q * 64 + lane;
            const int r = e >> 3, g = (e & 7) ^ (((r >> 1) & 1) << 2);
            const float* src = (e < gran_valid) ? src0 + r * DFT_C + g * 4 : dft_zero16;
            __builtin_amdgcn_global_load_lds((dft_glb_ptr_t)src, (dft_lds_ptr_t)(smem + buf * BUF + q * 256), 16, 0, 0);
        }
    };
    const int first = blockIdx.x, stride = gridDim.x;
    const int nmine = first < p.nb ? (p.nb - 1 - first) / stride + 1 : 0;
    for (int a = 0; a < p.nbuf - 1 && a < nmine; ++a) issue(first + a * stride, a);

    // A operand: F^T tile, lane (i = output row 16mt + j (clamped to P-1), kq) holds F[4s + kq][row]
    float fa[KMAX];
    {
        const int pr = min(16 * mt + j, p.P - 1);
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            const int r = 4 * s + kq;
            fa[s] = (r < p.n) ? p.F[(int64_t)r * p.P + pr] : 0.f;
        }
        // pin the fragments before the loop: otherwise the compiler's own wait for these loads lands inside
        // the loop body as a vmcnt(0) per line, which would also drain the prefetched lines
#pragma unroll
        for (int s = 0; s < KMAX; ++s) asm volatile("" : "+v"(fa[s]));
    }
    // B operand read offsets: row 4s + kq, column 16nt + j -> granule (4nt + j/4) ^ f(row); f depends on kq only
    const int slot = ((4 * nt + (j >> 2)) ^ (((kq >> 1) & 1) << 2));
    const int boff = kq * DFT_C + slot * 4 + (j & 3);
    int buf = 0;
    for (int t = 0; t < nmine; ++t) {
        // lines t+1 .. t+nbuf-2 may stay in flight, plus the 4 stores of each line issued after them
        dft_wait_vm((p.nbuf - 2) * (LPI + 4) + (t > 0 ? 4 : 0));
        asm volatile("s_barrier" ::: "memory");
        if (t + p.nbuf - 1 < nmine) {
            int nb = buf + p.nbuf - 1;
            if (nb >= p.nbuf) nb -= p.nbuf;
            issue(first + (t + p.nbuf - 1) * stride, nb);
        } else {
            // keep the per-line instruction count constant for the counted waits
#pragma unroll
            for (int i = 0; i < LPI; ++i)
                __builtin_amdgcn_global_load_lds((dft_glb_ptr_t)dft_zero16, (dft_lds_ptr_t)(smem + p.nbuf * BUF), 16, 0, 0);
        }
        const float* xb = smem + buf * BUF + boff;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KMAX; s += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], xb[(4 * s) * DFT_C], acc0, 0, 0, 0);
            if (s + 1 < KMAX)
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s + 1], xb[(4 * s + 4) * DFT_C], acc1, 0, 0, 0);
        }
        float* y = p.Y + (int64_t)(first + t * stride) * p.P * DFT_C + 16 * nt + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pr = min(16 * mt + 4 * kq + r, p.P - 1);
            y[pr * DFT_C] = acc0[r] + acc1[r];
        }
        if (++buf == p.nbuf) buf = 0;
    }
}

struct DftSP {
    const float* F; const float* Z; const float* X2; const float* W2; const float* bias;
    float* Y; float* pre;
    int nb, n, P, act;
    const float* gate = nullptr;       // gt_dft_synthesis_gated: Y *= silu'(gate) (same layout as Y)
};

// MAXT = output row tiles (16 rows) per wave: ceil(ceil(n/16) / 4)
template <int MAXT, bool PRE>
__global__ __launch_bounds__(256, 2) void dft_synthesis_kernel(const DftSP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int xch = (p.n * DFT_C * 4 + 1023) / 1024;             // chunks of the X2 line
    const int zch = (p.P * DFT_C * 4 + 1023) / 1024;             // chunks of the Z line (linear copy)
    const int nch = (xch + zch + 3) / 4 * 4, lpi = nch / 4, BUF = nch * 256;
    const int xg = p.n * (DFT_C / 4), zg = p.P * (DFT_C / 4);
    const int mtiles = (p.n + 15) / 16;
    const int nmw = wave < mtiles ? (mtiles - 1 - wave) / 4 + 1 : 0;   // row tiles of this wave: wave, wave+4, ..
    const int spl = nmw * (PRE ? 8 : 4);                          // stores per line of this wave
    const int ks1 = p.P / 4;

    auto issue = [&](int item, int buf) {
        const float* xs = p.X2 + (int64_t)item * p.n * DFT_C;
        const float* zs = p.Z + (int64_t)item * p.P * DFT_C;
        for (int i = 0; i < lpi; ++i) {
            const int q = wave + 4 * i, e = q * 64 + lane;
            const float* src;
            if (q < xch) {
                const int r = e >> 3, g = (e & 7) ^ (r & 7);
                src = (e < xg) ? xs + r * DFT_C + g * 4 : dft_zero16;
            } else {
                const int ez = e - xch * 64;
                src = (ez < zg) ? zs + ez * 4 : dft_zero16;
            }
            __builtin_amdgcn_global_load_lds((dft_glb_ptr_t)src, (dft_lds_ptr_t)(smem + buf * BUF + q * 256), 16, 0, 0);
        }
    };
    const int first = blockIdx.x, stride = gridDim.x;
    const int nmine = first < p.nb ? (p.nb - 1 - first) / stride + 1 : 0;
    if (nmine > 0) issue(first, 0);

    // item-independent operands in registers.  Output column of lane j in column tile nt: 2j + nt (a lane owns
    // two adjacent columns -> float2 stores, full 128-byte rows per 16 lanes)
    float fa[MAXT][8], w2[8][2], bv[2];
#pragma unroll
    for (int mi = 0; mi < MAXT; ++mi) {
        const int row = min(16 * (wave + 4 * mi) + j, p.n - 1);
#pragma unroll
        for (int s = 0; s < 8; ++s) fa[mi][s] = (s < ks1 && mi < nmw) ? p.F[(int64_t)row * p.P + 4 * s + kq] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) w2[s][nt] = p.W2[(4 * s + kq) * DFT_C + 2 * j + nt];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bv[nt] = p.bias ? p.bias[2 * j + nt] : 0.f;
    // pin the register operands before the loop (see dft_analysis_kernel)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int mi = 0; mi < MAXT; ++mi) asm volatile("" : "+v"(fa[mi][s]));
        asm volatile("" : "+v"(w2[s][0]), "+v"(w2[s][1]));
    }
    asm volatile("" : "+v"(bv[0]), "+v"(bv[1]));

    for (int t = 0; t < nmine; ++t) {
        const int buf = t & 1;
        dft_wait_vm(t > 0 ? spl : 0);                 // only the stores of line t-1 may still be in flight
        asm volatile("s_barrier" ::: "memory");
        if (t + 1 < nmine) issue(first + (t + 1) * stride, buf ^ 1);
        const float* xb = smem + buf * BUF;
        const float* zb = xb + xch * 256;
        // B operand of the first product: Z[4s + kq][2j + nt]
        f32x2 zf[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
            zf[s] = (s < ks1) ? *reinterpret_cast<const f32x2*>(&zb[(4 * s + kq) * DFT_C + 2 * j]) : f32x2{0.f, 0.f};
        const int64_t obase = (int64_t)(first + t * stride) * p.n * DFT_C;
#pragma unroll
        for (int mi = 0; mi < MAXT; ++mi) {
            if (mi < nmw) {
                const int m0 = 16 * (wave + 4 * mi);
                const int row = min(m0 + j, p.n - 1);
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int s = 0; s < 8; ++s) {           // k-steps beyond P/4 carry zero fragments on both sides
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi][s], zf[s][nt], acc[nt], 0, 0, 0);
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float a2 = xb[row * DFT_C + ((s ^ (row & 7)) << 2) + kq];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, w2[s][nt], acc[nt], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int orow = min(m0 + 4 * kq + r, p.n - 1);
                    f32x2 v = {acc[0][r] + bv[0], acc[1][r] + bv[1]};
                    if (PRE) *reinterpret_cast<f32x2*>(p.pre + obase + (int64_t)orow * DFT_C + 2 * j) = v;
                    if (p.act == GT_ACT_SILU) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); }
                    else if (p.act == GT_ACT_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
                    if (p.gate) {
                        const f32x2 gt = *reinterpret_cast<const f32x2*>(p.gate + obase + (int64_t)orow * DFT_C + 2 * j);
                        v[0] *= dsilu_f(gt[0]); v[1] *= dsilu_f(gt[1]);
                    }
                    *reinterpret_cast<f32x2*>(p.Y + obase + (int64_t)orow * DFT_C + 2 * j) = v;
                }
            }
        }
    }
}

}  // namespace gt

using namespace gt;

static int dft_blocks(int nb, int per_cu) { return std::min(nb, per_cu * 256); }

extern "C" int gt_dft_analysis(const float* F, const float* X, float* Y, int32_t nb, int32_t n, int32_t P,
                               int32_t C, void* stream) {
    if (!F || !X || !Y || nb <= 0 || n <= 0 || P <= 0 || C <= 0) return GT_EINVAL;
    if (C != DFT_C || P > 32 || n > 224) return GT_ENOTSUP;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(Y) & 3)) return GT_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int kmax = n <= 64 ? 16 : (n <= 144 ? 36 : 56);
    const int nch = (kmax * 4 * DFT_C * 4 / 1024 + 3) / 4 * 4;
    const int nbuf = (3 * nch + 1) * 1024 <= 65536 ? 3 : 2;
    const size_t lds = (size_t)(nbuf * nch + 1) * 1024;              // + 1 KiB sink for the balancing loads
    DftAP p{F, X, Y, nb, n, P, nbuf};
    dim3 grid((unsigned)dft_blocks(nb, 2));
    switch (kmax) {
        case 16: hipLaunchKernelGGL((dft_analysis_kernel<16>), grid, dim3(256), lds, st, p); break;
        case 36: hipLaunchKernelGGL((dft_analysis_kernel<36>), grid, dim3(256), lds, st, p); break;
        default: hipLaunchKernelGGL((dft_analysis_kernel<56>), grid, dim3(256), lds, st, p); break;
    }
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_dft_synthesis(const float* F, const float* Z, float* Y, int32_t nb, int32_t n, int32_t P,
                                int32_t Co, const float* X2, const float* W2, int32_t C2, const float* bias,
                                int32_t act, float* pre, void* stream) {
    return gt_dft_synthesis_gated(F, Z, Y, nb, n, P, Co, X2, W2, C2, bias, act, pre, nullptr, stream);
}

extern "C" int gt_dft_synthesis_gated(const float* F, const float* Z, float* Y, int32_t nb, int32_t n, int32_t P,
                                      int32_t Co, const float* X2, const float* W2, int32_t C2, const float* bias,
                                      int32_t act, float* pre, const float* out_gate, void* stream) {
    if (!F || !Z || !Y || nb <= 0 || n <= 0 || P <= 0 || Co <= 0) return GT_EINVAL;
    if (out_gate && (act != GT_ACT_NONE || pre)) return GT_EINVAL;
    if (reinterpret_cast<uintptr_t>(out_gate) & 7) return GT_EALIGN;
    if (act != GT_ACT_NONE && act != GT_ACT_RELU && act != GT_ACT_SILU) return GT_EINVAL;
    if (Co != DFT_C || C2 != DFT_C || !X2 || !W2 || P > 32 || (P & 3) || n > 256) return GT_ENOTSUP;
    const uintptr_t al = reinterpret_cast<uintptr_t>(X2) | reinterpret_cast<uintptr_t>(Z);
    if ((al & 15) || (reinterpret_cast<uintptr_t>(Y) & 7) || (reinterpret_cast<uintptr_t>(pre) & 7)) return GT_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int xch = (n * DFT_C * 4 + 1023) / 1024, zch = (P * DFT_C * 4 + 1023) / 1024;
    const int nch = (xch + zch + 3) / 4 * 4;
    const size_t lds = (size_t)(2 * nch) * 1024;
    if (lds > 65536) return GT_ENOTSUP;
    DftSP p{F, Z, X2, W2, bias, Y, pre, nb, n, P, act, out_gate};
    const int maxt = ((n + 15) / 16 + 3) / 4;
    dim3 grid((unsigned)dft_blocks(nb, lds <= 53 * 1024 ? 3 : 2));
#define GT_DS(T)                                                                                         \
    case T:                                                                                              \
        if (pre) hipLaunchKernelGGL((dft_synthesis_kernel<T, true>), grid, dim3(256), lds, st, p);       \
        else hipLaunchKernelGGL((dft_synthesis_kernel<T, false>), grid, dim3(256), lds, st, p);          \
        break;
    switch (maxt) {
        GT_DS(1) GT_DS(2) GT_DS(3) GT_DS(4)
        default: return GT_ENOTSUP;
    }
#undef GT_DS
    GT_LAUNCH_CHECK();
    return 0;
}
