// Bias of a split-operand accumulation chain on the gfx950 matrix pipe, and what removes it.
//
// One wave per trial computes a 32 x 32 tile  C[i][j] = sum_k A[i][k] B[j][k]  with K = 16 S  from fp32 operands that the
// host has split exactly into three bf16 planes (the arithmetic of gt_gemm_x3.hip), in several ways:
//   chain      six plane products per stage into ONE accumulator, smallest first (what the kernels do)
//   chain_neg  the same with the M-side operand negated, result negated back (does the bias follow the accumulator's sign?)
//   stage0     the six products of a stage into a zero accumulator, added to the running sum with a VALU fp32 add
//   maincorr   a0 b0 in one accumulator, the five correction products in a second one, summed at the end
//   altrow     chain, odd rows i of the M-side operand negated (and negated back): per-row alternating bias sign
//   f32        8 x v_mfma_f32_32x32x2_f32 per stage on the unsplit operands
//   f16x2      two fp16 terms per operand (scaled by 2^e), three products per stage into one accumulator
// The host compares with the exact sum in long double and prints, per variant: mean SIGNED error and rms error, both
// relative to sum_k |a||b| (so a rounding that chops toward -inf shows as a negative mean whatever the operand signs).
//
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_chain_probe.hip -o tools/_bin/mfma_chain_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { CHAIN = 0, CHAIN_NEG, STAGE0, MAINCORR, ALTROW, F32, F16X2, NVAR };
static const char* kNames[NVAR] = {"chain", "chain_neg", "stage0", "maincorr", "altrow", "f32", "f16x2"};

// A, B: [trial][32 rows][K] fp32.  out: [trial][32 i][32 j].  scale_log2: f16x2 operand scale (power of two)
template <int VAR>
__global__ void chain(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int S, int e_a, int e_b) {
    const int t = blockIdx.x, lane = threadIdx.x, lr = lane & 31, lh = lane >> 5, K = 16 * S;
    const float* arow = A + ((size_t)t * 32 + lr) * K;      // M-side row lr  (the MFMA's second operand: output column index)
    const float* brow = B + ((size_t)t * 32 + lr) * K;      // N-side row lr  (the MFMA's first operand: output row index)
    f32x16 acc = {0}, acc2 = {0};
    const float sgn = (VAR == CHAIN_NEG) ? -1.f : (VAR == ALTROW && (lr & 1)) ? -1.f : 1.f;
    for (int s = 0; s < S; ++s) {
        float av[8], bv[8];
        for (int e = 0; e < 8; ++e) { av[e] = sgn * arow[16 * s + 8 * lh + e]; bv[e] = brow[16 * s + 8 * lh + e]; }
        if (VAR == F32) {
            for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[e], av[e], acc, 0, 0, 0);
            continue;
        }
        if (VAR == F16X2) {
            f16x8 a0, a1, b0, b1;
            const float sa = ldexpf(1.f, e_a), sb = ldexpf(1.f, e_b);
            for (int e = 0; e < 8; ++e) {
                const float x = av[e] * sa, y = bv[e] * sb;
                a0[e] = (_Float16)x; a1[e] = (_Float16)(x - (float)a0[e]);
                b0[e] = (_Float16)y; b1[e] = (_Float16)(y - (float)b0[e]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, acc, 0, 0, 0);
            continue;
        }
        bf16x8 a[3], b[3];
        for (int e = 0; e < 8; ++e) {
            float r = av[e], q = bv[e];
            for (int pl = 0; pl < 3; ++pl) {
                a[pl][e] = (__bf16)r; r -= (float)a[pl][e];
                b[pl][e] = (__bf16)q; q -= (float)b[pl][e];
            }
        }
        if (VAR == STAGE0) {
            f32x16 tacc = {0};
            for (int sum = 2; sum >= 0; --sum)
                for (int pa = 0; pa < 3; ++pa) {
                    const int pb = sum - pa;
                    if (pb < 0 || pb > 2) continue;
                    tacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[pb], a[pa], tacc, 0, 0, 0);
                }
            for (int e = 0; e < 16; ++e) acc[e] += tacc[e];
        } else if (VAR == MAINCORR) {
            for (int sum = 2; sum >= 1; --sum)
                for (int pa = 0; pa < 3; ++pa) {
                    const int pb = sum - pa;
                    if (pb < 0 || pb > 2) continue;
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[pb], a[pa], acc2, 0, 0, 0);
                }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[0], acc, 0, 0, 0);
        } else {
            for (int sum = 2; sum >= 0; --sum)
                for (int pa = 0; pa < 3; ++pa) {
                    const int pb = sum - pa;
                    if (pb < 0 || pb > 2) continue;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[pb], a[pa], acc, 0, 0, 0);
                }
        }
    }
    if (VAR == MAINCORR) for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
    if (VAR == F16X2) { const float us = ldexpf(1.f, -(e_a + e_b)); for (int e = 0; e < 16; ++e) acc[e] *= us; }
    // result layout (first operand = N-side rows -> output "row" index n, second = M-side -> lane's own m = lr):
    // register 4 g + t of lane (lr, lh) holds  n = 8 g + 4 lh + t,  m = lr
    for (int g = 0; g < 4; ++g)
        for (int tt = 0; tt < 4; ++tt)
            out[((size_t)t * 32 + lr) * 32 + 8 * g + 4 * lh + tt] = sgn * acc[4 * g + tt];
}

template <int VAR>
static void launch(const float* A, const float* B, float* o, int T, int S, int ea, int eb) {
    hipLaunchKernelGGL(chain<VAR>, dim3(T), dim3(64), 0, 0, A, B, o, S, ea, eb);
}

int main() {
    const int T = 256;
    std::mt19937_64 rng(99);
    std::normal_distribution<float> nd(0.f, 1.f);
    printf("{\"tile_outputs_per_case\": %d, \"results\": [\n", T * 1024);
    bool first = true;
    struct Case { const char* name; int S; bool positive; float offset; float amp_a; float amp_b; };
    const Case cases[] = {{"random signs, K=128", 8, false, 0.f, 1.f, 1.f},    {"random signs, K=1152", 72, false, 0.f, 1.f, 1.f},
                          {"positive, K=128", 8, true, 0.f, 1.f, 1.f},         {"positive, K=1152", 72, true, 0.f, 1.f, 1.f},
                          {"random signs, K=4096", 256, false, 0.f, 1.f, 1.f}, {"A = 3 + randn, K=1152", 72, false, 3.f, 1.f, 1.f},
                          {"gradient-like: A ~ 1e-5 randn, K=1152", 72, false, 0.f, 1e-5f, 1.f}};
    for (const Case& cs : cases) {
        const int K = 16 * cs.S;
        std::vector<float> A((size_t)T * 32 * K), B((size_t)T * 32 * K), o((size_t)T * 1024);
        float amax_a = 0, amax_b = 0;
        for (size_t i = 0; i < A.size(); ++i) {
            float x = nd(rng) * cs.amp_a + cs.offset * cs.amp_a, y = nd(rng) * cs.amp_b;
            if (cs.positive) { x = fabsf(x); y = fabsf(y); }
            A[i] = x; B[i] = y;
            amax_a = fmaxf(amax_a, fabsf(x)); amax_b = fmaxf(amax_b, fabsf(y));
        }
        // f16x2 scale: amax -> [2^13, 2^14)
        const int ea = 13 - (int)floorf(log2f(amax_a)), eb = 13 - (int)floorf(log2f(amax_b));
        float *dA, *dB, *dO;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, o.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<long double> ref((size_t)T * 1024), sab((size_t)T * 1024);
        for (int t = 0; t < T; ++t)
            for (int m = 0; m < 32; ++m)
                for (int n = 0; n < 32; ++n) {
                    long double r = 0, s = 0;
                    const float* a = &A[((size_t)t * 32 + m) * K];
                    const float* b = &B[((size_t)t * 32 + n) * K];
                    for (int k = 0; k < K; ++k) { r += (long double)a[k] * b[k]; s += fabsl((long double)a[k] * b[k]); }
                    ref[((size_t)t * 32 + m) * 32 + n] = r; sab[((size_t)t * 32 + m) * 32 + n] = s;
                }
        for (int v = 0; v < NVAR; ++v) {
            switch (v) {
                case CHAIN: launch<CHAIN>(dA, dB, dO, T, cs.S, ea, eb); break;
                case CHAIN_NEG: launch<CHAIN_NEG>(dA, dB, dO, T, cs.S, ea, eb); break;
                case STAGE0: launch<STAGE0>(dA, dB, dO, T, cs.S, ea, eb); break;
                case MAINCORR: launch<MAINCORR>(dA, dB, dO, T, cs.S, ea, eb); break;
                case ALTROW: launch<ALTROW>(dA, dB, dO, T, cs.S, ea, eb); break;
                case F32: launch<F32>(dA, dB, dO, T, cs.S, ea, eb); break;
                default: launch<F16X2>(dA, dB, dO, T, cs.S, ea, eb); break;
            }
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
            hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost);
            long double se = 0, s2 = 0, mx = 0, num = 0, den = 0;
            for (size_t i = 0; i < o.size(); ++i) {
                const long double d = (long double)o[i] - ref[i], e = d / sab[i];
                se += e; s2 += e * e; if (fabsl(e) > mx) mx = fabsl(e);
                num += d * d; den += ref[i] * ref[i];
            }
            const double n = (double)o.size();
            printf("%s {\"case\": \"%s\", \"variant\": \"%s\", \"mean_signed\": %.3e, \"rms\": %.3e, \"max\": %.3e, \"rel_l2\": %.3e, "
                   "\"bias_over_rms\": %.3f}", first ? " " : ",\n ", cs.name, kNames[v], (double)(se / n), (double)sqrtl(s2 / n),
                   (double)mx, (double)sqrtl(num / den), (double)((se / n) / sqrtl(s2 / n)));
            first = false;
        }
        hipFree(dA); hipFree(dB); hipFree(dO);
    }
    printf("\n]}\n");
    return 0;
}
