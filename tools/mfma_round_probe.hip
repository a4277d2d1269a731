// How does a gfx950 MFMA round when it adds its dot product to the accumulator?  ("test whether
// v_mfma_f32_32x32x16_bf16's internal accumulation is biased".)
//
// One wave per trial.  Every row of the MFMA "A" operand is the same 16-vector a[k] and every column of "B" the same
// b[k], so all 1024 results of one instruction are  dot(a, b) + c[i][j]  with 1024 different accumulator inputs c --
// the operand-to-lane layout does not matter (a lane's eight k values are a_lo for lanes 0..31, a_hi for 32..63).
// The host evaluates dot + c exactly (long double: bf16 / f16 products are exact, 16 of them and c fit) and classifies
// every result: equal to round-to-nearest-even of the exact sum, equal to truncation (round toward zero), or neither;
// and reports the mean signed error in units of the result's ulp.
//
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_round_probe.hip -o tools/_bin/mfma_round_probe && tools/_bin/mfma_round_probe
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
typedef float f32x4 __attribute__((ext_vector_type(4)));

// a, b: [trial][2 halves][8] floats (already representable in the operand type); c, d: [trial][64 lanes][16]
template <int MODE>      // 0: 32x32x16 bf16   1: 32x32x16 f16   2: 16 steps of 32x32x1... (see below) fp32 16x16x4 chain
__global__ void probe(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                      float* __restrict__ d) {
    const int t = blockIdx.x, lane = threadIdx.x, lh = lane >> 5;
    const float* av = a + (t * 2 + lh) * 8;
    const float* bv = b + (t * 2 + lh) * 8;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = c[(t * 64 + lane) * 16 + e];
    if (MODE == 0) {
        bf16x8 x, y;
        for (int e = 0; e < 8; ++e) { x[e] = (__bf16)av[e]; y[e] = (__bf16)bv[e]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
    } else if (MODE == 1) {
        f16x8 x, y;
        for (int e = 0; e < 8; ++e) { x[e] = (_Float16)av[e]; y[e] = (_Float16)bv[e]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0);
    } else {
        // fp32 MFMA 32x32x2: lane half lh supplies k = lh; eight instructions walk this lane-half's eight values, so the
        // sixteen products enter as eight chained two-term steps (what gt_gemm.hip's fp32 kernels do, two k per step)
        for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) d[(t * 64 + lane) * 16 + e] = acc[e];
}

static float to_bf16(float x) {          // RNE to bf16, returned as float
    uint32_t u; memcpy(&u, &x, 4);
    u += 0x7FFF + ((u >> 16) & 1); u &= 0xFFFF0000u;
    float r; memcpy(&r, &u, 4); return r;
}
static float to_f16(float x) { return (float)(_Float16)x; }

static float rz(long double x) {          // truncation of x to float
    float r = (float)x;                   // RNE
    if (fabsl((long double)r) > fabsl(x)) r = nextafterf(r, 0.f);
    return r;
}

struct Stat { long n = 0, rn = 0, tz = 0, both = 0, other = 0; double serr = 0, aerr = 0; };

int main() {
    const int T = 512;
    std::mt19937_64 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    float *da, *db, *dc, *dd;
    hipMalloc(&da, T * 16 * 4); hipMalloc(&db, T * 16 * 4); hipMalloc(&dc, T * 1024 * 4); hipMalloc(&dd, T * 1024 * 4);
    std::vector<float> a(T * 16), b(T * 16), c(T * 1024), d(T * 1024);
    struct Case { const char* name; float pscale; bool same_sign; float cscale; };
    // pscale: magnitude of a product relative to |c| ~ cscale
    const Case cases[] = {
        {"products ~ c, random signs", 1.f, false, 1.f},
        {"products ~ c, all positive", 1.f, true, 1.f},
        {"products ~ 2^-8 c (correction planes), random signs", 1.f / 256, false, 1.f},
        {"products ~ 2^-8 c, all positive", 1.f / 256, true, 1.f},
        {"products ~ 2^-16 c, all positive", 1.f / 65536, true, 1.f},
        {"c = 0 (first step of a chain)", 1.f, false, 0.f},
        {"c ~ 64 x products (long chain), all positive", 1.f, true, 64.f},
        {"c ~ 1024 x products (long chain), all positive", 1.f, true, 1024.f},
    };
    const char* modes[] = {"v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16", "8 x v_mfma_f32_32x32x2_f32"};
    printf("{\"trials\": %d, \"results\": [\n", T * 1024);
    bool first = true;
    for (int mode = 0; mode < 3; ++mode)
        for (const Case& cs : cases) {
            for (int i = 0; i < T * 16; ++i) {
                float x = nd(rng), y = nd(rng) * cs.pscale;
                if (cs.same_sign) { x = fabsf(x); y = fabsf(y); }
                if (mode == 0) { x = to_bf16(x); y = to_bf16(y); }
                else if (mode == 1) { x = to_f16(x); y = to_f16(y); }
                else { x = to_bf16(x); y = to_bf16(y); }      // exact products for the fp32 pipe too
                a[i] = x; b[i] = y;
            }
            for (int i = 0; i < T * 1024; ++i) c[i] = cs.cscale * (cs.same_sign ? fabsf(nd(rng)) + 0.5f : nd(rng));
            hipMemcpy(da, a.data(), T * 16 * 4, hipMemcpyHostToDevice);
            hipMemcpy(db, b.data(), T * 16 * 4, hipMemcpyHostToDevice);
            hipMemcpy(dc, c.data(), T * 1024 * 4, hipMemcpyHostToDevice);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(T), dim3(64), 0, 0, da, db, dc, dd);
            else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(T), dim3(64), 0, 0, da, db, dc, dd);
            else hipLaunchKernelGGL(probe<2>, dim3(T), dim3(64), 0, 0, da, db, dc, dd);
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
            hipMemcpy(d.data(), dd, T * 1024 * 4, hipMemcpyDeviceToHost);
            Stat s;
            for (int t = 0; t < T; ++t) {
                long double dot = 0;
                for (int k = 0; k < 16; ++k) dot += (long double)a[t * 16 + k] * (long double)b[t * 16 + k];
                for (int i = 0; i < 1024; ++i) {
                    const long double ex = dot + (long double)c[t * 1024 + i];
                    const float got = d[t * 1024 + i], rn = (float)ex, tz = rz(ex);
                    const float ulp = fabsf(nextafterf(fabsf(rn), INFINITY) - fabsf(rn));
                    ++s.n;
                    if (got == rn && got == tz) ++s.both;
                    else if (got == rn) ++s.rn;
                    else if (got == tz) ++s.tz;
                    else ++s.other;
                    const double e = (double)(((long double)got - ex) / ulp) * (ex < 0 ? -1.0 : 1.0);   // > 0: away from zero
                    s.serr += e; s.aerr += fabs(e);
                }
            }
            printf("%s {\"mode\": \"%s\", \"case\": \"%s\", \"exact_either\": %.4f, \"only_nearest\": %.4f, \"only_truncation\": %.4f, "
                   "\"neither\": %.4f, \"mean_signed_err_ulp_away_from_zero\": %.4f, \"mean_abs_err_ulp\": %.4f}",
                   first ? " " : ",\n ", modes[mode], cs.name, (double)s.both / s.n, (double)s.rn / s.n, (double)s.tz / s.n,
                   (double)s.other / s.n, s.serr / s.n, s.aerr / s.n);
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
