// Shared device helpers for libgt_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gt_hip.h"

namespace gt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------
// Stateless dropout RNG.  key = mix(seed, salt) is wave-uniform (computed once per kernel),
// the per-element part is a murmur3 finaliser over (idx * golden + key).
// ------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__host__ __device__ inline uint32_t drop_key(uint64_t seed, uint32_t salt) {
    uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    return fmix32(lo ^ fmix32(hi + 0x9e3779b9u * (salt + 1u)));
}
__host__ __device__ inline uint32_t drop_hash(uint32_t key, uint32_t idx) {
    return fmix32(idx * 0x9e3779b1u + key);
}
// p in (0,1): threshold such that keep <=> hash >= thresh
__host__ inline uint32_t drop_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0) t = 0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

struct DropDev {          // device-side view of gt_dropout
    uint32_t thresh;      // 0 => disabled
    uint32_t salt;
    float scale;          // 1/(1-p) (possibly signed)
    const uint64_t* seed;
};
inline DropDev make_drop(const gt_dropout* d, float sign = 1.f) {
    DropDev r{0u, 0u, sign, nullptr};
    if (d && d->p > 0.f) {
        r.thresh = drop_thresh(d->p);
        r.salt = d->salt;
        r.scale = sign / (1.f - d->p);
        r.seed = d->seed;
    }
    return r;
}
__device__ inline uint32_t drop_key_dev(const DropDev& d) {
    return d.thresh ? drop_key(*d.seed, d.salt) : 0u;
}
__device__ inline float drop_mul(const DropDev& d, uint32_t key, uint32_t idx) {
    return (drop_hash(key, idx) >= d.thresh) ? d.scale : 0.f;
}

// Two fp32 values times a power of two s -> packed fp16 heads h0 = f16(x s) and residuals h1 = f16(x s - h0), in FOUR instructions:
// v_fma_mix{lo,hi}_f16 evaluate the fp32 fma and round it (RNE, gradual underflow) into one half of the destination -- no
// separate multiply, no conversions.  Bit-identical to  r = x s; h0 = cvt_pk_f16_f32(r); h1 = cvt_pk_f16_f32(r - f32(h0))
// (eight issue slots per pair: a packed fp32 instruction costs two, tools/valu_rate_probe.hip) except for the sign of exact
// zeros (tools/fma_mix_split_probe.hip, checked on gfx950 incl. fp16-subnormal heads and residuals).
// The operands must not be fresh MFMA results: hipcc's hazard recogniser does not look into inline asm (gt_fourier16.hip).
__device__ __forceinline__ void f16_mulsplit_pair(float a0, float m0, float a1, float m1, uint32_t& hi, uint32_t& lo) {
    uint32_t h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(m0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(m1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(m0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(m1), "v"(h));
    hi = h;
    lo = l;
}

// SiLU on the hardware exp2 / rcp units (1 ulp each; 5 VALU ops instead of ~27 for expf + IEEE division).
__device__ inline float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ inline float silu_f(float x) { return x * sigmoid_f(x); }
__device__ inline float dsilu_f(float x) {
    const float s = sigmoid_f(x);
    return s * (1.f + x * (1.f - s));
}
__device__ inline void silu_both(float x, float& a, float& da) {   // value and derivative from one sigmoid
    const float s = sigmoid_f(x);
    a = x * s;
    da = s * (1.f + x * (1.f - s));
}

// erf GELU (torch.nn.GELU's default, reference layers.py:968): value Phi(x) x and derivative Phi(x) + x phi(x)
__device__ inline void gelu_both(float x, float& a, float& da) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    a = x * cdf;
    da = cdf + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// value of lane (l ^ 32): v_permlane32_swap (gfx950) exchanges the wave's halves on the VALU -- no LDS round trip and no
// s_waitcnt as with __shfl_xor(v, 32) = ds_bpermute_b32.  Returns (own half's values, other half's values) merged per lane.
__device__ __forceinline__ float xor32(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // sw[0]: lanes 0..31 everywhere, sw[1]: lanes 32..63
    return __uint_as_float((threadIdx.x & 32) ? sw[0] : sw[1]);
}
__device__ __forceinline__ float xor32_max(float v) {                       // max(v[l], v[l ^ 32]) without the select
    const uint32_t u = __float_as_uint(v);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}
__device__ __forceinline__ float xor32_sum(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Wave64 sum on the DPP path: six v_add_f32 with a lane-permuted operand, no LDS traffic and no waits (a __shfl_xor step is
// a ds_bpermute_b32 + s_waitcnt).  The TOTAL ends up in lane 63 only (quad swaps, row shifts by 4 and 8, then the row
// broadcasts 15 / 31 of the gfx9 DPP set); for kernels that reduce many values per lane and let one lane store them.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v) {      // the permuted operand; 0 where a lane has no source / is masked
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_term<0xb1, 0xf>(v);      // quad_perm [1,0,3,2]
    v += dpp_term<0x4e, 0xf>(v);      // quad_perm [2,3,0,1]
    v += dpp_term<0x114, 0xf>(v);     // row_shr:4
    v += dpp_term<0x118, 0xf>(v);     // row_shr:8   -> lanes 12..15 of a row hold the row sum
    v += dpp_term<0x142, 0xa>(v);     // row_bcast:15 into rows 1, 3
    v += dpp_term<0x143, 0xc>(v);     // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave sum
    return v;
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// tall-skinny weight-gradient path of gt_gemm (gt_tsmm.hip)
bool tsmm_eligible(const gt_gemm_desc* d);
int64_t tsmm_ws_bytes(const gt_gemm_desc* d);
const char* tsmm_kernel_name(const gt_gemm_desc* d);
int tsmm_run(const gt_gemm_desc* d, void* ws, int64_t ws_bytes, void* stream);

#define GT_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

}  // namespace gt
