// Optimizer step of the training loop on ONE flat fp32 bucket (reference: libs/utils_ft.py:676-681 --
// nn.utils.clip_grad_norm_(model.parameters(), grad_clip); optimizer.step() with torch.optim.Adam): global gradient
// norm -> clip coefficient -> Adam moments and parameter update, three launches for the whole model instead of a norm
// / scale / Adam multi-tensor pass per call.  HBM-bound: the step reads g, p, m, v once and writes p, m, v once.
// Everything the host would have to read back (norm, step count, learning rate) stays in device memory, so the step is
// graph-capturable and the same captured launch serves every iteration.
#include <algorithm>
#include <cmath>

#include "gt_common.h"

namespace gt {

constexpr int OPT_BLOCKS = 1024;        // partial sums of the norm pass (fixed: deterministic two-pass reduction)

// partial[b] = sum over this block's grid-stride range of g[i]^2  (double accumulation inside a thread's chain would
// cost nothing here, but torch's foreach norm is fp32 too; the fixed order makes replays bit-identical)
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n,
                                                             float* __restrict__ partial) {
    __shared__ float red[4];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int64_t n4 = n >> 2;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = g4[i];
        s0 = fmaf(v[0], v[0], s0); s1 = fmaf(v[1], v[1], s1); s2 = fmaf(v[2], v[2], s2); s3 = fmaf(v[3], v[3], s3);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        s0 = fmaf(v, v, s0);
    }
    float s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = scale^2 * sum of the partials (one block, fixed order)
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partial, int nblk, float scale,
                                                           float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = scale * scale * ((red[0] + red[1]) + (red[2] + red[3]));
}

struct AdamP {
    float* p; const float* g; float* m; float* v; int64_t n;
    const float* sqnorm; float gscale, max_norm;
    const float* lr; float beta1, beta2, eps, weight_decay;
    const uint64_t* step; const float* beta1_dev;
};

__global__ __launch_bounds__(256) void adam_clip_kernel(const AdamP a) {
    __shared__ float sh[4];
    const float beta1 = a.beta1_dev ? a.beta1_dev[0] : a.beta1;      // OneCycleLR cycles Adam's beta1 with the rate
    if (threadIdx.x == 0) {
        // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to <= 1
        float coef = a.gscale;
        if (a.max_norm > 0.f) coef *= fminf(1.f, a.max_norm / (sqrtf(a.sqnorm[0]) + 1e-6f));
        const double t = (double)a.step[0];
        const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)a.beta2, t);
        sh[0] = coef;
        sh[1] = (float)((double)a.lr[0] / bc1);           // step_size
        sh[2] = (float)(1.0 / sqrt(bc2));                 // 1 / sqrt(bias_correction2)
    }
    __syncthreads();
    const float coef = sh[0], step_size = sh[1], rbc2 = sh[2];
    const float b1 = beta1, b2 = a.beta2, ob1 = 1.f - beta1, ob2 = 1.f - a.beta2;
    const int64_t n4 = a.n >> 2;
    f32x4* p4 = reinterpret_cast<f32x4*>(a.p);
    f32x4* m4 = reinterpret_cast<f32x4*>(a.m);
    f32x4* v4 = reinterpret_cast<f32x4*>(a.v);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(a.g);
    auto upd = [&](float& p, float g, float& m, float& v) {
        g = g * coef + a.weight_decay * p;
        m = b1 * m + ob1 * g;                              // lerp(m, g, 1 - beta1)
        v = b2 * v + ob2 * g * g;
        p -= step_size * (m / (sqrtf(v) * rbc2 + a.eps));
    };
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 p = p4[i], m = m4[i], v = v4[i];
        const f32x4 g = g4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = p[j], mj = m[j], vj = v[j];
            upd(pj, g[j], mj, vj);
            p[j] = pj; m[j] = mj; v[j] = vj;
        }
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        upd(p, a.g[i], m, v);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
}

}  // namespace gt

using namespace gt;

extern "C" int64_t gt_grad_sqnorm_ws_bytes(void) { return OPT_BLOCKS * (int64_t)sizeof(float); }

extern "C" int gt_grad_sqnorm(const float* g, int64_t n, float scale, float* out, void* ws, int64_t ws_bytes,
                              void* stream) {
    if (!g || !out || n <= 0) return GT_EINVAL;
    if (reinterpret_cast<uintptr_t>(g) & 15) return GT_EALIGN;
    if (!ws || ws_bytes < gt_grad_sqnorm_ws_bytes()) return GT_EWS;
    float* partial = reinterpret_cast<float*>(ws);
    const int nblk = (int)std::min<int64_t>(OPT_BLOCKS, std::max<int64_t>(1, ((n >> 2) + 255) / 256));
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, n, partial);
    GT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, scale, out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sqnorm,
                                 float gscale, float max_norm, const float* lr, float beta1, float beta2, float eps,
                                 float weight_decay, const uint64_t* step, const float* beta1_dev, void* stream) {
    if (!p || !g || !m || !v || !lr || !step || n <= 0) return GT_EINVAL;
    if (max_norm > 0.f && !sqnorm) return GT_EINVAL;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return GT_EALIGN;
    const AdamP a{p, g, m, v, n, sqnorm, gscale, max_norm, lr, beta1, beta2, eps, weight_decay, step, beta1_dev};
    const int nblk = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ((n >> 2) + 255) / 256));
    hipLaunchKernelGGL(adam_clip_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    GT_LAUNCH_CHECK();
    return 0;
}
