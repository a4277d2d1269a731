// Bilinear resize (align_corners=True) of the CNN down/up-scalers, forward and backward, with the
// NCHW <-> NHWC layout change of the scaler boundaries fused in (reference: F.interpolate at
// libs/layers.py:483-512, 658-670; the permutes at libs/model.py:675-687, 740-749).
//
// HBM-bound: every input element is read once from HBM (the 4 taps of neighbouring outputs hit L1/L2),
// every output element is written once.  A block owns a [64 channels] x [32 x-positions] tile of one
// output row; when the input and output layouts differ the tile is transposed through LDS so both the
// loads and the stores stay coalesced (x-contiguous for NCHW, channel-contiguous float4 for NHWC).
// Backward is a gather over the (contiguous) range of outputs that touch an input pixel: no atomics,
// so it is deterministic (the reference warns that F.interpolate's backward is not,
// examples/README.md:6-7).
#include "gt_common.h"
#include <algorithm>

namespace gt {

constexpr int RS_TC = 64;   // channels per tile
constexpr int RS_TX = 32;   // x positions per tile

struct Axis {               // source index / weights of one output coordinate (torch's align_corners rule)
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Axis axis_of(int o, float scale, int ni) {
    // The reference rounds scale*o to fp32 before taking floor and fraction.  Letting the compiler
    // contract `scale*o - i0` into one fma changes the weights by up to 1 ulp of src (~1e-5 relative at
    // o ~ 100), so contraction is switched off for this function.
#pragma clang fp contract(off)
    const float src = scale * (float)o;
    int i0 = (int)src;
    i0 = min(i0, ni - 1);
    Axis a;
    a.i0 = i0;
    a.i1 = i0 + (i0 < ni - 1 ? 1 : 0);
    a.l1 = src - (float)i0;
    a.l0 = 1.f - a.l1;
    return a;
}
// first output index whose i0 can reach i-1 (conservative estimate, fixed up by the caller's loop)
__device__ __forceinline__ int first_out(int i, float scale, int no) {
    if (scale <= 0.f || i <= 1) return 0;
    int o = (int)((float)(i - 1) / scale) - 2;
    return max(0, min(o, no));
}

struct ResizeP {
    const float* x; float* y;
    const float* gate;          // fwd: unused.  bwd: saved activated output (ReLU gate on g), may be null
    int B, C, Hi, Wi, Ho, Wo;
    float sy, sx;
    int act;                    // fwd: GT_ACT_NONE / GT_ACT_RELU applied to the output
    int xtiles;
};

template <bool NHWC>
__device__ __forceinline__ int64_t addr(int b, int c, int y, int x, int C, int H, int W) {
    return NHWC ? (((int64_t)b * H + y) * W + x) * C + c : (((int64_t)b * C + c) * H + y) * W + x;
}

// ------------------------------------------------------------------------------------------ forward
template <bool IN_NHWC, bool OUT_NHWC>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const ResizeP p) {
    __shared__ float tile[RS_TC][RS_TX + 1];
    const int t = threadIdx.x;
    const int xt = blockIdx.x % p.xtiles, ct = blockIdx.x / p.xtiles;
    const int ox0 = xt * RS_TX, c0 = ct * RS_TC, oy = blockIdx.y, b = blockIdx.z;
    const Axis ay = axis_of(oy, p.sy, p.Hi);

    if (!IN_NHWC) {
        const int ox_l = t & 31, cg = t >> 5;
        const int ox = ox0 + ox_l;
        if (ox < p.Wo) {
            const Axis ax = axis_of(ox, p.sx, p.Wi);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c_l = cg + 8 * k, c = c0 + c_l;
                if (c < p.C) {
                    const float* r0 = p.x + addr<false>(b, c, ay.i0, 0, p.C, p.Hi, p.Wi);
                    const float* r1 = p.x + addr<false>(b, c, ay.i1, 0, p.C, p.Hi, p.Wi);
                    float v = ay.l0 * (ax.l0 * r0[ax.i0] + ax.l1 * r0[ax.i1]) +
                              ay.l1 * (ax.l0 * r1[ax.i0] + ax.l1 * r1[ax.i1]);
                    if (p.act == GT_ACT_RELU) v = fmaxf(v, 0.f);
                    if (!OUT_NHWC) p.y[addr<false>(b, c, oy, ox, p.C, p.Ho, p.Wo)] = v;
                    else tile[c_l][ox_l] = v;
                }
            }
        }
    } else {
        const int c_l = (t & 15) * 4, xg = t >> 4;
        const int c = c0 + c_l;
        if (c < p.C) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ox_l = xg + 16 * k, ox = ox0 + ox_l;
                if (ox < p.Wo) {
                    const Axis ax = axis_of(ox, p.sx, p.Wi);
                    const f32x4 v00 = *reinterpret_cast<const f32x4*>(p.x + addr<true>(b, c, ay.i0, ax.i0, p.C, p.Hi, p.Wi));
                    const f32x4 v01 = *reinterpret_cast<const f32x4*>(p.x + addr<true>(b, c, ay.i0, ax.i1, p.C, p.Hi, p.Wi));
                    const f32x4 v10 = *reinterpret_cast<const f32x4*>(p.x + addr<true>(b, c, ay.i1, ax.i0, p.C, p.Hi, p.Wi));
                    const f32x4 v11 = *reinterpret_cast<const f32x4*>(p.x + addr<true>(b, c, ay.i1, ax.i1, p.C, p.Hi, p.Wi));
                    f32x4 v = ay.l0 * (ax.l0 * v00 + ax.l1 * v01) + ay.l1 * (ax.l0 * v10 + ax.l1 * v11);
                    if (p.act == GT_ACT_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    if (OUT_NHWC) {
                        *reinterpret_cast<f32x4*>(p.y + addr<true>(b, c, oy, ox, p.C, p.Ho, p.Wo)) = v;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) tile[c_l + j][ox_l] = v[j];
                    }
                }
            }
        }
    }
    if (IN_NHWC == OUT_NHWC) return;
    __syncthreads();
    if (OUT_NHWC) {          // tile -> NHWC float4 stores
        const int c_l = (t & 15) * 4, xg = t >> 4;
        const int c = c0 + c_l;
        if (c < p.C) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ox_l = xg + 16 * k, ox = ox0 + ox_l;
                if (ox < p.Wo) {
                    const f32x4 v = {tile[c_l][ox_l], tile[c_l + 1][ox_l], tile[c_l + 2][ox_l], tile[c_l + 3][ox_l]};
                    *reinterpret_cast<f32x4*>(p.y + addr<true>(b, c, oy, ox, p.C, p.Ho, p.Wo)) = v;
                }
            }
        }
    } else {                  // tile -> NCHW x-contiguous stores
        const int ox_l = t & 31, cg = t >> 5;
        const int ox = ox0 + ox_l;
        if (ox < p.Wo) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c_l = cg + 8 * k, c = c0 + c_l;
                if (c < p.C) p.y[addr<false>(b, c, oy, ox, p.C, p.Ho, p.Wo)] = tile[c_l][ox_l];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
// p.x = upstream gradient g (output-shaped, layout G_NHWC), p.y = dx (input-shaped, layout DX_NHWC),
// p.gate = saved activated forward output (same shape/layout as g) or null.  Hi/Wi are the sizes of
// the FORWARD input (= dx), Ho/Wo of the forward output (= g).  blockIdx.y = input row iy.
template <bool G_NHWC, bool DX_NHWC>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const ResizeP p) {
    __shared__ float tile[RS_TC][RS_TX + 1];
    const int t = threadIdx.x;
    const int xt = blockIdx.x % p.xtiles, ct = blockIdx.x / p.xtiles;
    const int ix0 = xt * RS_TX, c0 = ct * RS_TC, iy = blockIdx.y, b = blockIdx.z;

    // output rows touching iy: i0(oy) in {iy-1, iy}
    int oy_lo = first_out(iy, p.sy, p.Ho);
    while (oy_lo < p.Ho && axis_of(oy_lo, p.sy, p.Hi).i0 < iy - 1) ++oy_lo;
    int oy_hi = oy_lo;
    while (oy_hi < p.Ho && axis_of(oy_hi, p.sy, p.Hi).i0 <= iy) ++oy_hi;

    if (!G_NHWC) {
        const int ix_l = t & 31, cg = t >> 5;
        const int ix = ix0 + ix_l;
        if (ix < p.Wi) {
            int ox_lo = first_out(ix, p.sx, p.Wo);
            while (ox_lo < p.Wo && axis_of(ox_lo, p.sx, p.Wi).i0 < ix - 1) ++ox_lo;
            int ox_hi = ox_lo;
            while (ox_hi < p.Wo && axis_of(ox_hi, p.sx, p.Wi).i0 <= ix) ++ox_hi;
#pragma unroll 1
            for (int k = 0; k < 8; ++k) {
                const int c_l = cg + 8 * k, c = c0 + c_l;
                if (c >= p.C) break;
                float acc = 0.f;
                for (int oy = oy_lo; oy < oy_hi; ++oy) {
                    const Axis ay = axis_of(oy, p.sy, p.Hi);
                    const float wy = (ay.i0 == iy ? ay.l0 : 0.f) + (ay.i1 == iy ? ay.l1 : 0.f);
                    const int64_t ro = addr<false>(b, c, oy, 0, p.C, p.Ho, p.Wo);
                    float racc = 0.f;
                    for (int ox = ox_lo; ox < ox_hi; ++ox) {
                        const Axis ax = axis_of(ox, p.sx, p.Wi);
                        const float wx = (ax.i0 == ix ? ax.l0 : 0.f) + (ax.i1 == ix ? ax.l1 : 0.f);
                        float g = p.x[ro + ox];
                        if (p.gate && !(p.gate[ro + ox] > 0.f)) g = 0.f;
                        racc = fmaf(wx, g, racc);
                    }
                    acc = fmaf(wy, racc, acc);
                }
                if (!DX_NHWC) p.y[addr<false>(b, c, iy, ix, p.C, p.Hi, p.Wi)] = acc;
                else tile[c_l][ix_l] = acc;
            }
        }
    } else {
        const int c_l = (t & 15) * 4, xg = t >> 4;
        const int c = c0 + c_l;
        if (c < p.C) {
#pragma unroll 1
            for (int k = 0; k < 2; ++k) {
                const int ix_l = xg + 16 * k, ix = ix0 + ix_l;
                if (ix >= p.Wi) break;
                int ox_lo = first_out(ix, p.sx, p.Wo);
                while (ox_lo < p.Wo && axis_of(ox_lo, p.sx, p.Wi).i0 < ix - 1) ++ox_lo;
                int ox_hi = ox_lo;
                while (ox_hi < p.Wo && axis_of(ox_hi, p.sx, p.Wi).i0 <= ix) ++ox_hi;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int oy = oy_lo; oy < oy_hi; ++oy) {
                    const Axis ay = axis_of(oy, p.sy, p.Hi);
                    const float wy = (ay.i0 == iy ? ay.l0 : 0.f) + (ay.i1 == iy ? ay.l1 : 0.f);
                    f32x4 racc = {0.f, 0.f, 0.f, 0.f};
                    for (int ox = ox_lo; ox < ox_hi; ++ox) {
                        const Axis ax = axis_of(ox, p.sx, p.Wi);
                        const float wx = (ax.i0 == ix ? ax.l0 : 0.f) + (ax.i1 == ix ? ax.l1 : 0.f);
                        const int64_t o = addr<true>(b, c, oy, ox, p.C, p.Ho, p.Wo);
                        f32x4 g = *reinterpret_cast<const f32x4*>(p.x + o);
                        if (p.gate) {
                            const f32x4 y = *reinterpret_cast<const f32x4*>(p.gate + o);
#pragma unroll
                            for (int j = 0; j < 4; ++j) if (!(y[j] > 0.f)) g[j] = 0.f;
                        }
                        racc += wx * g;
                    }
                    acc += wy * racc;
                }
                if (DX_NHWC) {
                    *reinterpret_cast<f32x4*>(p.y + addr<true>(b, c, iy, ix, p.C, p.Hi, p.Wi)) = acc;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) tile[c_l + j][ix_l] = acc[j];
                }
            }
        }
    }
    if (G_NHWC == DX_NHWC) return;
    __syncthreads();
    if (DX_NHWC) {
        const int c_l = (t & 15) * 4, xg = t >> 4;
        const int c = c0 + c_l;
        if (c < p.C) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ix_l = xg + 16 * k, ix = ix0 + ix_l;
                if (ix < p.Wi) {
                    const f32x4 v = {tile[c_l][ix_l], tile[c_l + 1][ix_l], tile[c_l + 2][ix_l], tile[c_l + 3][ix_l]};
                    *reinterpret_cast<f32x4*>(p.y + addr<true>(b, c, iy, ix, p.C, p.Hi, p.Wi)) = v;
                }
            }
        }
    } else {
        const int ix_l = t & 31, cg = t >> 5;
        const int ix = ix0 + ix_l;
        if (ix < p.Wi) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c_l = cg + 8 * k, c = c0 + c_l;
                if (c < p.C) p.y[addr<false>(b, c, iy, ix, p.C, p.Hi, p.Wi)] = tile[c_l][ix_l];
            }
        }
    }
}

static int check_resize(const void* x, const void* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                        int in_nhwc, int out_nhwc) {
    if (!x || !y || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return GT_EINVAL;
    if ((in_nhwc | out_nhwc) & ~1) return GT_EINVAL;
    if (B > 65535 || Ho > 65535 || Hi > 65535) return GT_EINVAL;
    if ((in_nhwc || out_nhwc) && (C & 3)) return GT_ENOTSUP;          // NHWC side moves float4 channel groups
    if (in_nhwc && (reinterpret_cast<uintptr_t>(x) & 15)) return GT_EALIGN;
    if (out_nhwc && (reinterpret_cast<uintptr_t>(y) & 15)) return GT_EALIGN;
    return 0;
}
static inline float scale_of(int ni, int no) { return (no > 1) ? (float)(ni - 1) / (float)(no - 1) : 0.f; }

}  // namespace gt

using namespace gt;

extern "C" int gt_bilinear2d_fwd(const float* x, float* y, int32_t B, int32_t C, int32_t Hi, int32_t Wi,
                                 int32_t Ho, int32_t Wo, int32_t in_nhwc, int32_t out_nhwc, int32_t act,
                                 void* stream) {
    if (int rc = check_resize(x, y, B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc)) return rc;
    if (act != GT_ACT_NONE && act != GT_ACT_RELU) return GT_ENOTSUP;
    ResizeP p{x, y, nullptr, B, C, Hi, Wi, Ho, Wo, scale_of(Hi, Ho), scale_of(Wi, Wo), act, ceil_div(Wo, RS_TX)};
    dim3 grid((unsigned)(p.xtiles * ceil_div(C, RS_TC)), (unsigned)Ho, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (!in_nhwc && !out_nhwc) hipLaunchKernelGGL((resize_fwd_kernel<false, false>), grid, dim3(256), 0, st, p);
    else if (!in_nhwc && out_nhwc) hipLaunchKernelGGL((resize_fwd_kernel<false, true>), grid, dim3(256), 0, st, p);
    else if (in_nhwc && !out_nhwc) hipLaunchKernelGGL((resize_fwd_kernel<true, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((resize_fwd_kernel<true, true>), grid, dim3(256), 0, st, p);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_bilinear2d_bwd(const float* g, const float* y_saved, float* dx, int32_t B, int32_t C,
                                 int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t in_nhwc,
                                 int32_t out_nhwc, int32_t act, void* stream) {
    // g (and y_saved) have the forward OUTPUT shape/layout, dx the forward INPUT shape/layout
    if (int rc = check_resize(dx, g, B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc)) return rc;
    if (act != GT_ACT_NONE && act != GT_ACT_RELU) return GT_ENOTSUP;
    if (act == GT_ACT_RELU && !y_saved) return GT_EINVAL;
    if (out_nhwc && y_saved && (reinterpret_cast<uintptr_t>(y_saved) & 15)) return GT_EALIGN;
    ResizeP p{g, dx, act == GT_ACT_RELU ? y_saved : nullptr, B, C, Hi, Wi, Ho, Wo, scale_of(Hi, Ho),
              scale_of(Wi, Wo), act, ceil_div(Wi, RS_TX)};
    dim3 grid((unsigned)(p.xtiles * ceil_div(C, RS_TC)), (unsigned)Hi, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (!out_nhwc && !in_nhwc) hipLaunchKernelGGL((resize_bwd_kernel<false, false>), grid, dim3(256), 0, st, p);
    else if (!out_nhwc && in_nhwc) hipLaunchKernelGGL((resize_bwd_kernel<false, true>), grid, dim3(256), 0, st, p);
    else if (out_nhwc && !in_nhwc) hipLaunchKernelGGL((resize_bwd_kernel<true, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((resize_bwd_kernel<true, true>), grid, dim3(256), 0, st, p);
    GT_LAUNCH_CHECK();
    return 0;
}
