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
#include <cstdlib>
#include <cstring>
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
    // optional affine term added to the resized value of channel c at output pixel q (NHWC outputs only):
    //   + bias[c] + sum_j rp_a[q*rp_lda + j] * rp_b[c*rp_ldb + j]
    const float* bias; int rp; const float* rp_a; int64_t rp_lda; const float* rp_b; int64_t rp_ldb;
    // channels-last kernels only: the INPUT side (fwd: x, bwd: dx) is a padded concatenation of three column segments of
    // segp channels each (ops.scaler_conv_chain): real channel c lives at padded column c + (segp - seg) * min(c / seg, 2);
    // seg == 0: dense.
    int seg, segp;
    // backward with segments only: the forward input itself (padded layout, the output of a ReLU): dx is zeroed where it
    // is not positive, which is the first step of its producer's backward (ops.ScalerConvChainFn) done on the way out
    const float* in_gate;
    // act == GT_ACT_SILU (channels-last kernels): fwd writes silu'(resized value) here, bwd reads it through `gate` as a factor
    float* dact;
    int gate_mul;               // bwd: in_gate is a factor (dx *= in_gate) instead of the ReLU test
};

// padded column of real channel c
__device__ __forceinline__ int seg_col(const ResizeP& p, int c) { return c + (p.segp - p.seg) * min(c / p.seg, 2); }
// 4 consecutive real channels c .. c+3 of the padded-segment pixel at px (floats)
// (c is a multiple of 4; seg and segp are even, so the pairs (c, c+1) and (c+2, c+3) never straddle a segment: two 8-byte loads)
__device__ __forceinline__ f32x4 seg_load4(const ResizeP& p, const float* __restrict__ px, int c) {
    const f32x2 a = *reinterpret_cast<const f32x2*>(px + seg_col(p, c));
    const f32x2 b = *reinterpret_cast<const f32x2*>(px + seg_col(p, c + 2));
    return f32x4{a[0], a[1], b[0], b[1]};
}

__device__ __forceinline__ f32x4 resize_affine(const ResizeP& p, f32x4 v, int b, int c, int oy, int ox) {
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + c);
    if (p.rp) {
        const float* ga = p.rp_a + (((int64_t)b * p.Ho + oy) * p.Wo + ox) * p.rp_lda;
        for (int j = 0; j < p.rp; ++j) {
            const float a = ga[j];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = fmaf(a, p.rp_b[(int64_t)(c + t) * p.rp_ldb + j], v[t]);
        }
    }
    return v;
}

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
                    if (OUT_NHWC) v = resize_affine(p, v, b, c, oy, ox);
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
//
// The outputs that touch input index i form a contiguous range; their weights are gathered once per
// thread (x) / per block (y) into a small register table, so the channel loop is pure load + fma.
constexpr int RS_MAXT = 6;          // taps per axis held in registers (covers up-sampling factors < 2.5)
struct Taps {
    int lo, n;
    float w[RS_MAXT];
};
__device__ __forceinline__ Taps taps_of(int i, float scale, int ni, int no) {
    Taps t;
    int lo = first_out(i, scale, no);
    while (lo < no && axis_of(lo, scale, ni).i0 < i - 1) ++lo;
    t.lo = lo;
    t.n = 0;
#pragma unroll
    for (int j = 0; j < RS_MAXT; ++j) t.w[j] = 0.f;
#pragma unroll
    for (int j = 0; j < RS_MAXT; ++j) {
        const int o = lo + j;
        if (o < no) {
            const Axis a = axis_of(o, scale, ni);
            if (a.i0 <= i) {
                t.w[j] = (a.i0 == i ? a.l0 : 0.f) + (a.i1 == i ? a.l1 : 0.f);
                t.n = j + 1;
            }
        }
    }
    // more than RS_MAXT contributing outputs (very strong up-sampling): flag with n = -1
    if (lo + RS_MAXT < no && axis_of(lo + RS_MAXT, scale, ni).i0 <= i) t.n = -1;
    return t;
}

template <bool G_NHWC, bool DX_NHWC>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const ResizeP p) {
    __shared__ float tile[RS_TC][RS_TX + 1];
    const int t = threadIdx.x;
    const int xt = blockIdx.x % p.xtiles, ct = blockIdx.x / p.xtiles;
    const int ix0 = xt * RS_TX, c0 = ct * RS_TC, iy = blockIdx.y, b = blockIdx.z;
    const Taps ty = taps_of(iy, p.sy, p.Hi, p.Ho);

    if (!G_NHWC) {
        const int ix_l = t & 31, cg = t >> 5;
        const int ix = ix0 + ix_l;
        if (ix < p.Wi) {
            const Taps tx = taps_of(ix, p.sx, p.Wi, p.Wo);
            if (ty.n >= 0 && tx.n >= 0) {
                float accs[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {           // 8 independent channels in flight per lane
                    const int c = c0 + cg + 8 * k;
                    float acc = 0.f;
                    if (c < p.C) {
#pragma unroll
                        for (int jy = 0; jy < RS_MAXT; ++jy) {
                            if (jy < ty.n) {
                                const int64_t ro = addr<false>(b, c, ty.lo + jy, tx.lo, p.C, p.Ho, p.Wo);
                                float racc = 0.f;
#pragma unroll
                                for (int jx = 0; jx < RS_MAXT; ++jx) {
                                    if (jx < tx.n) {
                                        float g = p.x[ro + jx];
                                        if (p.gate && !(p.gate[ro + jx] > 0.f)) g = 0.f;
                                        racc = fmaf(tx.w[jx], g, racc);
                                    }
                                }
                                acc = fmaf(ty.w[jy], racc, acc);
                            }
                        }
                    }
                    accs[k] = acc;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int c_l = cg + 8 * k, c = c0 + c_l;
                    if (c < p.C) {
                        if (!DX_NHWC) p.y[addr<false>(b, c, iy, ix, p.C, p.Hi, p.Wi)] = accs[k];
                        else tile[c_l][ix_l] = accs[k];
                    }
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < 8; ++k) {           // generic path: arbitrary number of taps
                    const int c_l = cg + 8 * k, c = c0 + c_l;
                    if (c >= p.C) break;
                    float acc = 0.f;
                    for (int oy = ty.lo; oy < p.Ho; ++oy) {
                        const Axis ay = axis_of(oy, p.sy, p.Hi);
                        if (ay.i0 > iy) break;
                        const float wy = (ay.i0 == iy ? ay.l0 : 0.f) + (ay.i1 == iy ? ay.l1 : 0.f);
                        const int64_t ro = addr<false>(b, c, oy, 0, p.C, p.Ho, p.Wo);
                        float racc = 0.f;
                        for (int ox = tx.lo; ox < p.Wo; ++ox) {
                            const Axis ax = axis_of(ox, p.sx, p.Wi);
                            if (ax.i0 > ix) break;
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
        }
    } else {
        const int c_l = (t & 15) * 4, xg = t >> 4;
        const int c = c0 + c_l;
        if (c < p.C) {
#pragma unroll 1
            for (int k = 0; k < 2; ++k) {
                const int ix_l = xg + 16 * k, ix = ix0 + ix_l;
                if (ix >= p.Wi) break;
                const Taps tx = taps_of(ix, p.sx, p.Wi, p.Wo);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (ty.n >= 0 && tx.n >= 0) {
#pragma unroll
                    for (int jy = 0; jy < RS_MAXT; ++jy) {
                        if (jy < ty.n) {
                            f32x4 racc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int jx = 0; jx < RS_MAXT; ++jx) {
                                if (jx < tx.n) {
                                    const int64_t o = addr<true>(b, c, ty.lo + jy, tx.lo + jx, p.C, p.Ho, p.Wo);
                                    f32x4 g = *reinterpret_cast<const f32x4*>(p.x + o);
                                    if (p.gate) {
                                        const f32x4 y = *reinterpret_cast<const f32x4*>(p.gate + o);
#pragma unroll
                                        for (int j = 0; j < 4; ++j) if (!(y[j] > 0.f)) g[j] = 0.f;
                                    }
                                    racc += tx.w[jx] * g;
                                }
                            }
                            acc += ty.w[jy] * racc;
                        }
                    }
                } else {
                    for (int oy = ty.lo; oy < p.Ho; ++oy) {
                        const Axis ay = axis_of(oy, p.sy, p.Hi);
                        if (ay.i0 > iy) break;
                        const float wy = (ay.i0 == iy ? ay.l0 : 0.f) + (ay.i1 == iy ? ay.l1 : 0.f);
                        f32x4 racc = {0.f, 0.f, 0.f, 0.f};
                        for (int ox = tx.lo; ox < p.Wo; ++ox) {
                            const Axis ax = axis_of(ox, p.sx, p.Wi);
                            if (ax.i0 > ix) break;
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

// NCHW -> NCHW backward over flattened planes: a thread owns one input pixel (iy, ix) of 8 channel planes,
// so the stores of a wave are 256 contiguous bytes per plane regardless of the (odd) row length.
__global__ __launch_bounds__(256) void resize_bwd_planar_kernel(const ResizeP p) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.Hi * p.Wi) return;
    const int iy = e / p.Wi, ix = e - iy * p.Wi;
    const int b = blockIdx.z, cbase = blockIdx.y * 8;
    const Taps ty = taps_of(iy, p.sy, p.Hi, p.Ho);
    const Taps tx = taps_of(ix, p.sx, p.Wi, p.Wo);
    const int64_t plane_o = (int64_t)p.Ho * p.Wo, plane_i = (int64_t)p.Hi * p.Wi;
    if (ty.n >= 0 && tx.n >= 0) {
        float accs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = cbase + k;
            float acc = 0.f;
            if (c < p.C) {
                const int64_t po = ((int64_t)b * p.C + c) * plane_o;
#pragma unroll
                for (int jy = 0; jy < RS_MAXT; ++jy) {
                    if (jy < ty.n) {
                        const int64_t ro = po + (int64_t)(ty.lo + jy) * p.Wo + tx.lo;
                        float racc = 0.f;
#pragma unroll
                        for (int jx = 0; jx < RS_MAXT; ++jx) {
                            if (jx < tx.n) {
                                float g = p.x[ro + jx];
                                if (p.gate && !(p.gate[ro + jx] > 0.f)) g = 0.f;
                                racc = fmaf(tx.w[jx], g, racc);
                            }
                        }
                        acc = fmaf(ty.w[jy], racc, acc);
                    }
                }
            }
            accs[k] = acc;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (cbase + k < p.C) p.y[((int64_t)b * p.C + cbase + k) * plane_i + e] = accs[k];
    } else {
        for (int k = 0; k < 8 && cbase + k < p.C; ++k) {
            const int64_t po = ((int64_t)b * p.C + cbase + k) * plane_o;
            float acc = 0.f;
            for (int oy = ty.lo; oy < p.Ho; ++oy) {
                const Axis ay = axis_of(oy, p.sy, p.Hi);
                if (ay.i0 > iy) break;
                const float wy = (ay.i0 == iy ? ay.l0 : 0.f) + (ay.i1 == iy ? ay.l1 : 0.f);
                float racc = 0.f;
                for (int ox = tx.lo; ox < p.Wo; ++ox) {
                    const Axis ax = axis_of(ox, p.sx, p.Wi);
                    if (ax.i0 > ix) break;
                    const float wx = (ax.i0 == ix ? ax.l0 : 0.f) + (ax.i1 == ix ? ax.l1 : 0.f);
                    float g = p.x[po + (int64_t)oy * p.Wo + ox];
                    if (p.gate && !(p.gate[po + (int64_t)oy * p.Wo + ox] > 0.f)) g = 0.f;
                    racc = fmaf(wx, g, racc);
                }
                acc = fmaf(wy, racc, acc);
            }
            p.y[((int64_t)b * p.C + cbase + k) * plane_i + e] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------ conv0 + resize
// First stage of the down-scaler fused into one pass (layers.py:483-495: Conv2dResBlock(in -> out, 3x3,
// padding 1, no bias) -> dropout -> act, then F.interpolate -> act):
//     y0[b,c,iy,ix] = relu( keep(b,c,iy,ix) * sum_{ci,dy,dx} W[c,ci,dy,dx] x[b,ci,iy+dy-1,ix+dx-1] )
//     y [b,c,oy,ox] = relu( bilinear(y0)[oy,ox] )
// The input has one (few) channel(s) while y0 has `out` channels at the fine resolution (651 MB at
// 141^2 x 128 x batch 64): y0 is never written -- the conv is re-evaluated at the 4 source pixels of each
// output (36 fma per output and input channel), and in backward at every fine pixel, where the gathered
// gradient is turned straight into the 3x3 weight gradient.  The dropout mask uses the linear NCHW index
// of y0, so the fused op draws exactly the mask the unfused conv -> gt_dropout_apply sequence would.
constexpr int CR_MAXCI = 4;         // input channels supported by the fused path
constexpr int CR_CH = 16;           // output channels per block (forward)

struct ConvResizeP {
    const float* x; const float* w; float* y;          // fwd: y output.  bwd: y = saved forward output
    const float* g; float* partial;                    // bwd only
    int B, Cin, Cout, H, W, Ho, Wo;
    float sy, sx;
    DropDev drop;
    int y_nhwc;                                        // y (and g) channels-last [B, Ho, Wo, Cout] instead of channels-first
    int nstrips;                                       // bwd, channels-last: pixel strips per image (1-D grid, see kernel)
    // channels-last only, optional: the forward's decisions, 4 bits per (output pixel, channel) -- bit t: source pixel t of
    // the bilinear stencil was kept by the dropout AND positive; all four cleared when the resized value itself is <= 0 (its
    // gradient is zero then).  [B][Cout / 16][Ho * Wo] 64-bit words (a wave's 64 pixels are 512 contiguous bytes for the
    // writer and for the reader), nibble c % 16 of word c / 16.  With it the backward neither re-evaluates the convolution
    // nor re-draws the dropout mask, and does not read y.
    unsigned long long* bits;
};

__device__ __forceinline__ void load_patch(const float* __restrict__ xp, int H, int W, int iy, int ix,
                                           float (&pt)[9]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = iy + dy - 1;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = ix + dx - 1;
            pt[dy * 3 + dx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xp[(int64_t)yy * W + xx] : 0.f;
        }
    }
}
// the same patch without branches (the backward requests 36 of these values per pixel in front of a long arithmetic block):
// the load goes to a clamped (valid) address, the select zeroes what lies outside the picture.  The forward is faster with
// the predicated form above (271 vs 353 us at B = 128), the backward with this one.
__device__ __forceinline__ void load_patch_clamped(const float* __restrict__ xp, int H, int W, int iy, int ix,
                                                   float (&pt)[9]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = iy + dy - 1;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = ix + dx - 1;
            const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const float v = xp[yc * W + xc];
            pt[dy * 3 + dx] = (yy == yc && xx == xc) ? v : 0.f;
        }
    }
}

// Parity hook (gt_debug_conv0_mask): when set, the forward records the ReLU decision it takes for every fine-grid value
// of the fused convolution it evaluates -- mask[(b * Cout + c) * H * W + pixel] = 1 (kept and positive) or 0 -- so a
// float64 checker can replay exactly these decisions (tests/test_bench_kernels_gpu.py; pixels no output touches stay as
// the caller initialised them).  One pointer load per thread when unset.
__device__ unsigned char* g_conv0_mask = nullptr;

template <int CIN, int ACT = GT_ACT_RELU>
__global__ __launch_bounds__(256) void conv_resize_fwd_kernel(const ConvResizeP p) {
    __shared__ float sw[CR_CH * CIN * 9];
    // channels-last output: the channel groups of a pixel strip are neighbouring blocks (they complete the strip's
    // 512-byte rows together); channels-first: the pixel strips of a channel group are
    const int bc = p.y_nhwc ? blockIdx.x : blockIdx.y, bx = p.y_nhwc ? blockIdx.y : blockIdx.x;
    const int c0 = bc * CR_CH, b = blockIdx.z;
    for (int i = threadIdx.x; i < CR_CH * CIN * 9; i += 256) {
        const int c = c0 + i / (CIN * 9);
        sw[i] = (c < p.Cout) ? p.w[(int64_t)c * CIN * 9 + i % (CIN * 9)] : 0.f;
    }
    __syncthreads();
    const int e = bx * 256 + threadIdx.x;
    if (e >= p.Ho * p.Wo) return;
    const int oy = e / p.Wo, ox = e - oy * p.Wo;
    const Axis ay = axis_of(oy, p.sy, p.H), ax = axis_of(ox, p.sx, p.W);
    const uint32_t key = drop_key_dev(p.drop);
    unsigned char* const dbg_mask = g_conv0_mask;
    // 3x3 input patches around the 4 source pixels, kept in registers for every output channel
    float pt[CIN][4][9];
    uint32_t toff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int iy = (t & 2) ? ay.i1 : ay.i0, ix = (t & 1) ? ax.i1 : ax.i0;
        toff[t] = (uint32_t)(iy * p.W + ix);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
            load_patch(p.x + ((int64_t)b * CIN + ci) * p.H * p.W, p.H, p.W, iy, ix, pt[ci][t]);
    }
    const uint32_t plane = (uint32_t)(p.H * p.W);
    const float w00 = ay.l0 * ax.l0, w01 = ay.l0 * ax.l1, w10 = ay.l1 * ax.l0, w11 = ay.l1 * ax.l1;
    unsigned long long nib = 0ull;                  // p.bits: the decisions of this pixel's CR_CH = 16 channels
#pragma unroll 1
    for (int j4 = 0; j4 < CR_CH; j4 += 4) {
        if (c0 + j4 >= p.Cout) break;
        float r4[4];
        unsigned n16 = 0u;                          // the four channels' nibbles
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = j4 + jj, c = c0 + j;
            float cv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float wv = sw[(j * CIN + ci) * 9 + k];           // zero for c >= Cout
#pragma unroll
                    for (int t = 0; t < 4; ++t) cv[t] = fmaf(wv, pt[ci][t][k], cv[t]);
                }
            const uint32_t cbase = ((uint32_t)b * (uint32_t)p.Cout + (uint32_t)c) * plane;   // mod 2^32, like the
#pragma unroll                                                                                // stand-alone dropout
            for (int t = 0; t < 4; ++t) {
                const float m = p.drop.thresh ? drop_mul(p.drop, key, cbase + toff[t]) : p.drop.scale;
                if (ACT == GT_ACT_SILU) cv[t] = silu_f(cv[t] * m);
                else {
                    cv[t] = fmaxf(cv[t] * m, 0.f);
                    if (dbg_mask && c < p.Cout) dbg_mask[cbase + toff[t]] = cv[t] > 0.f ? 1 : 0;
                }
            }
            // same association as the stand-alone resize: l0y*(l0x*v00 + l1x*v01) + l1y*(l0x*v10 + l1x*v11)
            const float rz = ay.l0 * (ax.l0 * cv[0] + ax.l1 * cv[1]) + ay.l1 * (ax.l0 * cv[2] + ax.l1 * cv[3]);
            r4[jj] = ACT == GT_ACT_SILU ? silu_f(rz) : fmaxf(rz, 0.f);
            const unsigned d4 = (cv[0] > 0.f ? 1u : 0u) | (cv[1] > 0.f ? 2u : 0u) | (cv[2] > 0.f ? 4u : 0u) | (cv[3] > 0.f ? 8u : 0u);
            n16 |= (r4[jj] > 0.f ? d4 : 0u) << (4 * jj);
        }
        nib |= (unsigned long long)n16 << (4 * j4);
        (void)w00; (void)w01; (void)w10; (void)w11;
        const int c = c0 + j4;
        if (p.y_nhwc) {                       // a pixel's four channels: one 16-byte store (Cout % 4 == 0 checked on the host)
            *reinterpret_cast<f32x4*>(p.y + ((int64_t)b * p.Ho * p.Wo + e) * p.Cout + c) = f32x4{r4[0], r4[1], r4[2], r4[3]};
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (c + jj < p.Cout) p.y[((int64_t)b * p.Cout + c + jj) * p.Ho * p.Wo + e] = r4[jj];
        }
    }
    if (p.bits) p.bits[((int64_t)b * (p.Cout >> 4) + bc) * (p.Ho * p.Wo) + e] = nib;   // CR_CH == 16: one word per thread
}
static_assert(CR_CH == 16, "conv_resize_fwd_kernel packs the decisions of its 16 channels into one 64-bit word");

// Backward: weight gradient only (the fused path is used when the input needs no gradient).
// Output-side formulation: with R the bilinear operator, G = g .* [y > 0], and D = keep .* [y0 > 0],
//     dW[c][ci][k] = sum_px (R^T G)[px] D[px] x[px + off_k]  =  sum_o G[o] * sum_{4 taps t} w_t D[src_t] x[src_t + off_k]
// so a thread walks OUTPUT pixels (coalesced reads of g and y, no gather, the same 3x3 patches as the forward)
// and accumulates dW for CRB_CG channels in registers over CRB_PXT pixels before one block reduction.
constexpr int CRB_PXT = 8;
#ifndef GT_CRB_CG
#define GT_CRB_CG 8
#endif
constexpr int CRB_CG = GT_CRB_CG;
static_assert(CRB_CG % 4 == 0 && CRB_CG >= 4, "the channels-last paths read a pixel's CRB_CG channels as float4 groups");
#ifndef GT_CRB_WAVES                               // resident waves per SIMD the one-channel instance is compiled for
#define GT_CRB_WAVES 2
#endif
template <int CIN, bool BITS = false, int ACT = GT_ACT_RELU>
__global__ __launch_bounds__(256, (CIN == 1 ? GT_CRB_WAVES : 1)) void conv_resize_bwd_kernel(const ConvResizeP p) {
    static_assert(!BITS || CRB_CG == 8, "the recorded decisions are read as one 32-bit half word: eight channels per thread");
    static_assert(!BITS || ACT == GT_ACT_RELU, "decision bits describe ReLUs");
    __shared__ float sw[BITS ? 1 : CRB_CG * CIN * 9];
    __shared__ float red[4][CRB_CG * CIN * 9];
    // channels-first: blockIdx = (pixel strip, channel group).  channels-last: a strip's channel groups read the same
    // 512-byte rows of g and y, 32 bytes each: they are put on ONE XCD next to each other (1-D grid, block id % 8 = XCD), so
    // a row is fetched into one L2 once instead of into all eight (measured 3.3 GB -> of HBM reads for 0.8 GB of g and y)
    int bc, bx, nbx, b;
    if (p.y_nhwc) {                                 // strips numbered over the whole batch: every XCD gets work
        const int ncg = (p.Cout + CRB_CG - 1) / CRB_CG;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int gs = xcd + 8 * (j / ncg);
        bc = j % ncg;
        nbx = p.nstrips;
        if (gs >= nbx * p.B) return;
        b = gs / nbx;
        bx = gs - b * nbx;
    } else {
        bc = blockIdx.y; bx = blockIdx.x; nbx = gridDim.x; b = blockIdx.z;
    }
    const int c0 = bc * CRB_CG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!BITS) {
        for (int i = threadIdx.x; i < CRB_CG * CIN * 9; i += 256) {
            const int c = c0 + i / (CIN * 9);
            sw[i] = (c < p.Cout) ? p.w[(int64_t)c * CIN * 9 + i % (CIN * 9)] : 0.f;
        }
        __syncthreads();
    }
    const uint32_t plane = (uint32_t)(p.H * p.W);
    const int oplane = p.Ho * p.Wo;
    const uint32_t key = drop_key_dev(p.drop);
    float acc[CRB_CG][CIN * 9];
#pragma unroll
    for (int j = 0; j < CRB_CG; ++j)
#pragma unroll
        for (int k = 0; k < CIN * 9; ++k) acc[j][k] = 0.f;

#pragma unroll 1
    for (int it = 0; it < CRB_PXT; ++it) {
        const int e = (bx * CRB_PXT + it) * 256 + threadIdx.x;
        if (e >= oplane) continue;
        const int oy = e / p.Wo, ox = e - oy * p.Wo;
        Axis ay = axis_of(oy, p.sy, p.H), ax = axis_of(ox, p.sx, p.W);
        float pt[CIN][4][9];
        uint32_t toff[4];
        if (BITS) {
            // The four 3x3 patches are windows of ONE 4x4 neighbourhood around (i0 - 1, i0 - 1) when i1 = i0 + 1: 16 loads
            // instead of 36.  At the last row / column i1 = i0: both taps of that axis are the same source pixel (same
            // patch, same recorded decision), so its weight moves to tap 0 and tap 1 (which would read the window one
            // further, i.e. something else) gets weight zero.
            if (ay.i1 == ay.i0) { ay.l0 += ay.l1; ay.l1 = 0.f; }
            if (ax.i1 == ax.i0) { ax.l0 += ax.l1; ax.l1 = 0.f; }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float* xp = p.x + ((int64_t)b * CIN + ci) * plane;
                float nb[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int yy = ay.i0 - 1 + r, yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int xx = ax.i0 - 1 + q, xc = xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx);
                        const float v = xp[yc * p.W + xc];
                        nb[r][q] = (yy == yc && xx == xc) ? v : 0.f;
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) pt[ci][t][dy * 3 + dx] = nb[(t >> 1) + dy][(t & 1) + dx];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int iy = (t & 2) ? ay.i1 : ay.i0, ix = (t & 1) ? ax.i1 : ax.i0;
                toff[t] = (uint32_t)(iy * p.W + ix);
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci)
                    load_patch_clamped(p.x + ((int64_t)b * CIN + ci) * plane, p.H, p.W, iy, ix, pt[ci][t]);
            }
        }
        const float wt[4] = {ay.l0 * ax.l0, ay.l0 * ax.l1, ay.l1 * ax.l0, ay.l1 * ax.l1};
        float gl[CRB_CG], yl[CRB_CG];       // channels-last: the pixel's eight channels are 32 contiguous bytes of g and y
        uint32_t dec = 0u;                  // BITS: the forward's decisions for these eight channels, 4 bits each
        if (p.y_nhwc) {
            const int64_t o8 = ((int64_t)b * oplane + e) * p.Cout + c0;           // Cout % 8 == 0 checked on the host
#pragma unroll
            for (int h = 0; h < CRB_CG / 4; ++h) {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.g + o8 + 4 * h);
#pragma unroll
                for (int t = 0; t < 4; ++t) gl[4 * h + t] = g4[t];
                if (!BITS && ACT == GT_ACT_RELU) {
                    const f32x4 y4 = *reinterpret_cast<const f32x4*>(p.y + o8 + 4 * h);
#pragma unroll
                    for (int t = 0; t < 4; ++t) yl[4 * h + t] = y4[t];
                }
            }
            if (BITS) {
                const unsigned long long w64 = p.bits[((int64_t)b * (p.Cout >> 4) + (c0 >> 4)) * oplane + e];
                dec = (c0 & 8) ? (uint32_t)(w64 >> 32) : (uint32_t)w64;
            }
        }
#pragma unroll          // full unroll: acc[j][..] must be statically indexed to stay in registers
        for (int j = 0; j < CRB_CG; ++j) {
            const int c = min(c0 + j, p.Cout - 1);                 // clamped: tail channels are not stored
            float go;
            float coef[4];
            if (BITS) {                     // decisions recorded by the forward (they include [y > 0]): no conv, no mask draw
                go = gl[j] * p.drop.scale;
#pragma unroll
                for (int t = 0; t < 4; ++t) coef[t] = (dec & (1u << (4 * j + t))) ? wt[t] * go : 0.f;
            } else {
                if (ACT == GT_ACT_SILU) go = p.y_nhwc ? gl[j] : p.g[((int64_t)b * p.Cout + c) * oplane + e];
                else if (p.y_nhwc) go = (yl[j] > 0.f) ? gl[j] : 0.f;
                else {
                    const int64_t o = ((int64_t)b * p.Cout + c) * oplane + e;
                    go = (p.y[o] > 0.f) ? p.g[o] : 0.f;
                }
                float cv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const float wv = sw[(j * CIN + ci) * 9 + k];
#pragma unroll
                        for (int t = 0; t < 4; ++t) cv[t] = fmaf(wv, pt[ci][t][k], cv[t]);
                    }
                const uint32_t cbase = ((uint32_t)b * (uint32_t)p.Cout + (uint32_t)c) * plane;
                if (ACT == GT_ACT_SILU) {
                    // both SiLUs re-evaluated: a_t = silu(m_t conv_t), r = bilinear(a), d out / d conv_t = silu'(r) w_t m_t silu'(m_t conv_t)
                    float av[4], dav[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float m = p.drop.thresh ? drop_mul(p.drop, key, cbase + toff[t]) : p.drop.scale;
                        silu_both(cv[t] * m, av[t], dav[t]);
                        dav[t] *= m;
                    }
                    const float rz = ay.l0 * (ax.l0 * av[0] + ax.l1 * av[1]) + ay.l1 * (ax.l0 * av[2] + ax.l1 * av[3]);
                    go *= dsilu_f(rz);
#pragma unroll
                    for (int t = 0; t < 4; ++t) coef[t] = wt[t] * dav[t] * go;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float m = p.drop.thresh ? drop_mul(p.drop, key, cbase + toff[t]) : p.drop.scale;
                        coef[t] = (cv[t] * m > 0.f) ? wt[t] * m * go : 0.f;
                    }
                }
            }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    float a = acc[j][ci * 9 + k];
#pragma unroll
                    for (int t = 0; t < 4; ++t) a = fmaf(coef[t], pt[ci][t][k], a);
                    acc[j][ci * 9 + k] = a;
                }
        }
    }
    // wave reduction, then the 4 waves through LDS (fixed order -> deterministic)
#pragma unroll
    for (int j = 0; j < CRB_CG; ++j)
#pragma unroll
        for (int k = 0; k < CIN * 9; ++k) {
            const float v = wave_sum_lane63(acc[j][k]);      // 72 sums per lane: DPP adds (shuffles: 432 LDS round trips)
            if (lane == 63) red[wave][j * CIN * 9 + k] = v;
        }
    __syncthreads();
    if (threadIdx.x < CRB_CG * CIN * 9) {
        const int c = c0 + threadIdx.x / (CIN * 9);
        if (c < p.Cout) {
            float* part = p.partial + ((int64_t)(b * nbx + bx) * p.Cout) * CIN * 9;
            part[(int64_t)c0 * CIN * 9 + threadIdx.x] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
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
// ---- channels-last on both sides (the fine-grid resize of the regressor input): one thread per (pixel, 4
// channels), flat over a row, so narrow channel counts (C = 32) keep every lane busy; RPT output rows per thread
// put 4*RPT independent float4 loads in flight.
constexpr int RN_RPT = 4;
__global__ __launch_bounds__(256) void resize_nhwc_fwd_kernel(const ResizeP p) {
    const int C4 = p.C >> 2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.Wo * C4) return;
    const int ox = e / C4, c = (e - ox * C4) * 4;
    const int oy0 = blockIdx.y * RN_RPT, b = blockIdx.z;
    const Axis ax = axis_of(ox, p.sx, p.Wi);
    Axis ay[RN_RPT];
    f32x4 v[RN_RPT][4];
    // Affine epilogue (gt_bilinear2d_fwd_affine: bias + up to two rank-1 terms, the regressor's fc(cat[x, grid]) commuted in
    // front of the resize): the bias and the weights of the thread's four channels do not depend on the row -- fetched ONCE,
    // here, and the rows' coefficients with the rows' taps, so that no load stands between the interpolation and the store
    // (resize_affine ran a dynamic loop of dependent scalar loads per row: 260 us against 118 us without the epilogue; 155 now).
    const bool rp2 = p.rp == 1 || p.rp == 2;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, w0 = bv, w1 = bv;
    float a0[RN_RPT], a1[RN_RPT];
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + c);
    if (rp2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            w0[t] = p.rp_b[(int64_t)(c + t) * p.rp_ldb];
            w1[t] = p.rp == 2 ? p.rp_b[(int64_t)(c + t) * p.rp_ldb + 1] : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < RN_RPT; ++r) {
        const int oy = min(oy0 + r, p.Ho - 1);
        ay[r] = axis_of(oy, p.sy, p.Hi);
        a0[r] = a1[r] = 0.f;
        if (rp2) {
            const float* ga = p.rp_a + (((int64_t)b * p.Ho + oy) * p.Wo + ox) * p.rp_lda;
            a0[r] = ga[0];
            if (p.rp == 2) a1[r] = ga[1];
        }
        if (p.seg) {          // padded three-segment input: gather the four real channels (block-uniform branch)
            const int CP3 = 3 * p.segp;
            v[r][0] = seg_load4(p, p.x + addr<true>(b, 0, ay[r].i0, ax.i0, CP3, p.Hi, p.Wi), c);
            v[r][1] = seg_load4(p, p.x + addr<true>(b, 0, ay[r].i0, ax.i1, CP3, p.Hi, p.Wi), c);
            v[r][2] = seg_load4(p, p.x + addr<true>(b, 0, ay[r].i1, ax.i0, CP3, p.Hi, p.Wi), c);
            v[r][3] = seg_load4(p, p.x + addr<true>(b, 0, ay[r].i1, ax.i1, CP3, p.Hi, p.Wi), c);
            continue;
        }
        v[r][0] = *reinterpret_cast<const f32x4*>(p.x + addr<true>(b, c, ay[r].i0, ax.i0, p.C, p.Hi, p.Wi));
        v[r][1] = *reinterpret_cast<const f32x4*>(p.x + addr<true>(b, c, ay[r].i0, ax.i1, p.C, p.Hi, p.Wi));
        v[r][2] = *reinterpret_cast<const f32x4*>(p.x + addr<true>(b, c, ay[r].i1, ax.i0, p.C, p.Hi, p.Wi));
        v[r][3] = *reinterpret_cast<const f32x4*>(p.x + addr<true>(b, c, ay[r].i1, ax.i1, p.C, p.Hi, p.Wi));
    }
#pragma unroll
    for (int r = 0; r < RN_RPT; ++r) {
        const int oy = oy0 + r;
        if (oy < p.Ho) {
            f32x4 o = ay[r].l0 * (ax.l0 * v[r][0] + ax.l1 * v[r][1]) + ay[r].l1 * (ax.l0 * v[r][2] + ax.l1 * v[r][3]);
            if (rp2 || !p.rp) {                 // same operations in the same order as resize_affine
                if (p.bias) o += bv;
                if (rp2) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        o[t] = fmaf(a0[r], w0[t], o[t]);
                        if (p.rp == 2) o[t] = fmaf(a1[r], w1[t], o[t]);
                    }
                }
            } else {
                o = resize_affine(p, o, b, c, oy, ox);
            }
            if (p.act == GT_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
            } else if (p.act == GT_ACT_SILU) {          // + the derivative the backward multiplies g with
                f32x4 d;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a, da;
                    silu_both(o[j], a, da);
                    o[j] = a; d[j] = da;
                }
                if (p.dact) *reinterpret_cast<f32x4*>(p.dact + addr<true>(b, c, oy, ox, p.C, p.Ho, p.Wo)) = d;
            }
            *reinterpret_cast<f32x4*>(p.y + addr<true>(b, c, oy, ox, p.C, p.Ho, p.Wo)) = o;
        }
    }
}

// gather form of the backward (no atomics), same thread mapping over an input row; falls back to the tiled
// kernel's generic loop when an axis has more than RS_MAXT contributing outputs
__global__ __launch_bounds__(256) void resize_nhwc_bwd_kernel(const ResizeP p) {
    // dx has p.C channels, or (p.seg != 0) the three padded column segments of 3 * p.segp channels: a thread owns four
    // consecutive dx columns; with segments each maps to a real channel of g or to a padding column (gradient zero)
    const int CX = p.seg ? 3 * p.segp : p.C;
    const int C4 = CX >> 2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int iy = blockIdx.y, b = blockIdx.z;
    // The tap tables (first output, count, weights: float divisions and a dozen axis_of evaluations each) are the same for every
    // channel group of a pixel and -- along y -- for the whole block: worked out ONCE per block by the first threads and read
    // back from LDS, instead of twice by every thread (they were a large part of the kernel's instructions).
    constexpr int TXMAX = 66;                        // pixels a block's 256 four-channel groups can touch when C4 >= 4
    __shared__ Taps s_tx[TXMAX];
    __shared__ Taps s_ty;
    const int px0 = (blockIdx.x * 256) / C4;
    const bool shared_taps = C4 >= 4;
    if (shared_taps) {
        const int npx = min((blockIdx.x * 256 + 255) / C4, p.Wi - 1) - px0 + 1;
        if ((int)threadIdx.x < npx) s_tx[threadIdx.x] = taps_of(px0 + threadIdx.x, p.sx, p.Wi, p.Wo);
        if (threadIdx.x == 255) s_ty = taps_of(iy, p.sy, p.Hi, p.Ho);
        __syncthreads();
    }
    if (e >= p.Wi * C4) return;
    const int ix = e / C4, c = (e - ix * C4) * 4;
    int cr[4] = {c, c + 1, c + 2, c + 3};          // real channel of each column, -1: padding
    if (p.seg) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sgi = (c + j) / p.segp, r = (c + j) - sgi * p.segp;
            const int width = sgi < 2 ? p.seg : p.C - 2 * p.seg;
            cr[j] = r < width ? sgi * p.seg + r : -1;
        }
    }
    const Taps ty = shared_taps ? s_ty : taps_of(iy, p.sy, p.Hi, p.Ho);
    const Taps tx = shared_taps ? s_tx[ix - px0] : taps_of(ix, p.sx, p.Wi, p.Wo);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jy = 0; jy < RS_MAXT; ++jy) {
        if (jy < ty.n) {
            f32x4 racc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jx = 0; jx < RS_MAXT; ++jx) {
                if (jx < tx.n) {
                    f32x4 g;
                    if (p.seg) {          // the column pairs (0,1) and (2,3) are real together or padding together
                        const int64_t o = addr<true>(b, 0, ty.lo + jy, tx.lo + jx, p.C, p.Ho, p.Wo);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x2 gv = {0.f, 0.f};
                            if (cr[2 * h] >= 0) {
                                gv = *reinterpret_cast<const f32x2*>(p.x + o + cr[2 * h]);
                                if (p.gate) {
                                    const f32x2 y = *reinterpret_cast<const f32x2*>(p.gate + o + cr[2 * h]);
                                    if (p.act == GT_ACT_SILU) { gv[0] *= y[0]; gv[1] *= y[1]; }      // y = silu'(resized)
                                    else {
                                        if (!(y[0] > 0.f)) gv[0] = 0.f;
                                        if (!(y[1] > 0.f)) gv[1] = 0.f;
                                    }
                                }
                            }
                            g[2 * h] = gv[0]; g[2 * h + 1] = gv[1];
                        }
                    } else {
                        const int64_t o = addr<true>(b, c, ty.lo + jy, tx.lo + jx, p.C, p.Ho, p.Wo);
                        g = *reinterpret_cast<const f32x4*>(p.x + o);
                        if (p.gate) {
                            const f32x4 y = *reinterpret_cast<const f32x4*>(p.gate + o);
                            if (p.act == GT_ACT_SILU) g *= y;
                            else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) if (!(y[j] > 0.f)) g[j] = 0.f;
                            }
                        }
                    }
                    racc += tx.w[jx] * g;
                }
            }
            acc += ty.w[jy] * racc;
        }
    }
    const int64_t od = addr<true>(b, c, iy, ix, CX, p.Hi, p.Wi);
    if (p.in_gate) {
        const f32x4 xin = *reinterpret_cast<const f32x4*>(p.in_gate + od);
        if (p.gate_mul) acc *= xin;
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (!(xin[j] > 0.f)) acc[j] = 0.f;
        }
    }
    *reinterpret_cast<f32x4*>(p.y + od) = acc;
}

// true when no input index of the axis has more than RS_MAXT contributing outputs (host-side bound:
// an input cell's support spans at most 2 * (no-1)/(ni-1) outputs)
static inline bool taps_fit(int ni, int no) {
    return ni <= 1 ? no <= RS_MAXT : 2.0 * (double)(no - 1) / (double)(ni - 1) + 2.0 <= (double)RS_MAXT;
}

static inline float scale_of(int ni, int no) { return (no > 1) ? (float)(ni - 1) / (float)(no - 1) : 0.f; }

}  // namespace gt

using namespace gt;

extern "C" int gt_bilinear2d_fwd_affine(const float* x, float* y, int32_t B, int32_t C, int32_t Hi, int32_t Wi,
                                        int32_t Ho, int32_t Wo, int32_t in_nhwc, int32_t out_nhwc, int32_t act,
                                        const gt_resize_affine* aff, void* stream) {
    if (int rc = check_resize(x, y, B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc)) return rc;
    if (act != GT_ACT_NONE && act != GT_ACT_RELU) return GT_ENOTSUP;
    ResizeP p{x, y, nullptr, B, C, Hi, Wi, Ho, Wo, scale_of(Hi, Ho), scale_of(Wi, Wo), act, ceil_div(Wo, RS_TX),
              nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, nullptr};
    if (aff && (aff->bias || aff->rp)) {
        if (!(in_nhwc && out_nhwc)) return GT_ENOTSUP;
        if (aff->rp < 0 || aff->rp > 8 || (aff->rp && (!aff->rp_a || !aff->rp_b))) return GT_EINVAL;
        if (aff->bias && (reinterpret_cast<uintptr_t>(aff->bias) & 15)) return GT_EALIGN;
        p.bias = aff->bias; p.rp = aff->rp; p.rp_a = aff->rp_a; p.rp_lda = aff->rp_lda;
        p.rp_b = aff->rp_b; p.rp_ldb = aff->rp_ldb;
    }
    dim3 grid((unsigned)(p.xtiles * ceil_div(C, RS_TC)), (unsigned)Ho, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (!in_nhwc && !out_nhwc) hipLaunchKernelGGL((resize_fwd_kernel<false, false>), grid, dim3(256), 0, st, p);
    else if (!in_nhwc && out_nhwc) hipLaunchKernelGGL((resize_fwd_kernel<false, true>), grid, dim3(256), 0, st, p);
    else if (in_nhwc && !out_nhwc) hipLaunchKernelGGL((resize_fwd_kernel<true, false>), grid, dim3(256), 0, st, p);
    else if ((C & 3) == 0 && ceil_div(Ho, RN_RPT) <= 65535) {
        dim3 ng((unsigned)ceil_div((int64_t)Wo * (C / 4), 256), (unsigned)ceil_div(Ho, RN_RPT), (unsigned)B);
        hipLaunchKernelGGL(resize_nhwc_fwd_kernel, ng, dim3(256), 0, st, p);
    } else hipLaunchKernelGGL((resize_fwd_kernel<true, true>), grid, dim3(256), 0, st, p);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_bilinear2d_fwd(const float* x, float* y, int32_t B, int32_t C, int32_t Hi, int32_t Wi,
                                 int32_t Ho, int32_t Wo, int32_t in_nhwc, int32_t out_nhwc, int32_t act,
                                 void* stream) {
    return gt_bilinear2d_fwd_affine(x, y, B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc, act, nullptr, stream);
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
              scale_of(Wi, Wo), act, ceil_div(Wi, RS_TX), nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, nullptr};
    dim3 grid((unsigned)(p.xtiles * ceil_div(C, RS_TC)), (unsigned)Hi, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (!out_nhwc && !in_nhwc) {
        if (ceil_div(C, 8) > 65535) return GT_EINVAL;
        dim3 pg((unsigned)ceil_div((int64_t)Hi * Wi, 256), (unsigned)ceil_div(C, 8), (unsigned)B);
        hipLaunchKernelGGL(resize_bwd_planar_kernel, pg, dim3(256), 0, st, p);
    }
    else if (!out_nhwc && in_nhwc) hipLaunchKernelGGL((resize_bwd_kernel<false, true>), grid, dim3(256), 0, st, p);
    else if (out_nhwc && !in_nhwc) hipLaunchKernelGGL((resize_bwd_kernel<true, false>), grid, dim3(256), 0, st, p);
    else if ((C & 3) == 0 && taps_fit(Hi, Ho) && taps_fit(Wi, Wo)) {
        dim3 ng((unsigned)ceil_div((int64_t)Wi * (C / 4), 256), (unsigned)Hi, (unsigned)B);
        hipLaunchKernelGGL(resize_nhwc_bwd_kernel, ng, dim3(256), 0, st, p);
    } else hipLaunchKernelGGL((resize_bwd_kernel<true, true>), grid, dim3(256), 0, st, p);
    GT_LAUNCH_CHECK();
    return 0;
}

// Channels-last resize whose INPUT is the padded three-segment buffer of ops.scaler_conv_chain (gt_hip.h)
static int check_seg(int C, int seg, int segp) {
    if (seg <= 0 || (seg & 1) || segp < seg || (segp & 3) || (C & 3) || C <= 2 * seg || C - 2 * seg > segp) return GT_EINVAL;
    return 0;
}

extern "C" int gt_bilinear2d_seg_fwd(const float* x, float* y, int32_t B, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                                     int32_t Wo, int32_t act, int32_t seg, int32_t segp, float* dact, void* stream) {
    if (int rc = check_resize(x, y, B, C, Hi, Wi, Ho, Wo, 1, 1)) return rc;
    if (int rc = check_seg(C, seg, segp)) return rc;
    if (act != GT_ACT_NONE && act != GT_ACT_RELU && act != GT_ACT_SILU) return GT_ENOTSUP;
    if (dact && (act != GT_ACT_SILU || (reinterpret_cast<uintptr_t>(dact) & 15))) return GT_EINVAL;
    if (ceil_div(Ho, RN_RPT) > 65535) return GT_EINVAL;
    ResizeP p{x, y, nullptr, B, C, Hi, Wi, Ho, Wo, scale_of(Hi, Ho), scale_of(Wi, Wo), act, ceil_div(Wo, RS_TX),
              nullptr, 0, nullptr, 0, nullptr, 0, seg, segp, nullptr, dact, 0};
    dim3 ng((unsigned)ceil_div((int64_t)Wo * (C / 4), 256), (unsigned)ceil_div(Ho, RN_RPT), (unsigned)B);
    hipLaunchKernelGGL(resize_nhwc_fwd_kernel, ng, dim3(256), 0, (hipStream_t)stream, p);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_bilinear2d_seg_bwd(const float* g, const float* y_saved, float* dx, int32_t B, int32_t C, int32_t Hi,
                                     int32_t Wi, int32_t Ho, int32_t Wo, int32_t act, int32_t seg, int32_t segp,
                                     const float* x_gate, int32_t gate_mul, void* stream) {
    if (int rc = check_resize(dx, g, B, C, Hi, Wi, Ho, Wo, 1, 1)) return rc;
    if (int rc = check_seg(C, seg, segp)) return rc;
    if (act != GT_ACT_NONE && act != GT_ACT_RELU && act != GT_ACT_SILU) return GT_ENOTSUP;
    if (act != GT_ACT_NONE && !y_saved) return GT_EINVAL;          // ReLU: the activated output; SiLU: the forward's dact
    if (y_saved && (reinterpret_cast<uintptr_t>(y_saved) & 15)) return GT_EALIGN;
    if (!taps_fit(Hi, Ho) || !taps_fit(Wi, Wo)) return GT_ENOTSUP;
    ResizeP p{g, dx, act != GT_ACT_NONE ? y_saved : nullptr, B, C, Hi, Wi, Ho, Wo, scale_of(Hi, Ho),
              scale_of(Wi, Wo), act, ceil_div(Wi, RS_TX), nullptr, 0, nullptr, 0, nullptr, 0, seg, segp, x_gate, nullptr,
              gate_mul != 0};
    if (x_gate && (reinterpret_cast<uintptr_t>(x_gate) & 15)) return GT_EALIGN;
    dim3 ng((unsigned)ceil_div((int64_t)Wi * (3 * segp / 4), 256), (unsigned)Hi, (unsigned)B);
    hipLaunchKernelGGL(resize_nhwc_bwd_kernel, ng, dim3(256), 0, (hipStream_t)stream, p);
    GT_LAUNCH_CHECK();
    return 0;
}

static int check_conv_resize(const void* x, const void* w, const void* y, int B, int Cin, int Cout, int H, int W,
                             int Ho, int Wo, const gt_dropout* drop, int act) {
    if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return GT_EINVAL;
    if (Cin > CR_MAXCI || (act != GT_ACT_RELU && act != GT_ACT_SILU)) return GT_ENOTSUP;
    if (B > 65535) return GT_EINVAL;
    if (drop && drop->p > 0.f && !drop->seed) return GT_EINVAL;
    if (drop && (drop->p < 0.f || drop->p >= 1.f)) return GT_EINVAL;
    return 0;
}

static int conv_resize_fwd(const float* x, const float* w, float* y, int32_t B, int32_t Cin,
                           int32_t Cout, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                           const gt_dropout* drop, int32_t act, int y_nhwc, void* bits, void* stream) {
    if (int rc = check_conv_resize(x, w, y, B, Cin, Cout, H, W, Ho, Wo, drop, act)) return rc;
    if (y_nhwc && ((Cout & 7) || (reinterpret_cast<uintptr_t>(y) & 15))) return GT_ENOTSUP;
    if (bits && (!y_nhwc || (Cout & 15) || (reinterpret_cast<uintptr_t>(bits) & 7) || act != GT_ACT_RELU)) return GT_ENOTSUP;
    ConvResizeP p{x, w, y, nullptr, nullptr, B, Cin, Cout, H, W, Ho, Wo, scale_of(H, Ho), scale_of(W, Wo),
                  make_drop(drop), y_nhwc, 0, reinterpret_cast<unsigned long long*>(bits)};
    dim3 grid((unsigned)ceil_div((int64_t)Ho * Wo, 256), (unsigned)ceil_div(Cout, CR_CH), (unsigned)B);
    if (y_nhwc) std::swap(grid.x, grid.y);
    hipStream_t st = (hipStream_t)stream;
    if (act == GT_ACT_SILU) switch (Cin) {
        case 1: hipLaunchKernelGGL((conv_resize_fwd_kernel<1, GT_ACT_SILU>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((conv_resize_fwd_kernel<2, GT_ACT_SILU>), grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((conv_resize_fwd_kernel<3, GT_ACT_SILU>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((conv_resize_fwd_kernel<4, GT_ACT_SILU>), grid, dim3(256), 0, st, p); break;
    } else switch (Cin) {
        case 1: hipLaunchKernelGGL(conv_resize_fwd_kernel<1>, grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(conv_resize_fwd_kernel<2>, grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(conv_resize_fwd_kernel<3>, grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(conv_resize_fwd_kernel<4>, grid, dim3(256), 0, st, p); break;
    }
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_debug_conv0_mask(void* mask, void* stream) {
    unsigned char* m = reinterpret_cast<unsigned char*>(mask);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GT_EINVAL;     // launches in flight keep their setting
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv0_mask), &m, sizeof(m), 0, hipMemcpyHostToDevice);
}

extern "C" int gt_conv3x3_resize_fwd(const float* x, const float* w, float* y, int32_t B, int32_t Cin,
                                     int32_t Cout, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                     const gt_dropout* drop, int32_t act, void* stream) {
    return conv_resize_fwd(x, w, y, B, Cin, Cout, H, W, Ho, Wo, drop, act, 0, nullptr, stream);
}
extern "C" int gt_conv3x3_resize_fwd_nhwc(const float* x, const float* w, float* y, int32_t B, int32_t Cin,
                                          int32_t Cout, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                          const gt_dropout* drop, int32_t act, void* relu_bits, void* stream) {
    return conv_resize_fwd(x, w, y, B, Cin, Cout, H, W, Ho, Wo, drop, act, 1, relu_bits, stream);
}
extern "C" int64_t gt_conv3x3_resize_bits_bytes(int32_t B, int32_t Cout, int32_t Ho, int32_t Wo) {
    if (B <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 || (Cout & 15)) return 0;
    return (int64_t)B * Ho * Wo * (Cout / 16) * 8;
}

extern "C" int64_t gt_conv3x3_resize_bwd_ws_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W) {
    (void)H; (void)W;      // partial slabs are per (image, strip of OUTPUT pixels): bounded by the input size
    return (int64_t)B * ceil_div((int64_t)H * W, 256 * CRB_PXT) * Cout * Cin * 9 * (int64_t)sizeof(float);
}

static int conv_resize_bwd(const float* g, const float* y, const float* x, const float* w, int32_t B,
                           int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                           const gt_dropout* drop, int32_t act, float* dw, void* ws, int64_t ws_bytes, int y_nhwc,
                           const void* bits, void* stream) {
    if (bits && (!y_nhwc || (Cout & 15) || CRB_CG != 8 || (reinterpret_cast<uintptr_t>(bits) & 7) || act != GT_ACT_RELU))
        return GT_ENOTSUP;
    const bool no_y = bits || act == GT_ACT_SILU;    // the SiLU backward re-evaluates both activations: y is not read
    if (int rc = check_conv_resize(x, w, no_y ? (const void*)g : (const void*)y, B, Cin, Cout, H, W, Ho, Wo, drop, act)) return rc;
    if (!g || !dw) return GT_EINVAL;
    if (no_y && !y) y = g;                           // not read (alignment checks below see a valid pointer)
    // channels-last: a block walks whole channel groups of CRB_CG (a build-time constant) as aligned float4s
    if (y_nhwc && ((Cout & 7) || (Cout % CRB_CG) || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(g)) & 15)))
        return GT_ENOTSUP;
    if (!ws || ws_bytes < gt_conv3x3_resize_bwd_ws_bytes(B, Cin, Cout, H, W)) return GT_EWS;
    ConvResizeP p{x, w, const_cast<float*>(y), g, reinterpret_cast<float*>(ws), B, Cin, Cout, H, W, Ho, Wo,
                  scale_of(H, Ho), scale_of(W, Wo), make_drop(drop), y_nhwc, 0,
                  reinterpret_cast<unsigned long long*>(const_cast<void*>(bits))};
    if (ceil_div((int64_t)Ho * Wo, 256 * CRB_PXT) > ceil_div((int64_t)H * W, 256 * CRB_PXT)) return GT_ENOTSUP;
    const int nx = ceil_div((int64_t)Ho * Wo, 256 * CRB_PXT);
    dim3 grid((unsigned)nx, (unsigned)ceil_div(Cout, CRB_CG), (unsigned)B);
    if (y_nhwc) {
        p.nstrips = nx;
        grid = dim3((unsigned)(ceil_div(Cout, CRB_CG) * (((int64_t)nx * B + 7) / 8 * 8)), 1u, 1u);
    }
    hipStream_t st = (hipStream_t)stream;
    if (bits) {
        switch (Cin) {
            case 1: hipLaunchKernelGGL((conv_resize_bwd_kernel<1, true>), grid, dim3(256), 0, st, p); break;
            case 2: hipLaunchKernelGGL((conv_resize_bwd_kernel<2, true>), grid, dim3(256), 0, st, p); break;
            case 3: hipLaunchKernelGGL((conv_resize_bwd_kernel<3, true>), grid, dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL((conv_resize_bwd_kernel<4, true>), grid, dim3(256), 0, st, p); break;
        }
    } else if (act == GT_ACT_SILU) switch (Cin) {
        case 1: hipLaunchKernelGGL((conv_resize_bwd_kernel<1, false, GT_ACT_SILU>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((conv_resize_bwd_kernel<2, false, GT_ACT_SILU>), grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((conv_resize_bwd_kernel<3, false, GT_ACT_SILU>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((conv_resize_bwd_kernel<4, false, GT_ACT_SILU>), grid, dim3(256), 0, st, p); break;
    } else switch (Cin) {
        case 1: hipLaunchKernelGGL(conv_resize_bwd_kernel<1>, grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(conv_resize_bwd_kernel<2>, grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(conv_resize_bwd_kernel<3>, grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(conv_resize_bwd_kernel<4>, grid, dim3(256), 0, st, p); break;
    }
    GT_LAUNCH_CHECK();
    const int64_t n = (int64_t)Cout * Cin * 9;
    return gt_slab_reduce(p.partial, n, B * nx, n, 1.f, dw, stream);
}

extern "C" int gt_conv3x3_resize_bwd(const float* g, const float* y, const float* x, const float* w, int32_t B,
                                     int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                     const gt_dropout* drop, int32_t act, float* dw, void* ws, int64_t ws_bytes,
                                     void* stream) {
    return conv_resize_bwd(g, y, x, w, B, Cin, Cout, H, W, Ho, Wo, drop, act, dw, ws, ws_bytes, 0, nullptr, stream);
}
extern "C" int gt_conv3x3_resize_bwd_nhwc(const float* g, const float* y, const float* x, const float* w, int32_t B,
                                          int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                          const gt_dropout* drop, int32_t act, const void* relu_bits, float* dw,
                                          void* ws, int64_t ws_bytes, void* stream) {
    return conv_resize_bwd(g, y, x, w, B, Cin, Cout, H, W, Ho, Wo, drop, act, dw, ws, ws_bytes, 1, relu_bits, stream);
}
