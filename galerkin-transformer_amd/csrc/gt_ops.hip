// Memory-bound and small-matrix kernels of the encoder / spectral hot path (gfx950).
//   - stateless dropout helpers, column sums, slab reductions (deterministic two-pass)
//   - per-head LayerNorm + position concat (fwd/bwd)
//   - Galerkin small-matrix stage: M = mask .* (K'^T V')/n, P = M Wfc_h^T (fwd/bwd)
//   - row LayerNorm (fwd/bwd), activation backward
//   - spectral complex mode mixing (fwd/bwd)
// All launches go to the caller's stream; no allocation, no synchronisation.
#include "gt_common.h"
#include <algorithm>
#include <cstring>

namespace gt {

// ------------------------------------------------------------------------------------------ misc
__global__ void seed_advance_kernel(uint64_t* s, uint64_t inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *s += inc;
}

__global__ void dropout_apply_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                     DropDev d) {
    const uint32_t key = drop_key_dev(d);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] * (d.thresh ? drop_mul(d, key, (uint32_t)i) : d.scale);
}

// y = act2(drop2(act1(drop1(x)))) in one pass (BWD: gx = gy * dy/dx, recomputed from x): the elementwise tail
// of Conv2dResBlock (layers.py:88-150) and of Interp2dUpsample's conv branch (layers.py:658-668), where the
// reference runs up to four separate elementwise kernels over a [B, C, H, W] map.  Mask index = element index.
__device__ __forceinline__ void act_pair(int act, float v, float& a, float& da) {
    if (act == GT_ACT_SILU) silu_both(v, a, da);
    else if (act == GT_ACT_RELU) { a = fmaxf(v, 0.f); da = v > 0.f ? 1.f : 0.f; }
    else if (act == GT_ACT_GELU) gelu_both(v, a, da);
    else { a = v; da = 1.f; }
}
template <bool BWD>
__device__ __forceinline__ float dropact_one(float x, float gy, uint32_t idx, const DropDev& d1, uint32_t k1, int a1,
                                             const DropDev& d2, uint32_t k2, int a2) {
    const float m1 = d1.thresh ? drop_mul(d1, k1, idx) : d1.scale;
    const float m2 = d2.thresh ? drop_mul(d2, k2, idx) : d2.scale;
    float a, da, b, db;
    act_pair(a1, x * m1, a, da);
    act_pair(a2, a * m2, b, db);
    return BWD ? gy * db * m2 * da * m1 : b;
}
template <bool BWD>
__global__ __launch_bounds__(256) void dropact_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                      float* __restrict__ out, int64_t n, DropDev d1, int a1,
                                                      DropDev d2, int a2, int vec) {
    const uint32_t k1 = drop_key_dev(d1), k2 = drop_key_dev(d2);
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nth) {
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
            f32x4 gv = {0.f, 0.f, 0.f, 0.f}, o;
            if (BWD) gv = reinterpret_cast<const f32x4*>(gy)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = dropact_one<BWD>(xv[j], gv[j], (uint32_t)(4 * i + j), d1, k1, a1, d2, k2, a2);
            reinterpret_cast<f32x4*>(out)[i] = o;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nth)
            out[i] = dropact_one<BWD>(x[i], BWD ? gy[i] : 0.f, (uint32_t)i, d1, k1, a1, d2, k2, a2);
    } else {
        for (int64_t i = tid; i < n; i += nth)
            out[i] = dropact_one<BWD>(x[i], BWD ? gy[i] : 0.f, (uint32_t)i, d1, k1, a1, d2, k2, a2);
    }
}

// out[i] = alpha * sum_k slabs[k*stride + i].  A block owns 16 consecutive outputs; its 16 slab
// lanes each walk every 16th slab (fixed order -> deterministic), then a fixed LDS tree combines.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int64_t stride,
                                                          int n_slabs, int64_t n, float alpha,
                                                          float* __restrict__ out) {
    __shared__ float red[16][17];
    const int ox = threadIdx.x & 15, sy = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + ox;
    float s = 0.f;
    if (i < n) {
        // 4 independent partial sums per lane keep 4 loads in flight; combined in a fixed order
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = sy;
        for (; k + 48 < n_slabs; k += 64) {
            s0 += slabs[(int64_t)k * stride + i];
            s1 += slabs[(int64_t)(k + 16) * stride + i];
            s2 += slabs[(int64_t)(k + 32) * stride + i];
            s3 += slabs[(int64_t)(k + 48) * stride + i];
        }
        for (; k < n_slabs; k += 16) s0 += slabs[(int64_t)k * stride + i];
        s = (s0 + s1) + (s2 + s3);
    }
    red[sy][ox] = s;
    __syncthreads();
    if (sy == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][ox];
        out[i] = alpha * t;
    }
}

// partial[g][n] = sum over rows r = g*16+ry, stepping gridDim.y*16, of A[r][n]*keep(r,n).
// Block = 16 float4 column groups (64 columns) x 16 row lanes; fixed-order LDS combine (deterministic).
constexpr int CS_MAXG = 512;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int64_t lda, int M,
                                                     int N, DropDev d, int vec, float* __restrict__ partial) {
    __shared__ float red[16][65];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int n = blockIdx.x * 64 + 4 * cx;
    const uint32_t key = drop_key_dev(d);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        const bool full = vec && (n + 3 < N);
        for (int r = blockIdx.y * 16 + ry; r < M; r += gridDim.y * 16) {
            const float* p = A + (int64_t)r * lda + n;
            float v0, v1 = 0.f, v2 = 0.f, v3 = 0.f;
            if (full) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p);
                v0 = t[0]; v1 = t[1]; v2 = t[2]; v3 = t[3];
            } else {
                v0 = p[0];
                if (n + 1 < N) v1 = p[1];
                if (n + 2 < N) v2 = p[2];
                if (n + 3 < N) v3 = p[3];
            }
            if (d.thresh) {
                const uint32_t di = (uint32_t)((int64_t)r * N + n);
                v0 *= drop_mul(d, key, di); v1 *= drop_mul(d, key, di + 1);
                v2 *= drop_mul(d, key, di + 2); v3 *= drop_mul(d, key, di + 3);
            }
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
    }
    red[ry][4 * cx + 0] = s0; red[ry][4 * cx + 1] = s1; red[ry][4 * cx + 2] = s2; red[ry][4 * cx + 3] = s3;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int nn = blockIdx.x * 64 + threadIdx.x;
        if (nn < N) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
            partial[(int64_t)blockIdx.y * N + nn] = t * (d.thresh ? 1.f : d.scale);
        }
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ pre,
                               float* __restrict__ dpre, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float x = pre[i], g = dout[i];
        dpre[i] = (act == GT_ACT_SILU) ? g * dsilu_f(x) : (act == GT_ACT_RELU ? (x > 0.f ? g : 0.f) : g);
    }
}

// ------------------------------------------------------------------------------------------ head norm
// One block handles HN_TOK tokens.  LDS image: seg[tok][3h][dk+1] (pad 1 -> a thread walking its own
// segment is conflict-free against its neighbours).
constexpr int HN_TOK_MAX = 16;
// tokens per block such that the LDS image stays <= ~48 KiB
static inline int hn_tok(int per_token_floats) {
    int t = 12000 / std::max(per_token_floats, 1);
    return std::max(1, std::min(t, HN_TOK_MAX));
}

__global__ __launch_bounds__(256) void headnorm_fwd_kernel(
    const float* __restrict__ qkv, const float* __restrict__ pos, const float* __restrict__ gamma,
    const float* __restrict__ beta, int T, int h, int dk, int p, int DP, int norm_mask, float eps,
    float* __restrict__ out, float* __restrict__ stats, int HN_TOK) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int d3 = 3 * h * dk, S = 3 * h, pitch = dk + 1;
    const int t0 = blockIdx.x * HN_TOK, nt = min(HN_TOK, T - t0);
    for (int e = threadIdx.x; e < nt * d3; e += blockDim.x) {
        const int tok = e / d3, f = e % d3;
        lds[(tok * S + f / dk) * pitch + (f % dk)] = qkv[(int64_t)(t0 + tok) * d3 + f];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < nt * S; it += blockDim.x) {
        const int tok = it / S, s = it % S, stream = s / h, head = s % h;
        if (!((norm_mask >> stream) & 1)) continue;
        int ni = 0;
        for (int q = 0; q < stream; ++q) ni += (norm_mask >> q) & 1;
        float* v = lds + it * pitch;
        float mu = 0.f;
        for (int j = 0; j < dk; ++j) mu += v[j];
        mu /= dk;
        float var = 0.f;
        for (int j = 0; j < dk; ++j) { const float c = v[j] - mu; var += c * c; }
        var /= dk;
        const float rstd = 1.f / sqrtf(var + eps);
        const float* g = gamma + (ni * h + head) * dk;
        const float* b = beta + (ni * h + head) * dk;
        for (int j = 0; j < dk; ++j) v[j] = (v[j] - mu) * rstd * g[j] + b[j];
        float* st = stats + (((int64_t)ni * T + t0 + tok) * h + head) * 2;
        st[0] = mu;
        st[1] = rstd;
    }
    __syncthreads();
    const int per_stream = nt * h * DP;
    for (int e = threadIdx.x; e < 3 * per_stream; e += blockDim.x) {
        const int stream = e / per_stream, r = e % per_stream;
        const int tok = r / (h * DP), head = (r / DP) % h, c = r % DP;
        float val = 0.f;
        if (c < p) val = pos[(int64_t)(t0 + tok) * p + c];
        else if (c < p + dk) val = lds[(tok * S + stream * h + head) * pitch + (c - p)];
        out[((int64_t)stream * T + t0) * h * DP + r] = val;
    }
}

__global__ __launch_bounds__(256) void headnorm_bwd_kernel(
    const float* __restrict__ d_out, const float* __restrict__ qkv, const float* __restrict__ gamma,
    const float* __restrict__ stats, int T, int h, int dk, int p, int DP, int norm_mask,
    float* __restrict__ d_qkv, float* __restrict__ partial /* [nblk][2(dg,db)][2][h][dk] */, int HN_TOK) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int d3 = 3 * h * dk, S = 3 * h, pitch = dk + 1;
    float* xs = lds;                              // raw -> xhat   [HN_TOK][S][pitch]
    float* dy = lds + HN_TOK * S * pitch;         // upstream grad [HN_TOK][S][pitch]
    float* m1 = dy + HN_TOK * S * pitch;          // [HN_TOK*S]
    float* m2 = m1 + HN_TOK * S;
    float* rs = m2 + HN_TOK * S;
    const int ngroups = (T + HN_TOK - 1) / HN_TOK;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const bool first = (grp == (int)blockIdx.x);
    const int t0 = grp * HN_TOK, nt = min(HN_TOK, T - t0);
    __syncthreads();
    for (int e = threadIdx.x; e < nt * d3; e += blockDim.x) {
        const int tok = e / d3, f = e % d3, s = f / dk, j = f % dk;
        xs[(tok * S + s) * pitch + j] = qkv[(int64_t)(t0 + tok) * d3 + f];
        const int stream = s / h, head = s % h;
        dy[(tok * S + s) * pitch + j] =
            d_out[(((int64_t)stream * T + t0 + tok) * h + head) * DP + p + j];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < nt * S; it += blockDim.x) {
        const int tok = it / S, s = it % S, stream = s / h, head = s % h;
        if (!((norm_mask >> stream) & 1)) continue;
        int ni = 0;
        for (int q = 0; q < stream; ++q) ni += (norm_mask >> q) & 1;
        const float* st = stats + (((int64_t)ni * T + t0 + tok) * h + head) * 2;
        const float mu = st[0], rstd = st[1];
        const float* g = gamma + (ni * h + head) * dk;
        float* x = xs + it * pitch;
        const float* gy = dy + it * pitch;
        float a1 = 0.f, a2 = 0.f;
        for (int j = 0; j < dk; ++j) {
            const float xh = (x[j] - mu) * rstd;
            x[j] = xh;
            const float gg = gy[j] * g[j];
            a1 += gg;
            a2 += gg * xh;
        }
        m1[it] = a1 / dk;
        m2[it] = a2 / dk;
        rs[it] = rstd;
    }
    __syncthreads();
    // dgamma/dbeta partial sums over this block's tokens: one thread per (ni, head, j)
    const int nn = ((norm_mask & 1) + ((norm_mask >> 1) & 1) + ((norm_mask >> 2) & 1));
    const int hd = h * dk;
    float* pg = partial + (int64_t)blockIdx.x * 2 * 2 * hd;
    for (int e = threadIdx.x; e < 2 * hd; e += blockDim.x) {
        const int ni = e / hd, head = (e % hd) / dk, j = e % dk;
        float sg = 0.f, sb = 0.f;
        if (ni < nn) {
            int stream = -1, cnt = -1;
            for (int q = 0; q < 3; ++q)
                if ((norm_mask >> q) & 1) { if (++cnt == ni) { stream = q; break; } }
            const int s = stream * h + head;
            for (int tok = 0; tok < nt; ++tok) {
                const float gyv = dy[(tok * S + s) * pitch + j];
                sg += gyv * xs[(tok * S + s) * pitch + j];
                sb += gyv;
            }
        }
        pg[e] = first ? sg : pg[e] + sg;
        pg[2 * hd + e] = first ? sb : pg[2 * hd + e] + sb;
    }
    for (int e = threadIdx.x; e < nt * d3; e += blockDim.x) {
        const int tok = e / d3, f = e % d3, s = f / dk, j = f % dk, stream = s / h, head = s % h;
        const int it = tok * S + s;
        float g = dy[it * pitch + j];
        if ((norm_mask >> stream) & 1) {
            int ni = 0;
            for (int q = 0; q < stream; ++q) ni += (norm_mask >> q) & 1;
            const float gm = gamma[(ni * h + head) * dk + j];
            g = rs[it] * (g * gm - m1[it] - xs[it * pitch + j] * m2[it]);
        }
        d_qkv[(int64_t)(t0 + tok) * d3 + f] = g;
    }
    }   // token groups
}

// ---- bandwidth-shaped head norm (dk % 4 == 0) ------------------------------------------------------
// Thread layout: PT = 3h*G lanes per token (G = pow2 >= dk/4 lanes per head segment, one float4 each),
// R = blockDim/PT tokens in flight per pass; a lane keeps its (segment, quarter) for the whole kernel, so
// gamma/beta stay in registers and (backward) the affine gradients accumulate in registers.  Segment
// statistics are G-lane shuffle reductions.  qkv / d_qkv move as aligned float4; the head tiles (offset by
// the p coordinate columns) move as float2 when p is even, scalars otherwise.
struct HeadGeom {
    int T, h, dk, p, DP, norm_mask, G, PT, R, tpb;
};

__device__ __forceinline__ float group_sum(float v, int G) {
    for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void tile_store4(float* __restrict__ dst, int p, f32x4 v) {
    if ((p & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
    else if ((p & 1) == 0) {
        *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v[2], v[3]};
    } else { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
}
__device__ __forceinline__ f32x4 tile_load4(const float* __restrict__ src, int p) {
    if ((p & 3) == 0) return *reinterpret_cast<const f32x4*>(src);
    if ((p & 1) == 0) {
        const f32x2 a = *reinterpret_cast<const f32x2*>(src), b = *reinterpret_cast<const f32x2*>(src + 2);
        return f32x4{a[0], a[1], b[0], b[1]};
    }
    return f32x4{src[0], src[1], src[2], src[3]};
}

__global__ void headnorm_fwd_v2_kernel(const float* __restrict__ qkv, const float* __restrict__ pos,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       HeadGeom g, float eps, float* __restrict__ out,
                                       float* __restrict__ stats) {
    const int r = threadIdx.x / g.PT, l = threadIdx.x % g.PT;
    if (r >= g.R) return;
    const int seg = l / g.G, q = l % g.G, Q4 = g.dk >> 2;
    const bool active = q < Q4;
    const int stream = seg / g.h, head = seg % g.h;
    const bool normed = (g.norm_mask >> stream) & 1;
    const int ni = __popc(g.norm_mask & ((1 << stream) - 1));
    f32x4 gm = {1.f, 1.f, 1.f, 1.f}, bt = {0.f, 0.f, 0.f, 0.f};
    if (normed && active) {
        gm = *reinterpret_cast<const f32x4*>(gamma + (ni * g.h + head) * g.dk + 4 * q);
        bt = *reinterpret_cast<const f32x4*>(beta + (ni * g.h + head) * g.dk + 4 * q);
    }
    const int d3 = 3 * g.h * g.dk;
    const float inv = 1.f / (float)g.dk;
    const int t_end = min(g.T, (int)(blockIdx.x + 1) * g.tpb);
    // two tokens per trip: both loads are requested before either is consumed
    for (int t = blockIdx.x * g.tpb + r; t < t_end; t += 2 * g.R) {
        const bool two = t + g.R < t_end;
        f32x4 xx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (active) {
            xx[0] = *reinterpret_cast<const f32x4*>(qkv + (int64_t)t * d3 + seg * g.dk + 4 * q);
            if (two) xx[1] = *reinterpret_cast<const f32x4*>(qkv + (int64_t)(t + g.R) * d3 + seg * g.dk + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const int tt = t + u * g.R;
            const f32x4 x = xx[u];
            f32x4 y = x;
            if (normed) {
                const float mu = group_sum(x[0] + x[1] + x[2] + x[3], g.G) * inv;
                f32x4 c = x - mu;
                if (!active) c = f32x4{0.f, 0.f, 0.f, 0.f};
                const float var = group_sum(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3], g.G) * inv;
                const float rstd = 1.f / sqrtf(var + eps);
                y = c * rstd * gm + bt;
                if (q == 0)
                    *reinterpret_cast<f32x2*>(stats + (((int64_t)ni * g.T + tt) * g.h + head) * 2) = f32x2{mu, rstd};
            }
            float* row = out + (((int64_t)stream * g.T + tt) * g.h + head) * g.DP;
            if (active) tile_store4(row + g.p + 4 * q, g.p, y);
            if (q == 0)
                for (int j = 0; j < g.p; ++j) row[j] = pos[(int64_t)tt * g.p + j];
            if (q == Q4 - 1)
                for (int j = g.p + g.dk; j < g.DP; ++j) row[j] = 0.f;
        }
    }
}

__global__ void headnorm_bwd_v2_kernel(const float* __restrict__ d_out, const float* __restrict__ qkv,
                                       const float* __restrict__ gamma, const float* __restrict__ stats,
                                       HeadGeom g, float* __restrict__ d_qkv,
                                       float* __restrict__ partial /* [nblk][dg: 2*h*dk | db: 2*h*dk] */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [R][PT][8]
    const int r = threadIdx.x / g.PT, l = threadIdx.x % g.PT;
    const int hd = g.h * g.dk;
    if (r < g.R) {
        const int seg = l / g.G, q = l % g.G, Q4 = g.dk >> 2;
        const bool active = q < Q4;
        const int stream = seg / g.h, head = seg % g.h;
        const bool normed = (g.norm_mask >> stream) & 1;
        const int ni = __popc(g.norm_mask & ((1 << stream) - 1));
        f32x4 gm = {1.f, 1.f, 1.f, 1.f};
        if (normed && active) gm = *reinterpret_cast<const f32x4*>(gamma + (ni * g.h + head) * g.dk + 4 * q);
        f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
        const int d3 = 3 * hd;
        const float inv = 1.f / (float)g.dk;
        const int t_end = min(g.T, (int)(blockIdx.x + 1) * g.tpb);
        for (int t = blockIdx.x * g.tpb + r; t < t_end; t += 2 * g.R) {      // two tokens per trip (see forward)
            const bool two = t + g.R < t_end;
            f32x4 gyy[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, xx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            f32x2 stt[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                const int tt = t + u * g.R;
                if (active) {
                    gyy[u] = tile_load4(d_out + (((int64_t)stream * g.T + tt) * g.h + head) * g.DP + g.p + 4 * q, g.p);
                    if (normed) xx[u] = *reinterpret_cast<const f32x4*>(qkv + (int64_t)tt * d3 + seg * g.dk + 4 * q);
                }
                if (normed) stt[u] = *reinterpret_cast<const f32x2*>(stats + (((int64_t)ni * g.T + tt) * g.h + head) * 2);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                const int tt = t + u * g.R;
                const f32x4 gy = gyy[u], x = xx[u];
                f32x4 dx = gy;
                if (normed) {
                    const float mu = stt[u][0], rstd = stt[u][1];
                    f32x4 xh = (x - mu) * rstd;
                    if (!active) xh = f32x4{0.f, 0.f, 0.f, 0.f};
                    const f32x4 gg = gy * gm;
                    const float m1 = group_sum(gg[0] + gg[1] + gg[2] + gg[3], g.G) * inv;
                    const float m2 = group_sum(gg[0] * xh[0] + gg[1] * xh[1] + gg[2] * xh[2] + gg[3] * xh[3], g.G) * inv;
                    dx = rstd * (gg - m1 - xh * m2);
                    dg += gy * xh;
                    db += gy;
                }
                if (active) *reinterpret_cast<f32x4*>(d_qkv + (int64_t)tt * d3 + seg * g.dk + 4 * q) = dx;
            }
        }
        float* me = lds + ((size_t)r * g.PT + l) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) { me[j] = dg[j]; me[4 + j] = db[j]; }
    }
    __syncthreads();
    // fixed-order combine over the R token rows, one thread per (lane slot, component)
    float* pg = partial + (int64_t)blockIdx.x * 4 * hd;
    const int nn = __popc(g.norm_mask & 7);
    for (int e = threadIdx.x; e < 4 * hd; e += blockDim.x)          // slots of absent norm streams
        if ((e % (2 * hd)) / hd >= nn) pg[e] = 0.f;
    for (int e = threadIdx.x; e < g.PT * 8; e += blockDim.x) {
        const int ll = e >> 3, comp = e & 7;
        const int seg = ll / g.G, q = ll % g.G;
        const int stream = seg / g.h, head = seg % g.h;
        if (q >= (g.dk >> 2) || !((g.norm_mask >> stream) & 1)) continue;
        const int ni = __popc(g.norm_mask & ((1 << stream) - 1));
        float s = 0.f;
        for (int rr = 0; rr < g.R; ++rr) s += lds[((size_t)rr * g.PT + ll) * 8 + comp];
        const int idx = ni * hd + head * g.dk + 4 * q + (comp & 3);
        pg[(comp < 4 ? 0 : 2 * hd) + idx] = s;
    }
}

// ------------------------------------------------------------------------------------------ galerkin dK', dV'
// dK'[t] = V'[t] dM^T and dV'[t] = K'[t] dM for every token of one (batch, head)  -- the backward of
// M = K'^T V' (layers.py:723) -- as one streaming pass: the two DP x DP operands live in registers as MFMA A
// fragments, token rows go from HBM straight into B fragments.  A row of DP = 16G + 4 floats is G*4 + 1
// float4: lane (row j, kq) loads float4 number kq + 4g (g < G) and the last one; k-step (g, c) contracts
// k = 4(kq + 4g) + c (component c of the lane's g-th float4), the final step k = 16G + kq (component kq of the
// shared last float4) -- every k exactly once, every load a full 64-byte run per row.  Result tiles come out
// transposed (output column x row), i.e. one float4 of the output row per lane.
struct DkvP {
    const float* Kp; const float* Vp; const float* dM; float* dKp; float* dVp;
    int n, h;
};
template <int G>
__global__ __launch_bounds__(256, 2) void galerkin_dkv_kernel(const DkvP p) {
    constexpr int DP = 16 * G + 4, NS = 4 * G + 1, NMT = G + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int head = blockIdx.x % p.h, b = blockIdx.x / p.h;
    const int64_t hD = (int64_t)p.h * DP;
    const int64_t base = ((int64_t)b * p.n) * hD + (int64_t)head * DP;
    const float* dm = p.dM + ((int64_t)b * p.h + head) * DP * DP;
    // A fragments: lane (i = output column 16mt + j, kq); a1 -> dK' (dM[col][k]), a2 -> dV' (dM[k][col])
    float a1[NMT][NS], a2[NMT][NS];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const int col = 16 * mt + j, cc = min(col, DP - 1);
        const float live = col < DP ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = (s < 4 * G) ? 4 * (kq + 4 * (s >> 2)) + (s & 3) : 16 * G + kq;
            a1[mt][s] = live * dm[cc * DP + k];
            a2[mt][s] = live * dm[k * DP + cc];
        }
    }
    // token tiles of this (batch, head) are shared out over gridDim.y blocks (ex4: B h = 16 would leave 240 CUs idle)
    const int ntile = (p.n + 15) >> 4, per = (ntile + gridDim.y - 1) / gridDim.y;
    const int tend = min(ntile, (int)(blockIdx.y + 1) * per);
    for (int tile = blockIdx.y * per + wave; tile < tend; tile += 4) {
        const int t = 16 * tile + j, tc = min(t, p.n - 1);
        const float* kr = p.Kp + base + (int64_t)tc * hD;
        const float* vr = p.Vp + base + (int64_t)tc * hD;
        f32x4 kk[G + 1], vv[G + 1];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            kk[g] = *reinterpret_cast<const f32x4*>(kr + 4 * (kq + 4 * g));
            vv[g] = *reinterpret_cast<const f32x4*>(vr + 4 * (kq + 4 * g));
        }
        kk[G] = *reinterpret_cast<const f32x4*>(kr + 16 * G);
        vv[G] = *reinterpret_cast<const f32x4*>(vr + 16 * G);
        f32x4 acc1[NMT], acc2[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) acc1[mt] = acc2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float bv, bk;
            if (s < 4 * G) { bv = vv[s >> 2][s & 3]; bk = kk[s >> 2][s & 3]; }
            else {
                bv = kq == 0 ? vv[G][0] : (kq == 1 ? vv[G][1] : (kq == 2 ? vv[G][2] : vv[G][3]));
                bk = kq == 0 ? kk[G][0] : (kq == 1 ? kk[G][1] : (kq == 2 ? kk[G][2] : kk[G][3]));
            }
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt][s], bv, acc1[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[mt][s], bk, acc2[mt], 0, 0, 0);
            }
        }
        if (t < p.n) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int col = 16 * mt + 4 * kq;
                if (col < DP) {
                    *reinterpret_cast<f32x4*>(p.dKp + base + (int64_t)t * hD + col) = acc1[mt];
                    *reinterpret_cast<f32x4*>(p.dVp + base + (int64_t)t * hD + col) = acc2[mt];
                }
            }
        }
    }
}

// ---------------------------------------------------------------- galerkin dK', dV' with the LayerNorm backward behind them
// The two products of galerkin_dkv_kernel, and on the same registers the per-head LayerNorm backward of
// headnorm_bwd_v2_kernel for the K and V streams (layers.py:841-874 backwards): the head-tile gradients dK', dV'
// ([T][h][DP], 2 x 136 MB at B = 128) are never written or read back.  A lane holds four consecutive tile columns
// 16 mt + 4 kq .. + 3 of token row j (columns in value order, see the kernel): a row's dk values sit in the four kq
// lanes of its j, so the two row means are a local sum and two cross-lane adds.  d(gamma), d(beta): per-lane running sums
// over the block's tokens, folded over the 16 token lanes, then over the four waves in LDS in a fixed order; block
// (b, head) owns the head's dk-slice of partial[b][dg K | dg V | db K | db V] (the layout gt_headnorm_bwd reduces).
struct DkvLnP {
    const float* Kp; const float* Vp; const float* dM;
    const float* qkv; const float* gamma; const float* stats;
    float* d_qkv; float* partial;
    int n, h, dk, p, T;
    const float* beta;                              // PLAIN only
};
// PLAIN: the head tiles hold the normalised values WITHOUT the LayerNorm affine (gt_hip.h: hn_plain): K' = gamma_K xh + beta_K
// is never formed -- gamma scales the rows of the dM fragments (the contraction index is the tile column), beta dM is a
// per-output-column constant added to the products, and xh for the LayerNorm backward is the tile itself: the raw
// projection is neither stored by the forward nor read here.
// G = 3 (DP = 52: ex3's 48-wide heads): the two sets of dM fragments alone are 104 registers -- at two blocks per CU the
// kernel spilled 220 (PLAIN) / 119 registers to scratch and ran 2.7x slower per token than G = 2 (499 vs 181 us for the same
// bytes, round 6 profile); one block per CU opens the whole 512-entry register file (no scratch).
template <int G, bool PLAIN>
__global__ __launch_bounds__(256, (G >= 3 ? 1 : 2)) void galerkin_dkv_ln_kernel(const DkvLnP p) {
    constexpr int DP = 16 * G + 4, NS = 4 * G + 1, NMT = G + 1;
    __shared__ float red[4][4][NMT][4][4];          // [wave][kq][mt][c][dgK, dbK, dgV, dbV]
    __shared__ __attribute__((aligned(16))) float cst[2][4][NMT][4];    // PLAIN: [dK' | dV'][kq][mt][c] = (beta dM) of the lane's columns
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int head = blockIdx.x % p.h, b = blockIdx.x / p.h;
    const int64_t hD = (int64_t)p.h * DP;
    const int64_t base = ((int64_t)b * p.n) * hD + (int64_t)head * DP;
    const float* dm = p.dM + ((int64_t)b * p.h + head) * DP * DP;
    const int hd = p.h * p.dk, d3 = 3 * hd;
    const float inv = 1.f / (float)p.dk;
    // output columns in VALUE order: column c' < dk is value c' (tile column p + c'), the coordinate columns follow, then
    // the pad -- a permutation of the rows of the dM fragments, so that the lane's four consecutive outputs are an
    // aligned float4 of the raw projection row and of its gradient
    float a1[NMT][NS], a2[NMT][NS];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const int cp = 16 * mt + j;
        const int col = cp < p.dk ? cp + p.p : (cp < p.dk + p.p ? cp - p.dk : cp), cc = min(col, DP - 1);
        const float live = cp < DP ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = (s < 4 * G) ? 4 * (kq + 4 * (s >> 2)) + (s & 3) : 16 * G + kq;
            float sv = 1.f, sk = 1.f;               // PLAIN: the operand rows carry xh; its gamma moves onto dM
            if (PLAIN && k >= p.p && k < p.p + p.dk) {
                sv = p.gamma[hd + (int64_t)head * p.dk + k - p.p];
                sk = p.gamma[(int64_t)head * p.dk + k - p.p];
            }
            a1[mt][s] = live * sv * dm[cc * DP + k];
            a2[mt][s] = live * sk * dm[k * DP + cc];
        }
    }
    if (PLAIN) {                                     // beta dM of every output column, once per block
        for (int e = threadIdx.x; e < 2 * 4 * NMT * 4; e += blockDim.x) {
            const int c = e & 3, mt = (e >> 2) % NMT, kq2 = (e / (4 * NMT)) & 3, which = e / (16 * NMT);
            const int cp = 16 * mt + 4 * kq2 + c;
            const int col = cp < p.dk ? cp + p.p : (cp < p.dk + p.p ? cp - p.dk : cp);
            float acc = 0.f;
            if (cp < DP)
                for (int v = 0; v < p.dk; ++v) {
                    const int k = p.p + v;
                    acc += which == 0 ? p.beta[hd + (int64_t)head * p.dk + v] * dm[col * DP + k]      // dK' = V' dM^T
                                      : p.beta[(int64_t)head * p.dk + v] * dm[k * DP + col];           // dV' = K' dM
                }
            cst[which][kq2][mt][c] = acc;
        }
        __syncthreads();
    }
    bool ok[NMT];                                   // the lane's float4 of group mt holds values (dk % 4 == 0: all or none)
    f32x4 gmK[NMT], gmV[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const int v0 = 16 * mt + 4 * kq;
        ok[mt] = v0 < p.dk;
        gmK[mt] = gmV[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok[mt]) {
            gmK[mt] = *reinterpret_cast<const f32x4*>(p.gamma + (int64_t)head * p.dk + v0);
            gmV[mt] = *reinterpret_cast<const f32x4*>(p.gamma + hd + (int64_t)head * p.dk + v0);
        }
    }
    f32x4 dgK[NMT], dbK[NMT], dgV[NMT], dbV[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) dgK[mt] = dbK[mt] = dgV[mt] = dbV[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // a (batch, head)'s token tiles are shared out over gridDim.y blocks (small batches: B h blocks alone leave the chip idle)
    const int ntile = (p.n + 15) >> 4, per = (ntile + gridDim.y - 1) / gridDim.y;
    const int tend = min(ntile, (int)(blockIdx.y + 1) * per);
    for (int tile = blockIdx.y * per + wave; tile < tend; tile += 4) {
        const int t = 16 * tile + j, tc = min(t, p.n - 1);
        const int64_t tok = (int64_t)b * p.n + tc;
        const float* kr = p.Kp + base + (int64_t)tc * hD;
        const float* vr = p.Vp + base + (int64_t)tc * hD;
        f32x4 kk[G + 1], vv[G + 1];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            kk[g] = *reinterpret_cast<const f32x4*>(kr + 4 * (kq + 4 * g));
            vv[g] = *reinterpret_cast<const f32x4*>(vr + 4 * (kq + 4 * g));
        }
        kk[G] = *reinterpret_cast<const f32x4*>(kr + 16 * G);
        vv[G] = *reinterpret_cast<const f32x4*>(vr + 16 * G);
        // raw projection rows (PLAIN: the normalised values themselves, in value order, out of the tile rows just
        // requested -- cache-hot) and statistics of this token: requested before the products, used after them
        const float* xk = PLAIN ? kr + p.p + 4 * kq : p.qkv + tok * d3 + hd + head * p.dk + 4 * kq;       // + 16 mt
        const float* xv = PLAIN ? vr + p.p + 4 * kq : xk + hd;
        f32x4 xK[NMT], xV[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            xK[mt] = xV[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok[mt]) {
                xK[mt] = PLAIN ? tile_load4(xk + 16 * mt, p.p) : *reinterpret_cast<const f32x4*>(xk + 16 * mt);
                xV[mt] = PLAIN ? tile_load4(xv + 16 * mt, p.p) : *reinterpret_cast<const f32x4*>(xv + 16 * mt);
            }
        }
        const f32x2 stK = *reinterpret_cast<const f32x2*>(p.stats + (tok * p.h + head) * 2);
        const f32x2 stV = *reinterpret_cast<const f32x2*>(p.stats + (((int64_t)p.T + tok) * p.h + head) * 2);

        f32x4 acc1[NMT], acc2[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            acc1[mt] = acc2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (PLAIN) {
                acc1[mt] = *reinterpret_cast<const f32x4*>(&cst[0][kq][mt][0]);
                acc2[mt] = *reinterpret_cast<const f32x4*>(&cst[1][kq][mt][0]);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float bv, bk;
            if (s < 4 * G) { bv = vv[s >> 2][s & 3]; bk = kk[s >> 2][s & 3]; }
            else {
                bv = kq == 0 ? vv[G][0] : (kq == 1 ? vv[G][1] : (kq == 2 ? vv[G][2] : vv[G][3]));
                bk = kq == 0 ? kk[G][0] : (kq == 1 ? kk[G][1] : (kq == 2 ? kk[G][2] : kk[G][3]));
            }
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt][s], bv, acc1[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[mt][s], bk, acc2[mt], 0, 0, 0);
            }
        }
        const bool live = t < p.n;
        // LayerNorm backward of one stream on the lane's columns: gy = d(normalised head row), x = raw row (PLAIN: xh)
        auto ln_bwd = [&](const f32x4 (&gy)[NMT], const f32x4 (&x)[NMT], const f32x4 (&gm)[NMT], f32x2 st,
                          f32x4 (&dg)[NMT], f32x4 (&db)[NMT], float* __restrict__ dst) {
            const float mu = st[0], rstd = st[1];
            f32x4 xh[NMT], gg[NMT];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    xh[mt][c] = ok[mt] ? (PLAIN ? x[mt][c] : (x[mt][c] - mu) * rstd) : 0.f;
                    gg[mt][c] = ok[mt] ? gy[mt][c] * gm[mt][c] : 0.f;
                    s1 += gg[mt][c];
                    s2 += gg[mt][c] * xh[mt][c];
                }
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            const float m1 = s1 * inv, m2 = s2 * inv;
            if (!live) return;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                if (!ok[mt]) continue;
                f32x4 dx;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dx[c] = rstd * (gg[mt][c] - m1 - xh[mt][c] * m2);
                    dg[mt][c] += gy[mt][c] * xh[mt][c];
                    db[mt][c] += gy[mt][c];
                }
                *reinterpret_cast<f32x4*>(dst + 16 * mt) = dx;
            }
        };
        float* dk_row = p.d_qkv + tok * d3 + hd + head * p.dk + 4 * kq;
        ln_bwd(acc1, xK, gmK, stK, dgK, dbK, dk_row);
        ln_bwd(acc2, xV, gmV, stV, dgV, dbV, dk_row + hd);
    }
    // fold the 16 token lanes, then the four waves (fixed order)
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[4] = {dgK[mt][c], dbK[mt][c], dgV[mt][c], dbV[mt][c]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) v[q] += __shfl_xor(v[q], m, 64);
                if (j == 0) red[wave][kq][mt][c][q] = v[q];
            }
        }
    __syncthreads();
    float* pg = p.partial + ((int64_t)b * gridDim.y + blockIdx.y) * 4 * hd + (int64_t)head * p.dk;
    for (int e = threadIdx.x; e < 4 * NMT * 4 * 4; e += blockDim.x) {
        const int q = e & 3, c = (e >> 2) & 3, mt = (e >> 4) % NMT, kq2 = e / (16 * NMT);
        const int v = 16 * mt + 4 * kq2 + c;
        if (v >= p.dk) continue;
        const float sum = ((red[0][kq2][mt][c][q] + red[1][kq2][mt][c][q]) + red[2][kq2][mt][c][q]) + red[3][kq2][mt][c][q];
        // q: 0 dg K, 1 db K, 2 dg V, 3 db V   ->  partial row [dg K | dg V | db K | db V], each h*dk wide
        pg[((q & 1) * 2 + (q >> 1)) * hd + v] = sum;
    }
}

// Q stream of the galerkin backward (not normalised): drop the coordinate / pad columns of dQ' [T][h][DP] into the Q block
// of d_qkv [T][3 h dk]
__global__ __launch_bounds__(256) void headtile_unpad_kernel(const float* __restrict__ src, float* __restrict__ d_qkv,
                                                             int64_t total4, int h, int dk, int p, int DP) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total4) return;
    const int Q4 = dk >> 2;
    const int q = (int)(e % Q4), head = (int)((e / Q4) % h);
    const int64_t t = e / ((int64_t)Q4 * h);
    const f32x4 v = tile_load4(src + (t * h + head) * DP + p + 4 * q, p);
    *reinterpret_cast<f32x4*>(d_qkv + t * 3 * h * dk + head * dk + 4 * q) = v;
}

static bool head_geom(int T, int h, int dk, int p, int norm_mask, int max_blocks, HeadGeom* g, int* threads,
                      int* blocks) {
    if (dk & 3) return false;
    int G = 1;
    while (G < dk / 4) G <<= 1;
    if (G > 64) return false;
    const int PT = 3 * h * G;
    if (PT > 1024) return false;
    // whole waves with no idle lanes when PT and the wave size have a small common multiple (PT = 96 -> 384)
    int lcm = PT;
    while (lcm % 64) lcm += PT;
    int thr = lcm <= 512 ? lcm * std::max(1, 384 / lcm) : std::max(256, ((PT + 63) / 64) * 64);
    const int R = thr / PT;
    int nblk = std::min(max_blocks, ceil_div(T, R * 8));
    nblk = std::max(nblk, 1);
    int tpb = ceil_div(T, nblk);
    tpb = ceil_div(tpb, R) * R;
    nblk = ceil_div(T, tpb);
    *g = HeadGeom{T, h, dk, p, (dk + p + 3) & ~3, norm_mask, G, PT, R, tpb};
    *threads = thr;
    *blocks = nblk;
    return true;
}

// ------------------------------------------------------------------------------------------ galerkin K^T V
// M[b,h] = K'^T V' over the tokens of one sample (layers.py:723), K', V' in the head-tile layout
// [T][h][DP] = [pos(p) | values(dk) | pad].  Streaming kernel: every token row is read exactly once,
// straight from HBM into MFMA operand registers (no LDS): lane (i = lane&15, k = lane>>4) of a wave holds
// K'[t0+k][p+16a+i] and V'[t0+k][p+16b+i] for 4 tokens per step, i.e. the A = K^T (16 x 4) and B = V
// (4 x 16) fragments of v_mfma_f32_16x16x4_f32.  The dk x dk core accumulates on the matrix pipe, the p-wide
// coordinate borders (P^T P, P^T V, K^T P) on the VALU beside it.  A block = 4 waves = 4 heads (looped if
// h > 4) of one token chunk of one sample; chunks write partial slabs that gt_galerkin_finalize_fwd sums.
template <int NB>
__global__ __launch_bounds__(256) void galerkin_ktv_kernel(const float* __restrict__ Kp, const float* __restrict__ Vp,
                                                           int n, int h, int DP, int p, int chunk,
                                                           float* __restrict__ slabs, int B,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int t_lo = ch * chunk, t_hi = min(n, t_lo + chunk);
    const int64_t hD = (int64_t)h * DP;
    for (int head = wave; head < h; head += 4) {
        f32x4 acc[NB][NB];
        float kp[NB][2], pv[NB][2], pp[2][2];
#pragma unroll
        for (int a = 0; a < NB; ++a) {
#pragma unroll
            for (int c = 0; c < NB; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            kp[a][0] = kp[a][1] = pv[a][0] = pv[a][1] = 0.f;
        }
        pp[0][0] = pp[0][1] = pp[1][0] = pp[1][1] = 0.f;
        const float* kb = Kp + ((int64_t)b * n) * hD + (int64_t)head * DP;
        const float* vb = Vp + ((int64_t)b * n) * hD + (int64_t)head * DP;
        // "plain" head tiles (gt_hip.h: hn_plain) hold the normalised values without the LayerNorm affine: K' = gamma_K xh +
        // beta_K is formed as the operand is loaded (gamma, beta [2][h][16 NB]: K then V); NULL = tiles hold K', V'
        float gk[NB], bk[NB], gv[NB], bv[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            gk[c] = gv[c] = 1.f;
            bk[c] = bv[c] = 0.f;
            if (gamma) {
                const int o = head * 16 * NB + 16 * c + i, hd = h * 16 * NB;
                gk[c] = gamma[o]; bk[c] = beta[o]; gv[c] = gamma[hd + o]; bv[c] = beta[hd + o];
            }
        }
        for (int tb = t_lo; tb < t_hi; tb += 16) {
          // 4 independent 4-token steps in flight: every load of the 16 tokens is requested before the first is used
          float a[4][NB], v[4][NB], pk[4][2];
          bool okk[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = tb + 4 * u + kq;
            okk[u] = t < t_hi;
            const float* kr = kb + (int64_t)t * hD;
            const float* vr = vb + (int64_t)t * hD;
            pk[u][0] = pk[u][1] = 0.f;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                a[u][c] = okk[u] ? kr[p + 16 * c + i] : 0.f;
                v[u][c] = okk[u] ? vr[p + 16 * c + i] : 0.f;
            }
            if (p > 0) pk[u][0] = okk[u] ? kr[0] : 0.f;
            if (p > 1) pk[u][1] = okk[u] ? kr[1] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (gamma) {                                       // plain tiles: the LayerNorm affine on the way in
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    a[u][c] = okk[u] ? fmaf(a[u][c], gk[c], bk[c]) : 0.f;
                    v[u][c] = okk[u] ? fmaf(v[u][c], gv[c], bv[c]) : 0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < NB; ++c)
#pragma unroll
                for (int e = 0; e < NB; ++e)
                    acc[c][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c], v[u][e], acc[c][e], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                kp[c][0] = fmaf(a[u][c], pk[u][0], kp[c][0]); kp[c][1] = fmaf(a[u][c], pk[u][1], kp[c][1]);
                pv[c][0] = fmaf(pk[u][0], v[u][c], pv[c][0]); pv[c][1] = fmaf(pk[u][1], v[u][c], pv[c][1]);
            }
            pp[0][0] = fmaf(pk[u][0], pk[u][0], pp[0][0]); pp[0][1] = fmaf(pk[u][0], pk[u][1], pp[0][1]);
            pp[1][0] = fmaf(pk[u][1], pk[u][0], pp[1][0]); pp[1][1] = fmaf(pk[u][1], pk[u][1], pp[1][1]);
          }
        }
        // borders: combine the 4 token lanes
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                kp[c][e] += __shfl_xor(kp[c][e], 16, 64); kp[c][e] += __shfl_xor(kp[c][e], 32, 64);
                pv[c][e] += __shfl_xor(pv[c][e], 16, 64); pv[c][e] += __shfl_xor(pv[c][e], 32, 64);
            }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 2; ++e) { pp[c][e] += __shfl_xor(pp[c][e], 16, 64); pp[c][e] += __shfl_xor(pp[c][e], 32, 64); }

        float* M = slabs + ((((int64_t)ch * B + b) * h + head) * DP) * DP;
        const int Dr = p + 16 * NB;
        // core: D layout of the 16x16 tile: row = 4*kq + r, col = i
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int e = 0; e < NB; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) M[(int64_t)(p + 16 * c + 4 * kq + r) * DP + p + 16 * e + i] = acc[c][e][r];
        if (kq == 0) {
#pragma unroll
            for (int c = 0; c < NB; ++c)
                for (int e = 0; e < p; ++e) {
                    M[(int64_t)(p + 16 * c + i) * DP + e] = kp[c][e];         // K^T P
                    M[(int64_t)e * DP + p + 16 * c + i] = pv[c][e];           // P^T V
                }
            if (i == 0)
                for (int c = 0; c < p; ++c)
                    for (int e = 0; e < p; ++e) M[(int64_t)c * DP + e] = pp[c][e];
        }
        for (int e = lane; e < DP * DP; e += 64) {               // zero padding rows / columns
            const int rr = e / DP, cc = e % DP;
            if (rr >= Dr || cc >= Dr) M[e] = 0.f;
        }
    }
}

// Same product with the token rows staged through LDS.  The kernel above feeds the MFMA operands with 4-byte loads of 64-byte
// row pieces at an 8-byte offset (the coordinates sit in front of the values): 2.4-2.8 TB/s.  A tile of 16 tokens of all h
// heads is ONE contiguous 16 * h * DP * 4-byte piece of the head-tile array, so the block copies it with 16-byte loads
// (every byte of every line used, one request per 1 KiB) into LDS, register-staged one tile ahead, and the waves (one head
// each) read their operands from there (consecutive lanes on consecutive banks).  h <= 4, 16 * h * DP floats <= 4096.
constexpr int KTV_TT = 16;          // tokens per tile
constexpr int KTV_MAXG = 4;         // 16-byte granules per thread and operand tile (h * DP <= 256)
template <int NB>
__global__ __launch_bounds__(256) void galerkin_ktv_lds_kernel(const float* __restrict__ Kp, const float* __restrict__ Vp,
                                                               int n, int h, int DP, int p, int chunk,
                                                               float* __restrict__ slabs, int B,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta) {
    extern __shared__ __attribute__((aligned(16))) float ktv_lds[];      // [2 buffers][K | V][KTV_TT * hD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int i = lane & 15, kq = lane >> 4;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int t_lo = ch * chunk, t_hi = min(n, t_lo + chunk);
    const int hD = h * DP, tile_f = KTV_TT * hD, ng = tile_f >> 2;
    const int head = wave;
    const bool active = head < h;

    f32x4 acc[NB][NB];
    float kp[NB][2], pv[NB][2], pp[2][2];
#pragma unroll
    for (int a = 0; a < NB; ++a) {
#pragma unroll
        for (int c = 0; c < NB; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        kp[a][0] = kp[a][1] = pv[a][0] = pv[a][1] = 0.f;
    }
    pp[0][0] = pp[0][1] = pp[1][0] = pp[1][1] = 0.f;
    float gk[NB], bk[NB], gv[NB], bv[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        gk[c] = gv[c] = 1.f;
        bk[c] = bv[c] = 0.f;
        if (gamma && active) {
            const int o = head * 16 * NB + 16 * c + i, hd = h * 16 * NB;
            gk[c] = gamma[o]; bk[c] = beta[o]; gv[c] = gamma[hd + o]; bv[c] = beta[hd + o];
        }
    }
    const float* kbase = Kp + (int64_t)b * n * hD;
    const float* vbase = Vp + (int64_t)b * n * hD;
    f32x4 rk[KTV_MAXG], rv[KTV_MAXG];
    auto gload = [&](int tb) {                    // tile tb .. tb + 15 -> registers (zeros past the chunk)
        const int valid_f = min(KTV_TT, t_hi - tb) * hD;
#pragma unroll
        for (int q = 0; q < KTV_MAXG; ++q) {
            const int g4 = (tid + 256 * q) * 4;
            const bool ok = g4 < valid_f;         // granules never straddle tokens (hD % 4 == 0)
            rk[q] = ok ? *reinterpret_cast<const f32x4*>(kbase + (int64_t)tb * hD + g4) : f32x4{0.f, 0.f, 0.f, 0.f};
            rv[q] = ok ? *reinterpret_cast<const f32x4*>(vbase + (int64_t)tb * hD + g4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto sstore = [&](int buf) {
        float* ks = ktv_lds + buf * 2 * tile_f;
#pragma unroll
        for (int q = 0; q < KTV_MAXG; ++q) {
            const int g = tid + 256 * q;
            if (g < ng) {
                *reinterpret_cast<f32x4*>(ks + 4 * g) = rk[q];
                *reinterpret_cast<f32x4*>(ks + tile_f + 4 * g) = rv[q];
            }
        }
    };
    int buf = 0;
    if (t_lo < t_hi) gload(t_lo);
    for (int tb = t_lo; tb < t_hi; tb += KTV_TT) {
        sstore(buf);
        __syncthreads();                           // tile tb is in LDS; everybody is done with the other buffer
        if (tb + KTV_TT < t_hi) gload(tb + KTV_TT);
        if (active) {
            const float* ks = ktv_lds + buf * 2 * tile_f + head * DP;
            const float* vs = ks + tile_f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = 4 * u + kq;
                const bool ok = tb + tt < t_hi;
                float a[NB], v[NB], pk[2] = {0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    a[c] = ks[tt * hD + p + 16 * c + i];
                    v[c] = vs[tt * hD + p + 16 * c + i];
                    if (gamma) {                   // plain tiles: the LayerNorm affine on the way in (rows past the chunk stay 0)
                        a[c] = ok ? fmaf(a[c], gk[c], bk[c]) : 0.f;
                        v[c] = ok ? fmaf(v[c], gv[c], bv[c]) : 0.f;
                    }
                }
                if (p > 0) pk[0] = ks[tt * hD];
                if (p > 1) pk[1] = ks[tt * hD + 1];
#pragma unroll
                for (int c = 0; c < NB; ++c)
#pragma unroll
                    for (int e = 0; e < NB; ++e)
                        acc[c][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], v[e], acc[c][e], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    kp[c][0] = fmaf(a[c], pk[0], kp[c][0]); kp[c][1] = fmaf(a[c], pk[1], kp[c][1]);
                    pv[c][0] = fmaf(pk[0], v[c], pv[c][0]); pv[c][1] = fmaf(pk[1], v[c], pv[c][1]);
                }
                pp[0][0] = fmaf(pk[0], pk[0], pp[0][0]); pp[0][1] = fmaf(pk[0], pk[1], pp[0][1]);
                pp[1][0] = fmaf(pk[1], pk[0], pp[1][0]); pp[1][1] = fmaf(pk[1], pk[1], pp[1][1]);
            }
        }
        buf ^= 1;
    }
    if (!active) return;
    // borders: combine the 4 token lanes
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            kp[c][e] += __shfl_xor(kp[c][e], 16, 64); kp[c][e] += __shfl_xor(kp[c][e], 32, 64);
            pv[c][e] += __shfl_xor(pv[c][e], 16, 64); pv[c][e] += __shfl_xor(pv[c][e], 32, 64);
        }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) { pp[c][e] += __shfl_xor(pp[c][e], 16, 64); pp[c][e] += __shfl_xor(pp[c][e], 32, 64); }

    float* M = slabs + ((((int64_t)ch * B + b) * h + head) * DP) * DP;
    const int Dr = p + 16 * NB;
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int e = 0; e < NB; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(int64_t)(p + 16 * c + 4 * kq + r) * DP + p + 16 * e + i] = acc[c][e][r];
    if (kq == 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c)
            for (int e = 0; e < p; ++e) {
                M[(int64_t)(p + 16 * c + i) * DP + e] = kp[c][e];         // K^T P
                M[(int64_t)e * DP + p + 16 * c + i] = pv[c][e];           // P^T V
            }
        if (i == 0)
            for (int c = 0; c < p; ++c)
                for (int e = 0; e < p; ++e) M[(int64_t)c * DP + e] = pp[c][e];
    }
    for (int e = lane; e < DP * DP; e += 64) {               // zero padding rows / columns
        const int rr = e / DP, cc = e % DP;
        if (rr >= Dr || cc >= Dr) M[e] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------ galerkin finalize
// One block per (batch, head, group of FIN_RB rows of M): row j of P needs row j of M only.  (Round 5: one block per
// (batch, head) walked the whole 52 x 52 matrix with n_slabs dependent loads per element -- 178 us at ex4's 16 x 64 slabs.)
// Rows per block: a quarter of the matrix (round 6; four until then).  Every block stages all of W_h (d x DP floats: 18 KB at
// d = 128, 40 KB at d = 192) for its rows' products -- with four rows per block that staging was most of the kernel's traffic
// (4 608 blocks x 18 KB at C2, 6 656 x 40 KB at C4).
static inline int fin_rows_per_block(int DP) { return std::max(4, (DP + 3) / 4); }
__global__ __launch_bounds__(256) void galerkin_fin_fwd_kernel(
    const float* __restrict__ slabs, int n_slabs, int64_t slab_stride, int h, int DP, int Dr, int d,
    float inv_n, const float* __restrict__ mask, DropDev drop, const float* __restrict__ Wfc,
    float* __restrict__ Mt, float* __restrict__ P, float* __restrict__ Pv, int pdim, int FIN_RB) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // Both operands of the product loop are read as 16-byte vectors along the contraction (DP % 4 == 0; W_h's rows are padded
    // to DP columns with zeros): a quarter of the LDS instructions of the scalar loop, which was what the kernel waited for.
    float* sM = lds;                    // [FIN_RB][DP]
    float* sW = lds + FIN_RB * DP;      // [d][DP]
    const int bh = blockIdx.x, b = bh / h, hh = bh % h;
    const int j0 = blockIdx.y * FIN_RB, nr = min(FIN_RB, DP - j0);
    const uint32_t key = drop_key_dev(drop);
    const int64_t mo = (int64_t)bh * DP * DP;
    for (int le = threadIdx.x; le < nr * DP; le += blockDim.x) {
        const int e = j0 * DP + le, j = e / DP, c = e % DP;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;           // four independent chains: the loads overlap
        int k = 0;
        for (; k + 4 <= n_slabs; k += 4) {
            s0 += slabs[(k + 0) * slab_stride + mo + e];
            s1 += slabs[(k + 1) * slab_stride + mo + e];
            s2 += slabs[(k + 2) * slab_stride + mo + e];
            s3 += slabs[(k + 3) * slab_stride + mo + e];
        }
        for (; k < n_slabs; ++k) s0 += slabs[k * slab_stride + mo + e];
        const float s = (s0 + s1) + (s2 + s3);
        float mul = inv_n;
        if (mask) mul *= mask[mo + e];
        else if (drop.thresh) mul *= drop_mul(drop, key, (uint32_t)(mo + e));
        const float v = (j < Dr && c < Dr) ? s * mul : 0.f;
        sM[le] = v;
        Mt[mo + e] = v;
    }
    for (int e = threadIdx.x; e < d * DP; e += blockDim.x) {
        const int c = e / DP, ee = e % DP;
        sW[e] = ee < Dr ? Wfc[(int64_t)c * (h * Dr) + hh * Dr + ee] : 0.f;
    }
    __syncthreads();
    float* Pb = P + ((int64_t)b * h * DP + (int64_t)hh * DP) * d;
    for (int le = threadIdx.x; le < nr * d; le += blockDim.x) {
        const int jl = le / d, c = le % d, j = j0 + jl;
        float acc = 0.f;
        if (j < Dr) {
            const f32x4* mr = reinterpret_cast<const f32x4*>(sM + jl * DP);
            const f32x4* wr = reinterpret_cast<const f32x4*>(sW + c * DP);
            for (int q = 0; q < (DP >> 2); ++q) {           // the scalar loop's order (the zero columns behind Dr add nothing)
                const f32x4 m4 = mr[q], w4 = wr[q];
                acc = fmaf(m4[0], w4[0], acc); acc = fmaf(m4[1], w4[1], acc);
                acc = fmaf(m4[2], w4[2], acc); acc = fmaf(m4[3], w4[3], acc);
            }
        }
        Pb[(int64_t)j * d + c] = acc;
        // the value rows of P once more, compact [B][h dk][d]: the B operand of the backward's dQ product (it used to be
        // sliced out of P by an ATen copy in every backward)
        if (Pv && j >= pdim && j < Dr) Pv[((int64_t)b * h * (Dr - pdim) + (int64_t)hh * (Dr - pdim) + (j - pdim)) * d + c] = acc;
    }
}

__global__ __launch_bounds__(256) void galerkin_fin_bwd_kernel(
    const float* __restrict__ dPt, const float* __restrict__ Mt, const float* __restrict__ mask,
    DropDev drop, const float* __restrict__ Wfc, int h, int DP, int Dr, int d, float inv_n,
    float* __restrict__ dM, float* __restrict__ dWfc_slabs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // gridDim.y blocks share the two output loops of one (batch, head): part q owns ROWS [j0, j1) of dM and feature COLUMNS
    // [c0, c1) of dWfc, and stages only what those need -- the row slice and the column slice of dP_h, all of W_h and M.
    // (Round 6: every part used to stage all of dP_h; at d = 192, DP = 52 that was 89 KB of LDS, one block per CU and
    // 252 us per launch -- now 69 KB at four parts, two blocks per CU.)
    const int part = blockIdx.y, parts = gridDim.y;
    const int jr = (DP + parts - 1) / parts, j0 = part * jr, j1 = min(DP, j0 + jr), nj = max(0, j1 - j0);
    const int cr = (d + parts - 1) / parts, c0 = part * cr, c1 = min(d, c0 + cr), nc = max(0, c1 - c0);
    const int dpitch = d + 1, cpitch = cr + 1;
    // W_h (rows padded to DP columns with zeros) and M first: both are read as 16-byte vectors along ee (four outputs per
    // thread: one broadcast scalar + one vector read per four FMAs, where the scalar loops issued two reads per FMA)
    float* sW = lds;                        // [d][DP]
    float* sM = sW + d * DP;                // [DP][DP]
    float* sdr = sM + DP * DP;              // [jr][d+1]    dP_h[j0 + j][c]        (rows of this part, every feature)
    float* sdc = parts == 1 ? sdr : sdr + jr * dpitch;    // [DP][cr+1]   dP_h[j][c0 + c]   (every row, features of this part;
                                                          //  one part: the same image as sdr)
    const int bh = blockIdx.x, b = bh / h, hh = bh % h;
    const uint32_t key = drop_key_dev(drop);
    const int64_t mo = (int64_t)bh * DP * DP;
    const float* src = dPt + (int64_t)b * d * (h * DP) + hh * DP;
    for (int e = threadIdx.x; e < d * nj; e += blockDim.x) {
        const int c = e / nj, j = e % nj;
        sdr[j * dpitch + c] = src[(int64_t)c * (h * DP) + j0 + j];
    }
    if (parts > 1)
        for (int e = threadIdx.x; e < nc * DP; e += blockDim.x) {
            const int c = e / DP, j = e % DP;
            sdc[j * cpitch + c] = src[(int64_t)(c0 + c) * (h * DP) + j];
        }
    for (int e = threadIdx.x; e < d * DP; e += blockDim.x) {
        const int c = e / DP, ee = e % DP;
        sW[e] = ee < Dr ? Wfc[(int64_t)c * (h * Dr) + hh * Dr + ee] : 0.f;
    }
    for (int e = threadIdx.x; e < DP * DP; e += blockDim.x) sM[e] = Mt[mo + e];
    __syncthreads();
    const int Q4 = DP >> 2;
    for (int e = threadIdx.x; e < nj * Q4; e += blockDim.x) {
        const int jl = e / Q4, q = e - jl * Q4, j = j0 + jl;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (j < Dr) {
            const float* dp = sdr + jl * dpitch;
            for (int c = 0; c < d; ++c) acc += dp[c] * *reinterpret_cast<const f32x4*>(sW + c * DP + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ee = 4 * q + t;
                float mul = inv_n;
                if (mask) mul *= mask[mo + j * DP + ee];
                else if (drop.thresh) mul *= drop_mul(drop, key, (uint32_t)(mo + j * DP + ee));
                acc[t] = ee < Dr ? acc[t] * mul : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(dM + mo + j * DP + 4 * q) = acc;
    }
    float* dst = dWfc_slabs + (int64_t)b * d * (h * Dr) + hh * Dr;
    for (int e = threadIdx.x; e < nc * Q4; e += blockDim.x) {
        const int cl = e / Q4, q = e - cl * Q4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < Dr; ++j) acc += sdc[j * cpitch + cl] * *reinterpret_cast<const f32x4*>(sM + j * DP + 4 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * q + t < Dr) dst[(int64_t)(c0 + cl) * (h * Dr) + 4 * q + t] = acc[t];
    }
}

// ------------------------------------------------------------------------------------------ row layernorm
// one wave per row, 4 rows per block
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int T, int d,
                                                            float eps, float* __restrict__ y,
                                                            float* __restrict__ stats) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    if (row >= T) return;
    const float* xr = x + (int64_t)row * d;
    float s = 0.f;
    for (int j = lane; j < d; j += 64) s += xr[j];
    const float mu = wave_sum(s) / d;
    float v = 0.f;
    for (int j = lane; j < d; j += 64) { const float c = xr[j] - mu; v += c * c; }
    const float rstd = 1.f / sqrtf(wave_sum(v) / d + eps);
    float* yr = y + (int64_t)row * d;
    for (int j = lane; j < d; j += 64) yr[j] = (xr[j] - mu) * rstd * gamma[j] + beta[j];
    if (lane == 0) { stats[2 * (int64_t)row] = mu; stats[2 * (int64_t)row + 1] = rstd; }
}

// Narrow rows (d <= 64, d % 4 == 0; ex4's d = 48): a row is 16 lanes x one float4 each, four rows per wave -- the generic
// kernels below spend a wave, twelve cross-lane exchanges and (backward) an LDS read-modify-write per element on one 192-byte
// row (layernorm_bwd 50.6 us, layernorm_fwd 19.1 us for [65536, 48]: profiles/r06e_rocprofv3_steady_ex4_ns_after_dkv_fin.txt).
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}
__global__ __launch_bounds__(256) void layernorm_fwd16_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int T, int d, float eps,
                                                              float* __restrict__ y, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63, q = lane & 15, rg = (threadIdx.x >> 4);      // 16 row slots per block
    const bool on = 4 * q < d;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 g = on ? *reinterpret_cast<const f32x4*>(gamma + 4 * q) : z;
    const f32x4 be = on ? *reinterpret_cast<const f32x4*>(beta + 4 * q) : z;
    const float inv_d = 1.f / (float)d;
    for (int64_t row = (int64_t)blockIdx.x * 16 + rg; row < T; row += (int64_t)gridDim.x * 16) {
        const f32x4 v = on ? *reinterpret_cast<const f32x4*>(x + row * d + 4 * q) : z;
        const float mu = group16_sum((v[0] + v[1]) + (v[2] + v[3])) * inv_d;
        f32x4 c = v - mu;
        if (!on) c = z;
        const float var = group16_sum((c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3])) * inv_d;
        const float rstd = 1.f / sqrtf(var + eps);
        if (on) *reinterpret_cast<f32x4*>(y + row * d + 4 * q) = c * rstd * g + be;
        if (q == 0) { stats[2 * row] = mu; stats[2 * row + 1] = rstd; }
    }
}
__global__ __launch_bounds__(256) void layernorm_bwd16_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ stats, int T, int d, float* __restrict__ dx, float* __restrict__ partial /* [nblk][2][d] */) {
    __shared__ __attribute__((aligned(16))) float red[4][2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane & 15, rg = (threadIdx.x >> 4);
    const bool on = 4 * q < d;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 g = on ? *reinterpret_cast<const f32x4*>(gamma + 4 * q) : z;
    const float inv_d = 1.f / (float)d;
    f32x4 mg = z, mb = z;
    for (int64_t row = (int64_t)blockIdx.x * 16 + rg; row < T; row += (int64_t)gridDim.x * 16) {
        const float mu = stats[2 * row], rstd = stats[2 * row + 1];
        const f32x4 xv = on ? *reinterpret_cast<const f32x4*>(x + row * d + 4 * q) : z;
        const f32x4 gr = on ? *reinterpret_cast<const f32x4*>(dy + row * d + 4 * q) : z;
        f32x4 xh = (xv - mu) * rstd;
        if (!on) xh = z;
        const f32x4 gg = gr * g, gx = gg * xh;
        const float a1 = group16_sum((gg[0] + gg[1]) + (gg[2] + gg[3])) * inv_d;
        const float a2 = group16_sum((gx[0] + gx[1]) + (gx[2] + gx[3])) * inv_d;
        mg += gr * xh;
        mb += gr;
        if (on) *reinterpret_cast<f32x4*>(dx + row * d + 4 * q) = (gg - a1 - xh * a2) * rstd;
    }
    // the wave's four row slots (lanes q, q + 16, q + 32, q + 48), then the four waves through LDS: fixed order
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        mg[c] += __shfl_xor(mg[c], 16, 64); mg[c] += __shfl_xor(mg[c], 32, 64);
        mb[c] += __shfl_xor(mb[c], 16, 64); mb[c] += __shfl_xor(mb[c], 32, 64);
    }
    if (lane < 16) {
        *reinterpret_cast<f32x4*>(&red[w][0][4 * q]) = mg;
        *reinterpret_cast<f32x4*>(&red[w][1][4 * q]) = mb;
    }
    __syncthreads();
    float* pg = partial + (int64_t)blockIdx.x * 2 * d;
    for (int jj = threadIdx.x; jj < 2 * d; jj += blockDim.x) {
        const int which = jj / d, col = jj % d;
        pg[jj] = (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
    }
}
static inline bool ln_narrow(const void* a, const void* b, const void* c, int d) {
    return d <= 64 && d % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

constexpr int LN_ROWS = 64;   // rows per block in backward (partial dgamma/dbeta per block)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ stats, int T, int d, float* __restrict__ dx,
    float* __restrict__ partial /* [nblk][2][d] */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [4][2][d]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* mg = lds + (w * 2) * d;
    float* mb = mg + d;
    for (int j = lane; j < d; j += 64) { mg[j] = 0.f; mb[j] = 0.f; }
    for (int r0 = blockIdx.x * LN_ROWS; r0 < T; r0 += gridDim.x * LN_ROWS) {
    const int r1 = min(T, r0 + LN_ROWS);
    for (int row = r0 + w; row < r1; row += 4) {
        const float mu = stats[2 * (int64_t)row], rstd = stats[2 * (int64_t)row + 1];
        const float* xr = x + (int64_t)row * d;
        const float* gr = dy + (int64_t)row * d;
        float a1 = 0.f, a2 = 0.f;
        for (int j = lane; j < d; j += 64) {
            const float xh = (xr[j] - mu) * rstd, gg = gr[j] * gamma[j];
            a1 += gg;
            a2 += gg * xh;
            mg[j] += gr[j] * xh;
            mb[j] += gr[j];
        }
        a1 = wave_sum(a1) / d;
        a2 = wave_sum(a2) / d;
        float* dr = dx + (int64_t)row * d;
        for (int j = lane; j < d; j += 64) {
            const float xh = (xr[j] - mu) * rstd;
            dr[j] = rstd * (gr[j] * gamma[j] - a1 - xh * a2);
        }
    }
    }   // row groups
    __syncthreads();
    float* pg = partial + (int64_t)blockIdx.x * 2 * d;
    for (int j = threadIdx.x; j < 2 * d; j += blockDim.x)
        pg[j] = lds[j] + lds[2 * d + j] + lds[4 * d + j] + lds[6 * d + j];
}

// ------------------------------------------------------------------------------------------ mode mixing
// One block per retained mode q.  X: [B][2][Qx][Cin], Y: [B][2][Qy][Cout] (re plane, im plane),
// W: [Cin][Cout][Q][2].  Complex product, no conjugate (layers.py:1143-1151).
constexpr int MM_BCH = 8;   // batch entries staged per pass
__global__ __launch_bounds__(256) void modemix_fwd_kernel(const float* __restrict__ X,
                                                          const float* __restrict__ W, int B, int Q,
                                                          int Cin, int Cout, int64_t xbs, int64_t ybs,
                                                          int Qx, int Qy, int qoff,
                                                          float* __restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sWr = lds;                       // [Cin][Cout]
    float* sWi = sWr + Cin * Cout;          // [Cin][Cout]
    float* sX = sWi + Cin * Cout;           // [MM_BCH][2][Cin]
    const int q = blockIdx.x;
    for (int e = threadIdx.x; e < Cin * Cout; e += blockDim.x) {
        const float2 w = *reinterpret_cast<const float2*>(W + ((int64_t)e * Q + q) * 2);
        sWr[e] = w.x;
        sWi[e] = w.y;
    }
    // the batch is cut into gridDim.y slices: Q = m*m blocks alone (144 for the Darcy decoder) leave the chip half empty
    const int bchunk = (B + gridDim.y - 1) / gridDim.y;
    const int bend = min(B, (int)(blockIdx.y + 1) * bchunk);
    for (int bb = blockIdx.y * bchunk; bb < bend; bb += MM_BCH) {
        const int nb = min(MM_BCH, bend - bb);
        __syncthreads();
        for (int e = threadIdx.x; e < nb * 2 * Cin; e += blockDim.x) {
            const int b = e / (2 * Cin), ri = (e / Cin) & 1, i = e % Cin;
            sX[e] = X[(int64_t)(bb + b) * xbs + ((int64_t)ri * Qx + qoff + q) * Cin + i];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nb * Cout; e += blockDim.x) {
            const int b = e / Cout, o = e % Cout;
            const float* xr = sX + b * 2 * Cin;
            const float* xi = xr + Cin;
            float yr = 0.f, yi = 0.f;
            for (int i = 0; i < Cin; ++i) {
                const float wr = sWr[i * Cout + o], wi = sWi[i * Cout + o];
                yr = fmaf(xr[i], wr, yr); yr = fmaf(-xi[i], wi, yr);
                yi = fmaf(xi[i], wr, yi); yi = fmaf(xr[i], wi, yi);
            }
            float* yp = Y + (int64_t)(bb + b) * ybs + ((int64_t)qoff + q) * Cout + o;
            yp[0] = yr;
            yp[(int64_t)Qy * Cout] = yi;
        }
    }
}

template <int MAXP>      // (i, o) weight-gradient pairs per thread: Cin * Cout <= 256 * MAXP
__global__ __launch_bounds__(256) void modemix_bwd_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ dY, int B, int Q,
    int Cin, int Cout, int64_t xbs, int64_t ybs, int Qx, int Qy, int qoff, float* __restrict__ dX,
    float* __restrict__ dW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // weight rows padded by one float: the dX loop below reads W[i][o] with the lanes running over i -- at a row pitch of
    // Cout = 32 floats every lane of a wave hit the same LDS bank (a 32-way conflict on both reads of each of the Cout steps)
    const int CW = Cout + 1;
    float* sWr = lds;                       // [Cin][Cout + 1]
    float* sWi = sWr + Cin * CW;
    float* sX = sWi + Cin * CW;             // [MM_BCH][2][Cin]
    float* sG = sX + MM_BCH * 2 * Cin;      // [MM_BCH][2][Cout]
    const int q = blockIdx.x;
    for (int e = threadIdx.x; e < Cin * Cout; e += blockDim.x) {
        const float2 w = *reinterpret_cast<const float2*>(W + ((int64_t)e * Q + q) * 2);
        const int i = e / Cout, o = e - i * Cout;
        sWr[i * CW + o] = w.x;
        sWi[i * CW + o] = w.y;
    }
    // each thread owns up to MAXP (i,o) pairs of dW, accumulated over the whole batch in registers
    float gr[MAXP], gi[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) gr[k] = gi[k] = 0.f;
    const int bchunk = (B + gridDim.y - 1) / gridDim.y;      // batch slice of this block (dW: one partial per slice)
    const int bend = min(B, (int)(blockIdx.y + 1) * bchunk);
    for (int bb = blockIdx.y * bchunk; bb < bend; bb += MM_BCH) {
        const int nb = min(MM_BCH, bend - bb);
        __syncthreads();
        for (int e = threadIdx.x; e < nb * 2 * Cin; e += blockDim.x) {
            const int b = e / (2 * Cin), ri = (e / Cin) & 1, i = e % Cin;
            sX[e] = X[(int64_t)(bb + b) * xbs + ((int64_t)ri * Qx + qoff + q) * Cin + i];
        }
        for (int e = threadIdx.x; e < nb * 2 * Cout; e += blockDim.x) {
            const int b = e / (2 * Cout), ri = (e / Cout) & 1, o = e % Cout;
            sG[e] = dY[(int64_t)(bb + b) * ybs + ((int64_t)ri * Qy + qoff + q) * Cout + o];
        }
        __syncthreads();
        // dX[b][.][q][i] = sum_o dY (x) conj(W)
        for (int e = threadIdx.x; e < nb * Cin; e += blockDim.x) {
            const int b = e / Cin, i = e % Cin;
            const float* g_r = sG + b * 2 * Cout;
            const float* g_i = g_r + Cout;
            float xr = 0.f, xi = 0.f;
            for (int o = 0; o < Cout; ++o) {
                const float wr = sWr[i * CW + o], wi = sWi[i * CW + o];
                xr = fmaf(g_r[o], wr, xr); xr = fmaf(g_i[o], wi, xr);
                xi = fmaf(g_i[o], wr, xi); xi = fmaf(-g_r[o], wi, xi);
            }
            float* xp = dX + (int64_t)(bb + b) * xbs + ((int64_t)qoff + q) * Cin + i;
            xp[0] = xr;
            xp[(int64_t)Qx * Cin] = xi;
        }
        // dW[i][o] += sum_b conj(X) (x) dY
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int e = threadIdx.x + k * 256;
            if (e < Cin * Cout) {
                const int i = e / Cout, o = e % Cout;
                for (int b = 0; b < nb; ++b) {
                    const float xr = sX[b * 2 * Cin + i], xi = sX[b * 2 * Cin + Cin + i];
                    const float g_r = sG[b * 2 * Cout + o], g_i = sG[b * 2 * Cout + Cout + o];
                    gr[k] = fmaf(xr, g_r, gr[k]); gr[k] = fmaf(xi, g_i, gr[k]);
                    gi[k] = fmaf(xr, g_i, gi[k]); gi[k] = fmaf(-xi, g_r, gi[k]);
                }
            }
        }
    }
    float* dWs = dW + (int64_t)blockIdx.y * Cin * Cout * Q * 2;       // slab of this batch slice (gridDim.y == 1: dW itself)
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int e = threadIdx.x + k * 256;
        if (e < Cin * Cout)
            *reinterpret_cast<float2*>(dWs + ((int64_t)e * Q + q) * 2) = make_float2(gr[k], gi[k]);
    }
}

// batch slices per mode block: enough blocks for ~3 per CU, at least 8 samples per slice
static inline int modemix_slices(int B, int Q) { return std::max(1, std::min(ceil_div(768, Q), ceil_div(B, 8))); }

// kernels whose dynamic LDS may exceed 64 KiB opt in once (host-side attribute, not a stream op)
template <typename K>
static int allow_big_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    if (bytes > 160 * 1024) return GT_ENOTSUP;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e == hipSuccess ? 0 : (int)e;
}

static inline int grid_for(int64_t n, int block = 256, int cap = 4096) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((n + block - 1) / block, cap));
}

}  // namespace gt

using namespace gt;

extern "C" int gt_abi_version(void) { return GT_ABI_VERSION; }
extern "C" const char* gt_target_arch(void) { return "gfx950"; }

extern "C" int gt_seed_advance(uint64_t* seed, uint64_t inc, void* stream) {
    if (!seed) return GT_EINVAL;
    hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, seed, inc);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_dropout_apply(const float* x, float* out, int64_t n, const gt_dropout* d, void* stream) {
    if (!x || !out || n < 0) return GT_EINVAL;
    if (d && d->p > 0.f && !d->seed) return GT_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, out,
                       n, make_drop(d));
    GT_LAUNCH_CHECK();
    return 0;
}

static int dropact_launch(bool bwd, const float* x, const float* gy, float* out, int64_t n, const gt_dropout* d1,
                          int32_t act1, const gt_dropout* d2, int32_t act2, void* stream) {
    if (!x || !out || n < 0 || (bwd && !gy)) return GT_EINVAL;
    if ((d1 && d1->p > 0.f && !d1->seed) || (d2 && d2->p > 0.f && !d2->seed)) return GT_EINVAL;
    if (act1 < GT_ACT_NONE || act1 > GT_ACT_GELU || act2 < GT_ACT_NONE || act2 > GT_ACT_GELU) return GT_EINVAL;
    if (n == 0) return 0;
    const int vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const int grid = grid_for((n + 3) / 4, 256, 8192);
    if (bwd) hipLaunchKernelGGL(dropact_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gy, out, n,
                                make_drop(d1), act1, make_drop(d2), act2, vec);
    else hipLaunchKernelGGL(dropact_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gy, out, n,
                            make_drop(d1), act1, make_drop(d2), act2, vec);
    GT_LAUNCH_CHECK();
    return 0;
}
extern "C" int gt_dropact_fwd(const float* x, float* y, int64_t n, const gt_dropout* d1, int32_t act1,
                              const gt_dropout* d2, int32_t act2, void* stream) {
    return dropact_launch(false, x, nullptr, y, n, d1, act1, d2, act2, stream);
}
extern "C" int gt_dropact_bwd(const float* x, const float* gy, float* gx, int64_t n, const gt_dropout* d1,
                              int32_t act1, const gt_dropout* d2, int32_t act2, void* stream) {
    return dropact_launch(true, x, gy, gx, n, d1, act1, d2, act2, stream);
}

extern "C" int gt_slab_reduce(const float* slabs, int64_t stride, int32_t n_slabs, int64_t n, float alpha,
                              float* out, void* stream) {
    if (!slabs || !out || n_slabs <= 0 || n <= 0) return GT_EINVAL;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(ceil_div(n, 16)), dim3(256), 0, (hipStream_t)stream, slabs,
                       stride, n_slabs, n, alpha, out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_colsum(const float* A, int64_t lda, int32_t M, int32_t N, const gt_dropout* a_drop,
                         float a_sign, float* out, void* ws, int64_t ws_bytes, void* stream) {
    if (!A || !out || M <= 0 || N <= 0) return GT_EINVAL;
    if (a_drop && a_drop->p > 0.f && !a_drop->seed) return GT_EINVAL;
    const int colb = ceil_div(N, 64);
    const int chunks = std::max(1, std::min({ceil_div(M, 128), CS_MAXG, std::max(1, 768 / colb)}));
    if (!ws || ws_bytes < (int64_t)chunks * N * (int64_t)sizeof(float)) return GT_EWS;
    float* partial = reinterpret_cast<float*>(ws);
    const int vec = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((lda & 3) == 0);
    hipLaunchKernelGGL(colsum_kernel, dim3(colb, chunks), dim3(256), 0, (hipStream_t)stream, A, lda, M, N,
                       make_drop(a_drop, a_sign), vec, partial);
    GT_LAUNCH_CHECK();
    return gt_slab_reduce(partial, N, chunks, N, 1.f, out, stream);
}

extern "C" int gt_act_bwd(const float* dout, const float* pre, float* dpre, int64_t n, int32_t act,
                          void* stream) {
    if (!dout || !pre || !dpre || n <= 0) return GT_EINVAL;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dout, pre,
                       dpre, n, act);
    GT_LAUNCH_CHECK();
    return 0;
}

static inline int round4(int v) { return (v + 3) & ~3; }

extern "C" int gt_headnorm_fwd(const float* qkv, const float* pos, const float* gamma, const float* beta,
                               int32_t T, int32_t h, int32_t dk, int32_t p, int32_t norm_mask, float eps,
                               float* out, float* stats, void* stream) {
    if (!qkv || !out || T <= 0 || h <= 0 || dk <= 0 || p < 0) return GT_EINVAL;
    if (p > 0 && !pos) return GT_EINVAL;
    if (norm_mask & ~7) return GT_EINVAL;
    if (norm_mask && (!gamma || !beta || !stats)) return GT_EINVAL;
    const int DP = round4(dk + p);
    {
        HeadGeom g; int thr, nblk;
        const bool al = ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out) |
                          reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
                          reinterpret_cast<uintptr_t>(stats)) & 15) == 0;
        if (al && head_geom(T, h, dk, p, norm_mask, 1 << 20, &g, &thr, &nblk)) {
            hipLaunchKernelGGL(headnorm_fwd_v2_kernel, dim3(nblk), dim3(thr), 0, (hipStream_t)stream, qkv, pos,
                               gamma, beta, g, eps, out, stats);
            GT_LAUNCH_CHECK();
            return 0;
        }
    }
    const int tok = hn_tok(3 * h * (dk + 1));
    const size_t lds = (size_t)tok * 3 * h * (dk + 1) * sizeof(float);
    if (lds > 64 * 1024) return GT_ENOTSUP;
    hipLaunchKernelGGL(headnorm_fwd_kernel, dim3(ceil_div(T, tok)), dim3(256), lds, (hipStream_t)stream,
                       qkv, pos, gamma, beta, T, h, dk, p, DP, norm_mask, eps, out, stats, tok);
    GT_LAUNCH_CHECK();
    return 0;
}

static inline int hn_tok_bwd(int h, int dk) { return hn_tok(2 * 3 * h * (dk + 1) + 9 * h); }
constexpr int HN_MAXB = 1024;      // bound on blocks (= dgamma/dbeta partials) of the backward
static inline int hn_blocks_bwd(int T, int h, int dk) {
    return std::min(ceil_div(T, hn_tok_bwd(h, dk)), HN_MAXB);
}
extern "C" int64_t gt_headnorm_bwd_ws_bytes(int32_t T, int32_t h, int32_t dk) {
    (void)T;
    return (int64_t)HN_MAXB * 4 * h * dk * (int64_t)sizeof(float);      // upper bound for both kernels
}

extern "C" int gt_headnorm_bwd(const float* d_out, const float* qkv, const float* gamma, const float* stats,
                               int32_t T, int32_t h, int32_t dk, int32_t p, int32_t norm_mask, float* d_qkv,
                               float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, void* stream) {
    if (!d_out || !qkv || !d_qkv || T <= 0 || h <= 0 || dk <= 0 || p < 0) return GT_EINVAL;
    if (norm_mask & ~7) return GT_EINVAL;
    if (norm_mask && (!gamma || !stats || !dgamma || !dbeta)) return GT_EINVAL;
    if (!ws || ws_bytes < gt_headnorm_bwd_ws_bytes(T, h, dk)) return GT_EWS;
    const int DP = round4(dk + p);
    const int S = 3 * h;
    const int tok = hn_tok_bwd(h, dk);
    const size_t lds = ((size_t)2 * tok * S * (dk + 1) + 3 * tok * S) * sizeof(float);
    int nblk = hn_blocks_bwd(T, h, dk);
    float* partial = reinterpret_cast<float*>(ws);
    HeadGeom g; int thr, nb2;
    const bool al = ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(d_qkv) |
                      reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(stats) |
                      reinterpret_cast<uintptr_t>(d_out)) & 15) == 0;
    if (al && head_geom(T, h, dk, p, norm_mask, HN_MAXB, &g, &thr, &nb2)) {
        nblk = nb2;
        const size_t lds2 = (size_t)g.R * g.PT * 8 * sizeof(float);
        hipLaunchKernelGGL(headnorm_bwd_v2_kernel, dim3(nblk), dim3(thr), lds2, (hipStream_t)stream, d_out,
                           qkv, gamma, stats, g, d_qkv, partial);
    } else {
        if (lds > 64 * 1024) return GT_ENOTSUP;
        hipLaunchKernelGGL(headnorm_bwd_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream, d_out, qkv,
                           gamma, stats, T, h, dk, p, DP, norm_mask, d_qkv, partial, tok);
    }
    GT_LAUNCH_CHECK();
    if (norm_mask) {
        const int hd = h * dk;
        // partial: [nblk][ (dg: 2*hd) | (db: 2*hd) ]
        int rc = gt_slab_reduce(partial, 4 * hd, nblk, 2 * hd, 1.f, dgamma, stream);
        if (rc) return rc;
        rc = gt_slab_reduce(partial + 2 * hd, 4 * hd, nblk, 2 * hd, 1.f, dbeta, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int gt_galerkin_dkv(const float* Kp, const float* Vp, const float* dM, float* dKp, float* dVp,
                               int32_t B, int32_t n, int32_t h, int32_t DP, void* stream) {
    if (!Kp || !Vp || !dM || !dKp || !dVp || B <= 0 || n <= 0 || h <= 0) return GT_EINVAL;
    if (DP != 20 && DP != 36 && DP != 52) return GT_ENOTSUP;
    if ((reinterpret_cast<uintptr_t>(Kp) | reinterpret_cast<uintptr_t>(Vp) | reinterpret_cast<uintptr_t>(dKp) |
         reinterpret_cast<uintptr_t>(dVp)) & 15)
        return GT_EALIGN;
    DkvP p{Kp, Vp, dM, dKp, dVp, n, h};
    const int ntile = (n + 15) / 16;
    const int chunks = std::max(1, std::min({(1024 + B * h - 1) / (B * h), (ntile + 7) / 8, 65535}));
    dim3 grid((unsigned)(B * h), (unsigned)chunks);
    hipStream_t st = (hipStream_t)stream;
    if (DP == 20) hipLaunchKernelGGL(galerkin_dkv_kernel<1>, grid, dim3(256), 0, st, p);
    else if (DP == 36) hipLaunchKernelGGL(galerkin_dkv_kernel<2>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(galerkin_dkv_kernel<3>, grid, dim3(256), 0, st, p);
    GT_LAUNCH_CHECK();
    return 0;
}

static inline int dkv_ln_chunks(int B, int h) { return std::max(1, std::min(16, 512 / std::max(1, B * h))); }
extern "C" int64_t gt_galerkin_dkv_ln_ws_bytes(int32_t B, int32_t h, int32_t dk) {
    return (int64_t)B * dkv_ln_chunks(B, h) * 4 * h * dk * (int64_t)sizeof(float);
}

extern "C" int gt_galerkin_dkv_ln(const float* Kp, const float* Vp, const float* dM, const float* dQp, const float* qkv,
                                  const float* gamma, const float* stats, int32_t B, int32_t n, int32_t h, int32_t dk,
                                  int32_t p, float* d_qkv, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes,
                                  void* stream) {
    if (!qkv) return GT_EINVAL;
    return gt_galerkin_dkv_ln_plain(Kp, Vp, dM, dQp, qkv, gamma, nullptr, stats, B, n, h, dk, p, d_qkv, dgamma, dbeta, ws,
                                    ws_bytes, stream);
}

// beta != NULL: "plain" head tiles (gt_hip.h: hn_plain), qkv unused (may be NULL)
extern "C" int gt_galerkin_dkv_ln_plain(const float* Kp, const float* Vp, const float* dM, const float* dQp,
                                        const float* qkv, const float* gamma, const float* beta, const float* stats,
                                        int32_t B, int32_t n, int32_t h, int32_t dk, int32_t p, float* d_qkv,
                                        float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, void* stream) {
    if (!Kp || !Vp || !dM || (!qkv && !beta) || !gamma || !stats || !d_qkv || !dgamma || !dbeta) return GT_EINVAL;
    if (B <= 0 || n <= 0 || h <= 0 || dk <= 0 || p < 0) return GT_EINVAL;
    const int DP = round4(dk + p);
    if ((DP != 20 && DP != 36 && DP != 52) || (dk & 3)) return GT_ENOTSUP;
    if ((reinterpret_cast<uintptr_t>(Kp) | reinterpret_cast<uintptr_t>(Vp) | reinterpret_cast<uintptr_t>(dQp) |
         reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(d_qkv) | reinterpret_cast<uintptr_t>(stats) |
         reinterpret_cast<uintptr_t>(gamma)) & 15)                                      // dQp may be 0
        return GT_EALIGN;
    if (!ws || ws_bytes < gt_galerkin_dkv_ln_ws_bytes(B, h, dk)) return GT_EWS;
    hipStream_t st = (hipStream_t)stream;
    const int hd = h * dk;
    float* partial = reinterpret_cast<float*>(ws);
    DkvLnP q{Kp, Vp, dM, qkv, gamma, stats, d_qkv, partial, n, h, dk, p, B * n, beta};
    const int chunks = dkv_ln_chunks(B, h);
    dim3 grid((unsigned)(B * h), (unsigned)chunks);
    if (beta) {
        if (DP == 20) hipLaunchKernelGGL((galerkin_dkv_ln_kernel<1, true>), grid, dim3(256), 0, st, q);
        else if (DP == 36) hipLaunchKernelGGL((galerkin_dkv_ln_kernel<2, true>), grid, dim3(256), 0, st, q);
        else hipLaunchKernelGGL((galerkin_dkv_ln_kernel<3, true>), grid, dim3(256), 0, st, q);
    } else {
        if (DP == 20) hipLaunchKernelGGL((galerkin_dkv_ln_kernel<1, false>), grid, dim3(256), 0, st, q);
        else if (DP == 36) hipLaunchKernelGGL((galerkin_dkv_ln_kernel<2, false>), grid, dim3(256), 0, st, q);
        else hipLaunchKernelGGL((galerkin_dkv_ln_kernel<3, false>), grid, dim3(256), 0, st, q);
    }
    GT_LAUNCH_CHECK();
    if (dQp) {                                     // NULL: the caller's dQ product wrote the Q block itself
        const int64_t total4 = (int64_t)B * n * h * (dk >> 2);
        hipLaunchKernelGGL(headtile_unpad_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, dQp, d_qkv,
                           total4, h, dk, p, DP);
        GT_LAUNCH_CHECK();
    }
    int rc = gt_slab_reduce(partial, 4 * hd, B * chunks, 2 * hd, 1.f, dgamma, stream);
    if (rc) return rc;
    return gt_slab_reduce(partial + 2 * hd, 4 * hd, B * chunks, 2 * hd, 1.f, dbeta, stream);
}

extern "C" int gt_galerkin_finalize_fwd(const float* slabs, int32_t n_slabs, int64_t slab_stride, int32_t B,
                                        int32_t h, int32_t DP, int32_t Dr, int32_t d, int32_t n_tokens,
                                        const float* mask, const gt_dropout* drop, const float* Wfc,
                                        float* Mt, float* P, float* Pv, int32_t pos_dim, void* stream) {
    if (!slabs || !Wfc || !Mt || !P || n_slabs <= 0 || B <= 0 || h <= 0 || Dr <= 0 || DP < Dr || d <= 0 ||
        n_tokens <= 0 || pos_dim < 0 || pos_dim >= Dr)
        return GT_EINVAL;
    if (drop && drop->p > 0.f && !drop->seed) return GT_EINVAL;
    if (DP & 3) return GT_EINVAL;
    // (few (batch, head) pairs: four rows per block as before, for the parallelism of the slab sums)
    const int FIN_RB = B * h >= 256 ? fin_rows_per_block(DP) : 4;
    const size_t lds = ((size_t)FIN_RB * DP + (size_t)d * DP) * sizeof(float);
    if (int rc = allow_big_lds(galerkin_fin_fwd_kernel, lds)) return rc;
    hipLaunchKernelGGL(galerkin_fin_fwd_kernel, dim3(B * h, (DP + FIN_RB - 1) / FIN_RB), dim3(256), lds, (hipStream_t)stream, slabs,
                       n_slabs, slab_stride, h, DP, Dr, d, 1.f / (float)n_tokens, mask,
                       make_drop(mask ? nullptr : drop), Wfc, Mt, P, Pv, pos_dim, FIN_RB);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_galerkin_finalize_bwd(const float* dPt, const float* Mt, const float* mask,
                                        const gt_dropout* drop, const float* Wfc, int32_t B, int32_t h,
                                        int32_t DP, int32_t Dr, int32_t d, int32_t n_tokens, float* dM,
                                        float* dWfc_slabs, void* stream) {
    if (!dPt || !Mt || !Wfc || !dM || !dWfc_slabs || B <= 0 || h <= 0 || Dr <= 0 || DP < Dr || d <= 0 ||
        n_tokens <= 0)
        return GT_EINVAL;
    if (drop && drop->p > 0.f && !drop->seed) return GT_EINVAL;
    // parts: enough blocks to fill the chip (each part stages W_h and M in full, its slices of dP_h); wide models (d >= 160:
    // the full staging would leave one block per CU) always take four
    const int parts = std::max(d >= 160 ? 4 : 1, std::min(4, 1024 / (B * h)));
    const size_t lds = ((size_t)((DP + parts - 1) / parts) * (d + 1) + (parts > 1 ? (size_t)DP * ((d + parts - 1) / parts + 1) : 0) +
                        (size_t)d * DP + (size_t)DP * DP) * sizeof(float);
    if ((DP & 3) || (reinterpret_cast<uintptr_t>(dM) & 15)) return GT_EINVAL;
    if (int rc = allow_big_lds(galerkin_fin_bwd_kernel, lds)) return rc;
    hipLaunchKernelGGL(galerkin_fin_bwd_kernel, dim3(B * h, parts), dim3(256), lds, (hipStream_t)stream, dPt, Mt,
                       mask, make_drop(mask ? nullptr : drop), Wfc, h, DP, Dr, d, 1.f / (float)n_tokens, dM,
                       dWfc_slabs);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_layernorm_fwd(const float* x, const float* gamma, const float* beta, int32_t T, int32_t d,
                                float eps, float* y, float* stats, void* stream) {
    if (!x || !gamma || !beta || !y || !stats || T <= 0 || d <= 0) return GT_EINVAL;
    if (ln_narrow(x, y, gamma, d) && (reinterpret_cast<uintptr_t>(beta) & 15) == 0)
        hipLaunchKernelGGL(layernorm_fwd16_kernel, dim3(std::min(ceil_div(T, 16), 4096)), dim3(256), 0, (hipStream_t)stream, x,
                           gamma, beta, T, d, eps, y, stats);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ceil_div(T, 4)), dim3(256), 0, (hipStream_t)stream, x,
                           gamma, beta, T, d, eps, y, stats);
    GT_LAUNCH_CHECK();
    return 0;
}

constexpr int LN_MAXB = 512;
static inline int ln_blocks_bwd(int T) { return std::min(ceil_div(T, LN_ROWS), LN_MAXB); }
extern "C" int64_t gt_layernorm_bwd_ws_bytes(int32_t T, int32_t d) {
    return (int64_t)ln_blocks_bwd(T) * 2 * d * (int64_t)sizeof(float);
}

extern "C" int gt_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stats,
                                int32_t T, int32_t d, float* dx, float* dgamma, float* dbeta, void* ws,
                                int64_t ws_bytes, void* stream) {
    if (!dy || !x || !gamma || !stats || !dx || !dgamma || !dbeta || T <= 0 || d <= 0) return GT_EINVAL;
    if (!ws || ws_bytes < gt_layernorm_bwd_ws_bytes(T, d)) return GT_EWS;
    const int nblk = ln_blocks_bwd(T);
    const size_t lds = (size_t)8 * d * sizeof(float);
    if (lds > 64 * 1024) return GT_ENOTSUP;
    float* partial = reinterpret_cast<float*>(ws);
    if (ln_narrow(dy, x, dx, d) && (reinterpret_cast<uintptr_t>(gamma) & 15) == 0)
        hipLaunchKernelGGL(layernorm_bwd16_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, stats, T, d,
                           dx, partial);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream, dy, x, gamma,
                           stats, T, d, dx, partial);
    GT_LAUNCH_CHECK();
    int rc = gt_slab_reduce(partial, 2 * d, nblk, d, 1.f, dgamma, stream);
    if (rc) return rc;
    return gt_slab_reduce(partial + d, 2 * d, nblk, d, 1.f, dbeta, stream);
}

extern "C" int gt_modemix_fwd(const float* X, const float* W, int32_t B, int32_t Q, int32_t Cin, int32_t Cout,
                              int64_t x_bstride, int64_t y_bstride, int32_t q_total_x, int32_t q_total_y,
                              int32_t q_off, float* Y, void* stream) {
    if (!X || !W || !Y || B <= 0 || Q <= 0 || Cin <= 0 || Cout <= 0 || q_off < 0 ||
        q_off + Q > q_total_x || q_off + Q > q_total_y)
        return GT_EINVAL;
    if ((reinterpret_cast<uintptr_t>(W) & 7) != 0) return GT_EALIGN;
    const size_t lds = ((size_t)2 * Cin * Cout + (size_t)MM_BCH * 2 * Cin) * sizeof(float);
    if (int rc = allow_big_lds(modemix_fwd_kernel, lds)) return rc;
    hipLaunchKernelGGL(modemix_fwd_kernel, dim3(Q, modemix_slices(B, Q)), dim3(256), lds, (hipStream_t)stream, X, W, B,
                       Q, Cin, Cout, x_bstride, y_bstride, q_total_x, q_total_y, q_off, Y);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gt_modemix_bwd_ws_bytes(int32_t B, int32_t Q, int32_t Cin, int32_t Cout) {
    const int S = modemix_slices(B, Q);
    return S > 1 ? (int64_t)S * Cin * Cout * Q * 2 * (int64_t)sizeof(float) : 0;
}

extern "C" int gt_modemix_bwd(const float* X, const float* W, const float* dY, int32_t B, int32_t Q,
                              int32_t Cin, int32_t Cout, int64_t x_bstride, int64_t y_bstride,
                              int32_t q_total_x, int32_t q_total_y, int32_t q_off, float* dX, float* dW,
                              void* ws, int64_t ws_bytes, void* stream) {
    if (!X || !W || !dY || !dX || !dW || B <= 0 || Q <= 0 || Cin <= 0 || Cout <= 0 || q_off < 0 ||
        q_off + Q > q_total_x || q_off + Q > q_total_y)
        return GT_EINVAL;
    if (((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(dW)) & 7) != 0) return GT_EALIGN;
    if (Cin * Cout > 24 * 256) return GT_ENOTSUP;          // 96 x 48 (ex1 as shipped) = 18 pairs per thread
    const size_t lds =
        ((size_t)2 * Cin * (Cout + 1) + (size_t)MM_BCH * 2 * Cin + (size_t)MM_BCH * 2 * Cout) * sizeof(float);
    const int S = modemix_slices(B, Q);
    const int64_t nW = (int64_t)Cin * Cout * Q * 2;
    float* dWk = dW;                                       // S == 1: the kernel writes dW directly
    if (S > 1) {
        if (!ws || ws_bytes < gt_modemix_bwd_ws_bytes(B, Q, Cin, Cout)) return GT_EWS;
        if (reinterpret_cast<uintptr_t>(ws) & 7) return GT_EALIGN;
        dWk = reinterpret_cast<float*>(ws);
    }
    const dim3 grid((unsigned)Q, (unsigned)S);
    if (Cin * Cout <= 8 * 256) {
        if (int rc = allow_big_lds(modemix_bwd_kernel<8>, lds)) return rc;
        hipLaunchKernelGGL(modemix_bwd_kernel<8>, grid, dim3(256), lds, (hipStream_t)stream, X, W, dY, B, Q, Cin,
                           Cout, x_bstride, y_bstride, q_total_x, q_total_y, q_off, dX, dWk);
    } else {
        if (int rc = allow_big_lds(modemix_bwd_kernel<24>, lds)) return rc;
        hipLaunchKernelGGL(modemix_bwd_kernel<24>, grid, dim3(256), lds, (hipStream_t)stream, X, W, dY, B, Q, Cin,
                           Cout, x_bstride, y_bstride, q_total_x, q_total_y, q_off, dX, dWk);
    }
    GT_LAUNCH_CHECK();
    if (S > 1) return gt_slab_reduce(dWk, nW, S, nW, 1.f, dW, stream);      // fixed order: deterministic
    return 0;
}

extern "C" int32_t gt_galerkin_ktv_slabs(int32_t B, int32_t n) {
    // token chunks per sample: enough blocks to fill the chip (~4 per CU), at least 64 tokens per chunk
    int chunks = std::max(1, std::min(ceil_div(1024, std::max(B, 1)), ceil_div(n, 64)));
    return chunks;
}

extern "C" int gt_galerkin_ktv(const float* Kp, const float* Vp, int32_t B, int32_t n, int32_t h, int32_t dk,
                               int32_t p, float* slabs, int32_t n_slabs, void* stream) {
    return gt_galerkin_ktv_affine(Kp, Vp, nullptr, nullptr, B, n, h, dk, p, slabs, n_slabs, stream);
}

extern "C" int gt_galerkin_ktv_affine(const float* Kp, const float* Vp, const float* gamma, const float* beta, int32_t B,
                                      int32_t n, int32_t h, int32_t dk, int32_t p, float* slabs, int32_t n_slabs,
                                      void* stream) {
    if (!Kp || !Vp || !slabs || B <= 0 || n <= 0 || h <= 0 || dk <= 0 || p < 0 || n_slabs <= 0) return GT_EINVAL;
    if ((gamma == nullptr) != (beta == nullptr)) return GT_EINVAL;
    if ((dk & 15) || dk > 96 || p > 2) return GT_ENOTSUP;
    if (B > 65535) return GT_EINVAL;
    const int DP = round4(dk + p);
    const int chunk = ((ceil_div(n, n_slabs) + 3) / 4) * 4;
    if ((int64_t)chunk * n_slabs < n) return GT_EINVAL;
    dim3 grid((unsigned)n_slabs, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    // LDS-staged rows (16-byte loads of whole contiguous token tiles): one head per wave, 16 * h * DP floats per operand tile
    static const int lds_on = [] { const char* e = getenv("GT_KTV_LDS"); return e ? atoi(e) : 1; }();
    if (lds_on && h <= 4 && h * DP <= 256 && ((reinterpret_cast<uintptr_t>(Kp) | reinterpret_cast<uintptr_t>(Vp)) & 15) == 0) {
        const size_t lds = (size_t)2 * 2 * KTV_TT * h * DP * sizeof(float);
        switch (dk / 16) {
            case 1: hipLaunchKernelGGL(galerkin_ktv_lds_kernel<1>, grid, dim3(256), lds, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
            case 2: hipLaunchKernelGGL(galerkin_ktv_lds_kernel<2>, grid, dim3(256), lds, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
            case 3: hipLaunchKernelGGL(galerkin_ktv_lds_kernel<3>, grid, dim3(256), lds, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
            case 4: hipLaunchKernelGGL(galerkin_ktv_lds_kernel<4>, grid, dim3(256), lds, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
            case 6: hipLaunchKernelGGL(galerkin_ktv_lds_kernel<6>, grid, dim3(256), lds, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
            default: return GT_ENOTSUP;
        }
        GT_LAUNCH_CHECK();
        return 0;
    }
    switch (dk / 16) {
        case 1: hipLaunchKernelGGL(galerkin_ktv_kernel<1>, grid, dim3(256), 0, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
        case 2: hipLaunchKernelGGL(galerkin_ktv_kernel<2>, grid, dim3(256), 0, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
        case 3: hipLaunchKernelGGL(galerkin_ktv_kernel<3>, grid, dim3(256), 0, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
        case 4: hipLaunchKernelGGL(galerkin_ktv_kernel<4>, grid, dim3(256), 0, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
        case 6: hipLaunchKernelGGL(galerkin_ktv_kernel<6>, grid, dim3(256), 0, st, Kp, Vp, n, h, DP, p, chunk, slabs, B, gamma, beta); break;
        default: return GT_ENOTSUP;
    }
    GT_LAUNCH_CHECK();
    return 0;
}
