/*
 * gt_hip.h -- C ABI of libgt_hip.so: the MI355X (gfx950) device path for the Galerkin /
 * Fourier simple-attention encoder layer and the spectral-convolution decoder.
 *
 * Boundary contract
 *   - plain C: device pointers, sizes, POD descriptors; no torch / C++ types.
 *   - every entry point enqueues work on `stream` (a hipStream_t passed as void*) and
 *     returns immediately; no hidden synchronisation, no allocation (graph-capturable).
 *     Scratch memory is supplied by the caller (`ws`, `ws_bytes`).
 *   - all tensors are fp32, row-major, device-resident.
 *   - return value: 0 = success, >0 = hipError_t from the launch, <0 = GT_E* argument error.
 *
 * The reference (scaomath/galerkin-transformer) has no FFI of its own: its hot path is a chain
 * of ATen calls issued from Python.  Each entry point below names the reference call site(s)
 * (file:line under /root/reference) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 */
#ifndef GT_HIP_H
#define GT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GT_ABI_VERSION 21

/* argument errors */
#define GT_EINVAL   (-1)   /* bad shape / flag combination            */
#define GT_EALIGN   (-2)   /* pointer or leading dimension misaligned */
#define GT_EWS      (-3)   /* scratch buffer too small                */
#define GT_ENOTSUP  (-4)   /* combination not implemented             */

/* activations */
#define GT_ACT_NONE 0
#define GT_ACT_RELU 1
#define GT_ACT_SILU 2
#define GT_ACT_GELU 3   /* erf GELU; gt_dropact_fwd / gt_dropact_bwd only (every other entry point: GT_EINVAL) */
#define GT_ACT_DROP_SILU 4   /* gt_gemm only: the dropout sits in FRONT of the SiLU -- Conv2dResBlock's conv -> dropout -> activation
                                (layers.py:139-149) with a non-ReLU activation, where the two do not commute:
                                    u = keepscale(m, n) * v;   result = silu(u);   pre[m][n] (optional) = keepscale(m, n) * silu'(u)
                                `drop` is consumed here (not applied again behind the activation), and `pre` receives the factor the
                                backward multiplies the output gradient with (GT_AUX_MUL) instead of the pre-activation */

#define GT_ACT_SILU2 5       /* gt_gemm only (ABI v21): two SiLUs in a row -- Interp2dUpsample's conv block -> activation -> activation
                                (layers.py:642-650: Conv2dResBlock's own activation, then the up-scaler's) when its dropouts are off:
                                    result = silu(silu(v));   pre[m][n] (optional) = silu'(v) * silu'(silu(v))
                                `pre` is again the factor of the backward (GT_AUX_MUL on the product that forms the result's
                                gradient); `drop` must be NULL / p = 0 (GT_EINVAL) */

/* gt_gemm_desc.aux_op: multiply the result by a function of aux[m][n] */
#define GT_AUX_NONE      0
#define GT_AUX_GT0       1   /* v *= (aux > 0) ? aux_scale : 0      (ReLU' with the dropout scale folded in) */
#define GT_AUX_DSILU     2   /* v *= d/dx silu(aux)                 (aux = saved pre-activation)             */
#define GT_AUX_MUL       3   /* v *= aux * aux_scale                (explicit multiplicative mask replay)    */

int gt_abi_version(void);
/* Name of the code object's target ("gfx950").  Does not touch the device. */
const char* gt_target_arch(void);

/* ---------------------------------------------------------------------------------------------
 * Stateless dropout.  keep(idx) = hash(seed[0], salt, idx) >= p*2^32, scale 1/(1-p).  `seed`
 * is a DEVICE pointer to one uint64 so a captured graph replays with fresh masks after the
 * host bumps it (gt_seed_advance).  The same (seed, salt, idx) regenerates the mask in backward.
 * Replaces F.dropout / nn.Dropout at layers.py:701,731,981 and model.py:125,132.
 * ------------------------------------------------------------------------------------------- */
typedef struct gt_dropout {
    float p;                    /* 0 => disabled                                   */
    uint32_t salt;              /* distinguishes call sites                        */
    const uint64_t* seed;       /* device pointer, may be NULL iff p == 0          */
} gt_dropout;

/* seed[0] += inc  (one tiny kernel; capturable) */
int gt_seed_advance(uint64_t* seed, uint64_t inc, void* stream);
/* out[i] = x[i] * keepscale(i)   -- standalone elementwise dropout, n elements, used by tests */
int gt_dropout_apply(const float* x, float* out, int64_t n, const gt_dropout* d, void* stream);

/* y[i] = act2(drop2(act1(drop1(x[i]))))  in one elementwise pass (mask index = i, the index gt_dropout_apply uses;
 * d1 / d2 may be NULL or p = 0, act* = GT_ACT_*): the tail of Conv2dResBlock (layers.py:88-150) and of
 * Interp2dUpsample's conv branch (layers.py:658-668).  Backward recomputes from x:  gx[i] = gy[i] * dy/dx. */
int gt_dropact_fwd(const float* x, float* y, int64_t n, const gt_dropout* d1, int32_t act1, const gt_dropout* d2,
                   int32_t act2, void* stream);
int gt_dropact_bwd(const float* x, const float* gy, float* gx, int64_t n, const gt_dropout* d1, int32_t act1,
                   const gt_dropout* d2, int32_t act2, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batched fp32 GEMM on the MFMA pipe (v_mfma_f32_16x16x4_f32) with fused prologue/epilogue.
 *
 *   for z = (b0, b1) in [0,batch0) x [0,batch1):
 *     acc[m][n] = sum_k  A_z(m,k) * keepA(m,k) * B_z(k,n)
 *     v   = alpha*acc + bias[n] + sum_{j<rp} rp_a[m][j]*rp_b[n][j] + add_z[m][n]
 *     pre[m][n] = v                                  (optional)
 *     v   = act(v) ;  v *= f(aux[m][n]) ;  v = dropout(v)          (act = GT_ACT_DROP_SILU: see there)
 *     C_z[m][n] = res_z[m][n] + out_scale*v          (res optional)
 *
 *   layout_a = 0 : A(m,k) = A[m*lda + k]     (k contiguous; activations [tokens, features])
 *   layout_a = 1 : A(m,k) = A[k*lda + m]     (m contiguous; transposed use, reduction over rows)
 *   layout_b = 0 : B(k,n) = B[n*ldb + k]     (nn.Linear weight [out, in])
 *   layout_b = 1 : B(k,n) = B[k*ldb + n]
 *
 * Replaces every nn.Linear / torch.matmul / einsum contraction on the path:
 *   layers.py:837-839 (QKV), :723,:733 (K^T V, Q M), :687,:703 (Q K^T, S V), :897 (fc),
 *   :979-987 (FFN), :1087-1098 / :1172-1189 (truncated DFT stages), model.py:615-629, and the
 *   autograd backward of each (addmm / bmm backward = the transposed layouts).
 *
 * split_k: 0 = let the library decide, 1 = never split, >1 = that many K-slices (reduced in a
 * second deterministic pass; only alpha is allowed as epilogue then).  ws/ws_bytes: scratch for
 * the slabs (gt_gemm_ws_bytes gives an upper bound).
 * ------------------------------------------------------------------------------------------- */
typedef struct gt_gemm_desc {
    int32_t M, N, K;
    int32_t layout_a, layout_b;
    int32_t batch0, batch1;
    int32_t split_k;

    const float* A; int64_t lda, a_bs0, a_bs1;
    const float* B; int64_t ldb, b_bs0, b_bs1;
    float*       C; int64_t ldc, c_bs0, c_bs1;

    /* prologue: stateless dropout mask on A.  The mask index of A's element (outer, inner) is
       z*a_drop_bstride + outer*a_drop_ld + inner  (outer = the lda-strided index).          */
    gt_dropout a_drop; float a_drop_sign; int64_t a_drop_ld, a_drop_bstride;
    /* optional by-product (layout_a = 1 only): a_colsum[m] = sum over batch and k of A_z(m,k)*keepA(m,k),
       i.e. the column sums of the [K, M] tensor behind A -- the bias gradient that goes with a weight
       gradient (autograd of nn.Linear: layers.py:811,823,964,976).  Deterministic (per-slice partials in
       `ws`, reduced in a fixed order). */
    float* a_colsum;

    /* epilogue */
    float alpha;
    const float* bias;
    int32_t rp; const float* rp_a; int64_t rp_lda, rp_a_bs0; const float* rp_b; int64_t rp_ldb;
    const float* add; int64_t ldadd, add_bs0, add_bs1;   /* full-matrix addend before the activation */
    float* pre; int64_t ldpre;                  /* batch strides = c_bs0/c_bs1 scaled by ldpre/ldc is NOT assumed: dense [batch][M][ldpre] */
    int32_t act;
    int32_t aux_op; const float* aux; int64_t ldaux, aux_bs0, aux_bs1; float aux_scale;
    gt_dropout drop;                            /* mask index = (z*M + m)*N + n */
    const float* res; int64_t ldr, r_bs0, r_bs1;
    float out_scale;

    /* Fused two-layer pointwise head  y = W2 act(W1 x + b1) + b2  with a narrow second layer (n_out <= 4):
     * SpectralRegressor / PointwiseRegressor tail, model.py:575-580, 625-629.  Needs N <= 128 (one tile
     * column), no batching, no split-K; of the epilogue fields above only alpha, bias and act are honoured.
     *   GT_EP_ROWDOT  : out2[m][o] = sum_n act(alpha*acc + bias)[m][n] * w2[o*ldw2 + n] + b2[o];  C is NOT
     *                   written -- the [M, N] hidden activation never reaches HBM.
     *   GT_EP_MLP_BWD : h = alpha*acc + bias is the recomputed pre-activation;
     *                   C[m][n] = (sum_o g2[m][o] * w2[o*ldw2 + n]) * act'(h[m][n])       (= dL/dh)
     *                   dw2[o][n] = sum_m g2[m][o] * act(h[m][n])                          (by-product)   */
    int32_t ep_mode, n_out;
    const float* w2; int64_t ldw2;
    const float* b2;        /* ROWDOT: [n_out] or NULL */
    float* out2;            /* ROWDOT: [M, n_out] */
    const float* g2;        /* MLP_BWD: [M, n_out] */
    float* dw2;             /* MLP_BWD: [n_out, N] (row stride N) */

    /* Optional second product accumulated into the same tile before the epilogue:
     *     acc += sum_{k < K2} A2_z(m,k) * B2_z(k,n)          (same layout_a / layout_b as the first product)
     * SpectralConv's  act(irfft-stage(Z) + Linear(x))  (layers.py:1172-1189, 1087-1098) becomes one launch:
     * first product = c2r DFT stage, second = the residual Linear on the block's own input rows; likewise
     * its backward  dx = rfft-stage^T(dX1) + dpre W.  No split-K, no A dropout, v1 kernel only. */
    int32_t K2;
    const float* A2; int64_t lda2, a2_bs0, a2_bs1;
    const float* B2; int64_t ldb2, b2_bs0, b2_bs1;

    /* GT_EP_HEADNORM: the packed QKV projection (layers.py:838-840) with the per-head LayerNorm + position columns of
     * gt_headnorm_fwd (layers.py:841-874) fused behind it.  C [M, N = 3*h*dk] is written as usual (the backward reads
     * the raw projection) and in the same pass every row's head segments go to hn_out [3][M][h][DP] (normalised where
     * hn_norm_mask says so, coordinates in columns [0, hn_p), zero pad) and hn_stats [#normed][M][h][2] = (mean,
     * rstd).  Runs on the split-operand ring kernel only (precision != GT_PREC_F32, 16-byte aligned operands,
     * K % 4 == 0): dk in {16, 32, 64}, hn_p <= 4, layout_a = layout_b = 0, no batching, no split-K, no other epilogue field but
     * alpha and bias; anything else returns GT_ENOTSUP and the caller runs gt_gemm + gt_headnorm_fwd. */
    const float* hn_gamma; const float* hn_beta; const float* hn_pos;
    float* hn_out; float* hn_stats;
    int32_t hn_h, hn_dk, hn_p, hn_norm_mask;
    float hn_eps;

    /* Arithmetic of the contraction (GT_PREC_*).  Operands, accumulator and result are fp32 in every mode; what
     * changes is the MFMA instruction the products run on:
     *   GT_PREC_F32    v_mfma_f32_16x16x4_f32, bit-for-bit an fp32 FMA chain (157 TFLOP/s peak);
     *   GT_PREC_BF16X3 every operand value is split exactly into three bf16 terms while it is staged
     *                  (a = h0 + h1 + h2 to 2^-24) and the six plane products down to 2^-16 are accumulated on
     *                  v_mfma_f32_32x32x16_bf16 in fp32: same rounding class as GT_PREC_F32 (the 1e-5 parity gate
     *                  holds), 16/6 of its matrix rate;
     *   GT_PREC_BF16X2 two terms / three products (~2^-16 relative);
     *   GT_PREC_BF16   operands rounded to bf16, one product: throughput mode with its own (3e-3) gate;
     *   GT_PREC_F16X2  every operand value is split into TWO fp16 terms (11 + 11 significand bits) after scaling by a
     *                  power of two that tracks the data (weights: per 32-column tile when they are packed; activation rows:
     *                  a running per-row exponent inside the kernel, the accumulator rescaled when it has to drop), three
     *                  products on v_mfma_f32_32x32x16_f16: fp32-class results at half the matrix work of GT_PREC_BF16X3.
     *                  Implemented by the packed-B kernels (a weight against >= 16384 token rows, the implicit
     *                  convolutions); every other launch of this mode runs GT_PREC_BF16X3.
     * The split kernel serves the plain-epilogue products with M, N >= 96 (whole 128 x 128 tiles); everything else
     * (fused heads, head-norm epilogue, narrow or tiny problems, the tall-skinny path) stays on the fp32 pipe in
     * every mode. */
    int32_t precision;

    /* Implicit 3x3 convolution (stride 1, zero padding 1) on channels-last activations: cv_c > 0 makes A a
     * [B, cv_h, cv_w, cv_c] image, M = B*cv_h*cv_w its pixels, K = 9*cv_c, and the contraction index
     *     k = (c / CB) * 9*CB + tap * CB + (c % CB),      CB = 32 if cv_c % 32 == 0 else 16,  tap = 3*(dy+1) + (dx+1)
     * reads A[pixel + (dy, dx)][c] (zero outside the image) -- the im2col matrix is never written; the nine taps of a
     * channel block are adjacent in k so that they hit the same cache lines back to back.  B is the filter as
     * [N][9*cv_c] in that k order (layout_b = 0): an nn.Conv2d weight [N][c][3][3] -> [N][c/CB][tap][CB]
     * for the forward product (layers.py:98-100 `nn.Conv2d(.., kernel_size=3, padding=1, bias=False)` of the scaler
     * blocks), or the tap-reversed, in/out-swapped filter for the data gradient.  Split-operand ring kernel only:
     * precision != GT_PREC_F32, layout_a = layout_b = 0, cv_c % 16 == 0, no batching, no split-K, no A dropout,
     * M, N >= 96 (N >= 32 with M >= 16384: narrow outputs run on 128 x 64 tiles of the packed-B kernel); anything else
     * returns GT_ENOTSUP (the caller then uses its library convolution).  lda > cv_c is the pixel pitch of the image:
     * the convolution then reads the cv_c-channel column slice A .. A + cv_c of a wider channels-last buffer in place
     * (lda % 4 == 0); the weight-gradient form takes ldb the same way.
     *
     * cv_wgrad != 0 selects the weight gradient of the same convolution instead:
     *     C_tap[m][n] = sum_pixels A[pixel][m] * X[pixel + (dy, dx)][n]          tap = 0..8 = the batch index
     * with A = the output gradient [pixels, M] (layout_a = 1, lda = M's row length), B = X the channels-last input
     * [B, cv_h, cv_w, cv_c] (layout_b = 1, N = cv_c), K = the pixel count, batch0 = 9, batch1 = 1, C = [9][M][N]
     * (c_bs0 = the tap stride), split_k = 0 (the library cuts K into chunks; the nine taps of a chunk run next to each
     * other on one XCD and share its L2).  cv_w >= 16, no epilogue fields. */
    int32_t cv_h, cv_w, cv_c, cv_wgrad;

    /* GT_EP_HEADNORM: streams (bit 0 Q, 1 K, 2 V) whose RAW projection is NOT written to C.  The backward needs the raw
     * rows of the normalised streams only (LayerNorm backward); a training forward passes ~hn_norm_mask & 7 and saves
     * a third of C's write traffic.  Zero (the default) writes all of C. */
    int32_t hn_skip_raw_mask;

    /* GT_EP_HEADNORM, hn_plain != 0: the head tiles of the normalised streams hold the normalised values WITHOUT the
     * LayerNorm affine (xh = (x - mean) * rstd); gamma and beta are applied by the consumers (gt_galerkin_ktv_affine
     * on load, gt_galerkin_dkv_ln_plain folded into dM), whose LayerNorm backward then needs no raw projection at all:
     * with hn_skip_raw_mask = 7 the launch writes nothing to C and C may be NULL. */
    int32_t hn_plain;

    /* Pre-packed B (round 5, ABI v19): NULL, or the buffer gt_gemm_pack_b_many filled for exactly this product (same B,
     * layout_b, ldb, N, K, precision = GT_PREC_F16X2) since B last changed -- gt_gemm then skips its own pack launch and
     * needs no workspace.  Only where gt_gemm_packed_b_bytes(d) > 0. */
    const void* b_packed;

    /* Second output (ABI v20, round 6): c_masked[m][n] = C[m][n] * keepscale_{c_mask}(m, n)  -- the finished result once more
     * under a stateless dropout mask (index (z*M + m)*N + n, the index gt_dropout_apply uses on the dense [M, N] tensor), row
     * pitch ldc_masked.  The data gradient a layer hands to the layer in front of it is needed twice there: as it is (residual
     * branch) and under that layer's output-dropout mask (nn.Dropout of model.py:125,132 backwards: it feeds three
     * contractions) -- the product that computes it writes both and the elementwise gt_dropout_apply pass between two layers'
     * backward passes disappears.  No batching, no split-K, GT_EP_NORMAL only; NULL = off. */
    float* c_masked; int64_t ldc_masked;
    gt_dropout c_mask;
} gt_gemm_desc;

#define GT_PREC_F32    0
#define GT_PREC_BF16X3 1
#define GT_PREC_BF16X2 2
#define GT_PREC_BF16   3
#define GT_PREC_F16X2  4

#define GT_EP_NORMAL  0
#define GT_EP_ROWDOT  1
#define GT_EP_MLP_BWD 2
#define GT_EP_HEADNORM 3

void    gt_gemm_desc_init(gt_gemm_desc* d);            /* zero + alpha=1, out_scale=1, batch=1 */
int64_t gt_gemm_ws_bytes(const gt_gemm_desc* d);
int     gt_gemm(const gt_gemm_desc* d, void* ws, int64_t ws_bytes, void* stream);
/* Weights packed ahead, all in one launch (the model-API counterpart of "one pack launch per step": every nn.Linear weight of
 * layers.py:811,823,964,976 feeds 2 token products per step).  gt_gemm_packed_b_bytes: size of d's packed weight if gt_gemm(d)
 * is a single packed-B launch, else 0.  gt_gemm_pack_b_many: descs[i].B -> outs[i] for n <= 64 products (GT_PREC_F16X2). */
int64_t gt_gemm_packed_b_bytes(const gt_gemm_desc* d);
int     gt_gemm_pack_b_many(const gt_gemm_desc* descs, void* const* outs, int32_t n, void* stream);
/* debugging/tests: which tile configuration and split the library picks */
int     gt_gemm_plan(const gt_gemm_desc* d, int32_t* bm, int32_t* bn, int32_t* split);
/* symbol of the kernel instance gt_gemm would launch for d, as a profiler prints it */
int     gt_gemm_kernel_name(const gt_gemm_desc* d, char* buf, int32_t n);

/* ---------------------------------------------------------------------------------------------
 * Fused FeedForward forward (ABI v20, round 6; layers.py:979-987 with the residual of model.py:131-132) in GT_PREC_F16X2:
 *     hid[t][:] = dropout_h( act( x[t] W1^T + b1 ) )             [T, f]  written to HBM (the backward reads it)
 *     out[t][:] = res[t] + dropout_o( hid[t] W2^T + b2 )         [T, d]  (res may be NULL)
 * in ONE launch: a block owns 64 token rows and keeps their hidden tile in LDS between the two contractions, so the hidden
 * activation is never read back in the forward.  Value for value the arithmetic of the two gt_gemm launches it replaces
 * (same packed planes, same running exponents, same product order, same dropout indices t*f + j / t*d + c): bit-identical
 * results.  x [T, d], W1 [f, d], W2 [d, f] dense fp32, 16-byte aligned; act = GT_ACT_RELU | GT_ACT_NONE.
 * w1_packed / w2_packed: both NULL (the call packs the weights into ws, gt_ffn_fwd_ws_bytes) or the buffers
 * gt_gemm_pack_b_many filled for the products [T, d] x W1^T and [T, f] x W2^T (gt_gemm_desc.b_packed of those).
 * Implemented for d = 128, f = 256, T >= 16384 (else GT_ENOTSUP: two gt_gemm launches do the same).
 * ------------------------------------------------------------------------------------------- */
int64_t gt_ffn_fwd_ws_bytes(int64_t T, int32_t d, int32_t f);
int gt_ffn_fwd(const float* x, int64_t T, int32_t d, int32_t f, const float* W1, const float* b1, const float* W2,
               const float* b2, const float* res, const gt_dropout* drop_h, const gt_dropout* drop_o, int32_t act,
               float* hid, float* out, void* relu_bits, const void* w1_packed, const void* w2_packed, void* ws,
               int64_t ws_bytes, void* stream);
/* relu_bits (optional, gt_ffn_bits_bytes(T) bytes, 16-byte aligned): one bit per hidden value -- kept by the dropout AND
 * positive -- in the kernel's own register layout (opaque to the caller), for gt_ffn_bwd.
 *
 * The data half of the backward (autograd of the above, nn.Linear / ReLU / nn.Dropout backwards) in the same geometry:
 *     gh[t][:] = (gm[t] W2) .* bits .* hid_scale          [T, f]  written to HBM (dW1 = gh^T x needs it)
 *     dx[t][:] = res[t] + gh[t] W1                        [T, d]  (res = the unmasked incoming gradient of a residual layer, or NULL)
 *     dx_masked = dx .* keepscale_{mask2}                 optional second output (gt_gemm_desc.c_masked's twin)
 * gm = the incoming gradient under the output-dropout mask; hid_scale = 1 / (1 - p_h).  The hidden activation itself is not
 * read (its 242 MB at B = 128 were re-read as a 1-bit decision by the unfused hidden-gradient launch).  w2_packed / w1_packed:
 * the packs of the products [T, d] x W2 (layout_b = 1, N = f) and [T, f] x W1 (layout_b = 1, N = d), or both NULL + ws
 * (gt_ffn_fwd_ws_bytes).  Same shapes as gt_ffn_fwd, else GT_ENOTSUP. */
int64_t gt_ffn_bits_bytes(int64_t T);
int gt_ffn_bwd(const float* gm, int64_t T, int32_t d, int32_t f, const float* W2, const float* W1, const void* relu_bits,
               float hid_scale, const float* res, float* gh, float* dx, float* dx_masked, const gt_dropout* mask2,
               const void* w2_packed, const void* w1_packed, void* ws, int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * out[n] (+)= sum_m A[m*lda + n] * keepA(m,n)  -- bias gradients.  Two deterministic passes.
 * ------------------------------------------------------------------------------------------- */
int gt_colsum(const float* A, int64_t lda, int32_t M, int32_t N, const gt_dropout* a_drop,
              float a_sign, float* out, void* ws, int64_t ws_bytes, void* stream);
/* dpre[i] = dout[i] * act'(pre[i])   (GT_ACT_*), n elements */
int gt_act_bwd(const float* dout, const float* pre, float* dpre, int64_t n, int32_t act, void* stream);
/* out[i] = alpha * sum_s slabs[s*stride + i], i < n */
int gt_slab_reduce(const float* slabs, int64_t stride, int32_t n_slabs, int64_t n, float alpha,
                   float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-head LayerNorm + position concat (layers.py:841-874).
 *   qkv   [T, 3*d]   raw projections (T = B*n tokens), d = h*dk
 *   pos   [T, p]     coordinates (p may be 0 -> pos may be NULL)
 *   gamma/beta [2][h][dk]: affine of the two normalised streams, in stream order
 *   norm_mask: bit s set => stream s (0=Q,1=K,2=V) is normalised (galerkin: K,V = 6; fourier:
 *              Q,K = 3; no attn norm: 0)
 *   out   [3][T][h][DP]  with DP = round_up(dk+p, 4); columns [0,p) = pos, [p,p+dk) = values,
 *                        rest zero.  stats [2][T][h][2] = (mean, rstd) of the normalised streams.
 * Backward: d_out -> d_qkv [T,3d], dgamma/dbeta [2][h][dk] (deterministic two-pass).
 * ------------------------------------------------------------------------------------------- */
int gt_headnorm_fwd(const float* qkv, const float* pos, const float* gamma, const float* beta,
                    int32_t T, int32_t h, int32_t dk, int32_t p, int32_t norm_mask, float eps,
                    float* out, float* stats, void* stream);
int gt_headnorm_bwd(const float* d_out, const float* qkv, const float* gamma, const float* stats,
                    int32_t T, int32_t h, int32_t dk, int32_t p, int32_t norm_mask,
                    float* d_qkv, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes,
                    void* stream);
int64_t gt_headnorm_bwd_ws_bytes(int32_t T, int32_t h, int32_t dk);

/* ---------------------------------------------------------------------------------------------
 * Galerkin attention core, small-matrix stage (layers.py:723-733 + :897 folded):
 *   Mt[b,h]   = mask .* ( sum_s slabs[s][b,h] ) / n          (DPxDP, the "attn_weight")
 *   P[b][h*DP + j][c] = sum_e Mt[b,h][j][e] * Wfc[c][h*Dr + e]    (Dr = dk+p unpadded)
 * so that   attn_out[b] = [Q'](n x h*DP) @ P[b]  is one GEMM.  mask: explicit tensor
 * [B,h,DP,DP] of multipliers (replay mode), or NULL with drop.p (0.5 in the reference) for the
 * stateless RNG, or both NULL/0 for identity.
 * Backward: from dPt[b][c][h*DP+j] (= d attn_out^T Q') produce
 *   dM[b,h]  = mask .* (dP_h Wfc_h) / n        and      dWfc slabs [B][d][h*Dr].
 * ------------------------------------------------------------------------------------------- */
/* K'^T V' (layers.py:723) as a streaming kernel on the head-tile layout [B*n][h][DP] (what
 * gt_headnorm_fwd writes): token rows go from HBM straight into MFMA operand registers, the dk x dk core
 * accumulates on the matrix pipe and the p-wide coordinate borders on the VALU.  Writes n_slabs partial
 * [B,h,DP,DP] slabs (token chunks) that gt_galerkin_finalize_fwd sums.  dk % 16 == 0, dk <= 96, p <= 2
 * (else GT_ENOTSUP: use gt_gemm).  gt_galerkin_ktv_slabs suggests n_slabs. */
int32_t gt_galerkin_ktv_slabs(int32_t B, int32_t n);
int gt_galerkin_ktv(const float* Kp, const float* Vp, int32_t B, int32_t n, int32_t h, int32_t dk, int32_t p,
                    float* slabs, int32_t n_slabs, void* stream);
/* The same for "plain" head tiles (gt_gemm_desc.hn_plain): K' = gamma_K xh + beta_K, V' likewise, formed while the
 * operands are loaded; gamma, beta [2][h][dk] (K then V).  NULL gamma / beta = gt_galerkin_ktv. */
int gt_galerkin_ktv_affine(const float* Kp, const float* Vp, const float* gamma, const float* beta, int32_t B, int32_t n,
                           int32_t h, int32_t dk, int32_t p, float* slabs, int32_t n_slabs, void* stream);
int gt_galerkin_finalize_fwd(const float* slabs, int32_t n_slabs, int64_t slab_stride,
                             int32_t B, int32_t h, int32_t DP, int32_t Dr, int32_t d, int32_t n_tokens,
                             const float* mask, const gt_dropout* drop, const float* Wfc,
                             float* Mt, float* P, float* Pv, int32_t pos_dim, void* stream);
/* (Pv, optional: the value rows pos_dim .. Dr-1 of every head's block of P once more, compact [B][h (Dr - pos_dim)][d] --
 * the B operand of the backward's dQ product.) */
int gt_galerkin_finalize_bwd(const float* dPt, const float* Mt, const float* mask,
                             const gt_dropout* drop, const float* Wfc,
                             int32_t B, int32_t h, int32_t DP, int32_t Dr, int32_t d, int32_t n_tokens,
                             float* dM, float* dWfc_slabs, void* stream);

/* dK'[t] = V'[t] dM^T,  dV'[t] = K'[t] dM  for every token of every (batch, head): the backward of
 * M = K'^T V' (layers.py:723) as one streaming pass over the head tiles [B*n][h][DP] (dM [B,h,DP,DP]).
 * DP in {20, 36, 52}, else GT_ENOTSUP (two batched gt_gemm launches do the same). */
int gt_galerkin_dkv(const float* Kp, const float* Vp, const float* dM, float* dKp, float* dVp, int32_t B,
                    int32_t n, int32_t h, int32_t DP, void* stream);

/* The same two products with the per-head LayerNorm backward of gt_headnorm_bwd (norm_mask = K and V, layers.py:841-874
 * backwards) applied on the way out: dK', dV' are never written.  Writes all three blocks of d_qkv [B*n][3 h dk] (K, V:
 * LayerNorm backward of the products; Q: dQp [B*n][h][DP] with its coordinate / pad columns dropped -- or, with
 * dQp = NULL, left to the caller, whose dQ product can write the value columns straight into the Q block) and
 * dgamma / dbeta [2][h][dk] (K then V).  qkv = the raw projection, gamma [2][h][dk], stats [2][B*n][h][2] as
 * gt_headnorm_fwd left them.  DP = round4(dk + p) in {20, 36, 52} and dk % 4 == 0, else GT_ENOTSUP (gt_galerkin_dkv +
 * gt_headnorm_bwd do the same in two passes). */
int64_t gt_galerkin_dkv_ln_ws_bytes(int32_t B, int32_t h, int32_t dk);
int gt_galerkin_dkv_ln(const float* Kp, const float* Vp, const float* dM, const float* dQp, const float* qkv,
                       const float* gamma, const float* stats, int32_t B, int32_t n, int32_t h, int32_t dk, int32_t p,
                       float* d_qkv, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, void* stream);
/* The same on "plain" head tiles: beta [2][h][dk] given, qkv unused (may be NULL); gamma moves onto the rows of dM, beta dM
 * is added to the products, the tiles themselves are the xh of the LayerNorm backward. */
int gt_galerkin_dkv_ln_plain(const float* Kp, const float* Vp, const float* dM, const float* dQp, const float* qkv,
                             const float* gamma, const float* beta, const float* stats, int32_t B, int32_t n, int32_t h,
                             int32_t dk, int32_t p, float* d_qkv, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused Fourier-type attention (layers.py:672-705):  out = ((Q' K'^T) * scale .* mask) V'  on the head-tile
 * layout [B*n][h][DP], without writing the n x n score matrix: score tiles are scaled, masked (stateless
 * dropout with mask index ((b*h+head)*n + query)*n + key -- the index the materialising gt_gemm path uses --
 * or an explicit [B,h,n,n] mask) and consumed in registers.  One entry point, three uses:
 *     forward        F1 = Q',  F2 = NULL, T1 = K', T2 = V',  owner_is_key = 0  ->  O1 = out
 *     d/dQ'          F1 = dO,  F2 = NULL, T1 = V', T2 = K',  owner_is_key = 0  ->  O1 = dQ'
 *     d/dV', d/dK'   F1 = K',  F2 = V',   T1 = Q', T2 = dO,  owner_is_key = 1  ->  O1 = dV', O2 = dK'
 * DP in {20, 36, 52} (else GT_ENOTSUP: materialise through gt_gemm).
 * ------------------------------------------------------------------------------------------- */
int gt_fourier_attn(const float* F1, const float* F2, const float* T1, const float* T2, float* O1, float* O2,
                    int32_t B, int32_t n, int32_t h, int32_t DP, float scale, const float* mask,
                    const gt_dropout* drop, int32_t owner_is_key, void* stream);

/* The same operator (layers.py:672-705, the three uses above) in the two-term fp16 arithmetic (GT_PREC_F16X2: three products
 * per contraction on v_mfma_f32_16x16x32_f16, fp32 accumulation, fp32-class results) -- ABI v19.  The head tiles are split
 * ONCE per use into "images": per (batch, head, tile of 32 token rows) the two fp16 planes in MFMA fragment order, in the
 * layout of each of the two products, with the tile's power-of-two exponent and its largest row norm in a header.
 *     gt_fourier16_image_bytes   size of ONE tensor's image block (header + images) for head tiles [B*n][h][DP]
 *     gt_fourier16_presplit      X0..X3 (up to four head-tile tensors, NULL-terminated) -> image blocks I0..I3, one launch
 *     gt_fourier16_attn          as gt_fourier_attn with F1, F2, T1, T2 = image blocks of the tensors named there
 * DP in {20, 36, 52}, else GT_ENOTSUP. */
int64_t gt_fourier16_image_bytes(int32_t B, int32_t n, int32_t h, int32_t DP);
int gt_fourier16_presplit(const float* X0, const float* X1, const float* X2, const float* X3, void* I0, void* I1, void* I2,
                          void* I3, int32_t B, int32_t n, int32_t h, int32_t DP, void* stream);
int gt_fourier16_attn(const void* F1, const void* F2, const void* T1, const void* T2, float* O1, float* O2, int32_t B,
                      int32_t n, int32_t h, int32_t DP, float scale, const float* mask, const gt_dropout* drop,
                      int32_t block16, int32_t owner_is_key, void* stream);
/* block16 != 0 (p must be 0.5 -- the reference's always-on F.dropout(p_attn), layers.py:700-701): the attention-score mask
 * is drawn per 4 x 4 block of the [n x n] matrix of each (batch, head): ONE hash per block,
 *     x = fmix32( (((b h + head) nq4 + (query >> 2)) nq4 + (key >> 2)) * 0x9e3779b1 + key(seed, salt) ),  nq4 = ceil(n / 4),
 *     keep(query, key) = bit 16 + 4 (query & 3) + (key & 3) of x,
 * instead of one hash per element (block16 = 0: the index documented at gt_fourier_attn).  gt_dropout_block16 applies the
 * same mask (and the 1 / (1 - p) rescale) in place to a materialised score matrix S [BH][n][n]: the path that returns the
 * attention weights runs it behind its score gt_gemm, so both paths draw one mask. */
int gt_dropout_block16(float* S, int64_t BH, int32_t n, const gt_dropout* drop, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Truncated-DFT stages along the contiguous grid axis of SpectralConv2d (layers.py:1176 rfft2 and :1187
 * irfft2 restricted to the kept modes; the residual nn.Linear of :1128 / :1196 rides on the synthesis).
 * One batch item = one grid line; all tensors dense fp32:
 *     analysis    Y[b] (P x C)  = F^T X[b]                 F [n, P],  X [nb, n, C],  Y [nb, P, C]
 *     synthesis   Y[b] (n x Co) = act( F Z[b] + X2[b] W2 + bias ),   pre (optional) = the argument of act
 *                 F [n, P], Z [nb, P, Co], X2 [nb, n, C2], W2 [C2, Co], bias [Co] or NULL, Y / pre [nb, n, Co]
 * The backward of one is the other with the transposed basis (spectral.py).  Implemented for C = Co = C2 =
 * 32 channels, P = 2*modes <= 32 (synthesis: P % 4 == 0), n <= 224; anything else returns GT_ENOTSUP and the
 * caller runs the same product through gt_gemm.
 * ------------------------------------------------------------------------------------------- */
int gt_dft_analysis(const float* F, const float* X, float* Y, int32_t nb, int32_t n, int32_t P, int32_t C,
                    void* stream);
int gt_dft_synthesis(const float* F, const float* Z, float* Y, int32_t nb, int32_t n, int32_t P, int32_t Co,
                     const float* X2, const float* W2, int32_t C2, const float* bias, int32_t act, float* pre,
                     void* stream);
/* ABI v21: the same with  Y *= silu'(out_gate)  on the store (out_gate [nb, n, Co]; act must be GT_ACT_NONE, pre NULL).  In the
 * backward of a SpectralConv2d stack (model.py:569-572 backwards) the synthesis of layer l + 1 forms the gradient of layer l's
 * activated output: with out_gate = layer l's saved pre-activation it hands over the gradient of the pre-activation, and
 * layer l's own gt_act_bwd pass disappears. */
int gt_dft_synthesis_gated(const float* F, const float* Z, float* Y, int32_t nb, int32_t n, int32_t P, int32_t Co,
                           const float* X2, const float* W2, int32_t C2, const float* bias, int32_t act, float* pre,
                           const float* out_gate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused pointwise regression head  out[t] = w2 . act(W1 x[t] + b1) + b2  (model.py:575-580, 625-629: the
 * Linear(K -> N), activation, Linear(N -> n_out) tail of SpectralRegressor at every fine-grid point) and its
 * complete backward in one pass over x: the [T, N] hidden activation and dL/dh never reach HBM.
 *     X [T, K] dense, W1 [N, K], b1 [N] or NULL, w2 [n_out, N], b2 [n_out] or NULL, out [T, n_out]
 *     backward: g [T, n_out] = dL/dout  ->  dX [T, K] (or NULL), dW1 [N, K], db1 [N] / dw2 [n_out, N] /
 *     db2 [n_out] (each may be NULL); ws >= gt_mlp_head_bwd_ws_bytes(T) bytes of scratch (deterministic
 *     fixed-order reduction).  Implemented for K = 32, N = 128, n_out = 1 (else GT_ENOTSUP: use gt_gemm's
 *     ep_mode GT_EP_ROWDOT / GT_EP_MLP_BWD path, which covers N <= 128, n_out <= 4).
 *     precision: GT_PREC_F16X2 = two fp16 terms per operand, three products on the fp16 matrix pipe (per-row / per-tensor
 *     / running power-of-two exponents, gt_head.hip); any other GT_PREC_* = the fp32-MFMA kernels (exact fp32 products).
 * ------------------------------------------------------------------------------------------- */
int gt_mlp_head_fwd(const float* X, int64_t T, int32_t K, int32_t N, int32_t n_out, const float* W1,
                    const float* b1, const float* w2, const float* b2, int32_t act, int32_t precision, float* out,
                    void* stream);
int64_t gt_mlp_head_bwd_ws_bytes(int64_t T);
int gt_mlp_head_bwd(const float* X, int64_t T, int32_t K, int32_t N, int32_t n_out, const float* W1,
                    const float* b1, const float* w2, int32_t act, int32_t precision, const float* g, float* dX,
                    float* dW1, float* db1, float* dw2, float* db2, void* ws, int64_t ws_bytes, void* stream);
/* ABI v21: dX *= silu'(dx_gate) on the store (dx_gate [T, K] = the pre-activation of the layer whose SiLU produced X, e.g.
 * the last SpectralConv2d in front of the head, model.py:572-575); NULL = gt_mlp_head_bwd. */
int gt_mlp_head_bwd_gated(const float* X, int64_t T, int32_t K, int32_t N, int32_t n_out, const float* W1,
                          const float* b1, const float* w2, int32_t act, int32_t precision, const float* g, float* dX,
                          const float* dx_gate, float* dW1, float* db1, float* dw2, float* db2, void* ws, int64_t ws_bytes,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row LayerNorm over the feature axis (model.py:128-129,134-135 when layer_norm=True).
 * ------------------------------------------------------------------------------------------- */
int gt_layernorm_fwd(const float* x, const float* gamma, const float* beta, int32_t T, int32_t d,
                     float eps, float* y, float* stats /* [T][2] mean,rstd */, void* stream);
int gt_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stats,
                     int32_t T, int32_t d, float* dx, float* dgamma, float* dbeta,
                     void* ws, int64_t ws_bytes, void* stream);
int64_t gt_layernorm_bwd_ws_bytes(int32_t T, int32_t d);

/* ---------------------------------------------------------------------------------------------
 * Spectral mode mixing (layers.py:1066-1075, 1143-1151): per retained mode q, complex
 *   Y[b][ri][q][o] = sum_i X[b][ri'][q][i] (x) W[i][o][q]      (complex product, real-pair weights)
 * X, Y layout: [B][2][Q][C] (re plane, im plane), Q = number of retained modes, C channels.
 * W layout:    [Cin][Cout][Q][2]  (the reference's parameter layout; for 2-D the two corner
 *              blocks are passed as two calls or as Q = modes*modes each via `w_qstride`).
 * Backward: dX from dY and W ;  dW from X and dY (summed over the batch).
 * The batch is cut into slices so that Q x slices blocks fill the chip; the backward's dW partials (one per slice) go
 * through ws (gt_modemix_bwd_ws_bytes, 0 when one slice suffices) and are summed in a fixed order.
 * ------------------------------------------------------------------------------------------- */
int gt_modemix_fwd(const float* X, const float* W, int32_t B, int32_t Q, int32_t Cin, int32_t Cout,
                   int64_t x_bstride, int64_t y_bstride, int32_t q_total_x, int32_t q_total_y,
                   int32_t q_off, float* Y, void* stream);
int64_t gt_modemix_bwd_ws_bytes(int32_t B, int32_t Q, int32_t Cin, int32_t Cout);
int gt_modemix_bwd(const float* X, const float* W, const float* dY, int32_t B, int32_t Q,
                   int32_t Cin, int32_t Cout, int64_t x_bstride, int64_t y_bstride,
                   int32_t q_total_x, int32_t q_total_y, int32_t q_off,
                   float* dX, float* dW, void* ws, int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bilinear resize, align_corners=True (F.interpolate at layers.py:483-512, 658-670), with the
 * channels-first <-> channels-last change of the scaler boundaries (model.py:675-687, 740-749)
 * fused in.  x: [B,C,Hi,Wi] (in_nhwc=0) or [B,Hi,Wi,C] (in_nhwc=1); y likewise with Ho,Wo and
 * out_nhwc.  act: GT_ACT_NONE or GT_ACT_RELU applied to the resized output.  NHWC sides need
 * C % 4 == 0 and 16-byte aligned pointers.
 * Backward: g, y_saved (the activated forward output; only read when act == GT_ACT_RELU) have the
 * forward OUTPUT shape/layout, dx the forward INPUT shape/layout.  Gather formulation: no atomics,
 * bitwise deterministic.
 * ------------------------------------------------------------------------------------------- */
int gt_bilinear2d_fwd(const float* x, float* y, int32_t B, int32_t C, int32_t Hi, int32_t Wi,
                      int32_t Ho, int32_t Wo, int32_t in_nhwc, int32_t out_nhwc, int32_t act,
                      void* stream);
/* Same, plus an affine term per output pixel q and channel c (NHWC in and out only):
 *     y = act( resize(x) + bias[c] + sum_j rp_a[q*rp_lda + j] * rp_b[c*rp_ldb + j] )
 * This is what makes  fc(cat[upsample(x), grid])  (model.py:740-749 followed by model.py:615-617)
 * computable as  upsample(x W_x^T) + grid W_g^T + b : the pointwise Linear commutes with the bilinear
 * interpolation (its weights sum to one), so the Linear runs at the coarse resolution and only the
 * freq_dim-channel result is interpolated. */
typedef struct gt_resize_affine {
    const float* bias;                 /* [C] or NULL */
    int32_t rp;                        /* 0..8 */
    const float* rp_a; int64_t rp_lda; /* [B*Ho*Wo, rp] */
    const float* rp_b; int64_t rp_ldb; /* [C, rp] */
} gt_resize_affine;
int gt_bilinear2d_fwd_affine(const float* x, float* y, int32_t B, int32_t C, int32_t Hi, int32_t Wi,
                             int32_t Ho, int32_t Wo, int32_t in_nhwc, int32_t out_nhwc, int32_t act,
                             const gt_resize_affine* aff, void* stream);
int gt_bilinear2d_bwd(const float* g, const float* y_saved, float* dx, int32_t B, int32_t C,
                      int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t in_nhwc,
                      int32_t out_nhwc, int32_t act, void* stream);
/* Channels-last resize whose INPUT is the padded three-segment buffer the down-scaler's convolution chain writes
 * (Interp2dEncoder, layers.py:497-512: cat[x1, x2, x3] -> F.interpolate -> activation): x [B, Hi, Wi, 3*segp] holds the
 * C real channels in three column segments of segp channels each -- real channel c at column c + (segp - seg) * min(c / seg, 2),
 * i.e. segment widths seg, seg, C - 2*seg, zeros behind them -- and y [B, Ho, Wo, C] is dense.  The concatenation is never
 * materialised.  Backward: g, y_saved dense like y; dx in the padded layout (padding columns get zero); x_gate (optional)
 * = the forward input x itself when it is the output of a ReLU: dx is zeroed where x <= 0, i.e. the gradient leaves
 * already multiplied by the derivative of the ReLU that produced x (one elementwise pass less in its producer).
 * ABI v20 -- act = GT_ACT_SILU (Interp2dEncoder's default activation_type, layers.py:446-456; ex3's down-scaler): the forward
 * also writes dact [B, Ho, Wo, C] = silu'(resized value) and the backward takes THAT buffer as y_saved (its gradient is
 * g * dact); gate_mul != 0 makes x_gate a multiplicative factor (dx *= x_gate: the keepscale * silu' buffer a
 * GT_ACT_DROP_SILU product left in `pre`) instead of the ReLU test. */
int gt_bilinear2d_seg_fwd(const float* x, float* y, int32_t B, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                          int32_t Wo, int32_t act, int32_t seg, int32_t segp, float* dact, void* stream);
int gt_bilinear2d_seg_bwd(const float* g, const float* y_saved, float* dx, int32_t B, int32_t C, int32_t Hi,
                          int32_t Wi, int32_t Ho, int32_t Wo, int32_t act, int32_t seg, int32_t segp,
                          const float* x_gate, int32_t gate_mul, void* stream);

/* ---------------------------------------------------------------------------------------------
 * First stage of the CNN down-scaler in one pass (layers.py:483-495 with Conv2dResBlock :88-150):
 *     y = relu( resize( relu( dropout( conv3x3(x; w), p ) ), (Ho, Wo) ) )
 * x [B,Cin,H,W] (Cin <= 4), w [Cout,Cin,3,3] (padding 1, stride 1, no bias), y [B,Cout,Ho,Wo], all
 * channels-first.  The Cout-channel fine-resolution map is never materialised.  The dropout mask is
 * indexed by the linear NCHW index of the (virtual) conv output, i.e. identical to running
 * gt_dropout_apply on it.  act: GT_ACT_RELU, or (ABI v20) GT_ACT_SILU = the same pass with SiLU on both sides of the resize
 * (Interp2dEncoder's default activation_type='silu', layers.py:446-456: ex3's down-scaler); the SiLU backward re-evaluates
 * the convolution at the four source pixels (no decision bits: relu_bits must be NULL) and does not read y.
 * Backward produces the weight gradient only (dw [Cout,Cin,3,3]; two-pass deterministic reduction
 * through ws); callers that need d/dx use the unfused operators.
 * ------------------------------------------------------------------------------------------- */
int gt_conv3x3_resize_fwd(const float* x, const float* w, float* y, int32_t B, int32_t Cin, int32_t Cout,
                          int32_t H, int32_t W, int32_t Ho, int32_t Wo, const gt_dropout* drop, int32_t act,
                          void* stream);
int gt_conv3x3_resize_bwd(const float* g, const float* y, const float* x, const float* w, int32_t B,
                          int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                          const gt_dropout* drop, int32_t act, float* dw, void* ws, int64_t ws_bytes,
                          void* stream);
int64_t gt_conv3x3_resize_bwd_ws_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W);
/* The same with y (and g, y of the backward) channels-last [B, Ho, Wo, Cout]: what the channels-last convolution chain
 * of the down-scaler consumes. */
int gt_conv3x3_resize_fwd_nhwc(const float* x, const float* w, float* y, int32_t B, int32_t Cin, int32_t Cout,
                               int32_t H, int32_t W, int32_t Ho, int32_t Wo, const gt_dropout* drop, int32_t act,
                               void* relu_bits, void* stream);
int gt_conv3x3_resize_bwd_nhwc(const float* g, const float* y, const float* x, const float* w, int32_t B,
                               int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                               const gt_dropout* drop, int32_t act, const void* relu_bits, float* dw, void* ws,
                               int64_t ws_bytes, void* stream);
/* relu_bits (optional, NULL = off; Cout % 16 == 0, 8-byte aligned, gt_conv3x3_resize_bits_bytes() bytes): the forward
 * records 4 bits per (output pixel, channel) -- source pixel t of the bilinear stencil kept by the dropout AND positive,
 * all four cleared when the resized value is <= 0 -- and the backward, handed the same buffer, takes its decisions from
 * there: it neither re-evaluates the convolution at the four source pixels nor re-draws the mask, and does not read y
 * (y may be NULL then).  Same weight gradient bit for bit decisions, half the instructions, 0.45 instead of 0.81 GB read. */
int64_t gt_conv3x3_resize_bits_bytes(int32_t B, int32_t Cout, int32_t Ho, int32_t Wo);

/* ---------------------------------------------------------------------------------------------
 * Weight gradient of a NARROW channels-last 3x3 convolution (padding 1, stride 1, no bias): the three convolutions of the
 * down-scaler's chain, reference libs/layers.py:463-482 (Interp2dEncoder conv1 / conv2 / conv3 = Conv2dResBlock,
 * layers.py:88-150), whose weight gradients autograd would compute with cudnn/MIOpen's conv2d weight backward:
 *     dw[co][ci][ky][kx] = alpha * sum_{b,y,x} gy[(b H + y) W + x][co] * x[(b H + y + ky - 1) W + x + kx - 1][ci]
 * gy / x: channels-last images with pixel pitches ldg >= Cout / ldx >= Cin (a column segment of a wider buffer is read in
 * place), 16-byte aligned, pitches multiples of 4.  dw: [Cout][Cin][3][3], the reference's layout.  precision: GT_PREC_BF16X3
 * (three bf16 planes, six products) or GT_PREC_F16X2 (two fp16 planes under one running power-of-two scale per operand and
 * block, three products); each operand value is split once per block (gt_convw.hip).
 * GT_ENOTSUP unless (Cout == 48, Cin % 16 == 0: the down-scaler's padded narrow convolutions) or (Cout % 64 == 0,
 * Cin % 32 == 0: the up-scaler's 128 -> 128 convolution).  Rows wider than 80 pixels are worked in equal x-segments of at
 * most 80 (round 5: the 113 / 114-pixel rows of the 211 x 211 configuration).  Deterministic: partial results per (image, row
 * chunk, segment) in ws
 * (>= gt_conv3x3_wgrad_nhwc_ws_bytes), summed in a fixed order.
 * ------------------------------------------------------------------------------------------- */
int gt_conv3x3_wgrad_nhwc(const float* gy, int64_t ldg, const float* x, int64_t ldx, float* dw, int32_t B, int32_t H,
                          int32_t W, int32_t Cin, int32_t Cout, float alpha, int32_t precision, void* ws, int64_t ws_bytes,
                          void* stream);
int64_t gt_conv3x3_wgrad_nhwc_ws_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout);

/* Parity hook for the fused convolution's ReLU (reference: Conv2dResBlock's activation, layers.py:139-149, which the fused
 * pass never materialises): with a device buffer `mask` [B, Cout, H, W] of bytes registered, every following
 * gt_conv3x3_resize_fwd(_nhwc) records the decision it takes for each fine-grid value it evaluates (1: kept by the dropout
 * and positive, 0: not; values no output pixel touches keep what the caller stored).  NULL switches it off.  The pointer is
 * a process-wide device variable: tests only. */
int gt_debug_conv0_mask(void* mask, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer step on ONE flat fp32 bucket: what utils_ft.py:676-681 does per batch
 * (nn.utils.clip_grad_norm_(model.parameters(), grad_clip); optimizer.step() with torch.optim.Adam), as three
 * launches for the whole model.  Norm, step count and learning rate are read from DEVICE memory (no host sync; one
 * captured launch serves every iteration).
 *   gt_grad_sqnorm    out[0] = sum_i (scale * g[i])^2            (deterministic two-pass; ws >= gt_grad_sqnorm_ws_bytes())
 *   gt_adam_clip_step g' = gscale * clip * g + weight_decay * p,  clip = min(1, max_norm / (sqrt(sqnorm[0]) + 1e-6))
 *                     (max_norm <= 0: no clipping, sqnorm may be NULL);
 *                     m = beta1 m + (1-beta1) g';  v = beta2 v + (1-beta2) g'^2;  t = step[0] (already incremented:
 *                     bump it with gt_seed_advance before the call);
 *                     p -= lr[0] / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)      -- torch.optim.Adam.
 *   beta1_dev (optional device scalar) overrides beta1: OneCycleLR cycles Adam's beta1 together with the rate.
 *   gscale folds the 1/world of the gradient average after a sum-all-reduce (pass the same factor as `scale` to
 *   gt_grad_sqnorm so that the norm is the norm of the averaged gradient).  All pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int64_t gt_grad_sqnorm_ws_bytes(void);
int gt_grad_sqnorm(const float* g, int64_t n, float scale, float* out, void* ws, int64_t ws_bytes, void* stream);
int gt_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sqnorm, float gscale,
                      float max_norm, const float* lr, float beta1, float beta2, float eps, float weight_decay,
                      const uint64_t* step, const float* beta1_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GT_HIP_H */
