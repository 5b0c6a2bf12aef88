/* liblgm_hip.so — C-ABI of the MI355X (gfx950) hot path for lightning-generative-models.
 *
 * The reference (pure Python, /root/reference) has no FFI: every entry point below replaces
 * a group of ATen calls that the reference's nn.Modules issue on the hot path.  Each entry
 * cites the reference lines whose arithmetic it implements.  SURVEY.md §8(b) defines the
 * conventions:
 *   - plain pointers + sizes, no torch types; all device pointers are caller-owned;
 *   - the library never allocates, frees, retains or synchronises; all work is enqueued on
 *     the `stream` argument (a hipStream_t passed as void*);
 *   - return 0 = OK, <0 = invalid argument / unsupported shape, >0 = hipError_t;
 *     lgm_last_error() returns a thread-local description;
 *   - re-entrant per THREAD (safe from the autograd thread next to the main thread): an entry point keeps no state
 *     between calls.  What the library does hold, exactly:
 *       . thread-local: the last error string, the last kernel name (lgm_last_error / lgm_last_kernel), and - only
 *         INSIDE lgm_conv_bwd_pair[_post] - the pair-recording context in which that call's own input-gradient and
 *         weight-gradient dispatchers record instead of launching; it is opened and closed by the same call on the same
 *         thread and never visible between calls;
 *       . process-wide, written once: hipFuncSetAttribute one-shot flags per kernel (idempotent: a race sets the same
 *         attribute twice) and `static const` switches read from the environment on first use;
 *       . process-wide, diagnostics only: the cycle-stamp buffers of lgm_wino_set_debug_buffer /
 *         lgm_wino4_set_debug_buffer (NULL on the product path; a tool sets and clears them around its own launches,
 *         single-threaded).
 *     Callers that launch from several threads at once need no lock.
 *
 * Activation layout: NHWC fp32, addressed as (pointer, pitch) where pitch is the distance
 * in floats between consecutive pixels (>= channels; lets a tensor be a channel-slice of a
 * wider concat buffer).  Pointers and pitches of tensors read through the implicit-GEMM
 * convolutions must be 16-byte aligned / multiples of 4 floats.
 *
 * Convolution weights: physical layout [Nw][KH*KW][Cw] (= torch channels_last memory format
 * of the logical OIHW Conv2d weight; for ConvTranspose2d Nw = in_channels, Cw = out_channels).
 */
#ifndef LGM_HIP_H
#define LGM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGM_ABI_VERSION 7   /* 2: lgm_posemb takes a frequency table, BatchNorm entry points, *_planes / *_partial;
                             * 3: lgm_conv3x3_wino4*, lgm_gn_fwd_stats, lgm_conv3x3_wino_wgradn* + LgmWgradItem, a negative
                             *    dst offset in lgm_wino_weights table rows means "skip that copy", lgm_kernel_name*;
                             * 4: LgmPostOp carries the BatchNorm-backward sums (bn_*), lgm_bn_reduce3_coef_tiles;
                             * 5: lgm_set_cu_margin / lgm_cu_margin;
                             * 6: lgm_extract_axpby, lgm_model_predictions (GaussianDiffusion's per-sample-time algebra), lgm_time_mlp_*, lgm_weng_*;
                             * 7: lgm_gn_bwd_add, lgm_wgrad1x1_group*, lgm_wgrad_queue_* */
#define LGM_OK 0
#define LGM_ERR_INVALID (-1)
#define LGM_ERR_UNSUPPORTED (-2)

int lgm_abi_version(void);
const char* lgm_last_error(void);
/* diagnostic: name (as rocprofv3 prints it, without the argument list) of the primary kernel launched by the calling
 * thread's last convolution-family call (lgm_conv_xy / _yx / _wgrad / lgm_conv3x3_wino) */
const char* lgm_last_kernel(void);
/* diagnostic: every name lgm_last_kernel() can ever return (one entry per site in the library that notes a name, in link
 * order; duplicates possible).  Each is a prefix of the demangled name of a kernel in liblgm_hip.so - checked on the CPU by
 * tests/test_cabi.py, so that bench.py's per-kernel attribution and the rocprofv3 rows under profiles/ cannot drift apart. */
int lgm_kernel_name_count(void);
const char* lgm_kernel_name(int i);

/* ---------------------------------------------------------------------------------------
 * Convolution family (implicit GEMM on v_mfma_f32_32x32x2_f32, exact fp32).
 * Geometry is always that of the equivalent *forward convolution* X -> Y:
 *   X side: [B, H, W, Cw]   Y side: [B, Ho, Wo, Nw]   weight [Nw][KH*KW][Cw]
 *   Y[b,oh,ow,n] = sum_{kh,kw,c} X[b, oh*stride-pad+kh, ow*stride-pad+kw, c] * Wt[n][kh*KW+kw][c]
 * Replaces: nn.Conv2d / nn.ConvTranspose2d forward+backward at
 *   ddpm.py:96,103,160,187,213,215,252,253,304,377,413,422 (UNet),
 *   dcgan.py:79-87,150-158 (DCGAN G/D), vqvae.py:36-51,74-85, residual.py:14-18.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  int32_t B, H, W, Cw;    /* X side */
  int32_t Ho, Wo, Nw;     /* Y side */
  int32_t KH, KW, stride, pad;
} LgmConvGeom;

/* X -> Y  (Conv2d.forward; ConvTranspose2d input-gradient).
 * y = conv(x, w) + bias[n] + res   (bias, res optional = NULL; res may alias y). */
int lgm_conv_xy(const LgmConvGeom* g, const float* x, int64_t x_pitch, const float* w,
                const float* bias, const float* res, int64_t res_pitch,
                float* y, int64_t y_pitch, void* workspace, int64_t workspace_bytes, void* stream);

/* Y -> X  (Conv2d input-gradient; ConvTranspose2d.forward).
 * x = conv_transpose(y, w) + bias[c] + res   (bias, res optional; res may alias x). */
int lgm_conv_yx(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* w,
                const float* w_t, const float* bias, const float* res, int64_t res_pitch,
                float* x, int64_t x_pitch, void* workspace, int64_t workspace_bytes, void* stream);
/* w_t (optional): the same weights transposed to [Cw][KH*KW][Nw]; when given, the 3x3 input-gradient
 * kernel reads its weight fragments with the same 16-byte loads as the forward pass.
 * lgm_transpose_weights refreshes such copies for a whole flat parameter buffer in ONE launch:
 * table rows = (offset, Nw, T, Cw, first_block); dst uses the same offsets as src. */
int lgm_transpose_weights(const float* src, float* dst, const int32_t* table, int n_layers,
                          int total_blocks, void* stream);

/* Optional workspace (bytes) for lgm_conv_xy (yx = 0) / lgm_conv_yx (yx = 1): small-image 3x3
 * layers split the reduction over workgroups and combine the partials in a fixed order.
 * Passing NULL / too few bytes is legal (no split, slower on tiny feature maps). */
int64_t lgm_conv_workspace(const LgmConvGeom* g, int yx);

/* Post-op of a convolution's epilogue: out = act(conv + bias + res), then - backward passes - the ReLU / LeakyReLU
 * mask of a SAVED activation m (its forward output): out *= (m > 0 ? 1 : mask_slope).  Replaces the separate
 * activation launches around the reference's Conv2d -> ReLU pairs (vqvae.py:36-51,74-85, residual.py:14-20) and
 * their autograd mirror images.  act: 0 none, 3 ReLU, 4 LeakyReLU(slope); mask NULL: none.  Applied inside the
 * implicit-GEMM kernels' epilogue / split-K reducer; paths without an epilogue hook finish with one elementwise
 * launch - the result is the same on every path. */
typedef struct {
  int32_t act;
  float slope;
  const float* mask;
  int64_t mask_pitch;
  float mask_slope;
  /* ABI 4.  BatchNorm's backward sums from THIS epilogue (reference: autograd's backward of nn.BatchNorm2d behind a
   * convolution's input gradient, dcgan.py:150-161, wgan.py:117-156): when the finished output `out` is the gradient gn
   * that arrives at a train-mode BatchNorm whose input was `bn_a` (same [rows][channels] shape as `out`, statistics
   * bn_mean / bn_rstd), every full row tile also leaves (sum gn, sum gn * xhat, 0) per channel in
   * bn_partial[tile][3][N] - the layout lgm_bn_reduce3's first stage writes - and *bn_tiles receives the number of
   * tiles; lgm_bn_reduce3_coef_tiles finishes from there and the separate read pass over (gn, a) disappears.
   * *bn_tiles == 0: this launch could not (split-K, ragged tiles, a path without the hook): run lgm_bn_reduce3_coef.
   * bn_a NULL: off. */
  const float* bn_a;
  int64_t bn_a_pitch;
  const float* bn_mean;
  const float* bn_rstd;
  float* bn_partial;
  int64_t bn_partial_floats;     /* capacity of bn_partial */
  int32_t* bn_tiles;             /* out (host memory) */
} LgmPostOp;
int lgm_conv_xy_post(const LgmConvGeom* g, const float* x, int64_t x_pitch, const float* w, const float* bias,
                     const float* res, int64_t res_pitch, float* y, int64_t y_pitch, void* workspace,
                     int64_t workspace_bytes, const LgmPostOp* post, void* stream);
int lgm_conv_yx_post(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* w, const float* w_t,
                     const float* bias, const float* res, int64_t res_pitch, float* x, int64_t x_pitch,
                     void* workspace, int64_t workspace_bytes, const LgmPostOp* post, void* stream);
/* lgm_conv_bwd_pair with lgm_conv_yx_post's post-op on the INPUT gradient (the backward mask of an activation that sat in
 * front of this layer's input - VQ-VAE residual.py:14-20, vqvae.py:36-51): gx = post(W^T gy + res).  Keeps masked layers
 * on the one-launch path. */
int lgm_conv_bwd_pair_post(const LgmConvGeom* g, const float* gy, int64_t gy_pitch, const float* x, int64_t x_pitch,
                           const float* w, const float* w_t, const float* res, int64_t res_pitch, float* gx,
                           int64_t gx_pitch, void* dgrad_ws, int64_t dgrad_ws_bytes, float* gw, float* gbias,
                           float beta, void* wgrad_ws, int64_t wgrad_ws_bytes, int64_t* desc, const LgmPostOp* post,
                           void* stream);


/* Weight gradient: gw[n][tap][c] = beta*gw + sum_{b,oh,ow} Y[b,oh,ow,n] * X[b,ih,iw,c].
 * (Conv2d: Y = grad_output, X = input;  ConvTranspose2d: Y = input, X = grad_output.)
 * Deterministic: split-K partials go to `workspace` and are reduced in a fixed order.
 * lgm_conv_wgrad_workspace() returns the bytes needed for the given geometry. */
int64_t lgm_conv_wgrad_workspace(const LgmConvGeom* g);
/* gbias (optional, [Nw]): fused bias gradient gbias[n] = beta*gbias + sum_{b,oh,ow} Y[b,oh,ow,n]
 * (valid for Conv2d / Linear, where Y is the output gradient). */
int lgm_conv_wgrad(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* x,
                   int64_t x_pitch, float* gw, float* gbias, float beta, void* workspace,
                   int64_t workspace_bytes, void* stream);
/* Backward of ANY convolution / linear layer in one call: weight / bias gradient (lgm_conv_wgrad, or its deferred form
 * when desc != NULL) AND input gradient gx = W^T gy (+ res) (lgm_conv_yx).  When the dispatchers pick the two kernels
 * that can share a grid - the 1x1 convolutions and linears of the UNet at small row counts (to_qkv / to_out
 * ddpm.py:214-223, res_conv :184, Downsample :100-104) - both run in ONE launch (gemm_bwd_pair_kernel: same kernel
 * bodies, bit-identical results); otherwise exactly the two separate calls.  dgrad_ws (lgm_conv_workspace(g, 1) bytes)
 * and wgrad_ws (lgm_conv_wgrad_workspace(g) bytes) must not overlap. */
int lgm_conv_bwd_pair(const LgmConvGeom* g, const float* gy, int64_t gy_pitch, const float* x, int64_t x_pitch,
                      const float* w, const float* w_t, const float* res, int64_t res_pitch, float* gx,
                      int64_t gx_pitch, void* dgrad_ws, int64_t dgrad_ws_bytes, float* gw, float* gbias, float beta,
                      void* wgrad_ws, int64_t wgrad_ws_bytes, int64_t* desc, void* stream);
/* Deferred form (same arithmetic): writes only the per-split partial slabs into `workspace` (which must then
 * stay untouched until the reduction) and fills desc[8] = {workspace, slab stride, gw, n_w, gbias, n_b, splits,
 * beta bits}; lgm_wgrad_reduce_batch then performs the fixed-order slab reductions of MANY layers in one
 * launch (table rows = desc + first block, blocks per row = ceil((n_w + n_b) / 256); rows with splits == 1
 * must be left out: their result is already in gw).  Replaces ~75 tiny reduce launches per DDPM step. */
int lgm_conv_wgrad_deferred(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* x,
                            int64_t x_pitch, float* gw, float* gbias, float beta, void* workspace,
                            int64_t workspace_bytes, int64_t* desc, void* stream);
int lgm_wgrad_reduce_batch(const int64_t* table, int n_entries, int64_t total_blocks, void* stream);

/* Column sums of a [rows, cols] matrix with row pitch: out[c] = beta*out[c] + sum_r a[r,c].
 * Used for conv/linear bias gradients.  Deterministic two-stage; workspace >=
 * lgm_colsum_workspace(rows, cols) bytes. */
int64_t lgm_colsum_workspace(int64_t rows, int64_t cols);
int lgm_colsum(const float* a, int64_t pitch, int64_t rows, int64_t cols, float* out, float beta,
               void* workspace, void* stream);
/* Deferred form: the first stage's partial rows stay in `workspace` (must then stay untouched until the reduction) and
 * desc[8] receives their row for lgm_wgrad_reduce_batch - the bias gradients of transposed convolutions join the
 * bucket's one reduction launch.  cols % 4 == 0 and a 16-byte aligned `out` required. */
int lgm_colsum_deferred(const float* a, int64_t pitch, int64_t rows, int64_t cols, float* out, float beta,
                        void* workspace, int64_t* desc, void* stream);

/* ---------------------------------------------------------------------------------------
 * GroupNorm(+FiLM scale/shift)(+SiLU)(+residual)   — Block.forward ddpm.py:164-173
 *   z = GN(x; gamma, beta, G, eps) * (scale + 1) + shift ;  y = act ? silu(z) : z ;  y += res
 * x, y, res: [B, HW, C] NHWC with pitches.  ss (optional): [B, >=2C] rows, scale = ss[b, c],
 * shift = ss[b, C + c] (the chunk(2) of the time-MLP output, ddpm.py:192-194).
 * Outputs saved for backward: mean, rstd [B, G]; coefA, coefB [B, C]  (z = x*coefA + coefB).
 * ------------------------------------------------------------------------------------- */
int lgm_gn_fwd(const float* x, int64_t x_pitch, int B, int HW, int C, int G, float eps,
               const float* gamma, const float* beta, const float* ss, int64_t ss_pitch, int act,
               const float* res, int64_t res_pitch, float* y, int64_t y_pitch, float* mean,
               float* rstd, float* coefA, float* coefB, void* stream);
/* The same, with x still in pieces: the convolution that produces x (Block.proj, ddpm.py:160-171) split its reduction
 * and left `splits` partial planes [B*HW][C] (dense, `plane_stride` floats apart) instead of running its reducer
 * (lgm_conv3x3_wino_partial).  The planes are summed here in the reducer's fixed order (plane 0, 1, ..., then
 * conv_bias), the finished x is WRITTEN to `x` (the backward pass reads it) and normalised in the same pass: one
 * launch and one round trip of x less per convolution.  Only shapes with lgm_gn_planes_supported(...) == 1. */
int64_t lgm_gn_planes_supported(int B, int HW, int C, int G);
/* 1 when lgm_gn_fwd runs its one-pass kernel for this shape (the slices fit a block's registers), 0: two passes over x */
int64_t lgm_gn_fwd_fused_supported(int B, int HW, int C, int G);
int lgm_gn_fwd_planes(const float* planes, int64_t plane_stride, int splits, const float* conv_bias, float* x,
                      int64_t x_pitch, int B, int HW, int C, int G, float eps, const float* gamma,
                      const float* beta, const float* ss, int64_t ss_pitch, int act, const float* res,
                      int64_t res_pitch, float* y, int64_t y_pitch, float* mean, float* rstd, float* coefA,
                      float* coefB, void* stream);
/* The same, with the STATISTICS already gathered by the convolution that produced x (lgm_conv3x3_wino4_stats; Block.proj
 * -> Block.norm, ddpm.py:157-173, on maps whose (image, channel block) slices do not fit a one-pass GroupNorm block - the
 * 64 x 64 maps of configs/diffusion/ddpm_64.json): `stats` holds, per image, `parts_per_image` rows of (sum, sum of
 * squares) per channel of the convolution's outputs BEFORE its bias (`conv_bias`, may be NULL), layout
 * [b * parts + q][2][C].  A small kernel combines them in float64 in a fixed order into mean / rstd / coefA / coefB, then
 * the apply pass reads x once.  C / G must divide 64. */
int lgm_gn_fwd_stats(const float* stats, int parts_per_image, const float* conv_bias, const float* x, int64_t x_pitch,
                     int B, int HW, int C, int G, float eps, const float* gamma, const float* beta, const float* ss,
                     int64_t ss_pitch, int act, const float* res, int64_t res_pitch, float* y, int64_t y_pitch,
                     float* mean, float* rstd, float* coefA, float* coefB, void* stream);
/* Backward.  gx (optionally accumulated), ggamma/gbeta (= affine_beta*old + new), gss [B, >=2C]
 * (optional; = gss_beta*old + new).  workspace: 5*B*C floats. */
int lgm_gn_bwd(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch, int B, int HW,
               int C, int G, const float* gamma, const float* beta, const float* ss,
               int64_t ss_pitch, int act, const float* mean, const float* rstd, const float* coefA,
               const float* coefB, float* gx, int64_t gx_pitch, int accumulate_gx, float* ggamma,
               float* gbeta, float affine_beta, float* gss, int64_t gss_pitch, float gss_beta,
               float* workspace, void* stream);
/* Backward with gy given as the partial planes of the input-gradient convolution that produces it (ResnetBlock:
 * block2.proj's input gradient feeds only block1.norm's backward, ddpm.py:176-200); gy itself is never written.
 * rows / desc: both NULL (gamma / beta gradients complete on return) or the deferred form of lgm_gn_bwd_deferred. */
int lgm_gn_bwd_planes(const float* x, int64_t x_pitch, const float* gy_planes, int64_t plane_stride, int splits, int B,
                      int HW, int C, int G, const float* gamma, const float* beta, const float* ss,
                      int64_t ss_pitch, int act, const float* mean, const float* rstd, const float* coefA,
                      const float* coefB, float* gx, int64_t gx_pitch, int accumulate_gx, float* ggamma,
                      float* gbeta, float affine_beta, float* gss, int64_t gss_pitch, float gss_beta,
                      float* workspace, float* rows, int64_t* desc, void* stream);
/* ABI 7.  lgm_gn_bwd (rows = desc = NULL) / lgm_gn_bwd_deferred (both given) that ALSO adds gy to a second tensor,
 * add_out[b, p, c] += gy[b, p, c] (add_pitch % 4 == 0, >= C; not gx, not gy): the gradient of ResnetBlock's identity
 * residual - reference ddpm.py:187,200, `return h + self.res_conv(x)` with res_conv = nn.Identity - when the block's input
 * gradient is accumulated into a tensor that already holds another branch's gradient (the skip connections of the UNet's
 * down path, ddpm.py:440-447).  Replaces the separate `gx += gy` pass: block2's GroupNorm backward reads gy anyway. */
int lgm_gn_bwd_add(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch, int B, int HW,
                   int C, int G, const float* gamma, const float* beta, const float* ss,
                   int64_t ss_pitch, int act, const float* mean, const float* rstd, const float* coefA,
                   const float* coefB, float* gx, int64_t gx_pitch, int accumulate_gx, float* ggamma,
                   float* gbeta, float affine_beta, float* gss, int64_t gss_pitch, float gss_beta,
                   float* workspace, float* rows, int64_t* desc, float* add_out, int64_t add_pitch, void* stream);
/* Deferred form: when the one-pass kernel applies, the per-image rows [sc*S2 | sc*S1] (B x 2C floats) are left in
 * `rows` and `desc` is filled in the format of lgm_conv_wgrad_deferred (images play the role of splits), so that
 * lgm_wgrad_reduce_batch produces ggamma / gbeta of many layers in one launch; otherwise the gradients are
 * complete on return and desc[6] == 0. */
int lgm_gn_bwd_deferred(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch, int B, int HW,
                        int C, int G, const float* gamma, const float* beta, const float* ss,
                        int64_t ss_pitch, int act, const float* mean, const float* rstd, const float* coefA,
                        const float* coefB, float* gx, int64_t gx_pitch, int accumulate_gx, float* ggamma,
                        float* gbeta, float affine_beta, float* gss, int64_t gss_pitch, float gss_beta,
                        float* workspace, float* rows, int64_t* desc, void* stream);

/* RMSNorm over channels — ddpm.py:107-113: y = x / max(||x||_2, 1e-12) * g * sqrt(C) (+ res).
 * Backward: gx = [gx +] d/dx (+ res): `res` (or null) is the gradient arriving over the residual branch
 * around the PreNorm'd attention block (ddpm.py:187-193), added in the same pass. */
int lgm_rmsnorm_fwd(const float* x, int64_t x_pitch, const float* g, const float* res,
                    int64_t res_pitch, float* y, int64_t y_pitch, int64_t npix, int C, void* stream);
int64_t lgm_rmsnorm_bwd_workspace(int64_t npix, int C);
int lgm_rmsnorm_bwd(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch,
                    const float* g, float* gx, int64_t gx_pitch, int accumulate_gx, const float* res,
                    int64_t res_pitch, float* gg, float gg_beta, int64_t npix, int C, void* workspace, void* stream);
/* Deferred form: leaves the per-block partial rows of the g gradient in `workspace` and fills a descriptor in
 * the format of lgm_conv_wgrad_deferred (rows play the role of splits), so that lgm_wgrad_reduce_batch sums
 * the g gradients of all RMSNorm layers together with the weight-gradient slabs in one launch. */
int lgm_rmsnorm_bwd_deferred(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch,
                             const float* g, float* gx, int64_t gx_pitch, int accumulate_gx, const float* res,
                             int64_t res_pitch, float* gg, float gg_beta, int64_t npix, int C, void* workspace,
                             int64_t* desc, void* stream);

/* ---------------------------------------------------------------------------------------
 * Attention cores on qkv [B, n, 3*heads*32] (channel = which*hidden + head*32 + d).
 * LinearAttention ddpm.py:217-239 (mem_kv [2, heads, 32, M]); Attention ddpm.py:255-271 +
 * modules/attend.py:97-126 (mem_kv [2, heads, M, 32]).  out: [B, n, heads*32].
 * ------------------------------------------------------------------------------------- */
int lgm_linattn_fwd(const float* qkv, int64_t qkv_pitch, const float* mem_kv, int B, int n,
                    int heads, int dim_head, int M, float* out, int64_t out_pitch, float* ctx,
                    float* kmax, float* ksum, void* stream);
/* Head of the attention blocks' forward in one launch (ddpm.py:224-225, :262-263): xn = RMSNorm_g(x) (ddpm.py:115-121)
 * and qkv = to_qkv(xn), a bias-free 1x1 convolution with weight w [N][C]; N = 384, C in {64, 128, 256}
 * (lgm_rms_qkv_fused_supported: 0 = not built, 1 = built, 2 = built and faster than the two separate launches).  xn is
 * written for the weight gradient. */
int64_t lgm_rms_qkv_fused_supported(int C, int N);
int lgm_rms_qkv_fused(const float* x, int64_t x_pitch, const float* g, const float* w, int C, int N, int64_t npix,
                      float* xn, int64_t xn_pitch, float* qkv, int64_t qkv_pitch, void* stream);
/* Forward with its tail fused (ddpm.py:229-239 + the residual around the block): after the context launch ONE launch
 * computes out = softmax_d(q) scale ctx, o2 = to_out[0](out) (wout [Cout][heads*32], bout [Cout]) and
 * y = RMSNorm_g(o2) + x.  `out` and `o2` are written for the backward pass.  Cout in {64, 128, 256}
 * (lgm_linattn_fwd_fused_supported: 0 = not built, 1 = built, 2 = built and faster than the separate launches). */
int64_t lgm_linattn_fwd_fused_supported(int heads, int dim_head, int Cout);
int lgm_linattn_fwd_fused(const float* qkv, int64_t qkv_pitch, const float* mem_kv, int B, int n, int heads,
                          int dim_head, int M, const float* wout, const float* bout, const float* g, int Cout,
                          const float* x, int64_t x_pitch, float* out, int64_t out_pitch, float* o2,
                          int64_t o2_pitch, float* y, int64_t y_pitch, float* ctx, float* kmax, float* ksum,
                          void* stream);
int64_t lgm_linattn_bwd_workspace(int B, int heads, int dim_head, int M);
int lgm_linattn_bwd(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* gout,
                    int64_t gout_pitch, const float* ctx, const float* kmax, const float* ksum,
                    int B, int n, int heads, int dim_head, int M, float* gqkv, int64_t gqkv_pitch,
                    float* gmem_kv, float gmem_beta, void* workspace, void* stream);
/* The same backward with its tail fused (autograd of ddpm.py:214-239, i.e. LinearAttention.to_qkv's backward folded
 * into the attention core's): gq / gk / gv of a pixel tile stay on the chip; the launch that computes them also produces
 * to_qkv's input gradient gxn[B n, C] = gqkv Wqkv and to_qkv's weight gradient (per-workgroup slabs in `slabs`,
 * lgm_linattn_bwd_fused_slabs bytes, summed by the fixed-order reducer).  Built for C = 64 input channels
 * (lgm_linattn_bwd_fused_supported).  xn = the RMSNorm output to_qkv was applied to, wqkv_t = to_qkv.weight transposed,
 * [C][3*heads*32].  `gw_desc` / `gmem_desc` (8 int64, rows of lgm_wgrad_reduce_batch; [6] = 0: no row): non-null = the
 * reductions are left to the caller's batched reducer and `slabs` / `gmem_part` (B*2*heads*32*M floats) must stay
 * untouched until it has run; null = reduced here. */
int64_t lgm_linattn_bwd_fused_supported(int heads, int dim_head, int C);
int64_t lgm_linattn_bwd_fused_slabs(int B, int n, int C);
int64_t lgm_linattn_bwd_fused_workspace(int B, int heads, int dim_head);
int lgm_linattn_bwd_fused(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* gout,
                          int64_t gout_pitch, const float* ctx, const float* kmax, const float* ksum,
                          const float* xn, int64_t xn_pitch, const float* wqkv_t, int C, int B, int n,
                          int heads, int dim_head, int M, float* gxn, int64_t gxn_pitch, float* gwqkv,
                          float gw_beta, void* slabs, int64_t slab_bytes, int64_t* gw_desc, float* gmem_kv,
                          float gmem_beta, void* gmem_part, int64_t* gmem_desc, void* workspace, void* stream);
/* Deferred forms of the two backward entry points: the per-image partial rows of the mem_kv gradient are left in
 * `gmem_part` (B*2*heads*32*M floats; must stay untouched until the caller's lgm_wgrad_reduce_batch has run) and
 * `gmem_desc` (8 int64) receives their reducer row ([6] = 0: no row) - one batched reduction per exchange bucket instead
 * of a column-sum launch pair per attention block.  gmem_kv must be 16-byte aligned. */
int lgm_linattn_bwd_deferred(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* gout,
                             int64_t gout_pitch, const float* ctx, const float* kmax, const float* ksum,
                             int B, int n, int heads, int dim_head, int M, float* gqkv, int64_t gqkv_pitch,
                             float* gmem_kv, float gmem_beta, void* gmem_part, int64_t* gmem_desc,
                             void* workspace, void* stream);
int lgm_attn_bwd_deferred(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* out,
                          int64_t out_pitch, const float* gout, int64_t gout_pitch, const float* lse, int B,
                          int n, int heads, int dim_head, int M, float* gqkv, int64_t gqkv_pitch,
                          float* gmem_kv, float gmem_beta, void* gmem_part, int64_t* gmem_desc, void* stream);
int lgm_attn_fwd(const float* qkv, int64_t qkv_pitch, const float* mem_kv, int B, int n, int heads,
                 int dim_head, int M, float* out, int64_t out_pitch, float* lse, void* stream);
int64_t lgm_attn_bwd_workspace(int B, int heads, int dim_head, int M);
int lgm_attn_bwd(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* out,
                 int64_t out_pitch, const float* gout, int64_t gout_pitch, const float* lse, int B,
                 int n, int heads, int dim_head, int M, float* gqkv, int64_t gqkv_pitch,
                 float* gmem_kv, float gmem_beta, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------
 * Elementwise / data-movement kernels.
 * ------------------------------------------------------------------------------------- */
/* SinusoidalPosEmb ddpm.py:125-132: out[b] = cat(sin(t_b f), cos(t_b f)).  freqs[dim/2] (device) is the
 * table f_i = exp(-i ln(theta)/(dim/2-1)) computed by the caller on the host, as the reference does. */
int lgm_posemb(const int64_t* t, int B, int dim, const float* freqs, float* out, int64_t pitch, void* stream);

#define LGM_ACT_NONE 0
#define LGM_ACT_SILU 1  /* nn.SiLU  ddpm.py:162,180 */
#define LGM_ACT_GELU 2  /* nn.GELU (erf) ddpm.py:331 */
#define LGM_ACT_RELU 3  /* residual.py:13-16, vqvae.py:38-42, dcgan.py:90 */
#define LGM_ACT_LRELU 4 /* dcgan.py:161 (slope 0.2) */
#define LGM_ACT_TANH 5  /* dcgan.py:90, vqvae.py:85 */
/* y = act(x + bias) (+ res);  gx (+)= gy * act'(x + bias).  [rows, cols] matrices with pitches. */
int lgm_act_fwd(const float* x, int64_t x_pitch, const float* bias, const float* res,
                int64_t res_pitch, float* y, int64_t y_pitch, int64_t rows, int cols, int act,
                float slope, void* stream);
int lgm_act_bwd(const float* x, int64_t x_pitch, const float* bias, const float* gy,
                int64_t gy_pitch, float* gx, int64_t gx_pitch, int accumulate, int64_t rows,
                int cols, int act, float slope, void* stream);
/* y = alpha*a + beta*b (b optional) */
int lgm_axpby(const float* a, int64_t a_pitch, float alpha, const float* b, int64_t b_pitch,
              float beta, float* y, int64_t y_pitch, int64_t rows, int cols, void* stream);
/* nn.Upsample(scale_factor=2, mode="nearest") ddpm.py:95; x [B,H,W,C] -> y [B,2H,2W,C] */
int lgm_upsample2x_fwd(const float* x, int64_t x_pitch, float* y, int64_t y_pitch, int B, int H,
                       int W, int C, void* stream);
int lgm_upsample2x_bwd(const float* gy, int64_t gy_pitch, float* gx, int64_t gx_pitch, int B, int H,
                       int W, int C, int accumulate, void* stream);
/* Rearrange("b c (h p1) (w p2) -> b (c p1 p2) h w") ddpm.py:102 in NHWC.  Hlo/Wlo: low-res size,
 * C: high-res channels.  inverse=0: src hi-res [B,2H,2W,C] -> dst lo-res [B,H,W,4C];
 * inverse=1: src lo-res -> dst hi-res (the gradient), optionally accumulated. */
int lgm_pixel_unshuffle(const float* src, int64_t src_pitch, float* dst, int64_t dst_pitch, int B,
                        int Hlo, int Wlo, int C, int inverse, int accumulate, void* stream);
int lgm_nchw_to_nhwc(const float* src, float* dst, int64_t dst_pitch, int B, int C, int HW, int Cpad,
                     void* stream);
int lgm_nhwc_to_nchw(const float* src, int64_t src_pitch, float* dst, int B, int C, int HW,
                     void* stream);

/* Diffusion training elementwise: GaussianDiffusion.forward/q_sample/predict_v
 * ddpm.py:945, 869-876, 684-688.  img, noise: NCHW dense; xt, target: NHWC pitch (pad zeroed). */
int lgm_qsample_target(const float* img, const float* noise, const int64_t* t, const float* sqrt_ac,
                       const float* sqrt_1mac, int normalize, float* xt, float* target,
                       int64_t pitch, int B, int C, int HW, int Cpad, void* stream);
/* loss = mean_b( w[t_b] * mean_{chw} (out - target)^2 )  ddpm.py:921-925 (loss_weight NULL => 1) */
int lgm_weighted_mse_fwd(const float* out, const float* target, int64_t pitch, const int64_t* t,
                         const float* loss_weight, int B, int C, int HW, int Cpad, float* per_sample,
                         float* loss, void* stream);
int lgm_weighted_mse_bwd(const float* out, const float* target, int64_t pitch, const int64_t* t,
                         const float* loss_weight, const float* gloss, int B, int C, int HW, int Cpad,
                         float* gout, void* stream);

/* The UNet's time embedding in one launch: SinusoidalPosEmb ddpm.py:119-132 -> time_mlp (Linear, GELU, Linear) :328-333 ->
 * the SiLU every ResnetBlock.mlp applies first :181-183.  t [B] int64; freqs [dim/2] as lgm_posemb takes them; w1 [time_dim][dim],
 * w2 [time_dim][time_dim] (torch Linear layout, dense).  Outputs, all dense and all kept for the backward:
 *   pe [B][dim], a1 = pe w1^T + b1, h = gelu(a1), temb = h w2^T + b2, st = silu(temb)   ([B][time_dim] each).
 * A row's results do not depend on the batch it is part of (fixed per-row summation order). */
int64_t lgm_time_mlp_supported(int dim, int time_dim);   /* 1 / 0: query, not a status code */
int lgm_time_mlp_fwd(const int64_t* t, int B, int dim, const float* freqs, const float* w1, const float* b1,
                     const float* w2, const float* b2, int time_dim, float* pe, float* a1, float* h, float* temb,
                     float* st, void* stream);
/* Its backward from gst = d loss / d st (two launches: the row-local chain, then the batch reductions, rows in order):
 *   gtemb = gst silu'(temb), ga1 = (gtemb w2) gelu'(a1)   (scratch outputs [B][time_dim]),
 *   gw2 = beta gw2 + gtemb^T h, gb2 = beta gb2 + colsum(gtemb), gw1 = beta gw1 + ga1^T pe, gb1 = beta gb1 + colsum(ga1). */
int lgm_time_mlp_bwd(const float* gst, const float* pe, const float* a1, const float* h, const float* temb,
                     const float* w2, int B, int dim, int time_dim, float* gtemb, float* ga1, float* gw1, float* gb1,
                     float* gw2, float* gb2, float beta, void* stream);

/* GaussianDiffusion's per-sample-timestep algebra on dense NCHW tensors (`extract(table, t, shape) * ...`):
 *   out[b][i] = clamp?( (ta[t_b] * x[b][i] + sb * tb[t_b] * y[b][i]) / td[t_b] )     i < per_sample
 * ta / tb / td = NULL read as 1; products and the sum are rounded separately (no contraction), like the reference's ATen
 * expression.  One entry point for q_sample ddpm.py:869-876 (sqrt_ac, sqrt_1mac, +1), predict_v :684-688, predict_start_from_v
 * :690-694, predict_start_from_noise :673-677 (sb = -1), predict_noise_from_start :679-682 (ta = sqrt_recip, tb = NULL,
 * sb = -1, td = sqrt_recipm1) and q_posterior's mean :697-700.  clip != 0: clamp to [-1, 1] (maybe_clip :711-713). */
int lgm_extract_axpby(const float* ta, const float* tb, const float* td, const int64_t* t, const float* x,
                      const float* y, float sb, int clip, float* out, int B, int64_t per_sample, int n_table,
                      void* stream);
/* model_predictions ddpm.py:707-734, pred_v branch, in one pass over dense NCHW tensors with a per-sample t:
 *   x_start = maybe_clip(sqrt_ac[t] * x - sqrt_1mac[t] * v) ;  pred_noise = (sqrt_recip[t] * x - x_start) / sqrt_recipm1[t] */
int lgm_model_predictions(const float* x, const float* v, const int64_t* t, const float* sqrt_ac,
                          const float* sqrt_1mac, const float* sqrt_recip, const float* sqrt_recipm1, int clip,
                          float* pred_noise, float* x_start, int B, int64_t per_sample, int n_table, void* stream);

/* One reverse-diffusion update at a shared timestep (model_predictions ddpm.py:707-734 pred_v branch,
 * p_sample :748-757, ddim_sample loop body :805-829):
 *   x0 = clamp(A*x + Bv*v) ; eps = (R*x - x0)/Rm1 ; out = C0*x0 + C1*x + C2*eps + C3*noise
 * x, v, out, x0_out: dense NHWC with Cpad channels; noise: NCHW dense or NULL. */
int lgm_sample_step(const float* x, const float* v, const float* noise, float* out, float* x0_out,
                    int B, int C, int HW, int Cpad, float A, float Bv, int clip, float R, float Rm1,
                    float C0, float C1, float C2, float C3, void* stream);

/* The same update for a HIP-graph-replayed chain (p_sample_loop ddpm.py:759-780, ddim_sample :782-834 without the
 * per-step host round trip :775,829): scalars from row counter[0] of table[n_steps][8] = (A, Bv, R, Rm1, C0, C1, C2,
 * C3), x updated IN PLACE; lgm_sampler_time writes t[b] = ttable[counter[0]] for the UNet forward of the step;
 * advance != 0 appends counter[0] += 1 (the last node of a step). */
int lgm_sampler_time(const int64_t* ttable, const int32_t* counter, int64_t* t, int B, void* stream);
int lgm_sample_step_table(float* x, const float* v, const float* noise, float* x0_out, int B, int C, int HW,
                          int Cpad, const float* table, const int32_t* counter, int clip, int advance,
                          void* stream);

/* ---------------------------------------------------------------------------------------
 * Non-fused Winograd engine (csrc/winograd_eng.hip): input transform launch -> ONE batched weight-stationary fp32 MFMA GEMM
 * -> output transform launch, for the layers the fused Winograd kernels' workgroup shapes cannot fill:
 *   f43      3x3 / stride 1 / pad 1 as F(4x4, 3x3) (Block.proj ddpm.py:160 on the 4 x 4 maps: one tile per image);
 *   f42 xy   4x4 / stride 2 / pad 1 Conv2d forward (Discriminator dcgan.py:150-158) as F(4x4, 2x2) on the four pixel phases of
 *            the input, the phases summed inside the GEMM (k = (2p + q) * C + c): 100 products per 16 outputs instead of 256;
 *   f42 yx   the same layer's input gradient = ConvTranspose2d forward (Generator dcgan.py:79-87): four per-phase problems.
 * Tensors: V[xi][T][K], U[xi][N][K] (prepared by the caller: lgm_hip/weng.py), M[xi][T][N], all dense fp32; T = tiles.
 * ------------------------------------------------------------------------------------- */
/* C[b][m][n] = sum_k A[b][m][k] * Bm[b][n][k], b < batch (strides in floats).  K, lda, ldb, batch strides % 4 == 0. */
int lgm_weng_gemm(const float* A, const float* Bm, float* C, int M, int N, int K, int lda, int ldb, int ldc, int batch,
                  int64_t a_batch, int64_t b_batch, int64_t c_batch, void* stream);
/* The same kernel as ONE un-split GEMM with a bias / residual epilogue: C[m][n] = bias[n] + res[m][ldr ...] + sum_k A[m][k] Bm[n][k]
 * (a 1x1 Conv2d / Linear: ddpm.py:96-103, 187, 213-215, 252-253; weights [N][K] as the library stores them; res may alias C). */
int lgm_weng_gemm_epi(const float* A, const float* Bm, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                      const float* bias, const float* res, int ldr, void* stream);
/* x [B][H][W][C] (pitch) -> V[36][T][C], T = B (H/4) (W/4);  M[36][T][N] -> y [B][H][W][N] (pitch) + bias */
int lgm_weng_f43_in(const float* x, int64_t x_pitch, int B, int H, int W, int C, float* V, void* stream);
int lgm_weng_f43_out(const float* M, int B, int H, int W, int N, const float* bias, float* y, int64_t y_pitch,
                     void* stream);
/* x [B][H][W][C] -> V[25][T][4C], T = B (H/8) (W/8);  M[25][T][N] -> y [B][H/2][W/2][N] + bias */
int lgm_weng_f42_in_xy(const float* x, int64_t x_pitch, int B, int H, int W, int C, float* V, void* stream);
int lgm_weng_f42_out_xy(const float* M, int B, int Ho, int Wo, int N, const float* bias, float* y, int64_t y_pitch,
                        void* stream);
/* dy [B][Ho][Wo][Ny] -> V[4][25][T][Ny], T = B (Ho/4) (Wo/4);  M[4][25][T][C] -> dx [B][2 Ho][2 Wo][C] + bias */
int lgm_weng_f42_in_yx(const float* dy, int64_t pitch, int B, int Ho, int Wo, int Ny, float* V, void* stream);
int lgm_weng_f42_out_yx(const float* M, int B, int Ho, int Wo, int C, const float* bias, float* dx, int64_t pitch,
                        void* stream);
/* The output transforms with LgmPostOp's elementwise part in them: out = act(v + bias) (act 0 / 3 ReLU / 4 LeakyReLU), then
 * * (mask > 0 ? 1 : mask_slope) with mask = a saved activation of the output's shape (NULL: none). */
int lgm_weng_f42_out_xy_post(const float* M, int B, int Ho, int Wo, int N, const float* bias, float* y, int64_t y_pitch,
                             int act, float slope, const float* mask, int64_t mask_pitch, float mask_slope, void* stream);
int lgm_weng_f42_out_yx_post(const float* M, int B, int Ho, int Wo, int C, const float* bias, float* dx, int64_t pitch,
                             int act, float slope, const float* mask, int64_t mask_pitch, float mask_slope, void* stream);
/* U of a 4x4 / stride-2 layer for both directions from its weights w [Nw][16][Cw] (float64 arithmetic, rounded once):
 * Uxy [25][Nw][4 Cw], Uyx [4][25][Cw][Nw]; either may be NULL. */
int lgm_weng_f42_weights(const float* w, int Nw, int Cw, float* Uxy, float* Uyx, void* stream);

/* ---------------------------------------------------------------------------------------
 * Vector quantiser (VQ-VAE) — models/modules/vector_quantizer.py.
 * x: latents as [N = B*H*W, D] rows (NHWC), codebook [K, D].
 * ------------------------------------------------------------------------------------- */
/* _quantize :53-60: indices[n] = argmin_k ( ||x_n||^2 + ||e_k||^2 - (2 x_n).e_k ), int64, lowest
 * index on ties; min_dist optional.  The [N,K] distance matrix is never materialised. */
int lgm_vq_assign(const float* x, int64_t x_pitch, const float* codebook, int N, int K, int D,
                  int64_t* indices, float* min_dist, void* stream);
/* one-hot^T @ x and one-hot.sum(0) (:132-143) without the one-hot: dw [K,D], counts [K];
 * deterministic (rows visited in ascending order). */
int lgm_vq_segment_sum(const float* x, int64_t x_pitch, const int64_t* indices, int N, int K, int D,
                       float* dw, float* counts, void* stream);
/* VectorQuantizerEMA._ema_update :128-147 (in place on the two buffers and the codebook). */
int lgm_vq_ema_update(float* cluster_size, float* ema_embedding, float* codebook,
                      const float* counts, const float* dw, int K, int D, float decay, float eps,
                      void* stream);
/* q[n] = codebook[indices[n]]; out3 = { vq_loss = mse + commitment*mse (:76-78),
 * perplexity (:87-88), mse }.  workspace: lgm_vq_gather_workspace bytes. */
int64_t lgm_vq_gather_workspace(int N, int D);
int lgm_vq_gather_loss(const float* x, int64_t x_pitch, const float* codebook,
                       const int64_t* indices, const float* counts, int N, int K, int D,
                       float commitment, float* q, int64_t q_pitch, float* out3, void* workspace,
                       void* stream);
/* Backward: gx = gq (straight-through :90-93, optional) + g*commitment*2(x-q)/(ND);
 * gcodebook = beta*gcodebook + g*2(counts*e - dw)/(ND);  g = *g_vq_loss (device scalar). */
int lgm_vq_bwd(const float* x, int64_t x_pitch, const float* q, int64_t q_pitch, const float* gq,
               int64_t gq_pitch, const float* codebook, const float* dw, const float* counts,
               const float* g_vq_loss, float commitment, int N, int K, int D, float* gx,
               int64_t gx_pitch, float* gcodebook, float gcb_beta, void* stream);

/* ---------------------------------------------------------------------------------------
 * Train-mode BatchNorm2d primitives over [M = B*H*W, C] NHWC matrices (dcgan.py:86-90,158-161)
 * and the WGAN-GP loss pieces (wgan.py:84-156).  xhat = (a - mean) * rstd per channel.
 * ------------------------------------------------------------------------------------- */
int64_t lgm_bn_workspace(int64_t rows, int C);
/* batch mean / rstd (biased variance, eps), running stats updated with torch semantics */
int lgm_bn_stats(const float* a, int64_t a_pitch, int64_t rows, int C, float eps, float momentum,
                 float* mean, float* rstd, float* running_mean, float* running_var, void* workspace,
                 void* stream);
/* BatchNorm statistics folded into the producing convolution (dcgan.py:86-90,158-161: Conv2d / ConvTranspose2d
 * followed by BatchNorm2d): lgm_conv_xy_stats / lgm_conv_yx_stats are lgm_conv_xy / lgm_conv_yx without bias and
 * residual that also leave, per row tile, (sum, squared deviations from the tile's mean, rows) of every output
 * column in stats[tile][3][C]; *stats_tiles = tiles written, 0 when this geometry cannot (then call lgm_bn_stats).
 * stats holds lgm_conv_stats_floats(g, yx) floats.  lgm_bn_stats_from_tiles finishes mean / rstd / running stats. */
int64_t lgm_conv_stats_floats(const LgmConvGeom* g, int yx);
int lgm_conv_xy_stats(const LgmConvGeom* g, const float* x, int64_t x_pitch, const float* w, float* y,
                      int64_t y_pitch, void* workspace, int64_t workspace_bytes, float* stats,
                      int* stats_tiles, void* stream);
int lgm_conv_yx_stats(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* w,
                      const float* w_t, float* x, int64_t x_pitch, void* workspace, int64_t workspace_bytes,
                      float* stats, int* stats_tiles, void* stream);
int lgm_bn_stats_from_tiles(const float* partial, int ntiles, int C, int64_t rows, float eps, float momentum,
                            float* mean, float* rstd, float* running_mean, float* running_var,
                            void* stream);
/* sums3 = [sum v1, sum v1*xhat, sum v1*v2] per channel ([3][C]); a / v2 optional */
int lgm_bn_reduce3(const float* v1, int64_t v1_pitch, const float* v2, int64_t v2_pitch,
                   const float* a, int64_t a_pitch, const float* mean, const float* rstd,
                   int64_t rows, int C, float* sums3, void* workspace, void* stream);
/* per-channel coefficient vectors coef4 = [A1,A2,A3,A4][C]:
 * mode 0: BN backward operator T(v) = c (v - mean v - xhat mean(v xhat)); optional ggamma/gbeta
 *         (= beta_acc*old + sums) and m_out = [mean v, mean v*xhat];
 * mode 1: adjoint of T's dependence on the batch statistics (gradient-penalty second order). */
int lgm_bn_coef(int mode, const float* sums3, const float* gamma, const float* rstd,
                const float* saved_m, int C, int64_t M, float* coef4, float* ggamma, float* gbeta,
                float beta_acc, float* m_out, void* stream);
/* lgm_bn_reduce3 followed by lgm_bn_coef, the coefficient math run inside the second reduction stage (two
 * launches): modes bit 0 -> mode-0 set into coef8[0..4C) with ggamma0 / gbeta0 / m_out, bit 1 -> mode-1 set into
 * coef8[4C..8C) with ggamma1 (needs saved_m).  sums3 (optional) also receives [S1,S2,S3]. */
int lgm_bn_reduce3_coef(int modes, const float* v1, int64_t v1_pitch, const float* v2, int64_t v2_pitch,
                        const float* a, int64_t a_pitch, const float* mean, const float* rstd,
                        const float* gamma, const float* saved_m, int64_t rows, int C, float* coef8,
                        float* ggamma0, float* gbeta0, float beta_acc0, float* m_out, float* ggamma1,
                        float beta_acc1, float* sums3, void* workspace, void* stream);
/* the same second stage over partial sums a convolution's epilogue left (LgmPostOp.bn_partial, `tiles` row tiles) */
int lgm_bn_reduce3_coef_tiles(int modes, const float* partial, int tiles, const float* gamma, const float* rstd,
                              const float* saved_m, int64_t rows, int C, float* coef8, float* ggamma0, float* gbeta0,
                              float beta_acc0, float* m_out, float* ggamma1, float beta_acc1, float* sums3, void* stream);
/* out_k = A1k*v1 + A2k*v2 + A3k*xhat + A4k for the two coefficient sets of coef8 = [2][4][C] (set 0 without its
 * v2 term), one pass over v1, v2, a */
int lgm_bn_affine3x2(const float* v1, int64_t v1_pitch, const float* v2, int64_t v2_pitch, const float* a,
                     int64_t a_pitch, const float* mean, const float* rstd, const float* coef8, float* out0,
                     int64_t out0_pitch, float* out1, int64_t out1_pitch, int64_t rows, int C, void* stream);
/* out (+)= act( A1*v1 + A2*v2 + A3*xhat + A4 )   (any of the terms optional = NULL) */
int lgm_bn_affine3(const float* v1, int64_t v1_pitch, const float* v2, int64_t v2_pitch,
                   const float* a, int64_t a_pitch, const float* mean, const float* rstd,
                   const float* A1, const float* A2, const float* A3, const float* A4, float* out,
                   int64_t out_pitch, int accumulate, int act, float slope, int64_t rows, int C,
                   void* stream);
/* Input pipeline of the reference DataModule on the device (data/datamodule.py:41-53, data/utils.py:7-35):
 * src u8 [B,H,W,C] (decoded images) -> dst fp32 NCHW [B,C,S,S] = Normalize(ToTensor(.)), centre-cropped to
 * min(H,W), resized to SxS (bilinear, antialias=True), flipped horizontally where flip[b] != 0 (NULL: none). */
int lgm_image_transform(const unsigned char* src, int64_t B, int H, int W, int C, const unsigned char* flip,
                        float* dst, int S, void* stream);
/* interpolated_images = alpha*x + (1-alpha)*x_hat (wgan.py:137), alpha [B] */
int lgm_lerp_rows(const float* x, const float* y, const float* alpha, float* out, int64_t B,
                  int64_t rowlen, void* stream);
/* gradient penalty with the reference's channel-only norm (wgan.py:153-156):
 * loss_out = lambda * mean_p (||g_p||_c - 1)^2 ; gbar = d(gscale*loss)/dg.  g, gbar dense NHWC4. */
int64_t lgm_gp_penalty_workspace(int64_t npix);
int lgm_gp_penalty(const float* g, int64_t npix, int C, float lambda, const float* gscale,
                   float* loss_out, float* gbar, void* workspace, void* stream);
/* R1 regulariser on real data (r1gan.py:74-77): loss_out = 0.5 * mean_b sum_{chw} g^2 (the caller
 * applies hparams.r1_penalty); gbar = d(gscale*loss)/dg = gscale/B * g.  g, gbar dense NHWC4 with a
 * zero padding lane; workspace as lgm_gp_penalty_workspace(npix). */
int lgm_r1_penalty(const float* g, int64_t npix, int64_t B, const float* gscale, float* loss_out,
                   float* gbar, void* workspace, void* stream);
/* out = scale * mean_i v[i*pitch]  (critic score means, wgan.py:85-87) */
int lgm_mean_col(const float* v, int64_t pitch, int64_t n, float scale, float* out, void* stream);
/* out[r][c] = (c == col) ? scale * (vptr ? *vptr : 1) : 0 */
int lgm_fill_col(float* out, int64_t pitch, int64_t n, int ncols, int col, float scale,
                 const float* vptr, void* stream);
/* vals4 = (real, fake, gp, d_loss): d_loss = fake - real + gp (wgan.py:87,98) */
int lgm_wgan_dloss(float* vals4, void* stream);
/* VQVAE._common_step's scalar tail (vqvae.py:184-194) in one launch: vals4 = (recon * w_recon + vq * w_vq, recon, vq,
 * perplexity) with out3 = (vq_loss, perplexity, .) as lgm_vq_gather_loss leaves it; and its mirror image for the
 * backward pass: out2 = (g * w0, g * w1). */
int lgm_vqvae_loss(const float* recon, const float* out3, float w_recon, float w_vq, float* vals4, void* stream);
/* The same from the per-sample reconstruction terms lgm_weighted_mse_fwd leaves (its `loss` may be NULL: no mean launch):
 * their mean is taken here, in mean_kernel's order. */
int lgm_vqvae_loss_samples(const float* per_sample, int n, const float* out3, float w_recon, float w_vq, float* vals4,
                           void* stream);
/* VQ-VAE decoder end (vqvae.py:85-88, :130): x_hat = tanh(pre) and the per-sample reconstruction terms in one pass; and
 * its backward with the loss weights folded in - gpre = d loss / d pre for loss = w_recon mse + ..., g2 = (gloss w_recon,
 * gloss w_vq) left for the quantiser's backward (lgm_vq_bwd's g_vq_loss = g2 + 1). */
int lgm_tanh_mse_fwd(const float* pre, const float* target, int64_t pitch, int B, int C, int HW, int Cpad, float* xh,
                     float* per_sample, void* stream);
int lgm_tanh_mse_bwd(const float* xh, const float* target, int64_t pitch, const float* gloss, float w_recon, float w_vq,
                     int B, int C, int HW, int Cpad, float* gpre, float* g2, void* stream);

/* The VQ-VAE's ResidualStack forward in ONE launch (models/modules/residual.py:5-43; vqvae.py:45-47, :71-73): per layer
 * y = relu(conv3x3(cur)), cur' = relu(conv1x1(y) + cur).  x [B,H,W,Cin] arrives with the first in-place ReLU applied;
 * w3[l] [Rh][9][Cin], w1[l] [hidden][1][Rh] (the flat parameter layouts, no bias); y[l] [B,H,W,Rh] and z[l]
 * [B,H,W,hidden] (contiguous) receive every layer's two activations (the backward pass reads them).  Built for 4 x 4
 * maps, Cin = hidden = 128, Rh = 32, <= 4 layers (lgm_resstack_fwd_supported). */
int64_t lgm_resstack_fwd_supported(int H, int W, int Cin, int hidden, int Rh, int layers);
int lgm_resstack_fwd(const float* x, int64_t x_pitch, int B, int H, int W, int Cin, int hidden, int Rh, int layers,
                     const float* const* w3, const float* const* w1, float* const* y, float* const* z, void* stream);
int lgm_scale_pair(const float* g, float w0, float w1, float* out2, void* stream);

/* ---------------------------------------------------------------------------------------
 * OPT-IN split-precision 3x3 convolution (SURVEY.md "bf16x3 ... behind a flag, only if it holds 1e-4";
 * never used unless the caller asks for it): each fp32 operand is split exactly into three bf16 pieces
 * and the product evaluated with six bf16 MFMAs and fp32 accumulation (fp32-level error, measured in
 * tests/test_hip_bf16x3.py).  Same contract as lgm_conv_xy / lgm_conv_yx for 3x3 / stride 1 / pad 1
 * (Block.proj ddpm.py:160-171 and its input gradient), except that the weights are passed as the three
 * planes produced by lgm_split_bf16x3 from the fp32 weights [Nw][9][Cw] (mode 0, forward) or from their
 * transposed copy [Cw][9][Nw] (mode 1, input gradient); plane p starts at w_planes + p*plane_elems.
 * ------------------------------------------------------------------------------------- */
int64_t lgm_conv3x3_bf16x3_supported(const LgmConvGeom* g, int mode, int64_t a_pitch);   /* 1 / 0 */
/* table rows (int32 x 5): element offset of the slot (same in src and in every dst plane), rows, taps, K,
 * first chunk; rows %% 32 == 0, K %% 16 == 0; total_chunks = sum rows*taps*K/8.  The planes are written in the
 * MFMA fragment order the kernel reads (one contiguous 1 KB read per wave and fragment). */
int lgm_split_bf16x3(const float* src, uint16_t* dst, const int32_t* table, int n_slots, int64_t total_chunks,
                     int64_t plane_elems, void* stream);
int lgm_conv3x3_bf16x3(int mode, const LgmConvGeom* g, const float* a, int64_t a_pitch,
                       const uint16_t* w_planes, int64_t plane_elems, const float* bias, const float* res,
                       int64_t res_pitch, float* out, int64_t out_pitch, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Winograd F(2x2, 3x3) convolution in exact fp32 arithmetic (csrc/winograd.hip): the 3x3 / stride 1 / pad 1
 * layers (Block.proj ddpm.py:160-171, Upsample's conv :93-97, the last stages' 3x3 convs :377,413) and their
 * input gradients with 2.25x fewer MFMA FLOPs than the direct form.  Same contract as lgm_conv_xy (yx = 0) /
 * lgm_conv_yx (yx = 1) except that the weights are passed TRANSFORMED: U = G g G^T in MFMA fragment order, as
 * lgm_wino_weights writes them.
 *   lgm_wino_weights: table rows (int64 x 6) = source offset (floats) of a slot w[Np][9][Cp] inside `src`,
 *   Np, Cp (both multiples of 32), destination offset of the forward operand inside dst_f (Np*Cp*16 floats,
 *   layout [Np/32][Cp/8][16][2][32][4]), destination offset of the input-gradient operand inside dst_b
 *   (mirrored taps, roles of Np / Cp swapped), first block; one block per 32 x 32 (n, c) tile, total_blocks =
 *   sum Np/32 * Cp/32.  Either destination may be NULL.
 * ------------------------------------------------------------------------------------- */
int64_t lgm_conv3x3_wino_supported(const LgmConvGeom* g, int yx);   /* 1 / 0 (dense operands fit 32-bit offsets) */
/* 1 when the caller's pitched operands also fit the kernel's 32-bit byte offsets (res_pitch 0: no residual) */
int64_t lgm_conv3x3_wino_fits(const LgmConvGeom* g, int64_t a_pitch, int64_t out_pitch, int64_t res_pitch);
/* Partial form: the split-K planes stay in `workspace` for a consumer that sums them itself (lgm_gn_fwd_planes /
 * lgm_gn_bwd_planes) - no reducer launch, and the split count is planned without its cost.  partial[0] = planes
 * left (1: none, `out` is complete incl. bias), partial[1] = plane stride in floats.  With planes, `out` is not
 * written and `bias` not applied (the consumer adds it). */
int64_t lgm_conv3x3_wino_workspace_partial(const LgmConvGeom* g, int yx);
/* Backward PAIR of a 3x3 layer (autograd's conv backward, Block.proj ddpm.py:160-171): the input gradient
 * gx = W^T * gy (+ res) and the weight / bias gradient gw = beta*gw + gy (x) x in ONE launch - at the per-GPU batches
 * of a strong-scaled run each of them fills a fraction of the chip and is latency-bound, side by side they take
 * the time of one.  Same kernels' code and arithmetic as lgm_conv3x3_wino(yx = 1) + lgm_conv_wgrad: bit-identical.
 * u_b = the input-gradient operand lgm_wino_weights writes; dgrad_ws / partial as lgm_conv3x3_wino[_partial]
 * (partial NULL: finished gx); workspace sizes from lgm_conv3x3_wino_bwd_workspaces; desc NULL: gw complete on return, else
 * the deferred form of lgm_conv_wgrad_deferred. */
int64_t lgm_conv3x3_wino_bwd_supported(const LgmConvGeom* g, int64_t gy_pitch, int64_t x_pitch, int64_t gx_pitch,
                                       int64_t res_pitch);
/* out[0] / out[1] = bytes of dgrad_ws / wgrad_ws the pair needs (its two split counts are planned jointly, so they
 * differ from the stand-alone kernels'); partial != 0: for the partial-planes form */
int lgm_conv3x3_wino_bwd_workspaces(const LgmConvGeom* g, int partial, int64_t* out);
int lgm_conv3x3_wino_bwd(const LgmConvGeom* g, const float* gy, int64_t gy_pitch, const float* x, int64_t x_pitch,
                         const float* u_b, const float* res, int64_t res_pitch, float* gx, int64_t gx_pitch,
                         void* dgrad_ws, int64_t dgrad_ws_bytes, int64_t* partial, float* gw, float* gbias, float beta,
                         void* wgrad_ws, int64_t wgrad_ws_bytes, int64_t* desc, void* stream);
/* Weight gradients of TWO 3x3 layers in one launch (deferred slab reduction: both descriptors as lgm_conv_wgrad_deferred
 * fills them): the chip's workgroups are shared in proportion to the layers' work, each workgroup takes twice the pixels
 * with one prologue and one slab - for the large-map layers whose weights are small and whose input gradient runs apart
 * (lgm_conv3x3_wino4).  Workspace sizes: lgm_conv3x3_wino_wgrad2_workspaces. */
int64_t lgm_conv3x3_wino_wgrad2_supported(const LgmConvGeom* ga, const LgmConvGeom* gb);
/* The same for 2 ... 4 layers (up to 8 when every layer takes the F(4x4) weight-gradient kernel).  LgmWgradItem = one
 * layer's arguments of lgm_conv_wgrad_deferred. */
typedef struct {
  const LgmConvGeom* g;
  const float* y;
  int64_t y_pitch;
  const float* x;
  int64_t x_pitch;
  float* gw;
  float* gbias;
  float beta;
  void* ws;
  int64_t ws_bytes;
  int64_t* desc;
} LgmWgradItem;
/* ABI 7.  The same for the weight / bias gradients of 2 ... 4 1x1 convolutions (reference: autograd's backward of res_conv
 * ddpm.py:187, to_out :215,253, to_qkv :213,252): ONE launch of the streaming 1x1 weight-gradient kernel, every layer on its
 * share of the chip.  All layers must take that kernel on their own (channel counts in whole 64-blocks, whole 64-pixel
 * chunks) and use the same block tile (Nw % 128 and Cw % 128 agree).  Descriptors as lgm_conv_wgrad_deferred (desc[6] = 1:
 * that layer's gradient is complete on return). */
int64_t lgm_wgrad1x1_group_supported(int n, const LgmConvGeom* const* geoms);
/* ABI 7.  Stand-alone launches of the generic weight-gradient kernel may wait for each other (per calling thread): while
 * the queue is ON, lgm_conv_wgrad_deferred / lgm_conv_bwd_pair[_post] with a descriptor do not launch that kernel but queue
 * it; four queued layers - or lgm_wgrad_queue_flush - issue ONE launch with the layers' grids one after the other (each
 * layer's own plan: results are bit-identical to separate launches).  The caller keeps the operands (y, x, workspaces)
 * alive until the flush and flushes before lgm_wgrad_reduce_batch / any reader of the gradients.  lgm_wgrad_queue_enable
 * returns the previous state (on < 0: switch off AND discard what is queued - the start of a pass, in case an earlier one
 * was abandoned half-way); the queue is OFF by default.  (Reference: autograd's weight gradients of the 1x1 / 7x7 /
 * 2x2 convolutions and linears, ddpm.py:103,180,304,330-332,422 - none is read before the optimizer step.) */
int lgm_wgrad_queue_enable(int on);
int lgm_wgrad_queue_flush(void);
int lgm_wgrad1x1_group_workspaces(int n, const LgmConvGeom* const* geoms, int64_t* out);     /* bytes per layer */
int lgm_wgrad1x1_group(int n, const LgmWgradItem* items, void* stream);
int64_t lgm_conv3x3_wino_wgradn_supported(int n, const LgmConvGeom* const* geoms);
int lgm_conv3x3_wino_wgradn_workspaces(int n, const LgmConvGeom* const* geoms, int64_t* out);
int lgm_conv3x3_wino_wgradn(int n, const LgmWgradItem* items, void* stream);
int lgm_conv3x3_wino_wgrad2_workspaces(const LgmConvGeom* ga, const LgmConvGeom* gb, int64_t* out);   /* bytes, a / b */
int lgm_conv3x3_wino_wgrad2(const LgmConvGeom* ga, const float* ya, int64_t ya_pitch, const float* xa, int64_t xa_pitch,
                            float* gwa, float* gba, float beta_a, void* wsa, int64_t wsa_bytes, int64_t* desca,
                            const LgmConvGeom* gb, const float* yb, int64_t yb_pitch, const float* xb, int64_t xb_pitch,
                            float* gwb, float* gbb, float beta_b, void* wsb, int64_t wsb_bytes, int64_t* descb,
                            void* stream);
int lgm_conv3x3_wino_partial(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                             const float* bias, float* out, int64_t out_pitch, void* workspace,
                             int64_t workspace_bytes, int64_t* partial, void* stream);
int64_t lgm_conv3x3_wino_workspace(const LgmConvGeom* g, int yx);   /* split-K partial outputs (bytes) */
int lgm_wino_weights(const float* src, float* dst_f, float* dst_b, const int64_t* table, int n_slots,
                     int64_t total_blocks, void* stream);
/* diagnostic only: buf != NULL makes lgm_conv3x3_wino run its cycle-stamped build (64 int64 per workgroup) */
int lgm_wino_set_debug_buffer(void* buf, int mode);
int lgm_conv3x3_wino(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                     const float* bias, const float* res, int64_t res_pitch, float* out, int64_t out_pitch,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Winograd F(4x4, 3x3) convolution in fp32 (csrc/winograd4.hip): the same layers and the same contract as
 * lgm_conv3x3_wino on the LARGE maps (16 x 16, or H % 16 == 0 and W % 32 == 0; reduction channels % 32, produced
 * channels % 64): 2.25 multiplies per output instead of 4, 1.78x fewer MFMA FLOPs than F(2x2, 3x3) (reference ops:
 * Block.proj ddpm.py:157-173 and its input gradient).  fp32 throughout; error against float64 a few 1e-6 of the output
 * scale (interpolation points 0, +-1, +-2, inf).  Weights are passed TRANSFORMED by lgm_wino4_weights:
 *   table rows as lgm_wino_weights; forward operand Np*Cp*36 floats, layout [Np/64][Cp/8][36][2][2][32][4]; input-gradient
 *   operand with mirrored taps and the roles of Np / Cp swapped; one block per 32 x 32 (n, c) tile.
 * ------------------------------------------------------------------------------------- */
int64_t lgm_conv3x3_wino4_supported(const LgmConvGeom* g, int yx);
int64_t lgm_conv3x3_wino4_workspace(const LgmConvGeom* g, int yx);   /* split-K partial outputs (bytes) */
/* 1 when this layer is expected to run faster here than through lgm_conv3x3_wino (measured table: enough units to fill
 * the chip without deep split-K); callers take F(4x4) only then */
int64_t lgm_conv3x3_wino4_preferred(const LgmConvGeom* g, int yx);
/* partial form, as lgm_conv3x3_wino_partial (same workspace size as lgm_conv3x3_wino4_workspace) */
int lgm_conv3x3_wino4_partial(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                              const float* bias, float* out, int64_t out_pitch, void* workspace,
                              int64_t workspace_bytes, int64_t* partial, void* stream);
/* Forward with the consumer GroupNorm's raw statistics from the epilogue: floats the `stats` buffer needs (0: this
 * geometry does not take it - maps that are multiples of 16 x 32, an unsplit reduction) and the rows per image;
 * lgm_gn_fwd_stats consumes them.  No residual operand. */
int64_t lgm_conv3x3_wino4_stats_floats(const LgmConvGeom* g, int* parts_per_image);
int lgm_conv3x3_wino4_stats(const LgmConvGeom* g, const float* x, int64_t x_pitch, const float* u, const float* bias,
                            float* y, int64_t y_pitch, float* stats, int64_t stats_floats, void* stream);
int lgm_wino4_weights(const float* src, float* dst_f, float* dst_b, const int64_t* table, int n_slots,
                      int64_t total_blocks, void* stream);
/* diagnostic only: buf != NULL makes lgm_conv3x3_wino4 run its cycle-stamped build (32 int64 per workgroup) */
int lgm_wino4_set_debug_buffer(void* buf, int exp);   /* exp: attribution build (drops parts of the phase: wrong results) */
int lgm_conv3x3_wino4(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                      const float* bias, const float* res, int64_t res_pitch, float* out, int64_t out_pitch,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* The same convolution (same operands, same U from lgm_wino4_weights, same contract) from LIGHT workgroups
 * (csrc/winograd4l.hip; reference ops as above, ddpm.py:157-173): 16-tile units, 256 threads, one wave per SIMD, 74 KB of LDS,
 * so that two fit a CU and one fits beside a foreign resident workgroup (a collective's kernel).  Maps 8 x 8 (B % 4 == 0),
 * 16 x 16, or H % 8 == 0 and W % 32 == 0.  Direct entry points for tests and tools; lgm_conv3x3_wino4* choose between the
 * two workgroup sizes themselves (LGM_WINO4_LIGHT). */
/* diagnostic / tests: 1 = light workgroups wherever they take the geometry, 0 = 32-tile workgroups only, -1 = what the
 * environment says (LGM_WINO4_LIGHT; default: light when WORLD_SIZE > 1).  Process-wide; callers that cache plans (split counts, workspace sizes)
 * must not flip it between a size query and the launch it sizes. */
int lgm_wino4_set_light(int mode);

/* Workgroup slots the launch planners leave to a collective resident beside the step (one process per GPU, RCCL's kernel
 * holds a CU per channel): split-K / slab / persistent-range plans are sized for 256 - margin CUs.  Default: LGM_CU_MARGIN,
 * else 16 when WORLD_SIZE > 1, else 0; margin < 0 returns to that default; margin <= 128.  Plans made before the call keep
 * their grids (captured graphs included).  lgm_cu_margin() reads the value in force. */
int lgm_set_cu_margin(int margin);
int lgm_cu_margin(void);
int64_t lgm_conv3x3_wino4l_supported(const LgmConvGeom* g, int yx);
int64_t lgm_conv3x3_wino4l_workspace(const LgmConvGeom* g, int yx);   /* split-K partial outputs (bytes) */
int lgm_conv3x3_wino4l(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                       const float* bias, const float* res, int64_t res_pitch, float* out, int64_t out_pitch,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* Weight gradient of those layers in F(4x4,3x3) form (csrc/winograd4_wgrad.hip; maps with W % 16 == 0, H % 4 == 0,
 * Nw % 64 == 0, Cw % 32 == 0): dU = sum over tiles of (A dY A^T) (.) (B^T x B), dw = G^T dU G, fused bias sums.  Slabs only:
 * `workspace` receives desc[6] slabs for the batched fixed-order reducer; desc as lgm_conv_wgrad_deferred fills it. */
int64_t lgm_conv3x3_wino4_wgrad_supported(const LgmConvGeom* g);
int64_t lgm_conv3x3_wino4_wgrad_workspace(const LgmConvGeom* g);
int lgm_conv3x3_wino4_wgrad(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* x, int64_t x_pitch,
                            float* gw, float* gbias, float beta, void* workspace, int64_t workspace_bytes,
                            int64_t* desc, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimiser kernels on flat storage.
 * torch.optim.Adam (coupled L2; decoupled=1 gives AdamW) — ddpm.py:1053-1059, vqvae.py:207-214,
 * wgan.py:183-195.  step: 1-based count (host value, or read from step_dev when non-NULL).
 * ------------------------------------------------------------------------------------- */
int lgm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                  float b2, float eps, float weight_decay, float step, const float* step_dev,
                  float grad_scale, int decoupled, void* stream);
/* torch.optim.RMSprop with its defaults (momentum 0, not centred) - the critic / generator optimiser
 * of the weight-clipping WGAN (wgan.py:171-181): sq = alpha*sq + (1-alpha)*g^2; p -= lr*g/(sqrt(sq)+eps) */
int lgm_rmsprop_step(float* p, const float* g, float* sq, int64_t n, float lr, float alpha, float eps,
                     float weight_decay, float grad_scale, void* stream);
/* WGAN weight clipping, param.data.clamp_(-c, c) over the critic's flat parameter buffer (wgan.py:158-168) */
int lgm_clamp(float* x, int64_t n, float lo, float hi, void* stream);
/* ema_pytorch.EMA.update (ddpm.py:1047-1048): shadow += (online - shadow) * w   (w = 1 copies) */
int lgm_ema_lerp(float* shadow, const float* online, int64_t n, float w, void* stream);
int lgm_add_scalar(float* x, float a, void* stream);
int lgm_fill(float* x, int64_t n, float val, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LGM_HIP_H */
