/*
 * ganet.h — C ABI of the fused feature-net / loss kernels for MI355X (gfx950).
 *
 * These are the "next" rows of the hot-path scope table: pieces of the
 * per-iteration path that the reference leaves to generic torch ops and that dominate the
 * iteration once the rasterizer is fast.
 *
 *   ganet_linear_wgrad   weight/bias gradient of the decoder's 1x1-conv layers
 *                        (/root/reference/model/modules.py:554-582: conv1..conv8*): dW = g^T x,
 *                        db = sum_m g — a [N,M]x[M,K] GEMM with M = 262,144 and N,K <= 194, i.e.
 *                        a reduction-shaped GEMM that vendor libraries run at ~17 TF; here fp32
 *                        MFMA (32x32x2) with the huge dimension split over workgroups.
 *   ganet_bn_act_*       BatchNorm1d (training statistics) + softplus of the same layers
 *                        (modules.py:535-548,554-560), two passes over HBM instead of ~7.
 *   ganet_ssim_*         SSIM with the 11x11 Gaussian window (/root/reference/utils/loss_utils.py:
 *                        23-53) as one separable tiled pass forward and one backward.
 *
 * Conventions as in gsr.h: device pointers, caller-owned buffers, work enqueued on `stream`,
 * return 0 on success, ganet_last_error() for the text. All tensors fp32, row-major.
 */
#ifndef GANET_H
#define GANET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANET_ABI_VERSION 9
#define GANET_MAX_TERMS 8

/* ---- dW[N,K] = sum_m g[m,n] x[m,k] ; db[N] = sum_m g[m,n] (db may be NULL).
 * g: [M,N] row-major with leading dimension ldg, x: [M,K] with ldx. Supported: N <= 128,
 * K <= 224 (returns 4 = unsupported otherwise; the caller then uses a vendor GEMM).
 * workspace: ganet_linear_wgrad_workspace(M,N,K) bytes. */
size_t ganet_linear_wgrad_workspace(int64_t M, int32_t N, int32_t K);
int ganet_linear_wgrad(int64_t M, int32_t N, int32_t K, const float* g, int64_t ldg,
                       const float* x, int64_t ldx, float* dW, float* db, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ---- Training-mode BatchNorm over rows + activation, y = act(gamma * (x - mean) * rstd + beta).
 * x,y: [M,C] row-major contiguous, C <= 256. act: 0 = identity, 1 = softplus (beta=1,
 * threshold=20 as torch.nn.Softplus). Forward writes mean[C] and rstd[C] (biased variance,
 * eps as given) for the backward pass; if running_mean / running_var / num_batches_tracked are
 * non-NULL they are updated in place exactly as torch.nn.BatchNorm1d does in training mode
 * (unbiased variance, running = (1 - momentum) * running + momentum * batch).
 * workspace: ganet_bn_workspace(M,C) bytes. */
size_t ganet_bn_workspace(int64_t M, int32_t C);
int ganet_bn_act_fwd(int64_t M, int32_t C, const float* x, const float* gamma, const float* beta,
                     float eps, int32_t act, float* y, float* mean, float* rstd,
                     float* running_mean, float* running_var, float momentum,
                     int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes,
                     void* stream);
/* Backward: dy [M,C] -> dx [M,C], dgamma[C], dbeta[C]. */
int ganet_bn_act_bwd(int64_t M, int32_t C, const float* x, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, int32_t act, const float* dy, float* dx,
                     float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---- SSIM (window 11, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) and L1 between img1 and
 * img2, both [planes, H, W] (planes = batch*channels), in one pass over the images. Forward writes
 * sums[0] = norm * sum of the SSIM map and sums[1] = norm * sum |img1 - img2| (norm = 1/(planes*H*W)
 * gives utils/loss_utils.py's ssim(...) and l1_loss_w(...)), and three partial-derivative maps
 * [3, planes, H, W] into `partials` for the backward pass. Backward:
 * dimg1 = norm * (d_ssim[0] * d(sum SSIM)/dimg1 + d_l1[0] * sign(img1 - img2)); d_ssim / d_l1 are device
 * scalars, either may be NULL (= 0). `sums` holds ganet_ssim_sums_floats() floats (the two results first,
 * then the partial sums the waves spread their additions over); the call zeroes all of it. */
int64_t ganet_ssim_sums_floats(void);
int ganet_ssim_fwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2, float norm,
                   float* sums, float* partials, void* stream);
int ganet_ssim_bwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2,
                   const float* partials, float norm, const float* d_ssim, const float* d_l1, float* dimg1,
                   void* stream);

/* ---- fused decoder-MLP layers (ganet_mlp.hip): tall-skinny fp32-MFMA GEMMs whose A operand is
 * activated on load and whose epilogue produces the BatchNorm statistics of the output ----------
 *
 *   z[M,N] = [ x1 | softplus(in_scale . x2 + in_shift) ] [M,K1+K2] . W[N,K1+K2]^T + bias
 *
 * x2 is the previous layer's PRE-activation; in_scale = gamma*rstd and in_shift = beta - mean*in_scale
 * fold its BatchNorm, so Conv1d(k=1) -> BatchNorm1d -> Softplus of
 * /root/reference/model/modules.py:554-582 runs point-major without ever storing a normalised
 * activation. x1 (K1 columns, may be 0) is an un-activated operand (the decoder input / the skip
 * connection into conv5). K1 and K2 are multiples of 8, row strides multiples of 4 floats, operands
 * 16-byte aligned; supported shapes: (K1,K2,N) = (72,0,128) (0,128,128) (72,128,128) (0,128,<=32).
 * col_part (optional, ganet_mlp_stats_floats(N) floats) receives per-workgroup partial column sums and
 * sums of squares of z - stat_shift (stat_shift [N] or NULL = 0: a per-column shift, e.g. the layer's running
 * mean, against the cancellation of E[z^2] - mean^2; pass the same pointer to ganet_mlp_stats, it may alias
 * running_mean); ganet_mlp_stats reduces them to mean/rstd, the folded scale/shift for the
 * NEXT layer's prologue, and updates the running statistics like F.batch_norm(training=True).
 * ganet_wgrad_act is the matching weight gradient: dW[n,k] = sum_m g[m,n] softplus(in_scale_k
 * x[m,k] + in_shift_k), db[n] = sum_m g[m,n] (N, K <= 128). */
/* row_order (ganet_mlp_fwd, ganet_wgrad_act, ganet_mlp_bwd_data): the M rows are independent work, so
 * the order in which the chip sweeps them is free. GANET_ROWS_DEFAULT: the kernel's natural order (forward
 * / data gradient: a common front over all workgroups from row 0 up; weight gradient: one contiguous
 * range per wave). GANET_ROWS_UP / GANET_ROWS_DOWN: a common front first-to-last / last-to-first. A
 * caller that alternates UP and DOWN between consecutive launches over the same activations lets each
 * kernel start on the rows the previous one touched last, which the 256 MiB Infinity Cache still holds
 * (4-5 % per launch measured on the decoder's layers). Results do not depend on it beyond the
 * summation order of the weight gradient / column statistics. */
#define GANET_ROWS_DEFAULT 0
#define GANET_ROWS_UP 1
#define GANET_ROWS_DOWN 2
size_t ganet_mlp_stats_floats(int32_t N);
int ganet_mlp_fwd(int64_t M, int32_t N, int32_t K1, int32_t K2, const float* x1, int64_t ld1,
                  const float* x2, int64_t ld2, const float* in_scale, const float* in_shift,
                  const float* W, const float* bias, float* z, int64_t ldz, float* col_part,
                  const float* stat_shift, int32_t row_order, void* stream);
int ganet_mlp_stats(int64_t M, int32_t N, const float* col_part, const float* gamma,
                    const float* beta, float eps, float* mean, float* rstd, float* scale,
                    float* shift, float* running_mean, float* running_var, float momentum,
                    int64_t* num_batches_tracked, const float* stat_shift, void* stream);
size_t ganet_wgrad_act_workspace(int64_t M, int32_t N, int32_t K);
/* gz/gcoef (both or neither): the g operand is assembled on load as gcoef[0][n] g + gcoef[1][n] gz +
 * gcoef[2][n] — the BatchNorm backward of the layer folded to per-column coefficients (see
 * ganet_mlp_bwd_stats). in_scale/in_shift NULL: x is used as it is (decoder input operand). */
int ganet_wgrad_act(int64_t M, int32_t N, int32_t K, const float* g, int64_t ldg, const float* gz,
                    int64_t ldgz, const float* gcoef, const float* x, int64_t ldx,
                    const float* in_scale, const float* in_shift, float* dW, float* db,
                    void* workspace, size_t workspace_bytes, int32_t row_order, void* stream);
/* dW == NULL above leaves the per-workgroup partial sums in `workspace` (which must then stay untouched);
 * ganet_wgrad_reduce_batch finishes up to GANET_MAX_WGRAD_JOBS such calls with ONE launch (the decoder's
 * backward has 15 weight gradients; their reductions are off the dependency chain). jobs: HOST array. */
#define GANET_MAX_WGRAD_JOBS 20
typedef struct GanetWgradJob {
  const void* workspace;   /* as passed to ganet_wgrad_act */
  int64_t M;
  int32_t N, K;
  float* dW;               /* [N,K] */
  float* db;               /* [N] or NULL */
  int32_t nblocks;         /* 0: the partials of ganet_wgrad_act(M, ...); > 0: that many blocks of N*K+N
                              floats (ganet_mlp_head_bwd's wgrad_part: ganet_mlp_head_bwd_parts() blocks) */
} GanetWgradJob;
int ganet_wgrad_reduce_batch(int32_t n_jobs, const GanetWgradJob* jobs, void* stream);

/* ---- input-gradient side with the BatchNorm + softplus backward folded in (ganet_mlp_bwd.hip) ----
 * For a hidden layer with stored pre-activation z, u = scale z + shift, y = softplus(u):
 *   G  = dL/dy . softplus'(u);   dz = A G + q z + p   (per-column A, q, p from the sums of G and G z).
 * ganet_mlp_bwd_data:  out[M,O] (+)= (A g + q gz + p)[M,128] . W[128, 0:O]   (W = the layer's weight
 *   [128 out, in] or a column slice of it, row stride ldw; O = 128 or <= 96); with src_z != NULL the result is multiplied by softplus'(src_scale src_z +
 *   src_shift) — i.e. out = G of the SOURCE layer — and col_part ([ganet_mlp_bwd_data_parts()][2][128])
 *   receives the partial column sums of out and out . src_z.
 * ganet_mlp_head_bwd:  the same for the narrow output heads (g [M,N8], N8 <= 4, W8 [N8,128]):
 *   G[M,128] = (g W8) softplus'(scale z + shift), partial sums [ganet_mlp_head_bwd_parts()][2][128].
 *   wgrad_part (optional, [ganet_mlp_head_bwd_parts()][N8*128 + N8] floats): the head's own weight
 *   gradient rides along — per-workgroup partial sums of dW8[n,k] = sum_m g[m,n] softplus(scale_k z[m,k]
 *   + shift_k) and db8[n] = sum_m g[m,n], finished by ganet_wgrad_reduce_batch (job.nblocks =
 *   ganet_mlp_head_bwd_parts()); it saves the separate pass of ganet_wgrad_act over z.
 * ganet_mlp_bwd_stats: partial sums -> coef [3][128] = (A, q, p), d gamma, d beta.
 * Together they replace the autograd backward of Conv1d(k=1) -> BatchNorm1d -> Softplus
 * (/root/reference/model/modules.py:554-582) without materialising dz or dL/dy. */
int32_t ganet_mlp_bwd_data_parts(void);
int32_t ganet_mlp_head_bwd_parts(void);
/* ganet_mlp_bwd_fused (ganet_layer_bwd.hip): the backward of a hidden 128 -> 128 layer in ONE pass over the
 * activations — ganet_mlp_bwd_data (O = 128) AND the matching ganet_wgrad_act (N = K = 128, x = softplus(src_scale
 * src_z + src_shift)): 4 instead of 7 [M,128] tensors through HBM (g, gz, src_z in; out written). Same arithmetic as
 * the separate kernels (exactly split fp32 operands on the bf16 matrix pipe). g, gz, src_z, out: [M,128] contiguous,
 * 16-byte aligned, M a multiple of 32; W [128, >= 128] with row stride ldw (a column slice of a wider weight is
 * fine). accumulate: out += ...; apply_act: the result is multiplied by softplus'(src_scale src_z + src_shift) (out = G
 * of the source layer) and col_part [ganet_mlp_bwd_fused_parts()][2][128] receives the partial column sums of out and
 * out . src_z (apply_act = 0: out is the raw dL/dy contribution, col_part untouched — the pattern of several layers
 * accumulating into one source). wgrad_workspace (ganet_mlp_bwd_fused_workspace() bytes) receives
 * ganet_mlp_bwd_fused_parts() partial tiles [128*128 + 128] for ganet_wgrad_reduce_batch (job.nblocks =
 * ganet_mlp_bwd_fused_parts()). */
/* ganet_mlp_bwd_fused_input: the same one-pass backward for a layer whose input is the RAW decoder input x [M,72] (conv1,
 * and the input half of conv5): out [M,72] (+)= dz . W[128, 0:O] (columns >= O receive zeros), dW[n, 0:72] = sum_m
 * dz[m,n] x[m,k] — delivered as the left 72 columns of a 128 x 128 partial tile (reduce with N = K = 128), db likewise. */
int ganet_mlp_bwd_fused_input(int64_t M, const float* g, const float* gz, const float* gcoef, const float* W,
                              int64_t ldw, int32_t O, float* out, int64_t ldo, int32_t accumulate, const float* x,
                              void* wgrad_workspace, size_t workspace_bytes, int32_t row_order, void* stream);
int32_t ganet_mlp_bwd_fused_parts(void);
size_t ganet_mlp_bwd_fused_workspace(void);
int ganet_mlp_bwd_fused(int64_t M, const float* g, const float* gz, const float* gcoef, const float* W, int64_t ldw,
                        float* out, int32_t accumulate, const float* src_z, const float* src_scale,
                        const float* src_shift, int32_t apply_act, float* col_part, void* wgrad_workspace,
                        size_t workspace_bytes, int32_t row_order, void* stream);
int ganet_mlp_bwd_data(int64_t M, int32_t O, const float* g, int64_t ldg, const float* gz, int64_t ldgz,
                       const float* gcoef, const float* W, int64_t ldw, float* out, int64_t ldo,
                       int32_t accumulate,
                       const float* src_z, int64_t ld_src, const float* src_scale,
                       const float* src_shift, float* col_part, int32_t row_order, void* stream);
int ganet_mlp_head_bwd(int64_t M, int32_t N8, const float* g, const float* W8, const float* z,
                       int64_t ldz, const float* scale, const float* shift, float* G, int64_t ldG,
                       float* col_part, float* wgrad_part, void* stream);
int ganet_mlp_bwd_stats(int64_t M, int32_t nparts, const float* col_part, const float* mean,
                        const float* rstd, const float* scale, float* coef, float* dgamma,
                        float* dbeta, void* stream);

/* ---- the whole decoder MLP as ONE call each way (ganet_decoder.hip): the fixed launch sequence of the layer kernels
 * above for the reference's ShapeDecoder (/root/reference/model/modules.py:508-582; forward :554-582) in training mode —
 * what gaussianavatar_amd/fused.py::_DecoderFn otherwise issues one call at a time from Python (~70 calls per
 * iteration). Layer order: 0..4 = conv1..conv5 (conv5 = [x | act(conv4)]), then per head h = 0, 1, 2 (residual, scale,
 * colour): 5 + 2h = conv6*, 6 + 2h = conv7*; W8/b8 = conv8* ([n8,128], n8 <= 4). All pointers device memory; weights in
 * nn.Conv1d layout ([out, in] contiguous). x: [M,72] (the cin input columns, zero padded). running_mean /
 * running_var / num_batches_tracked NULL = statistics not tracked; they are updated like
 * F.batch_norm(training=True). saved (ganet_decoder_saved_floats(M) floats) keeps every layer's pre-activation and
 * BatchNorm statistics for the backward pass. out[h]: [M, n8[h]] logits. */
#define GANET_DEC_LAYERS 11
typedef struct GanetDecoderParams {
  int32_t cin;
  const float* W[GANET_DEC_LAYERS];
  const float* bias[GANET_DEC_LAYERS];
  const float* gamma[GANET_DEC_LAYERS];
  const float* beta[GANET_DEC_LAYERS];
  float* running_mean[GANET_DEC_LAYERS];
  float* running_var[GANET_DEC_LAYERS];
  int64_t* num_batches_tracked[GANET_DEC_LAYERS];
  float eps[GANET_DEC_LAYERS];
  float momentum[GANET_DEC_LAYERS];
  const float* W8[3];
  const float* b8[3];
  int32_t n8[3];
} GanetDecoderParams;
typedef struct GanetDecoderGrads {
  float* dW[GANET_DEC_LAYERS];      /* laid out like W */
  float* db[GANET_DEC_LAYERS];
  float* dgamma[GANET_DEC_LAYERS];
  float* dbeta[GANET_DEC_LAYERS];
  float* dW8[3];
  float* db8[3];
  float* dx;                        /* [M, x_cols] (columns >= cin stay unwritten) or NULL */
  int32_t x_cols;
} GanetDecoderGrads;
size_t ganet_decoder_saved_floats(int64_t M);
size_t ganet_decoder_fwd_workspace(void);
int ganet_decoder_fwd(int64_t M, const float* x, const GanetDecoderParams* params, float* saved, float* const* out,
                      void* workspace, size_t workspace_bytes, void* stream);
/* Backward: d_out[h] [M, n8[h]] (all three required), M a multiple of 32. side_stream (may be NULL): a second stream
 * for the weight-gradient launches that are off the dependency chain; the call orders it against `stream` with events
 * on both sides, so the caller only has to keep the buffers alive in `stream` order. */
size_t ganet_decoder_bwd_workspace(int64_t M);
int ganet_decoder_bwd(int64_t M, const float* x, const GanetDecoderParams* params, const float* saved,
                      const float* const* d_out, const GanetDecoderGrads* grads, void* workspace,
                      size_t workspace_bytes, void* stream, void* side_stream);

/* ---- decoder heads -> per-Gaussian records (ganet_pack.hip) ----------------------------------------
 * One kernel for pred_res * res_scale, the two sigmoid heads (x scale_mult for the scale warm-up), the
 * gather of the N valid texels (valid_index [N], int64, texel index in [0,HW)) and the two regulariser
 * sums: sums[0] = sq_norm * sum over ALL texels of (res_scale * res)^2 (offset regulariser), sums[1] =
 * scale_norm * sum over the valid texels of the scale (scale regulariser); the call zeroes sums first.
 * res [frames*HW,3], scale_logit [frames*HW,1], colour_logit [frames*HW,3];
 * out [frames*N*7] = residual [frames,N,3] | scale [frames,N] | colour [frames,N,3]: three contiguous
 * segments of one buffer (one all-reduce exchanges their gradient in the data-parallel path). Replaces
 * the element-wise chain of /root/reference/model/avatar_model.py:298-324,367-368 + the sigmoids of
 * model/network.py:79-81.
 * Backward: inv_index [HW] (int64; n for a valid texel, -1 otherwise); d_out laid out like out (may be
 * NULL = zero), d_sq / d_scale_sum device scalars (may be NULL); every element of the three gradient
 * tensors is written. */
int ganet_decode_pack_fwd(int32_t frames, int64_t HW, int64_t N, const float* res,
                          const float* scale_logit, const float* colour_logit,
                          const int64_t* valid_index, float res_scale, float scale_mult, float sq_norm,
                          float scale_norm, float* out, float* sums, void* stream);
int ganet_decode_pack_bwd(int32_t frames, int64_t HW, int64_t N, const float* res,
                          const float* scale_logit, const float* colour_logit,
                          const int64_t* inv_index, float res_scale, float scale_mult, float sq_norm,
                          float scale_norm, const float* d_out, const float* d_sq,
                          const float* d_scale_sum, float* d_res, float* d_scale_logit,
                          float* d_colour_logit, void* stream);
/* Gradient of decode_pack's record buffer [residual b*N*3 | scale b*N | colour b*N*3] from the gradients of its views
 * broadcast over B frames (b = 1: summed over the frames; b = B: per frame) with the scale broadcast over three axes:
 * g_res, g_scale3, g_col are [B, N, 3] (any may be NULL = zero). One launch instead of the expand-backward sums. */
int ganet_records_bwd(int32_t b, int32_t B, int64_t N, const float* g_res, const float* g_scale3, const float* g_col,
                      float* d_flat, void* stream);

/* out[0] = norm * sum_i x[i]^2 (x 16-byte aligned; geometry-feature regulariser,
 * /root/reference/model/avatar_model.py:367); backward dx = 2 * norm * d_out[0] * x. */
int ganet_mean_sq_fwd(int64_t n, const float* x, float norm, float* out, void* stream);
int ganet_mean_sq_bwd(int64_t n, const float* x, float norm, const float* d_out, float* dx, void* stream);

/* ---- Adam over a list of tensors in one launch (ganet_optim.hip): torch.optim.Adam's rule with amsgrad,
 * weight decay and maximize off — m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps) — for the optimiser step of
 * /root/reference/model/avatar_model.py:152-161,264-266. tensors: HOST array, n_tensors <=
 * GANET_MAX_ADAM_TENSORS; all pointers device float32, n elements each, contiguous.
 * skip_flag: device int32[1] or NULL. When it is non-zero the launch changes NOTHING (no parameter, no moment):
 * the rasterizer raises this flag when a backward pass had to drop a frame's gradient (include/gsr.h: gsr_backward,
 * `overflow_flag`), so that a step computed from incomplete gradients is never applied — without a host sync. */
#define GANET_MAX_ADAM_TENSORS 64
typedef struct GanetAdamTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;
  float lr, bias_correction1, bias_correction2;
} GanetAdamTensor;
int ganet_adam_step(int32_t n_tensors, const GanetAdamTensor* tensors, float beta1, float beta2, float eps,
                    const int32_t* skip_flag, void* stream);
/* Lower the flag again behind the step's last launch (stream-ordered 4-byte fill): the flag describes ONE optimisation
 * step, whatever the caller's loop does between steps. */
int ganet_flag_clear(int32_t* flag, void* stream);

/* out[0] = bias + sum_{i<n} weights[i] * terms[i][0]: the scalar objective of the training loop
 * (/root/reference/train.py:70-82) in one launch. terms: HOST array of n device pointers, weights: HOST
 * array, n <= GANET_MAX_TERMS. Backward: d_terms[i] = weights[i] * d_out[0] (d_out, d_terms on device). */
int ganet_weighted_sum_fwd(int32_t n, const float* const* terms, const float* weights, float bias,
                           float* out, void* stream);
int ganet_weighted_sum_bwd(int32_t n, const float* weights, const float* d_out, float* d_terms,
                           void* stream);

/* ---- geometry-feature convolutions (ganet_conv.hip): 5x5, stride 1, zero padding 2, no bias, 64 -> 64 channels
 * (/root/reference/model/modules.py:114-137, GeomConvLayers; replaces the three nn.Conv2d calls and their autograd
 * backward). Maps are channels-last fp32 [b][H][W][64] (W a multiple of 64); weights are nn.Conv2d's
 * [64][64][5][5]. ganet_conv5_pack splits the weights of up to GANET_CONV5_MAX convolutions into the bf16 planes
 * both passes read (forward and transposed / tap-flipped for the input gradient; ganet_conv5_packed_bytes bytes);
 * ganet_conv5_apply(..., conv, input_gradient, ...) is y = conv(x, w[conv]) or dL/dx = conv^T(dL/dy, w[conv]);
 * ganet_conv5_wgrad writes dL/dw [64][64][5][5] (workspace: ganet_conv5_wgrad_workspace bytes). Arithmetic as the
 * decoder GEMMs: exactly split fp32 operands on the bf16 matrix pipe, fp32 accumulation. */
#define GANET_CONV5_MAX 4
size_t ganet_conv5_packed_bytes(int32_t n_convs);
int ganet_conv5_pack(int32_t n_convs, const float* const* w, void* packed, void* stream);
int ganet_conv5_apply(int32_t b, int32_t H, int32_t W, const float* x, const void* packed, int32_t conv,
                      int32_t input_gradient, float* y, void* stream);
size_t ganet_conv5_wgrad_workspace(int32_t b, int32_t H, int32_t W);
int ganet_conv5_wgrad(int32_t b, int32_t H, int32_t W, const float* x, const float* dy, float* dw, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- bilinear up-sampling at the separable UV texel grid + decoder-input assembly (ganet_upsample.hip)
 * x[(i,j), 0:C] = sum_{a,b<2} row_w[i,a] col_w[j,b] feat[row_idx[i,a], col_idx[j,b], :], followed by the two
 * uv columns and zeros up to ldx — F.grid_sample(bilinear, align_corners=False, zero padding) at the
 * reference's texel grid + the concatenation with uv (/root/reference/model/network.py:60-66,
 * utils/general_utils.py:165-176). feat [frames,R,R,C] channels-last, C = 64; row/col taps [S,2] (index
 * clamped into range, weight 0 for a tap that falls into the zero padding); uv [frames,S*S,2];
 * x [frames*S*S, ldx]. Backward takes the transposed tap lists in CSR form (ptr [R+1], src, w), a scratch
 * tmp [frames,S,R,C] and writes dfeat [frames,R,R,C] in full. */
int ganet_upsample_cat_fwd(int32_t frames, int32_t S, int32_t R, int32_t C, const float* feat,
                           const int32_t* row_idx, const float* row_w, const int32_t* col_idx,
                           const float* col_w, const float* uv, float* x, int64_t ldx, void* stream);
int ganet_upsample_cat_bwd(int32_t frames, int32_t S, int32_t R, int32_t C, const float* dx, int64_t ldx,
                           const int32_t* row_ptr, const int32_t* row_src, const float* row_w,
                           const int32_t* col_ptr, const int32_t* col_src, const float* col_w,
                           float* tmp, float* dfeat, void* stream);

/* ---- the stage-2 pose encoder as ONE call each way (ganet_unet.hip): UnetNoCond5DS of
 * /root/reference/model/modules.py:185-232 (blocks :62-111) — five 4x4 / stride-2 convolutions down (nf, 2nf, 4nf, 8nf, 8nf
 * channels; BatchNorm2d(affine=False) after conv2..4; LeakyReLU(0.2) in front of conv2..5, in place, so the skip tensors
 * carry it too), five ReLU -> 4x4 / stride-2 transposed convolutions up with skip concatenations (BatchNorm after
 * upconv1..4, bias on upconv5). x: [B, cin, S, S] NCHW (S a power of two >= 32, cin <= 8); out: [B, S, S, cout]
 * channels-last; nf and cout multiples of 32. Weights in torch's layouts: Wd[k] = conv{k+1}.conv.weight
 * [co][ci][4][4], Wu[k] = upconv{k+1}.up.weight [ci][co][4][4], bias5 = upconv5.up.bias. BatchNorm index: 0..2 =
 * conv2..4, 3..6 = upconv1..4 (running statistics updated like F.batch_norm(training=True); training = 0: the running
 * statistics normalise). saved (ganet_unet_saved_floats floats) keeps the raw layer outputs and the statistics for
 * the backward pass, which expects the forward pass to have run in training mode. Replaces ~40 im2col / GEMM /
 * col2im / batch_norm / element-wise launches per pass. ganet_unet_bwd's side_stream (may be NULL) takes the weight
 * gradients, which hang off the input-gradient chain; it is ordered against `stream` inside the call (events both
 * ways), so every result is ready in `stream` order. */
#define GANET_UNET_BN 7
typedef struct GanetUnetParams {
  int32_t cin, nf, cout, S;
  const float* Wd[5];
  const float* Wu[5];
  const float* bias5;
  float* running_mean[GANET_UNET_BN];
  float* running_var[GANET_UNET_BN];
  int64_t* num_batches_tracked[GANET_UNET_BN];
  float eps, momentum;
} GanetUnetParams;
typedef struct GanetUnetGrads {
  float* dWd[5];
  float* dWu[5];
  float* dbias5;
} GanetUnetGrads;
size_t ganet_unet_saved_floats(const GanetUnetParams* params, int32_t B);
size_t ganet_unet_fwd_workspace(const GanetUnetParams* params, int32_t B);
int ganet_unet_fwd(const GanetUnetParams* params, int32_t B, const float* x, int32_t training, float* saved, float* out,
                   void* workspace, size_t workspace_bytes, void* stream);
size_t ganet_unet_bwd_workspace(const GanetUnetParams* params, int32_t B);
int ganet_unet_bwd(const GanetUnetParams* params, int32_t B, const float* x, const float* saved, const float* d_out,
                   const GanetUnetGrads* grads, void* workspace, size_t workspace_bytes, void* stream, void* side_stream);

/* Optional per-kernel timing (bench/profiling only). The caller owns a GanetProfile object and binds it to the
 * calling thread; from then on every instrumented launch this thread makes whose kernel id is in `mask` is bracketed by
 * hipEvents recorded on the launch stream and accounted to that object (bind NULL or mask 0 to stop). ganet_profile_read
 * waits for the recorded events and returns, per kernel id < ganet_profile_count(), the summed GPU time in ms and the
 * launch count since the last reset. No process-global state: the binding is thread-local, like the error text. */
typedef struct GanetProfile GanetProfile;
GanetProfile* ganet_profile_create(void);
void ganet_profile_destroy(GanetProfile* profile);
int ganet_profile_bind(GanetProfile* profile, int mask);   /* bit k = time kernel id k */
int ganet_profile_count(void);
int ganet_profile_read(GanetProfile* profile, double* ms_sum, int64_t* launches, int reset);
const char* ganet_profile_kernel_name(int id);

/* Decoder GEMM arithmetic (not switchable: the library has no process-global state):
 * fp32 operands split exactly into three bf16 pieces, six bf16 MFMAs per step accumulated in fp32 (csrc/ganet_split.h:
 * fp32-accurate against float64, 6/16 of the fp32-MFMA pipe time). */
const char* ganet_last_error(void);
int ganet_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GANET_H */
