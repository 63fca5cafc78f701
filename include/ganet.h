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

#define GANET_ABI_VERSION 1

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

/* ---- SSIM (window 11, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) between img1 and
 * img2, both [planes, H, W] (planes = batch*channels). Forward writes sum over all elements of
 * the SSIM map to ssim_sum[0] (caller divides by planes*H*W) and three partial-derivative maps
 * [3, planes, H, W] into `partials` for the backward pass. Backward: dL/dimg1 = scale *
 * d(ssim_sum)/dimg1 written to dimg1 [planes,H,W]. */
int ganet_ssim_fwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2,
                   float* ssim_sum, float* partials, void* stream);
int ganet_ssim_bwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2,
                   const float* partials, const float* scale_dev, float* dimg1, void* stream);

const char* ganet_last_error(void);
int ganet_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GANET_H */
