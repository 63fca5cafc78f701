// ganet_pack.hip — glue between the decoder heads and the per-Gaussian pipeline, fused.
//
// The reference does, on the decoder's three outputs ([B,3,HW], [B,1,HW], [B,3,HW]):
//   pred_res * 0.02, sigmoid heads, scale warm-up, permute, boolean-mask gather of the valid texels,
//   repeat of the scale to 3 channels, and mean(pred_res^2) for the offset regulariser
// (/root/reference/model/avatar_model.py:298-324, model/network.py:69-81) — about twenty element-wise
// launches with their autograd mirrors. Here: one forward kernel that reads the heads' logits and
// writes the packed per-Gaussian record [N,7] = (residual 3, scale 1, colour 3) plus the sum of
// squared residuals, and one backward kernel that writes the three logit gradients in full.
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"

namespace ganet {

namespace {

__device__ __forceinline__ float sigmoid_f(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v * 1.4426950408889634f));
}

__global__ void __launch_bounds__(256)
decode_pack_fwd_kernel(int64_t HW, int64_t N, const float* __restrict__ res,
                       const float* __restrict__ scale_logit, const float* __restrict__ colour_logit,
                       const int64_t* __restrict__ valid_index, float res_scale, float scale_mult,
                       float* __restrict__ packed, float* __restrict__ res_sq_sum) {
  const int64_t f = blockIdx.y;
  res += f * HW * 3; scale_logit += f * HW; colour_logit += f * HW * 3; packed += f * N * 7;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t n = t0; n < N; n += stride) {
    const int64_t m = valid_index[n];
    float* o = packed + n * 7;
    o[0] = res[3 * m] * res_scale;
    o[1] = res[3 * m + 1] * res_scale;
    o[2] = res[3 * m + 2] * res_scale;
    o[3] = sigmoid_f(scale_logit[m]) * scale_mult;
    o[4] = sigmoid_f(colour_logit[3 * m]);
    o[5] = sigmoid_f(colour_logit[3 * m + 1]);
    o[6] = sigmoid_f(colour_logit[3 * m + 2]);
  }
  // sum over ALL texels of (res_scale * res)^2
  float s = 0.f;
  for (int64_t i = t0; i < HW * 3; i += stride) {
    const float v = res[i] * res_scale;
    s = fmaf(v, v, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float s_red[4];
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(res_sq_sum, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
}

__global__ void __launch_bounds__(256)
decode_pack_bwd_kernel(int64_t HW, int64_t N, const float* __restrict__ res,
                       const float* __restrict__ scale_logit, const float* __restrict__ colour_logit,
                       const int64_t* __restrict__ inv_index, float res_scale, float scale_mult,
                       const float* __restrict__ d_packed, const float* __restrict__ d_sq_sum,
                       float* __restrict__ d_res, float* __restrict__ d_scale_logit,
                       float* __restrict__ d_colour_logit) {
  const int64_t f = blockIdx.y;
  res += f * HW * 3; scale_logit += f * HW; colour_logit += f * HW * 3; d_packed += f * N * 7;
  d_res += f * HW * 3; d_scale_logit += f * HW; d_colour_logit += f * HW * 3;
  const float k = d_sq_sum ? 2.0f * res_scale * res_scale * d_sq_sum[0] : 0.f;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < HW;
       m += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = inv_index[m];
    float g[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n >= 0) {
#pragma unroll
      for (int c = 0; c < 7; ++c) g[c] = d_packed[n * 7 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) d_res[3 * m + c] = fmaf(k, res[3 * m + c], g[c] * res_scale);
    const float ss = sigmoid_f(scale_logit[m]);
    d_scale_logit[m] = g[3] * scale_mult * ss * (1.0f - ss);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sg = sigmoid_f(colour_logit[3 * m + c]);
      d_colour_logit[3 * m + c] = g[4 + c] * sg * (1.0f - sg);
    }
  }
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

int ganet_decode_pack_fwd(int32_t frames, int64_t HW, int64_t N, const float* res,
                          const float* scale_logit, const float* colour_logit,
                          const int64_t* valid_index, float res_scale, float scale_mult, float* packed,
                          float* res_sq_sum, void* stream_) {
  if (frames <= 0 || HW <= 0 || N < 0 || N > HW || !res || !scale_logit || !colour_logit ||
      (N > 0 && (!valid_index || !packed)) || !res_sq_sum) {
    set_error("ganet_decode_pack_fwd: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_hip(hipMemsetAsync(res_sq_sum, 0, sizeof(float), stream), "memset res_sq_sum");
  if (rc) return rc;
  const int blocks = (int)((HW * 3 / 4 + 255) / 256 < 2048 ? (HW * 3 / 4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(decode_pack_fwd_kernel, dim3(blocks > 0 ? blocks : 1, frames), dim3(256), 0, stream, HW, N,
                     res, scale_logit, colour_logit, valid_index, res_scale, scale_mult, packed, res_sq_sum);
  return check_hip(hipGetLastError(), "decode_pack_fwd_kernel");
}

int ganet_decode_pack_bwd(int32_t frames, int64_t HW, int64_t N, const float* res,
                          const float* scale_logit, const float* colour_logit,
                          const int64_t* inv_index, float res_scale, float scale_mult,
                          const float* d_packed, const float* d_sq_sum, float* d_res,
                          float* d_scale_logit, float* d_colour_logit, void* stream_) {
  if (frames <= 0 || HW <= 0 || N < 0 || !res || !scale_logit || !colour_logit || !inv_index ||
      (N > 0 && !d_packed) || !d_res || !d_scale_logit || !d_colour_logit) {
    set_error("ganet_decode_pack_bwd: invalid arguments");
    return 1;
  }
  const int blocks = (int)((HW + 255) / 256 < 4096 ? (HW + 255) / 256 : 4096);
  hipLaunchKernelGGL(decode_pack_bwd_kernel, dim3(blocks, frames), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), HW, N, res, scale_logit, colour_logit, inv_index,
                     res_scale, scale_mult, d_packed, d_sq_sum, d_res, d_scale_logit, d_colour_logit);
  return check_hip(hipGetLastError(), "decode_pack_bwd_kernel");
}

}  // extern "C"
