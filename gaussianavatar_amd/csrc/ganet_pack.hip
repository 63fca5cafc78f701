// ganet_pack.hip — glue between the decoder heads and the per-Gaussian pipeline, fused.
//
// The reference does, on the decoder's three outputs ([B,3,HW], [B,1,HW], [B,3,HW]):
//   pred_res * 0.02, sigmoid heads, scale warm-up, permute, boolean-mask gather of the valid texels,
//   repeat of the scale to 3 channels, and mean(pred_res^2) for the offset regulariser
// (/root/reference/model/avatar_model.py:298-324, model/network.py:69-81) — about twenty element-wise
// launches with their autograd mirrors. Here: one forward kernel that reads the heads' logits and
// writes the per-Gaussian residual / scale / colour arrays (three segments of one buffer) plus the two
// regulariser means (squared residuals over all texels, scales over the valid ones), and one backward
// kernel that writes the three logit gradients in full. Also here: the geometry-feature regulariser
// mean(x^2) and the weighted sum that composes the scalar objective — the zero-dimensional glue of the
// training loop as single launches.
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"

namespace ganet {

namespace {

__device__ __forceinline__ float sigmoid_f(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v * 1.4426950408889634f));
}

// out: residual [frames,N,3] | scale [frames,N] | colour [frames,N,3] — three contiguous segments of
// one buffer (one all-reduce in the data-parallel exchange, contiguous views for the consumers).
// sums[0] += sq_norm * sum over ALL texels of (res_scale * res)^2 ; sums[1] += scale_norm * sum of the
// valid texels' scales.
__global__ void __launch_bounds__(256)
decode_pack_fwd_kernel(int64_t HW, int64_t N, const float* __restrict__ res,
                       const float* __restrict__ scale_logit, const float* __restrict__ colour_logit,
                       const int64_t* __restrict__ valid_index, float res_scale, float scale_mult,
                       float sq_norm, float scale_norm, float* __restrict__ out, float* __restrict__ sums) {
  const int64_t f = blockIdx.y, F = gridDim.y;
  res += f * HW * 3; scale_logit += f * HW; colour_logit += f * HW * 3;
  float* o_res = out + f * N * 3;
  float* o_scale = out + F * N * 3 + f * N;
  float* o_col = out + F * N * 4 + f * N * 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // Few workgroups — each ends in two atomics on the same line, ~15 ns apiece: 768 of them were most of this kernel's
  // 15 us — and in exchange four texels per thread and step, their gathers issued together.
  float sc = 0.f;
  for (int64_t n0 = t0; n0 < N; n0 += 4 * stride) {
    int64_t m[4];
    float r[4][3], sl[4], cl[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) m[u] = valid_index[min(n0 + u * stride, N - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { r[u][c] = res[3 * m[u] + c]; cl[u][c] = colour_logit[3 * m[u] + c]; }
      sl[u] = scale_logit[m[u]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t n = n0 + u * stride;
      if (n >= N) break;
#pragma unroll
      for (int c = 0; c < 3; ++c) { o_res[3 * n + c] = r[u][c] * res_scale; o_col[3 * n + c] = sigmoid_f(cl[u][c]); }
      const float sv = sigmoid_f(sl[u]) * scale_mult;
      o_scale[n] = sv;
      sc += sv;
    }
  }
  // sum over ALL texels of (res_scale * res)^2
  float s = 0.f;
  for (int64_t i = t0; i < HW * 3; i += 4 * stride) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = res[min(i + u * stride, HW * 3 - 1)] * res_scale;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < HW * 3) s = fmaf(v[u], v[u], s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    sc += __shfl_xor(sc, o);
  }
  __shared__ float s_red[2][4];
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = s; s_red[1][threadIdx.x >> 6] = sc; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float* r = s_red[threadIdx.x];
    unsafeAtomicAdd(sums + threadIdx.x, ((r[0] + r[1]) + (r[2] + r[3])) * (threadIdx.x ? scale_norm : sq_norm));
  }
}

__global__ void __launch_bounds__(256)
decode_pack_bwd_kernel(int64_t HW, int64_t N, const float* __restrict__ res,
                       const float* __restrict__ scale_logit, const float* __restrict__ colour_logit,
                       const int64_t* __restrict__ inv_index, float res_scale, float scale_mult,
                       float sq_norm, float scale_norm, const float* __restrict__ d_out,
                       const float* __restrict__ d_sq, const float* __restrict__ d_scale_sum,
                       float* __restrict__ d_res, float* __restrict__ d_scale_logit,
                       float* __restrict__ d_colour_logit) {
  const int64_t f = blockIdx.y, F = gridDim.y;
  res += f * HW * 3; scale_logit += f * HW; colour_logit += f * HW * 3;
  const float* g_res = d_out ? d_out + f * N * 3 : nullptr;
  const float* g_scale = d_out ? d_out + F * N * 3 + f * N : nullptr;
  const float* g_col = d_out ? d_out + F * N * 4 + f * N * 3 : nullptr;
  d_res += f * HW * 3; d_scale_logit += f * HW; d_colour_logit += f * HW * 3;
  const float k = d_sq ? 2.0f * res_scale * res_scale * sq_norm * d_sq[0] : 0.f;
  const float ks = d_scale_sum ? scale_norm * d_scale_sum[0] : 0.f;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < HW;
       m += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = inv_index[m];
    float g[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n >= 0) {
      if (g_res) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { g[c] = g_res[3 * n + c]; g[4 + c] = g_col[3 * n + c]; }
        g[3] = g_scale[n];
      }
      g[3] += ks;
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

// Gradient of the packed record buffer from the gradients of its expanded views (avatar_model._decode: the stage-1
// decoder is evaluated once, its residual / scale / colour records are broadcast over the B frames and the scalar scale
// over three axes): d_flat[(f, n)] = sum over the frames that share record f (all B when b = 1) — and, for the scale,
// over its three copies. One launch instead of autograd's four expand-backward sums, a cat and its fills.
__global__ void __launch_bounds__(256)
records_bwd_kernel(int b, int B, int64_t N, const float* __restrict__ g_res, const float* __restrict__ g_scale3,
                   const float* __restrict__ g_col, float* __restrict__ d_flat) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (record frame f, n)
  if (i >= (int64_t)b * N) return;
  const int64_t f = i / N, n = i - f * N;
  const int f0 = b == 1 ? 0 : (int)f, f1 = b == 1 ? B : (int)f + 1;
  float r[3] = {0.f, 0.f, 0.f}, c[3] = {0.f, 0.f, 0.f}, sc = 0.f;
  for (int q = f0; q < f1; ++q) {
    const int64_t o = ((int64_t)q * N + n) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (g_res) r[k] += g_res[o + k];
      if (g_col) c[k] += g_col[o + k];
      if (g_scale3) sc += g_scale3[o + k];
    }
  }
  float* o_res = d_flat + i * 3;
  float* o_col = d_flat + (int64_t)b * N * 4 + i * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) { o_res[k] = r[k]; o_col[k] = c[k]; }
  d_flat[(int64_t)b * N * 3 + i] = sc;
}

// out[0] += norm * sum x^2   (geometry-feature regulariser, /root/reference/model/avatar_model.py:367)
__global__ void __launch_bounds__(256)
mean_sq_fwd_kernel(int64_t n, const float* __restrict__ x, float norm, float* __restrict__ out) {
  float s = 0.f;
  const int64_t n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  // few workgroups (every one ends in an atomic on the same address: ~15 ns each, 1024 of them were the kernel's
  // 15 us), four independent 16-byte loads per thread and step
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x4[min(i + u * stride, n4 - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < n4) s += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[4 * n4 + threadIdx.x]; s = fmaf(v, v, s); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float s_red[4];
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * norm);
}

__global__ void __launch_bounds__(256)
mean_sq_bwd_kernel(int64_t n, const float* __restrict__ x, float norm, const float* __restrict__ d_out,
                   float* __restrict__ dx) {
  const float k = 2.0f * norm * d_out[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = k * x[i];
}

// The scalar objective of the training loop as ONE launch: out = bias + sum_i w_i * term_i
// (/root/reference/train.py:70-82 composes it with ~9 zero-dimensional kernels forward, ~7 backward).
struct Terms {
  const float* p[GANET_MAX_TERMS];
  float w[GANET_MAX_TERMS];
};

__global__ void weighted_sum_kernel(int n, Terms t, float bias, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    float s = bias;
    for (int i = 0; i < n; ++i) s = fmaf(t.w[i], t.p[i][0], s);
    out[0] = s;
  }
}

__global__ void weighted_sum_bwd_kernel(int n, Terms t, const float* __restrict__ d_out,
                                        float* __restrict__ d_terms) {
  if ((int)threadIdx.x < n) d_terms[threadIdx.x] = t.w[threadIdx.x] * d_out[0];
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

int ganet_decode_pack_fwd(int32_t frames, int64_t HW, int64_t N, const float* res,
                          const float* scale_logit, const float* colour_logit,
                          const int64_t* valid_index, float res_scale, float scale_mult, float sq_norm,
                          float scale_norm, float* out, float* sums, void* stream_) {
  if (frames <= 0 || HW <= 0 || N < 0 || N > HW || !res || !scale_logit || !colour_logit ||
      (N > 0 && (!valid_index || !out)) || !sums) {
    set_error("ganet_decode_pack_fwd: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_hip(hipMemsetAsync(sums, 0, 2 * sizeof(float), stream), "memset sums");
  if (rc) return rc;
  const int blocks = (int)((HW * 3 / 16 + 255) / 256 < 192 ? (HW * 3 / 16 + 255) / 256 : 192);
  hipLaunchKernelGGL(decode_pack_fwd_kernel, dim3(blocks > 0 ? blocks : 1, frames), dim3(256), 0, stream, HW, N,
                     res, scale_logit, colour_logit, valid_index, res_scale, scale_mult, sq_norm, scale_norm,
                     out, sums);
  return check_hip(hipGetLastError(), "decode_pack_fwd_kernel");
}

int ganet_decode_pack_bwd(int32_t frames, int64_t HW, int64_t N, const float* res,
                          const float* scale_logit, const float* colour_logit,
                          const int64_t* inv_index, float res_scale, float scale_mult, float sq_norm,
                          float scale_norm, const float* d_out, const float* d_sq,
                          const float* d_scale_sum, float* d_res, float* d_scale_logit,
                          float* d_colour_logit, void* stream_) {
  if (frames <= 0 || HW <= 0 || N < 0 || !res || !scale_logit || !colour_logit || !inv_index ||
      !d_res || !d_scale_logit || !d_colour_logit) {
    set_error("ganet_decode_pack_bwd: invalid arguments");
    return 1;
  }
  const int blocks = (int)((HW + 255) / 256 < 4096 ? (HW + 255) / 256 : 4096);
  hipLaunchKernelGGL(decode_pack_bwd_kernel, dim3(blocks, frames), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), HW, N, res, scale_logit, colour_logit, inv_index,
                     res_scale, scale_mult, sq_norm, scale_norm, d_out, d_sq, d_scale_sum, d_res,
                     d_scale_logit, d_colour_logit);
  return check_hip(hipGetLastError(), "decode_pack_bwd_kernel");
}

int ganet_records_bwd(int32_t b, int32_t B, int64_t N, const float* g_res, const float* g_scale3, const float* g_col,
                      float* d_flat, void* stream_) {
  if (b <= 0 || B < b || (b != 1 && b != B) || N <= 0 || !d_flat) {
    set_error("ganet_records_bwd: invalid arguments (b = 1 or b = B)");
    return 1;
  }
  const int64_t n = (int64_t)b * N;
  hipLaunchKernelGGL(records_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     b, B, N, g_res, g_scale3, g_col, d_flat);
  return check_hip(hipGetLastError(), "records_bwd_kernel");
}

int ganet_mean_sq_fwd(int64_t n, const float* x, float norm, float* out, void* stream_) {
  if (n <= 0 || !x || !out || (reinterpret_cast<uintptr_t>(x) & 15)) {
    set_error("ganet_mean_sq_fwd: invalid arguments (x must be 16-byte aligned)");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_hip(hipMemsetAsync(out, 0, sizeof(float), stream), "memset mean_sq");
  if (rc) return rc;
  const int blocks = (int)((n / 16 + 255) / 256 < 64 ? (n / 16 + 255) / 256 : 64);
  hipLaunchKernelGGL(mean_sq_fwd_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, n, x, norm, out);
  return check_hip(hipGetLastError(), "mean_sq_fwd_kernel");
}

int ganet_mean_sq_bwd(int64_t n, const float* x, float norm, const float* d_out, float* dx, void* stream_) {
  if (n <= 0 || !x || !d_out || !dx) {
    set_error("ganet_mean_sq_bwd: invalid arguments");
    return 1;
  }
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(mean_sq_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), n, x,
                     norm, d_out, dx);
  return check_hip(hipGetLastError(), "mean_sq_bwd_kernel");
}

int ganet_weighted_sum_fwd(int32_t n, const float* const* terms, const float* weights, float bias,
                           float* out, void* stream_) {
  if (n <= 0 || n > GANET_MAX_TERMS || !terms || !weights || !out) {
    set_error("ganet_weighted_sum_fwd: invalid arguments (1 <= n <= %d)", GANET_MAX_TERMS);
    return 1;
  }
  Terms t{};
  for (int i = 0; i < n; ++i) {
    if (!terms[i]) { set_error("ganet_weighted_sum_fwd: term %d is NULL", i); return 1; }
    t.p[i] = terms[i];
    t.w[i] = weights[i];
  }
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_), n, t, bias, out);
  return check_hip(hipGetLastError(), "weighted_sum_kernel");
}

int ganet_weighted_sum_bwd(int32_t n, const float* weights, const float* d_out, float* d_terms,
                           void* stream_) {
  if (n <= 0 || n > GANET_MAX_TERMS || !weights || !d_out || !d_terms) {
    set_error("ganet_weighted_sum_bwd: invalid arguments");
    return 1;
  }
  Terms t{};
  for (int i = 0; i < n; ++i) t.w[i] = weights[i];
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_), n, t,
                     d_out, d_terms);
  return check_hip(hipGetLastError(), "weighted_sum_bwd_kernel");
}

}  // extern "C"
