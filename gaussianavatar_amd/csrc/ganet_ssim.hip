// ganet_ssim.hip — SSIM (11x11 Gaussian window, sigma 1.5, zero padding) forward and backward
// as separable, LDS-tiled passes.
//
// Reference: /root/reference/utils/loss_utils.py:23-53 — five grouped 11x11 convolutions
// (mu1, mu2, E[x1^2], E[x2^2], E[x1 x2]) plus ~15 element-wise kernels over [B,3,H,W], and
// their autograd counterparts. Here one workgroup owns a 32x16 output tile of one image plane:
// it stages the (32+10)x(16+10) input patch in LDS, runs the horizontal then the vertical 11-tap
// pass for all five moments, evaluates the SSIM map and — because the loss is always
// differentiated — the three partial derivatives dS/dmu1, dS/dE[x1^2], dS/dE[x1 x2] in the same
// pass. The backward pass convolves those three maps with the (symmetric) window and combines
//     dL/dx1 = scale * ( conv(dS/dmu1) + 2 x1 conv(dS/dE11) + x2 conv(dS/dE12) ).
#include <cmath>

#include "ganet.h"
#include "ganet_common.h"

namespace ganet {

namespace {

constexpr int TX = 32, TY = 16, R = 5, WIN = 11;
constexpr int PX = TX + 2 * R, PY = TY + 2 * R;   // staged patch 42 x 26
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

struct Window { float w[WIN]; };

Window make_window() {
  Window k;
  double s = 0.0, v[WIN];
  for (int i = 0; i < WIN; ++i) { v[i] = exp(-(double)((i - R) * (i - R)) / (2.0 * 1.5 * 1.5)); s += v[i]; }
  for (int i = 0; i < WIN; ++i) k.w[i] = (float)(v[i] / s);
  return k;
}

template <int NQ>
__device__ __forceinline__ void load_patch(const float* const* src, float (*patch)[PY][PX + 1], int H,
                                           int W, int x0, int y0) {
  for (int i = threadIdx.x; i < PX * PY; i += TX * TY) {
    const int py = i / PX, px = i - py * PX;
    const int gx = x0 + px - R, gy = y0 + py - R;
    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
#pragma unroll
    for (int q = 0; q < NQ; ++q) patch[q][py][px] = in ? src[q][(size_t)gy * W + gx] : 0.f;
  }
}

__global__ void __launch_bounds__(TX * TY)
ssim_fwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                float* __restrict__ ssim_sum, float* __restrict__ partials, size_t map_stride,
                Window k) {
  __shared__ float s_in[2][PY][PX + 1];
  __shared__ float s_h[5][PY][TX + 1];
  __shared__ float s_red[TX * TY / 64];
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const size_t poff = (size_t)plane * H * W;
  const float* src[2] = {img1 + poff, img2 + poff};
  load_patch<2>(src, s_in, H, W, x0, y0);
  __syncthreads();
  // horizontal pass: PY rows x TX columns, five moments
  for (int i = threadIdx.x; i < PY * TX; i += TX * TY) {
    const int py = i / TX, tx = i - py * TX;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      const float a = s_in[0][py][tx + t], b = s_in[1][py][tx + t], w = k.w[t];
      m1 = fmaf(w, a, m1); m2 = fmaf(w, b, m2);
      e11 = fmaf(w, a * a, e11); e22 = fmaf(w, b * b, e22); e12 = fmaf(w, a * b, e12);
    }
    s_h[0][py][tx] = m1; s_h[1][py][tx] = m2; s_h[2][py][tx] = e11; s_h[3][py][tx] = e22; s_h[4][py][tx] = e12;
  }
  __syncthreads();
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int gx = x0 + tx, gy = y0 + ty;
  float S = 0.f;
  if (gx < W && gy < H) {
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      const float w = k.w[t];
      m1 = fmaf(w, s_h[0][ty + t][tx], m1); m2 = fmaf(w, s_h[1][ty + t][tx], m2);
      e11 = fmaf(w, s_h[2][ty + t][tx], e11); e22 = fmaf(w, s_h[3][ty + t][tx], e22);
      e12 = fmaf(w, s_h[4][ty + t][tx], e12);
    }
    const float n1 = 2.f * m1 * m2 + C1;
    const float n2 = 2.f * (e12 - m1 * m2) + C2;
    const float d1 = m1 * m1 + m2 * m2 + C1;
    const float d2 = (e11 - m1 * m1) + (e22 - m2 * m2) + C2;
    const float inv = 1.0f / (d1 * d2);
    S = n1 * n2 * inv;
    const size_t o = poff + (size_t)gy * W + gx;
    partials[o] = 2.f * m2 * (n2 - n1) * inv - 2.f * m1 * S * (1.0f / d1 - 1.0f / d2);   // dS/dmu1
    partials[map_stride + o] = -S / d2;                                                    // dS/dE[x1^2]
    partials[2 * map_stride + o] = 2.f * n1 * inv;                                         // dS/dE[x1 x2]
  }
  // block sum of the SSIM map
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) S += __shfl_xor(S, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = S;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TX * TY / 64; ++i) t += s_red[i];
    atomicAdd(ssim_sum, t);
  }
}

__global__ void __launch_bounds__(TX * TY)
ssim_bwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                const float* __restrict__ partials, size_t map_stride,
                const float* __restrict__ scale_dev, float* __restrict__ dimg1, Window k) {
  __shared__ float s_in[3][PY][PX + 1];
  __shared__ float s_h[3][PY][TX + 1];
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const size_t poff = (size_t)plane * H * W;
  const float* src[3] = {partials + poff, partials + map_stride + poff, partials + 2 * map_stride + poff};
  load_patch<3>(src, s_in, H, W, x0, y0);
  __syncthreads();
  for (int i = threadIdx.x; i < PY * TX; i += TX * TY) {
    const int py = i / TX, tx = i - py * TX;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      const float w = k.w[t];
      a = fmaf(w, s_in[0][py][tx + t], a);
      b = fmaf(w, s_in[1][py][tx + t], b);
      c = fmaf(w, s_in[2][py][tx + t], c);
    }
    s_h[0][py][tx] = a; s_h[1][py][tx] = b; s_h[2][py][tx] = c;
  }
  __syncthreads();
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int gx = x0 + tx, gy = y0 + ty;
  if (gx < W && gy < H) {
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      const float w = k.w[t];
      a = fmaf(w, s_h[0][ty + t][tx], a);
      b = fmaf(w, s_h[1][ty + t][tx], b);
      c = fmaf(w, s_h[2][ty + t][tx], c);
    }
    const size_t o = poff + (size_t)gy * W + gx;
    dimg1[o] = scale_dev[0] * (a + 2.f * img1[o] * b + img2[o] * c);
  }
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

int ganet_ssim_fwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2,
                   float* ssim_sum, float* partials, void* stream_) {
  if (planes <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !ssim_sum || !partials) {
    set_error("ganet_ssim_fwd: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_hip(hipMemsetAsync(ssim_sum, 0, sizeof(float), stream), "memset ssim_sum");
  if (rc) return rc;
  const dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY, planes);
  ProfScope prof_(K_SSIM_FWD, stream);
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(TX * TY), 0, stream, H, W, img1, img2, ssim_sum,
                     partials, (size_t)planes * H * W, make_window());
  return check_hip(hipGetLastError(), "ssim_fwd_kernel");
}

int ganet_ssim_bwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2,
                   const float* partials, const float* scale_dev, float* dimg1, void* stream_) {
  if (planes <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !partials || !scale_dev || !dimg1) {
    set_error("ganet_ssim_bwd: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY, planes);
  ProfScope prof_(K_SSIM_BWD, stream);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(TX * TY), 0, stream, H, W, img1, img2, partials,
                     (size_t)planes * H * W, scale_dev, dimg1, make_window());
  return check_hip(hipGetLastError(), "ssim_bwd_kernel");
}

}  // extern "C"
