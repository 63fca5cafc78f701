// ganet_ssim.hip — SSIM (11x11 Gaussian window, sigma 1.5, zero padding) forward and backward as
// separable streaming passes.
//
// Reference: /root/reference/utils/loss_utils.py:23-53 — five grouped 11x11 convolutions
// (mu1, mu2, E[x1^2], E[x2^2], E[x1 x2]) plus ~15 element-wise kernels over [B,3,H,W], and
// their autograd counterparts. Here a WAVE owns a vertical strip of 64 columns x STRIP rows of one
// image plane and streams down its rows (no workgroup barriers): a row (+5 columns of halo each side)
// goes through a per-wave LDS line, every lane takes the 11 horizontal taps of its column, and the
// vertical pass is a ring of 11 partially accumulated output rows held in registers (each new input
// row is scattered into the 11 output rows it contributes to; the oldest one is then complete).
// The L1 loss between the same two images rides along (forward: |x1 - x2| of the strip's own pixels;
// backward: its sign term is added to the SSIM gradient), so the loop's two image losses cost one pass.
// Forward evaluates the SSIM map and — because the loss is always differentiated — the three partial
// derivatives dS/dmu1, dS/dE[x1^2], dS/dE[x1 x2] in the same pass; backward convolves those three maps
// with the (symmetric) window and combines
//     dL/dx1 = scale * ( conv(dS/dmu1) + 2 x1 conv(dS/dE11) + x2 conv(dS/dE12) ).
#include <cmath>

#include "ganet.h"
#include "ganet_common.h"

namespace ganet {

namespace {

constexpr int R = 5, WIN = 11;
#ifndef GANET_SSIM_STRIP
#define GANET_SSIM_STRIP 34
#endif
constexpr int STRIP = GANET_SSIM_STRIP;   // output rows per wave (STRIP + 10 input rows = whole rounds of 11)
constexpr int WAVES = 4;                  // strips stacked in a workgroup
constexpr int LINE = 64 + 2 * R + 6;      // LDS line per input map (padded to 80 floats)
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

struct Window { float w[WIN]; };

Window make_window() {
  Window k;
  double s = 0.0, v[WIN];
  for (int i = 0; i < WIN; ++i) { v[i] = exp(-(double)((i - R) * (i - R)) / (2.0 * 1.5 * 1.5)); s += v[i]; }
  for (int i = 0; i < WIN; ++i) k.w[i] = (float)(v[i] / s);
  return k;
}

// Streams the rows ys-5 .. ys+STRIP+4 of NIN input maps through the wave. `horiz(taps, h, own)` turns
// the 11 taps of every input map at this lane's column into NQ horizontally filtered values (`own`: the
// input row belongs to this wave's strip, i.e. taps[.][5] is a pixel no other wave visits as a centre);
// `emit(y, v)` receives the NQ fully filtered values of output row y (only rows of the strip that exist).
template <int NIN, int NQ, class Horiz, class Emit>
__device__ __forceinline__ void stream_strip(const float* const* src, int H, int W, int x0, int ys,
                                             float* line, const Window& k, Horiz horiz, Emit emit) {
  const int lane = threadIdx.x & 63;
  float acc[WIN][NQ];
#pragma unroll
  for (int j = 0; j < WIN; ++j)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[j][q] = 0.f;
  const int gx0 = x0 - R + lane;                 // first column this lane stages
  const int gx1 = x0 + 64 - R + lane;            // lanes 0..9: the right halo
  const bool c0 = gx0 >= 0 && gx0 < W, c1 = lane < 2 * R && gx1 < W;
  // the next row's values are fetched into registers while the current row is being filtered
  float nv0[NIN], nv1[NIN];
  auto fetch = [&](int y) {
    const bool rowok = y >= 0 && y < H;
#pragma unroll
    for (int q = 0; q < NIN; ++q) {
      const float* row = src[q] + (size_t)(rowok ? y : 0) * W;
      nv0[q] = (rowok && c0) ? row[gx0] : 0.f;
      nv1[q] = (rowok && c1) ? row[gx1] : 0.f;
    }
  };
  fetch(ys - R);
  for (int base = -R; base < STRIP + R; base += WIN) {
#pragma unroll
    for (int j = 0; j < WIN; ++j) {              // row r = base + j sits in ring slot j
      const int r = base + j;
      const int y = ys + r;
      __builtin_amdgcn_wave_barrier();           // previous row's LDS reads are done (in-order LDS)
#pragma unroll
      for (int q = 0; q < NIN; ++q) {
        line[q * LINE + lane] = nv0[q];
        if (lane < 2 * R) line[q * LINE + 64 + lane] = nv1[q];
      }
      __builtin_amdgcn_wave_barrier();
      fetch(y + 1);
      float taps[NIN][WIN];
#pragma unroll
      for (int q = 0; q < NIN; ++q)
#pragma unroll
        for (int t = 0; t < WIN; ++t) taps[q][t] = line[q * LINE + lane + t];
      float h[NQ];
      horiz(taps, h, r >= 0 && r < STRIP && y < H);
      // scatter into the output rows r-5 .. r+5 (ring slots (j + d) mod 11), weight w[5 - d]
#pragma unroll
      for (int d = -R; d <= R; ++d) {
        const int slot = (j + d + WIN) % WIN;
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[slot][q] = fmaf(k.w[R - d], h[q], acc[slot][q]);
      }
      // output row r-5 is complete
      const int done = (j + WIN - R) % WIN;
      const int yo = y - R;
      if (r - R >= 0 && r - R < STRIP && yo < H) emit(yo, acc[done]);
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[done][q] = 0.f;
    }
  }
}

__global__ void __launch_bounds__(64 * WAVES)
ssim_fwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                float norm, float* __restrict__ sums, float* __restrict__ partials, size_t map_stride,
                Window k) {
  __shared__ float s_line[WAVES][2 * LINE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * 64, ys = (blockIdx.y * WAVES + wave) * STRIP;
  if (ys >= H) return;
  const size_t poff = (size_t)plane * H * W;
  const float* src[2] = {img1 + poff, img2 + poff};
  const int gx = x0 + lane;
  float S = 0.f, L = 0.f;
  auto horiz = [&](const float (*taps)[WIN], float* h, bool own) {
    L += (own && gx < W) ? fabsf(taps[0][R] - taps[1][R]) : 0.f;      // L1 rides along
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      const float a = taps[0][t], b = taps[1][t];
      const float wa = k.w[t] * a, wb = k.w[t] * b;
      m1 += wa; m2 += wb;
      e11 = fmaf(wa, a, e11); e22 = fmaf(wb, b, e22); e12 = fmaf(wa, b, e12);
    }
    h[0] = m1; h[1] = m2; h[2] = e11; h[3] = e22; h[4] = e12;
  };
  auto emit = [&](int y, const float* v) {
    if (gx >= W) return;
    const float m1 = v[0], m2 = v[1], e11 = v[2], e22 = v[3], e12 = v[4];
    const float n1 = 2.f * m1 * m2 + C1;
    const float n2 = 2.f * (e12 - m1 * m2) + C2;
    const float d1 = m1 * m1 + m2 * m2 + C1;
    const float d2 = (e11 - m1 * m1) + (e22 - m2 * m2) + C2;
    const float inv = 1.0f / (d1 * d2);
    const float Sv = n1 * n2 * inv;
    S += Sv;
    const size_t o = poff + (size_t)y * W + gx;
    partials[o] = 2.f * m2 * (n2 - n1) * inv - 2.f * m1 * Sv * (1.0f / d1 - 1.0f / d2);   // dS/dmu1
    partials[map_stride + o] = -Sv / d2;                                                   // dS/dE[x1^2]
    partials[2 * map_stride + o] = 2.f * n1 * inv;                                         // dS/dE[x1 x2]
  };
  stream_strip<2, 5>(src, H, W, x0, ys, s_line[wave], k, horiz, emit);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    S += __shfl_xor(S, off);
    L += __shfl_xor(L, off);
  }
  if (lane < 2) atomicAdd(sums + lane, (lane ? L : S) * norm);
}

__global__ void __launch_bounds__(64 * WAVES)
ssim_bwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                const float* __restrict__ partials, size_t map_stride, float norm,
                const float* __restrict__ d_ssim, const float* __restrict__ d_l1,
                float* __restrict__ dimg1, Window k) {
  __shared__ float s_line[WAVES][3 * LINE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * 64, ys = (blockIdx.y * WAVES + wave) * STRIP;
  if (ys >= H) return;
  const size_t poff = (size_t)plane * H * W;
  const float* src[3] = {partials + poff, partials + map_stride + poff, partials + 2 * map_stride + poff};
  const int gx = x0 + lane;
  const float scale = d_ssim ? norm * d_ssim[0] : 0.f;
  const float l1s = d_l1 ? norm * d_l1[0] : 0.f;
  auto horiz = [&](const float (*taps)[WIN], float* h, bool) {
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      a = fmaf(k.w[t], taps[0][t], a);
      b = fmaf(k.w[t], taps[1][t], b);
      c = fmaf(k.w[t], taps[2][t], c);
    }
    h[0] = a; h[1] = b; h[2] = c;
  };
  auto emit = [&](int y, const float* v) {
    if (gx >= W) return;
    const size_t o = poff + (size_t)y * W + gx;
    const float a = img1[o], b = img2[o];
    const float sg = (a > b) ? l1s : ((a < b) ? -l1s : 0.f);
    dimg1[o] = fmaf(scale, v[0] + 2.f * a * v[1] + b * v[2], sg);
  };
  stream_strip<3, 3>(src, H, W, x0, ys, s_line[wave], k, horiz, emit);
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

int ganet_ssim_fwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2, float norm,
                   float* sums, float* partials, void* stream_) {
  if (planes <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !sums || !partials) {
    set_error("ganet_ssim_fwd: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_hip(hipMemsetAsync(sums, 0, 2 * sizeof(float), stream), "memset ssim sums");
  if (rc) return rc;
  const dim3 grid((W + 63) / 64, (H + STRIP * WAVES - 1) / (STRIP * WAVES), planes);
  ProfScope prof_(K_SSIM_FWD, stream);
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(64 * WAVES), 0, stream, H, W, img1, img2, norm, sums,
                     partials, (size_t)planes * H * W, make_window());
  return check_hip(hipGetLastError(), "ssim_fwd_kernel");
}

int ganet_ssim_bwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2,
                   const float* partials, float norm, const float* d_ssim, const float* d_l1,
                   float* dimg1, void* stream_) {
  if (planes <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !partials || !dimg1) {
    set_error("ganet_ssim_bwd: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 grid((W + 63) / 64, (H + STRIP * WAVES - 1) / (STRIP * WAVES), planes);
  ProfScope prof_(K_SSIM_BWD, stream);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(64 * WAVES), 0, stream, H, W, img1, img2, partials,
                     (size_t)planes * H * W, norm, d_ssim, d_l1, dimg1, make_window());
  return check_hip(hipGetLastError(), "ssim_bwd_kernel");
}

}  // extern "C"
