// ganet_upsample.hip — bilinear up-sampling of the geometry/pose feature map at the (separable) UV
// texel grid, writing the decoder's input rows directly.
//
// Reference: /root/reference/model/network.py:60-66 — F.grid_sample(feature map [B,C,R,R], uv grid
// [B,S,S,2], bilinear, align_corners=False, zero padding), reshape to [B,C,S*S], concatenation with the
// uv coordinates. The query grid of the reference is separable (utils/general_utils.py:165-176: texel
// (i,j) samples row tap(i), column tap(j)), so every output texel has 2 x 2 taps:
//     x[(i,j), c] = sum_{a,b<2} wr[i,a] wc[j,b] feat[pr[i,a], pc[j,b], c]
// One wave per output texel (lane = channel, C = 64, channels-last feature map: coalesced 256-byte reads
// that stay in L2 — the map is 4 MB), the same kernel appends the two uv columns and the zero padding of
// the fused decoder's 72-float rows: one pass over the 75 MB output instead of two dense GEMMs + a cat.
// Backward: two separable gather passes through the transposed tap lists (CSR), no atomics.
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"

namespace ganet {

namespace {

__global__ void __launch_bounds__(256)
upsample_cat_fwd_kernel(int S, int R, const float* __restrict__ feat, const int32_t* __restrict__ row_idx,
                        const float* __restrict__ row_w, const int32_t* __restrict__ col_idx,
                        const float* __restrict__ col_w, const float* __restrict__ uv,
                        float* __restrict__ x, int64_t ldx) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // texel indices stay scalar
  const int64_t f = blockIdx.y;
  feat += f * (int64_t)R * R * 64;
  x += f * (int64_t)S * S * ldx;
  uv += f * (int64_t)S * S * 2;
  // A wave takes 4 consecutive texels of one output row at a time; lane = (texel u = lane >> 4, channel quad
  // c4 = lane & 15): the four taps of a texel are four 16-byte loads per lane and the 64 interpolated channels leave
  // as one 16-byte store per lane (one lane per channel, 16 dword loads and 8 stores per group: 36 us; this: 30.5.
  // Measured without further gain: the rows staged through LDS into 1152-byte contiguous stores, two groups per
  // step with all loads issued first, grids of 2048 .. 16384 workgroups).
  const int u = lane >> 4, c4 = lane & 15;
  const bool wide = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  const int groups_per_row = (S + 3) / 4;
  const int ngroups = S * groups_per_row;
  for (int gidx = blockIdx.x * 4 + wave; gidx < ngroups; gidx += gridDim.x * 4) {
    const int i = gidx / groups_per_row, j0 = (gidx - i * groups_per_row) * 4;
    const int p0 = row_idx[2 * i], p1 = row_idx[2 * i + 1];
    const float a0 = row_w[2 * i], a1 = row_w[2 * i + 1];
    const int j = min(j0 + u, S - 1);
    const int q0 = col_idx[2 * j], q1 = col_idx[2 * j + 1];
    const float b0 = col_w[2 * j], b1 = col_w[2 * j + 1];
    const float4* f0 = reinterpret_cast<const float4*>(feat + (int64_t)p0 * R * 64) + c4;
    const float4* f1 = reinterpret_cast<const float4*>(feat + (int64_t)p1 * R * 64) + c4;
    const float4 v00 = f0[q0 * 16], v01 = f0[q1 * 16], v10 = f1[q0 * 16], v11 = f1[q1 * 16];
    if (j0 + u >= S) continue;
    const int64_t m = (int64_t)i * S + j0 + u;
    // same association as the two-GEMM formulation: columns first, then rows
    float4 o;
    o.x = a0 * (b0 * v00.x + b1 * v01.x) + a1 * (b0 * v10.x + b1 * v11.x);
    o.y = a0 * (b0 * v00.y + b1 * v01.y) + a1 * (b0 * v10.y + b1 * v11.y);
    o.z = a0 * (b0 * v00.z + b1 * v01.z) + a1 * (b0 * v10.z + b1 * v11.z);
    o.w = a0 * (b0 * v00.w + b1 * v01.w) + a1 * (b0 * v10.w + b1 * v11.w);
    float* row = x + m * ldx;
    if (wide) {
      *reinterpret_cast<float4*>(row + 4 * c4) = o;
      // the two uv columns and the zero padding: 16-byte stores by the group's first lanes
      const int tail4 = ((int)ldx - 64) >> 2;
      if (c4 < tail4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 == 0) { t.x = uv[m * 2]; t.y = uv[m * 2 + 1]; }
        *reinterpret_cast<float4*>(row + 64 + 4 * c4) = t;
      }
    } else {
      row[4 * c4] = o.x; row[4 * c4 + 1] = o.y; row[4 * c4 + 2] = o.z; row[4 * c4 + 3] = o.w;
      for (int k = c4; k < (int)ldx - 64; k += 16) row[64 + k] = k < 2 ? uv[m * 2 + k] : 0.f;
    }
  }
}

// Backward in two separable passes (every wave-uniform index is forced scalar so that the tap lists
// come through the scalar cache):
//   pass 1  tmp[i, q, c]   = sum_{j in cols(q)} cw dx[(i, j), c]        one wave per (i, q), reads dx once
//   pass 2  dfeat[p, q, c] = sum_{i in rows(p)} rw tmp[i, q, c]         one wave per (p, q)
__global__ void __launch_bounds__(256)
upsample_bwd_cols_kernel(int S, int R, const float* __restrict__ dx, int64_t ldx,
                         const int32_t* __restrict__ cptr, const int32_t* __restrict__ csrc,
                         const float* __restrict__ cw, float* __restrict__ tmp) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t f = blockIdx.y;
  dx += f * (int64_t)S * S * ldx;
  tmp += f * (int64_t)S * R * 64;
  for (int t = blockIdx.x * 4 + wave; t < S * R; t += gridDim.x * 4) {
    const int i = t / R, q = t - i * R;
    const float* row = dx + (int64_t)i * S * ldx + lane;
    float acc = 0.f;
    const int c0 = cptr[q], c1 = cptr[q + 1];
    // the first 8 taps (all of them for the x4 up-sampling of the reference) as one batch of independent
    // loads — a run-time loop waits for every 256-byte row before it asks for the next one
    float v[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int ci = min(c0 + u, c1 - 1);
      w[u] = (c0 + u < c1) ? cw[ci] : 0.f;
      v[u] = (c1 > c0) ? row[(int64_t)csrc[max(ci, c0)] * ldx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fmaf(w[u], v[u], acc);
    for (int ci = c0 + 8; ci < c1; ++ci) acc = fmaf(cw[ci], row[(int64_t)csrc[ci] * ldx], acc);
    tmp[(int64_t)t * 64 + lane] = acc;
  }
}

__global__ void __launch_bounds__(256)
upsample_bwd_rows_kernel(int S, int R, const float* __restrict__ tmp, const int32_t* __restrict__ rptr,
                         const int32_t* __restrict__ rsrc, const float* __restrict__ rw,
                         float* __restrict__ dfeat) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t f = blockIdx.y;
  tmp += f * (int64_t)S * R * 64;
  dfeat += f * (int64_t)R * R * 64;
  for (int t = blockIdx.x * 4 + wave; t < R * R; t += gridDim.x * 4) {
    const int p = t / R, q = t - p * R;
    float acc = 0.f;
    const int r0 = rptr[p], r1 = rptr[p + 1];
    for (int ri = r0; ri < r1; ++ri) acc = fmaf(rw[ri], tmp[((int64_t)rsrc[ri] * R + q) * 64 + lane], acc);
    dfeat[(int64_t)t * 64 + lane] = acc;
  }
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

int ganet_upsample_cat_fwd(int32_t frames, int32_t S, int32_t R, int32_t C, const float* feat,
                           const int32_t* row_idx, const float* row_w, const int32_t* col_idx,
                           const float* col_w, const float* uv, float* x, int64_t ldx, void* stream_) {
  if (frames <= 0 || S <= 0 || R <= 0 || C != 64 || !feat || !row_idx || !row_w || !col_idx || !col_w ||
      !uv || !x || ldx < 66 || ldx > 128) {
    set_error("ganet_upsample_cat_fwd: invalid arguments (C must be 64, 66 <= ldx <= 128)");
    return 1;
  }
  const int64_t ngroups = (int64_t)S * ((S + 3) / 4);
  const int blocks = (int)((ngroups + 3) / 4 < 8192 ? (ngroups + 3) / 4 : 8192);
  hipLaunchKernelGGL(upsample_cat_fwd_kernel, dim3(blocks, frames), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), S, R, feat, row_idx, row_w, col_idx, col_w, uv, x, ldx);
  return check_hip(hipGetLastError(), "upsample_cat_fwd_kernel");
}

int ganet_upsample_cat_bwd(int32_t frames, int32_t S, int32_t R, int32_t C, const float* dx, int64_t ldx,
                           const int32_t* row_ptr, const int32_t* row_src, const float* row_w,
                           const int32_t* col_ptr, const int32_t* col_src, const float* col_w,
                           float* tmp, float* dfeat, void* stream_) {
  if (frames <= 0 || S <= 0 || R <= 0 || C != 64 || !dx || !row_ptr || !row_src || !row_w || !col_ptr ||
      !col_src || !col_w || !tmp || !dfeat || ldx < 64) {
    set_error("ganet_upsample_cat_bwd: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int b1 = (S * R + 3) / 4 < 16384 ? (S * R + 3) / 4 : 16384;
  hipLaunchKernelGGL(upsample_bwd_cols_kernel, dim3(b1, frames), dim3(256), 0, stream, S, R, dx, ldx, col_ptr,
                     col_src, col_w, tmp);
  const int b2 = (R * R + 3) / 4 < 4096 ? (R * R + 3) / 4 : 4096;
  hipLaunchKernelGGL(upsample_bwd_rows_kernel, dim3(b2, frames), dim3(256), 0, stream, S, R, tmp, row_ptr,
                     row_src, row_w, dfeat);
  return check_hip(hipGetLastError(), "upsample_bwd kernels");
}

}  // extern "C"
