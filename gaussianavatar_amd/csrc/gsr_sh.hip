// gsr_sh.hip — view-dependent colours from spherical harmonics (the `shs` input of the rasterizer
// API). The avatar path never takes this branch (the
// reference always passes colors_precomp: /root/reference/model/avatar_model.py:350-351,
// gaussian_renderer/__init__.py:40-47), so it is kept out of K1/K7: two small per-Gaussian
// kernels that run right after K1 (colour -> ws.rgb, clamp flags -> ws.clamped) and right after
// K7 (dL/dsh, and the view-direction term added to dL/dmeans3D).
//
// Both kernels evaluate the 16 real SH basis polynomials b_k(x,y,z) of the unit view direction and,
// in backward, their gradients; colour_c = max(0, sum_k b_k sh[k][c] + 0.5).
#include "gsr_common.h"

namespace gsr {

namespace {

constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
constexpr float kC2a = 1.0925484305920792f, kC2b = 0.31539156525252005f, kC2c = 0.5462742152960396f;
constexpr float kC3a = 0.5900435899266435f, kC3b = 2.890611442640554f, kC3c = 0.4570457994644658f,
                kC3d = 0.3731763325901154f, kC3e = 1.445305721320277f;

// b[k] for k < (deg+1)^2; when GRAD also db[k][0..2] = d b_k / d(x,y,z).
template <bool GRAD>
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* b, float (*db)[3]) {
  auto set = [&](int k, float v, float gx, float gy, float gz) {
    b[k] = v;
    if (GRAD) { db[k][0] = gx; db[k][1] = gy; db[k][2] = gz; }
  };
  set(0, kC0, 0.f, 0.f, 0.f);
  if (deg < 1) return;
  set(1, -kC1 * y, 0.f, -kC1, 0.f);
  set(2, kC1 * z, 0.f, 0.f, kC1);
  set(3, -kC1 * x, -kC1, 0.f, 0.f);
  if (deg < 2) return;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  set(4, kC2a * xy, kC2a * y, kC2a * x, 0.f);
  set(5, -kC2a * yz, 0.f, -kC2a * z, -kC2a * y);
  set(6, kC2b * (2.f * zz - xx - yy), -2.f * kC2b * x, -2.f * kC2b * y, 4.f * kC2b * z);
  set(7, -kC2a * xz, -kC2a * z, 0.f, -kC2a * x);
  set(8, kC2c * (xx - yy), 2.f * kC2c * x, -2.f * kC2c * y, 0.f);
  if (deg < 3) return;
  set(9, -kC3a * y * (3.f * xx - yy), -kC3a * 6.f * xy, -kC3a * (3.f * xx - 3.f * yy), 0.f);
  set(10, kC3b * xy * z, kC3b * yz, kC3b * xz, kC3b * xy);
  set(11, -kC3c * y * (4.f * zz - xx - yy), kC3c * 2.f * xy, -kC3c * (4.f * zz - xx - 3.f * yy),
      -kC3c * 8.f * yz);
  set(12, kC3d * z * (2.f * zz - 3.f * xx - 3.f * yy), -kC3d * 6.f * xz, -kC3d * 6.f * yz,
      kC3d * (6.f * zz - 3.f * xx - 3.f * yy));
  set(13, -kC3c * x * (4.f * zz - xx - yy), -kC3c * (4.f * zz - 3.f * xx - yy), kC3c * 2.f * xy,
      -kC3c * 8.f * xz);
  set(14, kC3e * z * (xx - yy), kC3e * 2.f * xz, -kC3e * 2.f * yz, kC3e * (xx - yy));
  set(15, -kC3a * x * (xx - 3.f * yy), -kC3a * (3.f * xx - 3.f * yy), kC3a * 6.f * xy, 0.f);
}

__global__ void __launch_bounds__(256)
sh_color_kernel(int P, int M, int deg, const float* __restrict__ means3D,
                const float* __restrict__ campos, const float* __restrict__ shs, Workspace ws,
                Batch bt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int64_t f = blockIdx.y;
  means3D += f * bt.means; campos += f * bt.campos; shs += f * bt.shs;
  ws = frame_ws(ws, (size_t)f * bt.ws_stride);
  const float dx = means3D[3 * i] - campos[0], dy = means3D[3 * i + 1] - campos[1],
              dz = means3D[3 * i + 2] - campos[2];
  const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  float b[16];
  sh_basis<false>(deg, dx * inv, dy * inv, dz * inv, b, nullptr);
  const int K = (deg + 1) * (deg + 1);
  const float* sh = shs + (size_t)i * M * 3;
  float c[3] = {0.5f, 0.5f, 0.5f};
  for (int k = 0; k < K; ++k) {
    c[0] = fmaf(b[k], sh[3 * k], c[0]);
    c[1] = fmaf(b[k], sh[3 * k + 1], c[1]);
    c[2] = fmaf(b[k], sh[3 * k + 2], c[2]);
  }
  uint8_t* cl = ws.clamped + 4 * (size_t)i;
  cl[0] = c[0] < 0.f; cl[1] = c[1] < 0.f; cl[2] = c[2] < 0.f; cl[3] = 0;
  ws.rgb[i] = make_float4(fmaxf(c[0], 0.f), fmaxf(c[1], 0.f), fmaxf(c[2], 0.f), 0.f);
}

__global__ void __launch_bounds__(256)
sh_bwd_kernel(int P, int M, int deg, const float* __restrict__ means3D,
              const float* __restrict__ campos, const float* __restrict__ shs, Workspace ws,
              float* __restrict__ dL_dsh, float* __restrict__ dL_dmeans3D, Batch bt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int64_t f = blockIdx.y;
  means3D += f * bt.means; campos += f * bt.campos; shs += f * bt.shs;
  ws = frame_ws(ws, (size_t)f * bt.ws_stride);
  if (dL_dsh) dL_dsh += f * (int64_t)P * M * 3;
  if (dL_dmeans3D) dL_dmeans3D += f * (int64_t)P * 3;
  // dL/dcolour as accumulated by K6 (exactly 0 for Gaussians that were not rendered)
  const uint8_t* cl = ws.clamped + 4 * (size_t)i;
  float g[3];
  // (a frame whose pair buffer overflowed yields no gradient: gsr_preprocess.hip, preprocess_bwd_kernel)
  const bool overflowed = ws.status[1] != 0;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    g[c] = (cl[c] || overflowed) ? 0.f : ws.grad_acc[(size_t)i * GSR_GRAD_STRIDE + 6 + c];
  const float dx = means3D[3 * i] - campos[0], dy = means3D[3 * i + 1] - campos[1],
              dz = means3D[3 * i + 2] - campos[2];
  const float sum2 = dx * dx + dy * dy + dz * dz;
  const float inv = 1.0f / sqrtf(sum2);
  float b[16], db[16][3];
  sh_basis<true>(deg, dx * inv, dy * inv, dz * inv, b, db);
  const int K = (deg + 1) * (deg + 1);
  const float* sh = shs + (size_t)i * M * 3;
  float gd[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < M; ++k) {
    float o[3] = {0.f, 0.f, 0.f};
    if (k < K) {
      const float w = sh[3 * k] * g[0] + sh[3 * k + 1] * g[1] + sh[3 * k + 2] * g[2];
      gd[0] = fmaf(db[k][0], w, gd[0]);
      gd[1] = fmaf(db[k][1], w, gd[1]);
      gd[2] = fmaf(db[k][2], w, gd[2]);
      o[0] = b[k] * g[0]; o[1] = b[k] * g[1]; o[2] = b[k] * g[2];
    }
    if (dL_dsh) {
      float* d = dL_dsh + ((size_t)i * M + k) * 3;
      d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
    }
  }
  if (dL_dmeans3D) {
    // through dir = d / |d|:  (g - dir (dir . g)) / |d|
    const float x = dx * inv, y = dy * inv, z = dz * inv;
    const float dot = x * gd[0] + y * gd[1] + z * gd[2];
    dL_dmeans3D[3 * i] += (gd[0] - x * dot) * inv;
    dL_dmeans3D[3 * i + 1] += (gd[1] - y * dot) * inv;
    dL_dmeans3D[3 * i + 2] += (gd[2] - z * dot) * inv;
  }
}

}  // namespace

hipError_t launch_sh_color(const GsrSettings& s, const Dims& d, const float* means3D,
                           const float* shs, int sh_coeffs, const Workspace& ws, const Batch& bt,
                           hipStream_t stream) {
  if (d.P == 0) return hipSuccess;
  hipLaunchKernelGGL(sh_color_kernel, dim3((d.P + 255) / 256, bt.frames), dim3(256), 0, stream, d.P,
                     sh_coeffs, s.sh_degree, means3D, s.campos, shs, ws, bt);
  return hipGetLastError();
}

hipError_t launch_sh_bwd(const GsrSettings& s, const Dims& d, const float* means3D, const float* shs,
                         int sh_coeffs, const Workspace& ws, float* dL_dsh, float* dL_dmeans3D,
                         const Batch& bt, hipStream_t stream) {
  if (d.P == 0) return hipSuccess;
  hipLaunchKernelGGL(sh_bwd_kernel, dim3((d.P + 255) / 256, bt.frames), dim3(256), 0, stream, d.P,
                     sh_coeffs, s.sh_degree, means3D, s.campos, shs, ws, dL_dsh, dL_dmeans3D, bt);
  return hipGetLastError();
}

}  // namespace gsr
