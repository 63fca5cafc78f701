// ganet_optim.hip — the Adam update of every parameter tensor of the iteration in ONE launch.
//
// The reference steps torch.optim.Adam over the net's ~55 small tensors (+ the geometry feature map)
// (/root/reference/model/avatar_model.py:152-161,264-266); torch's multi-tensor implementation needs
// three launches of ~32 us for these 3 M elements. Here a table of (param, grad, exp_avg, exp_avg_sq,
// size, step size, bias correction) rows travels as the kernel argument; a workgroup owns one
// 4096-element chunk and finds its tensor with a scan over the table's chunk prefix (scalar loads).
// Update rule (torch.optim.Adam, amsgrad off, weight decay off, maximize off):
//   m = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g^2 ;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"

namespace ganet {

namespace {

constexpr int CHUNK = 4096;

struct AdamTable {
  float* p[GANET_MAX_ADAM_TENSORS];
  const float* g[GANET_MAX_ADAM_TENSORS];
  float* m[GANET_MAX_ADAM_TENSORS];
  float* v[GANET_MAX_ADAM_TENSORS];
  int chunk_end[GANET_MAX_ADAM_TENSORS];     // exclusive prefix of chunks, per tensor
  int n[GANET_MAX_ADAM_TENSORS];
  float step_size[GANET_MAX_ADAM_TENSORS];   // lr / bias_correction1
  float inv_sqrt_bc2[GANET_MAX_ADAM_TENSORS];
};

__global__ void __launch_bounds__(256)
adam_kernel(int nt, AdamTable t, float beta1, float beta2, float eps, const int32_t* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag != 0) return;                // uniform: one scalar load per workgroup
  const int c = blockIdx.x;
  int k = 0;
  while (k < nt - 1 && c >= t.chunk_end[k]) ++k;          // uniform: scalar loads and branches
  const int first = k ? t.chunk_end[k - 1] : 0;
  const int base = (c - first) * CHUNK;
  const int n = t.n[k];
  float* __restrict__ p = t.p[k];
  const float* __restrict__ g = t.g[k];
  float* __restrict__ m = t.m[k];
  float* __restrict__ v = t.v[k];
  const float ss = t.step_size[k], isb = t.inv_sqrt_bc2[k];
  const int end = min(base + CHUNK, n);
  auto update = [&](float gi, float& mi, float& vi, float& pi) {
    mi = fmaf(beta1, mi, (1.0f - beta1) * gi);
    vi = fmaf(beta2, vi, (1.0f - beta2) * gi * gi);
    pi -= ss * (mi / (sqrtf(vi) * isb + eps));
  };
  const bool wide = end - base == CHUNK && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                                              reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  if (wide) {      // a whole chunk of 16-byte aligned tensors: four 16-byte loads per array and thread, all issued first
    float4 G[4], Mv[4], V[4], P[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 4 * (threadIdx.x + 256 * u);
      G[u] = *reinterpret_cast<const float4*>(g + i);
      Mv[u] = *reinterpret_cast<const float4*>(m + i);
      V[u] = *reinterpret_cast<const float4*>(v + i);
      P[u] = *reinterpret_cast<const float4*>(p + i);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 4 * (threadIdx.x + 256 * u);
      update(G[u].x, Mv[u].x, V[u].x, P[u].x); update(G[u].y, Mv[u].y, V[u].y, P[u].y);
      update(G[u].z, Mv[u].z, V[u].z, P[u].z); update(G[u].w, Mv[u].w, V[u].w, P[u].w);
      *reinterpret_cast<float4*>(m + i) = Mv[u];
      *reinterpret_cast<float4*>(v + i) = V[u];
      *reinterpret_cast<float4*>(p + i) = P[u];
    }
    return;
  }
#pragma unroll 4
  for (int i = base + threadIdx.x; i < end; i += 256) {
    float mi = m[i], vi = v[i], pi = p[i];
    update(g[i], mi, vi, pi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

int ganet_adam_step(int32_t n_tensors, const GanetAdamTensor* tensors, float beta1, float beta2, float eps,
                    const int32_t* skip_flag, void* stream_) {
  if (n_tensors <= 0 || n_tensors > GANET_MAX_ADAM_TENSORS || !tensors) {
    set_error("ganet_adam_step: invalid arguments (1 <= n_tensors <= %d)", GANET_MAX_ADAM_TENSORS);
    return 1;
  }
  AdamTable t{};
  int chunks = 0;
  for (int k = 0; k < n_tensors; ++k) {
    const GanetAdamTensor& q = tensors[k];
    if (!q.param || !q.grad || !q.exp_avg || !q.exp_avg_sq || q.n <= 0 || q.n > (int64_t)1 << 30 ||
        !(q.bias_correction1 > 0.f) || !(q.bias_correction2 > 0.f)) {
      set_error("ganet_adam_step: tensor %d invalid", k);
      return 1;
    }
    t.p[k] = q.param; t.g[k] = q.grad; t.m[k] = q.exp_avg; t.v[k] = q.exp_avg_sq;
    t.n[k] = (int)q.n;
    chunks += (int)((q.n + CHUNK - 1) / CHUNK);
    t.chunk_end[k] = chunks;
    t.step_size[k] = q.lr / q.bias_correction1;
    t.inv_sqrt_bc2[k] = 1.0f / sqrtf(q.bias_correction2);
  }
  hipLaunchKernelGGL(adam_kernel, dim3(chunks), dim3(256), 0, static_cast<hipStream_t>(stream_), n_tensors, t,
                     beta1, beta2, eps, skip_flag);
  return check_hip(hipGetLastError(), "adam_kernel");
}

int ganet_flag_clear(int32_t* flag, void* stream_) {
  if (!flag) { set_error("ganet_flag_clear: flag is NULL"); return 1; }
  return check_hip(hipMemsetAsync(flag, 0, sizeof(int32_t), static_cast<hipStream_t>(stream_)), "hipMemsetAsync");
}

}  // extern "C"
