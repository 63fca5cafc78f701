// ganet_bn.hip — training-mode BatchNorm over the rows of a point-major activation matrix,
// fused with the softplus that follows it in every hidden layer of the decoder
// (/root/reference/model/modules.py:535-548,554-575: actv_fn(bn_k(conv_k(.)))).
//
// HBM traffic per layer (x, y: [M,C] fp32, M = 262,144, C = 128 -> 134 MB each):
//   forward   statistics pass (read x) + apply pass (read x, write y)            = 3 tensors
//   backward  reduction pass (read x, dy) + apply pass (read x, dy, write dx)    = 5 tensors
// versus ~7 / ~10 tensor passes through separate batch_norm, softplus and their backward
// kernels in eager torch. Statistics use per-thread sums over short row runs combined with
// Chan's parallel (count, mean, M2) update, so there is no E[x^2]-E[x]^2 cancellation.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "ganet.h"
#include "ganet_common.h"

namespace ganet {

namespace {

thread_local char g_err[512] = "";

constexpr int WG = 256;
constexpr int MAX_C = 256;
constexpr int MAX_BLOCKS = 1024;

struct Plan { int nblocks; int64_t rows_per_block; int lanes_per_row; int rows_per_step; };

Plan make_plan(int64_t M, int C) {
  Plan p;
  p.lanes_per_row = C / 4;                       // each thread owns 4 consecutive channels
  p.rows_per_step = WG / p.lanes_per_row;        // rows covered by the workgroup per step
  int64_t rpb = (M + MAX_BLOCKS - 1) / MAX_BLOCKS;
  rpb = ((rpb + p.rows_per_step - 1) / p.rows_per_step) * p.rows_per_step;
  if (rpb < p.rows_per_step) rpb = p.rows_per_step;
  p.rows_per_block = rpb;
  p.nblocks = (int)((M + rpb - 1) / rpb);
  return p;
}

// softplus(u) = log1p(exp(u)) (torch.nn.Softplus: beta 1, threshold 20). For tiny e = exp(u) the
// series e - e^2/2 keeps full relative precision; elsewhere the hardware log of (1 + e) is within
// 1e-7 absolute.
__device__ __forceinline__ float act_fwd(float u, int act) {
  if (act == 0) return u;
  const float e = __expf(u);
  const float sp = e < 1e-3f ? e * (1.0f - 0.5f * e) : __logf(1.0f + e);
  return u > 20.0f ? u : sp;
}
__device__ __forceinline__ float act_grad(float u, int act) {
  if (act == 0) return 1.0f;
  const float z = __expf(fminf(u, 20.0f));
  return u > 20.0f ? 1.0f : z * __builtin_amdgcn_rcpf(z + 1.0f);
}

// (count, mean, M2) merge
__device__ __forceinline__ void chan(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
  if (nb == 0.f) return;
  const float nn = n + nb;
  const float d = meanb - mean;
  mean += d * (nb / nn);
  m2 += m2b + d * d * (n * nb / nn);
  n = nn;
}

// ---- forward statistics: partial (count, mean, M2) per block and channel
__global__ void __launch_bounds__(WG)
bn_stats_kernel(int64_t M, int C, const float* __restrict__ x, float* __restrict__ part, Plan p) {
  __shared__ float s_n[WG], s_mean[WG][4], s_m2[WG][4];
  const int tid = threadIdx.x;
  const int lane_c = tid % p.lanes_per_row, rgrp = tid / p.lanes_per_row;
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = min(r0 + p.rows_per_block, M);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
  float cnt = 0.f;
  if (rgrp < p.rows_per_step) {
    for (int64_t r = r0 + rgrp; r < r1; r += p.rows_per_step) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * C + lane_c * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
      cnt += 1.f;
    }
  }
  const float inv = cnt > 0.f ? 1.0f / cnt : 0.f;
  float mean[4] = {s.x * inv, s.y * inv, s.z * inv, s.w * inv};
  float m2[4] = {q.x - s.x * mean[0], q.y - s.y * mean[1], q.z - s.z * mean[2], q.w - s.w * mean[3]};
  s_n[tid] = cnt;
#pragma unroll
  for (int k = 0; k < 4; ++k) { s_mean[tid][k] = mean[k]; s_m2[tid][k] = fmaxf(m2[k], 0.f); }
  __syncthreads();
  if (tid < p.lanes_per_row) {
    float n = 0.f, mu[4] = {0.f, 0.f, 0.f, 0.f}, mm[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < p.rows_per_step; ++g) {
      const int t = g * p.lanes_per_row + tid;
      float nk = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        nk = n;
        chan(nk, mu[k], mm[k], s_n[t], s_mean[t][k], s_m2[t][k]);
      }
      n = nk;
    }
    float* o = part + (size_t)blockIdx.x * (1 + 2 * C);
    if (tid == 0) o[0] = n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[1 + tid * 4 + k] = mu[k];
      o[1 + C + tid * 4 + k] = mm[k];
    }
  }
}

// one wave per channel: lanes stride over the per-block partials, then a butterfly of Chan merges
__global__ void __launch_bounds__(64)
bn_stats_final_kernel(int nblocks, int C, float eps, const float* __restrict__ part,
                      float* __restrict__ mean, float* __restrict__ rstd,
                      float* __restrict__ running_mean, float* __restrict__ running_var,
                      float momentum, long long* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  float n = 0.f, mu = 0.f, m2 = 0.f;
  for (int b = lane; b < nblocks; b += 64) {
    const float* o = part + (size_t)b * (1 + 2 * C);
    chan(n, mu, m2, o[0], o[1 + c], o[1 + C + c]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float nb = __shfl_xor(n, off), mub = __shfl_xor(mu, off), m2b = __shfl_xor(m2, off);
    chan(n, mu, m2, nb, mub, m2b);
  }
  if (lane == 0) {
    mean[c] = mu;
    rstd[c] = rsqrtf(m2 / n + eps);       // biased variance, as F.batch_norm in training mode
    // running statistics exactly as torch: unbiased variance, running = (1-m) running + m batch
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (m2 / fmaxf(n - 1.0f, 1.0f));
    if (num_batches_tracked && c == 0) num_batches_tracked[0] += 1;
  }
}

// Element-wise passes: a thread always lands on the same 4 channels (grid stride is a multiple
// of C/4), so the per-channel constants are loaded once.
__global__ void __launch_bounds__(WG)
bn_apply_kernel(int64_t M, int C, const float* __restrict__ x, const float* __restrict__ gamma,
                const float* __restrict__ beta, const float* __restrict__ mean,
                const float* __restrict__ rstd, int act, float* __restrict__ y) {
  const int64_t total4 = M * C / 4;
  const int c4 = C / 4;
  const int64_t first = (int64_t)blockIdx.x * WG + threadIdx.x;
  const int c = (int)(first % c4) * 4;
  float a[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = gamma[c + k] * rstd[c + k];
    b[k] = beta[c + k] - mean[c + k] * a[k];
  }
  for (int64_t i = first; i < total4; i += (int64_t)gridDim.x * WG) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = act_fwd(fmaf(a[0], v.x, b[0]), act);
    o.y = act_fwd(fmaf(a[1], v.y, b[1]), act);
    o.z = act_fwd(fmaf(a[2], v.z, b[2]), act);
    o.w = act_fwd(fmaf(a[3], v.w, b[3]), act);
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

// ---- backward reduction: partial sum(ds), sum(ds * xhat) per block and channel
__global__ void __launch_bounds__(WG)
bn_bwd_reduce_kernel(int64_t M, int C, const float* __restrict__ x, const float* __restrict__ dy,
                     const float* __restrict__ gamma, const float* __restrict__ beta,
                     const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                     float* __restrict__ part, Plan p) {
  __shared__ float s_a[WG][4], s_b[WG][4];
  const int tid = threadIdx.x;
  const int lane_c = tid % p.lanes_per_row, rgrp = tid / p.lanes_per_row;
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = min(r0 + p.rows_per_block, M);
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  if (rgrp < p.rows_per_step) {
    const int c = lane_c * 4;
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mu[k] = mean[c + k]; rs[k] = rstd[c + k]; ga[k] = gamma[c + k]; be[k] = beta[c + k]; }
    for (int64_t r = r0 + rgrp; r < r1; r += p.rows_per_step) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * C + c);
      const float4 g = *reinterpret_cast<const float4*>(dy + r * C + c);
      const float in[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (in[k] - mu[k]) * rs[k];
        const float ds = gg[k] * act_grad(fmaf(ga[k], xh, be[k]), act);
        sa[k] += ds;
        sb[k] += ds * xh;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { s_a[tid][k] = sa[k]; s_b[tid][k] = sb[k]; }
  __syncthreads();
  if (tid < p.lanes_per_row) {
    float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < p.rows_per_step; ++g) {
      const int t = g * p.lanes_per_row + tid;
#pragma unroll
      for (int k = 0; k < 4; ++k) { ta[k] += s_a[t][k]; tb[k] += s_b[t][k]; }
    }
    float* o = part + (size_t)blockIdx.x * (2 * C);
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[tid * 4 + k] = ta[k]; o[C + tid * 4 + k] = tb[k]; }
  }
}

__global__ void __launch_bounds__(64)
bn_bwd_final_kernel(int nblocks, int C, const float* __restrict__ part, float* __restrict__ dgamma,
                    float* __restrict__ dbeta, float* __restrict__ sums) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  float a = 0.f, b = 0.f;
  for (int k = lane; k < nblocks; k += 64) {
    a += part[(size_t)k * 2 * C + c];
    b += part[(size_t)k * 2 * C + C + c];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off);
    b += __shfl_xor(b, off);
  }
  if (lane == 0) {
    sums[c] = a;
    sums[C + c] = b;
    if (dbeta) dbeta[c] = a;
    if (dgamma) dgamma[c] = b;
  }
}

__global__ void __launch_bounds__(WG)
bn_bwd_apply_kernel(int64_t M, int C, const float* __restrict__ x, const float* __restrict__ dy,
                    const float* __restrict__ gamma, const float* __restrict__ beta,
                    const float* __restrict__ mean, const float* __restrict__ rstd,
                    const float* __restrict__ sums, int act, float* __restrict__ dx) {
  const int64_t total4 = M * C / 4;
  const int c4 = C / 4;
  const float invM = 1.0f / (float)M;
  const int64_t first = (int64_t)blockIdx.x * WG + threadIdx.x;
  const int c = (int)(first % c4) * 4;
  float mu[4], rs[4], ga[4], be[4], k0[4], k1[4], k2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mu[k] = mean[c + k]; rs[k] = rstd[c + k]; ga[k] = gamma[c + k]; be[k] = beta[c + k];
    k0[k] = ga[k] * rs[k];                      // dx = k0 * (ds - k1 - xhat * k2)
    k1[k] = sums[c + k] * invM;
    k2[k] = sums[C + c + k] * invM;
  }
  for (int64_t i = first; i < total4; i += (int64_t)gridDim.x * WG) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    const float in[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (in[k] - mu[k]) * rs[k];
      const float ds = gg[k] * act_grad(fmaf(ga[k], xh, be[k]), act);
      o[k] = k0[k] * (ds - k1[k] - xh * k2[k]);
    }
    reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

int elementwise_grid(int64_t total4) {
  int64_t g = (total4 + WG - 1) / WG;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

int check_shape(int64_t M, int C) {
  if (M <= 0 || C <= 0 || C > MAX_C || (C % 4) != 0 || (WG % (C / 4)) != 0) {
    set_error("ganet_bn: unsupported shape M=%lld C=%d (C %% 4 == 0, C <= 256, 256 %% (C/4) == 0)",
              (long long)M, C);
    return 4;
  }
  return 0;
}

}  // namespace

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- opt-in profiler
// A caller-owned object (include/ganet.h: GanetProfile) bound to the calling thread; nothing process-global.
}  // namespace ganet

struct GanetProfile {
  struct Rec { hipEvent_t start, stop; int id; };
  std::mutex mu;                      // read may come from another thread than the one that launches
  std::vector<Rec> recs;              // recorded, not yet read
  std::vector<Rec> free_;             // recycled event pairs
  double ms[ganet::K_COUNT] = {0};
  long long n[ganet::K_COUNT] = {0};
};

namespace ganet {

namespace {
thread_local GanetProfile* t_prof = nullptr;
thread_local unsigned t_prof_mask = 0;     // bit k: time kernel id k
}  // namespace

ProfScope::ProfScope(KernelId id, hipStream_t s) : slot(-1), stream(s) {
  GanetProfile* p = t_prof;
  if (!p || !((t_prof_mask >> id) & 1u)) return;
  std::lock_guard<std::mutex> lk(p->mu);
  GanetProfile::Rec r;
  if (!p->free_.empty()) { r = p->free_.back(); p->free_.pop_back(); }
  else { if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return; }
  r.id = id;
  if (hipEventRecord(r.start, s) != hipSuccess) return;
  p->recs.push_back(r);
  slot = (int)p->recs.size() - 1;
}

ProfScope::~ProfScope() {
  GanetProfile* p = t_prof;
  if (slot < 0 || !p) return;
  std::lock_guard<std::mutex> lk(p->mu);
  if (slot < (int)p->recs.size()) (void)hipEventRecord(p->recs[slot].stop, stream);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(e));
  return 3;
}

}  // namespace ganet

using namespace ganet;

extern "C" {

size_t ganet_bn_workspace(int64_t M, int32_t C) {
  if (check_shape(M, C)) return 0;
  const Plan p = make_plan(M, C);
  return ((size_t)p.nblocks * (1 + 2 * (size_t)C) + 2 * (size_t)C) * sizeof(float);
}

int ganet_bn_act_fwd(int64_t M, int32_t C, const float* x, const float* gamma, const float* beta,
                     float eps, int32_t act, float* y, float* mean, float* rstd,
                     float* running_mean, float* running_var, float momentum,
                     int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes,
                     void* stream_) {
  int rc = check_shape(M, C);
  if (rc) return rc;
  if (!x || !gamma || !beta || !y || !mean || !rstd || !workspace ||
      workspace_bytes < ganet_bn_workspace(M, C)) {
    set_error("ganet_bn_act_fwd: NULL argument or workspace too small");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const Plan p = make_plan(M, C);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(p.nblocks), dim3(WG), 0, stream, M, C, x, part, p);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(64), 0, stream, p.nblocks, C, eps, part,
                     mean, rstd, running_mean, running_var, momentum,
                     reinterpret_cast<long long*>(num_batches_tracked));
  hipLaunchKernelGGL(bn_apply_kernel, dim3(elementwise_grid(M * C / 4)), dim3(WG), 0, stream, M, C, x,
                     gamma, beta, mean, rstd, act, y);
  return check_hip(hipGetLastError(), "ganet_bn_act_fwd");
}

int ganet_bn_act_bwd(int64_t M, int32_t C, const float* x, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, int32_t act, const float* dy, float* dx,
                     float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                     void* stream_) {
  int rc = check_shape(M, C);
  if (rc) return rc;
  if (!x || !gamma || !beta || !mean || !rstd || !dy || !dx || !workspace ||
      workspace_bytes < ganet_bn_workspace(M, C)) {
    set_error("ganet_bn_act_bwd: NULL argument or workspace too small");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const Plan p = make_plan(M, C);
  float* part = static_cast<float*>(workspace);
  float* sums = part + (size_t)p.nblocks * (1 + 2 * (size_t)C);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(p.nblocks), dim3(WG), 0, stream, M, C, x, dy, gamma,
                     beta, mean, rstd, act, part, p);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(64), 0, stream, p.nblocks, C, part, dgamma,
                     dbeta, sums);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(elementwise_grid(M * C / 4)), dim3(WG), 0, stream, M, C,
                     x, dy, gamma, beta, mean, rstd, sums, act, dx);
  return check_hip(hipGetLastError(), "ganet_bn_act_bwd");
}

GanetProfile* ganet_profile_create(void) { return new (std::nothrow) GanetProfile(); }

void ganet_profile_destroy(GanetProfile* p) {
  if (!p) return;
  if (t_prof == p) { t_prof = nullptr; t_prof_mask = 0; }
  for (auto* v : {&p->recs, &p->free_})
    for (auto& r : *v) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  delete p;
}

int ganet_profile_bind(GanetProfile* p, int mask) {
  t_prof = (p && mask) ? p : nullptr;
  t_prof_mask = p ? (unsigned)mask : 0u;
  return 0;
}

int ganet_profile_count(void) { return K_COUNT; }

int ganet_profile_read(GanetProfile* p, double* ms_sum, int64_t* launches, int reset) {
  if (!p) { set_error("ganet_profile_read: profile is NULL"); return 1; }
  std::lock_guard<std::mutex> lk(p->mu);
  for (auto& r : p->recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.stop) == hipSuccess &&
        hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      p->ms[r.id] += ms;
      p->n[r.id] += 1;
    }
    p->free_.push_back(r);
  }
  p->recs.clear();
  for (int k = 0; k < K_COUNT; ++k) {
    if (ms_sum) ms_sum[k] = p->ms[k];
    if (launches) launches[k] = p->n[k];
    if (reset) { p->ms[k] = 0; p->n[k] = 0; }
  }
  return 0;
}

const char* ganet_profile_kernel_name(int id) {
  static const char* names[K_COUNT] = {"mlp_fwd", "mlp_stats", "wgrad_act", "wgrad_reduce",
                                       "mlp_bwd_data", "head_bwd", "bwd_stats", "ssim_fwd", "ssim_bwd",
                                       "layer_bwd"};
  return (id >= 0 && id < K_COUNT) ? names[id] : "";
}

const char* ganet_last_error(void) { return g_err; }
int ganet_abi_version(void) { return GANET_ABI_VERSION; }

}  // extern "C"
