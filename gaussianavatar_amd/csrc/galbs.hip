// galbs.hip — SMPL joint transforms and fused point skinning for gfx950 (include/galbs.h).
//
// Replaces on the hot path:
//   batch_rodrigues + batch_rigid_transform + "+= transl" + "@ inv_mats"
//       /root/reference/submodules/smplx/lbs.py:299-333,349-405,
//       /root/reference/submodules/smplx/body_models.py:383, /root/reference/model/avatar_model.py:296
//       (dozens of tiny torch kernels and a python loop over joints) -> ONE launch, one wave/frame
//   the two skinning einsums /root/reference/model/avatar_model.py:311-314
//       (materialise pt_mats [B,N,4,4]) -> one fused kernel, blend matrices kept in registers
#include <cstdarg>
#include <cstdio>

#include <hip/hip_runtime.h>

#include "galbs.h"

namespace {

thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(e));
  return 3;
}

constexpr int WAVE = 64;
constexpr int SAVED_PER_JOINT = 21;   // R (9) + G (3x4 = 12)

// ------------------------------------------------------------------ small 3x3 / 3x4 helpers
struct Aff {   // 3x4 affine [R | t], row-major
  float m[12];
};

__device__ __forceinline__ Aff compose(const Aff& a, const Aff& b) {   // a . b
  Aff r;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      r.m[i * 4 + j] = a.m[i * 4 + 0] * b.m[0 * 4 + j] + a.m[i * 4 + 1] * b.m[1 * 4 + j] +
                       a.m[i * 4 + 2] * b.m[2 * 4 + j];
    r.m[i * 4 + 3] = a.m[i * 4 + 0] * b.m[3] + a.m[i * 4 + 1] * b.m[7] + a.m[i * 4 + 2] * b.m[11] +
                     a.m[i * 4 + 3];
  }
  return r;
}

// R = I + sin(t) K + (1 - cos(t)) K^2, t = |v + 1e-8|, K = skew(v / t)   (lbs.py:299-333)
__device__ __forceinline__ void rodrigues(const float v[3], float R[9]) {
  const float ex = v[0] + 1e-8f, ey = v[1] + 1e-8f, ez = v[2] + 1e-8f;
  const float t = sqrtf(ex * ex + ey * ey + ez * ez);
  const float kx = v[0] / t, ky = v[1] / t, kz = v[2] / t;
  const float s = sinf(t), c1 = 1.0f - cosf(t);
  const float kk = kx * kx + ky * ky + kz * kz;
  // K^2 = k k^T - (k.k) I
  R[0] = 1.0f + c1 * (kx * kx - kk);
  R[1] = -s * kz + c1 * kx * ky;
  R[2] = s * ky + c1 * kx * kz;
  R[3] = s * kz + c1 * kx * ky;
  R[4] = 1.0f + c1 * (ky * ky - kk);
  R[5] = -s * kx + c1 * ky * kz;
  R[6] = -s * ky + c1 * kx * kz;
  R[7] = s * kx + c1 * ky * kz;
  R[8] = 1.0f + c1 * (kz * kz - kk);
}

// dL/dv from dL/dR for the map above.
__device__ __forceinline__ void rodrigues_bwd(const float v[3], const float dR[9], float dv[3]) {
  const float e[3] = {v[0] + 1e-8f, v[1] + 1e-8f, v[2] + 1e-8f};
  const float t = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  const float k[3] = {v[0] / t, v[1] / t, v[2] / t};
  const float s = sinf(t), c = cosf(t), c1 = 1.0f - c;
  const float kk = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
  const float K[9] = {0.f, -k[2], k[1], k[2], 0.f, -k[0], -k[1], k[0], 0.f};
  float K2[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) K2[a * 3 + b] = k[a] * k[b] - (a == b ? kk : 0.f);
  float dt = 0.f;
#pragma unroll
  for (int q = 0; q < 9; ++q) dt += dR[q] * (c * K[q] + s * K2[q]);
  const float tr = dR[0] + dR[4] + dR[8];
  float dk[3];
  dk[0] = s * (dR[7] - dR[5]);
  dk[1] = s * (dR[2] - dR[6]);
  dk[2] = s * (dR[3] - dR[1]);
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    float acc = -2.0f * k[m] * tr;
#pragma unroll
    for (int b = 0; b < 3; ++b) acc += dR[m * 3 + b] * k[b] + dR[b * 3 + m] * k[b];
    dk[m] += c1 * acc;
  }
  // k = v / t, t = |v + eps|
  const float dkv = dk[0] * v[0] + dk[1] * v[1] + dk[2] * v[2];
  const float dt_total = dt - dkv / (t * t);
#pragma unroll
  for (int m = 0; m < 3; ++m) dv[m] = dk[m] / t + dt_total * e[m] / t;
}

// ------------------------------------------------------------------ joint transforms
__global__ void __launch_bounds__(WAVE)
joint_fwd_kernel(int J, const float* __restrict__ pose, const float* __restrict__ transl,
                 const float* __restrict__ joints_rest, const int32_t* __restrict__ parents,
                 const float* __restrict__ inv_mats, int64_t inv_stride, float* __restrict__ A,
                 float* __restrict__ cano2live, float* __restrict__ saved) {
  __shared__ Aff s_local[GALBS_MAX_JOINTS];
  __shared__ Aff s_glob[GALBS_MAX_JOINTS];
  __shared__ int s_par[GALBS_MAX_JOINTS];      // the chain below must not wait on global loads
  const int b = blockIdx.x;
  const int j = threadIdx.x;
  for (int q = j; q < J; q += WAVE) s_par[q] = parents[q];
  float R[9];
  if (j < J) {
    const float v[3] = {pose[(size_t)b * J * 3 + j * 3], pose[(size_t)b * J * 3 + j * 3 + 1],
                        pose[(size_t)b * J * 3 + j * 3 + 2]};
    rodrigues(v, R);
    const int p = parents[j];
    Aff L;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) L.m[r * 4 + c] = R[r * 3 + c];
      L.m[r * 4 + 3] = joints_rest[j * 3 + r] - (j > 0 ? joints_rest[p * 3 + r] : 0.f);
    }
    s_local[j] = L;
  }
  __syncthreads();
  if (j == 0) {   // the kinematic chain is inherently sequential (parents[i] < i)
    s_glob[0] = s_local[0];
    for (int i = 1; i < J; ++i) s_glob[i] = compose(s_glob[s_par[i]], s_local[i]);
  }
  __syncthreads();
  if (j < J) {
    const Aff G = s_glob[j];
    float* sv = saved + ((size_t)b * J + j) * SAVED_PER_JOINT;
#pragma unroll
    for (int q = 0; q < 9; ++q) sv[q] = R[q];
#pragma unroll
    for (int q = 0; q < 12; ++q) sv[9 + q] = G.m[q];
    float a[16];
    const float jx = joints_rest[j * 3], jy = joints_rest[j * 3 + 1], jz = joints_rest[j * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a[r * 4 + c] = G.m[r * 4 + c];
      a[r * 4 + 3] = G.m[r * 4 + 3] - (G.m[r * 4] * jx + G.m[r * 4 + 1] * jy + G.m[r * 4 + 2] * jz);
      if (transl) a[r * 4 + 3] += transl[b * 3 + r];
    }
    a[12] = 0.f; a[13] = 0.f; a[14] = 0.f; a[15] = 1.f;
    float* Ao = A + ((size_t)b * J + j) * 16;
#pragma unroll
    for (int q = 0; q < 16; ++q) Ao[q] = a[q];
    const float* inv = inv_mats + (size_t)b * inv_stride + (size_t)j * 16;
    float* Mo = cano2live + ((size_t)b * J + j) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        Mo[r * 4 + c] = a[r * 4] * inv[c] + a[r * 4 + 1] * inv[4 + c] + a[r * 4 + 2] * inv[8 + c] +
                        a[r * 4 + 3] * inv[12 + c];
  }
}

__global__ void __launch_bounds__(WAVE)
joint_bwd_kernel(int J, const float* __restrict__ pose, const float* __restrict__ joints_rest,
                 const int32_t* __restrict__ parents, const float* __restrict__ inv_mats,
                 int64_t inv_stride, const float* __restrict__ saved,
                 const float* __restrict__ dM, const float* __restrict__ dA_in,
                 float* __restrict__ dpose, float* __restrict__ dtransl) {
  __shared__ Aff s_dG[GALBS_MAX_JOINTS];    // gradient w.r.t. global transforms
  __shared__ float s_dRl[GALBS_MAX_JOINTS][9];
  __shared__ float s_dt[GALBS_MAX_JOINTS][3];
  // the serial sweep below reads the saved forward state joint by joint: staged in LDS first, so that
  // each of its steps costs LDS latency instead of a dependent global load (21 -> ~5 us)
  __shared__ float s_saved[GALBS_MAX_JOINTS][SAVED_PER_JOINT];
  __shared__ float s_jr[GALBS_MAX_JOINTS][3];
  __shared__ int s_par[GALBS_MAX_JOINTS];
  const int b = blockIdx.x;
  const int j = threadIdx.x;
  for (int q = j; q < J * SAVED_PER_JOINT; q += WAVE)
    s_saved[q / SAVED_PER_JOINT][q % SAVED_PER_JOINT] = saved[(size_t)b * J * SAVED_PER_JOINT + q];
  for (int q = j; q < J * 3; q += WAVE) s_jr[q / 3][q % 3] = joints_rest[q];
  for (int q = j; q < J; q += WAVE) s_par[q] = parents[q];
  if (j < J) {
    float dA[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) dA[q] = dA_in ? dA_in[((size_t)b * J + j) * 16 + q] : 0.f;
    if (dM) {
      const float* g = dM + ((size_t)b * J + j) * 16;
      const float* inv = inv_mats + (size_t)b * inv_stride + (size_t)j * 16;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          dA[r * 4 + c] += g[r * 4] * inv[c * 4] + g[r * 4 + 1] * inv[c * 4 + 1] +
                           g[r * 4 + 2] * inv[c * 4 + 2] + g[r * 4 + 3] * inv[c * 4 + 3];
    }
    const float jx = joints_rest[j * 3], jy = joints_rest[j * 3 + 1], jz = joints_rest[j * 3 + 2];
    Aff dG;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      dG.m[r * 4 + 0] = dA[r * 4 + 0] - dA[r * 4 + 3] * jx;
      dG.m[r * 4 + 1] = dA[r * 4 + 1] - dA[r * 4 + 3] * jy;
      dG.m[r * 4 + 2] = dA[r * 4 + 2] - dA[r * 4 + 3] * jz;
      dG.m[r * 4 + 3] = dA[r * 4 + 3];
      s_dt[j][r] = dA[r * 4 + 3];
    }
    s_dG[j] = dG;
  }
  __syncthreads();
  if (j == 0) {   // reverse sweep over the tree
    for (int i = J - 1; i >= 1; --i) {
      const int p = s_par[i];
      const float* svp = &s_saved[p][9];   // G_p (3x4)
      const float* svi = &s_saved[i][0];   // R_i
      const Aff dGi = s_dG[i];
      float Lt[3];
      for (int r = 0; r < 3; ++r) Lt[r] = s_jr[i][r] - s_jr[p][r];
      // local rotation gradient: dR_i = G_p.R^T dG_i.R
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          s_dRl[i][r * 3 + c] = svp[0 * 4 + r] * dGi.m[0 * 4 + c] + svp[1 * 4 + r] * dGi.m[1 * 4 + c] +
                                svp[2 * 4 + r] * dGi.m[2 * 4 + c];
      // parent: dG_p.R += dG_i.R R_i^T + dG_i.t (x) t_i ; dG_p.t += dG_i.t
      Aff dGp = s_dG[p];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          dGp.m[r * 4 + c] += dGi.m[r * 4 + 0] * svi[c * 3 + 0] + dGi.m[r * 4 + 1] * svi[c * 3 + 1] +
                              dGi.m[r * 4 + 2] * svi[c * 3 + 2] + dGi.m[r * 4 + 3] * Lt[c];
        dGp.m[r * 4 + 3] += dGi.m[r * 4 + 3];
      }
      s_dG[p] = dGp;
    }
    const Aff d0 = s_dG[0];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) s_dRl[0][r * 3 + c] = d0.m[r * 4 + c];
    if (dtransl) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
      for (int i = 0; i < J; ++i) { t0 += s_dt[i][0]; t1 += s_dt[i][1]; t2 += s_dt[i][2]; }
      dtransl[b * 3] = t0; dtransl[b * 3 + 1] = t1; dtransl[b * 3 + 2] = t2;
    }
  }
  __syncthreads();
  if (j < J && dpose) {
    const float v[3] = {pose[(size_t)b * J * 3 + j * 3], pose[(size_t)b * J * 3 + j * 3 + 1],
                        pose[(size_t)b * J * 3 + j * 3 + 2]};
    float dR[9], dv[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) dR[q] = s_dRl[j][q];
    rodrigues_bwd(v, dR, dv);
#pragma unroll
    for (int m = 0; m < 3; ++m) dpose[(size_t)b * J * 3 + j * 3 + m] = dv[m];
  }
}

// ------------------------------------------------------------------ point skinning
constexpr int SKIN_THREADS = 256;
constexpr int SKIN_FB = 8;   // frames handled per block (blend matrices staged in LDS)

// JT: compile-time joint count (24 = SMPL: the weight row is loaded as six 16-byte words up front and the
// blend loop is fully unrolled; with a run-time J the loop issues one dependent 4-byte load per joint and
// the kernel is latency-bound, 38 us instead of ~12 for 200k points x 2 frames); 0 = run-time J.
template <bool BWD, int JT>
__global__ void __launch_bounds__(SKIN_THREADS)
skin_kernel(int B, int N, int J, const float* __restrict__ points, int64_t pts_stride,
            const float* __restrict__ res, int64_t res_stride,
            const float* __restrict__ weights, int64_t w_stride,
            const float* __restrict__ mats, float* __restrict__ out,
            const float* __restrict__ dout, float* __restrict__ dres, float* __restrict__ dmats) {
  __shared__ float s_m[SKIN_FB][GALBS_MAX_JOINTS][12];
  __shared__ float s_dm[BWD ? SKIN_FB : 1][BWD ? GALBS_MAX_JOINTS : 1][12];
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int b0 = blockIdx.y * SKIN_FB;
  const int nb = min(SKIN_FB, B - b0);
  for (int q = tid; q < nb * J * 12; q += SKIN_THREADS) {
    const int f = q / (J * 12), rem = q % (J * 12), jj = rem / 12, e = rem % 12;
    s_m[f][jj][e] = mats[((size_t)(b0 + f) * J + jj) * 16 + e];
    if (BWD) s_dm[f][jj][e] = 0.f;
  }
  __syncthreads();
  for (int n0 = blockIdx.x * SKIN_THREADS; n0 < N; n0 += gridDim.x * SKIN_THREADS) {
    const int n = n0 + tid;
    const bool valid = n < N;
    for (int f = 0; f < nb; ++f) {
      const int b = b0 + f;
      const float* wrow = weights + (size_t)b * w_stride + (size_t)n * J;
      float x[3] = {0.f, 0.f, 0.f};
      if (valid) {
        const float* p = points + (size_t)b * pts_stride + (size_t)n * 3;
        x[0] = p[0]; x[1] = p[1]; x[2] = p[2];
        if (res) {
          const float* r = res + (size_t)b * res_stride + (size_t)n * 3;
          x[0] += r[0]; x[1] += r[1]; x[2] += r[2];
        }
      }
      float T[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = 0.f;
      if (JT > 0) {
        float4 wv[JT > 0 ? JT / 4 : 1];
        const float4* w4 = reinterpret_cast<const float4*>(weights + (size_t)b * w_stride + (size_t)(valid ? n : 0) * JT);
#pragma unroll
        for (int q = 0; q < JT / 4; ++q) wv[q] = w4[q];
#pragma unroll
        for (int q = 0; q < JT / 4; ++q) {
          const float wq[4] = {wv[q].x, wv[q].y, wv[q].z, wv[q].w};
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = fmaf(wq[c], s_m[f][4 * q + c][e], T[e]);
        }
      } else if (valid) {
        for (int jj = 0; jj < J; ++jj) {
          const float w = wrow[jj];
#pragma unroll
          for (int e = 0; e < 12; ++e) T[e] = fmaf(w, s_m[f][jj][e], T[e]);
        }
      }
      if (!BWD) {
        if (valid) {
          float* o = out + ((size_t)b * N + n) * 3;
#pragma unroll
          for (int r = 0; r < 3; ++r)
            o[r] = T[r * 4] * x[0] + T[r * 4 + 1] * x[1] + T[r * 4 + 2] * x[2] + T[r * 4 + 3];
        }
      } else {
        float g[3] = {0.f, 0.f, 0.f};
        if (valid) {
          const float* gp = dout + ((size_t)b * N + n) * 3;
          g[0] = gp[0]; g[1] = gp[1]; g[2] = gp[2];
          if (dres) {
            float* d = dres + ((size_t)b * N + n) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = T[c] * g[0] + T[4 + c] * g[1] + T[8 + c] * g[2];
          }
        }
        if (dmats) {
          // dL/dM_j = sum_n w_nj [g (x) x | g] ; wave-reduce per joint, skip joints that no
          // lane of the wave is bound to (skinning weights are sparse)
          float o12[12];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            o12[r * 4] = g[r] * x[0]; o12[r * 4 + 1] = g[r] * x[1];
            o12[r * 4 + 2] = g[r] * x[2]; o12[r * 4 + 3] = g[r];
          }
          for (int jj = 0; jj < J; ++jj) {
            const float w = valid ? wrow[jj] : 0.f;
            if (__ballot(w != 0.f) == 0ull) continue;
#pragma unroll
            for (int e = 0; e < 12; ++e) {
              float v = w * o12[e];
#pragma unroll
              for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
              if (lane == 0) atomicAdd(&s_dm[f][jj][e], v);
            }
          }
        }
      }
    }
  }
  if (BWD && dmats) {
    __syncthreads();
    for (int q = tid; q < nb * J * 12; q += SKIN_THREADS) {
      const int f = q / (J * 12), rem = q % (J * 12), jj = rem / 12, e = rem % 12;
      const float v = s_dm[f][jj][e];
      if (v != 0.f) unsafeAtomicAdd(&dmats[((size_t)(b0 + f) * J + jj) * 16 + e], v);
    }
  }
}

// dL/dM_j = sum_n w_nj [g_n (x) (x_n, 1)] as a skinny GEMM on the matrix cores:  D[j][e] += W^T[j][n] O[n][e]
// with O[n][4r+c] = g_n[r] * (x_n, 1)[c] (12 columns). v_mfma_f32_32x32x2_f32 takes two texels per step:
// lane (col, half) supplies w[n0+half][col] as the A operand and builds its own B element from g and x of
// texel n0+half. The operands of a chunk of 32 texels are contiguous in memory (32 J weights, 96 + 96 (+ 96)
// floats of dout / points / res): a wave fetches them with a handful of 16-byte buffer loads (rows past N read
// as 0), one chunk ahead, parks them in its own slice of LDS and picks the MFMA operands out of it with
// 4-byte LDS reads — the first version loaded every operand element straight from global memory, 64 load
// instructions per chunk instead of 6, and was bound by the address unit (52 us for 53 MB). The 16 waves of a
// workgroup add their tiles in LDS; one global atomic per (joint, entry) and workgroup.
typedef float skin_f32x16 __attribute__((ext_vector_type(16)));
typedef float skin_f32x4 __attribute__((ext_vector_type(4)));
constexpr int DM_BLOCKS = 128;
constexpr int DM_THREADS = 1024;        // 16 waves
constexpr int DM_CHUNK = 32;            // texels per chunk

__device__ __forceinline__ skin_f32x4 dm_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(skin_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

template <int JT>     // joint tiles of 32 (1: SMPL, 2: SMPL-X)
__global__ void __launch_bounds__(DM_THREADS)
skin_dmats_kernel(int N, int J, const float* __restrict__ points, int64_t pts_stride,
                  const float* __restrict__ res, int64_t res_stride,
                  const float* __restrict__ weights, int64_t w_stride,
                  const float* __restrict__ dout, float* __restrict__ dmats) {
  constexpr int WPB = DM_THREADS / 64;
  constexpr int NW = JT * DM_CHUNK * 32 / 256;           // 16-byte loads per lane that cover a chunk's weights
  constexpr int SLICE = JT * DM_CHUNK * 32 + 2 * 128;    // floats per wave: weights, dout, points (+ res)
  static_assert(SLICE >= JT * 32 * 12, "the wave's result tile reuses its slice");
  __shared__ __attribute__((aligned(16))) float s_all[WPB][SLICE];
  const int b = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int half = lane >> 5, col = lane & 31;
  const uint32_t kSkip = 0xffffffffu;
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(weights + (size_t)b * w_stride), 0, N * J * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_g = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dout + (size_t)b * N * 3), 0, N * 12, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_p = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(points + (size_t)b * pts_stride), 0, N * 12, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(res ? res + (size_t)b * res_stride : points), 0, res ? N * 12 : 0, 0x00020000);
  float* s_w = s_all[wave];
  float* s_g = s_w + JT * DM_CHUNK * 32;
  float* s_x = s_g + 128;
  const int chunks = (N + DM_CHUNK - 1) / DM_CHUNK, stride = gridDim.x * WPB;
  const int wlen = DM_CHUNK * J;                         // floats of weights per chunk

  skin_f32x4 wv[NW], gv, pv;
  auto fetch = [&](int c) {
    const bool live = c < chunks;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int f = 4 * lane + 256 * i;
      wv[i] = dm_load4(r_w, (live && f < wlen) ? (uint32_t)(c * wlen + f) * 4u : kSkip);
    }
    const uint32_t o = (live && lane < 24) ? (uint32_t)(c * 96 + 4 * lane) * 4u : kSkip;
    gv = dm_load4(r_g, o);
    pv = dm_load4(r_p, o) + dm_load4(r_r, o);
  };
  skin_f32x16 acc[JT];
#pragma unroll
  for (int t = 0; t < JT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int er = col >> 2, ec = col & 3;       // B element (row of g, column of (x,1)) of this lane
  const bool eok = col < 12;
  int c = blockIdx.x * WPB + wave;
  fetch(c);
  for (; c < chunks; c += stride) {
    __builtin_amdgcn_wave_barrier();            // the previous chunk's LDS reads were issued before (in-order LDS)
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (4 * lane + 256 * i < wlen) *reinterpret_cast<skin_f32x4*>(s_w + 4 * lane + 256 * i) = wv[i];
    if (lane < 24) {
      *reinterpret_cast<skin_f32x4*>(s_g + 4 * lane) = gv;
      *reinterpret_cast<skin_f32x4*>(s_x + 4 * lane) = pv;
    }
    __builtin_amdgcn_wave_barrier();
    fetch(c + stride);
#pragma unroll
    for (int st = 0; st < DM_CHUNK / 2; ++st) {
      const int n = 2 * st + half;
      const float g = s_g[n * 3 + (eok ? er : 0)];
      const float x = ec < 3 ? s_x[n * 3 + ec] : 1.f;
      const float bval = eok ? g * x : 0.f;
#pragma unroll
      for (int t = 0; t < JT; ++t) {
        const int j = t * 32 + col;
        const float a = s_w[n * J + (j < J ? j : 0)];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(j < J ? a : 0.f, bval, acc[t], 0, 0, 0);
      }
    }
  }
  // D layout: column e = lane & 31, row j = (reg & 3) + 8 * (reg >> 2) + 4 * half
  __builtin_amdgcn_wave_barrier();
  if (eok) {
#pragma unroll
    for (int t = 0; t < JT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_w[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 12 + col] = acc[t][r];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < J * 12; q += DM_THREADS) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) v += s_all[w][q];
    if (v != 0.f) unsafeAtomicAdd(&dmats[((size_t)b * J + q / 12) * 16 + q % 12], v);
  }
}

int check_common(int B, int J) {
  if (B <= 0 || J <= 0 || J > GALBS_MAX_JOINTS) {
    set_error("bad sizes: B=%d J=%d (J <= %d)", B, J, GALBS_MAX_JOINTS);
    return 1;
  }
  return 0;
}

}  // namespace

extern "C" {

size_t galbs_joint_saved_floats(int32_t J) { return (size_t)J * SAVED_PER_JOINT; }

int galbs_joint_transforms_fwd(int32_t B, int32_t J, const float* pose, const float* transl,
                               const float* joints_rest, const int32_t* parents,
                               const float* inv_mats, int64_t inv_batch_stride, float* A,
                               float* cano2live, float* saved, void* stream) {
  if (check_common(B, J)) return 1;
  if (!pose || !joints_rest || !parents || !inv_mats || !A || !cano2live || !saved) {
    set_error("galbs_joint_transforms_fwd: NULL argument");
    return 1;
  }
  hipLaunchKernelGGL(joint_fwd_kernel, dim3(B), dim3(WAVE), 0, static_cast<hipStream_t>(stream), J,
                     pose, transl, joints_rest, parents, inv_mats, inv_batch_stride, A, cano2live,
                     saved);
  return check_hip(hipGetLastError(), "joint_fwd_kernel");
}

int galbs_joint_transforms_bwd(int32_t B, int32_t J, const float* pose, const float* joints_rest,
                               const int32_t* parents, const float* inv_mats,
                               int64_t inv_batch_stride, const float* saved,
                               const float* dL_dcano2live, const float* dL_dA, float* dL_dpose,
                               float* dL_dtransl, void* stream) {
  if (check_common(B, J)) return 1;
  if (!pose || !joints_rest || !parents || !inv_mats || !saved || (!dL_dcano2live && !dL_dA)) {
    set_error("galbs_joint_transforms_bwd: NULL argument");
    return 1;
  }
  hipLaunchKernelGGL(joint_bwd_kernel, dim3(B), dim3(WAVE), 0, static_cast<hipStream_t>(stream), J,
                     pose, joints_rest, parents, inv_mats, inv_batch_stride, saved, dL_dcano2live,
                     dL_dA, dL_dpose, dL_dtransl);
  return check_hip(hipGetLastError(), "joint_bwd_kernel");
}

static dim3 skin_grid(int B, int N) {
  int gx = (N + SKIN_THREADS - 1) / SKIN_THREADS;
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  return dim3(gx, (B + SKIN_FB - 1) / SKIN_FB);
}

int galbs_skin_fwd(int32_t B, int32_t N, int32_t J, const float* points, int64_t pts_batch_stride,
                   const float* res, int64_t res_batch_stride, const float* weights,
                   int64_t w_batch_stride, const float* mats, float* out, void* stream) {
  if (check_common(B, J) || N < 0) { if (N < 0) set_error("N < 0"); return 1; }
  if (N == 0) return 0;
  if (!points || !weights || !mats || !out) { set_error("galbs_skin_fwd: NULL argument"); return 1; }
  // SMPL's 24 joints with 16-byte aligned weight rows take the unrolled variant
  const bool j24 = J == 24 && (reinterpret_cast<uintptr_t>(weights) & 15) == 0 && (w_batch_stride % 4) == 0;
  if (j24)
    hipLaunchKernelGGL((skin_kernel<false, 24>), skin_grid(B, N), dim3(SKIN_THREADS), 0,
                       static_cast<hipStream_t>(stream), B, N, J, points, pts_batch_stride, res,
                       res_batch_stride, weights, w_batch_stride, mats, out, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((skin_kernel<false, 0>), skin_grid(B, N), dim3(SKIN_THREADS), 0,
                       static_cast<hipStream_t>(stream), B, N, J, points, pts_batch_stride, res,
                       res_batch_stride, weights, w_batch_stride, mats, out, nullptr, nullptr, nullptr);
  return check_hip(hipGetLastError(), "skin_kernel<fwd>");
}

int galbs_skin_bwd(int32_t B, int32_t N, int32_t J, const float* points, int64_t pts_batch_stride,
                   const float* res, int64_t res_batch_stride, const float* weights,
                   int64_t w_batch_stride, const float* mats, const float* dL_dout, float* dL_dres,
                   float* dL_dmats, void* stream) {
  if (check_common(B, J) || N < 0) { if (N < 0) set_error("N < 0"); return 1; }
  if (!dL_dres && !dL_dmats) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dL_dmats) {
    int rc = check_hip(hipMemsetAsync(dL_dmats, 0, (size_t)B * J * 16 * sizeof(float), s), "memset dmats");
    if (rc) return rc;
  }
  if (N == 0) return 0;
  if (!points || !weights || !mats || !dL_dout) { set_error("galbs_skin_bwd: NULL argument"); return 1; }
  if (dL_dmats) {      // matrix gradients: skinny GEMM on the matrix cores (skin_dmats_kernel)
    const dim3 g2(DM_BLOCKS, B);
    if (J <= 32)
      hipLaunchKernelGGL(skin_dmats_kernel<1>, g2, dim3(DM_THREADS), 0, s, N, J, points, pts_batch_stride, res,
                         res_batch_stride, weights, w_batch_stride, dL_dout, dL_dmats);
    else
      hipLaunchKernelGGL(skin_dmats_kernel<2>, g2, dim3(DM_THREADS), 0, s, N, J, points, pts_batch_stride, res,
                         res_batch_stride, weights, w_batch_stride, dL_dout, dL_dmats);
    int rc = check_hip(hipGetLastError(), "skin_dmats_kernel");
    if (rc) return rc;
  }
  if (dL_dres) {
    const bool j24 = J == 24 && (reinterpret_cast<uintptr_t>(weights) & 15) == 0 && (w_batch_stride % 4) == 0;
    if (j24)
      hipLaunchKernelGGL((skin_kernel<true, 24>), skin_grid(B, N), dim3(SKIN_THREADS), 0, s, B, N, J, points,
                         pts_batch_stride, res, res_batch_stride, weights, w_batch_stride, mats, nullptr,
                         dL_dout, dL_dres, nullptr);
    else
      hipLaunchKernelGGL((skin_kernel<true, 0>), skin_grid(B, N), dim3(SKIN_THREADS), 0, s, B, N, J, points,
                         pts_batch_stride, res, res_batch_stride, weights, w_batch_stride, mats, nullptr,
                         dL_dout, dL_dres, nullptr);
  }
  return check_hip(hipGetLastError(), "skin_kernel<bwd>");
}

const char* galbs_last_error(void) { return g_err; }
int galbs_abi_version(void) { return GALBS_ABI_VERSION; }

}  // extern "C"
