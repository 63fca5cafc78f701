// ganet_wgrad.hip — weight/bias gradient of the decoder's 1x1-conv (point-wise linear) layers.
//
//   dW[n,k] = sum_m g[m,n] * x[m,k]      db[n] = sum_m g[m,n]        M = 262,144, N,K <= 224
//
// (the layers of /root/reference/model/modules.py:554-582 evaluated point-major). This is a
// GEMM whose reduction dimension is three orders of magnitude larger than its output, the shape
// vendor libraries handle worst (measured 17 TF/s = 487 us per layer on MI355X). Here:
//   * fp32-input MFMA v_mfma_f32_32x32x2_f32 — exact fp32, 157 TF peak;
//   * both operands are read straight from HBM in MFMA fragment layout: for one reduction step
//     (two rows m, m+1) lane l needs g[m + (l>>5)][n0 + (l&31)] and x[m + (l>>5)][k0 + (l&31)] —
//     two fully coalesced 128-byte segments per operand, no LDS staging, no transposes;
//   * the reduction dimension is split over up to 512 workgroups (2 per CU, 4 waves each = all 4
//     SIMDs' matrix pipes busy); per-workgroup partial tiles go to a workspace and a second tiny
//     kernel sums them (deterministic, no atomics).
#include <cstdarg>
#include <cstdio>

#include <hip/hip_runtime.h>

#include "ganet.h"
#include "ganet_common.h"

namespace ganet {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG = 256;
constexpr int UNROLL = 4;      // reduction steps (of 2 rows) whose loads are issued together
constexpr int MAX_BLOCKS = 512;

// SPLIT_N: wave w owns output row-tile w (32 rows of dW) and all KTW column tiles.
// !SPLIT_N (N <= 32): every wave owns row-tile 0 and column tiles w, w+4, ...
template <bool SPLIT_N, int KTW>
__global__ void __launch_bounds__(WG)
wgrad_kernel(int64_t M, int N, int K, const float* __restrict__ g, int64_t ldg,
             const float* __restrict__ x, int64_t ldx, float* __restrict__ partial,
             int64_t rows_per_block) {
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, col = lane & 31;
  const int NT = (N + 31) / 32, KT = (K + 31) / 32;
  const int ntile = SPLIT_N ? wave : 0;
  const bool owner = SPLIT_N ? (wave < NT) : true;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(r0 + rows_per_block, M);
  float* out = partial + (size_t)blockIdx.x * ((size_t)N * K + N);

  f32x16 acc[KTW];
#pragma unroll
  for (int j = 0; j < KTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float bias = 0.f;

  if (owner) {
    const int ncol = ntile * 32 + col;
    const bool nok = ncol < N;
    int kcol[KTW];
    bool kok[KTW];
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      const int kt = SPLIT_N ? j : wave + 4 * j;
      kcol[j] = kt * 32 + col;
      kok[j] = (kt < KT) && (kcol[j] < K);
    }
    for (int64_t m = r0; m < r1; m += 2 * UNROLL) {
      float a[UNROLL], b[UNROLL][KTW];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t row = m + 2 * u + half;
        const bool rok = row < r1;
        a[u] = (rok && nok) ? g[row * ldg + ncol] : 0.f;
#pragma unroll
        for (int j = 0; j < KTW; ++j) b[u][j] = (rok && kok[j]) ? x[row * ldx + kcol[j]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        bias += a[u];
#pragma unroll
        for (int j = 0; j < KTW; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][j], acc[j], 0, 0, 0);
      }
    }
    // C/D layout of 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      if (!kok[j]) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = ntile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < N) out[(size_t)n * K + kcol[j]] = acc[j][r];
      }
    }
    bias += __shfl_xor(bias, 32);
    if ((SPLIT_N || wave == 0) && half == 0 && nok) out[(size_t)N * K + ncol] = bias;
  }
}

// Sum the per-workgroup partial tiles. 64 output elements per block, 4 threads per element
// (each walks a quarter of the partials), combined through LDS.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(int nblocks, int N, int K, const float* __restrict__ partial,
                    float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float s_part[4][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const int total = N * K + N;
  float s0 = 0.f, s1 = 0.f;
  if (e < total) {
    int b = part;
    for (; b + 4 < nblocks; b += 8) {
      s0 += partial[(size_t)b * total + e];
      s1 += partial[(size_t)(b + 4) * total + e];
    }
    if (b < nblocks) s0 += partial[(size_t)b * total + e];
  }
  s_part[part][lane] = s0 + s1;
  __syncthreads();
  if (part == 0 && e < total) {
    const float s = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
    if (e < N * K) dW[e] = s;
    else if (db) db[e - N * K] = s;
  }
}

int plan_blocks(int64_t M, int64_t* rows_per_block) {
  int64_t rpb = (M + MAX_BLOCKS - 1) / MAX_BLOCKS;
  rpb = ((rpb + 2 * UNROLL - 1) / (2 * UNROLL)) * (2 * UNROLL);
  if (rpb < 2 * UNROLL) rpb = 2 * UNROLL;
  *rows_per_block = rpb;
  return (int)((M + rpb - 1) / rpb);
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

size_t ganet_linear_wgrad_workspace(int64_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int64_t rpb;
  const int nb = plan_blocks(M, &rpb);
  return (size_t)nb * ((size_t)N * K + N) * sizeof(float);
}

int ganet_linear_wgrad(int64_t M, int32_t N, int32_t K, const float* g, int64_t ldg,
                       const float* x, int64_t ldx, float* dW, float* db, void* workspace,
                       size_t workspace_bytes, void* stream_) {
  if (M <= 0 || N <= 0 || K <= 0 || !g || !x || !dW || ldg < N || ldx < K) {
    set_error("ganet_linear_wgrad: invalid arguments");
    return 1;
  }
  if (N > 128 || K > 224) {
    set_error("ganet_linear_wgrad: unsupported shape N=%d K=%d (N <= 128, K <= 224)", N, K);
    return 4;
  }
  int64_t rpb;
  const int nb = plan_blocks(M, &rpb);
  const size_t need = (size_t)nb * ((size_t)N * K + N) * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    set_error("ganet_linear_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    return 2;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* partial = static_cast<float*>(workspace);
  const int KT = (K + 31) / 32;
  const dim3 grid(nb), block(WG);
#define LAUNCH(SN, KTW) \
  hipLaunchKernelGGL((wgrad_kernel<SN, KTW>), grid, block, 0, stream, M, N, K, g, ldg, x, ldx, partial, rpb)
  if (N > 32) {
    switch (KT) {
      case 1: LAUNCH(true, 1); break;
      case 2: LAUNCH(true, 2); break;
      case 3: LAUNCH(true, 3); break;
      case 4: LAUNCH(true, 4); break;
      case 5: LAUNCH(true, 5); break;
      case 6: LAUNCH(true, 6); break;
      default: LAUNCH(true, 7); break;
    }
  } else {
    if (KT <= 4) LAUNCH(false, 1); else LAUNCH(false, 2);
  }
#undef LAUNCH
  int rc = check_hip(hipGetLastError(), "wgrad_kernel");
  if (rc) return rc;
  const int total = N * K + N;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, nb, N, K,
                     partial, dW, db);
  return check_hip(hipGetLastError(), "wgrad_reduce_kernel");
}

}  // extern "C"
