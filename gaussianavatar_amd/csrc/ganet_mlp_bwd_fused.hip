// ganet_mlp_bwd_fused.hip — backward of a hidden decoder layer (128 -> 128) in ONE pass over its activations:
// the data gradient (-> G_src, what ganet_mlp_bwd_data computes) AND the weight gradient (what ganet_wgrad_act
// computes) of layer i from the same 32-row slabs of (G_i, z_i, z_src).
//
// Separately the two kernels move 7 activation tensors per layer (G_i, z_i, z_src -> G_src; G_i, z_i, z_src)
// = 0.94 GB at M = 262,144 and took 123 + 99 us; fused, HBM sees 4 (G_i, z_i, z_src in, G_src out) and both
// matrix products (17.2 GFLOP per layer) run back to back on the same wave:
//
//   phase D  dY = dz_i . W_i           dz_i = A G_i + q z_i + p assembled on load, ROW layout (lane = row,
//                                      16-byte loads along the contracted index n); W^T resident in LDS;
//                                      4 accumulators (32 x 128 tile)                        — 256 MFMAs
//   epilogue G_src = dY . softplus'(u_src), written out, column sums for the BatchNorm backward of the
//            source layer; softplus(u_src) — the OTHER product's operand x — falls out of the same
//            exponential and stays in registers in the MFMA C/D layout (lane = column, register = row)
//   phase W  dW_i += dz_i^T . x        contraction over the slab's ROWS: both operands need lane = column.
//                                      x is already there (epilogue); dz_i is re-read in that layout with
//                                      dword loads (the slab was streamed a moment ago: L2 hits, no HBM),
//                                      16 accumulators = the whole 128 x 128 tile in AGPRs     — 256 MFMAs
//
// (The two products contract dz over different indices — columns for the data gradient, rows for the weight
// gradient — and MFMA operand layouts put the contracted index along registers/steps and the free index along
// lanes, so one of them needs dz "transposed"; re-reading it from cache in the other fragment layout costs no
// LDS traffic at all.)
// One wave per SIMD (512 registers), 4 per workgroup, 256 workgroups; all latency hiding is explicit: the source
// layer's z rows are requested at the start of phase D, the phase-W operands before the epilogue, the next
// slab's phase-D operands during phase W. The four waves' weight-gradient tiles are combined through LDS in a
// fixed order; per-workgroup partials go to the workspace ganet_wgrad_reduce_batch sums (same format as
// ganet_wgrad_act's). Replaces, per layer, /root/reference/model/modules.py:554-582's autograd graph of
// conv1d backward (two GEMMs) + BatchNorm backward + softplus backward.
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"

namespace ganet {

namespace {

constexpr int FWG = 256;
constexpr int FWAVES = FWG / 64;
constexpr int SLAB = 32;
constexpr int FUSED_BLOCKS = 256;
constexpr int K = 128;                       // layer width (rows of W, columns of G/z)
constexpr int LDW4 = K / 4 + 1;              // float4 pitch of the W^T image
constexpr int D = 4;                         // phase-D operand ring (k-blocks of 8 columns)
constexpr int TILE = K * K + K;              // floats of a weight-gradient tile + bias row
constexpr int XSLAB = SLAB * K;               // floats of one wave's z_src slab staged in LDS
constexpr size_t FUSED_LDS_MAIN = ((size_t)K * LDW4 + 96) * sizeof(float4) + (size_t)FWAVES * XSLAB * sizeof(float);
constexpr size_t FUSED_LDS_END = (2 * (size_t)TILE + (size_t)FWAVES * 256) * sizeof(float);
constexpr size_t FUSED_LDS = FUSED_LDS_MAIN > FUSED_LDS_END ? FUSED_LDS_MAIN : FUSED_LDS_END;

__global__ void __attribute__((amdgpu_flat_work_group_size(FWG, FWG), amdgpu_waves_per_eu(1, 1)))
mlp_bwd_fused_kernel(int64_t M, const float* __restrict__ g, const float* __restrict__ gz,
                     const float* __restrict__ gcoef, const float* __restrict__ W, float* __restrict__ out,
                     const float* __restrict__ src_z, const float* __restrict__ src_scale,
                     const float* __restrict__ src_shift, float* __restrict__ col_part,
                     float* __restrict__ wpartial, int reverse) {
  extern __shared__ float4 s_mem[];     // main loop: Wt [128][LDW4] | A [32] | q [32] | p [32] (float4 units) |
                                        // z_src slab [4 waves][32][128] floats (LDS-DMA target);
                                        // at the end: two weight-gradient tiles + column-sum scratch
  float4* s_w = s_mem;
  float4* s_cA = s_mem + K * LDW4;
  float4* s_cq = s_cA + 32;
  float4* s_cp = s_cq + 32;
  float4* s_x4 = s_cp + 32;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, col = lane & 31;

  {   // stage W[n][o] transposed into s_w[o][n]
    constexpr int PER = K * K / FWG;
    float* s_wf = reinterpret_cast<float*>(s_w);
    float wv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) wv[j] = W[threadIdx.x + j * FWG];          // i = n * 128 + o
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = threadIdx.x + j * FWG;
      const int n = i >> 7, o = i & 127;
      s_wf[o * (4 * LDW4) + n] = wv[j];
    }
  }
  for (int i = threadIdx.x; i < 32; i += FWG) {
    s_cA[i] = *reinterpret_cast<const float4*>(gcoef + 4 * i);
    s_cq[i] = *reinterpret_cast<const float4*>(gcoef + K + 4 * i);
    s_cp[i] = *reinterpret_cast<const float4*>(gcoef + 2 * K + 4 * i);
  }
  __syncthreads();

  // per-lane column constants (lane = column 32 t + col of the C/D layout)
  float ssc[4], ssh[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int o = t * 32 + col;
    ssc[t] = src_scale[o] * kLog2e;          // log2 units (ganet_mlp_common.h)
    ssh[t] = src_shift[o] * kLog2e;
  }
  // phase W reads its per-column coefficients (A, q, p)[32 j + col] from the LDS image (12 registers saved)
  const float* s_cAf = reinterpret_cast<const float*>(s_cA);
  const float* s_cqf = reinterpret_cast<const float*>(s_cq);
  const float* s_cpf = reinterpret_cast<const float*>(s_cp);
  float csum[4] = {0.f, 0.f, 0.f, 0.f}, csz[4] = {0.f, 0.f, 0.f, 0.f}, bias[4] = {0.f, 0.f, 0.f, 0.f};
  f32x16 accw[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[j][i][r] = 0.f;

  const int64_t nslab = M / SLAB;
  const int64_t wave_global = (int64_t)blockIdx.x * FWAVES + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * FWAVES;
  auto phys = [&](int64_t slab) { return reverse ? nslab - 1 - slab : slab; };

  // phase-D operand ring of the first slab
  float4 ag[D], az[D];
  {
    const int64_t row = phys(min(wave_global, nslab - 1)) * SLAB + col;
    const float* pg = g + row * K + 4 * h;
    const float* pz = gz + row * K + 4 * h;
#pragma unroll
    for (int b = 0; b < D; ++b) {
      ag[b] = *reinterpret_cast<const float4*>(pg + 8 * b);
      az[b] = *reinterpret_cast<const float4*>(pz + 8 * b);
    }
  }

  for (int64_t slab = wave_global; slab < nslab; slab += wave_stride) {
    const int64_t row0 = phys(slab) * SLAB;
    const float* pgc = g + (row0 + col) * K + 4 * h;              // row layout (phase D)
    const float* pzc = gz + (row0 + col) * K + 4 * h;
    const int64_t cd0 = (row0 + 4 * h) * K + col;                 // C/D layout: + (8 q + u) * K + 32 t
    // the source layer's z rows of this slab: global -> LDS DMA (no registers), row-major [32][128] image of
    // this wave; instruction i moves rows 2 i, 2 i + 1 (lane l: 16 bytes at float4 index 64 i + l). Requested
    // now, read by the epilogue.
    {
      const float* zs = src_z + row0 * K + 4 * lane;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds(zs + i * 256, (__attribute__((address_space(3))) void*)(s_x4 + wave * (XSLAB / 4) + i * 64),
                                         16, 0, 0);
    }
    const float* s_xf = reinterpret_cast<const float*>(s_x4) + wave * XSLAB + 4 * h * K + col;
    __builtin_amdgcn_sched_barrier(0);

    // ---------------------------------------------------------------- phase D: dY = dz . W
    f32x16 accd[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) accd[t][r] = 0.f;
    int woff = col * LDW4 + h;
    int soff = h;
    asm volatile("" : "+v"(woff), "+v"(soff));      // keep Wt's fragment in LDS (see ganet_mlp.hip)
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int slot = b % D;
      const float4 cA = s_cA[soff + 2 * b], cq = s_cq[soff + 2 * b], cp = s_cp[soff + 2 * b];
      const float av0 = fmaf(cA.x, ag[slot].x, fmaf(cq.x, az[slot].x, cp.x));
      const float av1 = fmaf(cA.y, ag[slot].y, fmaf(cq.y, az[slot].y, cp.y));
      const float av2 = fmaf(cA.z, ag[slot].z, fmaf(cq.z, az[slot].z, cp.z));
      const float av3 = fmaf(cA.w, ag[slot].w, fmaf(cq.w, az[slot].w, cp.w));
      __builtin_amdgcn_sched_barrier(kSchedMask);
      if (b + D < 16) {
        ag[slot] = *reinterpret_cast<const float4*>(pgc + 8 * (b + D));
        az[slot] = *reinterpret_cast<const float4*>(pzc + 8 * (b + D));
      }
      __builtin_amdgcn_sched_barrier(kSchedMask);
      float4 bw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bw[t] = s_w[woff + t * 32 * LDW4 + 2 * b];
#pragma unroll
      for (int t = 0; t < 4; ++t) accd[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bw[t].x, accd[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) accd[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bw[t].y, accd[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) accd[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av2, bw[t].z, accd[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) accd[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av3, bw[t].w, accd[t], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the slab's DMA (issued 256 MFMAs ago) has landed
    __builtin_amdgcn_sched_barrier(0);
    // phase-W operands of the first row pair: in flight during the epilogue. Pair p = rows 8 (p >> 1) +
    // 2 (p & 1) + {0, 1} (+ 4 h) = C/D registers r = 4 (p >> 1) + 2 (p & 1) + {0, 1}
    float gc[3][4][2], zc[3][4][2];
    auto load_pair = [&](int buf, int p) {
      // one 64-bit base per pair, the 8 loads of a tensor differ by instruction immediates
      const int64_t base = cd0 + (int64_t)(8 * (p >> 1) + 2 * (p & 1)) * K;
      const float* gp = g + base;
      const float* zp = gz + base;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gc[buf][j][u] = gp[u * K + 32 * j];
          zc[buf][j][u] = zp[u * K + 32 * j];
        }
    };
    load_pair(0, 0);
    load_pair(1, 1);
    __builtin_amdgcn_sched_barrier(0);

    // ---------------------------------------------------------------- epilogue
    // sigmoid and softplus of the same argument share the exponential:
    // e = 2^-|u|, tt = 1 + e: softplus / ln 2 = max(u, 0) + log2(tt); sigmoid = (u >= 0 ? 1 : e) / tt
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float zv = s_xf[((r & 3) + 8 * (r >> 2)) * K + 32 * t];
        const float u2 = fmaf(ssc[t], zv, ssh[t]);
        const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(u2));
        const float tt = 1.0f + e;
        const float sg = (u2 >= 0.f ? 1.0f : e) * __builtin_amdgcn_rcpf(tt);
        const float v = accd[t][r] * sg;
        accd[t][r] = __builtin_fmaxf(u2, 0.0f) + __builtin_amdgcn_logf(tt);   // softplus / ln 2: phase W's operand
        csum[t] += v;
        csz[t] = fmaf(v, zv, csz[t]);
        out[cd0 + (int64_t)((r & 3) + 8 * (r >> 2)) * K + 32 * t] = v;
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---------------------------------------------------------------- phase W: dW += dz^T . x
    const bool more = slab + wave_stride < nslab;
    const int64_t nrow = phys(more ? slab + wave_stride : slab) * SLAB + col;
    float wA[4], wq[4], wp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { wA[j] = s_cAf[32 * j + col]; wq[j] = s_cqf[32 * j + col]; wp[j] = s_cpf[32 * j + col]; }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int cur = p % 3;
      if (p + 2 < 8) load_pair((p + 2) % 3, p + 2);
      if (p == 5) {          // the next slab's first phase-D operands
        const float* pg = g + nrow * K + 4 * h;
        const float* pz = gz + nrow * K + 4 * h;
#pragma unroll
        for (int b = 0; b < D; ++b) {
          ag[b] = *reinterpret_cast<const float4*>(pg + 8 * b);
          az[b] = *reinterpret_cast<const float4*>(pz + 8 * b);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = 4 * (p >> 1) + 2 * (p & 1) + u;
        float dz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dz[j] = fmaf(wA[j], gc[cur][j][u], fmaf(wq[j], zc[cur][j][u], wp[j]));
          bias[j] += dz[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            accw[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz[j], accd[i][r], accw[j][i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ------------------------------------------------------------------ combine the four waves
  __syncthreads();                                  // W^T image is dead from here on
  float* s_tile = reinterpret_cast<float*>(s_mem);
  float* tile = s_tile + (size_t)(wave & 1) * TILE;
  auto lds_index = [&](int j, int i, int r) { return (j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * K + i * 32 + col; };
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] += __shfl_xor(bias[j], 32);
  if (wave < 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[lds_index(j, i, r)] = accw[j][i][r] * kLn2;
      if (h == 0) tile[K * K + j * 32 + col] = bias[j];
    }
  }
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[lds_index(j, i, r)] += accw[j][i][r] * kLn2;
      if (h == 0) tile[K * K + j * 32 + col] += bias[j];
    }
  }
  // column sums of G_src and G_src z_src: [gridDim.x][2][128], waves combined in fixed order
  float* s_red = s_tile + 2 * TILE;                 // [FWAVES][256]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float s = csum[t] + __shfl_xor(csum[t], 32);
    const float q = csz[t] + __shfl_xor(csz[t], 32);
    if (h == 0) { s_red[wave * 256 + t * 32 + col] = s; s_red[wave * 256 + 128 + t * 32 + col] = q; }
  }
  __syncthreads();
  float* wout = wpartial + (size_t)blockIdx.x * TILE;
  const float* t0 = s_tile;
  const float* t1 = s_tile + TILE;
  for (int e = threadIdx.x; e < TILE; e += FWG) wout[e] = t0[e] + t1[e];
  {
    const int i = threadIdx.x;                       // 256 threads <-> 256 sums
    col_part[(size_t)blockIdx.x * 256 + i] = (s_red[i] + s_red[256 + i]) + (s_red[512 + i] + s_red[768 + i]);
  }
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

int32_t ganet_mlp_bwd_fused_parts(void) { return FUSED_BLOCKS; }

size_t ganet_mlp_bwd_fused_workspace(void) { return (size_t)FUSED_BLOCKS * TILE * sizeof(float); }

int ganet_mlp_bwd_fused(int64_t M, const float* g, const float* gz, const float* gcoef, const float* W,
                        float* out, const float* src_z, const float* src_scale, const float* src_shift,
                        float* col_part, void* wgrad_workspace, size_t workspace_bytes, int32_t row_order,
                        void* stream_) {
  if (M <= 0 || (M % SLAB) || !g || !gz || !gcoef || !W || !out || !src_z || !src_scale || !src_shift ||
      !col_part || !wgrad_workspace || !aligned16(g) || !aligned16(gz) || !aligned16(gcoef)) {
    set_error("ganet_mlp_bwd_fused: invalid arguments (M must be a multiple of %d, all tensors [M,128] / "
              "[128,128] contiguous, 16-byte aligned)", SLAB);
    return 1;
  }
  if (workspace_bytes < ganet_mlp_bwd_fused_workspace()) {
    set_error("ganet_mlp_bwd_fused: workspace too small (%zu < %zu)", workspace_bytes,
              ganet_mlp_bwd_fused_workspace());
    return 2;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  static bool attr_set = false;
  if (!attr_set) {
    if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_fused_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS),
                  "hipFuncSetAttribute")) return 3;
    attr_set = true;
  }
  ProfScope prof_(K_BWD_DATA, stream);
  hipLaunchKernelGGL(mlp_bwd_fused_kernel, dim3(FUSED_BLOCKS), dim3(FWG), FUSED_LDS, stream, M, g, gz, gcoef, W,
                     out, src_z, src_scale, src_shift, col_part, static_cast<float*>(wgrad_workspace),
                     row_order == 2 ? 1 : 0);
  return check_hip(hipGetLastError(), "mlp_bwd_fused_kernel");
}

}  // extern "C"
