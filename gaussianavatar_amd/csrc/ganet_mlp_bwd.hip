// ganet_mlp_bwd.hip — input-gradient ("dgrad") side of the fused decoder layers, with the whole
// BatchNorm + softplus backward folded into the GEMMs' prologue and epilogue.
//
// For a hidden layer i with pre-activation z_i, u_i = scale_i z_i + shift_i, y_i = softplus(u_i):
//
//   G_i  = dL/dy_i . softplus'(u_i)                                   (gradient w.r.t. u_i)
//   dz_i = scale_i (G_i - mean_m G_i - xhat_i mean_m(G_i xhat_i))     (BatchNorm backward)
//        = A_i G_i + q_i z_i + p_i         per column: A = scale, q = -scale c2 rstd,
//                                          p = -scale c1 + scale c2 rstd mean, c1 = mean(G),
//                                          c2 = mean(G xhat) = rstd (mean(G z) - mean mean(G))
//
// so once the two column sums sum_m G_i and sum_m G_i z_i are known, dz_i is a per-column affine
// combination of the two stored tensors G_i and z_i and never has to be materialised:
//   * mlp_bwd_kernel computes dL/dy_src = dz_i . W_i (same streamed-M MFMA pipeline as the forward
//     kernel, A operand assembled from G_i and z_i on load) and, in its epilogue, multiplies by
//     softplus'(u_src) — i.e. writes G_src — and accumulates sum G_src, sum G_src z_src;
//   * bwd_stats_kernel turns those sums into (A, q, p) of the source layer plus d gamma / d beta;
//   * the weight gradient (ganet_wgrad_act, GPRO variant) assembles dz_i the same way.
// This replaces, per layer, a vendor GEMM + two BatchNorm-backward passes + the softplus backward
// of /root/reference/model/modules.py:554-582's autograd graph.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"

namespace ganet {

namespace {

constexpr int WG = 512;
constexpr int WAVES = WG / 64;
constexpr int SLAB = 32;
constexpr int BWD_BLOCKS = 256;
constexpr int HEAD_BLOCKS = 512;

// out[M,O] (+)= (A G + q Z + p)[M,128] . W[128, 0:O] ; SIG: out *= softplus'(src_scale src_z + src_shift)
template <int NT, bool ACCUM, bool SIG>
__global__ void __attribute__((amdgpu_flat_work_group_size(WG, WG), amdgpu_waves_per_eu(2, 2)))
mlp_bwd_kernel(int64_t M, int O, const float* __restrict__ g, int64_t ldg,
               const float* __restrict__ gz, int64_t ldgz, const float* __restrict__ gcoef,
               const float* __restrict__ W, int64_t ldw, float* __restrict__ out, int64_t ldo,
               const float* __restrict__ src_z, int64_t ld_src, const float* __restrict__ src_scale,
               const float* __restrict__ src_shift, float* __restrict__ col_part, int reverse) {
  constexpr int KB = 16, K = 128;
  constexpr int LDW4 = K / 4 + 1;
  constexpr int NP = NT * 32;
  constexpr int D = 4;                  // ring depth (k-blocks); two operands per slot
  extern __shared__ float4 s_mem[];     // Wt [NP][LDW4] | A [32] | q [32] | p [32]   (float4 units)
  float4* s_w = s_mem;
  float4* s_cA = s_mem + NP * LDW4;
  float4* s_cq = s_cA + 32;
  float4* s_cp = s_cq + 32;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, col = lane & 31;

  {   // stage W[128(n)][ldw] transposed into s_w[o][n] (row o of the LDS image = column o of W):
      // all global loads first (scalar: W may be a column slice with any alignment), then LDS stores
    constexpr int PER = (K * NP + WG - 1) / WG;            // elements per thread
    float* s_wf = reinterpret_cast<float*>(s_w);
    float wv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = min((int)threadIdx.x + j * WG, K * NP - 1);
      const int n = i / NP, o = i - n * NP;               // consecutive threads: consecutive columns o
      wv[j] = W[(size_t)n * ldw + min(o, O - 1)];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = threadIdx.x + j * WG;
      const int n = i / NP, o = i - n * NP;
      if (i < K * NP) s_wf[o * (4 * LDW4) + n] = o < O ? wv[j] : 0.f;
    }
  }
  for (int i = threadIdx.x; i < 32; i += WG) {
    s_cA[i] = *reinterpret_cast<const float4*>(gcoef + 4 * i);
    s_cq[i] = *reinterpret_cast<const float4*>(gcoef + K + 4 * i);
    s_cp[i] = *reinterpret_cast<const float4*>(gcoef + 2 * K + 4 * i);
  }
  __syncthreads();

  float csum[NT], csz[NT], ssc[NT], ssh[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    csum[t] = 0.f; csz[t] = 0.f;
    const int o = min(t * 32 + col, O - 1);
    ssc[t] = SIG ? src_scale[o] * kLog2e : 0.f;          // log2 units (ganet_mlp_common.h)
    ssh[t] = SIG ? src_shift[o] * kLog2e : 0.f;
  }

  const int64_t nslab = (M + SLAB - 1) / SLAB;
  const int64_t wave_global = (int64_t)blockIdx.x * WAVES + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * WAVES;

  const float *pgc, *pzc, *pgn, *pzn;
  auto phys = [&](int64_t slab) { return reverse ? nslab - 1 - slab : slab; };   // see mlp_fwd_kernel
  auto point_at = [&](int64_t slab, const float*& qg, const float*& qz) {
    const int64_t row = min(phys(slab) * SLAB + col, M - 1);
    qg = g + row * ldg + 4 * h;
    qz = gz + row * ldgz + 4 * h;
  };
  float4 ag[D], az[D];
  point_at(min(wave_global, nslab - 1), pgc, pzc);
#pragma unroll
  for (int b = 0; b < D; ++b) {
    ag[b] = *reinterpret_cast<const float4*>(pgc + 8 * b);
    az[b] = *reinterpret_cast<const float4*>(pzc + 8 * b);
  }

  for (int64_t slab = wave_global; slab < nslab; slab += wave_stride) {
    point_at(min(slab + wave_stride, nslab - 1), pgn, pzn);
    int woff = col * LDW4 + h;
    int soff = h;
    asm volatile("" : "+v"(woff), "+v"(soff));      // keep Wt's fragment in LDS (see ganet_mlp.hip)
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      const int slot = b % D;
      const float4 cA = s_cA[soff + 2 * b], cq = s_cq[soff + 2 * b], cp = s_cp[soff + 2 * b];
      const float av0 = fmaf(cA.x, ag[slot].x, fmaf(cq.x, az[slot].x, cp.x));
      const float av1 = fmaf(cA.y, ag[slot].y, fmaf(cq.y, az[slot].y, cp.y));
      const float av2 = fmaf(cA.z, ag[slot].z, fmaf(cq.z, az[slot].z, cp.z));
      const float av3 = fmaf(cA.w, ag[slot].w, fmaf(cq.w, az[slot].w, cp.w));
      // raw values dead: refill the slot, pinned between the MFMA groups (see ganet_mlp.hip)
      __builtin_amdgcn_sched_barrier(kSchedMask);
      if (b + D < KB) {
        ag[slot] = *reinterpret_cast<const float4*>(pgc + 8 * (b + D));
        az[slot] = *reinterpret_cast<const float4*>(pzc + 8 * (b + D));
      } else {
        ag[slot] = *reinterpret_cast<const float4*>(pgn + 8 * (b + D - KB));
        az[slot] = *reinterpret_cast<const float4*>(pzn + 8 * (b + D - KB));
      }
      __builtin_amdgcn_sched_barrier(kSchedMask);
      float4 bw[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bw[t] = s_w[woff + t * 32 * LDW4 + 2 * b];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bw[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bw[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av2, bw[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av3, bw[t].w, acc[t], 0, 0, 0);
    }
    pgc = pgn; pzc = pzn;
    // epilogue. C/D layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int64_t row0 = phys(slab) * SLAB;
    const bool full = row0 + SLAB <= M && O == NP;
    // The source layer's pre-activations are not cached: issue the loads of ALL tiles before the
    // first use, so that a slab pays one memory latency instead of one per tile.
    constexpr bool PRELOAD = SIG && !ACCUM;     // (the accumulate variant would spill: per tile there)
    float sz[PRELOAD ? NT : 1][16];
    if (PRELOAD) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int o = t * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const bool ok = full || (row < M && o < O);
          sz[t][r] = src_z[(ok ? row : 0) * ld_src + (ok ? o : 0)];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int o = t * 32 + col;
      float ex[16], szt[16];
      if (ACCUM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const bool ok = full || (row < M && o < O);
          ex[r] = out[(ok ? row : 0) * ldo + (ok ? o : 0)];
          if (SIG) szt[r] = src_z[(ok ? row : 0) * ld_src + (ok ? o : 0)];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool ok = full || (row < M && o < O);
        float v = acc[t][r];
        if (ACCUM) v += ex[r];
        if (SIG) {
          const float zv = PRELOAD ? sz[PRELOAD ? t : 0][r] : szt[r];
          v *= sigmoid_log2(fmaf(ssc[t], zv, ssh[t]));
          if (ok) { csum[t] += v; csz[t] = fmaf(v, zv, csz[t]); }
        }
        if (ok) out[row * ldo + o] = v;
      }
    }
  }
  if (SIG && col_part) {
    // per-workgroup partial sums -> [gridDim.x][2][128], combined through LDS in fixed order
    __syncthreads();
    float* s_red = reinterpret_cast<float*>(s_mem);        // [WAVES][256]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float s = csum[t] + __shfl_xor(csum[t], 32);
      const float q = csz[t] + __shfl_xor(csz[t], 32);
      if (h == 0) { s_red[wave * 256 + t * 32 + col] = s; s_red[wave * 256 + 128 + t * 32 + col] = q; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += WG) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += s_red[w * 256 + i];
      col_part[(size_t)blockIdx.x * 256 + i] = v;
    }
  }
}

// Output heads (conv8*, N8 <= 4 columns): G[m,k] = (sum_n g[m,n] W8[n,k]) softplus'(scale_k z[m,k] +
// shift_k) for the 128 columns of the head's last hidden layer, plus the two column sums.
// WG: the head's own weight gradient rides along (it needs the same g and z rows): per-workgroup
// partial sums of dW8[n,k] = sum_m g[m,n] softplus(u[m,k]) and db8[n] = sum_m g[m,n], in the layout of
// ganet_wgrad_act's workspace ([block][N8*128 + N8]) for ganet_wgrad_reduce_batch.
template <bool WG>
__global__ void __launch_bounds__(256)
head_bwd_kernel(int64_t M, int N8, const float* __restrict__ g, const float* __restrict__ W8,
                const float* __restrict__ z, int64_t ldz, const float* __restrict__ scale,
                const float* __restrict__ shift, float* __restrict__ G, int64_t ldG,
                float* __restrict__ col_part, float* __restrict__ wgrad_part) {
  __shared__ float4 s_red[2][8][32];
  const int cg = threadIdx.x & 31, rsub = threadIdx.x >> 5;
  float4 w[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
    w[n] = n < N8 ? *reinterpret_cast<const float4*>(W8 + (size_t)n * 128 + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 sc = *reinterpret_cast<const float4*>(scale + 4 * cg);
  float4 sh = *reinterpret_cast<const float4*>(shift + 4 * cg);
  sc.x *= kLog2e; sc.y *= kLog2e; sc.z *= kLog2e; sc.w *= kLog2e;
  sh.x *= kLog2e; sh.y *= kLog2e; sh.z *= kLog2e; sh.w *= kLog2e;
  float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgz = sg;
  float4 dw[WG ? 4 : 1];
  float dbv[WG ? 4 : 1];
#pragma unroll
  for (int n = 0; n < (WG ? 4 : 1); ++n) { dw[n] = make_float4(0.f, 0.f, 0.f, 0.f); dbv[n] = 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * 8 + rsub; row < M; row += (int64_t)gridDim.x * 8) {
    float gv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) gv[n] = n < N8 ? g[row * N8 + n] : 0.f;
    const float4 zv = *reinterpret_cast<const float4*>(z + row * ldz + 4 * cg);
    float4 d;
    d.x = gv[0] * w[0].x + gv[1] * w[1].x + gv[2] * w[2].x + gv[3] * w[3].x;
    d.y = gv[0] * w[0].y + gv[1] * w[1].y + gv[2] * w[2].y + gv[3] * w[3].y;
    d.z = gv[0] * w[0].z + gv[1] * w[1].z + gv[2] * w[2].z + gv[3] * w[3].z;
    d.w = gv[0] * w[0].w + gv[1] * w[1].w + gv[2] * w[2].w + gv[3] * w[3].w;
    if (WG) {
      // sigmoid and softplus of the same argument share the exponential:
      // e = 2^-|u|, t = 1 + e: softplus/ln2 = max(u,0) + log2(t); sigmoid = (u >= 0 ? 1 : e) / t
      float u[4] = {fmaf(sc.x, zv.x, sh.x), fmaf(sc.y, zv.y, sh.y), fmaf(sc.z, zv.z, sh.z), fmaf(sc.w, zv.w, sh.w)};
      float sp[4], sgm[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(u[c]));
        const float t = 1.0f + e;
        sp[c] = __builtin_fmaxf(u[c], 0.0f) + __builtin_amdgcn_logf(t);
        sgm[c] = (u[c] >= 0.f ? 1.0f : e) * __builtin_amdgcn_rcpf(t);
      }
      d.x *= sgm[0]; d.y *= sgm[1]; d.z *= sgm[2]; d.w *= sgm[3];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        dw[n].x = fmaf(gv[n], sp[0], dw[n].x); dw[n].y = fmaf(gv[n], sp[1], dw[n].y);
        dw[n].z = fmaf(gv[n], sp[2], dw[n].z); dw[n].w = fmaf(gv[n], sp[3], dw[n].w);
        dbv[n] += gv[n];
      }
    } else {
      d.x *= sigmoid_log2(fmaf(sc.x, zv.x, sh.x));
      d.y *= sigmoid_log2(fmaf(sc.y, zv.y, sh.y));
      d.z *= sigmoid_log2(fmaf(sc.z, zv.z, sh.z));
      d.w *= sigmoid_log2(fmaf(sc.w, zv.w, sh.w));
    }
    *reinterpret_cast<float4*>(G + row * ldG + 4 * cg) = d;
    sg.x += d.x; sg.y += d.y; sg.z += d.z; sg.w += d.w;
    sgz.x = fmaf(d.x, zv.x, sgz.x); sgz.y = fmaf(d.y, zv.y, sgz.y);
    sgz.z = fmaf(d.z, zv.z, sgz.z); sgz.w = fmaf(d.w, zv.w, sgz.w);
  }
  s_red[0][rsub][cg] = sg;
  s_red[1][rsub][cg] = sgz;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5;
    float4 a = s_red[which][0][cg];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      const float4 b = s_red[which][r][cg];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *reinterpret_cast<float4*>(col_part + (size_t)blockIdx.x * 256 + which * 128 + 4 * cg) = a;
  }
  if (WG) {
    // the 8 row groups of the workgroup combine through LDS, two head rows (n) per round; the softplus
    // was accumulated in log2 units: ln 2 is applied here
    float* part = wgrad_part + (size_t)blockIdx.x * (N8 * 128 + N8);
#pragma unroll
    for (int n0 = 0; n0 < 4; n0 += 2) {
      __syncthreads();
      s_red[0][rsub][cg] = dw[n0];
      s_red[1][rsub][cg] = dw[n0 + 1];
      __syncthreads();
      if (threadIdx.x < 64) {
        const int which = threadIdx.x >> 5;
        float4 a = s_red[which][0][cg];
#pragma unroll
        for (int r = 1; r < 8; ++r) {
          const float4 b = s_red[which][r][cg];
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const int n = n0 + which;
        if (n < N8) {
          float* o = part + n * 128 + 4 * cg;      // rows of 128 floats: 16-byte aligned only if N8 % 4 == 0
          o[0] = a.x * kLn2; o[1] = a.y * kLn2; o[2] = a.z * kLn2; o[3] = a.w * kLn2;
        }
      }
    }
    __syncthreads();
    float* s_db = reinterpret_cast<float*>(s_red);
    if (cg == 0) {
#pragma unroll
      for (int n = 0; n < 4; ++n) s_db[rsub * 4 + n] = dbv[n];
    }
    __syncthreads();
    if ((int)threadIdx.x < N8) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) a += s_db[r * 4 + threadIdx.x];
      part[N8 * 128 + threadIdx.x] = a;
    }
  }
}

// One workgroup per column: sum the partials in double, then the BatchNorm-backward coefficients
// (A, q, p) the consumers apply, and d gamma / d beta.
__global__ void __launch_bounds__(256)
bwd_stats_kernel(int nparts, int64_t M, const float* __restrict__ col_part,
                 const float* __restrict__ mean, const float* __restrict__ rstd,
                 const float* __restrict__ scale, float* __restrict__ coef,
                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double s_s[256], s_q[256];
  const int n = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 256) {
    s += (double)col_part[(size_t)p * 256 + n];
    q += (double)col_part[(size_t)p * 256 + 128 + n];
  }
  s_s[threadIdx.x] = s; s_q[threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { s_s[threadIdx.x] += s_s[threadIdx.x + o]; s_q[threadIdx.x] += s_q[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double sumG = s_s[0], sumGz = s_q[0];
    const double mu = mean[n], rs = rstd[n], sc = scale[n];
    const double dg = rs * (sumGz - mu * sumG);          // sum_m G xhat
    const double c1 = sumG / (double)M, c2 = dg / (double)M;
    coef[n] = (float)sc;
    coef[128 + n] = (float)(-sc * c2 * rs);
    coef[256 + n] = (float)(-sc * c1 + sc * c2 * rs * mu);
    dgamma[n] = (float)dg;
    dbeta[n] = (float)sumG;
  }
}

}  // namespace

}  // namespace ganet

using namespace ganet;

// Row sweep of the data-gradient kernels: -1 (default) = follow the call's row_order (GANET_ROWS_DOWN -> last row
// first), 0 / 1 = force up / down (dev switch: ganet_dev_set_reverse_bwd, or GANET_BWD_SWEEP=up|down in the environment)
int g_reverse_bwd = -2;
extern "C" void ganet_dev_set_reverse_bwd(int r) { g_reverse_bwd = r; }
static int bwd_reverse(int row_order) {
  if (g_reverse_bwd == -2) {
    const char* e = getenv("GANET_BWD_SWEEP");
    g_reverse_bwd = !e ? -1 : (!strcmp(e, "up") ? 0 : (!strcmp(e, "down") ? 1 : -1));
  }
  return g_reverse_bwd >= 0 ? g_reverse_bwd : (row_order == 2 ? 1 : 0);
}

extern "C" {

int32_t ganet_mlp_bwd_data_parts(void) { return BWD_BLOCKS; }
int32_t ganet_mlp_head_bwd_parts(void) { return HEAD_BLOCKS; }

int ganet_mlp_bwd_data(int64_t M, int32_t O, const float* g, int64_t ldg, const float* gz, int64_t ldgz,
                       const float* gcoef, const float* W, int64_t ldw, float* out, int64_t ldo,
                       int32_t accumulate,
                       const float* src_z, int64_t ld_src, const float* src_scale,
                       const float* src_shift, float* col_part, int32_t row_order, void* stream_) {
  const bool sig = src_z != nullptr;
  if (M <= 0 || O <= 0 || O > 128 || !g || !gz || !gcoef || !W || ldw < O || !out || ldo < O || (ldg % 4) ||
      (ldgz % 4) || ldg < 128 || ldgz < 128 || !aligned16(g) || !aligned16(gz) || !aligned16(gcoef) ||
      (sig && (!src_scale || !src_shift || !col_part || ld_src < O))) {
    set_error("ganet_mlp_bwd_data: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (mfma_mode() == 1) {
    const int rc = mlp_bwd_split(M, O, g, ldg, gz, ldgz, gcoef, W, ldw, out, ldo, accumulate != 0, src_z, ld_src,
                                 src_scale, src_shift, col_part, bwd_reverse(row_order), stream);
    if (rc >= 0) return rc;
  }
  const dim3 grid(BWD_BLOCKS), block(WG);
  const int nt = O > 96 ? 4 : 3;
#define LAUNCH(T, AC, SG)                                                                          \
  do {                                                                                             \
    const size_t lds = ((size_t)(T) * 32 * 33 + 96) * sizeof(float4);                              \
    static bool attr_set = false;                                                                  \
    if (!attr_set) {                                                                               \
      if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<T, AC, SG>),  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),     \
                    "hipFuncSetAttribute")) return 3;                                              \
      attr_set = true;                                                                             \
    }                                                                                              \
    ProfScope prof_(K_BWD_DATA, stream);                                                           \
    hipLaunchKernelGGL((mlp_bwd_kernel<T, AC, SG>), grid, block, lds, stream, M, O, g, ldg, gz, ldgz, \
                       gcoef, W, ldw, out, ldo, src_z, ld_src, src_scale, src_shift, col_part,     \
                       bwd_reverse(row_order));                                                    \
  } while (0)
  const bool acc = accumulate != 0;
  if (nt == 4 && !acc && sig) LAUNCH(4, false, true);
  else if (nt == 4 && !acc && !sig) LAUNCH(4, false, false);
  else if (nt == 4 && acc && !sig) LAUNCH(4, true, false);
  else if (nt == 4 && acc && sig) LAUNCH(4, true, true);
  else if (nt == 3 && !acc && !sig) LAUNCH(3, false, false);
  else if (nt == 3 && acc && !sig) LAUNCH(3, true, false);
  else {
    set_error("ganet_mlp_bwd_data: unsupported combination O=%d accumulate=%d sig=%d", O, (int)acc,
              (int)sig);
    return 4;
  }
#undef LAUNCH
  return check_hip(hipGetLastError(), "mlp_bwd_kernel");
}

int ganet_mlp_head_bwd(int64_t M, int32_t N8, const float* g, const float* W8, const float* z,
                       int64_t ldz, const float* scale, const float* shift, float* G, int64_t ldG,
                       float* col_part, float* wgrad_part, void* stream_) {
  if (M <= 0 || N8 <= 0 || N8 > 4 || !g || !W8 || !z || !scale || !shift || !G || !col_part ||
      (ldz % 4) || (ldG % 4) || ldz < 128 || ldG < 128 || !aligned16(W8) || !aligned16(z) ||
      !aligned16(G) || !aligned16(scale) || !aligned16(shift) || !aligned16(col_part)) {
    set_error("ganet_mlp_head_bwd: invalid arguments");
    return 1;
  }
  ProfScope prof_(K_HEAD_BWD, static_cast<hipStream_t>(stream_));
  if (wgrad_part)
    hipLaunchKernelGGL(head_bwd_kernel<true>, dim3(HEAD_BLOCKS), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), M, N8, g, W8, z, ldz, scale, shift, G, ldG,
                       col_part, wgrad_part);
  else
    hipLaunchKernelGGL(head_bwd_kernel<false>, dim3(HEAD_BLOCKS), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), M, N8, g, W8, z, ldz, scale, shift, G, ldG,
                       col_part, nullptr);
  return check_hip(hipGetLastError(), "head_bwd_kernel");
}

int ganet_mlp_bwd_stats(int64_t M, int32_t nparts, const float* col_part, const float* mean,
                        const float* rstd, const float* scale, float* coef, float* dgamma,
                        float* dbeta, void* stream_) {
  if (M <= 0 || nparts <= 0 || !col_part || !mean || !rstd || !scale || !coef || !dgamma || !dbeta) {
    set_error("ganet_mlp_bwd_stats: invalid arguments");
    return 1;
  }
  ProfScope prof_(K_BWD_STATS, static_cast<hipStream_t>(stream_));
  hipLaunchKernelGGL(bwd_stats_kernel, dim3(128), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     nparts, M, col_part, mean, rstd, scale, coef, dgamma, dbeta);
  return check_hip(hipGetLastError(), "bwd_stats_kernel");
}

}  // extern "C"
