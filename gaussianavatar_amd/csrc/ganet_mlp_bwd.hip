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
//   * the data-gradient kernel (mlp_bwd_split_kernel, ganet_mlp_split.hip; ganet_layer_bwd.hip where the weight
//     gradient rides along) computes dL/dy_src = dz_i . W_i (A operand assembled from G_i and z_i on load) and, in its epilogue, multiplies by
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


// Output heads (conv8*, N8 <= 4 columns): G[m,k] = (sum_n g[m,n] W8[n,k]) softplus'(scale_k z[m,k] +
// shift_k) for the 128 columns of the head's last hidden layer, plus the two column sums.
// WG: the head's own weight gradient rides along (it needs the same g and z rows): per-workgroup
// partial sums of dW8[n,k] = sum_m g[m,n] softplus(u[m,k]) and db8[n] = sum_m g[m,n], in the layout of
// ganet_wgrad_act's workspace ([block][N8*128 + N8]) for ganet_wgrad_reduce_batch.
template <bool WG, int N8>      // N8: the head's width, a compile-time count (no branch around a load)
__global__ void __launch_bounds__(256)
head_bwd_kernel(int64_t M, const float* __restrict__ g, const float* __restrict__ W8,
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
  // U rows per thread and step with all their loads issued first: 512 workgroups x 8 rows x one 16-byte load were
  // 2 MB in flight chip-wide — a fraction of what the HBM latency needs (61 us per head; 268 MB = 43 us at 6.3 TB/s)
  constexpr int U = 4;
  // a workgroup's step = 8 U consecutive rows (16 KB of z), row group rsub taking U consecutive ones
  const int64_t rstep = 1;
  for (int64_t row0 = ((int64_t)blockIdx.x * 8 + rsub) * U; row0 < M; row0 += (int64_t)gridDim.x * 8 * U) {
    float gvs[U][4];
    float4 zvs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * rstep;
      const int64_t rr = row < M ? row : row0;
#pragma unroll
      for (int n = 0; n < 4; ++n) gvs[u][n] = n < N8 ? g[rr * N8 + n] : 0.f;     // a clamped row: no branch
      zvs[u] = *reinterpret_cast<const float4*>(z + rr * ldz + 4 * cg);
    }
    // every load above is issued before the first value is touched, and "touched" whatever `row < M` says below
    // (otherwise the compiler sinks the loads of a row into that row's branch, one round trip per row)
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int n = 0; n < N8; ++n) asm volatile("" : "+v"(gvs[u][n]));
      asm volatile("" : "+v"(zvs[u].x), "+v"(zvs[u].y), "+v"(zvs[u].z), "+v"(zvs[u].w));
      if (row0 + u * rstep >= M) {
#pragma unroll
        for (int n = 0; n < 4; ++n) gvs[u][n] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int64_t row = row0 + u * rstep;
    if (row >= M) break;
    const float (&gv)[4] = gvs[u];
    const float4 zv = zvs[u];
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

// One wave per column: sum the partials in double, then the BatchNorm-backward coefficients
// (A, q, p) the consumers apply, and d gamma / d beta.
struct BwdStatsJobs { BwdStatsJob j[kMaxStatsJobs]; };
__global__ void __launch_bounds__(64)
bwd_stats_kernel(int64_t M, BwdStatsJobs jobs) {
  const BwdStatsJob& J = jobs.j[blockIdx.y];
  const int n = blockIdx.x;
  const double mu = J.mean[n], rs = J.rstd[n], sc = J.scale[n];   // requested before the partials: one round trip, not two
  double sumG, sumGz;
  column_sums_wave(J.col_part, J.nparts, 256, 128, n, sumG, sumGz);
  if (threadIdx.x == 0) {
    const double dg = rs * (sumGz - mu * sumG);          // sum_m G xhat
    const double c1 = sumG / (double)M, c2 = dg / (double)M;
    J.coef[n] = (float)sc;
    J.coef[128 + n] = (float)(-sc * c2 * rs);
    J.coef[256 + n] = (float)(-sc * c1 + sc * c2 * rs * mu);
    J.dgamma[n] = (float)dg;
    J.dbeta[n] = (float)sumG;
  }
}

}  // namespace

int bwd_stats_launch(int njobs, const BwdStatsJob* jobs, int64_t M, hipStream_t stream) {
  if (njobs <= 0 || njobs > kMaxStatsJobs) { set_error("bwd_stats_launch: 1..%d jobs", kMaxStatsJobs); return 1; }
  BwdStatsJobs js;
  for (int i = 0; i < kMaxStatsJobs; ++i) js.j[i] = jobs[i < njobs ? i : 0];
  ProfScope prof_(K_BWD_STATS, stream);
  hipLaunchKernelGGL(bwd_stats_kernel, dim3(128, njobs), dim3(64), 0, stream, M, js);
  return check_hip(hipGetLastError(), "bwd_stats_kernel");
}

}  // namespace ganet

using namespace ganet;

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
  const int rc = mlp_bwd_split(M, O, g, ldg, gz, ldgz, gcoef, W, ldw, out, ldo, accumulate != 0, src_z, ld_src,
                               src_scale, src_shift, col_part, row_order == GANET_ROWS_DOWN ? 1 : 0, stream);
  if (rc >= 0) return rc;
  set_error("ganet_mlp_bwd_data: unsupported combination O=%d (64 < O <= 128) accumulate=%d src=%d", O,
            (int)(accumulate != 0), (int)sig);
  return 4;
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
#define LAUNCH(WG_, N_)                                                                                     \
  hipLaunchKernelGGL((head_bwd_kernel<WG_, N_>), dim3(HEAD_BLOCKS), dim3(256), 0, static_cast<hipStream_t>(stream_), \
                     M, g, W8, z, ldz, scale, shift, G, ldG, col_part, wgrad_part)
  if (wgrad_part) {
    if (N8 == 1) LAUNCH(true, 1); else if (N8 == 2) LAUNCH(true, 2); else if (N8 == 3) LAUNCH(true, 3); else LAUNCH(true, 4);
  } else {
    if (N8 == 1) LAUNCH(false, 1); else if (N8 == 2) LAUNCH(false, 2); else if (N8 == 3) LAUNCH(false, 3); else LAUNCH(false, 4);
  }
#undef LAUNCH
  return check_hip(hipGetLastError(), "head_bwd_kernel");
}

int ganet_mlp_bwd_stats(int64_t M, int32_t nparts, const float* col_part, const float* mean,
                        const float* rstd, const float* scale, float* coef, float* dgamma,
                        float* dbeta, void* stream_) {
  if (M <= 0 || nparts <= 0 || !col_part || !mean || !rstd || !scale || !coef || !dgamma || !dbeta) {
    set_error("ganet_mlp_bwd_stats: invalid arguments");
    return 1;
  }
  const BwdStatsJob job{col_part, nparts, mean, rstd, scale, coef, dgamma, dbeta};
  return bwd_stats_launch(1, &job, M, static_cast<hipStream_t>(stream_));
}

}  // extern "C"
