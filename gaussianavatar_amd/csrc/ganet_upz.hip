// ganet_upz.hip — the decoder's two input GEMMs commuted with the bilinear up-sampling.
//
// The reference evaluates the decoder on x[m] = [ grid_sample(f)[m] | uv[m] ] for every texel m of the S x S query
// map (/root/reference/model/network.py:60-81): the first layer (conv1) and the input half of the skip layer (conv5)
// are 1 x 1 convolutions of that 66-column row (/root/reference/model/modules.py:555,559). Bilinear sampling is linear
// in the feature map f [R x R x 64] (R = S / 4), so
//
//     W . x[m] = sum_taps w_t (W_f f)[src_t(m)] + W_uv uv[m]            W = [ W_f (64 columns) | W_uv (2 columns) ]
//
// i.e. the GEMM runs ONCE at the feature map's resolution (R^2 = 16,384 rows instead of S^2 = 262,144: `rowgemm`), and
// the up-sampling gathers 128-column rows of P = f . [W1_f | W5_f]^T. The up-sampled input tensor x [M,72] (75 MB), its
// gradient, the two KIN = 72 GEMM launches each way and the separate up-sampling kernels disappear from the iteration:
//
//   forward   rowgemm                 P [frames R^2, 256] = f [., 64] . Wf^T               (0.5 GFLOP, L2-resident)
//             upsample_z_fwd          z1 [M,128] = bilinear(P[:, 0:128]) + W1_uv uv + b1, with conv1's BatchNorm column
//                                     sums in the epilogue: one pass over the 134 MB it writes
//             (skip layer)            layer_fwd_spec_kernel<1, true> (ganet_layer_fwd.hip) adds bilinear(P[:, 128:256])
//                                     + W5_uv uv to its output tile: a plain 128 -> 128 layer launch
//   backward  dz_upsample_t           dP[p, q, :] = sum over the texels whose taps hit (p, q) of w . dz[m, :], with
//                                     dz = A G + q Z + p assembled on load (BatchNorm backward folded, ganet.h): the
//                                     transposed up-sampling of a tensor that is never stored; the same sweep
//                                     accumulates dW_uv = sum dz uv^T and db = sum dz
//             rowgemm                 df [., 64] = dP [., 256] . Wf
//             ganet_wgrad_act         dWf [256, 64] = dP^T f (reduction over the 16,384 map pixels)
//
// All arithmetic fp32 (the small GEMMs on the fp32 matrix instruction, v_mfma_f32_32x32x2_f32).
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"

namespace ganet {

namespace {

// ---------------------------------------------------------------------------------------------------------------
// C[M,N] (+)= A[M,K] . Bt[N,K]^T     M, N multiples of 32, K a multiple of 64; operands 16-byte aligned rows.
// One wave per 32 x 32 tile of C. MFMA step s of a 64-wide k chunk pairs k = s (lanes 0-31) with k = 32 + s (lanes
// 32-63) — the reduction order is free —, so a lane's operands are 32 CONSECUTIVE floats of its row: eight 16-byte
// loads per operand and chunk, every byte of the touched lines used. No LDS: the operands live in L2 (<= 17 MB).
template <bool SPLITK>
__global__ void __launch_bounds__(256)
rowgemm_kernel(int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ Bt,
               int64_t ldb, float* __restrict__ C, int64_t ldc, int accumulate) {
  // SPLITK: the four waves of a workgroup share ONE tile, wave w takes the 64-wide k chunks w, w + 4, ...; their
  // partial tiles are added through LDS (few tiles, long K: dL/dfeat = dP [., 256] . Wf has 1,024 tiles)
  __shared__ float s_part[SPLITK ? 3 : 1][SPLITK ? 16 : 1][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, r = lane & 31;
  const int tiles_n = N / 32;
  const int64_t tile = SPLITK ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wave;
  const int64_t rb = tile / tiles_n;
  const int cb = (int)(tile - rb * tiles_n);
  if (rb * 32 >= M) return;
  const float* ap = A + (rb * 32 + r) * lda + 32 * h;
  const float* bp = Bt + (int64_t)(cb * 32 + r) * ldb + 32 * h;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int k0 = SPLITK ? 64 * wave : 0; k0 < K; k0 += SPLITK ? 256 : 64) {
    float4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = *reinterpret_cast<const float4*>(ap + k0 + 4 * u);
      b[u] = *reinterpret_cast<const float4*>(bp + k0 + 4 * u);
    }
    __builtin_amdgcn_sched_barrier(0);      // all sixteen loads in flight before the first MFMA waits for one
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
  }
  if (SPLITK) {
    if (wave > 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) s_part[wave - 1][q][lane] = acc[q];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] += (s_part[0][q][lane] + s_part[1][q][lane]) + s_part[2][q][lane];
  }
  // C/D layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* cp = C + (rb * 32 + 4 * h) * ldc + cb * 32 + r;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2);
    float v = acc[q];
    if (accumulate) v += cp[(int64_t)row * ldc];
    cp[(int64_t)row * ldc] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// z[m, 0:128] = sum_{a,b<2} rw[i,a] cw[j,b] P[f, ri[i,a], ci[j,b], 0:128] + Wuv[:, 0] u_m + Wuv[:, 1] v_m + bias
// for texel m = (f, i, j); column sums of (z - shift) and (z - shift)^2 per workgroup (what ganet_mlp_stats reads:
// [256 workgroups][2][128]). A wave takes two consecutive texels per step (lane = (texel, four columns)): the four
// taps are four 16-byte loads from the L2-resident P, the result leaves as one 16-byte store — 1 KB contiguous per wave.
constexpr int UPZ_BLOCKS = 256;       // = rows of the column-sum partials (FWD_BLOCKS of ganet_mlp.hip)
constexpr int UPZ_WG = 1024;

__global__ void __launch_bounds__(UPZ_WG)
upsample_z_fwd_kernel(UpGrid g, const float* __restrict__ P, int64_t ldp, const float* __restrict__ Wuv,
                      const float* __restrict__ bias, const float* __restrict__ stat_shift, float* __restrict__ z,
                      float* __restrict__ col_part) {
  __shared__ float s_red[UPZ_WG / 64][2][128];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int u = lane >> 5, c4 = lane & 31;
  const int S = g.S, R = g.R;
  const int64_t SS = (int64_t)S * S;
  const int64_t M = (int64_t)g.frames * SS;
  // this lane's four columns: uv weights, bias, statistics shift
  float wu[4], wv[4], bs[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    wu[e] = Wuv[(4 * c4 + e) * 2];
    wv[e] = Wuv[(4 * c4 + e) * 2 + 1];
    bs[e] = bias ? bias[4 * c4 + e] : 0.f;
    sh[e] = stat_shift ? stat_shift[4 * c4 + e] : 0.f;
  }
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t npair = (M + 1) / 2;
  const int64_t stride = (int64_t)gridDim.x * (UPZ_WG / 64);
  for (int64_t t = (int64_t)blockIdx.x * (UPZ_WG / 64) + wave; t < npair; t += stride) {
    const int64_t m = 2 * t + u;
    const int64_t mc = m < M ? m : M - 1;
    const int f = (int)(mc / SS);
    const int rem = (int)(mc - f * SS);
    const int i = rem / S, j = rem - i * S;
    const int2 ri = *reinterpret_cast<const int2*>(g.row_idx + 2 * i);
    const float2 rw = *reinterpret_cast<const float2*>(g.row_w + 2 * i);
    const int2 ci = *reinterpret_cast<const int2*>(g.col_idx + 2 * j);
    const float2 cw = *reinterpret_cast<const float2*>(g.col_w + 2 * j);
    const float2 uvv = *reinterpret_cast<const float2*>(g.uv + (int64_t)f * g.uv_frame_stride + (int64_t)rem * 2);
    const float* Pf = P + (int64_t)f * R * R * ldp + 4 * c4;
    const float4 v00 = *reinterpret_cast<const float4*>(Pf + ((int64_t)ri.x * R + ci.x) * ldp);
    const float4 v01 = *reinterpret_cast<const float4*>(Pf + ((int64_t)ri.x * R + ci.y) * ldp);
    const float4 v10 = *reinterpret_cast<const float4*>(Pf + ((int64_t)ri.y * R + ci.x) * ldp);
    const float4 v11 = *reinterpret_cast<const float4*>(Pf + ((int64_t)ri.y * R + ci.y) * ldp);
    if (m >= M) continue;
    // same association as the up-sampling of the input tensor (ganet_upsample.hip): columns first, then rows
    float o[4];
    o[0] = rw.x * (cw.x * v00.x + cw.y * v01.x) + rw.y * (cw.x * v10.x + cw.y * v11.x);
    o[1] = rw.x * (cw.x * v00.y + cw.y * v01.y) + rw.y * (cw.x * v10.y + cw.y * v11.y);
    o[2] = rw.x * (cw.x * v00.z + cw.y * v01.z) + rw.y * (cw.x * v10.z + cw.y * v11.z);
    o[3] = rw.x * (cw.x * v00.w + cw.y * v01.w) + rw.y * (cw.x * v10.w + cw.y * v11.w);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] += fmaf(wu[e], uvv.x, fmaf(wv[e], uvv.y, bs[e]));
      const float d = o[e] - sh[e];
      cs[e] += d;
      cq[e] = fmaf(d, d, cq[e]);
    }
    *reinterpret_cast<float4*>(z + m * 128 + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (!col_part) return;
  // the two texel halves of a wave, then the workgroup's waves (fixed order)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cs[e] += __shfl_xor(cs[e], 32);
    cq[e] += __shfl_xor(cq[e], 32);
  }
  if (u == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s_red[wave][0][4 * c4 + e] = cs[e];
      s_red[wave][1][4 * c4 + e] = cq[e];
    }
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int k = threadIdx.x >> 7, n = threadIdx.x & 127;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < UPZ_WG / 64; ++w) s += s_red[w][k][n];
    col_part[(size_t)blockIdx.x * 256 + k * 128 + n] = s;
  }
}

// The production form (S even): the per-pair kernel above as a software pipeline. What bounds that kernel is neither
// the HBM writes (a fill of the same 134 MB takes 23 us) nor its arithmetic, but a wave's serial chain "taps -> four
// gathers from the L2 / Infinity Cache -> arithmetic -> store" with ONE step in flight: 74 us. Here
//   * a step's taps are wave-uniform — the two texels of a pair differ only in the half-wave — so they come through the
//     SCALAR cache (row taps, the column taps and uv of both texels), requested a step ahead, and select by half-wave;
//   * the four 16-byte gathers of step n + 1 are issued before the arithmetic of step n (two register sets);
//   * 16 waves per CU.
// Measured on the way (512^2 map): eight texels per step, taps prefetched, 8 waves per CU: 50 us; the same with the
// rows of P parked in LDS per 4 x 4 patch (4x fewer gather bytes, 8 waves per CU at 190 registers): 71 us.
constexpr int UPQ_WG = 1024;

__global__ void __launch_bounds__(UPQ_WG)
upsample_z_fwd_pipe_kernel(UpGrid g, const float* __restrict__ P, int64_t ldp, const float* __restrict__ Wuv,
                           const float* __restrict__ bias, const float* __restrict__ stat_shift, float* __restrict__ z,
                           float* __restrict__ col_part) {
  __shared__ float s_red[UPQ_WG / 64][2][128];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int u = lane >> 5, c4 = lane & 31;
  const int S = g.S, R = g.R;
  const int gpr = S / 2;                                // pairs per texel row
  const int64_t npair = (int64_t)g.frames * S * gpr;
  float wu[4], wv[4], bs[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    wu[e] = Wuv[(4 * c4 + e) * 2];
    wv[e] = Wuv[(4 * c4 + e) * 2 + 1];
    bs[e] = bias ? bias[4 * c4 + e] : 0.f;
    sh[e] = stat_shift ? stat_shift[4 * c4 + e] : 0.f;
  }
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  // Work assignment: XCD x (= blockIdx % 8) owns a contiguous eighth of the texels, its workgroups contiguous ranges in
  // it, the waves of a workgroup consecutive pairs: the rows of P an XCD gathers are then a band of the map (1 MB of the
  // 8 MB at 512^2) that stays in ITS L2, each row re-used by the 16 texels under it. With the texels dealt round-robin
  // over the workgroups every XCD gathers from the WHOLE of P with a re-use distance of four texel rows: half of the
  // 537 MB of gathers then cross the fabric from the Infinity Cache, next to the 134 MB the kernel writes.
  const int nb = (int)gridDim.x;
  const int64_t lblock = (nb % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  const int64_t ppb = (npair + nb - 1) / nb;            // pairs per workgroup
  const int64_t first = lblock * ppb, last = first + ppb < npair ? first + ppb : npair;
  const int64_t stride = UPQ_WG / 64;
  const int64_t slot0 = first + wave;
  // everything a step needs that is wave-uniform (scalar registers)
  struct Geo { int64_t zoff, pb0, pb1; float a0, a1; int q0[2], q1[2]; float b0[2], b1[2], ux[2], uy[2]; };
  // (no division in the loop: a 64-bit divide is ~100 instructions, more than the step's own arithmetic; the wave's
  // position advances by `stride` pairs per step)
  int cur_f, cur_i, cur_j;                              // position of the NEXT geom() call (wave-uniform)
  {
    const int64_t pr0 = slot0 < last ? slot0 : (last > 0 ? last - 1 : 0);
    const int64_t row = pr0 / gpr;
    cur_j = (int)(pr0 - row * gpr) * 2;
    cur_f = (int)(row / S);
    cur_i = (int)(row - (int64_t)cur_f * S);
  }
  int64_t cur_pr = slot0;
  // the row-dependent part (taps of texel row i, bases of its two rows of P) changes once per texel row: 16 steps of a
  // wave at S = 512 — recomputed only then (the counters showed 92 scalar + 81 vector instructions per step, the SIMDs
  // 87 % busy issuing: this kernel is issue-bound, not memory-bound)
  int64_t row_pb0 = 0, row_pb1 = 0, row_z = 0, row_uv = 0;
  float row_a0 = 0.f, row_a1 = 0.f;
  auto row_setup = [&]() {
    const int f = cur_f, i = cur_i;
    const int p0 = g.row_idx[2 * i], p1 = g.row_idx[2 * i + 1];
    row_a0 = g.row_w[2 * i]; row_a1 = g.row_w[2 * i + 1];
    row_pb0 = ((int64_t)f * R + p0) * R * ldp;
    row_pb1 = ((int64_t)f * R + p1) * R * ldp;
    row_z = ((int64_t)f * S + i) * S * 128;
    row_uv = (int64_t)f * g.uv_frame_stride + (int64_t)i * S * 2;
  };
  row_setup();
  auto geom = [&]() {
    Geo o;
    const int j0 = cur_j;
    o.a0 = row_a0; o.a1 = row_a1; o.pb0 = row_pb0; o.pb1 = row_pb1;
    o.zoff = row_z + (int64_t)j0 * 128;
    const float* uvp = g.uv + row_uv + j0 * 2;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      o.q0[t] = g.col_idx[2 * (j0 + t)]; o.q1[t] = g.col_idx[2 * (j0 + t) + 1];
      o.b0[t] = g.col_w[2 * (j0 + t)]; o.b1[t] = g.col_w[2 * (j0 + t) + 1];
      o.ux[t] = uvp[2 * t]; o.uy[t] = uvp[2 * t + 1];
    }
    // advance (past the end of the range: stay on the last pair — a prefetch that is never used)
    if (cur_pr + stride < last) {
      cur_pr += stride;
      cur_j += 2 * (int)stride;
      if (cur_j >= S) {
        while (cur_j >= S) { cur_j -= S; if (++cur_i >= S) { cur_i = 0; ++cur_f; } }
        row_setup();
      }
    }
    return o;
  };
  struct Step { f32x4 v00, v01, v10, v11; float w00, w01, w10, w11, ux, uy; int64_t zoff; };
  auto issue = [&](Step& st, const Geo& o) {            // this lane's texel = j0 + u
    const int q0 = u ? o.q0[1] : o.q0[0], q1 = u ? o.q1[1] : o.q1[0];
    const float b0 = u ? o.b0[1] : o.b0[0], b1 = u ? o.b1[1] : o.b1[0];
    st.w00 = o.a0 * b0; st.w01 = o.a0 * b1; st.w10 = o.a1 * b0; st.w11 = o.a1 * b1;
    st.ux = u ? o.ux[1] : o.ux[0]; st.uy = u ? o.uy[1] : o.uy[0];
    st.zoff = o.zoff;
    const float* Pc = P + 4 * c4;
    st.v00 = *reinterpret_cast<const f32x4*>(Pc + o.pb0 + (int64_t)q0 * ldp);
    st.v01 = *reinterpret_cast<const f32x4*>(Pc + o.pb0 + (int64_t)q1 * ldp);
    st.v10 = *reinterpret_cast<const f32x4*>(Pc + o.pb1 + (int64_t)q0 * ldp);
    st.v11 = *reinterpret_cast<const f32x4*>(Pc + o.pb1 + (int64_t)q1 * ldp);
  };
  auto finish = [&](const Step& st) {
    float o[4];
    // (the four tap weights are multiplied out once per texel: one instruction per tap and channel)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = fmaf(st.w00, st.v00[e], fmaf(st.w01, st.v01[e], fmaf(st.w10, st.v10[e], st.w11 * st.v11[e])));
      o[e] += fmaf(wu[e], st.ux, fmaf(wv[e], st.uy, bs[e]));
      const float d = o[e] - sh[e];
      cs[e] += d;
      cq[e] = fmaf(d, d, cq[e]);
    }
    *reinterpret_cast<f32x4*>(z + st.zoff + (int64_t)u * 128 + 4 * c4) = f32x4{o[0], o[1], o[2], o[3]};
  };
  const int64_t nstep = slot0 < last ? (last - slot0 + stride - 1) / stride : 0;
  if (nstep > 0) {
    Step sa, sb2;
    Geo gn = geom();
    issue(sa, gn);
    gn = geom();
    for (int64_t n = 0; n < nstep; n += 2) {
      issue(sb2, gn);                                   // step n + 1's gathers
      gn = geom();
      __builtin_amdgcn_sched_barrier(0);
      finish(sa);
      __builtin_amdgcn_sched_barrier(0);
      if (n + 1 >= nstep) break;
      issue(sa, gn);                                    // step n + 2's gathers
      gn = geom();
      __builtin_amdgcn_sched_barrier(0);
      finish(sb2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (!col_part) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cs[e] += __shfl_xor(cs[e], 32);
    cq[e] += __shfl_xor(cq[e], 32);
  }
  if (u == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s_red[wave][0][4 * c4 + e] = cs[e];
      s_red[wave][1][4 * c4 + e] = cq[e];
    }
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int k = threadIdx.x >> 7, n = threadIdx.x & 127;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < UPQ_WG / 64; ++w) sum += s_red[w][k][n];
    col_part[(size_t)blockIdx.x * 256 + k * 128 + n] = sum;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Transposed up-sampling of dz = A G + q Z + p (never stored), one wave per map pixel (f, p, q), lane = two columns:
//   dP[f, p, q, n] = sum_{i in rows(p)} rw sum_{j in cols(q)} cw dz[(f, i, j), n]
// through the transposed tap lists (CSR). Every texel is read by the (<= 2 x 2) pixels its taps hit; a workgroup is
// 2 x 4 neighbouring pixels, so most of the sharing stays inside a CU and the rest in the L2 / Infinity Cache.
// The same sweep accumulates db[n] = sum_m dz[m,n] and dWuv[n, 0:2] = sum_m dz[m,n] uv[m, 0:2]: texel (i, j) is counted
// by the pixel that holds its FIRST (largest-weight) row and column tap. Partials per workgroup: [128 x 2 | 128] floats
// (= a ganet_wgrad_reduce_batch job with N = 128, K = 2).
constexpr int DZT_WG = 512;
constexpr int DZT_PART = 128 * 2 + 128;

__global__ void __launch_bounds__(DZT_WG)
dz_upsample_t_kernel(UpGrid g, const int32_t* __restrict__ rptr, const int32_t* __restrict__ rsrc,
                     const float* __restrict__ rwt, const int32_t* __restrict__ cptr, const int32_t* __restrict__ csrc,
                     const float* __restrict__ cwt, const float* __restrict__ G, const float* __restrict__ Z,
                     const float* __restrict__ coef, float* __restrict__ dP, int64_t ldp, float* __restrict__ partial) {
  __shared__ float s_red[DZT_WG / 64][6][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int pp = wave >> 2, qq = wave & 3;
  const int S = g.S, R = g.R;
  const int64_t SS = (int64_t)S * S;
  const float2 cA = *reinterpret_cast<const float2*>(coef + 2 * lane);
  const float2 cQ = *reinterpret_cast<const float2*>(coef + 128 + 2 * lane);
  const float2 cP = *reinterpret_cast<const float2*>(coef + 256 + 2 * lane);
  float sb0 = 0.f, sb1 = 0.f, su0 = 0.f, su1 = 0.f, sv0 = 0.f, sv1 = 0.f;
  const int pbn = (R + 1) / 2, qbn = (R + 3) / 4;
  const int64_t nitem = (int64_t)g.frames * pbn * qbn;
  for (int64_t item = blockIdx.x; item < nitem; item += gridDim.x) {
    const int f = (int)(item / ((int64_t)pbn * qbn));
    const int rem = (int)(item - (int64_t)f * pbn * qbn);
    const int pb = rem / qbn, qb = rem - pb * qbn;
    const int p = 2 * pb + pp, q = 4 * qb + qq;
    if (p >= R || q >= R) continue;                    // (uniform per wave)
    const int r0 = rptr[p], r1 = rptr[p + 1], c0 = cptr[q], c1 = cptr[q + 1];
    const float* Gf = G + (int64_t)f * SS * 128 + 2 * lane;
    const float* Zf = Z + (int64_t)f * SS * 128 + 2 * lane;
    const float* uvf = g.uv + (int64_t)f * g.uv_frame_stride;
    float a0 = 0.f, a1 = 0.f;
    for (int ri = r0; ri < r1; ++ri) {
      const int i = rsrc[ri];
      const float rw = rwt[ri];
      const bool own_row = g.row_idx[2 * i] == p;
      const int64_t rowbase = (int64_t)i * S;
      for (int cb = c0; cb < c1; cb += 8) {
        // eight column taps (all of them for the x4 up-sampling of the reference) as one batch of independent loads
        float2 gv[8], zv[8];
        float w[8];
        int jj[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int ci = min(cb + t, c1 - 1);
          jj[t] = csrc[ci];
          w[t] = (cb + t < c1) ? cwt[ci] : 0.f;
          const int64_t m = rowbase + jj[t];
          gv[t] = *reinterpret_cast<const float2*>(Gf + m * 128);
          zv[t] = *reinterpret_cast<const float2*>(Zf + m * 128);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float d0 = fmaf(cA.x, gv[t].x, fmaf(cQ.x, zv[t].x, cP.x));
          const float d1 = fmaf(cA.y, gv[t].y, fmaf(cQ.y, zv[t].y, cP.y));
          const float ww = rw * w[t];
          a0 = fmaf(ww, d0, a0);
          a1 = fmaf(ww, d1, a1);
          if (own_row && cb + t < c1 && g.col_idx[2 * jj[t]] == q) {      // wave-uniform
            const float2 uvv = *reinterpret_cast<const float2*>(uvf + (rowbase + jj[t]) * 2);
            sb0 += d0; sb1 += d1;
            su0 = fmaf(d0, uvv.x, su0); su1 = fmaf(d1, uvv.x, su1);
            sv0 = fmaf(d0, uvv.y, sv0); sv1 = fmaf(d1, uvv.y, sv1);
          }
        }
      }
    }
    *reinterpret_cast<float2*>(dP + (((int64_t)f * R + p) * R + q) * ldp + 2 * lane) = make_float2(a0, a1);
  }
  if (!partial) return;
  s_red[wave][0][lane] = sb0; s_red[wave][1][lane] = sb1;
  s_red[wave][2][lane] = su0; s_red[wave][3][lane] = su1;
  s_red[wave][4][lane] = sv0; s_red[wave][5][lane] = sv1;
  __syncthreads();
  if (threadIdx.x < 6 * 64) {
    const int k = threadIdx.x >> 6, l = threadIdx.x & 63;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < DZT_WG / 64; ++w) s += s_red[w][k][l];
    const int n = 2 * l + (k & 1);
    float* out = partial + (size_t)blockIdx.x * DZT_PART;
    if (k < 2) out[256 + n] = s;                 // db
    else out[n * 2 + ((k >> 1) - 1)] = s;        // dWuv[n][0] (u), dWuv[n][1] (v)
  }
}

// The production form of the same reduction ("march"): a workgroup owns a tile of DZM_P x DZM_Q map pixels, i.e. a
// block of (at most) DZM_ROWS x DZM_COLS texels, and walks through its texel rows, every element loaded ONCE:
//   * load phase — all 512 threads: thread (column slot, float4 of channels) loads G and Z of its texels of the row with
//     16-byte loads two rows ahead, forms dz = A G + q Z + p, adds it to its db / dW_uv sums when the texel is the
//     tile's own (first row tap and first column tap inside the tile: every texel counted by exactly one thread of
//     the launch) and writes it to the row image in LDS;
//   * reduce phase — wave w = pixel column w of the tile: h = sum_j cw dz[j] over its column taps (ds_read_b64 from the
//     row image), then the row's two vertical taps add rw . h into the wave's own slice of an LDS accumulator
//     [DZM_P][wave][128], written to dP once at the end.
// One barrier per row, row image double-buffered. Against one wave per pixel (dz_upsample_t_kernel: every element
// fetched by up to 2 x 2 waves of different workgroups — measured 104 us per launch at 512^2, 1.9x the algorithmic
// traffic) the only re-reads left are a tile's halo (2 texel rows / columns per side), and they are arranged to be L2
// hits: the workgroup order keeps neighbouring tiles on ONE XCD (blockIdx % 8 selects the XCD: logical tile = (blockIdx
// % 8) * (tiles / 8) + blockIdx / 8) and vertically neighbouring tiles sweep their rows in OPPOSITE directions, so that
// both reach the rows they share at the same time.
#ifndef GANET_DZM_ABLATE
#define GANET_DZM_ABLATE 0     // development (tools/dzm_ablate.sh): 1 no reduce phase, 2 no staging (loads only), 4 no Z loads,
#endif                         // 8 no barriers, 16 natural tile order and sweep direction, 32 no db / dW_uv sums
#ifndef GANET_DZM_WG
#define GANET_DZM_WG 512
#endif
#ifndef GANET_DZM_P
#define GANET_DZM_P 4
#endif
constexpr int DZM_WG = GANET_DZM_WG;   // 512 threads x 4 pixel rows per tile (two workgroups per CU, so that one computes while the other
                                       // waits at its row barrier). Measured on one box (512^2 map, median us per launch): 512 / 4: 80.8;
                                       // 1024 / 8 (16 waves, two load rounds per row, the second half empty): 92.9; 512 / 8: ~9 us
                                       // slower than 512 / 4 — the per-row arithmetic ADDS to the load time either way (ablation
                                       // builds: loads only 56-62 us). (A first 1024 / 8 figure of 72 us was of a build that loaded
                                       // 32 of a row's 48 columns: wrong results, caught by tests/test_decoder_map_gpu.py.)
constexpr int DZM_Q = 8;
constexpr int DZM_P = GANET_DZM_P;
constexpr int DZM_COLS = 48;           // texel columns a tile may span; x4 grid: 36
constexpr int DZM_SLOTS = DZM_WG / 32; // column slots of the load phase (x 32 float4 of channels)
constexpr int DZM_NR = (DZM_COLS + DZM_SLOTS - 1) / DZM_SLOTS;      // load rounds per row (the last one may be partial: jon[])
constexpr int DZM_NW = DZM_WG / 64;
constexpr int DZM_CPL = 128 / (64 * (DZM_NW / DZM_Q));   // channels per lane in the reduce phase (8 waves: 2, 16 waves: 1)
static_assert(DZM_NW == 8 || DZM_NW == 16, "reduce-phase mapping");

__global__ void __attribute__((amdgpu_flat_work_group_size(DZM_WG, DZM_WG), amdgpu_waves_per_eu(4, 4)))
dz_upsample_march_kernel(UpGrid g, const int32_t* __restrict__ rptr, const int32_t* __restrict__ rsrc,
                         const int32_t* __restrict__ cptr, const int32_t* __restrict__ csrc,
                         const float* __restrict__ cwt, const float* __restrict__ G, const float* __restrict__ Z,
                         const float* __restrict__ coef, float* __restrict__ dP, int64_t ldp, float* __restrict__ partial) {
  __shared__ float s_dz[2][DZM_COLS][128];            // 48 KB
  __shared__ float s_acc[DZM_P][DZM_Q][128];          // 16 KB (P = 4)
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c4 = tid & 31, slot = tid >> 5;            // load phase: channels 4 c4 .. + 3 of texel columns slot, slot + SLOTS, ...
  const int wq = wave & 7, ch = DZM_CPL * (64 * (wave >> 3) + lane);    // reduce phase: pixel column wq, channels ch .. + CPL - 1
  const int S = g.S, R = g.R;
  const int64_t SS = (int64_t)S * S;
  const int nchunk = (R + DZM_P - 1) / DZM_P, nstrip = (R + DZM_Q - 1) / DZM_Q;
  const int total = (int)gridDim.x;
  // XCD-aware order (total a multiple of 8): consecutive logical tiles share an XCD
  const int logical = (total % 8 == 0 && !(GANET_DZM_ABLATE & 16))
                          ? (int)(blockIdx.x % 8) * (total / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int strip = logical % nstrip;
  const int chunk = (logical / nstrip) % nchunk;
  const int f = logical / (nstrip * nchunk);
  const int q0 = strip * DZM_Q, q1 = min(R, q0 + DZM_Q);
  const int pc0 = chunk * DZM_P, pc1 = min(R, pc0 + DZM_P);
  const int q = q0 + wq;
#pragma unroll
  for (int p = 0; p < DZM_P; ++p)
#pragma unroll
    for (int e = 0; e < DZM_CPL; ++e) s_acc[p][wq][ch + e] = 0.f;
  float sb[4] = {0.f, 0.f, 0.f, 0.f}, su[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {0.f, 0.f, 0.f, 0.f};
  const int e0 = rptr[pc0], e1 = rptr[pc1], d0 = cptr[q0], d1 = cptr[q1];
  if (e1 > e0 && d1 > d0) {                            // (uniform) the tile has taps at all
    const int i_lo = rsrc[e0], i_hi = rsrc[e1 - 1];    // (lists ascending in the texel index)
    const int j_lo = csrc[d0], j_hi = csrc[d1 - 1];    // <= j_lo + DZM_COLS - 1 (GanetUpGrid.max_col_span)
    const int nrow = i_hi - i_lo + 1;
    const bool up = (chunk & 1) != 0 && !(GANET_DZM_ABLATE & 16);      // odd chunks sweep bottom-up
    auto row_at = [&](int k) { return up ? i_hi - min(k, nrow - 1) : i_lo + min(k, nrow - 1); };
    // this thread's texel columns and whether the tile owns them (first column tap inside the strip)
    int jc[DZM_NR];
    bool jon[DZM_NR];
    float jown[DZM_NR];
#pragma unroll
    for (int r = 0; r < DZM_NR; ++r) {
      const int j = j_lo + slot + DZM_SLOTS * r;
      jon[r] = j <= j_hi;
      jc[r] = jon[r] ? j : j_hi;
      const int qa = g.col_idx[2 * jc[r]];
      jown[r] = (jon[r] && qa >= q0 && qa < q1 && !(GANET_DZM_ABLATE & 32)) ? 1.f : 0.f;
    }
    const float4 cA = *reinterpret_cast<const float4*>(coef + 4 * c4);
    const float4 cQ = *reinterpret_cast<const float4*>(coef + 128 + 4 * c4);
    const float4 cP = *reinterpret_cast<const float4*>(coef + 256 + 4 * c4);
    const float* Gf = G + (int64_t)f * SS * 128 + 4 * c4;
    const float* Zf = Z + (int64_t)f * SS * 128 + 4 * c4;
    const float* uvf = g.uv + (int64_t)f * g.uv_frame_stride;
    // reduce phase: this wave's column taps (the first 8 in registers), as offsets into the row image
    const bool q_on = q < q1;
    const int c0 = q_on ? cptr[q] : 0, c1 = q_on ? cptr[q + 1] : 0;
    int jt[8];
    float wt[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const bool on = c0 + t < c1;
      jt[t] = (on ? csrc[c0 + t] : j_lo) - j_lo;
      wt[t] = on ? cwt[c0 + t] : 0.f;
    }
    // (uv rides with the row's loads: a load issued in stage_row would have to wait for ALL loads in flight — the
    // counter is in order — i.e. for the prefetched rows)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    struct Row { f32x4 gv[DZM_NR], zv[DZM_NR]; f32x2 uv[DZM_NR]; };
    auto load_row = [&](Row& r, int k) {
      const int64_t rowbase = (int64_t)row_at(k) * S;
#pragma unroll
      for (int t = 0; t < DZM_NR; ++t) {
        r.gv[t] = *reinterpret_cast<const f32x4*>(Gf + (rowbase + jc[t]) * 128);
        if (!(GANET_DZM_ABLATE & 4)) r.zv[t] = *reinterpret_cast<const f32x4*>(Zf + (rowbase + jc[t]) * 128);
        else r.zv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        r.uv[t] = *reinterpret_cast<const f32x2*>(uvf + (rowbase + jc[t]) * 2);
      }
    };
    auto stage_row = [&](const Row& r, int k) {        // dz of this thread's texels -> row image k & 1, + its sums
      const int i = row_at(k);
      const int pa = g.row_idx[2 * i];                 // scalar load
      const float own_row = (pa >= pc0 && pa < pc1) ? 1.f : 0.f;
#pragma unroll
      for (int t = 0; t < DZM_NR; ++t) {
        f32x4 d;
        d.x = fmaf(cA.x, r.gv[t].x, fmaf(cQ.x, r.zv[t].x, cP.x));
        d.y = fmaf(cA.y, r.gv[t].y, fmaf(cQ.y, r.zv[t].y, cP.y));
        d.z = fmaf(cA.z, r.gv[t].z, fmaf(cQ.z, r.zv[t].z, cP.z));
        d.w = fmaf(cA.w, r.gv[t].w, fmaf(cQ.w, r.zv[t].w, cP.w));
        if (GANET_DZM_ABLATE & 2) { sb[0] += d.x + d.y + d.z + d.w; continue; }
        if (jon[t]) *reinterpret_cast<f32x4*>(&s_dz[k & 1][slot + DZM_SLOTS * t][4 * c4]) = d;
        // db / dW_uv, branch-free: the texel counts (m = 1) when the tile owns it
        const float m = own_row * jown[t], mu = m * r.uv[t].x, mv = m * r.uv[t].y;
        sb[0] = fmaf(m, d.x, sb[0]); sb[1] = fmaf(m, d.y, sb[1]); sb[2] = fmaf(m, d.z, sb[2]); sb[3] = fmaf(m, d.w, sb[3]);
        su[0] = fmaf(mu, d.x, su[0]); su[1] = fmaf(mu, d.y, su[1]); su[2] = fmaf(mu, d.z, su[2]); su[3] = fmaf(mu, d.w, su[3]);
        sv[0] = fmaf(mv, d.x, sv[0]); sv[1] = fmaf(mv, d.y, sv[1]); sv[2] = fmaf(mv, d.z, sv[2]); sv[3] = fmaf(mv, d.w, sv[3]);
      }
    };
    auto reduce_row = [&](int k) {                     // wave = (pixel column q[, channel half]): horizontal taps, vertical scatter
      if (!q_on || (GANET_DZM_ABLATE & 1)) return;
      const int i = row_at(k);
      const int pa = g.row_idx[2 * i], pb = g.row_idx[2 * i + 1];        // scalar loads
      const float wa = g.row_w[2 * i], wb = g.row_w[2 * i + 1];
      float h[DZM_CPL];
#pragma unroll
      for (int e = 0; e < DZM_CPL; ++e) h[e] = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < DZM_CPL; ++e) h[e] = fmaf(wt[t], s_dz[k & 1][jt[t]][ch + e], h[e]);
      for (int cb = c0 + 8; cb < c1; ++cb)             // (more than 8 column taps: not the x4 grid)
#pragma unroll
        for (int e = 0; e < DZM_CPL; ++e) h[e] = fmaf(cwt[cb], s_dz[k & 1][csrc[cb] - j_lo][ch + e], h[e]);
      if (wa != 0.f && pa >= pc0 && pa < pc1)
#pragma unroll
        for (int e = 0; e < DZM_CPL; ++e) s_acc[pa - pc0][wq][ch + e] = fmaf(wa, h[e], s_acc[pa - pc0][wq][ch + e]);
      if (wb != 0.f && pb >= pc0 && pb < pc1)
#pragma unroll
        for (int e = 0; e < DZM_CPL; ++e) s_acc[pb - pc0][wq][ch + e] = fmaf(wb, h[e], s_acc[pb - pc0][wq][ch + e]);
    };
    // rows k = 0 .. nrow - 1; loads one row ahead (two register sets): the other workgroup of the CU covers the rest
    Row r0, r1;
    load_row(r0, 0);
    auto do_row = [&](Row& cur, Row& refill, int k) {
      load_row(refill, k + 1);
      __builtin_amdgcn_sched_barrier(0);
      stage_row(cur, k);
      if (!(GANET_DZM_ABLATE & 8)) __syncthreads();
      reduce_row(k);
    };
    for (int k = 0; k < nrow; k += 2) {
      do_row(r0, r1, k);
      if (k + 1 < nrow) do_row(r1, r0, k + 1);
    }
  }
  if (q < q1) {
#pragma unroll
    for (int p = 0; p < DZM_P; ++p)
      if (pc0 + p < pc1)
#pragma unroll
        for (int e = 0; e < DZM_CPL; ++e) dP[(((int64_t)f * R + pc0 + p) * R + q) * ldp + ch + e] = s_acc[p][wq][ch + e];
  }
  if (!partial) return;
  // db / dWuv: SLOTS column slots x 32 channel quads -> [128 x 2 | 128] per workgroup, through the (now idle) row image
  __syncthreads();
  float* red = &s_dz[0][0][0];                         // [SLOTS][3][128] <= the image's 48 KB
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[(slot * 3 + 0) * 128 + 4 * c4 + e] = sb[e];
    red[(slot * 3 + 1) * 128 + 4 * c4 + e] = su[e];
    red[(slot * 3 + 2) * 128 + 4 * c4 + e] = sv[e];
  }
  __syncthreads();
  if (tid < 3 * 128) {
    const int k = tid >> 7, n = tid & 127;
    float sum = 0.f;
#pragma unroll
    for (int sl = 0; sl < DZM_SLOTS; ++sl) sum += red[(sl * 3 + k) * 128 + n];
    float* out = partial + (size_t)blockIdx.x * DZT_PART;
    if (k == 0) out[256 + n] = sum;                    // db
    else out[n * 2 + (k - 1)] = sum;                   // dWuv[n][0] (u), dWuv[n][1] (v)
  }
}

}  // namespace

// ---- internal launchers (ganet_decoder.hip sequences them; the extern "C" entries below are thin wrappers)
int rowgemm_launch(int64_t M, int N, int K, const float* A, int64_t lda, const float* Bt, int64_t ldb, float* C,
                   int64_t ldc, int accumulate, hipStream_t stream) {
  if (M <= 0 || (M % 32) || N <= 0 || (N % 32) || K <= 0 || (K % 64) || !A || !Bt || !C || (lda % 4) || (ldb % 4) ||
      lda < K || ldb < K || ldc < N || !aligned16(A) || !aligned16(Bt)) {
    set_error("ganet_rowgemm: invalid arguments (M, N multiples of 32, K a multiple of 64, 16-byte aligned rows)");
    return 1;
  }
  const int64_t tiles = (M / 32) * (N / 32);
  ProfScope prof_(K_ROWGEMM, stream);
  if (K >= 256 && tiles <= 4096)
    hipLaunchKernelGGL(rowgemm_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, stream, M, N, K, A, lda, Bt, ldb, C, ldc,
                       accumulate);
  else
    hipLaunchKernelGGL(rowgemm_kernel<false>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, M, N, K, A, lda, Bt,
                       ldb, C, ldc, accumulate);
  return check_hip(hipGetLastError(), "rowgemm_kernel");
}

static bool grid_ok(const UpGrid& g) {
  return g.frames > 0 && g.S > 0 && g.R > 0 && g.row_idx && g.row_w && g.col_idx && g.col_w && g.uv &&
         (reinterpret_cast<uintptr_t>(g.row_idx) & 7) == 0 && (reinterpret_cast<uintptr_t>(g.col_idx) & 7) == 0 &&
         (reinterpret_cast<uintptr_t>(g.row_w) & 7) == 0 && (reinterpret_cast<uintptr_t>(g.col_w) & 7) == 0 &&
         (reinterpret_cast<uintptr_t>(g.uv) & 7) == 0 && (g.uv_frame_stride % 2) == 0;
}

int upsample_z_fwd_launch(const UpGrid& g, const float* P, int64_t ldp, const float* Wuv, const float* bias,
                          const float* stat_shift, float* z, float* col_part, hipStream_t stream) {
  if (!grid_ok(g) || !P || (ldp % 4) || ldp < 128 || !aligned16(P) || !Wuv || !z || !aligned16(z)) {
    set_error("ganet_upsample_z_fwd: invalid arguments");
    return 1;
  }
  ProfScope prof_(K_UPZ_FWD, stream);
  if (g.S % 2 == 0)
    hipLaunchKernelGGL(upsample_z_fwd_pipe_kernel, dim3(UPZ_BLOCKS), dim3(UPQ_WG), 0, stream, g, P, ldp, Wuv, bias, stat_shift,
                       z, col_part);
  else
    hipLaunchKernelGGL(upsample_z_fwd_kernel, dim3(UPZ_BLOCKS), dim3(UPZ_WG), 0, stream, g, P, ldp, Wuv, bias, stat_shift, z,
                       col_part);
  return check_hip(hipGetLastError(), "upsample_z_fwd_kernel");
}

int dz_upsample_t_blocks(const UpGrid& g) {      // = workgroups of the march kernel = rows of its partial sums
  return g.frames * ((g.R + DZM_P - 1) / DZM_P) * ((g.R + DZM_Q - 1) / DZM_Q);
}

int dz_upsample_t_launch(const UpGrid& g, const GanetUpGrid& t, const float* G, const float* Z, const float* coef,
                         float* dP, int64_t ldp, float* partial, hipStream_t stream) {
  if (!grid_ok(g) || !t.row_ptr || !t.row_src || !t.row_wt || !t.col_ptr || !t.col_src || !t.col_wt || !G || !Z ||
      !coef || !dP || (ldp % 2) || ldp < 128 || (reinterpret_cast<uintptr_t>(dP) & 7) || !aligned16(G) ||
      !aligned16(Z) || (reinterpret_cast<uintptr_t>(coef) & 7)) {
    set_error("ganet_dz_upsample_t: invalid arguments");
    return 1;
  }
  ProfScope prof_(K_DZ_UPT, stream);
  if (t.max_col_span > 0 && t.max_col_span <= DZM_COLS)
    hipLaunchKernelGGL(dz_upsample_march_kernel, dim3(dz_upsample_t_blocks(g)), dim3(DZM_WG), 0, stream, g, t.row_ptr,
                       t.row_src, t.col_ptr, t.col_src, t.col_wt, G, Z, coef, dP, ldp, partial);
  else      // any separable grid: one wave per map pixel (its workgroups cover 2 x 4 pixels: the same partial-row count)
    hipLaunchKernelGGL(dz_upsample_t_kernel, dim3(dz_upsample_t_blocks(g)), dim3(DZT_WG), 0, stream, g, t.row_ptr, t.row_src,
                       t.row_wt, t.col_ptr, t.col_src, t.col_wt, G, Z, coef, dP, ldp, partial);
  return check_hip(hipGetLastError(), "dz_upsample_t kernels");
}

UpGrid up_grid_of(const GanetUpGrid* t) {
  UpGrid g{};
  if (!t) return g;
  g.frames = t->frames; g.S = t->S; g.R = t->R;
  g.row_idx = t->row_idx; g.row_w = t->row_w; g.col_idx = t->col_idx; g.col_w = t->col_w;
  g.uv = t->uv; g.uv_frame_stride = t->uv_frame_stride;
  return g;
}

}  // namespace ganet

using namespace ganet;

extern "C" {

int ganet_rowgemm(int64_t M, int32_t N, int32_t K, const float* A, int64_t lda, const float* Bt, int64_t ldb, float* C,
                  int64_t ldc, int32_t accumulate, void* stream) {
  return rowgemm_launch(M, N, K, A, lda, Bt, ldb, C, ldc, accumulate, static_cast<hipStream_t>(stream));
}

int ganet_upsample_z_fwd(const GanetUpGrid* grid, const float* P, int64_t ldp, const float* Wuv, const float* bias,
                         const float* stat_shift, float* z, float* col_part, void* stream) {
  if (!grid) { set_error("ganet_upsample_z_fwd: grid is NULL"); return 1; }
  return upsample_z_fwd_launch(up_grid_of(grid), P, ldp, Wuv, bias, stat_shift, z, col_part,
                               static_cast<hipStream_t>(stream));
}

int ganet_mlp_fwd_add(const GanetUpGrid* grid, const float* x2, const float* in_scale, const float* in_shift,
                      const float* W, const float* bias, const float* P, int64_t ldp, const float* Wuv, float* z,
                      float* col_part, const float* stat_shift, int32_t row_order, void* stream) {
  if (!grid || !x2 || !in_scale || !in_shift || !W || !z || !aligned16(x2) || !aligned16(W) || !aligned16(in_scale) ||
      !aligned16(in_shift) || !grid_ok(up_grid_of(grid))) {
    set_error("ganet_mlp_fwd_add: invalid arguments");
    return 1;
  }
  const FwdAddend add{up_grid_of(grid), P, ldp, Wuv};
  const int64_t M = (int64_t)grid->frames * grid->S * grid->S;
  const int rc = layer_fwd_spec_add(M, x2, in_scale, in_shift, W, bias, z, col_part, stat_shift, add,
                                    row_order == GANET_ROWS_DOWN ? 1 : 0, static_cast<hipStream_t>(stream));
  if (rc < 0) { set_error("ganet_mlp_fwd_add: unsupported shape (S a multiple of 32, 16-byte aligned P and z)"); return 4; }
  return rc;
}

int32_t ganet_dz_upsample_t_parts(const GanetUpGrid* grid) { return grid ? dz_upsample_t_blocks(up_grid_of(grid)) : 0; }

int ganet_dz_upsample_t(const GanetUpGrid* grid, const float* G, const float* Z, const float* coef, float* dP, int64_t ldp,
                        float* partial, void* stream) {
  if (!grid) { set_error("ganet_dz_upsample_t: grid is NULL"); return 1; }
  return dz_upsample_t_launch(up_grid_of(grid), *grid, G, Z, coef, dP, ldp, partial, static_cast<hipStream_t>(stream));
}

}  // extern "C"
