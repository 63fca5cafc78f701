// ganet_unet.hip — the stage-2 pose encoder (UnetNoCond5DS, /root/reference/model/modules.py:185-232; blocks :62-111)
// forward and backward as hand-written kernels: no im2col / col2im, no vendor GEMM, no torch element-wise launches.
//
//   down  conv_k = Conv2d(4x4, stride 2, pad 1, no bias) [+ BatchNorm2d(affine=False)]       k = 1..5, 128^2 -> 4^2
//   up    upconv_k = ReLU -> ConvTranspose2d(4x4, stride 2, pad 1) [+ BatchNorm2d] -> cat skip   k = 1..5, 4^2 -> 128^2
//   (the reference's LeakyReLU(0.2, inplace) in front of conv2..5 also acts on the skip tensors: a_k = leaky(bn(conv_k)))
//
// Everything is channels-last ([B,H,W,C]) and NOTHING normalised or activated is ever stored: a layer writes its raw
// convolution output z (+ the BatchNorm column sums of z out of the epilogue), and every consumer applies
// act(z * scale + shift) while it loads its operand — the decoder's scheme (ganet.h). The concatenations are virtual: a
// consumer's K dimension runs over two source tensors ("segments"). Backward likewise: a layer keeps
// Gy = dL/d(BatchNorm output) (written, x act', by the dgrad kernel of its consumer(s), with the sums of Gy and Gy.y^
// in that epilogue) and dz = cA Gy + cQ z + cP is assembled on load by the layer's own dgrad / wgrad kernels.
//
// Two gather patterns cover all four convolution flavours (c = coarse grid [B,Hc,Wc], f = fine grid [B,2Hc,2Wc]):
//   S  out on c, 16 taps (ky,kx) from f at (2y + ky - 1, 2x + kx - 1)      conv forward, transposed-conv input gradient
//   T  out on f, per parity class of (y,x) 2 x 2 taps from c               transposed-conv forward, conv input gradient
// as one GEMM per output tile: a wave owns 32 pixels x 32 channels, the four waves of a workgroup split the taps and add
// their tiles through LDS; fp32 matrix instruction (v_mfma_f32_32x32x2_f32: exact fp32 products); lane (half, row)
// loads 16 consecutive channels of its pixel per 32-channel chunk (the reduction order over k is free). The weight
// gradient reduces over the coarse pixels with the pixel index as the MFMA's k: one dword per lane and operand per
// step, per-lane channel coefficients, deterministic partial tiles per 256-pixel chunk.
// The maps are tiny (64^2 .. 4^2 pixels x <= 512 channels): 1.2 GFLOP forward per frame, launch- and latency-bound —
// the point of this file is the ~40 torch / rocBLAS launches per pass it replaces.
#include <algorithm>
#include <cstdint>
#include <cstring>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"

namespace ganet {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// operand read with a per-channel prologue
//   mode 0: x                       mode 1: act(x * sc + sh), act = leaky(slope) (slope 0: ReLU, 1: identity)
//   mode 2: cf[0] g + cf[1] x + cf[2]   (dz of a layer with BatchNorm, from Gy = g and z = x)
struct USrc {
  const float* x;
  const float* g;
  const float* sc;      // mode 1: scale [C];  mode 2: coef [3][C]
  const float* sh;
  int C, mode;
  float slope;
};
// where a tile of output channels goes, and what the epilogue does with it
//   out (+)= val [+ bias];   dmode 1: val *= act'(z * sc + sh) (act = leaky(slope));
//   part: per-row-tile column sums [rows][2][C]: (val - shift, (val - shift)^2) (forward statistics, dmode 0) or
//         (val, val * (z * sc + sh)) of the STORED total (backward, dmode 1)
struct UDst {
  float* out;
  const float* z;
  const float* sc;
  const float* sh;
  const float* bias;
  const float* stat_shift;
  float* part;
  int C, dmode, accumulate;
  float slope;
};
struct UGemm {
  int B, Hc, Wc;          // coarse grid
  int N, Ctot;            // GEMM N (all destinations), K channels (all sources)
  int nsrc, ndst;
  USrc src[2];
  UDst dst[2];
  const float* Wp;        // [16][N][Ctot]
};

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

constexpr int UG_WG = 512;       // 8 waves split the taps (and, pattern T, the channel chunks): the maps are small, a wave's
                                 // serial chain of (tap, chunk) steps is what a launch takes

// PATTERN 0 = S, 1 = T (blockIdx.z = parity class)
template <int PATTERN>
__global__ void __launch_bounds__(UG_WG)
ugemm_kernel(UGemm p) {
  __shared__ float s_part[UG_WG / 64 - 1][16][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, r = lane & 31;
  const int Hc = p.Hc, Wc = p.Wc;                    // (powers of two)
  const int lw = 31 - __clz(Wc), lhw = lw + (31 - __clz(Hc));
  const int M = p.B * Hc * Wc;                       // output pixels (of this parity class)
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int cls = PATTERN ? (int)blockIdx.z : 0;
  const int py = cls >> 1, px = cls & 1;
  // this lane's output pixel (row r of the tile) on the COARSE index space
  const int m = m0 + r;
  const bool row_on = m < M;
  const int mm = row_on ? m : 0;
  const int b = mm >> lhw, yc = (mm >> lw) & (Hc - 1), xc = mm & (Wc - 1);
  const int Hs = PATTERN ? Hc : 2 * Hc, Ws = PATTERN ? Wc : 2 * Wc;     // source grid
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  // wave -> its taps: pattern S two of the 16 (t0, t0 + 8), pattern T one of the 4 and every second channel chunk
  constexpr int NW = UG_WG / 64;
  const int t0 = PATTERN ? (wave & 3) : wave;
  const int tstep = PATTERN ? 4 : NW, ntap = PATTERN ? 4 : 16;
  const int cg = PATTERN ? (wave >> 2) : 0, ncg = PATTERN ? NW / 4 : 1;
  int coff = 0, chunk = 0;
  for (int sidx = 0; sidx < p.nsrc; ++sidx) {
    const USrc& s = p.src[sidx];
    for (int c0 = 0; c0 < s.C; c0 += 32, ++chunk) {
      if ((chunk % ncg) != cg) continue;
      const int cl = c0 + 16 * h;                     // this lane's 16 channels of the chunk
      f32x4 k0[4], k1[4], k2[4];                      // per-channel coefficients (mode 1: sc, sh; mode 2: cA, cQ, cP)
      if (s.mode == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          k0[u] = *reinterpret_cast<const f32x4*>(s.sc + cl + 4 * u);
          k1[u] = *reinterpret_cast<const f32x4*>(s.sh + cl + 4 * u);
        }
      } else if (s.mode == 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          k0[u] = *reinterpret_cast<const f32x4*>(s.sc + cl + 4 * u);
          k1[u] = *reinterpret_cast<const f32x4*>(s.sc + s.C + cl + 4 * u);
          k2[u] = *reinterpret_cast<const f32x4*>(s.sc + 2 * s.C + cl + 4 * u);
        }
      }
      for (int t = t0; t < ntap; t += tstep) {
        // source pixel of this row for tap t, and the tap's index in the packed weights
        int sy, sx, wt;
        if (PATTERN == 0) {
          const int ky = t >> 2, kx = t & 3;
          sy = 2 * yc + ky - 1; sx = 2 * xc + kx - 1; wt = t;
        } else {
          // output (y, x) = (2 yc + py, 2 xc + px): ky = y + 1 - 2 iy in {1 - py, 3 - py}
          const int a = t >> 1, bb = t & 1;
          const int ky = (1 - py) + 2 * a, kx = (1 - px) + 2 * bb;
          sy = yc + py - a; sx = xc + px - bb; wt = ky * 4 + kx;
        }
        const bool on = row_on && sy >= 0 && sy < Hs && sx >= 0 && sx < Ws;
        const int64_t sp = on ? ((int64_t)(b * Hs + sy) * Ws + sx) * s.C + cl : cl;
        f32x4 a[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4*>(s.x + sp + 4 * u);
        const float* wp = p.Wp + ((int64_t)wt * p.N + n0 + r) * p.Ctot + coff + cl;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const f32x4*>(wp + 4 * u);
        if (s.mode == 2) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const f32x4 gq = *reinterpret_cast<const f32x4*>(s.g + sp + 4 * u);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[u][e] = fmaf(k0[u][e], gq[e], fmaf(k1[u][e], a[u][e], k2[u][e]));
          }
        } else if (s.mode == 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[u][e] = leaky(fmaf(a[u][e], k0[u][e], k1[u][e]), s.slope);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e)       // (zero padding / rows past the end: a select, not a product with 0)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(on ? a[u][e] : 0.f, w[u][e], acc, 0, 0, 0);
      }
    }
    coff += s.C;
  }
  // the waves' partial tiles
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) s_part[wave - 1][q][lane] = acc[q];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float v = acc[q];
#pragma unroll
    for (int w = 0; w < UG_WG / 64 - 1; ++w) v += s_part[w][q][lane];
    acc[q] = v;
  }
  // ---- epilogue: C/D layout column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  int doff = 0, di = 0;
  while (di + 1 < p.ndst && n0 >= doff + p.dst[di].C) { doff += p.dst[di].C; ++di; }
  const UDst& d = p.dst[di];
  const int n = n0 - doff + r;                         // channel inside the destination
  const float bias = d.bias ? d.bias[n] : 0.f;
  const float dsc = d.sc ? d.sc[n] : 1.f, dsh = d.sh ? d.sh[n] : 0.f;
  const float sshift = d.stat_shift ? d.stat_shift[n] : 0.f;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
    const int mr = m0 + row;
    if (mr >= M) continue;
    int64_t op;
    if (PATTERN == 0) {
      op = (int64_t)mr * d.C + n;
    } else {
      const int b2 = mr >> lhw, y2 = (mr >> lw) & (Hc - 1), x2 = mr & (Wc - 1);
      op = ((int64_t)(b2 * 2 * Hc + 2 * y2 + py) * (2 * Wc) + 2 * x2 + px) * d.C + n;
    }
    float val = acc[q] + bias;
    if (d.dmode == 1) {
      const float yv = fmaf(d.z[op], dsc, dsh);
      val *= yv > 0.f ? 1.f : d.slope;
      if (d.accumulate) val += d.out[op];
      d.out[op] = val;
      s1 += val;
      s2 = fmaf(val, yv, s2);
    } else {
      if (d.accumulate) val += d.out[op];
      d.out[op] = val;
      const float dv = val - sshift;
      s1 += dv;
      s2 = fmaf(dv, dv, s2);
    }
  }
  if (d.part) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (h == 0) {
      const int prow = cls * gridDim.x + blockIdx.x;
      d.part[((int64_t)prow * 2 + 0) * d.C + n] = s1;
      d.part[((int64_t)prow * 2 + 1) * d.C + n] = s2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient: D[t][i][j] = sum_m U[m, i] V[gather_S(m, t), j] over the coarse pixels m (U on the coarse grid, V
// gathered from the fine grid with the 16 taps of pattern S). conv: U = dz of the output, V = the activated input
// (dW[co][ci][t]); transposed conv: U = the activated input (two segments), V = dz of the output (dW[ci][co][t]).
// One wave per (tap, 32 x 32 tile of (i, j), 256-pixel chunk): partial tiles [chunk][16][I][J].
struct UWgrad {
  int B, Hc, Wc, I, J, nu, chunk_rows;
  USrc U[2];
  USrc V;
  float* part;
};

__device__ __forceinline__ float usrc_read(const USrc& s, int64_t idx, float k0, float k1, float k2) {
  const float x = s.x[idx];
  if (s.mode == 1) return leaky(fmaf(x, k0, k1), s.slope);
  if (s.mode == 2) return fmaf(k0, s.g[idx], fmaf(k1, x, k2));
  return x;
}

constexpr int UW_WG = 256;       // four waves share a (tap, tile, chunk): a quarter of the chunk's rows each

__global__ void __launch_bounds__(UW_WG)
uwgrad_kernel(UWgrad p) {
  __shared__ float s_part[3][16][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, r = lane & 31;
  const int t = blockIdx.x;
  const int jt = p.J / 32;
  const int i0 = ((int)blockIdx.y / jt) * 32, j0 = ((int)blockIdx.y % jt) * 32;
  const int Hc = p.Hc, Wc = p.Wc, M = p.B * Hc * Wc;
  const int lw = 31 - __clz(Wc), lhw = lw + (31 - __clz(Hc));
  const int clo = blockIdx.z * p.chunk_rows, chi = min(M, clo + p.chunk_rows);
  const int quarter = p.chunk_rows / 4;
  const int mlo = clo + wave * quarter, mhi = min(chi, mlo + quarter);
  // U segment of this i tile
  int uo = 0, ui = 0;
  while (ui + 1 < p.nu && i0 >= uo + p.U[ui].C) { uo += p.U[ui].C; ++ui; }
  const USrc& U = p.U[ui];
  const USrc& V = p.V;
  const int ic = i0 - uo + r, jc = j0 + r;
  float u0 = 0.f, u1 = 0.f, u2 = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f;
  if (U.mode == 1) { u0 = U.sc[ic]; u1 = U.sh[ic]; }
  else if (U.mode == 2) { u0 = U.sc[ic]; u1 = U.sc[U.C + ic]; u2 = U.sc[2 * U.C + ic]; }
  if (V.mode == 1) { v0 = V.sc[jc]; v1 = V.sh[jc]; }
  else if (V.mode == 2) { v0 = V.sc[jc]; v1 = V.sc[V.C + jc]; v2 = V.sc[2 * V.C + jc]; }
  const int ky = t >> 2, kx = t & 3;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int mb = mlo; mb < mhi; mb += 32) {
    float uu[16], vv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int m = mb + 2 * s + h;
      const bool on = m < mhi;
      const int mm = on ? m : mlo;
      const int b = mm >> lhw, yc = (mm >> lw) & (Hc - 1), xc = mm & (Wc - 1);
      const int sy = 2 * yc + ky - 1, sx = 2 * xc + kx - 1;
      const bool von = on && sy >= 0 && sy < 2 * Hc && sx >= 0 && sx < 2 * Wc;
      const int64_t vp = von ? ((int64_t)(b * 2 * Hc + sy) * (2 * Wc) + sx) * V.C + jc : jc;
      const float ur = usrc_read(U, (int64_t)mm * U.C + ic, u0, u1, u2);
      const float vr = usrc_read(V, vp, v0, v1, v2);
      uu[s] = on ? ur : 0.f;
      vv[s] = von ? vr : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(uu[s], vv[s], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) s_part[wave - 1][q][lane] = acc[q];
  }
  __syncthreads();
  if (wave > 0) return;
  float* out = p.part + ((int64_t)blockIdx.z * 16 + t) * p.I * p.J;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
    out[(int64_t)(i0 + row) * p.J + j0 + r] = (acc[q] + s_part[0][q][lane]) + (s_part[1][q][lane] + s_part[2][q][lane]);
  }
}

// dW[i * sI + j * sJ + t] = sum over chunks of part[chunk][t][i][j]
__global__ void __launch_bounds__(256)
uwgrad_reduce_kernel(int nchunk, int I, int J, int64_t sI, int64_t sJ, const float* __restrict__ part,
                     float* __restrict__ dW) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)16 * I * J;
  if (e >= total) return;
  const int t = (int)(e / ((int64_t)I * J));
  const int64_t ij = e - (int64_t)t * I * J;
  const int i = (int)(ij / J), j = (int)(ij - (int64_t)i * J);
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += part[(int64_t)c * total + e];
  dW[i * sI + j * sJ + t] = s;
}

// Wp[t][n][c] = W[n * sn + c * sc + t] for up to 9 layers in one launch (blockIdx.y = layer)
struct UPackJobs { const float* W[9]; float* Wp[9]; int N[9], C[9]; long long sn[9], sc[9]; };
__global__ void __launch_bounds__(256)
upack_kernel(UPackJobs jobs) {
  const int j = blockIdx.y;
  const int N = jobs.N[j], C = jobs.C[j];
  const int64_t total = (int64_t)16 * N * C;
  const float* __restrict__ W = jobs.W[j];
  float* __restrict__ Wp = jobs.Wp[j];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e / ((int64_t)N * C));
    const int64_t nc = e - (int64_t)t * N * C;
    const int n = (int)(nc / C), c = (int)(nc - (int64_t)n * C);
    Wp[e] = W[n * jobs.sn[j] + c * jobs.sc[j] + t];
  }
}

// BatchNorm statistics from the forward epilogue's partials (sums about the running mean) -> scale = rstd,
// shift = -mean rstd (affine = False), running statistics as F.batch_norm(training = True)
__global__ void __launch_bounds__(64)
ubn_fwd_kernel(int nparts, int C, float count, const float* __restrict__ part, const float* __restrict__ stat_shift,
               float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
               long long* __restrict__ nbt, float* __restrict__ sc, float* __restrict__ sh) {
  const int n = blockIdx.x;
  const float shv = stat_shift ? stat_shift[n] : 0.f;          // (may alias running_mean: read first)
  const float rm = running_mean ? running_mean[n] : 0.f, rv = running_var ? running_var[n] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  for (int pr = threadIdx.x; pr < nparts; pr += 64) {
    s1 += (double)part[((int64_t)pr * 2 + 0) * C + n];
    s2 += (double)part[((int64_t)pr * 2 + 1) * C + n];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if (threadIdx.x == 0) {
    const double dm = s1 / (double)count;
    const double mean = (double)shv + dm;
    double var = s2 / (double)count - dm * dm;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    sc[n] = rstd;
    sh[n] = -(float)mean * rstd;
    if (running_mean) {
      const double unbiased = var * ((double)count / (double)(count > 1.f ? count - 1.f : 1.f));
      running_mean[n] = (1.f - momentum) * rm + momentum * (float)mean;
      running_var[n] = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
    if (n == 0 && nbt) *nbt += 1;
  }
}

// evaluation mode: scale / shift from the running statistics
__global__ void ubn_eval_kernel(int C, const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                float* __restrict__ sc, float* __restrict__ sh) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= C) return;
  const float rstd = 1.0f / sqrtf(rv[n] + eps);
  sc[n] = rstd;
  sh[n] = -rm[n] * rstd;
}

// backward: sums of Gy and Gy y^ -> dz = cA Gy + cQ z + cP with y^ = z sc + sh:
//   dz = rstd (Gy - mean(Gy) - y^ mean(Gy y^))
__global__ void __launch_bounds__(64)
ubn_bwd_kernel(int nparts, int C, float count, const float* __restrict__ part, const float* __restrict__ sc,
               const float* __restrict__ sh, float* __restrict__ coef) {
  const int n = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int pr = threadIdx.x; pr < nparts; pr += 64) {
    s1 += (double)part[((int64_t)pr * 2 + 0) * C + n];
    s2 += (double)part[((int64_t)pr * 2 + 1) * C + n];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if (threadIdx.x == 0) {
    const float rstd = sc[n];
    const float mg = (float)(s1 / (double)count), mgy = (float)(s2 / (double)count);
    coef[n] = rstd;
    coef[C + n] = -rstd * mgy * rstd;
    coef[2 * C + n] = -rstd * (mg + mgy * sh[n]);
  }
}

// ---- the first layer (3 input channels: no GEMM shape). x is NCHW as the data loader hands it over.
// z1[b, oy, ox, co] = sum_{ci, ky, kx} W[co][ci][ky][kx] x[b, ci, 2 oy + ky - 1, 2 ox + kx - 1]
constexpr int UC1_MAXW = 8 * 16 * 64;      // cin <= 8, nf <= 64 weights in LDS (else global)
__global__ void __launch_bounds__(256)
uconv1_fwd_kernel(int B, int Cin, int S, int Cout, const float* __restrict__ x, const float* __restrict__ W,
                  float* __restrict__ z) {
  __shared__ float s_w[UC1_MAXW];
  const int nw = Cout * Cin * 16;
  const bool lds_w = nw <= UC1_MAXW;
  if (lds_w) for (int i = threadIdx.x; i < nw; i += 256) s_w[i] = W[i];
  __syncthreads();
  const int Ho = S / 2;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (pixel, co), co fastest
  if (e >= (int64_t)B * Ho * Ho * Cout) return;
  const int co = (int)(e % Cout);
  const int64_t pix = e / Cout;
  const int b = (int)(pix / (Ho * Ho)), rem = (int)(pix - (int64_t)b * Ho * Ho);
  const int oy = rem / Ho, ox = rem - oy * Ho;
  float acc = 0.f;
  for (int ci = 0; ci < Cin; ++ci)
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int sy = 2 * oy + ky - 1;
      if (sy < 0 || sy >= S) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int sx = 2 * ox + kx - 1;
        if (sx < 0 || sx >= S) continue;
        const int wi = ((co * Cin + ci) * 4 + ky) * 4 + kx;
        acc = fmaf(lds_w ? s_w[wi] : W[wi], x[(((int64_t)b * Cin + ci) * S + sy) * S + sx], acc);
      }
    }
  z[e] = acc;
}

// dW1[co][ci][ky][kx] partial over a chunk of output pixels: part[chunk][co * Cin * 16 + k]. Workgroup = (k, chunk):
// 8 row lanes x 32 output channels (Cout <= 32 per blockIdx.z slice), rows strided over the row lanes.
__global__ void __launch_bounds__(256)
uconv1_wgrad_kernel(int B, int Cin, int S, int Cout, int chunk_rows, const float* __restrict__ x,
                    const float* __restrict__ dz, float* __restrict__ part) {
  __shared__ float s_acc[8][32];
  const int Ho = S / 2;
  const int K = Cin * 16;
  const int k = blockIdx.x;
  const int ci = k / 16, ky = (k >> 2) & 3, kx = k & 3;
  const int rl = threadIdx.x >> 5, co = blockIdx.z * 32 + (threadIdx.x & 31);
  const int M = B * Ho * Ho;
  const int mlo = blockIdx.y * chunk_rows, mhi = min(M, mlo + chunk_rows);
  float acc = 0.f;
  if (co < Cout)
    for (int m = mlo + rl; m < mhi; m += 8) {
      const int b = m / (Ho * Ho), rem = m - b * Ho * Ho;
      const int oy = rem / Ho, ox = rem - oy * Ho;
      const int sy = 2 * oy + ky - 1, sx = 2 * ox + kx - 1;
      if (sy < 0 || sy >= S || sx < 0 || sx >= S) continue;
      acc = fmaf(dz[(int64_t)m * Cout + co], x[(((int64_t)b * Cin + ci) * S + sy) * S + sx], acc);
    }
  s_acc[rl][threadIdx.x & 31] = acc;
  __syncthreads();
  if (rl == 0 && co < Cout) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += s_acc[w][threadIdx.x & 31];
    part[(int64_t)blockIdx.y * K * Cout + (int64_t)co * K + k] = s;
  }
}

__global__ void __launch_bounds__(256)
usum_chunks_kernel(int nchunk, int64_t total, const float* __restrict__ part, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  float s = 0.f;
  int c = 0;
  for (; c + 8 <= nchunk; c += 8) {          // eight independent loads in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(c + u) * total + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; c < nchunk; ++c) s += part[(int64_t)c * total + e];
  out[e] = s;
}

// column sums of g [M, C] (the output bias gradient): partial per chunk of rows (4 row lanes x 64 channels per
// workgroup), then usum_chunks
__global__ void __launch_bounds__(256)
ucolsum_kernel(int M, int C, int chunk_rows, const float* __restrict__ g, float* __restrict__ part) {
  __shared__ float s_acc[4][64];
  const int rl = threadIdx.x >> 6, c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int mlo = blockIdx.y * chunk_rows, mhi = min(M, mlo + chunk_rows);
  float s = 0.f;
  if (c < C) for (int m = mlo + rl; m < mhi; m += 4) s += g[(int64_t)m * C + c];
  s_acc[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C)
    part[(int64_t)blockIdx.y * C + c] = (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x]) + (s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
}

// ones / zeros for the tensors without BatchNorm: sc[0] = sc[1] = 1, sh[0] = sh[1] = 0
__global__ void ufill_kernel(int n0, int n1, float* sc0, float* sh0, float* sc1, float* sh1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n0) { sc0[i] = 1.f; sh0[i] = 0.f; }
  if (i < n1) { sc1[i] = 1.f; sh1[i] = 0.f; }
}

// ------------------------------------------------------------------------------------------------- host side
inline size_t al(size_t v) { return (v + 63) / 64 * 64; }      // floats, 256-byte granules

struct Net {
  int B, S, cin, nf, cout;
  int C[6];        // C[1..5]: channels of z1..z5
  int Hd[6];       // Hd[k]: edge of z_k (k = 1..5): S / 2^k
  int CU[6];       // CU[k]: channels of zT_k (k = 1..5)
  int HU[6];       // HU[k]: edge of zT_k: Hd[5] * 2^k
};
Net make_net(const GanetUnetParams* p, int B) {
  Net n{};
  n.B = B; n.S = p->S; n.cin = p->cin; n.nf = p->nf; n.cout = p->cout;
  const int c[6] = {0, p->nf, 2 * p->nf, 4 * p->nf, 8 * p->nf, 8 * p->nf};
  for (int k = 1; k <= 5; ++k) { n.C[k] = c[k]; n.Hd[k] = p->S >> k; }
  const int cu[6] = {0, 8 * p->nf, 4 * p->nf, 2 * p->nf, p->nf, p->cout};
  for (int k = 1; k <= 5; ++k) { n.CU[k] = cu[k]; n.HU[k] = n.Hd[5] << k; }
  return n;
}
bool unet_ok(const GanetUnetParams* p, int B) {
  if (!p || B <= 0 || p->cin <= 0 || p->cin > 8 || p->nf <= 0 || (p->nf % 32) || p->cout <= 0 || (p->cout % 32) ||
      p->S < 32 || (p->S & (p->S - 1))) return false;      // S a power of two: every map edge is one (shifts)
  for (int k = 0; k < 5; ++k) if (!p->Wd[k] || !p->Wu[k]) return false;
  return p->bias5 != nullptr;
}
// saved tensor offsets (floats): z1..z5, zT1..zT5 (zT5 = the output, kept by the caller), then BN scale/shift
struct Saved {
  float* z[6]; float* zT[6];
  float* sc_d[6]; float* sh_d[6];     // BN of z2..z4 (z1, z5: ones / zeros)
  float* sc_u[6]; float* sh_u[6];     // BN of zT1..zT4
  size_t total;
};
Saved carve_saved(const Net& n, float* base) {
  Saved s{};
  size_t o = 0;
  for (int k = 1; k <= 5; ++k) { s.z[k] = base + o; o += al((size_t)n.B * n.Hd[k] * n.Hd[k] * n.C[k]); }
  for (int k = 1; k <= 4; ++k) { s.zT[k] = base + o; o += al((size_t)n.B * n.HU[k] * n.HU[k] * n.CU[k]); }
  for (int k = 1; k <= 5; ++k) { s.sc_d[k] = base + o; o += al(n.C[k]); s.sh_d[k] = base + o; o += al(n.C[k]); }
  for (int k = 1; k <= 4; ++k) { s.sc_u[k] = base + o; o += al(n.CU[k]); s.sh_u[k] = base + o; o += al(n.CU[k]); }
  s.total = o;
  return s;
}
size_t packed_floats(const Net& n) {      // all packed weights of one direction (forward or backward)
  size_t o = 0;
  for (int k = 2; k <= 5; ++k) o += al((size_t)16 * n.C[k] * n.C[k - 1]);
  const int cinu[6] = {0, n.C[5], n.CU[1] + n.C[4], n.CU[2] + n.C[3], n.CU[3] + n.C[2], n.CU[4] + n.C[1]};
  for (int k = 1; k <= 5; ++k) o += al((size_t)16 * n.CU[k] * cinu[k]);
  return o;
}
size_t max_part_floats(const Net& n) {    // BN partials of the widest case: rows x 2 x C
  size_t mx = 0;
  for (int k = 1; k <= 5; ++k) {
    mx = std::max(mx, (size_t)((n.B * n.Hd[k] * n.Hd[k] + 31) / 32) * 2 * n.C[k]);
    mx = std::max(mx, (size_t)4 * ((n.B * (n.HU[k] / 2) * (n.HU[k] / 2) + 31) / 32) * 2 * std::max(n.CU[k], n.C[1]));
  }
  return al(mx * 2);
}

int launch_ugemm(int pattern, const UGemm& g, hipStream_t st) {
  const int M = g.B * g.Hc * g.Wc;
  const dim3 grid((M + 31) / 32, g.N / 32, pattern ? 4 : 1);
  if (pattern) hipLaunchKernelGGL(ugemm_kernel<1>, grid, dim3(UG_WG), 0, st, g);
  else hipLaunchKernelGGL(ugemm_kernel<0>, grid, dim3(UG_WG), 0, st, g);
  return check_hip(hipGetLastError(), "ugemm_kernel");
}
struct PackList {
  UPackJobs jobs{};
  int n = 0;
  void add(int N, int C, int64_t sn, int64_t sc, const float* W, float* Wp) {
    jobs.W[n] = W; jobs.Wp[n] = Wp; jobs.N[n] = N; jobs.C[n] = C; jobs.sn[n] = sn; jobs.sc[n] = sc; ++n;
  }
  int launch(hipStream_t st) {
    if (!n) return 0;
    hipLaunchKernelGGL(upack_kernel, dim3(256, n), dim3(256), 0, st, jobs);
    return check_hip(hipGetLastError(), "upack_kernel");
  }
};
USrc src_raw(const float* x, int C) { USrc s{}; s.x = x; s.C = C; s.mode = 0; s.slope = 1.f; return s; }
USrc src_act(const float* z, int C, const float* sc, const float* sh, float slope) {
  USrc s{}; s.x = z; s.sc = sc; s.sh = sh; s.C = C; s.mode = 1; s.slope = slope; return s;
}
USrc src_dz(const float* g, const float* z, const float* coef, int C) {
  USrc s{}; s.x = z; s.g = g; s.sc = coef; s.C = C; s.mode = coef ? 2 : 0; s.slope = 1.f;
  if (!coef) s.x = g;                   // no BatchNorm: dz = Gy
  return s;
}

}  // namespace

}  // namespace ganet

using namespace ganet;

#define GA_TRY(call)            \
  do {                          \
    const int rc_ = (call);     \
    if (rc_) return rc_;        \
  } while (0)

extern "C" {

size_t ganet_unet_saved_floats(const GanetUnetParams* p, int32_t B) {
  if (!unet_ok(p, B)) return 0;
  return carve_saved(make_net(p, B), nullptr).total;
}

size_t ganet_unet_fwd_workspace(const GanetUnetParams* p, int32_t B) {
  if (!unet_ok(p, B)) return 0;
  const Net n = make_net(p, B);
  return (packed_floats(n) + max_part_floats(n)) * sizeof(float);
}

// x: [B, cin, S, S] NCHW; out: [B, S, S, cout] channels-last. training != 0: batch statistics (and the running
// statistics updated); else the running statistics normalise.
int ganet_unet_fwd(const GanetUnetParams* p, int32_t B, const float* x, int32_t training, float* saved, float* out,
                   void* workspace, size_t workspace_bytes, void* stream_) {
  if (!unet_ok(p, B) || !x || !saved || !out || !workspace) {
    set_error("ganet_unet_fwd: invalid arguments (nf and cout multiples of 32, S a multiple of 32, cin <= 8)");
    return 1;
  }
  if (workspace_bytes < ganet_unet_fwd_workspace(p, B)) { set_error("ganet_unet_fwd: workspace too small"); return 2; }
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const Net n = make_net(p, B);
  const Saved sv = carve_saved(n, saved);
  float* wsf = static_cast<float*>(workspace);
  float* part = wsf + packed_floats(n);
  // ones / zeros for the tensors without BatchNorm (z1, z5)
  hipLaunchKernelGGL(ufill_kernel, dim3((std::max(n.C[1], n.C[5]) + 255) / 256), dim3(256), 0, st, n.C[1], n.C[5], sv.sc_d[1],
                     sv.sh_d[1], sv.sc_d[5], sv.sh_d[5]);
  // every layer's weights in the [tap][n][c] order its GEMM reads, one launch
  float* wpk_d[6]; float* wpk_u[6];
  {
    PackList pl;
    float* q = wsf;
    for (int k = 2; k <= 5; ++k) {
      wpk_d[k] = q; q += al((size_t)16 * n.C[k] * n.C[k - 1]);
      pl.add(n.C[k], n.C[k - 1], (int64_t)n.C[k - 1] * 16, 16, p->Wd[k - 1], wpk_d[k]);
    }
    const int cinu[6] = {0, n.C[5], n.CU[1] + n.C[4], n.CU[2] + n.C[3], n.CU[3] + n.C[2], n.CU[4] + n.C[1]};
    for (int k = 1; k <= 5; ++k) {
      wpk_u[k] = q; q += al((size_t)16 * n.CU[k] * cinu[k]);
      pl.add(n.CU[k], cinu[k], 16, (int64_t)n.CU[k] * 16, p->Wu[k - 1], wpk_u[k]);      // WT[ci][co][16] -> [t][co][ci]
    }
    GA_TRY(pl.launch(st));
  }
  // BatchNorm of a layer's raw output from the partials its forward launch left
  auto bn = [&](int idx, int C, int nparts, float count, float* sc, float* sh) -> int {
    if (training) {
      hipLaunchKernelGGL(ubn_fwd_kernel, dim3(C), dim3(64), 0, st, nparts, C, count, part, p->running_mean[idx], p->eps,
                         p->momentum, p->running_mean[idx], p->running_var[idx],
                         reinterpret_cast<long long*>(p->num_batches_tracked[idx]), sc, sh);
    } else {
      if (!p->running_mean[idx] || !p->running_var[idx]) { set_error("ganet_unet_fwd: evaluation needs running statistics"); return 1; }
      hipLaunchKernelGGL(ubn_eval_kernel, dim3((C + 63) / 64), dim3(64), 0, st, C, p->running_mean[idx], p->running_var[idx],
                         p->eps, sc, sh);
    }
    return check_hip(hipGetLastError(), "ubn kernels");
  };
  // ---- conv1
  {
    const int64_t tot = (int64_t)B * n.Hd[1] * n.Hd[1] * n.C[1];
    hipLaunchKernelGGL(uconv1_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, B, n.cin, n.S, n.C[1], x,
                       p->Wd[0], sv.z[1]);
    GA_TRY(check_hip(hipGetLastError(), "uconv1_fwd_kernel"));
  }
  // ---- conv2..5: input a_{k-1} = leaky(bn(z_{k-1}))
  for (int k = 2; k <= 5; ++k) {
    const int ci = n.C[k - 1], co = n.C[k];
    UGemm g{};
    g.B = B; g.Hc = g.Wc = n.Hd[k]; g.N = co; g.Ctot = ci; g.nsrc = 1; g.ndst = 1; g.Wp = wpk_d[k];
    g.src[0] = src_act(sv.z[k - 1], ci, sv.sc_d[k - 1], sv.sh_d[k - 1], 0.2f);
    UDst d{}; d.out = sv.z[k]; d.C = co;
    const bool has_bn = k <= 4;
    const int bi = k - 2;                  // BatchNorm index 0..2 = conv2..4
    if (has_bn && training) { d.part = part; d.stat_shift = p->running_mean[bi]; }
    g.dst[0] = d;
    GA_TRY(launch_ugemm(0, g, st));
    if (has_bn) GA_TRY(bn(bi, co, (B * n.Hd[k] * n.Hd[k] + 31) / 32, (float)(B * n.Hd[k] * n.Hd[k]), sv.sc_d[k], sv.sh_d[k]));
  }
  // ---- upconv1..5: input relu(cat[bn(zT_{k-1}), a_{5-k+1}]) (upconv1: relu(z5))
  for (int k = 1; k <= 5; ++k) {
    const int co = n.CU[k];
    UGemm g{};
    g.B = B; g.Hc = g.Wc = n.HU[k] / 2; g.N = co; g.ndst = 1;
    if (k == 1) {
      g.nsrc = 1; g.src[0] = src_act(sv.z[5], n.C[5], sv.sc_d[5], sv.sh_d[5], 0.f);
    } else {
      const int skip = 6 - k;              // a_4, a_3, a_2, a_1
      g.nsrc = 2;
      g.src[0] = src_act(sv.zT[k - 1], n.CU[k - 1], sv.sc_u[k - 1], sv.sh_u[k - 1], 0.f);
      g.src[1] = src_act(sv.z[skip], n.C[skip], sv.sc_d[skip], sv.sh_d[skip], 0.f);     // relu(leaky(v)) = relu(v)
    }
    const int ci = g.src[0].C + (g.nsrc > 1 ? g.src[1].C : 0);
    g.Ctot = ci;
    g.Wp = wpk_u[k];
    UDst d{}; d.C = co;
    const bool has_bn = k <= 4;
    const int bi = 2 + k;                  // BatchNorm index 3..6 = upconv1..4
    d.out = has_bn ? sv.zT[k] : out;
    if (!has_bn) d.bias = p->bias5;
    if (has_bn && training) { d.part = part; d.stat_shift = p->running_mean[bi]; }
    g.dst[0] = d;
    GA_TRY(launch_ugemm(1, g, st));
    if (has_bn) {
      const int rows = (B * g.Hc * g.Wc + 31) / 32;
      GA_TRY(bn(bi, co, 4 * rows, (float)(B * n.HU[k] * n.HU[k]), sv.sc_u[k], sv.sh_u[k]));
    }
  }
  return 0;
}

size_t ganet_unet_bwd_workspace(const GanetUnetParams* p, int32_t B) {
  if (!unet_ok(p, B)) return 0;
  const Net n = make_net(p, B);
  size_t o = packed_floats(n) + max_part_floats(n);
  // Gy of z1..z5 and zT1..zT4, coefficients, weight-gradient partial tiles
  for (int k = 1; k <= 5; ++k) o += al((size_t)B * n.Hd[k] * n.Hd[k] * n.C[k]) + al(3 * n.C[k]);
  for (int k = 1; k <= 4; ++k) o += al((size_t)B * n.HU[k] * n.HU[k] * n.CU[k]) + al(3 * n.CU[k]);
  size_t wg = 0;
  for (int k = 2; k <= 5; ++k) {
    const int M = B * n.Hd[k] * n.Hd[k];
    wg = std::max(wg, (size_t)((M + 255) / 256) * 16 * n.C[k] * n.C[k - 1]);
  }
  const int cinu[6] = {0, n.C[5], n.CU[1] + n.C[4], n.CU[2] + n.C[3], n.CU[3] + n.C[2], n.CU[4] + n.C[1]};
  for (int k = 1; k <= 5; ++k) {
    const int M = B * (n.HU[k] / 2) * (n.HU[k] / 2);
    wg = std::max(wg, (size_t)((M + 255) / 256) * 16 * cinu[k] * n.CU[k]);
  }
  wg = std::max(wg, (size_t)((B * n.Hd[1] * n.Hd[1] + 255) / 256) * n.cin * 16 * n.C[1]);
  o += al(wg);
  return o * sizeof(float);
}

// d_out: [B, S, S, cout] channels-last. Gradients in the parameters' own layouts: dWd[k] like Wd[k]
// ([co][ci][4][4]), dWu[k] like Wu[k] ([ci][co][4][4]), dbias5 [cout]. (The input x has no gradient.)
int ganet_unet_bwd(const GanetUnetParams* p, int32_t B, const float* x, const float* saved_, const float* d_out,
                   const GanetUnetGrads* gr, void* workspace, size_t workspace_bytes, void* stream_, void* side_stream_) {
  if (!unet_ok(p, B) || !x || !saved_ || !d_out || !gr || !workspace || !gr->dbias5) {
    set_error("ganet_unet_bwd: invalid arguments");
    return 1;
  }
  for (int k = 0; k < 5; ++k) if (!gr->dWd[k] || !gr->dWu[k]) { set_error("ganet_unet_bwd: missing gradient buffer"); return 1; }
  if (workspace_bytes < ganet_unet_bwd_workspace(p, B)) { set_error("ganet_unet_bwd: workspace too small"); return 2; }
  hipStream_t st = static_cast<hipStream_t>(stream_);
  hipStream_t side = static_cast<hipStream_t>(side_stream_);
  // The input-gradient chain (dgrad_k -> BatchNorm coefficients -> dgrad_k-1 ...) is the critical path; a layer's weight
  // gradient hangs off it (it needs dz_k, nothing needs it): side stream, ordered by events as in ganet_decoder_bwd.
  // Both chains are launches of a few workgroups x a few microseconds, so they overlap almost entirely. The side stream's
  // launches share `wgpart` (serial on that stream) and only READ what the main stream produced before the fork.
  static thread_local hipEvent_t ev_main_dev[64] = {}, ev_side_dev[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  hipEvent_t& ev_main = ev_main_dev[dev_ & 63];
  hipEvent_t& ev_side = ev_side_dev[dev_ & 63];
  if (side && !ev_main) {
    GA_TRY(check_hip(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming), "hipEventCreate"));
    GA_TRY(check_hip(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming), "hipEventCreate"));
  }
  auto fork = [&](hipStream_t* out) -> int {      // the stream for work that depends on everything enqueued on `st` so far
    *out = st;
    if (!side) return 0;
    GA_TRY(check_hip(hipEventRecord(ev_main, st), "hipEventRecord"));
    GA_TRY(check_hip(hipStreamWaitEvent(side, ev_main, 0), "hipStreamWaitEvent"));
    *out = side;
    return 0;
  };
  const Net n = make_net(p, B);
  const Saved sv = carve_saved(n, const_cast<float*>(saved_));
  float* wsf = static_cast<float*>(workspace);
  float* wp = wsf;
  float* part = wsf + packed_floats(n);
  float* o = part + max_part_floats(n);
  float* Gd[6]; float* cf_d[6]; float* Gu[6]; float* cf_u[6];
  for (int k = 1; k <= 5; ++k) { Gd[k] = o; o += al((size_t)B * n.Hd[k] * n.Hd[k] * n.C[k]); cf_d[k] = o; o += al(3 * n.C[k]); }
  for (int k = 1; k <= 4; ++k) { Gu[k] = o; o += al((size_t)B * n.HU[k] * n.HU[k] * n.CU[k]); cf_u[k] = o; o += al(3 * n.CU[k]); }
  float* wgpart = o;
  // every layer's weights in the [tap][n][c] order its input-gradient GEMM reads, one launch
  float* wpk_d[6]; float* wpk_u[6];
  {
    PackList pl;
    float* q = wsf;
    const int cinu[6] = {0, n.C[5], n.CU[1] + n.C[4], n.CU[2] + n.C[3], n.CU[3] + n.C[2], n.CU[4] + n.C[1]};
    for (int k = 1; k <= 5; ++k) {
      wpk_u[k] = q; q += al((size_t)16 * n.CU[k] * cinu[k]);
      pl.add(cinu[k], n.CU[k], (int64_t)n.CU[k] * 16, 16, p->Wu[k - 1], wpk_u[k]);      // -> [t][n = ci][c = co]
    }
    for (int k = 2; k <= 5; ++k) {
      wpk_d[k] = q; q += al((size_t)16 * n.C[k] * n.C[k - 1]);
      pl.add(n.C[k - 1], n.C[k], 16, (int64_t)n.C[k - 1] * 16, p->Wd[k - 1], wpk_d[k]);   // W[co][ci][16] -> [t][n = ci][c = co]
    }
    GA_TRY(pl.launch(st));
  }
  (void)wp;
  auto wgrad = [&](int Hc, int I, int J, int nu, const USrc* U, const USrc& V, int64_t sI, int64_t sJ, float* dW) -> int {
    UWgrad w{};
    w.B = B; w.Hc = w.Wc = Hc; w.I = I; w.J = J; w.nu = nu; w.chunk_rows = 256;
    for (int i = 0; i < nu; ++i) w.U[i] = U[i];
    w.V = V; w.part = wgpart;
    const int M = B * Hc * Hc, nchunk = (M + 255) / 256;
    hipStream_t ws;
    GA_TRY(fork(&ws));
    hipLaunchKernelGGL(uwgrad_kernel, dim3(16, (I / 32) * (J / 32), nchunk), dim3(UW_WG), 0, ws, w);
    GA_TRY(check_hip(hipGetLastError(), "uwgrad_kernel"));
    const int64_t tot = (int64_t)16 * I * J;
    hipLaunchKernelGGL(uwgrad_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ws, nchunk, I, J, sI, sJ,
                       wgpart, dW);
    return check_hip(hipGetLastError(), "uwgrad_reduce_kernel");
  };
  auto bn_coef = [&](int C, int nparts, float count, const float* sc, const float* sh, float* coef) -> int {
    hipLaunchKernelGGL(ubn_bwd_kernel, dim3(C), dim3(64), 0, st, nparts, C, count, part, sc, sh, coef);
    return check_hip(hipGetLastError(), "ubn_bwd_kernel");
  };
  // ---- up path, k = 5 .. 1. dz of zT_k: k = 5: d_out itself; else assembled from (Gu[k], zT[k], cf_u[k])
  for (int k = 5; k >= 1; --k) {
    const int co = n.CU[k], Hc = n.HU[k] / 2;
    const USrc dz = k == 5 ? src_dz(d_out, nullptr, nullptr, co) : src_dz(Gu[k], sv.zT[k], cf_u[k], co);
    // the activated input (two segments)
    USrc in[2];
    int nin;
    if (k == 1) { nin = 1; in[0] = src_act(sv.z[5], n.C[5], sv.sc_d[5], sv.sh_d[5], 0.f); }
    else {
      const int skip = 6 - k;
      nin = 2;
      in[0] = src_act(sv.zT[k - 1], n.CU[k - 1], sv.sc_u[k - 1], sv.sh_u[k - 1], 0.f);
      in[1] = src_act(sv.z[skip], n.C[skip], sv.sc_d[skip], sv.sh_d[skip], 0.f);
    }
    const int ci = in[0].C + (nin > 1 ? in[1].C : 0);
    if (k == 5) {      // bias gradient: column sums of d_out
      const int M = B * n.S * n.S, nchunk = (M + 255) / 256;
      hipStream_t ws;
      GA_TRY(fork(&ws));
      hipLaunchKernelGGL(ucolsum_kernel, dim3((co + 63) / 64, nchunk), dim3(256), 0, ws, M, co, 256, d_out, wgpart);
      hipLaunchKernelGGL(usum_chunks_kernel, dim3((co + 255) / 256), dim3(256), 0, ws, nchunk, (int64_t)co, wgpart, gr->dbias5);
      GA_TRY(check_hip(hipGetLastError(), "ucolsum kernels"));
    }
    // weight gradient dWT[ci][co][t] = sum_m in[m, ci] dz[gather_S(m, t), co]
    GA_TRY(wgrad(Hc, ci, co, nin, in, dz, (int64_t)co * 16, 16, gr->dWu[k - 1]));
    // input gradient (pattern S over dz): din[b, iy, ix, ci] = sum dz[..., co] WT[ci][co][t]; times relu' into the Gy of
    // its two destinations: zT_{k-1} (first, only contribution) and the skip z_{6-k} (first contribution)
    UGemm g{};
    g.B = B; g.Hc = g.Wc = Hc; g.N = ci; g.Ctot = co; g.nsrc = 1; g.src[0] = dz; g.Wp = wpk_u[k];
    if (k == 1) {
      g.ndst = 1;
      UDst d{}; d.out = Gd[5]; d.z = sv.z[5]; d.sc = sv.sc_d[5]; d.sh = sv.sh_d[5]; d.C = n.C[5]; d.dmode = 1; d.slope = 0.f;
      g.dst[0] = d;                       // z5 has no BatchNorm: no sums
    } else {
      const int skip = 6 - k;
      g.ndst = 2;
      UDst d0{}; d0.out = Gu[k - 1]; d0.z = sv.zT[k - 1]; d0.sc = sv.sc_u[k - 1]; d0.sh = sv.sh_u[k - 1]; d0.C = n.CU[k - 1];
      d0.dmode = 1; d0.slope = 0.f; d0.part = part;
      UDst d1{}; d1.out = Gd[skip]; d1.z = sv.z[skip]; d1.sc = sv.sc_d[skip]; d1.sh = sv.sh_d[skip]; d1.C = n.C[skip];
      d1.dmode = 1; d1.slope = 0.f;       // the down path adds its contribution (and takes the sums) later
      g.dst[0] = d0; g.dst[1] = d1;
    }
    GA_TRY(launch_ugemm(0, g, st));
    if (k >= 2) GA_TRY(bn_coef(n.CU[k - 1], (B * Hc * Hc + 31) / 32, (float)(B * Hc * Hc), sv.sc_u[k - 1], sv.sh_u[k - 1],
                               cf_u[k - 1]));
  }
  // ---- down path, k = 5 .. 2: dz of z_k (k = 5: Gd[5] itself; else from (Gd[k], z[k], cf_d[k]))
  for (int k = 5; k >= 2; --k) {
    const int co = n.C[k], ci = n.C[k - 1], Hc = n.Hd[k];
    const USrc dz = k == 5 ? src_dz(Gd[5], nullptr, nullptr, co) : src_dz(Gd[k], sv.z[k], cf_d[k], co);
    const USrc in = src_act(sv.z[k - 1], ci, sv.sc_d[k - 1], sv.sh_d[k - 1], 0.2f);
    // dW[co][ci][t] = sum_m dz[m, co] in[gather_S(m, t), ci]
    GA_TRY(wgrad(Hc, co, ci, 1, &dz, in, (int64_t)ci * 16, 16, gr->dWd[k - 1]));
    // input gradient (pattern T over dz), times leaky', ADDED to the skip contribution already in Gd[k-1]
    UGemm g{};
    g.B = B; g.Hc = g.Wc = Hc; g.N = ci; g.Ctot = co; g.nsrc = 1; g.src[0] = dz; g.Wp = wpk_d[k]; g.ndst = 1;
    UDst d{}; d.out = Gd[k - 1]; d.z = sv.z[k - 1]; d.sc = sv.sc_d[k - 1]; d.sh = sv.sh_d[k - 1]; d.C = ci; d.dmode = 1;
    d.slope = 0.2f; d.accumulate = 1;
    const bool has_bn = (k - 1) >= 2;     // z2..z4
    if (has_bn) d.part = part;
    g.dst[0] = d;
    GA_TRY(launch_ugemm(1, g, st));
    if (has_bn) {
      const int rows = (B * Hc * Hc + 31) / 32;
      GA_TRY(bn_coef(ci, 4 * rows, (float)(B * n.Hd[k - 1] * n.Hd[k - 1]), sv.sc_d[k - 1], sv.sh_d[k - 1], cf_d[k - 1]));
    }
  }
  // ---- conv1: dW1 = sum dz1 x (dz1 = Gd[1]: no BatchNorm)
  {
    const int M = B * n.Hd[1] * n.Hd[1], nchunk = (M + 255) / 256;
    const int KC = n.cin * 16 * n.C[1];
    hipStream_t ws;
    GA_TRY(fork(&ws));
    hipLaunchKernelGGL(uconv1_wgrad_kernel, dim3(n.cin * 16, nchunk, (n.C[1] + 31) / 32), dim3(256), 0, ws, B, n.cin, n.S,
                       n.C[1], 256, x, Gd[1], wgpart);
    hipLaunchKernelGGL(usum_chunks_kernel, dim3((KC + 255) / 256), dim3(256), 0, ws, nchunk, (int64_t)KC, wgpart, gr->dWd[0]);
    GA_TRY(check_hip(hipGetLastError(), "uconv1_wgrad kernels"));
  }
  if (side) {      // every gradient is ready in main-stream order
    GA_TRY(check_hip(hipEventRecord(ev_side, side), "hipEventRecord"));
    GA_TRY(check_hip(hipStreamWaitEvent(st, ev_side, 0), "hipStreamWaitEvent"));
  }
  return 0;
}

}  // extern "C"
