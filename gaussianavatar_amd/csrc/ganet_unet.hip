// ganet_unet.hip — the stage-2 pose encoder (UnetNoCond5DS, /root/reference/model/modules.py:185-232; blocks :62-111)
// forward and backward as hand-written kernels: no im2col / col2im, no vendor GEMM, no torch element-wise launches.
//
//   down  conv_k = Conv2d(4x4, stride 2, pad 1, no bias) [+ BatchNorm2d(affine=False)]       k = 1..5, 128^2 -> 4^2
//   up    upconv_k = ReLU -> ConvTranspose2d(4x4, stride 2, pad 1) [+ BatchNorm2d] -> cat skip   k = 1..5, 4^2 -> 128^2
//   (the reference's LeakyReLU(0.2, inplace) in front of conv2..5 also acts on the skip tensors: a_k = leaky(bn(conv_k)))
//
// Everything is channels-last ([B,H,W,C]) and NOTHING normalised or activated is ever stored: a layer writes its raw
// convolution output z (+ the BatchNorm column sums of z out of the epilogue), and every consumer applies its operand's
// prologue while it loads — ONE formula, act(k0 g + k1 x + k2), with a [3][C] coefficient block per tensor written by
// whoever produces it (raw: (0, 1, 0); act(bn(z)): (0, scale, shift); dz = cA Gy + cQ z + cP: the BatchNorm backward
// folded) — the decoder's scheme (ganet.h). The concatenations are virtual: a consumer's K dimension runs over two source
// tensors ("segments"). Backward likewise: a layer keeps Gy = dL/d(BatchNorm output) (written, x act', by the dgrad
// kernel of its consumer(s), with the sums of Gy and Gy.y^ in that epilogue) and dz is assembled on load by the layer's
// own dgrad / wgrad kernels.
//
// Two gather patterns cover all four convolution flavours (c = coarse grid [B,Hc,Wc], f = fine grid [B,2Hc,2Wc]):
//   S  out on c, 16 taps (ky,kx) from f at (2y + ky - 1, 2x + kx - 1)      conv forward, transposed-conv input gradient
//   T  out on f, per parity class of (y,x) 2 x 2 taps from c               transposed-conv forward, conv input gradient
// as one GEMM per output tile of 32 pixels x 32 channels on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: exact
// fp32 products); lane (half, row) loads 16 consecutive channels of its pixel per 32-channel chunk (the reduction order
// over k is free). A tile's (chunk, tap) steps are cut over the waves of a workgroup (partial tiles through LDS) and over
// workgroups (partial tiles through global memory, a ticket per tile: ugemm_kernel). The weight gradient reduces over the
// coarse pixels with the pixel index as the MFMA's k: one dword per lane and operand per step, per-lane channel
// coefficients, deterministic partial tiles per 256-pixel chunk (uwgrad_kernel); ganet_unet_bwd issues it on a side stream.
// The maps are tiny (64^2 .. 4^2 pixels x <= 512 channels): 1.2 GFLOP forward per frame, launch- and latency-bound — what a
// launch costs here, and the load-scheduling mistakes that cost most, are written up in profiles/r05_pose_encoder.md.
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
  const float* k3;      // [3][C]: the prologue as act(k0 g + k1 x + k2) — mode 0: (0, 1, 0); mode 1: (0, scale, shift); mode 2:
                        // (cA, cQ, cP). Every producer of a scale / shift / coefficient writes this layout, so that a consumer
                        // stages it with three plain loads (a load under a test of the mode is a serialised round trip)
  int C, mode;
  float slope;          // of the activation (1: none)
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
  int nwk, ks;            // waves of a workgroup that share a tile (1, 2, 4, 8); workgroups that share a tile (K split)
  float* kpart;           // ks > 1: [tile][ks][16][64] partial tiles
  unsigned* ticket;       //         [tile] arrival counters (zero before the launch; the last workgroup leaves zero again)
};

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

constexpr int UG_WG = 512;       // 8 waves
constexpr int UG_MAXC = 1024;    // K channels whose prologue coefficients are staged in LDS

// One GEMM step = (32-channel chunk, tap): 16 MFMAs (~0.45 us) behind two or three 64-byte loads per lane that come from L2
// or HBM (1 - 3 us). The maps are tiny — M = 32 .. 8192 pixels a launch — so what a launch takes is the serial chain of
// steps of its slowest wave plus launch overhead, and the first version (8 waves of ONE workgroup per tile, each step's
// loads waited for before its MFMAs) spent 16 dependent round trips on a 4 x 4 map. Now:
//   * a tile's steps are cut nwk x ks ways: nwk waves of a workgroup (partial tiles added through LDS) and ks workgroups
//     (partial tiles through global memory; the LAST workgroup to arrive — a ticket per tile — adds them in a fixed
//     order and runs the epilogue: deterministic, no extra launch). The host picks them so that a launch has ~2048 waves;
//   * with nwk < 8 a workgroup holds 8 / nwk tiles (large maps: no partial sums at all);
//   * the loads of step s + 1 are in flight while step s computes (two register stages);
//   * the prologue coefficients of every K channel sit in LDS, the operand modes fold into one formula
//     act(k0 g + k1 x + k2) (HASG: the source is a dz assembled from (g, z)) — no branch between a load and its use.
// PATTERN 0 = S, 1 = T (blockIdx.z = parity class + 4 x K-split index)
template <int PATTERN, bool HASG>
__global__ void __launch_bounds__(UG_WG)
ugemm_kernel(UGemm p) {
  __shared__ float s_part[UG_WG / 64][16][64];
  __shared__ float s_coef[3][UG_MAXC];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, r = lane & 31;
  const int Hc = p.Hc, Wc = p.Wc;                    // (powers of two)
  const int lw = 31 - __clz(Wc), lhw = lw + (31 - __clz(Hc));
  const int M = p.B * Hc * Wc;                       // output pixels (of this parity class)
  const int mtiles = (M + 31) >> 5;
  constexpr int NCLS = PATTERN ? 4 : 1;
  constexpr int NTAP = PATTERN ? 4 : 16, LTAP = PATTERN ? 2 : 4;
  const int nwk = p.nwk, ks = p.ks;
  const int cls = PATTERN ? (int)(blockIdx.z & 3) : 0;
  const int ksi = PATTERN ? (int)(blockIdx.z >> 2) : (int)blockIdx.z;
  const int py = cls >> 1, px = cls & 1;
  const int lnwk = 31 - __clz(nwk);
  const int kw = wave & (nwk - 1);
  const int mtile = (int)blockIdx.x * (8 >> lnwk) + (wave >> lnwk);
  const bool tile_on = mtile < mtiles;
  const int m0 = mtile * 32, n0 = blockIdx.y * 32;
  // prologue coefficients of all K channels -> LDS: requested here, stored behind the first step's requests (below)
  constexpr int NST = UG_MAXC / UG_WG;
  float kst[2][NST][3];
#pragma unroll
  for (int sidx = 0; sidx < 2; ++sidx) {
    const USrc& s = p.src[min(sidx, p.nsrc - 1)];                 // (uniform index; a missing second segment re-reads the first)
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int c = min((int)threadIdx.x + j * UG_WG, s.C - 1);   // (clamped: not stored past the end)
      kst[sidx][j][0] = s.k3[c]; kst[sidx][j][1] = s.k3[s.C + c]; kst[sidx][j][2] = s.k3[2 * s.C + c];
    }
  }
  // this lane's output pixel (row r of the tile) on the COARSE index space
  const int m = m0 + r;
  const bool row_on = tile_on && m < M;
  const int mm = row_on ? m : 0;
  const int b = mm >> lhw, yc = (mm >> lw) & (Hc - 1), xc = mm & (Wc - 1);
  const int Hs = PATTERN ? Hc : 2 * Hc, Ws = PATTERN ? Wc : 2 * Wc;     // source grid
  const int nch0 = p.src[0].C >> 5;
  const int S = (p.Ctot >> 5) << LTAP;               // steps of a tile: (chunk, tap), tap fastest
  const int way = ksi * nwk + kw, P = nwk * ks;
  const int slo = tile_on ? (int)(((int64_t)way * S) / P) : 0;
  const int shi = tile_on ? (int)(((int64_t)(way + 1) * S) / P) : 0;
  const float slope0 = p.src[0].slope, slope1 = p.nsrc > 1 ? p.src[1].slope : 1.f;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  struct Stage { f32x4 a[4], g[4], w[4]; bool on; };
  auto issue = [&](int s, Stage& R) {
    const int c = s >> LTAP, t = s & (NTAP - 1);
    const bool seg1 = c >= nch0;
    const USrc& src = p.src[seg1 ? 1 : 0];
    const int cl = ((seg1 ? c - nch0 : c) << 5) + 16 * h;            // this lane's 16 channels of the chunk, in the segment
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
    R.on = row_on && sy >= 0 && sy < Hs && sx >= 0 && sx < Ws;
    const int64_t sp = R.on ? ((int64_t)(b * Hs + sy) * Ws + sx) * src.C + cl : cl;
    const float* wp = p.Wp + ((int64_t)wt * p.N + n0 + r) * p.Ctot + (c << 5) + 16 * h;
#pragma unroll
    for (int u = 0; u < 4; ++u) R.a[u] = *reinterpret_cast<const f32x4*>(src.x + sp + 4 * u);
    if (HASG) {
#pragma unroll
      for (int u = 0; u < 4; ++u) R.g[u] = *reinterpret_cast<const f32x4*>(src.g + sp + 4 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) R.w[u] = *reinterpret_cast<const f32x4*>(wp + 4 * u);
  };
  auto consume = [&](int s, Stage& R) {
    const int c = s >> LTAP;
    const float slope = c >= nch0 ? slope1 : slope0;
    const int ci = (c << 5) + 16 * h;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 k1 = *reinterpret_cast<const f32x4*>(&s_coef[1][ci + 4 * u]);
      const f32x4 k2 = *reinterpret_cast<const f32x4*>(&s_coef[2][ci + 4 * u]);
      f32x4 k0;
      if (HASG) k0 = *reinterpret_cast<const f32x4*>(&s_coef[0][ci + 4 * u]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = fmaf(k1[e], R.a[u][e], k2[e]);
        if (HASG) v = fmaf(k0[e], R.g[u][e], v);
        v = leaky(v, slope);
        // (zero padding / rows past the end: a select, not a product with 0)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(R.on ? v : 0.f, R.w[u][e], acc, 0, 0, 0);
      }
    }
  };
  // (The look-ahead is UNCONDITIONAL — past the last step it re-requests that step — because a load issued under a branch
  // makes the number of loads in flight unknown at the join, and the compiler then waits for all of them: vmcnt(0) in
  // front of every step, no overlap at all. That is what the first version of this loop compiled to.)
  Stage R0, R1;
  __builtin_amdgcn_sched_barrier(0);
  issue(min(slo, S - 1), R0);                  // (a wave without steps requests a valid address and drops it)
  __builtin_amdgcn_sched_barrier(0);
  {
    int coff = 0;
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      if (sidx < p.nsrc) {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
          const int c = threadIdx.x + j * UG_WG;
          if (c < p.src[sidx].C) {
            s_coef[0][coff + c] = kst[sidx][j][0]; s_coef[1][coff + c] = kst[sidx][j][1]; s_coef[2][coff + c] = kst[sidx][j][2];
          }
        }
        coff += p.src[sidx].C;
      }
    }
  }
  __syncthreads();
  if (slo < shi) {
    int s = slo;
    for (;;) {
      issue(min(s + 1, shi - 1), R1);
      __builtin_amdgcn_sched_barrier(0);       // (or the scheduler sinks the requests below the MFMAs: one stage again)
      consume(s, R0);
      if (++s >= shi) break;
      issue(min(s + 1, shi - 1), R0);
      __builtin_amdgcn_sched_barrier(0);
      consume(s, R1);
      if (++s >= shi) break;
    }
  }
  // the partial tiles of the waves that share this tile
  if (nwk > 1) {
    if (kw > 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) s_part[wave][q][lane] = acc[q];
    }
    __syncthreads();
    if (kw > 0) return;
    for (int w = 1; w < nwk; ++w) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] += s_part[wave + w][q][lane];
    }
  }
  if (!tile_on) return;
  // ... and of the workgroups that do
  if (ks > 1) {
    const int tile_id = (cls * mtiles + mtile) * (int)gridDim.y + (int)blockIdx.y;
    float* mine = p.kpart + ((int64_t)tile_id * ks + ksi) * 1024;
#pragma unroll
    for (int q = 0; q < 16; ++q) mine[q * 64 + lane] = acc[q];
    __threadfence();                                    // release: the partial tile before the ticket
    unsigned old = 0;
    if (lane == 0) old = atomicAdd(&p.ticket[tile_id], 1u);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(ks - 1)) return;
    __threadfence();                                    // acquire: the other workgroups' tiles after the ticket
    if (lane == 0) p.ticket[tile_id] = 0u;
    const float* all = p.kpart + (int64_t)tile_id * ks * 1024;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int k = 0; k < ks; ++k) {                      // fixed order: the sum does not depend on who came last
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] += all[(int64_t)k * 1024 + q * 64 + lane];
    }
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
  // (every load of the tile first — with a load, its use and a store per element inside the branches, the 16 elements
  // took 16 .. 32 memory round trips one after the other: most of what an input-gradient launch cost)
  int64_t op[16];
  bool ok[16];
  float zv[16], ov[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
    ok[q] = m0 + row < M;
    const int mr = ok[q] ? m0 + row : m0;
    if (PATTERN == 0) {
      op[q] = (int64_t)mr * d.C + n;
    } else {
      const int b2 = mr >> lhw, y2 = (mr >> lw) & (Hc - 1), x2 = mr & (Wc - 1);
      op[q] = ((int64_t)(b2 * 2 * Hc + 2 * y2 + py) * (2 * Wc) + 2 * x2 + px) * d.C + n;
    }
  }
  if (d.dmode == 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) zv[q] = d.z[op[q]];
  }
  if (d.accumulate) {
#pragma unroll
    for (int q = 0; q < 16; ++q) ov[q] = d.out[op[q]];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float val = acc[q] + bias;
    if (d.dmode == 1) {
      const float yv = fmaf(zv[q], dsc, dsh);
      val *= yv > 0.f ? 1.f : d.slope;
      if (d.accumulate) val += ov[q];
      if (ok[q]) {
        d.out[op[q]] = val;
        s1 += val;
        s2 = fmaf(val, yv, s2);
      }
    } else {
      if (d.accumulate) val += ov[q];
      if (ok[q]) {
        d.out[op[q]] = val;
        const float dv = val - sshift;
        s1 += dv;
        s2 = fmaf(dv, dv, s2);
      }
    }
  }
  if (d.part) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (h == 0) {
      const int prow = cls * mtiles + mtile;
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
  float* dW;              // != NULL (one chunk only): the tile goes straight to dW[i * sI + j * sJ + t], no reduction launch
  long long sI, sJ;
};

// operand element with its per-channel prologue as ONE formula, act(k0 g + k1 x + k2) (mode 0: 0, 1, 0; mode 1: 0, sc, sh
// with the activation's slope; mode 2: cA, cQ, cP), HASG = the operand has a g tensor. The mode is a launch constant, and
// testing it per element (as the first version did) put a branch around every load: the 32 .. 48 loads of a 32-row step
// waited for each other one by one, ~1 us each — that, not the arithmetic, was the 35 us a launch took.
struct UCoef { float k0, k1, k2, slope; };
__device__ __forceinline__ UCoef ucoef_of(const USrc& s, int c) {
  UCoef k;
  k.k0 = s.k3[c]; k.k1 = s.k3[s.C + c]; k.k2 = s.k3[2 * s.C + c]; k.slope = s.slope;
  return k;
}
__device__ __forceinline__ float ucoef_apply(const UCoef& k, float x, float g) {
  const float v = fmaf(k.k0, g, fmaf(k.k1, x, k.k2));
  return v > 0.f ? v : v * k.slope;
}

constexpr int UW_WG = 256;       // four waves share a (tap, tile, chunk): a quarter of the chunk's rows each

template <bool UG, bool VG>
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
  const UCoef ku = ucoef_of(U, ic), kv = ucoef_of(V, jc);
  const float* __restrict__ ux = U.x;
  const float* __restrict__ ug = UG ? U.g : U.x;
  const float* __restrict__ vx = V.x;
  const float* __restrict__ vg = VG ? V.g : V.x;
  const int UC = U.C, VC = V.C;
  const int ky = t >> 2, kx = t & 3;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int mb = mlo; mb < mhi; mb += 32) {
    float uxr[16], ugr[16], vxr[16], vgr[16];
    bool uon[16], von[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {                  // every load of the step first: one memory round trip
      const int m = mb + 2 * s + h;
      uon[s] = m < mhi;
      const int mm = uon[s] ? m : mlo;
      const int b = mm >> lhw, yc = (mm >> lw) & (Hc - 1), xc = mm & (Wc - 1);
      const int sy = 2 * yc + ky - 1, sx = 2 * xc + kx - 1;
      von[s] = uon[s] && sy >= 0 && sy < 2 * Hc && sx >= 0 && sx < 2 * Wc;
      const int64_t up = (int64_t)mm * UC + ic;
      const int64_t vp = von[s] ? ((int64_t)(b * 2 * Hc + sy) * (2 * Wc) + sx) * VC + jc : jc;
      uxr[s] = ux[up];
      ugr[s] = UG ? ug[up] : 0.f;
      vxr[s] = vx[vp];
      vgr[s] = VG ? vg[vp] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float uu = uon[s] ? ucoef_apply(ku, uxr[s], ugr[s]) : 0.f;
      const float vv = von[s] ? ucoef_apply(kv, vxr[s], vgr[s]) : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(uu, vv, acc, 0, 0, 0);
    }
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
    const float v = (acc[q] + s_part[0][q][lane]) + (s_part[1][q][lane] + s_part[2][q][lane]);
    if (p.dW) p.dW[(int64_t)(i0 + row) * p.sI + (int64_t)(j0 + r) * p.sJ + t] = v;
    else out[(int64_t)(i0 + row) * p.J + j0 + r] = v;
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

// Wp[t][n][c] = W[n * sn + c * sc + t] for up to 9 layers in one launch (blockIdx.y = layer): a thread takes one (n, c)
// pair — its 16 taps are contiguous in every source layout (64 bytes in, 16 stores that are coalesced across the
// threads' c). Workgroup (0, 0) also zeroes the K-split tickets of the GEMM launches that follow.
struct UPackJobs {
  const float* W[9]; float* Wp[9]; int N[9], C[9]; long long sn[9], sc[9];
  unsigned* ticket; int nticket;
  float* ident[2]; int identC[2];          // identity prologue blocks [3][C] = (0, 1, 0) to fill (C = 0: none)
};
__global__ void __launch_bounds__(256)
upack_kernel(UPackJobs jobs) {
  const int j = blockIdx.y;
  if (blockIdx.x == 0 && j == 0) {
    for (int i = threadIdx.x; i < jobs.nticket; i += 256) jobs.ticket[i] = 0u;
    for (int b = 0; b < 2; ++b)
      for (int i = threadIdx.x; i < 3 * jobs.identC[b]; i += 256)
        jobs.ident[b][i] = (i >= jobs.identC[b] && i < 2 * jobs.identC[b]) ? 1.f : 0.f;
  }
  const int N = jobs.N[j], C = jobs.C[j];
  const int pairs = N * C;
  const float* __restrict__ W = jobs.W[j];
  float* __restrict__ Wp = jobs.Wp[j];
  const long long sn = jobs.sn[j], sc = jobs.sc[j];
  for (int e = (int)blockIdx.x * 256 + (int)threadIdx.x; e < pairs; e += (int)gridDim.x * 256) {
    const int n = e / C, c = e - n * C;
    const f32x4* src = reinterpret_cast<const f32x4*>(W + n * sn + c * sc);
    const f32x4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Wp[(int64_t)(t + 0) * pairs + e] = v0[t];
      Wp[(int64_t)(t + 4) * pairs + e] = v1[t];
      Wp[(int64_t)(t + 8) * pairs + e] = v2[t];
      Wp[(int64_t)(t + 12) * pairs + e] = v3[t];
    }
  }
}

// BatchNorm statistics from the forward epilogue's partials (sums about the running mean) -> scale = rstd,
// shift = -mean rstd (affine = False), running statistics as F.batch_norm(training = True)
__global__ void __launch_bounds__(64)
ubn_fwd_kernel(int nparts, int C, float count, const float* __restrict__ part, const float* __restrict__ stat_shift,
               float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
               long long* __restrict__ nbt, float* __restrict__ k0, float* __restrict__ sc, float* __restrict__ sh) {
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
    k0[n] = 0.f;                       // (k0, sc, sh) = the consumers' [3][C] prologue block (USrc.k3)
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
                                float* __restrict__ k0, float* __restrict__ sc, float* __restrict__ sh) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= C) return;
  const float rstd = 1.0f / sqrtf(rv[n] + eps);
  k0[n] = 0.f;
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
  for (int ci = 0; ci < Cin; ++ci) {
    float xv[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {                   // the 16 taps' loads together (no branch between a load and its use)
      const int sy = 2 * oy + (t >> 2) - 1, sx = 2 * ox + (t & 3) - 1;
      const bool on = sy >= 0 && sy < S && sx >= 0 && sx < S;
      const float v = x[on ? (((int64_t)b * Cin + ci) * S + sy) * S + sx : 0];
      xv[t] = on ? v : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int wi = (co * Cin + ci) * 16 + t;
      acc = fmaf(lds_w ? s_w[wi] : W[wi], xv[t], acc);
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
  if (co < Cout) {
    const int lh = 31 - __clz(Ho);                   // (S is a power of two: unet_ok)
    for (int m0 = mlo + rl; m0 < mhi; m0 += 64) {    // eight rows' loads in flight
      float dv[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = m0 + 8 * u;
        const int mm = m < mhi ? m : mlo;
        const int b = mm >> (2 * lh), oy = (mm >> lh) & (Ho - 1), ox = mm & (Ho - 1);
        const int sy = 2 * oy + ky - 1, sx = 2 * ox + kx - 1;
        const bool on = m < mhi && sy >= 0 && sy < S && sx >= 0 && sx < S;
        dv[u] = dz[(int64_t)mm * Cout + co];
        const float xr = x[on ? (((int64_t)b * Cin + ci) * S + sy) * S + sx : 0];
        xv[u] = on ? xr : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(dv[u], xv[u], acc);
    }
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
  if (c < C) {
    int m = mlo + rl;
    for (; m + 28 < mhi; m += 32) {                  // eight independent loads in flight (the plain loop waited for each)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = g[(int64_t)(m + 4 * u) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; m < mhi; m += 4) s += g[(int64_t)m * C + c];
  }
  s_acc[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C)
    part[(int64_t)blockIdx.y * C + c] = (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x]) + (s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
}

// identity prologue blocks [3][n] = (0, 1, 0) for the tensors without BatchNorm
__global__ void ufill_kernel(int n0, int n1, float* k3a, float* k3b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n0) { k3a[i] = 0.f; k3a[n0 + i] = 1.f; k3a[2 * n0 + i] = 0.f; }
  if (i < n1) { k3b[i] = 0.f; k3b[n1 + i] = 1.f; k3b[2 * n1 + i] = 0.f; }
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
  // per tensor one [3][C] prologue block (k0 = 0 | scale | shift): USrc.k3 = sc - C
  for (int k = 1; k <= 5; ++k) { s.sc_d[k] = base + o + n.C[k]; s.sh_d[k] = base + o + 2 * n.C[k]; o += al(3 * n.C[k]); }
  for (int k = 1; k <= 4; ++k) { s.sc_u[k] = base + o + n.CU[k]; s.sh_u[k] = base + o + 2 * n.CU[k]; o += al(3 * n.CU[k]); }
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

// K-split scratch of the GEMM launches: partial tiles + tickets (carved from the workspace after the BatchNorm partials)
constexpr int UG_KTILES = 512;                 // tiles x ks of one launch (the heuristic below keeps it under this)
constexpr int UG_TICKETS = 512;
size_t ksplit_floats() { return al((size_t)UG_KTILES * 1024) + al(UG_TICKETS); }
struct KSplit { float* kpart; unsigned* ticket; };
KSplit carve_ksplit(float* base) { return {base, reinterpret_cast<unsigned*>(base + al((size_t)UG_KTILES * 1024))}; }

int launch_ugemm(int pattern, UGemm g, const KSplit& k, hipStream_t st) {
  const int M = g.B * g.Hc * g.Wc;
  const int mtiles = (M + 31) / 32, ncls = pattern ? 4 : 1;
  const int T = mtiles * (g.N / 32) * ncls;                  // tiles
  const int S = (g.Ctot / 32) * (pattern ? 4 : 16);          // steps of a tile
  if (g.Ctot > UG_MAXC) { set_error("ganet_unet: more than 1024 input channels in one convolution"); return 1; }
  // ~2048 waves a launch (two per SIMD): the waves of a workgroup first, then workgroups while a wave still has more than
  // two steps (a K split across workgroups costs a trip through memory and a ticket)
  int nwk = 1, ks = 1;
  while (nwk < 8 && T * nwk * 2 <= 2048 && nwk * 2 <= S) nwk *= 2;
  if (nwk == 8)
    while (S / (nwk * ks) > 2 && T * nwk * ks * 2 <= 4096 && T * ks * 2 <= UG_KTILES && T <= UG_TICKETS) ks *= 2;
  g.nwk = nwk; g.ks = ks; g.kpart = k.kpart; g.ticket = k.ticket;
  const int tpw = 8 / nwk;
  const dim3 grid((mtiles + tpw - 1) / tpw, g.N / 32, ncls * ks);
  const bool hasg = g.src[0].mode == 2;
  if (hasg && g.nsrc != 1) { set_error("ganet_unet: a dz source comes alone"); return 1; }
  if (pattern) {
    if (hasg) hipLaunchKernelGGL((ugemm_kernel<1, true>), grid, dim3(UG_WG), 0, st, g);
    else hipLaunchKernelGGL((ugemm_kernel<1, false>), grid, dim3(UG_WG), 0, st, g);
  } else {
    if (hasg) hipLaunchKernelGGL((ugemm_kernel<0, true>), grid, dim3(UG_WG), 0, st, g);
    else hipLaunchKernelGGL((ugemm_kernel<0, false>), grid, dim3(UG_WG), 0, st, g);
  }
  return check_hip(hipGetLastError(), "ugemm_kernel");
}
struct PackList {
  UPackJobs jobs{};
  int n = 0;
  void add(int N, int C, int64_t sn, int64_t sc, const float* W, float* Wp) {
    jobs.W[n] = W; jobs.Wp[n] = Wp; jobs.N[n] = N; jobs.C[n] = C; jobs.sn[n] = sn; jobs.sc[n] = sc; ++n;
  }
  int launch(const KSplit& k, hipStream_t st) {
    if (!n) return 0;
    jobs.ticket = k.ticket; jobs.nticket = UG_TICKETS;
    hipLaunchKernelGGL(upack_kernel, dim3(128, n), dim3(256), 0, st, jobs);
    return check_hip(hipGetLastError(), "upack_kernel");
  }
};
// act(bn(z)): sc / sh are the second and third row of the tensor's [3][C] prologue block (carve_saved)
USrc src_act(const float* z, int C, const float* sc, const float* sh, float slope) {
  USrc s{}; s.x = z; s.k3 = sc - C; s.C = C; s.mode = 1; s.slope = slope; (void)sh; return s;
}
// dz of a layer: cA g + cQ z + cP (coef [3][C], ubn_bwd_kernel), or g itself (no BatchNorm: `ident` = an identity block)
USrc src_dz(const float* g, const float* z, const float* coef, const float* ident, int C) {
  USrc s{}; s.x = z; s.g = g; s.k3 = coef; s.C = C; s.mode = coef ? 2 : 0; s.slope = 1.f;
  if (!coef) { s.x = g; s.k3 = ident; }
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
  return (packed_floats(n) + max_part_floats(n) + ksplit_floats()) * sizeof(float);
}

// x: [B, cin, S, S] NCHW; out: [B, S, S, cout] channels-last. training != 0: batch statistics (and the running
// statistics updated); else the running statistics normalise.
int ganet_unet_fwd(const GanetUnetParams* p, int32_t B, const float* x, int32_t training, float* saved, float* out,
                   void* workspace, size_t workspace_bytes, void* stream_) {
  if (!unet_ok(p, B) || !x || !saved || !out || !workspace) {
    set_error("ganet_unet_fwd: invalid arguments (nf and cout multiples of 32, S a power of two >= 32, cin <= 8)");
    return 1;
  }
  if (workspace_bytes < ganet_unet_fwd_workspace(p, B)) { set_error("ganet_unet_fwd: workspace too small"); return 2; }
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const Net n = make_net(p, B);
  const Saved sv = carve_saved(n, saved);
  float* wsf = static_cast<float*>(workspace);
  float* part = wsf + packed_floats(n);
  const KSplit ksp = carve_ksplit(part + max_part_floats(n));
  // ones / zeros for the tensors without BatchNorm (z1, z5)
  hipLaunchKernelGGL(ufill_kernel, dim3((std::max(n.C[1], n.C[5]) + 255) / 256), dim3(256), 0, st, n.C[1], n.C[5],
                     sv.sc_d[1] - n.C[1], sv.sc_d[5] - n.C[5]);
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
    GA_TRY(pl.launch(ksp, st));
  }
  // BatchNorm of a layer's raw output from the partials its forward launch left
  auto bn = [&](int idx, int C, int nparts, float count, float* sc, float* sh) -> int {
    if (training) {
      hipLaunchKernelGGL(ubn_fwd_kernel, dim3(C), dim3(64), 0, st, nparts, C, count, part, p->running_mean[idx], p->eps,
                         p->momentum, p->running_mean[idx], p->running_var[idx],
                         reinterpret_cast<long long*>(p->num_batches_tracked[idx]), sc - C, sc, sh);
    } else {
      if (!p->running_mean[idx] || !p->running_var[idx]) { set_error("ganet_unet_fwd: evaluation needs running statistics"); return 1; }
      hipLaunchKernelGGL(ubn_eval_kernel, dim3((C + 63) / 64), dim3(64), 0, st, C, p->running_mean[idx], p->running_var[idx],
                         p->eps, sc - C, sc, sh);
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
    GA_TRY(launch_ugemm(0, g, ksp, st));
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
    GA_TRY(launch_ugemm(1, g, ksp, st));
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
  size_t o = packed_floats(n) + max_part_floats(n) + ksplit_floats() + al(3 * n.cout) + al(3 * n.C[5]);
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
  const KSplit ksp = carve_ksplit(part + max_part_floats(n));
  float* o = part + max_part_floats(n) + ksplit_floats();
  float* ident_out = o; o += al(3 * n.cout);        // identity prologue blocks (d_out and Gd[5] are dz themselves)
  float* ident_z5 = o; o += al(3 * n.C[5]);
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
    pl.jobs.ident[0] = ident_out; pl.jobs.identC[0] = n.cout;
    pl.jobs.ident[1] = ident_z5; pl.jobs.identC[1] = n.C[5];
    GA_TRY(pl.launch(ksp, st));
  }
  (void)wp;
  auto wgrad = [&](int Hc, int I, int J, int nu, const USrc* U, const USrc& V, int64_t sI, int64_t sJ, float* dW) -> int {
    UWgrad w{};
    w.B = B; w.Hc = w.Wc = Hc; w.I = I; w.J = J; w.nu = nu; w.chunk_rows = 256;
    for (int i = 0; i < nu; ++i) w.U[i] = U[i];
    w.V = V; w.part = wgpart;
    const int M = B * Hc * Hc, nchunk = (M + 255) / 256;
    if (nchunk == 1) { w.dW = dW; w.sI = sI; w.sJ = sJ; }      // a single chunk's tiles ARE the gradient
    hipStream_t ws;
    GA_TRY(fork(&ws));
    bool ug = false;
    for (int i = 0; i < nu; ++i) ug = ug || U[i].mode == 2;
    const bool vg = V.mode == 2;
    const dim3 grid(16, (I / 32) * (J / 32), nchunk);
    if (ug && vg) { set_error("ganet_unet_bwd: both weight-gradient operands assembled from (g, z)"); return 1; }
    if (ug) hipLaunchKernelGGL((uwgrad_kernel<true, false>), grid, dim3(UW_WG), 0, ws, w);
    else if (vg) hipLaunchKernelGGL((uwgrad_kernel<false, true>), grid, dim3(UW_WG), 0, ws, w);
    else hipLaunchKernelGGL((uwgrad_kernel<false, false>), grid, dim3(UW_WG), 0, ws, w);
    GA_TRY(check_hip(hipGetLastError(), "uwgrad_kernel"));
    if (nchunk == 1) return 0;
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
    const USrc dz = k == 5 ? src_dz(d_out, nullptr, nullptr, ident_out, co) : src_dz(Gu[k], sv.zT[k], cf_u[k], nullptr, co);
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
    GA_TRY(launch_ugemm(0, g, ksp, st));
    if (k >= 2) GA_TRY(bn_coef(n.CU[k - 1], (B * Hc * Hc + 31) / 32, (float)(B * Hc * Hc), sv.sc_u[k - 1], sv.sh_u[k - 1],
                               cf_u[k - 1]));
  }
  // ---- down path, k = 5 .. 2: dz of z_k (k = 5: Gd[5] itself; else from (Gd[k], z[k], cf_d[k]))
  for (int k = 5; k >= 2; --k) {
    const int co = n.C[k], ci = n.C[k - 1], Hc = n.Hd[k];
    const USrc dz = k == 5 ? src_dz(Gd[5], nullptr, nullptr, ident_z5, co) : src_dz(Gd[k], sv.z[k], cf_d[k], nullptr, co);
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
    GA_TRY(launch_ugemm(1, g, ksp, st));
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
