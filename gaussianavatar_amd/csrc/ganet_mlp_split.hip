// ganet_mlp_split.hip — the streamed-M decoder GEMMs (forward layer, data gradient) on the bf16 matrix pipe
// with exactly split fp32 operands (ganet_split.h), behind the entry points of ganet_mlp.hip / ganet_mlp_bwd.hip:
//
//   forward        Z[M,N]  = [ X1 | softplus(scale . X2 + shift) ] . W^T + b,  column sums of Z in the epilogue
//   data gradient  out[M,O] (+)= (A G + q Z + p)[M,128] . W[128, 0:O],  x softplus'(u_src), sums of G_src, G_src z_src
//
// Mapping: a wave owns 32-row slabs x all columns. Lane (row = lane & 31, kg = lane >> 5) loads 8 consecutive
// k of its row per k-step of 16 (two 16-byte loads), applies the prologue in fp32, splits the 8 values into three
// bf16x8 fragments (44 VALU instructions) and issues 6 MFMAs per 32-column tile against the weight's three bf16
// planes, resident in LDS for the whole kernel (split once per workgroup while staging). A 128 x 128 layer needs
// 8 x 4 x 6 = 192 MFMAs per slab where the fp32 form needs 256 of twice the issue time; measured:
// the bf16 MFMA sustains ~44 cycles per instruction chip-wide, the VALU work (prologue + split, ~100 instructions
// per k-step) hides under it, a slab costs ~9 k cycles of k-loop + 2.5 k of epilogue, and the kernels run at
// 4.2-4.9 TB/s of algorithmic HBM traffic (forward 61 us, data gradient 103 us; fp32 form: 99 / 150 us).
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"
#include "ganet_split.h"

namespace ganet {

namespace {

constexpr int WG = 512;               // 8 waves, one workgroup per CU, two waves per SIMD
constexpr int WAVES = WG / 64;
constexpr int SLAB = 32;
constexpr int BLOCKS = 256;
#ifndef GANET_SPLIT_RING
#define GANET_SPLIT_RING 4
#endif
// scheduling barriers that pin the ring refills: classes that may cross (ganet_mlp_common.h: kSchedMask = VALU, SALU,
// LDS, transcendental; | 0x8 lets MFMAs cross as well)
#ifndef GANET_SPLIT_MASK
#define GANET_SPLIT_MASK kSchedMask
#endif
#ifndef GANET_SPLIT_RING_BWD
#define GANET_SPLIT_RING_BWD 2
#endif

// ---------------------------------------------------------------------------------------------
// forward.  K1 = width of the identity operand as passed (0 or 72), K1S = its k-steps (K1 padded to 16: the
// pad columns meet zero weights), K2S = k-steps of the activated operand (K2 = 16 K2S).
template <int K1, int K2S, int NT>
__global__ void __attribute__((amdgpu_flat_work_group_size(WG, WG), amdgpu_waves_per_eu(2, 2)))
mlp_fwd_split_kernel(int64_t M, int N, const float* __restrict__ x1, int64_t ld1,
                     const float* __restrict__ x2, int64_t ld2, const float* __restrict__ in_scale,
                     const float* __restrict__ in_shift, const float* __restrict__ W,
                     const float* __restrict__ bias, float* __restrict__ z, int64_t ldz,
                     float* __restrict__ col_part, const float* __restrict__ stat_shift, int reverse) {
  constexpr int K1S = (K1 + 15) / 16;
  constexpr int KS = K1S + K2S;
  constexpr int RU = 2 * KS;            // 16-byte units per weight row
  constexpr int NP = NT * 32;
  constexpr int K2 = 16 * K2S;
  constexpr int KW = K1 + K2;           // row length of W in global memory
  constexpr int D2 = (K2S % GANET_SPLIT_RING == 0) ? GANET_SPLIT_RING : K2S;
  constexpr bool TAIL = (K1 % 16) != 0; // last x1 step: the kg = 1 half lies beyond the row
  extern __shared__ u32x4 s_mem[];      // W planes 3 x [NP][RU] | scale [K2/4] | shift [K2/4]
  u32x4* s_w = s_mem;
  float4* s_sc = reinterpret_cast<float4*>(s_mem + 3 * NP * RU);
  float4* s_sh = s_sc + K2 / 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int kg = lane >> 5, col = lane & 31;

  // stage W: 8 consecutive k of one row per item, split into the three planes. Columns of the activated
  // operand carry the ln 2 of the log2-unit softplus (ganet_mlp_common.h).
  for (int i = threadIdx.x; i < NP * RU; i += WG) {
    const int n = i / RU, u = i - n * RU;
    float v[8];
    const bool ident = u < 2 * K1S;
    const int src = ident ? 8 * u : K1 + 8 * (u - 2 * K1S);
    const bool live = n < N && (!ident || 8 * u + 8 <= K1);
    if (live) {
      const float4 lo = *reinterpret_cast<const float4*>(W + (size_t)n * KW + src);
      const float4 hi = *reinterpret_cast<const float4*>(W + (size_t)n * KW + src + 4);
      const float f = ident ? 1.0f : kLn2;
      v[0] = lo.x * f; v[1] = lo.y * f; v[2] = lo.z * f; v[3] = lo.w * f;
      v[4] = hi.x * f; v[5] = hi.y * f; v[6] = hi.z * f; v[7] = hi.w * f;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    u32x4 p1, p2, p3;
    split8(v, p1, p2, p3);
    const int at = split_unit<RU>(n, u);
    s_w[at] = p1; s_w[NP * RU + at] = p2; s_w[2 * NP * RU + at] = p3;
  }
  for (int i = threadIdx.x; i < K2 / 4; i += WG) {      // folded BatchNorm, pre-multiplied by log2(e)
    const float4 a = *reinterpret_cast<const float4*>(in_scale + 4 * i);
    const float4 c = *reinterpret_cast<const float4*>(in_shift + 4 * i);
    s_sc[i] = make_float4(a.x * kLog2e, a.y * kLog2e, a.z * kLog2e, a.w * kLog2e);
    s_sh[i] = make_float4(c.x * kLog2e, c.y * kLog2e, c.z * kLog2e, c.w * kLog2e);
  }
  __syncthreads();

  float csum[NT], csq[NT], bias_r[NT], sshift[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    csum[t] = 0.f; csq[t] = 0.f;
    bias_r[t] = (bias && t * 32 + col < N) ? bias[t * 32 + col] : 0.f;
    sshift[t] = (stat_shift && t * 32 + col < N) ? stat_shift[t * 32 + col] : 0.f;   // see ganet_mlp.hip
  }

  const int64_t nslab = (M + SLAB - 1) / SLAB;
  const int64_t wave_global = (int64_t)blockIdx.x * WAVES + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * WAVES;
  auto phys = [&](int64_t slab) { return reverse ? nslab - 1 - slab : slab; };
  const float *p1c = nullptr, *p2c = nullptr, *p1n = nullptr, *p2n = nullptr;
  auto point_at = [&](int64_t slab, const float*& q1, const float*& q2) {
    const int64_t row = min(phys(slab) * SLAB + col, M - 1);
    if (K1S > 0) q1 = x1 + row * ld1 + 8 * kg;
    if (K2S > 0) q2 = x2 + row * ld2 + 8 * kg;
  };
  struct Raw { float4 lo, hi; };
  auto load1 = [&](const float* q, int s) -> Raw {
    // the tail step's kg = 1 lanes re-read the kg = 0 columns (finite values against zero weights)
    const float* p = q + 16 * s - ((TAIL && s == K1S - 1) ? 8 * kg : 0);
    return Raw{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)};
  };
  auto load2 = [&](const float* q, int s) -> Raw {
    return Raw{*reinterpret_cast<const float4*>(q + 16 * s), *reinterpret_cast<const float4*>(q + 16 * s + 4)};
  };
  Raw r1[K1S > 0 ? K1S : 1], r2[D2 > 0 ? D2 : 1];
  point_at(min(wave_global, nslab - 1), p1c, p2c);
#pragma unroll
  for (int s = 0; s < K1S; ++s) r1[s] = load1(p1c, s);
#pragma unroll
  for (int s = 0; s < (K2S > 0 ? D2 : 0); ++s) r2[s] = load2(p2c, s);

  // The A fragments of k-step s: prologue in fp32 on the step's ring slot, then the exact three-way split.
  struct Pieces { u32x4 a1, a2, a3; };
  auto make_pieces = [&](int s, int soff) -> Pieces {
    float v[8];
    if (s < K1S) {
      const Raw q = r1[s];
      v[0] = q.lo.x; v[1] = q.lo.y; v[2] = q.lo.z; v[3] = q.lo.w;
      v[4] = q.hi.x; v[5] = q.hi.y; v[6] = q.hi.z; v[7] = q.hi.w;
    } else {
      const int s2 = s >= K1S ? s - K1S : 0;
      const Raw q = r2[s2 % D2];
      const float4 sc0 = s_sc[soff + 4 * s2], sc1 = s_sc[soff + 4 * s2 + 1];
      const float4 sh0 = s_sh[soff + 4 * s2], sh1 = s_sh[soff + 4 * s2 + 1];
      v[0] = softplus_log2(fmaf(sc0.x, q.lo.x, sh0.x)); v[1] = softplus_log2(fmaf(sc0.y, q.lo.y, sh0.y));
      v[2] = softplus_log2(fmaf(sc0.z, q.lo.z, sh0.z)); v[3] = softplus_log2(fmaf(sc0.w, q.lo.w, sh0.w));
      v[4] = softplus_log2(fmaf(sc1.x, q.hi.x, sh1.x)); v[5] = softplus_log2(fmaf(sc1.y, q.hi.y, sh1.y));
      v[6] = softplus_log2(fmaf(sc1.z, q.hi.z, sh1.z)); v[7] = softplus_log2(fmaf(sc1.w, q.hi.w, sh1.w));
    }
    Pieces p;
    split8(v, p.a1, p.a2, p.a3);
    return p;
  };
  // slot of step s <- the step that will use it next (D2 steps on, or the same step of the wave's next slab)
  auto refill = [&](int s) {
    if (s < K1S) {
      r1[s] = load1(p1n, s);
    } else {
      const int s2 = s >= K1S ? s - K1S : 0;
      r2[s2 % D2] = (s2 + D2 < K2S) ? load2(p2c, s2 + D2) : load2(p2n, s2 + D2 - K2S);
    }
  };

  // Software pipeline over the k-steps: the fragments of step s + 1 (the next slab's step 0 at the end) are
  // computed next to the 6 NT MFMAs of step s, which do not depend on them.
  Pieces cur = make_pieces(0, 2 * kg);
  for (int64_t slab = wave_global; slab < nslab; slab += wave_stride) {
    point_at(min(slab + wave_stride, nslab - 1), p1n, p2n);
    // the weight fragments must stay in LDS: opaque offsets keep the compiler from hoisting the reads out of
    // the slab loop (and spilling them)
    int wrow = col, ukg = kg, soff = 2 * kg;
    asm volatile("" : "+v"(wrow), "+v"(ukg), "+v"(soff));
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // the slot of step s is free (its fragments are `cur`): refill it first, so that the load has D2 - 1
      // steps to land
      __builtin_amdgcn_sched_barrier(GANET_SPLIT_MASK);
      refill(s);
      __builtin_amdgcn_sched_barrier(GANET_SPLIT_MASK);
      const Pieces nxt = make_pieces((s + 1) % KS, soff);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int at = split_unit<RU>(wrow + t * 32, 2 * s + ukg);
        const u32x4 b1 = s_w[at], b2 = s_w[NP * RU + at], b3 = s_w[2 * NP * RU + at];
        GANET_SPLIT_PRODUCTS(acc[t], cur.a1, cur.a2, cur.a3, b1, b2, b3);
      }
      cur = nxt;
    }
    p1c = p1n; p2c = p2n;
    // epilogue: + bias, store, column statistics. C/D layout: column = lane & 31,
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int64_t row0 = phys(slab) * SLAB;
    if (row0 + SLAB <= M && N == NP) {
      float* zr = z + (row0 + 4 * kg) * ldz + col;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float val = acc[t][r] + bias_r[t];
          zr[((r & 3) + 8 * (r >> 2)) * ldz + t * 32] = val;
          const float d = val - sshift[t];
          csum[t] += d;
          csq[t] = fmaf(d, d, csq[t]);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = t * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
          if (row < M && n < N) {
            const float val = acc[t][r] + bias_r[t];
            z[row * ldz + n] = val;
            const float d = val - sshift[t];
            csum[t] += d;
            csq[t] = fmaf(d, d, csq[t]);
          }
        }
      }
    }
  }
  if (col_part) {       // per-workgroup partial column sums -> [gridDim.x][2][NP], fixed order
    __syncthreads();
    float* s_red = reinterpret_cast<float*>(s_mem);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float s = csum[t] + __shfl_xor(csum[t], 32);
      const float q = csq[t] + __shfl_xor(csq[t], 32);
      if (kg == 0) { s_red[wave * 2 * NP + t * 32 + col] = s; s_red[wave * 2 * NP + NP + t * 32 + col] = q; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NP; i += WG) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += s_red[w * 2 * NP + i];
      col_part[(size_t)blockIdx.x * 2 * NP + i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// data gradient (ganet_mlp_bwd.hip has the algebra): A operand = A G + q Z + p assembled on load.
template <int NT, bool ACCUM, bool SIG>
__global__ void __attribute__((amdgpu_flat_work_group_size(WG, WG), amdgpu_waves_per_eu(2, 2)))
mlp_bwd_split_kernel(int64_t M, int O, const float* __restrict__ g, int64_t ldg,
                     const float* __restrict__ gz, int64_t ldgz, const float* __restrict__ gcoef,
                     const float* __restrict__ W, int64_t ldw, float* __restrict__ out, int64_t ldo,
                     const float* __restrict__ src_z, int64_t ld_src, const float* __restrict__ src_scale,
                     const float* __restrict__ src_shift, float* __restrict__ col_part, int reverse) {
  constexpr int KS = 8, K = 128, RU = 16;
  constexpr int NP = NT * 32;
  constexpr int D = GANET_SPLIT_RING_BWD;   // k-steps in flight (two operands of 8 registers each per step)
  extern __shared__ u32x4 s_mem[];      // Wt planes 3 x [NP][RU] | A [32] | q [32] | p [32]  (float4)
  u32x4* s_w = s_mem;
  float4* s_cA = reinterpret_cast<float4*>(s_mem + 3 * NP * RU);
  float4* s_cq = s_cA + 32;
  float4* s_cp = s_cq + 32;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int kg = lane >> 5, col = lane & 31;

  // stage W[128 (n)][ldw] transposed: row o of the LDS image = column o of W, k = n. Consecutive threads take
  // consecutive columns o (W may be a column slice with any alignment: scalar, coalesced loads).
  for (int i = threadIdx.x; i < NP * RU; i += WG) {
    const int u = i / NP, o = i - u * NP;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = o < O ? W[(size_t)(8 * u + e) * ldw + o] : 0.f;
    u32x4 p1, p2, p3;
    split8(v, p1, p2, p3);
    const int at = split_unit<RU>(o, u);
    s_w[at] = p1; s_w[NP * RU + at] = p2; s_w[2 * NP * RU + at] = p3;
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
    ssc[t] = SIG ? src_scale[o] * kLog2e : 0.f;
    ssh[t] = SIG ? src_shift[o] * kLog2e : 0.f;
  }

  const int64_t nslab = (M + SLAB - 1) / SLAB;
  const int64_t wave_global = (int64_t)blockIdx.x * WAVES + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * WAVES;
  auto phys = [&](int64_t slab) { return reverse ? nslab - 1 - slab : slab; };
  const float *pgc, *pzc, *pgn, *pzn;
  auto point_at = [&](int64_t slab, const float*& qg, const float*& qz) {
    const int64_t row = min(phys(slab) * SLAB + col, M - 1);
    qg = g + row * ldg + 8 * kg;
    qz = gz + row * ldgz + 8 * kg;
  };
  struct Raw { float4 glo, ghi, zlo, zhi; };
  auto load = [&](const float* qg, const float* qz, int s) -> Raw {
    return Raw{*reinterpret_cast<const float4*>(qg + 16 * s), *reinterpret_cast<const float4*>(qg + 16 * s + 4),
               *reinterpret_cast<const float4*>(qz + 16 * s), *reinterpret_cast<const float4*>(qz + 16 * s + 4)};
  };
  Raw ring[D];
  point_at(min(wave_global, nslab - 1), pgc, pzc);
#pragma unroll
  for (int s = 0; s < D; ++s) ring[s] = load(pgc, pzc, s);

  for (int64_t slab = wave_global; slab < nslab; slab += wave_stride) {
    point_at(min(slab + wave_stride, nslab - 1), pgn, pzn);
    int wrow = col, ukg = kg, soff = 2 * kg;
    asm volatile("" : "+v"(wrow), "+v"(ukg), "+v"(soff));
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const Raw q = ring[s % D];
      const float4 a0 = s_cA[soff + 4 * s], a1c = s_cA[soff + 4 * s + 1];
      const float4 q0 = s_cq[soff + 4 * s], q1 = s_cq[soff + 4 * s + 1];
      const float4 c0 = s_cp[soff + 4 * s], c1 = s_cp[soff + 4 * s + 1];
      float v[8];
      v[0] = fmaf(a0.x, q.glo.x, fmaf(q0.x, q.zlo.x, c0.x)); v[1] = fmaf(a0.y, q.glo.y, fmaf(q0.y, q.zlo.y, c0.y));
      v[2] = fmaf(a0.z, q.glo.z, fmaf(q0.z, q.zlo.z, c0.z)); v[3] = fmaf(a0.w, q.glo.w, fmaf(q0.w, q.zlo.w, c0.w));
      v[4] = fmaf(a1c.x, q.ghi.x, fmaf(q1.x, q.zhi.x, c1.x)); v[5] = fmaf(a1c.y, q.ghi.y, fmaf(q1.y, q.zhi.y, c1.y));
      v[6] = fmaf(a1c.z, q.ghi.z, fmaf(q1.z, q.zhi.z, c1.z)); v[7] = fmaf(a1c.w, q.ghi.w, fmaf(q1.w, q.zhi.w, c1.w));
      u32x4 a1, a2, a3;
      split8(v, a1, a2, a3);
      __builtin_amdgcn_sched_barrier(GANET_SPLIT_MASK);
      ring[s % D] = (s + D < KS) ? load(pgc, pzc, s + D) : load(pgn, pzn, s + D - KS);
      __builtin_amdgcn_sched_barrier(GANET_SPLIT_MASK);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int at = split_unit<RU>(wrow + t * 32, 2 * s + ukg);
        const u32x4 b1 = s_w[at], b2 = s_w[NP * RU + at], b3 = s_w[2 * NP * RU + at];
        GANET_SPLIT_PRODUCTS(acc[t], a1, a2, a3, b1, b2, b3);
      }
    }
    pgc = pgn; pzc = pzn;
    // epilogue (as mlp_bwd_kernel): x softplus'(u_src), column sums of G_src and G_src z_src
    const int64_t row0 = phys(slab) * SLAB;
    const bool full = row0 + SLAB <= M && O == NP;
    constexpr bool PRELOAD = SIG && !ACCUM;
    float sz[PRELOAD ? NT : 1][16];
    if (PRELOAD) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int o = t * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
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
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
          const bool ok = full || (row < M && o < O);
          ex[r] = out[(ok ? row : 0) * ldo + (ok ? o : 0)];
          if (SIG) szt[r] = src_z[(ok ? row : 0) * ld_src + (ok ? o : 0)];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        const bool ok = full || (row < M && o < O);
        float val = acc[t][r];
        if (ACCUM) val += ex[r];
        if (SIG) {
          const float zv = PRELOAD ? sz[PRELOAD ? t : 0][r] : szt[r];
          val *= sigmoid_log2(fmaf(ssc[t], zv, ssh[t]));
          if (ok) { csum[t] += val; csz[t] = fmaf(val, zv, csz[t]); }
        }
        if (ok) out[row * ldo + o] = val;
      }
    }
  }
  if (SIG && col_part) {
    __syncthreads();
    float* s_red = reinterpret_cast<float*>(s_mem);        // [WAVES][256]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float s = csum[t] + __shfl_xor(csum[t], 32);
      const float q = csz[t] + __shfl_xor(csz[t], 32);
      if (kg == 0) { s_red[wave * 256 + t * 32 + col] = s; s_red[wave * 256 + 128 + t * 32 + col] = q; }
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

template <typename Kern>
int set_lds(Kern kern, size_t lds, PerDeviceFlag& done) {
  if (!!done) return 0;
  if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds), "hipFuncSetAttribute")) return 3;
  done = true;
  return 0;
}

}  // namespace

// Returns -1 when the shape has no split kernel (the caller then uses the fp32-MFMA one).
int mlp_fwd_split(int64_t M, int N, int K1, int K2, const float* x1, int64_t ld1, const float* x2, int64_t ld2,
                  const float* in_scale, const float* in_shift, const float* W, const float* bias, float* z,
                  int64_t ldz, float* col_part, const float* stat_shift, int reverse, hipStream_t stream) {
  const int nt = (N + 31) / 32;
  const dim3 grid(BLOCKS), block(WG);
#define LAUNCH(K1_, K2S_, T)                                                                                   \
  do {                                                                                                         \
    constexpr int RU_ = 2 * (((K1_) + 15) / 16 + (K2S_));                                                      \
    const size_t lds = (size_t)3 * (T) * 32 * RU_ * 16 + (size_t)2 * 16 * (K2S_) * 4;                          \
    static PerDeviceFlag attr_set;                                                                                     \
    if (int rc = set_lds(mlp_fwd_split_kernel<K1_, K2S_, T>, lds, attr_set)) return rc;                        \
    ProfScope prof_(K_MLP_FWD, stream);                                                                        \
    hipLaunchKernelGGL((mlp_fwd_split_kernel<K1_, K2S_, T>), grid, block, lds, stream, M, N, x1, ld1, x2, ld2, \
                       in_scale, in_shift, W, bias, z, ldz, col_part, stat_shift, reverse);                    \
    return check_hip(hipGetLastError(), "mlp_fwd_split_kernel");                                               \
  } while (0)
#ifndef GANET_NO_LAYER_FWD
  if (K1 == 0 && K2 == 128 && N == 128) {     // hidden layer: producer / consumer kernel (ganet_layer_fwd.hip)
    const int rc = layer_fwd_spec(M, x2, ld2, in_scale, in_shift, W, bias, z, ldz, col_part, stat_shift, reverse, stream);
    if (rc >= 0) return rc;
  }
#endif
  if (K1 == 0 && K2 == 128 && nt == 4) LAUNCH(0, 8, 4);
  if (K1 == 72 && K2 == 0 && nt == 4) LAUNCH(72, 0, 4);
  if (K1 == 72 && K2 == 128 && nt == 4) LAUNCH(72, 8, 4);
  if (K1 == 0 && K2 == 128 && nt == 1) LAUNCH(0, 8, 1);
#undef LAUNCH
  return -1;
}

int mlp_bwd_split(int64_t M, int O, const float* g, int64_t ldg, const float* gz, int64_t ldgz,
                  const float* gcoef, const float* W, int64_t ldw, float* out, int64_t ldo, bool accumulate,
                  const float* src_z, int64_t ld_src, const float* src_scale, const float* src_shift,
                  float* col_part, int reverse, hipStream_t stream) {
  const bool sig = src_z != nullptr;
  const int nt = O > 96 ? 4 : 3;
  if (O <= 64) return -1;
  const dim3 grid(BLOCKS), block(WG);
#define LAUNCH(T, AC, SG)                                                                                      \
  do {                                                                                                         \
    const size_t lds = (size_t)3 * (T) * 32 * 16 * 16 + 96 * 16;                                               \
    static PerDeviceFlag attr_set;                                                                                     \
    if (int rc = set_lds(mlp_bwd_split_kernel<T, AC, SG>, lds, attr_set)) return rc;                           \
    ProfScope prof_(K_BWD_DATA, stream);                                                                       \
    hipLaunchKernelGGL((mlp_bwd_split_kernel<T, AC, SG>), grid, block, lds, stream, M, O, g, ldg, gz, ldgz,    \
                       gcoef, W, ldw, out, ldo, src_z, ld_src, src_scale, src_shift, col_part, reverse);       \
    return check_hip(hipGetLastError(), "mlp_bwd_split_kernel");                                               \
  } while (0)
  if (nt == 4 && !accumulate && sig) LAUNCH(4, false, true);
  if (nt == 4 && !accumulate && !sig) LAUNCH(4, false, false);
  if (nt == 4 && accumulate && !sig) LAUNCH(4, true, false);
  if (nt == 4 && accumulate && sig) LAUNCH(4, true, true);
  if (nt == 3 && !accumulate && !sig) LAUNCH(3, false, false);
  if (nt == 3 && accumulate && !sig) LAUNCH(3, true, false);
#undef LAUNCH
  return -1;
}

}  // namespace ganet
