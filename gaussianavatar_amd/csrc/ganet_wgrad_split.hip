// ganet_wgrad_split.hip — weight gradient of a decoder layer on the bf16 matrix pipe with exactly split fp32
// operands (ganet_split.h); same interface, operand prologues and partial-tile workspace as wgrad_act_kernel
// (ganet_mlp.hip: the fp32-MFMA weight gradient, kept for the shapes that have no split kernel):
//
//   dW[n,k] = sum_m dz[m,n] . a[m,k],   db[n] = sum_m dz[m,n]        n < 128, k < 32 KT, reduction over M rows
//   dz = A G + q Z + p  (GPRO)  or  G;      a = softplus(scale x + shift)  (ACT)  or  x
//
// The reduction index m is the MFMA's k: lane (c, mg) of an operand fragment holds rows m0 + 8 mg .. + 7 of ONE
// column. So instead of every wave loading and converting everything it multiplies, the eight waves of a
// workgroup share the work through LDS, 16 rows at a time:
//   * produce: wave w of a group of four takes dz columns 32 w + c and a columns 32 w + c — per lane 8 rows of
//     each, dword loads that cover two 128-byte row segments per instruction — applies the prologues in fp32,
//     splits the 8 + 8 values into three bf16x8 fragments each and stores them as 16-byte units
//     [plane][mg][column]: every element of the step is loaded, activated and split exactly ONCE per CU;
//   * consume: wave w accumulates the 64 x 64 quadrant (w >> 1, w & 1) of the dW tile: 12 conflict-free
//     ds_read_b128 and 24 MFMAs per step, 64 accumulator registers.
// The two groups of four waves of a workgroup reduce different row blocks (two waves per SIMD). K = 128: while a
// group multiplies step t out of one LDS buffer it converts step t + 1 into the other, one workgroup barrier per
// step; K = 72: the groups alternate produce-only and consume-only phases in opposition (see the main loop). Loads
// are issued two steps ahead. The groups' tiles are added through LDS at the end: 256 partial tiles per launch,
// summed by the deterministic reduction kernel of ganet_mlp.hip.
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"
#include "ganet_split.h"

namespace ganet {

namespace {

constexpr int WGS = 512;          // two groups of four waves
constexpr int MAX_BLOCKS = 256;
constexpr int STEP = 16;          // rows per step
// scheduling barriers around the prefetch loads of the K = 128 loop: VALU, SALU, MFMA, LDS and transcendental
// instructions may cross (the producer's VALU interleaves with the consumer's MFMAs), VMEM may not (the loads stay
// where they are: left alone the scheduler sinks them to their use). 91 -> 89 us against full barriers (mask 0).
#ifndef GANET_WSPLIT_MASK
#define GANET_WSPLIT_MASK 0x78E
#endif
#ifndef GANET_WSPLIT_PF
#define GANET_WSPLIT_PF 2
#endif
constexpr int PF = GANET_WSPLIT_PF;   // steps whose loads are in flight

template <int KT, int LDX, bool ACT, bool GPRO, bool EXACT>
__global__ void __attribute__((amdgpu_flat_work_group_size(WGS, WGS), amdgpu_waves_per_eu(2, 2)))
wgrad_split_kernel(int64_t M, int K, const float* __restrict__ g, const float* __restrict__ gz,
                   const float* __restrict__ gcoef, const float* __restrict__ x,
                   const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                   float* __restrict__ partial, int64_t steps, int order) {
  constexpr int N = 128, LDG = 128;
  constexpr int KP = 32 * KT;
  constexpr int A_UNITS = 3 * 2 * N, B_UNITS = 3 * 2 * KP;      // per group: [plane][mg][column]
  constexpr int KTW = (KT + 1) / 2;                              // k-tiles of a consumer wave
  extern __shared__ u32x4 s_mem[];   // 2 groups x 2 step buffers (A_UNITS + B_UNITS); reused for the final tile + bias
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, w = wave & 3;
  const int lane = threadIdx.x & 63;
  const int mg = lane >> 5, c = lane & 31;
  // two step buffers per group: [grp][buf] (A_UNITS + B_UNITS); produce writes (pa, pb), consume reads (ca, cb)
  u32x4* const s_grp = s_mem + grp * 2 * (A_UNITS + B_UNITS);
  u32x4 *pa = s_grp, *pb = s_grp + A_UNITS;
  const u32x4 *ca = s_grp, *cb = s_grp + A_UNITS;

  // producer columns and their prologue coefficients
  const int ncol = 32 * w + c;
  const bool bprod = w < KT;                                     // wave-uniform
  const int kcol = min(32 * w + c, K - 1);
  const bool kok = 32 * w + c < K;
  const float cA = GPRO ? gcoef[ncol] : 1.f, cq = GPRO ? gcoef[N + ncol] : 0.f, cp = GPRO ? gcoef[2 * N + ncol] : 0.f;
  // log2-unit softplus: the ln 2 is applied once when the tile is written out
  const float sc = (ACT && bprod) ? in_scale[kcol] * kLog2e : 1.f, sh = (ACT && bprod) ? in_shift[kcol] * kLog2e : 0.f;

  // which steps this group reduces: a contiguous run (order 0) or — exact division only — a common front of
  // 16-row chunks swept first-to-last (1) / last-to-first (2), see include/ganet.h
  const int64_t gi = (int64_t)blockIdx.x * 2 + grp, tg = (int64_t)gridDim.x * 2;
  const bool front = EXACT && order != 0;
  auto row_of = [&](int64_t t) -> int64_t {
    const int64_t tt = min(t, steps - 1);                        // prefetch past the end re-reads the last step
    return (front ? ((order == 2 ? steps - 1 - tt : tt) * tg + gi) : (gi * steps + tt)) * STEP;
  };

  struct Raw { float g[8], z[GPRO ? 8 : 1], x[8]; };
  auto load = [&](Raw& r, int64_t m0) {
    if (EXACT) {
      const float* gp = g + (m0 + 8 * mg) * LDG + ncol;
      const float* zp = GPRO ? gz + (m0 + 8 * mg) * LDG + ncol : nullptr;
      const float* xp = x + (m0 + 8 * mg) * LDX + kcol;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        r.g[i] = gp[i * LDG];
        if (GPRO) r.z[i] = zp[i * LDG];
        r.x[i] = xp[i * LDX];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t row = min(m0 + 8 * mg + i, M - 1);
        r.g[i] = g[row * LDG + ncol];
        if (GPRO) r.z[i] = gz[row * LDG + ncol];
        r.x[i] = x[row * LDX + kcol];
      }
    }
  };

  float bias = 0.f;
  auto produce = [&](const Raw& r, int64_t m0, bool live) {      // live = false: a step nobody consumes (no bias)
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float d = GPRO ? fmaf(cA, r.g[i], fmaf(cq, r.z[i], cp)) : r.g[i];
      if (!EXACT) d = (m0 + 8 * mg + i < M) ? d : 0.f;           // rows beyond M contribute nothing
      v[i] = d;
      bias += live ? d : 0.f;
    }
    u32x4 p1, p2, p3;
    split8(v, p1, p2, p3);
    pa[(0 * 2 + mg) * N + ncol] = p1;
    pa[(1 * 2 + mg) * N + ncol] = p2;
    pa[(2 * 2 + mg) * N + ncol] = p3;
    if (bprod) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float a = ACT ? softplus_log2(fmaf(sc, r.x[i], sh)) : r.x[i];
        v[i] = kok ? a : 0.f;
      }
      split8(v, p1, p2, p3);
      pb[(0 * 2 + mg) * KP + 32 * w + c] = p1;
      pb[(1 * 2 + mg) * KP + 32 * w + c] = p2;
      pb[(2 * 2 + mg) * KP + 32 * w + c] = p3;
    }
  };

  // consumer quadrant
  const int jn = w >> 1, ik = w & 1;
  f32x16 acc[2][KTW];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < KTW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto consume = [&]() {
    u32x4 fa[2][3], fb[KTW][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[a][p] = ca[(p * 2 + mg) * N + (2 * jn + a) * 32 + c];
#pragma unroll
    for (int b = 0; b < KTW; ++b)
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[b][p] = cb[(p * 2 + mg) * KP + min(KTW * ik + b, KT - 1) * 32 + c];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < KTW; ++b)
        if (KTW * ik + b < KT) GANET_SPLIT_PRODUCTS(acc[a][b], fa[a][0], fa[a][1], fa[a][2], fb[b][0], fb[b][1], fb[b][2]);
  };

  Raw ring[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) load(ring[u], row_of(u));
  static_assert(PF == 2, "the ring indexing below is written for two register sets");
  if constexpr (KT == 4) {
    // One barrier per step: while the workgroup multiplies step t out of one buffer (consume: LDS reads + MFMAs) it
    // converts step t + 1 into the other (produce: VALU + LDS writes) — the two halves of a wave's instruction
    // stream are independent, so the compiler and the SIMD's two waves interleave them (128 x 128: 95 -> 90 us
    // against the two-phase schedule below).
    auto point = [&](int buf, bool producer) {
      u32x4* base = s_grp + buf * (A_UNITS + B_UNITS);
      if (producer) { pa = base; pb = base + A_UNITS; } else { ca = base; cb = base + A_UNITS; }
    };
    point(0, true);
    produce(ring[0], row_of(0), true);
    __builtin_amdgcn_sched_barrier(0);
    load(ring[0], row_of(PF));
    __syncthreads();
    for (int64_t t = 0; t < steps; t += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        // (no control flow around the loads — `steps` is a multiple of PF: with branches the compiler's s_waitcnt
        // bookkeeping falls back to vmcnt(0) and the prefetch is lost; the step after the last one is produced
        // from re-read rows and never consumed)
        point((u + 1) & 1, true);
        produce(ring[(u + 1) % PF], row_of(t + u + 1), t + u + 1 < steps);
        __builtin_amdgcn_sched_barrier(GANET_WSPLIT_MASK);
        load(ring[(u + 1) % PF], row_of(t + u + 1 + PF));
        __builtin_amdgcn_sched_barrier(GANET_WSPLIT_MASK);
        point(u & 1, false);
        consume();
        __syncthreads();
      }
    }
  } else {
    // K = 72 (three k-tiles: uneven producer / consumer work) is faster on the two-phase schedule (78 vs 94 us):
    // group 0: produce | consume | produce | ...      group 1: (wait) | produce | consume | ...   one buffer per
    // group, a barrier after every phase, so that each SIMD always holds one producing and one consuming wave
    if (grp == 1) __syncthreads();
    for (int64_t t = 0; t < steps; t += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        produce(ring[u], row_of(t + u), true);
        __builtin_amdgcn_sched_barrier(0);
        load(ring[u], row_of(t + u + PF));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        consume();
        __syncthreads();
      }
    }
    if (grp == 0) __syncthreads();
  }

  // combine the two groups' tiles through LDS (all step buffers are dead), write the partial tile
  float* tile = reinterpret_cast<float*>(s_mem);             // [N][KP] + [N] bias
  bias += __shfl_xor(bias, 32);
  auto tile_index = [&](int a, int b, int r) {
    return ((2 * jn + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * mg) * KP + (KTW * ik + b) * 32 + c;
  };
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < KTW; ++b)
        if (KTW * ik + b < KT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) tile[tile_index(a, b, r)] = acc[a][b][r];
        }
    if (mg == 0) tile[N * KP + ncol] = bias;
  }
  __syncthreads();
  if (grp == 1) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < KTW; ++b)
        if (KTW * ik + b < KT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) tile[tile_index(a, b, r)] += acc[a][b][r];
        }
    if (mg == 0) tile[N * KP + ncol] += bias;
  }
  __syncthreads();
  const float out_scale = ACT ? kLn2 : 1.f;
  float* out = partial + (size_t)blockIdx.x * ((size_t)N * K + N);
  for (int e = threadIdx.x; e < N * K; e += WGS) {
    const int n = e / K, k = e - n * K;
    out[e] = tile[n * KP + k] * out_scale;
  }
  for (int n = threadIdx.x; n < N; n += WGS) out[(size_t)N * K + n] = tile[N * KP + n];
}

}  // namespace

// `blocks` = the partial-tile count the caller's workspace and reduction are planned for (plan_wgrad in
// ganet_mlp.hip). Returns -1 when the shape has no split kernel.
int wgrad_split(int64_t M, int N, int K, const float* g, int64_t ldg, const float* gz, int64_t ldgz,
                const float* gcoef, const float* x, int64_t ldx, const float* in_scale, const float* in_shift,
                float* partial, int blocks, int order, hipStream_t stream) {
  const bool act = in_scale != nullptr, gpro = gz != nullptr;
  if (N != 128 || ldg != 128 || (gpro && ldgz != 128) || blocks <= 0) return -1;
  const bool main_shape = K == 128 && ldx == 128 && act;
  const bool input_shape = K == 72 && ldx == 72 && !act;
  if (!main_shape && !input_shape) return -1;
  const int64_t nsteps = (M + STEP - 1) / STEP;
  int64_t steps = (nsteps + 2 * blocks - 1) / (2 * blocks);           // per group
  steps = (steps + PF - 1) / PF * PF;
  const bool exact = (M % STEP) == 0 && steps * 2 * blocks == nsteps;
#define LAUNCH(KT_, LDX_, A, G, E)                                                                             \
  do {                                                                                                         \
    constexpr size_t step_bytes = (size_t)4 * (3 * 2 * 128 + 3 * 2 * 32 * (KT_)) * 16;                         \
    constexpr size_t tile_bytes = ((size_t)128 * 32 * (KT_) + 128) * 4;                                        \
    const size_t lds = step_bytes > tile_bytes ? step_bytes : tile_bytes;                                      \
    static PerDeviceFlag attr_set;                                                                                     \
    if (!attr_set) {                                                                                           \
      if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_kernel<KT_, LDX_, A, G, E>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),                 \
                    "hipFuncSetAttribute")) return 3;                                                          \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    ProfScope prof_(K_WGRAD, stream);                                                                          \
    hipLaunchKernelGGL((wgrad_split_kernel<KT_, LDX_, A, G, E>), dim3(blocks), dim3(WGS), lds, stream, M, K,   \
                       g, gz, gcoef, x, in_scale, in_shift, partial, steps, order);                            \
    return check_hip(hipGetLastError(), "wgrad_split_kernel");                                                 \
  } while (0)
  if (K == 128 && gpro && exact) LAUNCH(4, 128, true, true, true);
  if (K == 128 && gpro) LAUNCH(4, 128, true, true, false);
  if (K == 128 && exact) LAUNCH(4, 128, true, false, true);
  if (K == 128) LAUNCH(4, 128, true, false, false);
  if (K == 72 && gpro && exact) LAUNCH(3, 72, false, true, true);
  if (K == 72 && gpro) LAUNCH(3, 72, false, true, false);
  if (K == 72 && exact) LAUNCH(3, 72, false, false, true);
  if (K == 72) LAUNCH(3, 72, false, false, false);
#undef LAUNCH
  return -1;
}

}  // namespace ganet
