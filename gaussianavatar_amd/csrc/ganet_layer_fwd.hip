// ganet_layer_fwd.hip — forward of a hidden decoder layer (128 -> 128, the nine of the fourteen forward launches
// that carry most of the time) with the round's work split between producer and consumer waves, as in the
// one-pass backward (ganet_layer_bwd.hip, whose LDS image, swizzle and barrier scheme this file reuses):
//
//   Z[M,128] = softplus(scale . X + shift)[M,128] . W^T + b,   column sums of (Z - s) and (Z - s)^2 in the epilogue
//
// mlp_fwd_split_kernel (ganet_mlp_split.hip) lets every wave load its rows, activate, split and multiply: the
// VALU work (softplus + exact three-way split, ~100 instructions per k-step) and the MFMAs of a wave issue in program
// order, and its row-per-lane loads and stores touch 32 cache lines per instruction. Here, per 32-row slab:
//   * waves 4-11 = producers (slab r + 1), two per SIMD: coalesced 16-byte loads of X four slabs ahead, activation in
//     fp32, exact split, three ds_write_b128 into the slab's row-major bf16 image; they also store the finished output
//     tile of slab r - 1, which the consumers hand back through LDS (a consumer never waits for a store queue). A
//     lone wave issues an instruction every ~6 cycles, and the ~130 VALU instructions per 8 elements are the round's
//     longest chain: with four producer waves the round took 3.05 k cycles, the consumers idle for a quarter of it;
//   * waves 0-3 = consumers (slab r): wave w = output columns 32 w .. 32 w + 31: its W fragments (8 k-steps x 3
//     planes) live in 96 registers for the whole kernel, 48 MFMAs per slab, then + bias, the column statistics, and
//     the tile into LDS.
// One barrier per round, LDS double-buffered: 2 x (24 KB image + 16 KB output tile).
// Measured (M = 262,144, tools/lfwd_trace.py): round 3.2 k cycles = 1.5 us for 32 KB per workgroup — the loop runs at
// 5.25 TB/s of HBM traffic, its bound; prologue 3.2 us (W fragments, first slab), workgroups finish within 7 us of
// each other. In the training iteration the fourteen forward launches take ~850 us where they took ~866 with
// mlp_fwd_split_kernel for the hidden layers (stand-alone, inputs cold: 70 vs 71 us per launch): the forward layer
// is bound by the memory system, not by its instruction mix.
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"
#include "ganet_split.h"

namespace ganet {

namespace {

constexpr int FWG = 768;                    // 4 consumer + 8 producer waves: three waves per SIMD
constexpr int FBLOCKS = 256;                // = FWD_BLOCKS of ganet_mlp.hip: rows of the column-sum partials
constexpr int FSLAB = 32;
constexpr int PLANE = FSLAB * 256;          // bytes of one bf16 plane of a slab
constexpr int IMG = 3 * PLANE;              // 24 KB
constexpr int TILE = FSLAB * 128 * 4;       // fp32 output tile: 16 KB
constexpr int BUF = IMG + TILE;             // 40 KB

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef GANET_LFWD_TRACE
// development: phase time stamps (s_memtime) of block 0's consumer wave 0 and producer wave 4 (tools/lfwd_trace.py)
__device__ unsigned long long g_lfwd_trace[2][64][8];
__device__ unsigned long long g_lfwd_blocks[256][4];     // per workgroup: start, loop start, loop end, end (s_memrealtime)
#define LFWD_BLOCK(I) do { if (lane == 0 && wave == 0) g_lfwd_blocks[blockIdx.x][I] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define LFWD_STAMP(ROLE, R, I) do { if (blockIdx.x == 0 && lane == 0 && (R) < 64 && wave == ((ROLE) ? 4 : 0)) \
    g_lfwd_trace[ROLE][R][I] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LFWD_BLOCK(I) do {} while (0)
#define LFWD_STAMP(ROLE, R, I) do {} while (0)
#endif

#ifndef GANET_LFWD_ABLATE
#define GANET_LFWD_ABLATE 0   // development (tools/lfwd_ablate.sh): 1 = no softplus in the producers, 2 = one product of six,
#endif                        // 4 = output tiles not stored, 8 = every branch reads its own copy of the input
#ifndef GANET_LFWD_CHAINS
#define GANET_LFWD_CHAINS 1   // 2: even / odd k-steps on separate accumulators — 16 registers more than three waves per SIMD leave
#endif

// NB = 3: the three conv6 branches of the decoder (same input z5, three weight sets) in ONE launch. A branch is not a
// second loop over the rows — the register file has no room for three sets of W fragments — but its own workgroups:
// 240 = 8 XCDs x 10 groups x 3 branches, the three workgroups of a group on the SAME XCD (block indices 8 apart) walking
// the same slabs round by round, so that the input slab comes out of HBM once and the other two reads hit the XCD's L2.
struct FwdBranches {
  const float* W[3];
  const float* bias[3];
  float* z[3];
  float* col_part[3];
  const float* stat_shift[3];
};

template <int NB>
__global__ void __attribute__((amdgpu_flat_work_group_size(FWG, FWG), amdgpu_waves_per_eu(3, 3)))
layer_fwd_spec_kernel(int64_t M, const float* __restrict__ x, const float* __restrict__ in_scale,
                      const float* __restrict__ in_shift, FwdBranches br, int reverse) {
  extern __shared__ u32x4 s_mem[];
  char* const lds = reinterpret_cast<char*>(s_mem);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int kg = lane >> 5, c = lane & 31;
  const bool consumer = wave < 4;              // uniform
  LFWD_BLOCK(0);

  // branch of this workgroup, its index among the branch's workgroups and their number
  const int branch = NB == 1 ? 0 : (int)(blockIdx.x >> 3) % NB;
  const int vblock = NB == 1 ? (int)blockIdx.x : (int)(blockIdx.x / (8 * NB)) * 8 + (int)(blockIdx.x & 7);
  const int vgrid = NB == 1 ? (int)gridDim.x : (int)gridDim.x / NB;
#if GANET_LFWD_ABLATE & 8
  x += (size_t)branch * M * 128;
#endif
  const float* __restrict__ const W = br.W[branch];
  const float* __restrict__ const bias = br.bias[branch];
  float* __restrict__ const z = br.z[branch];
  float* __restrict__ const col_part = br.col_part[branch];
  const float* __restrict__ const stat_shift = br.stat_shift[branch];

  const int64_t nslab = M / FSLAB;
  const int rounds = (int)((nslab + vgrid - 1) / vgrid);
  const int roundsN = (rounds + 3) & ~3;            // both roles run the same number of barriers
  auto slab_of = [&](int r) -> int64_t { return (int64_t)r * vgrid + vblock; };
  auto phys = [&](int64_t slab) -> int64_t {
    const int64_t sl = slab < nslab ? slab : nslab - 1;        // past the end: re-read the last slab (never used)
    return reverse ? nslab - 1 - sl : sl;
  };

  if (!consumer) {
    // ---- producer: lane (prow, pq) owns columns 8 pq .. + 7 of row 4 pw + (lane >> 4)
    const int pw = wave - 4, pq = lane & 15;
    const int prow = 4 * pw + (lane >> 4);
    struct Raw { f32x4 s0, s1; };
    auto load_raw = [&](Raw& r, int64_t ps) {
      const int64_t off = (ps * FSLAB + prow) * 128 + 8 * pq;
      r.s0 = *reinterpret_cast<const f32x4*>(x + off);
      r.s1 = *reinterpret_cast<const f32x4*>(x + off + 4);
    };
    // folded BatchNorm of this lane's 8 columns, pre-multiplied by log2(e)
    float4 C0 = *reinterpret_cast<const float4*>(in_scale + 8 * pq), C1 = *reinterpret_cast<const float4*>(in_scale + 8 * pq + 4);
    float4 H0 = *reinterpret_cast<const float4*>(in_shift + 8 * pq), H1 = *reinterpret_cast<const float4*>(in_shift + 8 * pq + 4);
    C0.x *= kLog2e; C0.y *= kLog2e; C0.z *= kLog2e; C0.w *= kLog2e; C1.x *= kLog2e; C1.y *= kLog2e; C1.z *= kLog2e; C1.w *= kLog2e;
    H0.x *= kLog2e; H0.y *= kLog2e; H0.z *= kLog2e; H0.w *= kLog2e; H1.x *= kLog2e; H1.y *= kLog2e; H1.z *= kLog2e; H1.w *= kLog2e;
    const int pswz = ((prow & 3) << 2) | ((prow >> 2) & 3);
    const int p_img = prow * 256 + ((pq ^ pswz) << 4);
    auto produce = [&](const Raw& r, int buf) {
      char* const base = lds + buf * BUF;
      float v[8];
#if GANET_LFWD_ABLATE & 1
      v[0] = r.s0.x; v[1] = r.s0.y; v[2] = r.s0.z; v[3] = r.s0.w; v[4] = r.s1.x; v[5] = r.s1.y; v[6] = r.s1.z; v[7] = r.s1.w;
#else
      v[0] = softplus_log2(fmaf(C0.x, r.s0.x, H0.x)); v[1] = softplus_log2(fmaf(C0.y, r.s0.y, H0.y));
      v[2] = softplus_log2(fmaf(C0.z, r.s0.z, H0.z)); v[3] = softplus_log2(fmaf(C0.w, r.s0.w, H0.w));
      v[4] = softplus_log2(fmaf(C1.x, r.s1.x, H1.x)); v[5] = softplus_log2(fmaf(C1.y, r.s1.y, H1.y));
      v[6] = softplus_log2(fmaf(C1.z, r.s1.z, H1.z)); v[7] = softplus_log2(fmaf(C1.w, r.s1.w, H1.w));
#endif
      u32x4 p1, p2, p3;
      split8(v, p1, p2, p3);
      *reinterpret_cast<u32x4*>(base + p_img) = p1;
      *reinterpret_cast<u32x4*>(base + PLANE + p_img) = p2;
      *reinterpret_cast<u32x4*>(base + 2 * PLANE + p_img) = p3;
    };
    // the output tile of round r (buffer r & 1): the lane that owns those 32 bytes stores them
    auto drain = [&](int r) {
      if (r < 0 || !(slab_of(r) < nslab && r < rounds)) return;
#if GANET_LFWD_ABLATE & 4
      if (M > 0) return;
#endif
      const char* const zs = lds + (r & 1) * BUF + IMG + prow * 512 + pq * 32;
      const float4 o0 = *reinterpret_cast<const float4*>(zs);
      const float4 o1 = *reinterpret_cast<const float4*>(zs + 16);
      float* const op = z + (phys(slab_of(r)) * FSLAB + prow) * 128 + 8 * pq;
      *reinterpret_cast<float4*>(op) = o0;
      *reinterpret_cast<float4*>(op + 4) = o1;
    };
    // four slabs of raw rows in flight (32 KB per slab and workgroup); the round loop is unrolled by four so that a
    // set is a fixed group of registers (a loop-carried copy would wait for the loads)
    Raw q[4];
    load_raw(q[0], phys(slab_of(0)));
    produce(q[0], 0);
#pragma unroll
    for (int i = 1; i < 5; ++i) load_raw(q[i & 3], phys(slab_of(i)));
    __syncthreads();                                     // slab 0 in LDS
    // the raw values are consumed from this point on — and not earlier (ganet_layer_bwd.hip: without the pin the
    // compiler hoists the arithmetic of the unrolled body to its top and waits for every load there)
    auto pin = [&](Raw& s) { asm volatile("" : "+v"(s.s0), "+v"(s.s1) :: "memory"); };
    auto round = [&](Raw& s0, int r) {                   // slab r + 1 -> buffer (r + 1) & 1, then slab r + 5's loads
      LFWD_STAMP(1, r, 0);
      drain(r - 1);
      LFWD_STAMP(1, r, 1);
      pin(s0);
      LFWD_STAMP(1, r, 2);
      produce(s0, (r + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      load_raw(s0, phys(slab_of(r + 5)));
      __builtin_amdgcn_sched_barrier(0);
      LFWD_STAMP(1, r, 3);
      __syncthreads();
      LFWD_STAMP(1, r, 4);
    };
    for (int r = 0; r < roundsN; r += 4) {
      round(q[1], r);
      round(q[2], r + 1);
      round(q[3], r + 2);
      round(q[0], r + 3);
    }
    drain(roundsN - 1);
  } else {
    // ---- consumer: B fragment of k-step s = W[32 wave + c][16 s + 8 kg .. + 7] (x ln 2: softplus in log2 units)
    u32x4 Bw[8][3];
    {
      const float* wp = W + (size_t)(32 * wave + c) * 128 + 8 * kg;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float4 lo = *reinterpret_cast<const float4*>(wp + 16 * s);
        const float4 hi = *reinterpret_cast<const float4*>(wp + 16 * s + 4);
        const float v[8] = {lo.x * kLn2, lo.y * kLn2, lo.z * kLn2, lo.w * kLn2, hi.x * kLn2, hi.y * kLn2, hi.z * kLn2, hi.w * kLn2};
        split8(v, Bw[s][0], Bw[s][1], Bw[s][2]);
      }
    }
    const float bias_r = bias ? bias[32 * wave + c] : 0.f;
    const float sshift = stat_shift ? stat_shift[32 * wave + c] : 0.f;
    const int dswz = ((c & 3) << 2) | ((c >> 2) & 3);
    const int d_row = c * 256;
    float csum = 0.f, csq = 0.f;
    __syncthreads();                                     // slab 0 in LDS
    LFWD_BLOCK(1);
    for (int r = 0; r < roundsN; ++r) {
      const char* const base = lds + (r & 1) * BUF;
      const bool live = slab_of(r) < nslab && r < rounds;
      LFWD_STAMP(0, r, 0);
      f32x16 acc[GANET_LFWD_CHAINS];
#pragma unroll
      for (int h = 0; h < GANET_LFWD_CHAINS; ++h)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[h][q] = 0.f;
      auto read_a = [&](int s, u32x4 (&f)[3]) {
        const int at = d_row + (((2 * s + kg) ^ dswz) << 4);
        f[0] = *reinterpret_cast<const u32x4*>(base + at);
        f[1] = *reinterpret_cast<const u32x4*>(base + PLANE + at);
        f[2] = *reinterpret_cast<const u32x4*>(base + 2 * PLANE + at);
      };
      // the LDS reads of a k-step are issued before the MFMAs of the previous one (fences: the compiler otherwise
      // sinks every read to its use)
      u32x4 fcur[3], fnxt[3];
      read_a(0, fcur);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s < 7) read_a(s + 1, fnxt);
        __builtin_amdgcn_sched_barrier(0);
#if GANET_LFWD_ABLATE & 2
        acc[0] = mfma_bf16(fcur[0], Bw[s][0], acc[0]);
        asm volatile("" :: "v"(fcur[1]), "v"(fcur[2]));
#else
        GANET_SPLIT_PRODUCTS(acc[s % GANET_LFWD_CHAINS], fcur[0], fcur[1], fcur[2], Bw[s][0], Bw[s][1], Bw[s][2]);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) fcur[p] = fnxt[p];
      }
      // epilogue: + bias, the column statistics about the shift, and the tile into LDS (the producers store it).
      // C/D layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
      float* zs = reinterpret_cast<float*>(lds + (r & 1) * BUF + IMG) + (4 * kg) * 128 + 32 * wave + c;
      LFWD_STAMP(0, r, 1);
      if (live) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2);
          float val = acc[0][q];
          if (GANET_LFWD_CHAINS == 2) val += acc[GANET_LFWD_CHAINS - 1][q];
          val += bias_r;
          zs[row * 128] = val;
          const float d = val - sshift;
          csum += d;
          csq = fmaf(d, d, csq);
        }
      }
      LFWD_STAMP(0, r, 2);
      __syncthreads();
      LFWD_STAMP(0, r, 3);
    }
    LFWD_BLOCK(2);
    if (col_part) {      // [block][2][128]: one wave owns a column tile
      const float s = csum + __shfl_xor(csum, 32);
      const float q = csq + __shfl_xor(csq, 32);
      if (kg == 0) {
        col_part[(size_t)vblock * 256 + 32 * wave + c] = s;
        col_part[(size_t)vblock * 256 + 128 + 32 * wave + c] = q;
        // the statistics kernel adds up all FBLOCKS rows: the rows no workgroup of this branch owns are zeroed here
        for (int row = vblock + vgrid; NB > 1 && row < FBLOCKS; row += vgrid) {
          col_part[(size_t)row * 256 + 32 * wave + c] = 0.f;
          col_part[(size_t)row * 256 + 128 + 32 * wave + c] = 0.f;
        }
      }
    }
  }
}

}  // namespace

#if defined(GANET_LFWD_TRACE) || GANET_LFWD_ABLATE
// development: the three-branch launch on its own (tools/lfwd_trace.py, tools/lfwd_ablate.py)
extern "C" int ganet_dev_layer_fwd3(int64_t M, const float* x, const float* in_scale, const float* in_shift, const float* W,
                                    const float* bias, float* z, float* col_part, void* stream);
#endif
#ifdef GANET_LFWD_TRACE
extern "C" int ganet_dev_lfwd_blocks(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lfwd_blocks), sizeof(g_lfwd_blocks)); }
extern "C" int ganet_dev_lfwd_trace(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lfwd_trace), sizeof(g_lfwd_trace)); }
#endif

// Hidden layer forward, [M,128] -> [M,128], contiguous rows, M a multiple of 32: returns -1 for anything else (the
// caller then takes mlp_fwd_split_kernel).
template <int NB>
static int launch_layer_fwd(int blocks, int64_t M, const float* x, const float* in_scale, const float* in_shift,
                            const FwdBranches& br, int reverse, hipStream_t stream) {
  const size_t lds = (size_t)2 * BUF;
  static PerDeviceFlag attr_set;
  if (!attr_set) {
    if (int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_fwd_spec_kernel<NB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "layer_fwd lds"))
      return rc;
    attr_set = true;
  }
  ProfScope prof_(K_MLP_FWD, stream);
  hipLaunchKernelGGL((layer_fwd_spec_kernel<NB>), dim3(blocks), dim3(FWG), lds, stream, M, x, in_scale, in_shift, br,
                     reverse);
  return check_hip(hipGetLastError(), "layer_fwd_spec_kernel");
}

int layer_fwd_spec(int64_t M, const float* x, int64_t ldx, const float* in_scale, const float* in_shift, const float* W,
                   const float* bias, float* z, int64_t ldz, float* col_part, const float* stat_shift, int reverse,
                   hipStream_t stream) {
  if (M < FSLAB || (M % FSLAB) != 0 || ldx != 128 || ldz != 128 || !aligned16(z)) return -1;
  FwdBranches br{};
  br.W[0] = W; br.bias[0] = bias; br.z[0] = z; br.col_part[0] = col_part; br.stat_shift[0] = stat_shift;
  return launch_layer_fwd<1>(FBLOCKS, M, x, in_scale, in_shift, br, reverse, stream);
}

// Three layers on the same input (the decoder's conv6 branches) in one launch: -1 when the shape does not qualify.
int layer_fwd_spec3(int64_t M, const float* x, const float* in_scale, const float* in_shift, const float* const* W,
                    const float* const* bias, float* const* z, float* const* col_part, const float* const* stat_shift,
                    int reverse, hipStream_t stream) {
  constexpr int blocks = (FBLOCKS / 24) * 24;          // 240: 8 XCDs x 10 groups x 3 branches
  if (M < (int64_t)FSLAB * (blocks / 3) || (M % FSLAB) != 0) return -1;
  FwdBranches br{};
  for (int j = 0; j < 3; ++j) {
    if (!aligned16(z[j]) || !col_part[j]) return -1;
    br.W[j] = W[j]; br.bias[j] = bias[j]; br.z[j] = z[j]; br.col_part[j] = col_part[j]; br.stat_shift[j] = stat_shift[j];
  }
  return launch_layer_fwd<3>(blocks, M, x, in_scale, in_shift, br, reverse, stream);
}

}  // namespace ganet

#if defined(GANET_LFWD_TRACE) || GANET_LFWD_ABLATE
// W [3][128][128], bias [3][128], z [3][M][128], col_part [3][256][256]
extern "C" int ganet_dev_layer_fwd3(int64_t M, const float* x, const float* in_scale, const float* in_shift, const float* W,
                                    const float* bias, float* z, float* col_part, void* stream) {
  const float* Wp[3]; const float* bp[3]; float* zp[3]; float* cp[3]; const float* sp[3] = {nullptr, nullptr, nullptr};
  for (int j = 0; j < 3; ++j) { Wp[j] = W + j * 128 * 128; bp[j] = bias + j * 128; zp[j] = z + (size_t)j * M * 128; cp[j] = col_part + j * 256 * 256; }
  return ganet::layer_fwd_spec3(M, x, in_scale, in_shift, Wp, bp, zp, cp, sp, 0, static_cast<hipStream_t>(stream));
}
#endif
