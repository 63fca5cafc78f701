// ganet_layer_bwd.hip — one-pass backward of a hidden decoder layer on the bf16 matrix pipe with exactly split
// fp32 operands (ganet_split.h): the data gradient AND the weight gradient from ONE sweep over the activations.
//
//   dz     = A G + q Z + p                                  [M,128]  (BatchNorm backward folded, ganet_mlp_bwd.hip)
//   a      = softplus(scale_src Z_src + shift_src)          [M,128]  (the layer's input activation, never stored)
//   out    (+)= dz . W[128, 0:128]   (x softplus'(u_src) -> G_src, with column sums of G_src and G_src z_src)
//   dW[n,k] = sum_m dz[m,n] a[m,k],   db[n] = sum_m dz[m,n]
//
// The separate kernels (mlp_bwd_split + wgrad_split) each stream G, Z and Z_src: 7 [M,128] tensors per layer
// through HBM where 4 suffice (G, Z, Z_src in, G_src out). They are HBM-bound (0.60-0.64 of 8 TB/s) with the
// matrix pipe at 0.19, so the bytes are what is left to remove.
//
// Why this needs a transposition, and where it comes from. The data gradient multiplies dz from the left: its MFMA
// A fragment is "row m, 8 consecutive n". The weight gradient reduces over m: its fragments (of dz AND of a) are
// "column n, 8 consecutive m". Both are the same elements, split once (the three bf16 pieces are element-wise), but
// packed along different axes. gfx950's LDS has the transposing read for exactly this: ds_read_b64_tr_b16 hands a
// lane four 16-bit elements of one COLUMN of a row-major image. So a workgroup keeps, per 32-row slab, the three
// bf16 planes of dz and of a ROW-major in LDS (ds_write_b128 by the lanes that computed them); the data-gradient
// waves read rows (ds_read_b128), the weight-gradient waves read columns (ds_read_b64_tr_b16), nobody converts twice.
//
// Workgroup = 8 waves on one CU, one 32-row slab per round, LDS double-buffered, ONE barrier per round, the round's
// VALU work and its MFMA work on DIFFERENT waves of a SIMD (the matrix pipe and the vector ALU are separate pipes,
// but one wave issues in program order: a first version in which every wave produced AND multiplied measured 43 % of
// wave time stalled at issue with the matrix pipe busy 34 %):
//   * waves 4-7 = producers (slab r + 1): wave w takes rows 8 (w - 4) .. + 7: per lane 8 consecutive columns of two
//     rows of G, Z, Z_src (twelve 16-byte loads, issued two slabs ahead), dz and a in fp32, exact three-way split,
//     three ds_write_b128 per operand; Z_src itself goes to LDS as fp32 for the epilogue. They also store the
//     finished output tile, which the consumers hand back through LDS;
//   * waves 0-3 = consumers (slab r): wave w = data gradient of output columns 32 w .. 32 w + 31 (48 MFMAs; the W
//     fragments of that column tile — 8 k-steps x 3 planes — live in 96 registers for the whole kernel, so W needs no
//     LDS) AND weight-gradient quadrant (w >> 1, w & 1) (64 x 64 of dW, 64 accumulator registers, 48 MFMAs, 48
//     transposing reads), the LDS reads of a group of MFMAs issued before the previous group's MFMAs; then the
//     epilogue of its output tile (softplus', column sums), written over the slab's Z_src image.
//   Waves w and w + 4 share a SIMD: one consumer and one producer each.
// Measured (M = 262,144, MI355X, tools/lbwd_trace.py = s_memtime stamps per phase): round 6.7 k cycles = consumer
// 3.0 k (data gradient: one dependent accumulator chain) + 1.9 k (weight gradient) + 1.1 k (epilogue), producer 5.9 k;
// 120-135 us per launch against 185-210 us for the two separate kernels, 4.1-4.6 TB/s of algorithmic traffic.
// LDS image of a plane: [32 rows][256 B], 16-byte chunk q of row m stored at chunk q ^ swz(m), swz(m) =
// ((m & 3) << 2) | ((m >> 2) & 3): 16 rows of one chunk column cover the 16 chunk slots (ds_read_b128 of the data
// gradient: conflict-free), and the 4 rows x 4 chunks of a transposing half-wave read do too.
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"
#include "ganet_split.h"

namespace ganet {

namespace {

constexpr int LWG = 512;
constexpr int LBLOCKS = 256;
constexpr int LSLAB = 32;
constexpr int PLANE = LSLAB * 256;          // bytes of one bf16 plane of a slab
constexpr int IMG = 3 * PLANE;              // dz or a: 24 KB
constexpr int ZSRC = LSLAB * 128 * 4;       // fp32 Z_src of the slab: 16 KB
constexpr int BUF = 2 * IMG + ZSRC;         // 64 KB
constexpr int BUF_ACC = BUF + ZSRC;         // accumulating variants: + the slab of `out` to add to (80 KB; 2 x 80 = all of LDS)
constexpr int LTILE = 128 * 128 + 128;      // partial tile + bias, as wgrad_split writes it


#ifdef GANET_LBWD_TRACE
// development: phase time stamps (s_memtime) of block 0's consumer wave 0 and producer wave 4, 16 stamps per round
__device__ unsigned long long g_lbwd_trace[2][64][16];
__device__ unsigned long long g_lbwd_blocks[256][4];     // per workgroup: start, loop start, loop end, end (s_memrealtime, 100 MHz)
#define LBWD_STAMP(ROLE, R, I) do { if (blockIdx.x == 0 && lane == 0 && (R) < 64 && wave == ((ROLE) ? 4 : 0)) \
    g_lbwd_trace[ROLE][R][I] = __builtin_amdgcn_s_memtime(); } while (0)
#define LBWD_BLOCK(I) do { if (lane == 0 && wave == 0) g_lbwd_blocks[blockIdx.x][I] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LBWD_STAMP(ROLE, R, I) do {} while (0)
#define LBWD_BLOCK(I) do {} while (0)
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ uint2 read_tr(const char* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  return __builtin_bit_cast(uint2, v);
}

// KIN = 128: the layer's input is act(bn(src_z)) (a hidden layer); KIN = 72: the RAW decoder input x [M, 72] (conv1, and
// the input half of conv5): `src_z` = x, no activation, O = the input's real column count (<= 72) output columns with
// row stride ldo, no softplus' epilogue; dW comes out as the left 72 columns of the 128 x 128 partial tile.
template <bool ACCUM, bool SIG, int KIN>
__global__ void __attribute__((amdgpu_flat_work_group_size(LWG, LWG), amdgpu_waves_per_eu(2, 2)))
layer_bwd_spec_kernel(int64_t M, const float* __restrict__ g, const float* __restrict__ gz,
                      const float* __restrict__ gcoef, const float* __restrict__ W, int64_t ldw, int O,
                      float* __restrict__ out, const float* __restrict__ src_z, const float* __restrict__ src_scale,
                      const float* __restrict__ src_shift, float* __restrict__ col_part, float* __restrict__ wpartial,
                      int reverse) {
  extern __shared__ u32x4 s_mem[];
  char* const lds = reinterpret_cast<char*>(s_mem);
  constexpr int BUFB = ACCUM ? BUF_ACC : BUF;  // bytes of one slab buffer
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int kg = lane >> 5, c = lane & 31;
  const bool consumer = wave < 4;              // uniform
  constexpr int ldo = KIN == 128 ? 128 : KIN;  // row stride of `out`: [M,128], or [M,72] like the padded input
  LBWD_STAMP(0, 63, 0); LBWD_STAMP(1, 63, 0);
  LBWD_BLOCK(0);
  const int64_t nslab = M / LSLAB;
  const int rounds = (int)((nslab + gridDim.x - 1) / gridDim.x);
  auto slab_of = [&](int r) -> int64_t { return (int64_t)r * gridDim.x + blockIdx.x; };
  auto phys = [&](int64_t slab) -> int64_t {
    const int64_t sl = slab < nslab ? slab : nslab - 1;        // past the end: re-read the last slab (never used)
    return reverse ? nslab - 1 - sl : sl;
  };
  float* const wout = wpartial + (size_t)blockIdx.x * LTILE;
  float* const s_red = reinterpret_cast<float*>(lds);          // [4 producer waves][128] after the loop

  if (!consumer) {
    // ---- producer: lane (row group h, prow, pq) owns columns 8 pq .. + 7 of rows 8 pw + 4 h + (lane >> 4), h = 0, 1
    const int pw = wave - 4, pq = lane & 15;
    struct Raw { f32x4 g0, g1, z0, z1, s0, s1, o0, o1; };
    auto row_of = [&](int h) { return 8 * pw + 4 * h + (lane >> 4); };
    auto load_raw = [&](Raw& r, int64_t ps, int h) {
      const int64_t off = (ps * LSLAB + row_of(h)) * 128 + 8 * pq;
      r.g0 = *reinterpret_cast<const f32x4*>(g + off);
      r.g1 = *reinterpret_cast<const f32x4*>(g + off + 4);
      r.z0 = *reinterpret_cast<const f32x4*>(gz + off);
      r.z1 = *reinterpret_cast<const f32x4*>(gz + off + 4);
      if (KIN == 128) {
        r.s0 = *reinterpret_cast<const f32x4*>(src_z + off);
        r.s1 = *reinterpret_cast<const f32x4*>(src_z + off + 4);
      } else {      // raw input rows of 72 floats: chunks 0..8 (lanes beyond re-read chunk 8, masked in produce)
        const int64_t xo = (ps * LSLAB + row_of(h)) * KIN + 8 * (pq < KIN / 8 ? pq : KIN / 8 - 1);
        r.s0 = *reinterpret_cast<const f32x4*>(src_z + xo);
        r.s1 = *reinterpret_cast<const f32x4*>(src_z + xo + 4);
      }
      if (ACCUM) {      // the rows of `out` this slab adds to: handed to the consumers through LDS
        const int64_t oo = (ps * LSLAB + row_of(h)) * ldo + 8 * (KIN == 128 || pq < KIN / 8 ? pq : KIN / 8 - 1);
        r.o0 = *reinterpret_cast<const f32x4*>(out + oo);
        r.o1 = *reinterpret_cast<const f32x4*>(out + oo + 4);
      }
    };
    // The kernel's start is a chain of cold misses (2-4 k cycles each): everything a producer needs first is requested
    // at once — the rows of slab 0, then its 8 columns' coefficients straight from global memory (they used to be staged
    // through LDS behind a workgroup barrier: 21.8 k cycles of prologue, tools/lbwd_trace.py), then the rows of slab 1.
    Raw a0, a1, b0, b1;
    load_raw(a0, phys(slab_of(0)), 0); load_raw(a1, phys(slab_of(0)), 1);
    const float4* cf = reinterpret_cast<const float4*>(gcoef) + 2 * pq;
    const float4 A0 = cf[0], A1 = cf[1], Q0 = cf[32], Q1 = cf[33], P0 = cf[64], P1 = cf[65];
    float4 C0 = make_float4(0.f, 0.f, 0.f, 0.f), C1 = C0, H0 = C0, H1 = C0;
    if (KIN == 128) {
      const float4* sc4 = reinterpret_cast<const float4*>(src_scale) + 2 * pq;
      const float4* sh4 = reinterpret_cast<const float4*>(src_shift) + 2 * pq;
      C0 = sc4[0]; C1 = sc4[1]; H0 = sh4[0]; H1 = sh4[1];
    }
    load_raw(b0, phys(slab_of(1)), 0); load_raw(b1, phys(slab_of(1)), 1);
    if (KIN == 128) {      // softplus in log2 units
      C0.x *= kLog2e; C0.y *= kLog2e; C0.z *= kLog2e; C0.w *= kLog2e; C1.x *= kLog2e; C1.y *= kLog2e; C1.z *= kLog2e; C1.w *= kLog2e;
      H0.x *= kLog2e; H0.y *= kLog2e; H0.z *= kLog2e; H0.w *= kLog2e; H1.x *= kLog2e; H1.y *= kLog2e; H1.z *= kLog2e; H1.w *= kLog2e;
    }
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.f;
    auto produce = [&](const Raw& r, int buf, int h, bool live) {
      char* const base = lds + buf * BUFB;
      const int prow = row_of(h);
      const int pswz = ((prow & 3) << 2) | ((prow >> 2) & 3);
      const int p_img = prow * 256 + ((pq ^ pswz) << 4);
      const int p_zs = prow * 512 + pq * 32;
      const float lv = live ? 1.f : 0.f;      // a slab past the end contributes nothing to dW / db
      float v[8];
      v[0] = fmaf(A0.x, r.g0.x, fmaf(Q0.x, r.z0.x, P0.x)) * lv; v[1] = fmaf(A0.y, r.g0.y, fmaf(Q0.y, r.z0.y, P0.y)) * lv;
      v[2] = fmaf(A0.z, r.g0.z, fmaf(Q0.z, r.z0.z, P0.z)) * lv; v[3] = fmaf(A0.w, r.g0.w, fmaf(Q0.w, r.z0.w, P0.w)) * lv;
      v[4] = fmaf(A1.x, r.g1.x, fmaf(Q1.x, r.z1.x, P1.x)) * lv; v[5] = fmaf(A1.y, r.g1.y, fmaf(Q1.y, r.z1.y, P1.y)) * lv;
      v[6] = fmaf(A1.z, r.g1.z, fmaf(Q1.z, r.z1.z, P1.z)) * lv; v[7] = fmaf(A1.w, r.g1.w, fmaf(Q1.w, r.z1.w, P1.w)) * lv;
#pragma unroll
      for (int e = 0; e < 8; ++e) bias[e] += v[e];
      u32x4 p1, p2, p3;
      split8(v, p1, p2, p3);
      *reinterpret_cast<u32x4*>(base + p_img) = p1;
      *reinterpret_cast<u32x4*>(base + PLANE + p_img) = p2;
      *reinterpret_cast<u32x4*>(base + 2 * PLANE + p_img) = p3;
      if (KIN == 128) {
        v[0] = softplus_log2(fmaf(C0.x, r.s0.x, H0.x)); v[1] = softplus_log2(fmaf(C0.y, r.s0.y, H0.y));
        v[2] = softplus_log2(fmaf(C0.z, r.s0.z, H0.z)); v[3] = softplus_log2(fmaf(C0.w, r.s0.w, H0.w));
        v[4] = softplus_log2(fmaf(C1.x, r.s1.x, H1.x)); v[5] = softplus_log2(fmaf(C1.y, r.s1.y, H1.y));
        v[6] = softplus_log2(fmaf(C1.z, r.s1.z, H1.z)); v[7] = softplus_log2(fmaf(C1.w, r.s1.w, H1.w));
      } else {      // raw input; columns beyond KIN are zero (their dW columns are never read)
        const float on = pq < KIN / 8 ? 1.f : 0.f;
        v[0] = r.s0.x * on; v[1] = r.s0.y * on; v[2] = r.s0.z * on; v[3] = r.s0.w * on;
        v[4] = r.s1.x * on; v[5] = r.s1.y * on; v[6] = r.s1.z * on; v[7] = r.s1.w * on;
      }
      split8(v, p1, p2, p3);
      *reinterpret_cast<u32x4*>(base + IMG + p_img) = p1;
      *reinterpret_cast<u32x4*>(base + IMG + PLANE + p_img) = p2;
      *reinterpret_cast<u32x4*>(base + IMG + 2 * PLANE + p_img) = p3;
      if (SIG) {
        *reinterpret_cast<f32x4*>(base + 2 * IMG + p_zs) = r.s0;
        *reinterpret_cast<f32x4*>(base + 2 * IMG + p_zs + 16) = r.s1;
      }
      if (ACCUM) {
        *reinterpret_cast<f32x4*>(base + BUF + p_zs) = r.o0;
        *reinterpret_cast<f32x4*>(base + BUF + p_zs + 16) = r.o1;
      }
    };
    // two slabs of raw rows in flight: sets (a0, a1) and (b0, b1); the round loop is unrolled by two so that a set is
    // a fixed group of registers (a loop-carried copy would wait for the loads)
    produce(a0, 0, 0, slab_of(0) < nslab); produce(a1, 0, 1, slab_of(0) < nslab);
    load_raw(a0, phys(slab_of(2)), 0); load_raw(a1, phys(slab_of(2)), 1);
    __syncthreads();                                     // slab 0 in LDS
    const int rounds2 = (rounds + 1) & ~1;
    // The finished output tile comes back from the consumers through LDS (they wrote it over the slab's Z_src image,
    // each value where its z was): the producer lane that owns those 32 bytes stores them with two 16-byte stores
    // before it overwrites them with the next slab's Z_src. The consumers issue no global stores: a consumer blocked
    // at a full store queue stalls the matrix pipe (measured: 3-4 k cycles per round), a producer has slack.
    auto drain = [&](int r, int h) {                     // output tile of round r (buffer r & 1), row group h
      if (r < 0 || !(slab_of(r) < nslab && r < rounds)) return;
      const char* const zs = lds + (r & 1) * BUFB + 2 * IMG + row_of(h) * 512 + pq * 32;
      const float4 o0 = *reinterpret_cast<const float4*>(zs);
      const float4 o1 = *reinterpret_cast<const float4*>(zs + 16);
      if (KIN != 128 && pq >= KIN / 8) return;
      float* const op = out + (phys(slab_of(r)) * LSLAB + row_of(h)) * ldo + 8 * pq;
      *reinterpret_cast<float4*>(op) = o0;
      *reinterpret_cast<float4*>(op + 4) = o1;
    };
    // the raw values are consumed from this point on — and not earlier: without the pin the compiler hoists the
    // prologue arithmetic of all four produce calls of the unrolled body to its top and waits for every load there
    auto pin = [&](Raw& s) {
      asm volatile("" : "+v"(s.g0), "+v"(s.g1), "+v"(s.z0), "+v"(s.z1), "+v"(s.s0), "+v"(s.s1) :: "memory");
      if (ACCUM) asm volatile("" : "+v"(s.o0), "+v"(s.o1) :: "memory");
    };
    auto round = [&](Raw& s0, Raw& s1, int r) {          // slab r + 1 -> buffer (r + 1) & 1, then slab r + 3's loads
      const bool live = slab_of(r + 1) < nslab && r + 1 < rounds;
      LBWD_STAMP(1, r, 0);
      drain(r - 1, 0);
      pin(s0);
      LBWD_STAMP(1, r, 1);
      produce(s0, (r + 1) & 1, 0, live);
      LBWD_STAMP(1, r, 2);
      __builtin_amdgcn_sched_barrier(0);
      load_raw(s0, phys(slab_of(r + 3)), 0);
      __builtin_amdgcn_sched_barrier(0);
      LBWD_STAMP(1, r, 3);
      drain(r - 1, 1);
      pin(s1);
      produce(s1, (r + 1) & 1, 1, live);
      __builtin_amdgcn_sched_barrier(0);
      LBWD_STAMP(1, r, 4);
      load_raw(s1, phys(slab_of(r + 3)), 1);
      __builtin_amdgcn_sched_barrier(0);
      LBWD_STAMP(1, r, 5);
      __syncthreads();
      LBWD_STAMP(1, r, 6);
    };
    LBWD_STAMP(1, 63, 1);
    for (int r = 0; r < rounds2; r += 2) {
      round(b0, b1, r);
      round(a0, a1, r + 1);
    }
    LBWD_STAMP(1, 63, 2);
    drain(rounds2 - 1, 0);
    drain(rounds2 - 1, 1);
    // db: every lane holds 8 column sums over its rows
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float b = bias[e];
      b += __shfl_xor(b, 16);
      b += __shfl_xor(b, 32);
      if (lane < 16) s_red[pw * 128 + 8 * pq + e] = b;
    }
  } else {
    // ---- consumer
    u32x4 Bw[8][3];
    {
      const bool col_on = 32 * wave + c < O;
      const float* wp = W + (size_t)(8 * kg) * ldw + (col_on ? 32 * wave + c : 0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = col_on ? wp[(size_t)(16 * s + e) * ldw] : 0.f;
        split8(v, Bw[s][0], Bw[s][1], Bw[s][2]);
      }
    }
    const float ssc = SIG ? src_scale[32 * wave + c] * kLog2e : 0.f;
    const float ssh = SIG ? src_shift[32 * wave + c] * kLog2e : 0.f;
    const int dswz = ((c & 3) << 2) | ((c >> 2) & 3);
    const int d_row = c * 256;
    float csum = 0.f, csz = 0.f;
    const int jn = wave >> 1, ik = wave & 1;
    const int g16 = lane & 15, cc = (lane >> 4) & 1;
    auto tr_addr = [&](int tile, int hi) -> int {
      const int m = 8 * kg + 4 * hi + (g16 >> 2);
      const int q = 4 * tile + 2 * cc + ((g16 >> 1) & 1);
      const int sw = ((g16 >> 2) << 2) | (2 * kg + hi);
      return m * 256 + ((q ^ sw) << 4) + (g16 & 1) * 8;
    };
    // one base address per image; the other tile of the pair flips chunk bit 2 (^ 64 bytes), the upper four rows of
    // a k-group flip chunk bit 0 and lie 4 rows on (^ 16, + 1024): see tr_addr
    const int ta0 = tr_addr(2 * jn, 0), tb0 = IMG + tr_addr(2 * ik, 0);
    f32x16 wacc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[a][b][r] = 0.f;
    __syncthreads();                                     // slab 0 in LDS
    const int rounds2 = (rounds + 1) & ~1;
    LBWD_STAMP(0, 63, 1);
    LBWD_BLOCK(1);
    for (int r = 0; r < rounds2; ++r) {
      const char* const base = lds + (r & 1) * BUFB;
      const int64_t ps = phys(slab_of(r));
      const bool live = slab_of(r) < nslab && r < rounds;
      LBWD_STAMP(0, r, 0);
      // data gradient of this wave's 32 output columns, then the weight-gradient quadrant: 12 groups of MFMAs, the
      // LDS reads of a group issued BEFORE the MFMAs of the previous one (fences: the compiler otherwise sinks every
      // read to its use and the consumer — alone on its SIMD's matrix pipe — waits out each LDS round trip)
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      auto read_a = [&](int s, u32x4 (&f)[3]) {
        const int at = d_row + (((2 * s + kg) ^ dswz) << 4);
        f[0] = *reinterpret_cast<const u32x4*>(base + at);
        f[1] = *reinterpret_cast<const u32x4*>(base + PLANE + at);
        f[2] = *reinterpret_cast<const u32x4*>(base + 2 * PLANE + at);
      };
      auto read_frag = [&](int adr0, int t, int ms, u32x4 (&f)[3]) {
        const int lo_at = adr0 ^ (t * 64);
        const int hi_at = (lo_at ^ 16) + 1024;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const uint2 lo = read_tr(base + lo_at + ms * 4096 + p * PLANE);
          const uint2 hi = read_tr(base + hi_at + ms * 4096 + p * PLANE);
          f[p] = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
      };
      u32x4 fcur[3], fnxt[3], fb[2][3];
      read_a(0, fcur);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s < 7) read_a(s + 1, fnxt);
        else { read_frag(tb0, 0, 0, fb[0]); read_frag(tb0, 1, 0, fb[1]); read_frag(ta0, 0, 0, fnxt); }
        __builtin_amdgcn_sched_barrier(0);
        GANET_SPLIT_PRODUCTS(acc, fcur[0], fcur[1], fcur[2], Bw[s][0], Bw[s][1], Bw[s][2]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) fcur[p] = fnxt[p];
      }
      LBWD_STAMP(0, r, 1);
      // weight-gradient quadrant: groups (ms, a); fcur = A fragments of the group, fb = the step's two B tiles
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int ms = gq >> 1, a = gq & 1;
        if (gq == 0 || gq == 2) read_frag(ta0, 1, ms, fnxt);
        else if (gq == 1) read_frag(ta0, 0, 1, fnxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 2; ++b)
          GANET_SPLIT_PRODUCTS(wacc[a][b], fcur[0], fcur[1], fcur[2], fb[b][0], fb[b][1], fb[b][2]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) fcur[p] = fnxt[p];
        // the second row step's B tiles replace the first's (no registers for both: this read is waited for)
        if (gq == 1) { read_frag(tb0, 0, 1, fb[0]); read_frag(tb0, 1, 1, fb[1]); }
      }
      LBWD_STAMP(0, r, 2);
      // epilogue of the output tile: the result replaces z in the slab's Z_src image (the producers store it)
      float* zs = reinterpret_cast<float*>(lds + (r & 1) * BUFB + 2 * IMG) + (4 * kg) * 128 + 32 * wave + c;
      const float* olds = reinterpret_cast<const float*>(lds + (r & 1) * BUFB + BUF) + (4 * kg) * 128 + 32 * wave + c;
      if (live) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2);
          float val = acc[q];
          if (ACCUM) val += olds[row * 128];
          if (SIG) {
            const float zv = zs[row * 128];
            val *= sigmoid_log2(fmaf(ssc, zv, ssh));
            csum += val;
            csz = fmaf(val, zv, csz);
          }
          zs[row * 128] = val;
        }
      }
      LBWD_STAMP(0, r, 3);
      __syncthreads();
      LBWD_STAMP(0, r, 4);
    }
    LBWD_STAMP(0, 63, 2);
    LBWD_BLOCK(2);
    if (SIG) {       // column sums of G_src and G_src z_src: one wave owns a column tile
      const float s = csum + __shfl_xor(csum, 32);
      const float q = csz + __shfl_xor(csz, 32);
      if (kg == 0) {
        col_part[(size_t)blockIdx.x * 256 + 32 * wave + c] = s;
        col_part[(size_t)blockIdx.x * 256 + 128 + 32 * wave + c] = q;
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = (2 * jn + a) * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
          wout[n * 128 + (2 * ik + b) * 32 + c] = wacc[a][b][q] * (KIN == 128 ? kLn2 : 1.f);   // softplus in log2 units
        }
  }
  __syncthreads();
  for (int n = threadIdx.x; n < 128; n += LWG) wout[128 * 128 + n] = (s_red[n] + s_red[128 + n]) + (s_red[256 + n] + s_red[384 + n]);
  LBWD_STAMP(0, 63, 3); LBWD_STAMP(1, 63, 3);
  LBWD_BLOCK(3);
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

#ifdef GANET_LBWD_TRACE
int ganet_dev_lbwd_trace(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lbwd_trace), sizeof(g_lbwd_trace)); }
int ganet_dev_lbwd_blocks(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lbwd_blocks), sizeof(g_lbwd_blocks)); }
#endif
int32_t ganet_mlp_bwd_fused_parts(void) { return LBLOCKS; }

size_t ganet_mlp_bwd_fused_workspace(void) { return (size_t)LBLOCKS * LTILE * sizeof(float); }

static int layer_bwd_launch(int64_t M, int kin, const float* g, const float* gz, const float* gcoef, const float* W,
                            int64_t ldw, int O, float* out, int accumulate, const float* src, const float* src_scale,
                            const float* src_shift, int apply_act, float* col_part, void* wgrad_workspace,
                            size_t workspace_bytes, int32_t row_order, hipStream_t stream) {
  if (workspace_bytes < ganet_mlp_bwd_fused_workspace()) {
    set_error("ganet_mlp_bwd_fused: workspace too small (%zu < %zu)", workspace_bytes,
              ganet_mlp_bwd_fused_workspace());
    return 2;
  }
  const int reverse = row_order == GANET_ROWS_DOWN ? 1 : 0;
  const int64_t nslab = M / LSLAB;
  const int blocks = (int)(nslab < LBLOCKS ? nslab : LBLOCKS);
  float* wp = static_cast<float*>(wgrad_workspace);
  if (blocks < LBLOCKS)        // the reduction adds up all LBLOCKS partial tiles
    if (check_hip(hipMemsetAsync(wp + (size_t)blocks * LTILE, 0, (size_t)(LBLOCKS - blocks) * LTILE * sizeof(float),
                                 stream), "hipMemsetAsync")) return 3;
  if (apply_act && blocks < LBLOCKS)
    if (check_hip(hipMemsetAsync(col_part + (size_t)blocks * 256, 0, (size_t)(LBLOCKS - blocks) * 256 * sizeof(float),
                                 stream), "hipMemsetAsync")) return 3;
#define LAUNCH(AC, SG, KI)                                                                                      \
  do {                                                                                                          \
    static PerDeviceFlag attr_set;                                                                                      \
    if (!attr_set) {                                                                                            \
      if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_bwd_spec_kernel<AC, SG, KI>),       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * ((AC) ? BUF_ACC : BUF))), \
                    "hipFuncSetAttribute")) return 3;                                                           \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    ProfScope prof_(K_LAYER_BWD, stream);                                                                       \
    hipLaunchKernelGGL((layer_bwd_spec_kernel<AC, SG, KI>), dim3(blocks), dim3(LWG), 2 * ((AC) ? BUF_ACC : BUF), stream, M, g, gz, \
                       gcoef, W, ldw, O, out, src, src_scale, src_shift, col_part, wp, reverse);                \
    return check_hip(hipGetLastError(), "layer_bwd_kernel");                                                    \
  } while (0)
  if (kin == 72) {
    if (accumulate) LAUNCH(true, false, 72);
    LAUNCH(false, false, 72);
  }
  if (accumulate && apply_act) LAUNCH(true, true, 128);
  if (accumulate) LAUNCH(true, false, 128);
  if (apply_act) LAUNCH(false, true, 128);
  LAUNCH(false, false, 128);
#undef LAUNCH
}

int ganet_mlp_bwd_fused(int64_t M, const float* g, const float* gz, const float* gcoef, const float* W, int64_t ldw,
                        float* out, int32_t accumulate, const float* src_z, const float* src_scale,
                        const float* src_shift, int32_t apply_act, float* col_part, void* wgrad_workspace,
                        size_t workspace_bytes, int32_t row_order, void* stream_) {
  if (M <= 0 || (M % LSLAB) || !g || !gz || !gcoef || !W || ldw < 128 || !out || !src_z || !src_scale ||
      !src_shift || (apply_act && !col_part) || !wgrad_workspace || !aligned16(g) || !aligned16(gz) ||
      !aligned16(src_z) || !aligned16(gcoef) || !aligned16(src_scale) || !aligned16(src_shift)) {
    set_error("ganet_mlp_bwd_fused: invalid arguments (M must be a multiple of %d, activations [M,128] contiguous "
              "and 16-byte aligned like the coefficient vectors, W [128, >=128] with row stride ldw)", LSLAB);
    return 1;
  }
  return layer_bwd_launch(M, 128, g, gz, gcoef, W, ldw, 128, out, accumulate, src_z, src_scale, src_shift, apply_act,
                          col_part, wgrad_workspace, workspace_bytes, row_order, static_cast<hipStream_t>(stream_));
}

int ganet_mlp_bwd_fused_input(int64_t M, const float* g, const float* gz, const float* gcoef, const float* W,
                              int64_t ldw, int32_t O, float* out, int64_t ldo, int32_t accumulate, const float* x,
                              void* wgrad_workspace, size_t workspace_bytes, int32_t row_order, void* stream_) {
  if (M <= 0 || (M % LSLAB) || !g || !gz || !gcoef || !W || O <= 0 || O > 72 || ldw < O || !out || ldo != 72 ||
      !x || !wgrad_workspace || !aligned16(g) || !aligned16(gz) || !aligned16(x) || !aligned16(out) ||
      !aligned16(gcoef)) {
    set_error("ganet_mlp_bwd_fused_input: invalid arguments (M a multiple of %d; g, gz [M,128]; x and out [M,72] contiguous, "
              "all 16-byte aligned; W [128, >= O], O <= 72)", LSLAB);
    return 1;
  }
  return layer_bwd_launch(M, 72, g, gz, gcoef, W, ldw, O, out, accumulate, x, nullptr, nullptr, 0, nullptr,
                          wgrad_workspace, workspace_bytes, row_order, static_cast<hipStream_t>(stream_));
}

}  // extern "C"
