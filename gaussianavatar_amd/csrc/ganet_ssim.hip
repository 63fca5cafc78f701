// ganet_ssim.hip — SSIM (11x11 Gaussian window, sigma 1.5, zero padding) forward and backward as
// separable streaming passes.
//
// Reference: /root/reference/utils/loss_utils.py:23-53 — five grouped 11x11 convolutions
// (mu1, mu2, E[x1^2], E[x2^2], E[x1 x2]) plus ~15 element-wise kernels over [B,3,H,W], and
// their autograd counterparts. Here a WAVE owns a vertical strip of 64 columns x STRIP rows of one
// image plane and streams down its rows (no workgroup barriers): a row (+5 columns of halo each side)
// goes through a per-wave LDS line, every lane takes the 11 horizontal taps of its column, and the
// vertical pass is a ring of 11 partially accumulated output rows held in registers (each new input
// row is scattered into the 11 output rows it contributes to; the oldest one is then complete). A
// row's global loads are issued 11 rows before it is consumed (see the schedule note below).
// The L1 loss between the same two images rides along (forward: |x1 - x2| of the strip's own pixels;
// backward: its sign term is added to the SSIM gradient), so the loop's two image losses cost one pass.
// Forward evaluates the SSIM map and — because the loss is always differentiated — the three partial
// derivatives dS/dmu1, dS/dE[x1^2], dS/dE[x1 x2] in the same pass; backward convolves those three maps
// with the (symmetric) window and combines
//     dL/dx1 = scale * ( conv(dS/dmu1) + 2 x1 conv(dS/dE11) + x2 conv(dS/dE12) ).
#include <cmath>

#include "ganet.h"
#include "ganet_common.h"

namespace ganet {

namespace {

constexpr int R = 5, WIN = 11;
#ifndef GANET_SSIM_STRIP
#define GANET_SSIM_STRIP 34
#endif
#ifndef GANET_SSIM_WAVES
#define GANET_SSIM_WAVES 4
#endif
constexpr int STRIP = GANET_SSIM_STRIP;   // output rows per wave; STRIP + 10 input rows are streamed
constexpr int WAVES = GANET_SSIM_WAVES;   // strips stacked in a workgroup
#ifndef GANET_SSIM_AHEAD
#define GANET_SSIM_AHEAD 3
#endif
constexpr int AHEAD = GANET_SSIM_AHEAD;   // rows between a row's global loads and its filtering (1..11)
constexpr int ROUNDS = (STRIP + 2 * R + WIN - 1) / WIN;
constexpr int LINE = 64 + 2 * R + 6;      // LDS line (padded to 80 entries)
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
// The two sums: 3,000 waves adding into one cache line would serialise at the memory-side atomic unit
// (~15 ns each, 45 us — more than the kernel's arithmetic), so every wave adds into one of SUM_SLOTS
// partial sums on cache lines of their own and bumps that slot's arrival counter; the wave that completes
// a slot folds it into sums[0..1] (SUM_SLOTS additions on the result's line in total).
constexpr int SUM_SLOTS = 64, SUM_LINE = 64;             // floats per line (256 B)
constexpr int SUM_FLOATS = SUM_LINE * (1 + SUM_SLOTS);   // what `sums` must hold; zeroed by the host call

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Window { float w[WIN]; };

// The SIMD's arbiter issues oldest-wave-first. A wave alone can issue one instruction every ~6 cycles, so the
// three resident waves of a SIMD finished as a staircase (22 / 30 / 38 us, tools/ssim_trace.py): the oldest ran
// at its solo speed, the others in its gaps, and the last one finished alone. A wave therefore lowers its own
// priority as it advances (3 in the first quarter of its rounds ... 0 in the last): whoever is behind issues
// first, the waves stay within a round of each other and the VALU stays busy until they all end together.
__device__ __forceinline__ void progress_priority(int round) {
  if (round == 0) __builtin_amdgcn_s_setprio(3);
  else if (round == (ROUNDS + 3) / 4) __builtin_amdgcn_s_setprio(2);
  else if (round == (2 * ROUNDS + 3) / 4) __builtin_amdgcn_s_setprio(1);
  else if (round == (3 * ROUNDS + 3) / 4) __builtin_amdgcn_s_setprio(0);
}

#ifdef GANET_SSIM_TRACE   // dev build: per-wave residency (tools/ssim_trace.py)
__device__ unsigned long long g_ssim_trace[8192][4];
#endif

Window make_window() {
  Window k;
  double s = 0.0, v[WIN];
  for (int i = 0; i < WIN; ++i) { v[i] = exp(-(double)((i - R) * (i - R)) / (2.0 * 1.5 * 1.5)); s += v[i]; }
  for (int i = 0; i < WIN; ++i) k.w[i] = (float)(v[i] / s);
  return k;
}

// Buffer loads and stores: an element outside the plane (or a row this wave does not need) is addressed at
// offset ~0u, which the hardware answers with 0 (drops, for a store) without touching memory — the
// convolution's zero padding and every "skip this access" case cost no branch, so the prefetched values
// have no control dependence that would make the compiler wait for them early.
constexpr uint32_t kSkip = 0xffffffffu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const float* p, int H, int W) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, H * W * 4, 0x00020000);
}
// `col` is the lane's byte offset inside a row (kSkip: the column does not exist), `row` the row's byte
// offset (wave-uniform, travels in an SGPR), `rowok` whether this wave wants the row at all.
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, uint32_t col, uint32_t row, bool rowok) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, rowok ? col : kSkip, rowok ? row : 0u, 0));
}
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, uint32_t col, uint32_t row) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, col, row, 0);
}

// Both kernels share one schedule. Input row r = base + j (relative to the strip's first output row)
// lives in ring slot j of three rings that are all indexed statically (the j loop is fully unrolled):
//   * the PREFETCH ring: the row's global loads are issued one whole round (11 rows) before the row is
//     staged through LDS, so a wave has up to 11 rows in flight and never waits a memory round trip per row;
//   * the ACCUMULATOR ring: 11 partially summed output rows (each filtered input row is scattered into
//     the 11 output rows it contributes to; the oldest one is then complete);
//   * (backward) the SIDE ring: the image pixels the finished output row needs, fetched a round ahead too.
// The maps of a row travel through LDS interleaved (one 8- or 16-byte entry per column), so a tap is one
// ds_read and the filters run on packed pairs (v_pk_fma_f32).

__global__ void __launch_bounds__(64 * WAVES)
ssim_fwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                float norm, float* __restrict__ sums, float* __restrict__ partials, size_t map_stride,
                int strips, Window k) {
  __shared__ f32x2 s_line[WAVES][LINE];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * 64, ys = (blockIdx.y * WAVES + wave) * STRIP;
  if (ys >= H) return;
#ifdef GANET_SSIM_TRACE
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
  f32x2* line = s_line[wave];
  const size_t poff = (size_t)plane * H * W;
  const __amdgpu_buffer_rsrc_t a_img = plane_rsrc(img1 + poff, H, W), b_img = plane_rsrc(img2 + poff, H, W);
  const __amdgpu_buffer_rsrc_t o_mu = plane_rsrc(partials + poff, H, W),
                               o_11 = plane_rsrc(partials + map_stride + poff, H, W),
                               o_12 = plane_rsrc(partials + 2 * map_stride + poff, H, W);
  const int gx = x0 + lane;
  const int gx0 = x0 - R + lane;                 // first column this lane stages
  const int gx1 = x0 + 64 - R + lane;            // lanes 0..9: the right halo
  const uint32_t col0 = (gx0 >= 0 && gx0 < W) ? (uint32_t)gx0 * 4u : kSkip;
  const uint32_t col1 = (lane < 2 * R && gx1 < W) ? (uint32_t)gx1 * 4u : kSkip;
  const uint32_t colx = gx < W ? (uint32_t)gx * 4u : kSkip;
  const float inside = gx < W ? 1.f : 0.f;

  f32x2 p0[WIN], p1[WIN];                        // prefetch ring: (x1, x2) at gx0 / gx1
  auto fetch = [&](f32x2& v0, f32x2& v1, int r) {   // input row ys + r
    const int y = ys + r;
    const bool rowok = y >= 0 && y < H && r < STRIP + R;
    const uint32_t row = (uint32_t)(y * W) * 4u;
    v0 = f32x2{buf_load(a_img, col0, row, rowok), buf_load(b_img, col0, row, rowok)};
    v1 = f32x2{buf_load(a_img, col1, row, rowok), buf_load(b_img, col1, row, rowok)};
  };
  f32x2 acc_m[WIN], acc_e[WIN];                  // (mu1, mu2), (E11, E22)
  float acc_x[WIN];                              // E12
#pragma unroll
  for (int j = 0; j < WIN; ++j) {
    acc_m[j] = f32x2{0.f, 0.f};
    acc_e[j] = f32x2{0.f, 0.f};
    acc_x[j] = 0.f;
    if (j < AHEAD) fetch(p0[j], p1[j], j - R);
  }
  float S = 0.f, L = 0.f;
#pragma unroll 1
  for (int round = 0; round < ROUNDS; ++round) {
    const int base = -R + round * WIN;
    progress_priority(round);
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
      const int r = base + j;
      if (r >= STRIP + R) break;                 // uniform; only in the last round
      const int y = ys + r;
      __builtin_amdgcn_wave_barrier();           // the previous row's reads were issued before (in-order LDS)
      line[lane] = p0[j];
      if (lane < 2 * R) line[64 + lane] = p1[j];
      __builtin_amdgcn_wave_barrier();
      fetch(p0[(j + AHEAD) % WIN], p1[(j + AHEAD) % WIN], r + AHEAD);
      f32x2 t[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) t[i] = line[lane + i];
      if (r >= 0 && r < STRIP) L += fabsf(t[R].x - t[R].y);   // L1 rides along (rows/columns outside are 0 - 0)
      f32x2 m = f32x2{0.f, 0.f}, e = f32x2{0.f, 0.f};
      float x = 0.f;
#pragma unroll
      for (int i = 0; i < WIN; ++i) {
        const f32x2 wt = t[i] * k.w[i];
        m += wt;
        e = wt * t[i] + e;
        x = fmaf(wt.x, t[i].y, x);
      }
      // scatter into the output rows r-5 .. r+5 (ring slots (j + d) mod 11), weight w[5 - d]
#pragma unroll
      for (int d = -R; d <= R; ++d) {
        const int slot = (j + d + WIN) % WIN;
        const float wv = k.w[R - d];
        acc_m[slot] = m * wv + acc_m[slot];
        acc_e[slot] = e * wv + acc_e[slot];
        acc_x[slot] = fmaf(wv, x, acc_x[slot]);
      }
      const int done = (j + WIN - R) % WIN;      // output row r-5 is complete
      const int yo = y - R;
      if (r - R >= 0 && r - R < STRIP && yo < H) {          // uniform
        const float m1 = acc_m[done].x, m2 = acc_m[done].y, e11 = acc_e[done].x, e22 = acc_e[done].y;
        const float e12 = acc_x[done];
        const float n1 = 2.f * m1 * m2 + C1;
        const float n2 = 2.f * (e12 - m1 * m2) + C2;
        const float d1 = m1 * m1 + m2 * m2 + C1;
        const float d2 = (e11 - m1 * m1) + (e22 - m2 * m2) + C2;
        const float i1 = __builtin_amdgcn_rcpf(d1), i2 = __builtin_amdgcn_rcpf(d2);
        const float inv = i1 * i2;
        const float Sv = n1 * n2 * inv;
        S = fmaf(inside, Sv, S);
        const uint32_t row = (uint32_t)(yo * W) * 4u;
        buf_store(2.f * m2 * (n2 - n1) * inv - 2.f * m1 * Sv * (i1 - i2), o_mu, colx, row);   // dS/dmu1
        buf_store(-Sv * i2, o_11, colx, row);                                                 // dS/dE[x1^2]
        buf_store(2.f * n1 * inv, o_12, colx, row);                                           // dS/dE[x1 x2]
      }
      acc_m[done] = f32x2{0.f, 0.f};
      acc_e[done] = f32x2{0.f, 0.f};
      acc_x[done] = 0.f;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    S += __shfl_xor(S, off);
    L += __shfl_xor(L, off);
  }
  // the waves that exist (ys < H) are numbered densely: that is what the per-slot arrival counts assume
  const uint32_t wid = ((uint32_t)plane * gridDim.x + blockIdx.x) * strips + (blockIdx.y * WAVES + wave);
  const uint32_t total = gridDim.z * gridDim.x * strips;
  const uint32_t slot = wid % SUM_SLOTS, expected = (total - slot + SUM_SLOTS - 1) / SUM_SLOTS;
  float* sl = sums + SUM_LINE * (1 + slot);
  // Device-scope atomics are performed at the memory side, so "the addition's old value has come back"
  // means it is visible to every later atomic: order the arrival count behind it through a data dependence
  // (no fence — a fence would also write this XCD's L2 back, once per wave).
  float seen = 0.f;
  if (lane < 2) seen = atomicAdd(sl + lane, (lane ? L : S) * norm);
  uint32_t one = 1u;
  asm volatile("" : "+v"(one) : "v"(seen));
  uint32_t prev = 0;
  if (lane == 0) prev = atomicAdd(reinterpret_cast<uint32_t*>(sl + 2), one);
  if (__builtin_amdgcn_readfirstlane(prev) + 1 == expected) {
    if (lane < 2) atomicAdd(sums + lane, atomicAdd(sl + lane, 0.f));
  }
#ifdef GANET_SSIM_TRACE
  if (lane == 0 && wid < 8192) {
    g_ssim_trace[wid][0] = t_start;
    g_ssim_trace[wid][1] = __builtin_amdgcn_s_memrealtime();
    g_ssim_trace[wid][2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
    g_ssim_trace[wid][3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
  }
#endif
}

__global__ void __launch_bounds__(64 * WAVES)
ssim_bwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                const float* __restrict__ partials, size_t map_stride, float norm,
                const float* __restrict__ d_ssim, const float* __restrict__ d_l1,
                float* __restrict__ dimg1, Window k) {
  __shared__ f32x4 s_line[WAVES][LINE];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * 64, ys = (blockIdx.y * WAVES + wave) * STRIP;
  if (ys >= H) return;
  f32x4* line = s_line[wave];
  const size_t poff = (size_t)plane * H * W;
  const __amdgpu_buffer_rsrc_t pa = plane_rsrc(partials + poff, H, W),
                               pb = plane_rsrc(partials + map_stride + poff, H, W),
                               pc = plane_rsrc(partials + 2 * map_stride + poff, H, W),
                               xa = plane_rsrc(img1 + poff, H, W), xb = plane_rsrc(img2 + poff, H, W),
                               od = plane_rsrc(dimg1 + poff, H, W);
  const int gx = x0 + lane;
  const int gx0 = x0 - R + lane, gx1 = x0 + 64 - R + lane;
  const uint32_t col0 = (gx0 >= 0 && gx0 < W) ? (uint32_t)gx0 * 4u : kSkip;
  const uint32_t col1 = (lane < 2 * R && gx1 < W) ? (uint32_t)gx1 * 4u : kSkip;
  const uint32_t colx = gx < W ? (uint32_t)gx * 4u : kSkip;
  const float scale = d_ssim ? norm * d_ssim[0] : 0.f;
  const float l1s = d_l1 ? norm * d_l1[0] : 0.f;

  f32x4 p0[WIN], p1[WIN];                        // prefetch ring: the three derivative maps at gx0 / gx1
  f32x2 side[WIN];                               // side ring: (x1, x2) of the output row the slot completes
  auto fetch = [&](f32x4& v0, f32x4& v1, int r) {   // input row ys + r
    const int y = ys + r;
    const bool rowok = y >= 0 && y < H && r < STRIP + R;
    const uint32_t row = (uint32_t)(y * W) * 4u;
    v0 = f32x4{buf_load(pa, col0, row, rowok), buf_load(pb, col0, row, rowok), buf_load(pc, col0, row, rowok), 0.f};
    v1 = f32x4{buf_load(pa, col1, row, rowok), buf_load(pb, col1, row, rowok), buf_load(pc, col1, row, rowok), 0.f};
  };
  auto fetch_side = [&](f32x2& v, int ro) {      // output row ys + ro
    const int yo = ys + ro;
    const bool rowok = ro >= 0 && ro < STRIP && yo < H;
    const uint32_t row = (uint32_t)(yo * W) * 4u;
    v = f32x2{buf_load(xa, colx, row, rowok), buf_load(xb, colx, row, rowok)};
  };
  f32x2 acc_ab[WIN];
  float acc_c[WIN];
#pragma unroll
  for (int j = 0; j < WIN; ++j) {
    acc_ab[j] = f32x2{0.f, 0.f};
    acc_c[j] = 0.f;
    if (j < AHEAD) {
      fetch(p0[j], p1[j], j - R);
      fetch_side(side[j], j - 2 * R);
    }
  }
#pragma unroll 1
  for (int round = 0; round < ROUNDS; ++round) {
    const int base = -R + round * WIN;
    progress_priority(round);
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
      const int r = base + j;
      if (r >= STRIP + R) break;
      const int y = ys + r;
      __builtin_amdgcn_wave_barrier();
      line[lane] = p0[j];
      if (lane < 2 * R) line[64 + lane] = p1[j];
      __builtin_amdgcn_wave_barrier();
      fetch(p0[(j + AHEAD) % WIN], p1[(j + AHEAD) % WIN], r + AHEAD);
      f32x2 ab = f32x2{0.f, 0.f};
      float c = 0.f;
#pragma unroll
      for (int i = 0; i < WIN; ++i) {
        const f32x4 t = line[lane + i];
        ab = f32x2{t.x, t.y} * k.w[i] + ab;
        c = fmaf(k.w[i], t.z, c);
      }
#pragma unroll
      for (int d = -R; d <= R; ++d) {
        const int slot = (j + d + WIN) % WIN;
        const float wv = k.w[R - d];
        acc_ab[slot] = ab * wv + acc_ab[slot];
        acc_c[slot] = fmaf(wv, c, acc_c[slot]);
      }
      const int done = (j + WIN - R) % WIN;
      const int yo = y - R;
      if (r - R >= 0 && r - R < STRIP && yo < H) {          // uniform
        const float a = side[j].x, b = side[j].y;
        const float sg = (a > b) ? l1s : ((a < b) ? -l1s : 0.f);
        buf_store(fmaf(scale, acc_ab[done].x + 2.f * a * acc_ab[done].y + b * acc_c[done], sg), od, colx,
                  (uint32_t)(yo * W) * 4u);
      }
      fetch_side(side[(j + AHEAD) % WIN], r - R + AHEAD);
      acc_ab[done] = f32x2{0.f, 0.f};
      acc_c[done] = 0.f;
    }
  }
}

}  // namespace

}  // namespace ganet

using namespace ganet;

extern "C" {

#ifdef GANET_SSIM_TRACE
int ganet_dev_ssim_trace(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ssim_trace), sizeof(g_ssim_trace)); }
#endif

int64_t ganet_ssim_sums_floats(void) { return SUM_FLOATS; }

int ganet_ssim_fwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2, float norm,
                   float* sums, float* partials, void* stream_) {
  if (planes <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !sums || !partials) {
    set_error("ganet_ssim_fwd: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_hip(hipMemsetAsync(sums, 0, SUM_FLOATS * sizeof(float), stream), "memset ssim sums");
  if (rc) return rc;
  const int strips = (H + STRIP - 1) / STRIP;
  const dim3 grid((W + 63) / 64, (strips + WAVES - 1) / WAVES, planes);
  ProfScope prof_(K_SSIM_FWD, stream);
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(64 * WAVES), 0, stream, H, W, img1, img2, norm, sums,
                     partials, (size_t)planes * H * W, strips, make_window());
  return check_hip(hipGetLastError(), "ssim_fwd_kernel");
}

int ganet_ssim_bwd(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2,
                   const float* partials, float norm, const float* d_ssim, const float* d_l1,
                   float* dimg1, void* stream_) {
  if (planes <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !partials || !dimg1) {
    set_error("ganet_ssim_bwd: invalid arguments");
    return 1;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 grid((W + 63) / 64, (H + STRIP * WAVES - 1) / (STRIP * WAVES), planes);
  ProfScope prof_(K_SSIM_BWD, stream);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(64 * WAVES), 0, stream, H, W, img1, img2, partials,
                     (size_t)planes * H * W, norm, d_ssim, d_l1, dimg1, make_window());
  return check_hip(hipGetLastError(), "ssim_bwd_kernel");
}

}  // extern "C"
