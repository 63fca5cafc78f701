// ganet_conv.hip — the geometry-feature convolutions of the feature net: 5x5, stride 1, zero padding 2, no bias,
// 64 -> 64 channels on a 128 x 128 map (/root/reference/model/modules.py:114-137, GeomConvLayers: three of them in a
// row, no activation in between), forward, input gradient and weight gradient, on channels-last fp32 maps
// [b][H][W][64] — the layout the up-sampling kernel consumes (ganet_upsample.hip), so the NCHW <-> NHWC copies
// around the vendor kernels disappear.
//
// 3.36 GFLOP per pass on 4 MB tensors that live in the L2 / Infinity Cache: matrix-pipe work. As everywhere in the
// decoder the fp32 operands are split exactly into three bf16 pieces and multiplied on the bf16 pipe with fp32
// accumulation (ganet_split.h: six products per fp32 product, fp32-accurate).
//
//   conv5_kernel        implicit GEMM  y[p, n] = sum_{tap, k} x[p + tap, k] B_tap[k, n]:  a workgroup owns 64
//                       consecutive pixels of an image row x 64 output channels. Its 5 x 68 pixel halo is staged once in
//                       LDS (coalesced loads). Wave (output-channel half, k-quarter) accumulates two 32 pixel x 32
//                       channel tiles over its 16 of the tap's 64 input channels; its B fragments (pre-split by
//                       conv5_pack into bf16 planes laid out so that a fragment load is 2 x 512 contiguous bytes) come
//                       straight from the L2 into a register ring three taps ahead: the tap loop has no barrier and no
//                       LDS traffic but the halo reads (a first version that moved B through a double-buffered LDS image
//                       with a barrier per tap spent 2/3 of its time waiting). The k-quarters are added through LDS at
//                       the end. The input gradient is the same kernel on the transposed, tap-flipped weights
//                       (conv5_pack writes both variants).
//   conv5_wgrad_kernel  dW[n, k, dy, dx] = sum_p dy[p, n] x[p + (dy, dx), k]: the pixel index is the MFMA's reduction
//                       index. A workgroup = one kernel row dy x a chunk of 16-pixel runs; wave (n-tile, k-tile) keeps
//                       the five dx accumulators of its 32 x 32 tile: per run a lane loads 8 pixels of its dy column and
//                       the 12 pixels (8 + 4 shifts) of its x column, splits them once and forms the five shifted
//                       fragments by re-pairing the pieces (v_perm). Per-chunk partial tiles, deterministic reduction.
#include <cstdint>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"
#include "ganet_split.h"

namespace ganet {

namespace {

constexpr int C = 64;                       // channels in = out
constexpr int TAPS = 25;
constexpr int TAP_UNITS = 3 * 4 * 2 * C;                 // one tap's B image: [plane][k-step of 16][kg][n] 16-byte units:
                                                         // a wave's fragment load (plane, k-step) is 2 x 512 contiguous bytes
constexpr size_t PACKED_UNITS = (size_t)2 * TAPS * TAP_UNITS;   // one convolution: [variant][tap]

// variant 0 (forward):        B_tap[k = ci][n = co] = w[co][ci][dy][dx]
// variant 1 (input gradient): B_tap[k = co][n = ci] = w[co][ci][4 - dy][4 - dx]
struct PackJobs { const float* w[GANET_CONV5_MAX]; };
__global__ void __launch_bounds__(256)
conv5_pack_kernel(PackJobs jobs, u32x4* __restrict__ packed) {
  const int conv = blockIdx.y;
  const float* __restrict__ w = jobs.w[conv];
  const int i = blockIdx.x * 256 + threadIdx.x;          // (variant, tap, n, unit)
  if (i >= 2 * TAPS * C * 8) return;
  const int u = i & 7, n = (i >> 3) & 63, vt = i >> 9;
  const int tap = vt % TAPS, variant = vt / TAPS;
  const int dy = tap / 5, dx = tap % 5;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * u + e;
    v[e] = variant == 0 ? w[((size_t)n * C + k) * TAPS + dy * 5 + dx]
                        : w[((size_t)k * C + n) * TAPS + (4 - dy) * 5 + (4 - dx)];
  }
  u32x4 p1, p2, p3;
  split8(v, p1, p2, p3);
  u32x4* out = packed + (size_t)conv * PACKED_UNITS + (size_t)(variant * TAPS + tap) * TAP_UNITS;
  // unit u = 8 k of row n = (k-step u >> 1, kg u & 1)
  const int at = ((u >> 1) * 2 + (u & 1)) * C + n;
  out[0 * 8 * C + at] = p1;
  out[1 * 8 * C + at] = p2;
  out[2 * 8 * C + at] = p3;
}

#ifdef GANET_CONV_TRACE
// development (tools/conv_trace.py): s_memtime stamps of waves 0 and 7 of block 0, s_memrealtime start / end per block
__device__ unsigned long long g_conv_trace[2][8];
__device__ unsigned long long g_conv_blocks[1024][2];
__device__ unsigned long long g_wg_trace[2][8];
__device__ unsigned long long g_wg_blocks[1024][2];
#define WG_STAMP(I) do { if (blockIdx.x == 0 && blockIdx.y == 2 && lane == 0 && (wave == 0 || wave == 7)) g_wg_trace[wave == 7][I] = __builtin_amdgcn_s_memtime(); } while (0)
#define WG_BLOCK(I) do { if (lane == 0 && wave == 0) g_wg_blocks[blockIdx.y * gridDim.x + blockIdx.x][I] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CONV_STAMP(I) do { if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7)) g_conv_trace[wave == 7][I] = __builtin_amdgcn_s_memtime(); } while (0)
#define CONV_BLOCK(I) do { if (lane == 0 && wave == 0 && blockIdx.x < 1024) g_conv_blocks[blockIdx.x][I] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CONV_STAMP(I) do {} while (0)
#define CONV_BLOCK(I) do {} while (0)
#define WG_STAMP(I) do {} while (0)
#define WG_BLOCK(I) do {} while (0)
#endif

constexpr int WGC = 512;                                 // 8 waves: (32-channel half of the output) x (quarter of a tap's 64 k)
constexpr int HALO_W = 64 + 4;                           // pixels per halo row
constexpr int PIX_UNITS = 9;                             // 8 units of 8 bf16 channels + 1 pad: pixel records on different banks
constexpr int HALO_UNITS = 5 * HALO_W * PIX_UNITS;       // per plane
constexpr size_t CONV_LDS = (size_t)3 * HALO_UNITS * 16;
constexpr int LEAD = 3;                                  // taps whose B fragments are in flight

__global__ void __attribute__((amdgpu_flat_work_group_size(WGC, WGC), amdgpu_waves_per_eu(2, 2)))
conv5_kernel(int H, int W, const float* __restrict__ x, const u32x4* __restrict__ packed, float* __restrict__ y) {
  extern __shared__ u32x4 s_halo[];                      // [plane][5][HALO_W][PIX_UNITS] bf16 pieces; reused to add the k-quarters
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = wave & 1, kq = wave >> 1;
  const int lane = threadIdx.x & 63;
  const int kg = lane >> 5, li = lane & 31;
  const int per_row = W / 64;
  const int row = blockIdx.x / per_row;                  // image row, batch included
  const int py = row % H;
  const int px0 = (blockIdx.x - row * per_row) * 64;
  const float* img = x + (size_t)(row - py) * W * C;
  CONV_BLOCK(0); CONV_STAMP(0);

  // B fragments of this wave: (tap, plane) -> one coalesced 16-byte load per lane, straight from the packed image
  // (L2-resident, identical for every workgroup), LEAD taps ahead in a register ring — no LDS, no barrier per tap
  const u32x4* bp = packed + (kq * 2 + kg) * C + half * 32 + li;
  struct Frag { u32x4 b[3]; };
  auto load_b = [&](int tap, Frag& f) {
#pragma unroll
    for (int p = 0; p < 3; ++p) f.b[p] = bp[(size_t)tap * TAP_UNITS + p * 8 * C];
  };
  Frag ring[LEAD];
#pragma unroll
  for (int j = 0; j < LEAD; ++j) load_b(j, ring[j]);

  // stage the 5 x 68 pixel halo once: coalesced loads (a pixel = 8 lanes x 32 bytes), zeros outside the image, split
  // into the three bf16 planes here — every element is converted once, not once per tap and consumer wave
  // (all loads of a thread's items are issued first, from clamped addresses, and masked afterwards: with the loads
  // inside the bounds branch every item was its own memory round trip, six in a row before the first MFMA)
  constexpr int HALO_ITEMS = 5 * HALO_W * 8, HALO_IT = (HALO_ITEMS + WGC - 1) / WGC;
  float4 lo[HALO_IT], hi[HALO_IT];
#pragma unroll
  for (int k = 0; k < HALO_IT; ++k) {
    const int i = min((int)threadIdx.x + k * WGC, HALO_ITEMS - 1);
    const int u = i & 7, p = i >> 3;
    const int hy = p / HALO_W, hx = p - hy * HALO_W;
    const int yy = min(max(py + hy - 2, 0), H - 1), xx = min(max(px0 + hx - 2, 0), W - 1);
    const float* src = img + ((size_t)yy * W + xx) * C + 8 * u;
    lo[k] = *reinterpret_cast<const float4*>(src);
    hi[k] = *reinterpret_cast<const float4*>(src + 4);
  }
  CONV_STAMP(1);
#pragma unroll
  for (int k = 0; k < HALO_IT; ++k) {
    const int i = threadIdx.x + k * WGC;
    if (i >= HALO_ITEMS) break;
    const int u = i & 7, p = i >> 3;
    const int hy = p / HALO_W, hx = p - hy * HALO_W;
    const int yy = py + hy - 2, xx = px0 + hx - 2;
    const float on = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 1.f : 0.f;
    const float v[8] = {lo[k].x * on, lo[k].y * on, lo[k].z * on, lo[k].w * on, hi[k].x * on, hi[k].y * on, hi[k].z * on, hi[k].w * on};
    u32x4 p1, p2, p3;
    split8(v, p1, p2, p3);
    s_halo[p * PIX_UNITS + u] = p1;
    s_halo[HALO_UNITS + p * PIX_UNITS + u] = p2;
    s_halo[2 * HALO_UNITS + p * PIX_UNITS + u] = p3;
  }
  CONV_STAMP(2);
  __syncthreads();
  CONV_STAMP(3);

  // one accumulator per 32-pixel strip: consecutive MFMAs of a wave alternate between them
  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  // lane's pixel record at tap (0, 0); channels 16 kq + 8 kg .. + 7 are unit 2 kq + kg
  const u32x4* ap = s_halo + li * PIX_UNITS + 2 * kq + kg;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int dy = tap / 5, dx = tap - 5 * dy;
    const Frag f = ring[tap % LEAD];
    u32x4 a1[2], a2[2], a3[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const u32x4* at = ap + (dy * HALO_W + dx + 32 * st) * PIX_UNITS;
      a1[st] = at[0]; a2[st] = at[HALO_UNITS]; a3[st] = at[2 * HALO_UNITS];
    }
    __builtin_amdgcn_sched_barrier(kSchedMask);
    if (tap + LEAD < TAPS) load_b(tap + LEAD, ring[tap % LEAD]);
    __builtin_amdgcn_sched_barrier(kSchedMask);
    // the six products, the two strips interleaved
    acc[0] = mfma_bf16(a3[0], f.b[0], acc[0]); acc[1] = mfma_bf16(a3[1], f.b[0], acc[1]);
    acc[0] = mfma_bf16(a2[0], f.b[1], acc[0]); acc[1] = mfma_bf16(a2[1], f.b[1], acc[1]);
    acc[0] = mfma_bf16(a1[0], f.b[2], acc[0]); acc[1] = mfma_bf16(a1[1], f.b[2], acc[1]);
    acc[0] = mfma_bf16(a2[0], f.b[0], acc[0]); acc[1] = mfma_bf16(a2[1], f.b[0], acc[1]);
    acc[0] = mfma_bf16(a1[0], f.b[1], acc[0]); acc[1] = mfma_bf16(a1[1], f.b[1], acc[1]);
    acc[0] = mfma_bf16(a1[0], f.b[0], acc[0]); acc[1] = mfma_bf16(a1[1], f.b[0], acc[1]);
  }
  // the k-quarters 1..3 hand their tiles over through LDS; quarter 0 adds and stores.
  // C/D layout: column (output channel) = lane & 31, row (pixel of the strip) = (reg & 3) + 8 (reg >> 2) + 4 kg
  CONV_STAMP(4);
  __syncthreads();
  CONV_STAMP(5);
  float* s_t = reinterpret_cast<float*>(s_halo);         // [wave - 2][strip][reg][lane]
  if (kq > 0) {
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_t[(((wave - 2) * 2 + st) * 16 + r) * 64 + lane] = acc[st][r];
  }
  __syncthreads();
  CONV_STAMP(6);
  if (kq == 0) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      float* yo = y + ((size_t)row * W + px0 + st * 32 + 4 * kg) * C + half * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[st][r];
#pragma unroll
        for (int q = 1; q < 4; ++q) v += s_t[(((2 * q + half - 2) * 2 + st) * 16 + r) * 64 + lane];
        yo[(size_t)((r & 3) + 8 * (r >> 2)) * C] = v;
      }
    }
  }
  CONV_STAMP(7); CONV_BLOCK(1);
}

// ---------------------------------------------------------------------------------------------
constexpr int WG_STEPS = 22;      // 16-pixel runs per workgroup (1024 runs of a 128 x 128 map -> 47 chunks x 5 rows)

constexpr int WGW = 512;         // two groups of four waves take alternate runs of the chunk
__global__ void __attribute__((amdgpu_flat_work_group_size(WGW, WGW), amdgpu_waves_per_eu(2, 2)))
conv5_wgrad_kernel(int H, int W, int total_runs, const float* __restrict__ x, const float* __restrict__ dyv,
                   float* __restrict__ partial) {
  extern __shared__ float s_wg[];                        // the second group's tiles: [wave][dx][reg][lane]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  const int lane = threadIdx.x & 63;
  const int kg = lane >> 5, li = lane & 31;
  const int nt = w4 >> 1, kt = w4 & 1;                   // tile (output channel n, input channel k)
  const int dy = blockIdx.y;
  const int runs_per_row = W / 16;
  const int g0 = blockIdx.x * WG_STEPS + grp, g1 = min((int)(blockIdx.x + 1) * WG_STEPS, total_runs);
  WG_BLOCK(0); WG_STAMP(0);

  f32x16 acc[5];
#pragma unroll
  for (int d = 0; d < 5; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  struct Raw { float a[8], b[12]; };
  auto load = [&](int g, Raw& r) {
    const int gg = min(g, total_runs - 1);
    const int row = gg / runs_per_row, x0 = (gg - row * runs_per_row) * 16 + 8 * kg;
    const int py = row % H, yi = py + dy - 2;
    const bool rowok = yi >= 0 && yi < H;
    const float* ap = dyv + ((size_t)row * W + x0) * C + nt * 32 + li;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.a[i] = ap[(size_t)i * C];
    const float* bp = x + ((size_t)(row - py + (rowok ? yi : py)) * W) * C + kt * 32 + li;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int xi = x0 + j - 2;
      const bool ok = rowok && xi >= 0 && xi < W;
      const float val = bp[(size_t)(ok ? xi : x0) * C];
      r.b[j] = ok ? val : 0.f;
    }
  };
  auto compute = [&](const Raw& r) {
    u32x4 a1, a2, a3;
    split8(r.a, a1, a2, a3);
    // the 12 pixels once: pieces as fp32 bit patterns (their high halves are the bf16 values)
    float p1[12], p2[12], p3[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      p1[j] = r.b[j];
      p2[j] = r.b[j] - u2f(f2u(r.b[j]) & 0xffff0000u);
      p3[j] = p2[j] - u2f(f2u(p2[j]) & 0xffff0000u);
    }
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      u32x4 b1, b2, b3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        b1[q] = pack_hi(p1[d + 2 * q], p1[d + 2 * q + 1]);
        b2[q] = pack_hi(p2[d + 2 * q], p2[d + 2 * q + 1]);
        b3[q] = pack_hi(p3[d + 2 * q], p3[d + 2 * q + 1]);
      }
      GANET_SPLIT_PRODUCTS(acc[d], a1, a2, a3, b1, b2, b3);
    }
  };
  // this group's runs are g0, g0 + 2, ...; loads run two runs ahead in a ring of three
  Raw ring[3];
  load(g0, ring[0]);
  load(g0 + 2, ring[1]);
  WG_STAMP(1);
  for (int g = g0; g < g1; g += 6) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      load(g + 2 * u + 4, ring[(u + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 2 * u < g1) compute(ring[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  WG_STAMP(2);
  if (grp == 1) {
#pragma unroll
    for (int d = 0; d < 5; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_wg[((w4 * 5 + d) * 16 + r) * 64 + lane] = acc[d][r];
  }
  WG_STAMP(3);
  __syncthreads();
  WG_STAMP(4);
  if (grp == 0) {
    // partial[chunk][dy][dx][n][k]
    float* out = partial + (((size_t)blockIdx.x * 5 + dy) * 5) * C * C;
#pragma unroll
    for (int d = 0; d < 5; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        out[((size_t)d * C + n) * C + kt * 32 + li] = acc[d][r] + s_wg[((w4 * 5 + d) * 16 + r) * 64 + lane];
      }
  }
  WG_STAMP(5); WG_BLOCK(1);
}

// dW[n][k][dy][dx] = sum over chunks of partial[chunk][dy][dx][n][k]
__global__ void __launch_bounds__(256)
conv5_wgrad_reduce_kernel(int nchunks, const float* __restrict__ partial, float* __restrict__ dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // (tap, n, k), k fastest: coalesced reads
  if (i >= TAPS * C * C) return;
  float s = 0.f;
  for (int c0 = 0; c0 < nchunks; c0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = c0 + u < nchunks ? partial[(size_t)(c0 + u) * TAPS * C * C + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  const int k = i & 63, n = (i >> 6) & 63, tap = i >> 12;
  dw[((size_t)n * C + k) * TAPS + tap] = s;
}

int chunks_of(int total_runs) { return (total_runs + WG_STEPS - 1) / WG_STEPS; }
bool shape_ok(int b, int H, int W) { return b > 0 && H > 0 && W > 0 && (W % 64) == 0 && (int64_t)b * H * W < (1 << 24); }

}  // namespace

}  // namespace ganet

using namespace ganet;

#ifdef GANET_CONV_TRACE
extern "C" int ganet_dev_wgrad_trace(void* tr, void* bl) {
  if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_wg_trace), sizeof(g_wg_trace)) != hipSuccess) return 1;
  return (int)hipMemcpyFromSymbol(bl, HIP_SYMBOL(g_wg_blocks), sizeof(g_wg_blocks));
}
extern "C" int ganet_dev_conv_trace(void* tr, void* bl) {
  if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_conv_trace), sizeof(g_conv_trace)) != hipSuccess) return 1;
  return (int)hipMemcpyFromSymbol(bl, HIP_SYMBOL(g_conv_blocks), sizeof(g_conv_blocks));
}
#endif

extern "C" {

size_t ganet_conv5_packed_bytes(int32_t n_convs) { return n_convs > 0 ? (size_t)n_convs * PACKED_UNITS * 16 : 0; }

int ganet_conv5_pack(int32_t n_convs, const float* const* w, void* packed, void* stream_) {
  if (n_convs <= 0 || n_convs > GANET_CONV5_MAX || !w || !packed || !aligned16(packed)) {
    set_error("ganet_conv5_pack: invalid arguments (1 <= n_convs <= %d)", GANET_CONV5_MAX);
    return 1;
  }
  PackJobs jobs{};
  for (int i = 0; i < n_convs; ++i) {
    if (!w[i]) { set_error("ganet_conv5_pack: weight %d is NULL", i); return 1; }
    jobs.w[i] = w[i];
  }
  hipLaunchKernelGGL(conv5_pack_kernel, dim3((2 * TAPS * C * 8 + 255) / 256, n_convs), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), jobs, static_cast<u32x4*>(packed));
  return check_hip(hipGetLastError(), "conv5_pack_kernel");
}

int ganet_conv5_apply(int32_t b, int32_t H, int32_t W, const float* x, const void* packed, int32_t conv,
                      int32_t input_gradient, float* y, void* stream_) {
  if (!shape_ok(b, H, W) || !x || !packed || !y || conv < 0 || conv >= GANET_CONV5_MAX || !aligned16(x) ||
      !aligned16(packed) || x == y) {
    set_error("ganet_conv5_apply: invalid arguments (b=%d H=%d W=%d; W %% 64 == 0, 64 channels, channels-last, "
              "out of place)", b, H, W);
    return 1;
  }
  const u32x4* p = static_cast<const u32x4*>(packed) + (size_t)conv * PACKED_UNITS +
                   (size_t)(input_gradient ? 1 : 0) * TAPS * TAP_UNITS;
  const size_t lds = CONV_LDS;
  static PerDeviceFlag attr_set;       
  if (!attr_set) {
    if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(conv5_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute")) return 3;
    attr_set = true;
  }
  hipLaunchKernelGGL(conv5_kernel, dim3(b * H * (W / 64)), dim3(WGC), lds, static_cast<hipStream_t>(stream_), H, W, x,
                     p, y);
  return check_hip(hipGetLastError(), "conv5_kernel");
}

size_t ganet_conv5_wgrad_workspace(int32_t b, int32_t H, int32_t W) {
  if (!shape_ok(b, H, W)) return 0;
  return (size_t)chunks_of(b * H * (W / 16)) * TAPS * C * C * sizeof(float);
}

int ganet_conv5_wgrad(int32_t b, int32_t H, int32_t W, const float* x, const float* dy, float* dw, void* workspace,
                      size_t workspace_bytes, void* stream_) {
  if (!shape_ok(b, H, W) || !x || !dy || !dw) {
    set_error("ganet_conv5_wgrad: invalid arguments (b=%d H=%d W=%d; W %% 64 == 0)", b, H, W);
    return 1;
  }
  const size_t need = ganet_conv5_wgrad_workspace(b, H, W);
  if (!workspace || workspace_bytes < need) {
    set_error("ganet_conv5_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    return 2;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int runs = b * H * (W / 16), nchunks = chunks_of(runs);
  float* partial = static_cast<float*>(workspace);
  const size_t lds = (size_t)4 * 5 * 16 * 64 * sizeof(float);
  static PerDeviceFlag attr_set;       
  if (!attr_set) {
    if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(conv5_wgrad_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute")) return 3;
    attr_set = true;
  }
  hipLaunchKernelGGL(conv5_wgrad_kernel, dim3(nchunks, 5), dim3(WGW), lds, stream, H, W, runs, x, dy, partial);
  if (int rc = check_hip(hipGetLastError(), "conv5_wgrad_kernel")) return rc;
  hipLaunchKernelGGL(conv5_wgrad_reduce_kernel, dim3((TAPS * C * C + 255) / 256), dim3(256), 0, stream, nchunks,
                     partial, dw);
  return check_hip(hipGetLastError(), "conv5_wgrad_reduce_kernel");
}

}  // extern "C"
