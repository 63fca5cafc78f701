// gsr_common.h — shared declarations of the gfx950 rasterizer kernels (internal).
//
// Pipeline (one frame, everything on one stream, no host read-back):
//
//   K1 preprocess      per Gaussian : project, EWA covariance, conic, radius, tile rect;
//                                     per-tile histogram of (tile,Gaussian) pairs (a workgroup's rectangles counted
//                                     in an LDS window of the tile grid: difference array + prefix sum)
//   K2 tile_scan       one block    : exclusive scan of the histogram -> tile_offset, D; the tile order (longest list
//                                     first) and the work list of the long-list sort
//   K3 scatter         per Gaussian : append (depth_bits<<32 | index) to each touched tile (the same window count, one
//                                     returning atomic per (workgroup, tile), a wave's pairs dealt to its lanes)
//   K4 tile_sort       per tile     : merge sort of 2048-key chunks in LDS, chunk runs merged in LDS; lists beyond 8192
//                                     keys as independent ~4096-key output buckets (regular sampling) -> point_list
//   K5 render_fwd      per 4x4 block: cull the tile list, blend 64 survivors at a time (entry-parallel,
//                                     DPP wave scans); records the consumed segments
//   K6 render_bwd      per segment  : forward-order gradients of the segment's 64 entries summed over the block's
//                                     pixels in registers, one atomic record per (block, Gaussian) survivor
//   K7 preprocess_bwd  per Gaussian : screen-space grads -> means3D / scales / rotations
//
// Behavioural spec: SURVEY.md Appendix A (the reference's rasterizer is the un-vendored
// diff_gaussian_rasterization package used at /root/reference/gaussian_renderer/__init__.py:6).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include <type_traits>

#include "gsr.h"

#define GSR_TILE_PIX (GSR_TILE * GSR_TILE)   // 256 threads per tile = 4 wave64
#define GSR_WAVE 64
#define GSR_GRAD_STRIDE 16                   // floats per Gaussian in grad_acc: one 64-byte line per record, so a
                                             // wave's 9 atomics on a Gaussian touch exactly one line
#ifndef GSR_SUB
#define GSR_SUB 4                            // edge of a wave's pixel block in the render kernels: 4, or 8 (lane = pixel);
#endif                                       // 8 was built and measured in round 4: see gsr_render.hip
#define GSR_SEG_PIX (GSR_SUB * GSR_SUB)      // pixels of a render block = checkpoints per segment
#ifndef GSR_BWD_BLOCKS
#define GSR_BWD_BLOCKS 8192                  // workgroups of render_bwd per launch (round 6: 2048 -> 8192 with a limit on the waves
                                             // that take part, gsr_render.hip GSR_BWD_SEGS: 442 -> 403 us at 2.3 M pairs per frame,
                                             // 527 -> 481 at 5 M, 104 -> 104 at the avatar set; 4096 without the limit: 100 / 420 / 507)
#endif
#define GSR_SEG_BLOCKS ((GSR_TILE / GSR_SUB) * (GSR_TILE / GSR_SUB))   // render blocks of a tile: 16 (or 4)
#define GSR_PAIR_GRAD 9                      // floats of a per-pair gradient record (dxy2, dconic3, dopac1, drgb3)

namespace gsr {

// "done once per device" flag for per-function attributes (hipFuncSetAttribute is a per-device setting; a process may
// drive several GPUs): bit d = done on device d. Used as `static PerDeviceFlag f; if (!f) { ...; f = true; }`.
struct PerDeviceFlag {
  std::atomic<uint64_t> mask{0};
  static uint64_t bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
  bool operator!() const { return !(mask.load(std::memory_order_acquire) & bit()); }
  PerDeviceFlag& operator=(bool v) { if (v) mask.fetch_or(bit(), std::memory_order_release); return *this; }
};

struct Dims {
  int P, W, H, gx, gy, T;
  int64_t max_pairs;
  int seg_cap;        // capacity of the forward pass's segment records (see seg_capacity)
};

// Segment slots (64 survivors of one render block each) the forward pass may record. A block of a tile with n list
// entries records at most ceil(n / 64) segments; the B = GSR_SEG_BLOCKS blocks of tile t (list [o, o + n)) own the slots
//   B (o / 64 + t) + b c + s,   c = (o + n) / 64 - o / 64 + 1 >= ceil(n / 64)     (integer divisions)
// — disjoint between tiles, no counters, no overflow: sum over tiles <= B (D / 64 + T).
inline int seg_capacity(int T, int64_t max_pairs) {
  const int64_t c = max_pairs * GSR_SEG_BLOCKS / 64 + GSR_SEG_BLOCKS * (int64_t)T + GSR_SEG_BLOCKS;
  return (int)(c > 0x3fffffff ? 0x3fffffff : c);
}
__host__ __device__ inline int seg_block_capacity(int64_t start, int64_t end) { return (int)((end >> 6) - (start >> 6)) + 1; }
__host__ __device__ inline int64_t seg_first_slot(int64_t start, int tile) { return GSR_SEG_BLOCKS * ((start >> 6) + tile); }

// Work items of the tile sort's long-list pass (gsr_binning.hip: lists beyond GSR_SORT_LDS_KEYS keys are cut into output
// buckets of ~GSR_SORT_BUCKET keys): sum over such lists of ceil(n / GSR_SORT_BUCKET) <= max_pairs / GSR_SORT_BUCKET +
// (number of such lists <= max_pairs / GSR_SORT_LDS_KEYS).
#define GSR_SORT_LDS_KEYS 8192
#define GSR_SORT_BUCKET 4096
inline int64_t sort_work_capacity(int64_t max_pairs) { return 3 * (max_pairs / GSR_SORT_LDS_KEYS) + 16; }

// tile_scan_kernel (gsr_binning.hip) leaves the tiles in size order (largest list class first) when a thread of its
// one workgroup owns at most 8 tiles, i.e. T <= 8192; above that `tile_count` holds the identity order and every walk
// over it must skip empty tiles instead of stopping at the first one.
inline int tile_order_is_sorted(int T) { return (T + 1023) / 1024 <= 8 ? 1 : 0; }

inline Dims make_dims(int P, int W, int H, int64_t max_pairs) {
  Dims d;
  d.P = P; d.W = W; d.H = H;
  d.gx = (W + GSR_TILE - 1) / GSR_TILE;
  d.gy = (H + GSR_TILE - 1) / GSR_TILE;
  d.T = d.gx * d.gy;
  d.max_pairs = max_pairs;
  d.seg_cap = seg_capacity(d.T, max_pairs);
  return d;
}

// Resolved device pointers into the caller-owned workspace.
struct Workspace {
  float* depth;
  float2* xy;
  float4* xyext;
  float4* conic_opacity;
  float4* rgb;
  float* cov3d;
  int4* rect;
  uint32_t* tiles_touched;
  uint8_t* clamped;
  uint32_t* tile_count;
  uint32_t* tile_offset;
  uint32_t* tile_cursor;
  uint64_t* pair_key;
  uint32_t* point_list;
  uint64_t* pair_tmp;
  float* final_T;
  uint32_t* n_contrib;
  float* grad_acc;
  int32_t* status;
  int32_t* seg_heads;
  uint32_t* seg_count;
  uint2* seg_entries;
  float4* seg_ckpt;
  uint2* seg_info;
  float4* pix_accum;
  float* pair_grad;
  uint32_t* seg_list;
  uint32_t* sort_work;
};

// Batched launches: blockIdx.y = frame. Element strides between frames (0 = shared by all
// frames); outputs and gradients are contiguous [frames, ...].
struct Batch {
  int frames;
  size_t ws_stride;   // bytes between per-frame workspaces
  int64_t means, colors, opacities, scales, rotations, cov3d, view, proj, shs, campos;
};

__host__ __device__ inline Workspace frame_ws(Workspace w, size_t bytes) {
  auto mv = [bytes](auto*& p) { p = reinterpret_cast<std::remove_reference_t<decltype(p)>>(reinterpret_cast<char*>(p) + bytes); };
  mv(w.depth); mv(w.xy); mv(w.xyext); mv(w.conic_opacity); mv(w.rgb); mv(w.cov3d); mv(w.rect); mv(w.tiles_touched);
  mv(w.clamped); mv(w.tile_count); mv(w.tile_offset); mv(w.tile_cursor); mv(w.pair_key);
  mv(w.point_list); mv(w.pair_tmp); mv(w.final_T); mv(w.n_contrib); mv(w.grad_acc); mv(w.status); mv(w.seg_heads); mv(w.seg_count);
  mv(w.seg_entries); mv(w.seg_ckpt); mv(w.seg_info); mv(w.pix_accum); mv(w.pair_grad); mv(w.seg_list); mv(w.sort_work);
  return w;
}

// Half extents of the axis-aligned box around the region where Gaussian (conic A,B,C, opacity o) reaches
// alpha >= 1/255:  alpha >= 1/255  <=>  d^T Q d <= tau, tau = 2 ln(255 o); the box of that ellipse has
// half extents sqrt(tau Sigma_xx), sqrt(tau Sigma_yy) with Sigma = Q^-1. Slightly inflated (conservative);
// a Gaussian that can never reach 1/255 gets a negative extent (outside every box), NaN means "keep".
// Written once per Gaussian by K1 (xyext = centre + extents), consumed by the render kernels' culling.
__device__ __forceinline__ float2 alpha_extent(float4 co) {
  const float tau = 2.0f * logf(255.0f * co.w) + 1e-3f;
  const float det = co.x * co.z - co.y * co.y;
  const float inv = 1.0f / det;
  float hx = sqrtf(tau * co.z * inv) * 1.001f + 0.01f;
  float hy = sqrtf(tau * co.x * inv) * 1.001f + 0.01f;
  if (tau < 0.0f) hx = hy = -1e30f;
  return make_float2(hx, hy);
}

// ---- per-workgroup aggregation of (tile, Gaussian) pairs in LDS (K1 histogram, K3 scatter), round 6.
// The 256 Gaussians of a workgroup are UV neighbours: their tile rectangles lie in a WINDOW of the tile grid — the
// bounding box of the rectangles — that is a few hundred cells for an avatar and at most the whole grid. The pairs of the
// workgroup are counted per cell of that window in LDS, DIRECTLY INDEXED, as a 2-D difference array: a rectangle
// [x0,x1) x [y0,y1) is four marks (+1 at (x0,y0), -1 at (x1,y0), -1 at (x0,y1), +1 at (x1,y1)), a 2-D prefix sum over
// the window turns the marks into the count of rectangles covering each cell — the cost no longer depends on how many
// tiles a Gaussian touches (rounds 1-5 walked every rectangle lane by lane through a hash table keyed by tile id:
// 30 us at 4 tiles per Gaussian, 1.2 ms at 20: profiles/r05_dsweep.txt). Then ONE global atomic per (workgroup, cell).
// A window larger than the LDS array (only possible with more than GSR_WIN_CELLS tiles in the frame) falls back to the
// hash table (TileAgg below).
#define GSR_WIN_CELLS 8192                       // 32 KiB of LDS: the whole grid of a 1920x1080 frame (120 x 68)
struct TileWin { int x0, y0, w, h; };            // w == 0: no rectangle in the workgroup; w < 0: does not fit
// LDS bytes of the K1 / K3 launches for a frame of T tiles (the hash table needs 12 KiB whatever T is)
inline size_t win_lds_bytes(int T) {
  const int cells = T < GSR_WIN_CELLS ? T : GSR_WIN_CELLS;
  return (size_t)(cells < 3072 ? 3072 : cells) * 4;
}
// Bounding window of the rectangles of the workgroup's threads (`use` = this thread has one). All threads call;
// s_box = 4 ints of LDS; ends with a workgroup barrier.
__device__ __forceinline__ TileWin wg_tile_window(const int4& rc, bool use, int* s_box, int cells) {
  if (threadIdx.x == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = -1; s_box[3] = -1; }
  int x0 = use ? rc.x : 0x7fffffff, y0 = use ? rc.y : 0x7fffffff, x1 = use ? rc.z : -1, y1 = use ? rc.w : -1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
    x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && x1 >= 0) {
    atomicMin(&s_box[0], x0); atomicMin(&s_box[1], y0); atomicMax(&s_box[2], x1); atomicMax(&s_box[3], y1);
  }
  __syncthreads();
  TileWin wn;
  wn.x0 = s_box[0]; wn.y0 = s_box[1]; wn.w = s_box[2] - s_box[0]; wn.h = s_box[3] - s_box[1];
  if (s_box[2] < 0) { wn.x0 = wn.y0 = wn.w = wn.h = 0; }
  else if ((int64_t)wn.w * wn.h > cells) wn.w = -1;
  return wn;
}
// the four marks of one rectangle (marks on the window's right / bottom edge would only reach cells outside it)
__device__ __forceinline__ void win_mark(int* cell, const TileWin& wn, const int4& rc) {
  const int ax = rc.x - wn.x0, ay = rc.y - wn.y0, bx = rc.z - wn.x0, by = rc.w - wn.y0;
  atomicAdd(&cell[ay * wn.w + ax], 1);
  if (bx < wn.w) atomicAdd(&cell[ay * wn.w + bx], -1);
  if (by < wn.h) {
    atomicAdd(&cell[by * wn.w + ax], -1);
    if (bx < wn.w) atomicAdd(&cell[by * wn.w + bx], 1);
  }
}
// 2-D inclusive prefix sum over the window (marks -> rectangles covering each cell). All threads call; the marks must be
// visible (barrier before); ends with a barrier. Rows: one wave per row, 64 columns per step (consecutive addresses);
// columns: one thread per column (consecutive threads = consecutive banks), four rows in flight.
__device__ __forceinline__ void win_prefix(int* cell, const TileWin& wn) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < wn.h; r += nw) {
    int carry = 0;
    for (int c0 = 0; c0 < wn.w; c0 += 64) {
      const int c = c0 + lane;
      int v = c < wn.w ? cell[r * wn.w + c] : 0;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off);
        if (lane >= off) v += t;
      }
      v += carry;
      if (c < wn.w) cell[r * wn.w + c] = v;
      carry = __shfl(v, 63);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < wn.w; c += blockDim.x) {
    int run = 0, r = 0;
    for (; r + 4 <= wn.h; r += 4) {
      const int a0 = cell[r * wn.w + c], a1 = cell[(r + 1) * wn.w + c], a2 = cell[(r + 2) * wn.w + c],
                a3 = cell[(r + 3) * wn.w + c];
      run += a0; cell[r * wn.w + c] = run;
      run += a1; cell[(r + 1) * wn.w + c] = run;
      run += a2; cell[(r + 2) * wn.w + c] = run;
      run += a3; cell[(r + 3) * wn.w + c] = run;
    }
    for (; r < wn.h; ++r) { run += cell[r * wn.w + c]; cell[r * wn.w + c] = run; }
  }
  __syncthreads();
}

// ---- the fallback for windows beyond GSR_WIN_CELLS cells (rounds 1-5's path): a hash table keyed by tile id.
// The 256 Gaussians of a workgroup are UV neighbours and hit a few dozen distinct tiles. A serial chain
// of wave-aggregated global atomics paid one memory round trip per distinct tile and rect step
// (scatter: 24-30 us per 200k-Gaussian frame, 146 us at 300k / 1080p); here the pairs are first counted
// in an LDS hash table keyed by tile id, then ONE global atomic per (workgroup, tile) is issued — all of a
// workgroup's in one round — and the pairs take their slots from LDS cursors.
#define GSR_AGG_SLOTS 1024
#define GSR_AGG_PROBES 48
struct TileAgg {
  int key[GSR_AGG_SLOTS];          // tile id, -1 = free
  uint32_t cnt[GSR_AGG_SLOTS];     // pairs of this workgroup in the tile; reused as cursor
  uint32_t base[GSR_AGG_SLOTS];    // first global slot (scatter only)
};
__device__ __forceinline__ void agg_clear(TileAgg& t) {
  for (int s = threadIdx.x; s < GSR_AGG_SLOTS; s += blockDim.x) { t.key[s] = -1; t.cnt[s] = 0u; }
}
__device__ __forceinline__ int agg_hash(int tile) { return (int)(((uint32_t)tile * 2654435761u) >> 22); }
// slot of `tile`, claiming a free one if needed; -1 if the table is full along the probe sequence
__device__ __forceinline__ int agg_claim(TileAgg& t, int tile) {
  int h = agg_hash(tile);
  for (int p = 0; p < GSR_AGG_PROBES; ++p) {
    const int k = atomicCAS(&t.key[h], -1, tile);
    if (k == -1 || k == tile) return h;
    h = (h + 1) & (GSR_AGG_SLOTS - 1);
  }
  return -1;
}
// slot of a tile that agg_claim placed (or -1 if it had fallen back)
__device__ __forceinline__ int agg_find(const TileAgg& t, int tile) {
  int h = agg_hash(tile);
  for (int p = 0; p < GSR_AGG_PROBES; ++p) {
    const int k = t.key[h];
    if (k == tile) return h;
    if (k == -1) return -1;
    h = (h + 1) & (GSR_AGG_SLOTS - 1);
  }
  return -1;
}

// A Gaussian whose tile rectangle holds more than GSR_BIG_RECT tiles (a handful per frame at most, but thousands of
// tiles each: a stage-2 scene had a few that covered the screen) is not walked by its own lane through the LDS table
// — it would fill the table and turn every later claim of the workgroup into a full probe sequence (measured: 3.6 ms
// instead of 50 us for K3) — but by its whole wave, one tile per lane and step, with direct global atomics.
#define GSR_BIG_RECT 64
__device__ __forceinline__ bool rect_is_big(const int4& rc) {
  return (rc.z - rc.x) * (rc.w - rc.y) > GSR_BIG_RECT;
}
// calls f(tile, payload) for every tile of every big rectangle held by a lane of this wave (all lanes must call);
// payload = bcast(src lane), evaluated by ALL lanes before the tiles are dealt out (wave shuffles belong there)
template <typename G, typename F>
__device__ __forceinline__ void for_big_rects(const int4& rc, int gx, G bcast, F f) {
  uint64_t pending = __ballot(rect_is_big(rc));
  const int lane = threadIdx.x & 63;
  while (pending) {
    const int src = __ffsll((unsigned long long)pending) - 1;
    pending &= pending - 1;
    const int x0 = __shfl(rc.x, src), y0 = __shfl(rc.y, src), x1 = __shfl(rc.z, src), y1 = __shfl(rc.w, src);
    const int w = x1 - x0, n = w * (y1 - y0);
    const auto payload = bcast(src);
    for (int t = lane; t < n; t += 64) f((y0 + t / w) * gx + x0 + t % w, payload);
  }
}

int compute_layout(int P, int W, int H, int64_t max_pairs, GsrLayout* out);
Workspace resolve(void* base, const GsrLayout& L);
void set_error(const char* fmt, ...);
void trace_sync(hipStream_t stream, const char* what);      // gsr_set_trace: announce + wait (development)

// Kernel launchers (each returns a hipError_t from the launch).
hipError_t launch_preprocess(const GsrSettings& s, const Dims& d, const float* means3D,
                             const float* colors_precomp, const float* opacities,
                             const float* scales, const float* rotations,
                             const float* cov3D_precomp, const Workspace& ws, int32_t* radii,
                             const Batch& bt, hipStream_t stream);
hipError_t launch_binning(const Dims& d, const Workspace& ws, const Batch& bt, hipStream_t stream);
hipError_t launch_render_fwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             float* out_color, bool record, const Batch& bt, hipStream_t stream);
hipError_t launch_render_bwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             const float* dL_dout, const Batch& bt, hipStream_t stream);
hipError_t launch_preprocess_bwd(const GsrSettings& s, const Dims& d, const float* means3D,
                                 const float* scales, const float* rotations,
                                 const int32_t* radii, const Workspace& ws,
                                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors,
                                 float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                                 float* dL_dcov3D, int32_t* overflow_flag, const Batch& bt, hipStream_t stream);
// SH colour path (gsr_sh.hip): runs after K1 / after K7 when `shs` is given.
hipError_t launch_sh_color(const GsrSettings& s, const Dims& d, const float* means3D,
                           const float* shs, int sh_coeffs, const Workspace& ws, const Batch& bt,
                           hipStream_t stream);
hipError_t launch_sh_bwd(const GsrSettings& s, const Dims& d, const float* means3D, const float* shs,
                         int sh_coeffs, const Workspace& ws, float* dL_dsh, float* dL_dmeans3D,
                         const Batch& bt, hipStream_t stream);
// Opt-in per-kernel timing (gsr_profile_* in gsr.h).
enum KernelId { K_PREPROCESS = 0, K_SCAN, K_SCATTER, K_SORT, K_RENDER_FWD, K_RENDER_BWD,
                K_PREPROCESS_BWD, K_COUNT };
struct ProfScope {
  ProfScope(KernelId id, hipStream_t stream);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix,
                               uint8_t* out, hipStream_t stream);

}  // namespace gsr
