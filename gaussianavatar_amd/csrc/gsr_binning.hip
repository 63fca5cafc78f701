// gsr_binning.hip — K2 tile_scan, K3 scatter, K4 tile_sort.
//
// Replaces the reference's global pipeline (InclusiveSum over Gaussians -> blocking D2H read
// of the pair count -> duplicateWithKeys -> 64-bit global radix sort -> identifyTileRanges;
// SURVEY.md §2.1) with a per-tile one that never leaves the device:
//   histogram (in K1) -> one-block scan over tiles -> atomic append per tile -> per-tile sort
//   of (depth_bits<<32 | gaussian) keys in LDS.
// The per-tile order is the same total order (depth bits, then Gaussian index; Appendix A.2),
// so the lists are bit-identical to the reference's however the appends interleave.
#include "gsr_common.h"

namespace gsr {

namespace {

constexpr int SCAN_THREADS = 1024;
constexpr int MERGE_ITEMS = 8;        // outputs per thread per merge step

// ------------------------------------------------------------------ K2
__global__ void __launch_bounds__(SCAN_THREADS)
tile_scan_kernel(int T, int64_t max_pairs, uint32_t* __restrict__ tile_count,
                 uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ tile_cursor,
                 int32_t* __restrict__ status, size_t ws_stride) {
  {
    const size_t off = (size_t)blockIdx.y * ws_stride;   // batched launch: this frame's workspace
    tile_count = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tile_count) + off);
    tile_offset = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tile_offset) + off);
    tile_cursor = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tile_cursor) + off);
    status = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(status) + off);
  }
  __shared__ uint32_t s_wave[SCAN_THREADS / GSR_WAVE];
  __shared__ uint32_t s_max[SCAN_THREADS / GSR_WAVE];
  const int tid = threadIdx.x;
  const int per = (T + SCAN_THREADS - 1) / SCAN_THREADS;
  const int lo = tid * per;
  const int hi = min(lo + per, T);
  uint32_t sum = 0, mx = 0;
  for (int t = lo; t < hi; ++t) {
    const uint32_t c = tile_count[t];
    sum += c;
    mx = max(mx, c);
  }
  // inclusive scan of `sum` across the wave, then across the 16 waves
  const int lane = tid & (GSR_WAVE - 1), wave = tid / GSR_WAVE;
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < GSR_WAVE; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
#pragma unroll
  for (int off = GSR_WAVE / 2; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
  if (lane == GSR_WAVE - 1) s_wave[wave] = incl;
  if (lane == 0) s_max[wave] = mx;
  __syncthreads();
  uint32_t base = 0, total = 0, gmax = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / GSR_WAVE; ++w) {
    if (w < wave) base += s_wave[w];
    total += s_wave[w];
    gmax = max(gmax, s_max[w]);
  }
  uint32_t run = base + incl - sum;   // exclusive prefix of this thread's chunk
  for (int t = lo; t < hi; ++t) {
    tile_offset[t] = run;
    tile_cursor[t] = run;
    run += tile_count[t];
  }
  if (tid == 0) {
    tile_offset[T] = total;
    status[0] = (int32_t)total;
    status[1] = ((int64_t)total > max_pairs) ? 1 : 0;
    status[3] = (int32_t)gmax;
  }
  // The histogram is dead from here on: its buffer is reused for the tile ORDER the sort kernel walks —
  // tiles by size class (1 + floor(log2 count)), largest class first. The per-tile sort holds 64 KiB of
  // LDS, so only two of its workgroups fit on a CU and the launch is a list-scheduling problem: with
  // the long lists of the avatar's interior tiles dispatched first, the short ones fill the gaps instead
  // of a few long ones forming the tail. Order within a class is arbitrary (it only affects scheduling).
  constexpr int MAXPER = 8, NCLS = 34;              // MAXPER x SCAN_THREADS = 8192: gsr_common.h tile_order_is_sorted
  __shared__ uint32_t s_cls[NCLS];
  uint32_t pos[MAXPER];
  const bool ordered = per <= MAXPER;
  if (tid < NCLS) s_cls[tid] = 0;
  __syncthreads();
  auto cls_of = [](uint32_t c) { return c ? 32 - __clz(c) : 0; };   // 0 for empty, else 1..32
  // most tiles are empty (class 0): those are counted / placed once per wave (ballot), the rest per tile
  auto place = [&](int c, bool on) -> uint32_t {
    const unsigned long long empt = __ballot(on && c == 0);
    uint32_t base = 0;
    if (empt && lane == __ffsll((long long)empt) - 1) base = atomicAdd(&s_cls[0], (uint32_t)__popcll(empt));
    base = __shfl(base, empt ? __ffsll((long long)empt) - 1 : 0);
    if (on && c == 0) return base + (uint32_t)__popcll(empt & ((1ull << lane) - 1ull));
    return on ? atomicAdd(&s_cls[c], 1u) : 0u;
  };
  if (ordered)
    for (int k = 0; k < per; ++k) {                      // uniform trip count: the ballots need every lane
      const int t = lo + k;
      place(t < hi ? cls_of(tile_count[t]) : -1, t < hi);
    }
  __syncthreads();
  if (tid == 0) {
    uint32_t run2 = 0;
    for (int c = NCLS - 1; c >= 0; --c) { const uint32_t v = s_cls[c]; s_cls[c] = run2; run2 += v; }
  }
  __syncthreads();
  if (ordered)
    for (int k = 0; k < per; ++k) {
      const int t = lo + k;
      pos[k] = place(t < hi ? cls_of(tile_count[t]) : -1, t < hi);
    }
  __syncthreads();                    // every count has been read: the buffer may be overwritten
  for (int t = lo, k = 0; t < hi; ++t, ++k) tile_count[ordered ? pos[k] : t] = (uint32_t)t;
}

// ------------------------------------------------------------------ K3
// Append (depth_bits << 32 | index) to every tile a Gaussian touches. The workgroup's pairs are counted
// per tile in LDS (gsr_common.h: TileAgg), ONE returning global atomic per (workgroup, tile) reserves
// their slots — all of a workgroup's atomics fly in one round instead of one memory round trip per
// distinct tile and rect step — and the pairs take their positions from LDS cursors.
__global__ void __launch_bounds__(256)
scatter_kernel(int P, int gx, int64_t max_pairs, const int4* __restrict__ rect,
               const float* __restrict__ depth, uint32_t* __restrict__ tile_cursor,
               uint64_t* __restrict__ pair_key, size_t ws_stride) {
  __shared__ TileAgg s_agg;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  {
    const size_t off = (size_t)blockIdx.y * ws_stride;
    rect = reinterpret_cast<const int4*>(reinterpret_cast<const char*>(rect) + off);
    depth = reinterpret_cast<const float*>(reinterpret_cast<const char*>(depth) + off);
    tile_cursor = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tile_cursor) + off);
    pair_key = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(pair_key) + off);
  }
  agg_clear(s_agg);
  const int4 rc = i < P ? rect[i] : make_int4(0, 0, 0, 0);
  const uint64_t key = i < P ? (((uint64_t)__float_as_uint(depth[i]) << 32) | (uint32_t)i) : 0ull;
  const bool big = rect_is_big(rc);                     // walked by the whole wave below (gsr_common.h)
  __syncthreads();
  if (!big)
    for (int cy = rc.y; cy < rc.w; ++cy)
      for (int cx = rc.x; cx < rc.z; ++cx) {
        const int slot = agg_claim(s_agg, cy * gx + cx);
        if (slot >= 0) atomicAdd(&s_agg.cnt[slot], 1u);
      }
  __syncthreads();
  for (int sl = threadIdx.x; sl < GSR_AGG_SLOTS; sl += blockDim.x)
    if (s_agg.key[sl] >= 0) {
      s_agg.base[sl] = atomicAdd(&tile_cursor[s_agg.key[sl]], s_agg.cnt[sl]);
      s_agg.cnt[sl] = 0u;
    }
  __syncthreads();
  if (!big)
    for (int cy = rc.y; cy < rc.w; ++cy)
      for (int cx = rc.x; cx < rc.z; ++cx) {
        const int tile = cy * gx + cx;
        const int slot = agg_find(s_agg, tile);
        // (a tile that found no room in the table takes its slot directly)
        const uint32_t pos = slot >= 0 ? s_agg.base[slot] + atomicAdd(&s_agg.cnt[slot], 1u)
                                       : atomicAdd(&tile_cursor[tile], 1u);
        if ((int64_t)pos < max_pairs) pair_key[pos] = key;
      }
  for_big_rects(rc, gx,
                [&](int src) { return ((uint64_t)__shfl((uint32_t)(key >> 32), src) << 32) | __shfl((uint32_t)key, src); },
                [&](int tile, uint64_t k) {
                  const uint32_t pos = atomicAdd(&tile_cursor[tile], 1u);
                  if ((int64_t)pos < max_pairs) pair_key[pos] = k;
                });
}

// ------------------------------------------------------------------ K4
// Merge-path split: number of elements taken from A among the first `diag` outputs of
// merge(A[0..na), B[0..nb)); ties go to A (stable, although keys are unique here).
__device__ __forceinline__ int merge_split(const uint64_t* A, int na, const uint64_t* B, int nb,
                                           int diag) {
  int lo = max(0, diag - nb), hi = min(diag, na);
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (A[mid] <= B[diag - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Merge sort of n <= 8 NT keys in LDS by NT threads (ping-pong between a and b; returns the buffer that holds the
// result). Every thread owns 8 consecutive OUTPUT positions per level (merge path): a binary search for its split of the
// two runs (log2(run) dependent LDS reads), then 8 sequential merge steps — one LDS round trip per output. Against the
// bitonic network this replaces (62 -> 46 us for the bench scene's lists, 2 frames): n log2(n / 8) key moves instead of
// n log2^2(n) / 2, log2(n / 8) workgroup barriers instead of ~22 for 2048 keys, no padding to a power of two.
// (Measured and rejected: rank-scatter merging — a thread keeps 8 keys and binary-searches each one's rank in the
// sibling run, eight independent searches interleaved: 105 us, the random 8-byte LDS reads conflict on the banks; and
// the 8 outputs of a thread taken from two 8-key register windows with a bitonic half-cleaner + 12 compare-exchanges
// instead of 8 dependent LDS round trips: 50 us against 46 — the u64 network costs more than the round trips.)
__device__ __forceinline__ void cex(uint64_t& x, uint64_t& y) {
  const uint64_t lo = x < y ? x : y, hi = x < y ? y : x;
  x = lo; y = hi;
}
template <int NT>
__device__ __forceinline__ uint64_t* merge_sort_lds(uint64_t* a, uint64_t* b, int n, int tid) {
  constexpr int ITEMS = 8;
  const int g0 = tid * ITEMS;
  if (g0 < n) {       // sorting network on the thread's own 8 keys (19 compare-exchanges; absent keys = +inf)
    uint64_t k[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) k[i] = (g0 + i < n) ? a[g0 + i] : ~0ull;
    cex(k[0], k[1]); cex(k[2], k[3]); cex(k[4], k[5]); cex(k[6], k[7]);
    cex(k[0], k[2]); cex(k[1], k[3]); cex(k[4], k[6]); cex(k[5], k[7]);
    cex(k[1], k[2]); cex(k[5], k[6]); cex(k[0], k[4]); cex(k[3], k[7]);
    cex(k[1], k[5]); cex(k[2], k[6]);
    cex(k[1], k[4]); cex(k[3], k[6]);
    cex(k[2], k[4]); cex(k[3], k[5]);
    cex(k[3], k[4]);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) if (g0 + i < n) a[g0 + i] = k[i];
  }
  __syncthreads();
  uint64_t* src = a;
  uint64_t* dst = b;
  for (int run = ITEMS; run < n; run <<= 1) {
    if (g0 < n) {
      const int lo = (g0 / (2 * run)) * (2 * run);
      const int mid = min(lo + run, n), hi = min(lo + 2 * run, n);
      const uint64_t* A = src + lo;
      const uint64_t* B = src + mid;
      const int na = mid - lo, nb = hi - mid;
      int ia = merge_split(A, na, B, nb, g0 - lo);
      int ib = g0 - lo - ia;
      uint64_t va = ia < na ? A[ia] : ~0ull, vb = ib < nb ? B[ib] : ~0ull;
      const int cnt = min(ITEMS, hi - g0);
#pragma unroll
      for (int o = 0; o < ITEMS; ++o) {
        if (o < cnt) {
          const bool takeA = va <= vb;                // keys are unique and < ~0: an exhausted run never wins
          dst[g0 + o] = takeA ? va : vb;
          if (takeA) { ++ia; va = ia < na ? A[ia] : ~0ull; } else { ++ib; vb = ib < nb ? B[ib] : ~0ull; }
        }
      }
    }
    __syncthreads();
    uint64_t* t = src; src = dst; dst = t;
  }
  return src;
}

// ---- K4 as launched: chunk sorts + merges.
// One workgroup sorting a whole list makes the launch as long as its longest list: a 4096-key list is 30
// dependent LDS round trips (~1 us each), the training scene's densest tiles (~4600 entries) pad to 8192 keys
// — 100 us per 2 frames while the chip idles (sorting only the 64 longest tiles of a frame takes as long as
// sorting all 586). Measured and rejected on the way: two short lists side by side in one workgroup (no gain),
// pairing ALL lists (halves the long lists' thread count: 144 us), two size classes in two launches (144 us).
// So a list is cut into chunks of SORT_CHUNK keys, every chunk is sorted by its own workgroup (16 KiB of LDS:
// several per CU), and a list's sorted runs are merged with merge path inside LDS by one workgroup — two levels cover 4
// chunks (tile_merge_all_kernel); longer lists (none in avatar scenes; centimetre-sized Gaussians early in a from-scratch
// training do produce them) are merged per 8192-key block in LDS and then block against block through HBM by one workgroup. The order is
// the same total order (depth bits, then Gaussian index), whatever the decomposition.
#ifndef GSR_SORT_CHUNK
#define GSR_SORT_CHUNK 2048      // (tools/build_gsr_variant.sh -DGSR_SORT_CHUNK=1024: measured in round 4)
#endif
constexpr int SORT_CHUNK = GSR_SORT_CHUNK;
constexpr int SORT_MAX_CHUNKS = 8192 / SORT_CHUNK;      // the merge launch stages a whole list: <= 8192 keys of LDS per buffer

struct TileSpan { int64_t start; int n; };
__device__ __forceinline__ TileSpan tile_span(const uint32_t* tile_order, const uint32_t* tile_offset, int64_t cap,
                                              int rank) {
  const int tile = (int)tile_order[rank];                   // longest lists first (tile_scan_kernel)
  TileSpan t;
  t.start = min((int64_t)tile_offset[tile], cap);
  t.n = (int)(min((int64_t)tile_offset[tile + 1], cap) - t.start);
  return t;
}

#define GSR_FRAME_PTRS()                                                                                      \
  {                                                                                                           \
    const size_t off = (size_t)blockIdx.y * ws_stride;                                                        \
    tile_order = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(tile_order) + off);          \
    tile_offset = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(tile_offset) + off);        \
    pair_key = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(pair_key) + off);                          \
    pair_tmp = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(pair_tmp) + off);                          \
    point_list = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(point_list) + off);                      \
  }

// The grids are SORT_GRID workgroups per frame (and chunk / pair index) that stride over the ranks of the size
// order (descending size CLASS = floor(log2 n) + 1, tile_scan_kernel; `ordered` = that order exists, T <= 8192)
// and stop at the first list whose class needs no work: ~590 of 4096 tiles are occupied, and one workgroup per tile and
// chunk (32k mostly empty 1024-thread workgroups per frame over the four launches) cost more than the sort.
constexpr int SORT_GRID = 768;
constexpr int CHUNK_WG = SORT_CHUNK / 8;  // chunk sort: 8 keys per thread

// chunk blockIdx.z of the tiles of rank blockIdx.x, + SORT_GRID, ...: merge-sorted in LDS (merge_sort_lds); a
// single-chunk list goes straight to point_list, otherwise the sorted run replaces the chunk in pair_key
__global__ void __launch_bounds__(CHUNK_WG)
tile_sort_chunk_kernel(int T, int ordered, int64_t max_pairs, const uint32_t* __restrict__ tile_order,
                       const uint32_t* __restrict__ tile_offset, uint64_t* __restrict__ pair_key,
                       uint64_t* __restrict__ pair_tmp, uint32_t* __restrict__ point_list, size_t ws_stride) {
  __shared__ uint64_t s_key[2][SORT_CHUNK];
  GSR_FRAME_PTRS();
  const int tid = threadIdx.x;
  for (int rank = blockIdx.x; rank < T; rank += gridDim.x) {
    const TileSpan ts = tile_span(tile_order, tile_offset, max_pairs, rank);
    if (ts.n <= 0) { if (ordered) break; continue; }        // every later list is empty too
    // chunks blockIdx.z, + SORT_MAX_CHUNKS, ...: one pass for the lists the merge launch stages whole (<= MERGE_KEYS keys),
    // every chunk of a longer list too (it used to be sorted from scratch by its merge workgroup)
    for (int c0 = blockIdx.z * SORT_CHUNK; c0 < ts.n; c0 += SORT_MAX_CHUNKS * SORT_CHUNK) {
      const int m = min(SORT_CHUNK, ts.n - c0);
      uint64_t* keys = pair_key + ts.start + c0;
      __syncthreads();                                      // the previous chunk's LDS image is dead
      for (int i = tid; i < m; i += CHUNK_WG) s_key[0][i] = keys[i];
      __syncthreads();
      const uint64_t* sorted = merge_sort_lds<CHUNK_WG>(s_key[0], s_key[1], m, tid);
      if (ts.n <= SORT_CHUNK) {
        for (int i = tid; i < m; i += CHUNK_WG) point_list[ts.start + i] = (uint32_t)sorted[i];
      } else {
        for (int i = tid; i < m; i += CHUNK_WG) keys[i] = sorted[i];
      }
    }
  }
}

// Every merge pass of a list of 2 .. SORT_MAX_CHUNKS sorted chunks in ONE workgroup (it used to be one launch per pass plus
// one for the long lists: three launches of ~5-10 us, most of it launch latency, for the ~100 lists per frame that have
// more than one chunk): the runs are staged in LDS once (2 x 64 KiB, ping-pong), every thread takes 8 consecutive outputs of
// a level (merge path), the last level writes point_list. Lists beyond SORT_MAX_CHUNKS chunks (none in avatar scenes) take
// the whole-list path in the same launch, runs merged through HBM.
constexpr int MERGE_WG = 1024;
constexpr int MERGE_KEYS = SORT_MAX_CHUNKS * SORT_CHUNK;         // 8192 keys staged in LDS at once (64 KiB per buffer; avatar tiles reach ~4600)
static_assert(MERGE_KEYS == 8 * MERGE_WG, "one 8-key window per thread");
// The merge levels W = SORT_CHUNK, 2 SORT_CHUNK, ... of n <= MERGE_KEYS keys held in LDS as sorted chunks (src / dst ping-pong,
// MERGE_WG threads, 8 consecutive outputs per thread and level: merge path). The last level goes to out32 (the low words:
// the Gaussian indices) if given, else to LDS; returns the LDS buffer that holds the result in the second case.
__device__ __forceinline__ uint64_t* lds_merge_levels(uint64_t* src, uint64_t* dst, int n, int tid, uint32_t* out32) {
  const int g0 = tid * 8;
  for (int W = SORT_CHUNK; W < n; W <<= 1) {
    const bool last = 2 * W >= n;                             // this level leaves one run = the sorted list
    if (g0 < n) {
      const int lo = (g0 / (2 * W)) * (2 * W);
      const int mid = min(lo + W, n), hi = min(lo + 2 * W, n);
      const uint64_t* A = src + lo;
      const uint64_t* B = src + mid;
      const int na = mid - lo, nb = hi - mid;
      int ia = merge_split(A, na, B, nb, g0 - lo);
      int ib = g0 - lo - ia;
      const int cnt = min(8, hi - g0);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        if (o < cnt) {
          const bool takeA = (ib >= nb) || (ia < na && A[ia] <= B[ib]);
          const uint64_t v = takeA ? A[ia++] : B[ib++];
          if (last && out32) out32[g0 + o] = (uint32_t)v;
          else dst[g0 + o] = v;
        }
      }
    }
    __syncthreads();
    uint64_t* t = src; src = dst; dst = t;
  }
  return src;
}

// A list beyond MERGE_KEYS keys, its SORT_CHUNK-key chunks sorted (tile_sort_chunk_kernel): every MERGE_KEYS-key block is
// merged in LDS like a short list, then the blocks are merged through HBM (keys <-> pair_tmp ping-pong), all by this one
// workgroup — a thread takes MERGE_ITEMS consecutive outputs of a level: a binary search and MERGE_ITEMS dependent steps on
// global memory, which is why this path is slow (~50 us per 16 k keys) and why avatar-sized Gaussians never take it.
__device__ __forceinline__ void merge_long_list(uint64_t* keys, uint64_t* tmp, uint32_t* out, int n, uint64_t* s_merge, int tid) {
  for (int b0 = 0; b0 < n; b0 += MERGE_KEYS) {
    const int m = min(MERGE_KEYS, n - b0);
    __syncthreads();                                          // the previous block's LDS image is dead
    for (int i = tid; i < m; i += MERGE_WG) s_merge[i] = keys[b0 + i];
    __syncthreads();
    const uint64_t* r = lds_merge_levels(s_merge, s_merge + MERGE_KEYS, m, tid, nullptr);
    for (int i = tid; i < m; i += MERGE_WG) keys[b0 + i] = r[i];
  }
  __syncthreads();                                            // workgroup-scope visibility of the blocks (one CU, shared L1)
  uint64_t* src = keys;
  uint64_t* dst = tmp;
  for (int64_t width = MERGE_KEYS; width < n; width <<= 1) {
    const int nseg = (n + MERGE_ITEMS - 1) / MERGE_ITEMS;
    for (int seg = tid; seg < nseg; seg += MERGE_WG) {
      const int64_t g0 = (int64_t)seg * MERGE_ITEMS;
      const int64_t lo = (g0 / (2 * width)) * (2 * width);
      const int64_t mid = min(lo + width, (int64_t)n);
      const int64_t hi = min(lo + 2 * width, (int64_t)n);
      const uint64_t* A = src + lo;
      const uint64_t* B = src + mid;
      const int na = (int)(mid - lo), nb = (int)(hi - mid);
      const int diag = (int)(g0 - lo);
      int ia = merge_split(A, na, B, nb, diag);
      int ib = diag - ia;
      const int cnt = (int)min((int64_t)MERGE_ITEMS, hi - g0);
      for (int o = 0; o < cnt; ++o) {
        const bool takeA = (ib >= nb) || (ia < na && A[ia] <= B[ib]);
        dst[g0 + o] = takeA ? A[ia++] : B[ib++];
      }
    }
    __syncthreads();   // workgroup-scope visibility of dst (one CU, shared L1)
    uint64_t* t = src; src = dst; dst = t;
  }
  for (int i = tid; i < n; i += MERGE_WG) out[i] = (uint32_t)src[i];
}

__global__ void __launch_bounds__(MERGE_WG)
tile_merge_all_kernel(int T, int ordered, int64_t max_pairs, const uint32_t* __restrict__ tile_order,
                      const uint32_t* __restrict__ tile_offset, uint64_t* __restrict__ pair_key,
                      uint64_t* __restrict__ pair_tmp, uint32_t* __restrict__ point_list, size_t ws_stride) {
  extern __shared__ uint64_t s_merge[];                      // [2][MERGE_KEYS]
  GSR_FRAME_PTRS();
  const int tid = threadIdx.x;
  for (int rank = blockIdx.x; rank < T; rank += gridDim.x) {
    const TileSpan ts = tile_span(tile_order, tile_offset, max_pairs, rank);
    // one run: nothing to merge. The order is by size CLASS (1 + floor(log2 n), tile_scan_kernel): a list of exactly
    // SORT_CHUNK keys shares its class with lists that do need merging, so only a list BELOW that class ends the walk
    if (ts.n <= SORT_CHUNK) { if (ordered && ts.n < SORT_CHUNK) break; continue; }
    __syncthreads();                                          // the previous list's LDS image is dead
    if (ts.n > MERGE_KEYS) {
      merge_long_list(pair_key + ts.start, pair_tmp + ts.start, point_list + ts.start, ts.n, s_merge, tid);
      continue;
    }
    const int n = ts.n;
    for (int i = tid; i < n; i += MERGE_WG) s_merge[i] = pair_key[ts.start + i];
    __syncthreads();
    lds_merge_levels(s_merge, s_merge + MERGE_KEYS, n, tid, point_list + ts.start);
  }
}

}  // namespace

hipError_t launch_binning(const Dims& d, const Workspace& ws, const Batch& bt, hipStream_t stream) {
  {
    ProfScope prof_(K_SCAN, stream);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1, bt.frames), dim3(SCAN_THREADS), 0, stream, d.T, d.max_pairs,
                     ws.tile_count, ws.tile_offset, ws.tile_cursor, ws.status, bt.ws_stride);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (d.P > 0) {
    {
      ProfScope prof_(K_SCATTER, stream);
      hipLaunchKernelGGL(scatter_kernel, dim3((d.P + 255) / 256, bt.frames), dim3(256), 0, stream, d.P, d.gx,
                       d.max_pairs, ws.rect, ws.depth, ws.tile_cursor, ws.pair_key, bt.ws_stride);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    {
      ProfScope prof_(K_SORT, stream);
      const int gx = min(d.T, SORT_GRID);
      const int ordered = tile_order_is_sorted(d.T);                          // tile_scan_kernel: MAXPER
      hipLaunchKernelGGL(tile_sort_chunk_kernel, dim3(gx, bt.frames, SORT_MAX_CHUNKS), dim3(CHUNK_WG), 0, stream, d.T,
                         ordered, d.max_pairs, ws.tile_count, ws.tile_offset, ws.pair_key, ws.pair_tmp, ws.point_list,
                         bt.ws_stride);
      static PerDeviceFlag attr_set;
      constexpr size_t merge_lds = (size_t)2 * MERGE_KEYS * sizeof(uint64_t);
      if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_merge_all_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)merge_lds);
        if (e != hipSuccess) return e;
        attr_set = true;
      }
      hipLaunchKernelGGL(tile_merge_all_kernel, dim3(min(d.T, 256), bt.frames), dim3(MERGE_WG), merge_lds, stream, d.T, ordered,
                         d.max_pairs, ws.tile_count, ws.tile_offset, ws.pair_key, ws.pair_tmp, ws.point_list, bt.ws_stride);
    }
    e = hipGetLastError();
  }
  return e;
}

}  // namespace gsr
