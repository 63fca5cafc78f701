// gsr_binning.hip — K2 tile_scan, K3 scatter, K4 tile_sort.
//
// Replaces the reference's global pipeline (InclusiveSum over Gaussians -> blocking D2H read
// of the pair count -> duplicateWithKeys -> 64-bit global radix sort -> identifyTileRanges;
// SURVEY.md §2.1) with a per-tile one that never leaves the device:
//   histogram (in K1) -> one-block scan over tiles -> atomic append per tile -> per-tile sort
//   of (depth_bits<<32 | gaussian) keys in LDS (2048-key chunks, merged in LDS; lists beyond 8192 keys bucket by bucket).
// The per-tile order is the same total order (depth bits, then Gaussian index; Appendix A.2),
// so the lists are bit-identical to the reference's however the appends interleave.
#include "gsr_common.h"

namespace gsr {

namespace {

constexpr int SCAN_THREADS = 1024;
constexpr int MERGE_ITEMS = 8;        // outputs per thread per merge step
#ifndef GSR_SORT_CHUNK
#define GSR_SORT_CHUNK 2048      // (tools/build_gsr_variant.sh -DGSR_SORT_CHUNK=1024: measured in round 4)
#endif
constexpr int SORT_CHUNK = GSR_SORT_CHUNK;
// lists beyond GSR_SORT_LDS_KEYS keys: sorted chunks -> output buckets (psrs_bucket); a wave's lane owns a chunk there
constexpr int PSRS_MAX_RUNS = 64;
constexpr int PSRS_MAX_KEYS = PSRS_MAX_RUNS * SORT_CHUNK;

// ------------------------------------------------------------------ K2
__global__ void __launch_bounds__(SCAN_THREADS)
tile_scan_kernel(int T, int64_t max_pairs, uint32_t* __restrict__ tile_count,
                 uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ tile_cursor,
                 int32_t* __restrict__ status, uint32_t* __restrict__ sort_work, int work_cap, size_t ws_stride) {
  {
    const size_t off = (size_t)blockIdx.y * ws_stride;   // batched launch: this frame's workspace
    sort_work = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(sort_work) + off);
    tile_count = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tile_count) + off);
    tile_offset = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tile_offset) + off);
    tile_cursor = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tile_cursor) + off);
    status = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(status) + off);
  }
  __shared__ uint32_t s_wave[SCAN_THREADS / GSR_WAVE];
  __shared__ uint32_t s_max[SCAN_THREADS / GSR_WAVE];
  __shared__ uint32_t s_nwork;
  const int tid = threadIdx.x;
  if (tid == 0) s_nwork = 0u;
  const int per = (T + SCAN_THREADS - 1) / SCAN_THREADS;
  const int lo = tid * per;
  const int hi = min(lo + per, T);
  // a thread's counts stay in registers when it owns at most MAXPER tiles (T <= 8192: every frame up to 2048 x 1024):
  // the kernel is a chain of memory round trips, and the four passes over the histogram used to pay one each
  constexpr int MAXPER = 8, NCLS = 34;              // MAXPER x SCAN_THREADS = 8192: gsr_common.h tile_order_is_sorted
  const bool ordered = per <= MAXPER;
  uint32_t cnt_[MAXPER];
#pragma unroll
  for (int k = 0; k < MAXPER; ++k) cnt_[k] = (ordered && lo + k < hi) ? tile_count[lo + k] : 0u;
  uint32_t sum = 0, mx = 0;
  if (ordered) {
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) { sum += cnt_[k]; mx = max(mx, cnt_[k]); }
  } else {
    for (int t = lo; t < hi; ++t) {
      const uint32_t c = tile_count[t];
      sum += c;
      mx = max(mx, c);
    }
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
  auto emit = [&](int t, uint32_t c) {
    tile_offset[t] = run;
    tile_cursor[t] = run;
    // a list beyond the merge launch's LDS capacity is sorted bucket by bucket (tile_merge_all_kernel: psrs_bucket):
    // one work item per output bucket
    const int64_t n = min((int64_t)run + c, max_pairs) - min((int64_t)run, max_pairs);
    if (n > GSR_SORT_LDS_KEYS && n <= PSRS_MAX_KEYS) {
      const uint32_t nb = (uint32_t)((n + GSR_SORT_BUCKET - 1) / GSR_SORT_BUCKET);
      const uint32_t w0 = atomicAdd(&s_nwork, nb);
      for (uint32_t b = 0; b < nb; ++b)
        if (w0 + b < (uint32_t)work_cap) sort_work[w0 + b] = ((uint32_t)t << 8) | b;
    }
    run += c;
  };
  if (ordered) {
#pragma unroll
    for (int k = 0; k < MAXPER; ++k)
      if (lo + k < hi) emit(lo + k, cnt_[k]);
  } else {
    for (int t = lo; t < hi; ++t) emit(t, tile_count[t]);
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
  __shared__ uint32_t s_cls[NCLS];
  uint32_t pos[MAXPER];
  if (tid < NCLS) s_cls[tid] = 0;
  __syncthreads();
  auto cls_of = [](uint32_t c) { return c ? 32 - __clz(c) : 0; };   // 0 for empty, else 1..32
  // the class is that of the list AS THE LATER KERNELS SEE IT — cut off at the pair buffer's capacity: they walk this
  // order and stop at the first list too short to need them, so a tile whose pairs fell beyond an overflowed buffer
  // (length 0 for them, whatever its count) must sort behind every list that is still there. (Rounds 1-5 ordered by
  // the raw counts: after a heavy overflow the walks ended early and left point_list unwritten — a device fault in
  // render_fwd, found in round 6.)
  auto capped = [max_pairs](uint32_t start, uint32_t c) {
    return (uint32_t)(min((int64_t)start + c, max_pairs) - min((int64_t)start, max_pairs));
  };
  // most tiles are empty (class 0): those are counted / placed once per wave (ballot), the rest per tile
  auto place = [&](int c, bool on) -> uint32_t {
    const unsigned long long empt = __ballot(on && c == 0);
    uint32_t base = 0;
    if (empt && lane == __ffsll((long long)empt) - 1) base = atomicAdd(&s_cls[0], (uint32_t)__popcll(empt));
    base = __shfl(base, empt ? __ffsll((long long)empt) - 1 : 0);
    if (on && c == 0) return base + (uint32_t)__popcll(empt & ((1ull << lane) - 1ull));
    return on ? atomicAdd(&s_cls[c], 1u) : 0u;
  };
  const uint32_t run0 = base + incl - sum;
  if (ordered) {
    uint32_t r = run0;
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) {
      if (k < per) {                                     // uniform trip count: the ballots need every lane
        const int t = lo + k;
        place(t < hi ? cls_of(capped(r, cnt_[k])) : -1, t < hi);
        r += cnt_[k];
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    status[5] = (int32_t)min(s_nwork, (uint32_t)work_cap);
    uint32_t run2 = 0;
    for (int c = NCLS - 1; c >= 0; --c) { const uint32_t v = s_cls[c]; s_cls[c] = run2; run2 += v; }
    // ranks [0, status[6]) of the size order: the lists of SORT_CHUNK keys or more (only they can have a second chunk);
    // [0, status[7]): those of 2 SORT_CHUNK or more (a third, a fourth chunk) — the start of the class below in the order
    constexpr int CHUNK_CLS = 32 - __builtin_clz((unsigned)SORT_CHUNK);          // class of a list of exactly SORT_CHUNK keys
    status[6] = ordered ? (int32_t)s_cls[CHUNK_CLS - 1] : T;
    status[7] = ordered ? (int32_t)s_cls[CHUNK_CLS] : T;
  }
  __syncthreads();
  if (ordered) {
    uint32_t r = run0;
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) {
      if (k < per) {
        const int t = lo + k;
        pos[k] = place(t < hi ? cls_of(capped(r, cnt_[k])) : -1, t < hi);
        r += cnt_[k];
      }
    }
  }
  __syncthreads();                    // every count has been read: the buffer may be overwritten
  if (ordered) {
#pragma unroll
    for (int k = 0; k < MAXPER; ++k)
      if (lo + k < hi) tile_count[pos[k]] = (uint32_t)(lo + k);
  } else {
    for (int t = lo; t < hi; ++t) tile_count[t] = (uint32_t)t;
  }
}

// ------------------------------------------------------------------ K3
// Append (depth_bits << 32 | index) to every tile a Gaussian touches. The workgroup's pairs are counted per cell of its
// tile window in LDS (gsr_common.h: TileWin — the same difference array + prefix sum as K1's histogram), ONE returning
// global atomic per (workgroup, tile) reserves their slots — all of a workgroup's atomics fly in one round — and the
// pairs take their positions from the cells, which have become cursors. The pairs of a wave's 64 Gaussians are DEALT to
// its lanes (pair p of the wave -> lane p % 64: a 6-step search in the wave's prefix of rectangle sizes finds its
// Gaussian) instead of every lane walking its own rectangle: the wave's time is its pair count / 64, not its largest
// rectangle (scatter 26 -> 1,313 us between 4 and 20 tiles per Gaussian before, profiles/r05_dsweep.txt).
__global__ void __launch_bounds__(256)
scatter_kernel(int P, int gx, int64_t max_pairs, const int4* __restrict__ rect,
               const float* __restrict__ depth, uint32_t* __restrict__ tile_cursor,
               uint64_t* __restrict__ pair_key, size_t ws_stride, int win_cells) {
  extern __shared__ int s_dyn[];
  __shared__ int s_box[4];
  __shared__ int4 s_rect[256];
  __shared__ uint64_t s_key[256];
  __shared__ int s_incl[4][GSR_WAVE];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  {
    const size_t off = (size_t)blockIdx.y * ws_stride;
    rect = reinterpret_cast<const int4*>(reinterpret_cast<const char*>(rect) + off);
    depth = reinterpret_cast<const float*>(reinterpret_cast<const char*>(depth) + off);
    tile_cursor = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tile_cursor) + off);
    pair_key = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(pair_key) + off);
  }
  const int4 rc = i < P ? rect[i] : make_int4(0, 0, 0, 0);
  const uint64_t key = i < P ? (((uint64_t)__float_as_uint(depth[i]) << 32) | (uint32_t)i) : 0ull;
  const int ntile = (rc.z - rc.x) * (rc.w - rc.y);
  const TileWin wn = wg_tile_window(rc, ntile > 0, s_box, win_cells);
  if (wn.w == 0) return;
  if (wn.w > 0) {
    const int ncell = wn.w * wn.h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < ncell; e += blockDim.x) s_dyn[e] = 0;
    s_rect[threadIdx.x] = rc;
    s_key[threadIdx.x] = key;
    // inclusive prefix of the rectangle sizes over the wave
    int incl = ntile > 0 ? ntile : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    s_incl[wave][lane] = incl;
    const int total = __shfl(incl, 63);
    __syncthreads();
    if (ntile > 0) win_mark(s_dyn, wn, rc);
    __syncthreads();
    win_prefix(s_dyn, wn);
    for (int e = threadIdx.x; e < ncell; e += blockDim.x) {
      const int c = s_dyn[e];
      if (c > 0) s_dyn[e] = (int)atomicAdd(&tile_cursor[(wn.y0 + e / wn.w) * gx + wn.x0 + e % wn.w], (uint32_t)c);
    }
    __syncthreads();
    const int* incl_w = s_incl[wave];
    for (int p = lane; p < total; p += GSR_WAVE) {
      int j = 0;                                              // first Gaussian of the wave whose inclusive prefix exceeds p
#pragma unroll
      for (int step = 32; step > 0; step >>= 1)
        if (incl_w[j + step - 1] <= p) j += step;
      const int4 r = s_rect[wave * GSR_WAVE + j];
      const int rw = r.z - r.x;
      const int k = p - (incl_w[j] - rw * (r.w - r.y));       // index of the pair inside the rectangle, row-major
      const int ry = k / rw;
      const int cell = (r.y + ry - wn.y0) * wn.w + (r.x + (k - ry * rw) - wn.x0);
      const uint32_t pos = (uint32_t)atomicAdd(&s_dyn[cell], 1);
      if ((int64_t)pos < max_pairs) pair_key[pos] = s_key[wave * GSR_WAVE + j];
    }
    return;
  }
  // the window does not fit (more than GSR_WIN_CELLS tiles in the frame): rounds 1-5's hash-table path
  TileAgg& s_agg = *reinterpret_cast<TileAgg*>(s_dyn);
  agg_clear(s_agg);
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
template <class KA>
__device__ __forceinline__ int merge_split(const KA& A, int na, const KA& B, int nb, int diag) {
  int lo = max(0, diag - nb), hi = min(diag, na);
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (A[mid] <= B[diag - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// LDS key arrays are addressed through a swizzle (round 6). Every thread of the merge levels owns 8 consecutive output
// positions: with keys stored in index order a wave's `dst[8 t + o]` (ds_write_b64: 16-lane groups over 32 banks) puts 8 lanes on
// one bank pair — an 8-way conflict on every one of the 64 key writes of a 2048-key chunk sort, and the same on the window
// reads (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.77, profiles/r06_pmc_raster_*.txt). Slot of key j: j ^ ((j >> 4) & 7) —
// a permutation inside every aligned block of 8 keys that gives the 16 lanes of a write group 16 different slots of the
// 128-byte bank row (reads of a window: 2-way); conflict cycles of the chunk sort 6.4 M -> 1.3 M per launch. Arrays hold a
// multiple of 8 slots; LKeys is a run inside such an array. Used by the chunk sort (merge_sort_lds) only, see PKeys.
__device__ __forceinline__ int ksl(int j) { return j ^ ((j >> 4) & 7); }
struct LKeys {
  uint64_t* buf; int off;
  __device__ __forceinline__ uint64_t operator[](int i) const { return buf[ksl(off + i)]; }
  __device__ __forceinline__ void set(int i, uint64_t v) const { buf[ksl(off + i)] = v; }
  __device__ __forceinline__ LKeys operator+(int d) const { return LKeys{buf, off + d}; }
};
// the same interface over keys in index order: the merge launch's levels (one 1024-thread workgroup per CU: a pure
// latency chain — the swizzle's three extra address instructions per dependent read made it slower, 162 -> 179 us at 5 M
// pairs per frame, while the chunk sort, five workgroups per CU on one LDS pipe, gained: 178 -> 147 us)
struct PKeys {
  uint64_t* buf; int off;
  __device__ __forceinline__ uint64_t operator[](int i) const { return buf[off + i]; }
  __device__ __forceinline__ void set(int i, uint64_t v) const { buf[off + i] = v; }
  __device__ __forceinline__ PKeys operator+(int d) const { return PKeys{buf, off + d}; }
};

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
#ifdef GSR_SORT_TRACE
#define SORT_TRACE_CHUNKS (8192 / GSR_SORT_CHUNK)
// development: s_memrealtime stamps (100 MHz) of the chunk sort's phases, workgroup (0, 0, 0), thread 0
__device__ unsigned long long g_sort_trace[32];
#define SORT_STAMP(I) do { if (tid == 0 && blockIdx.x == GSR_SORT_TRACE && blockIdx.y == 0 && blockIdx.z == gridDim.z - 1) g_sort_trace[I] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SORT_STAMP(I) do {} while (0)
#endif
template <int NT>
__device__ __forceinline__ LKeys merge_sort_lds(uint64_t* a_, uint64_t* b_, int n, int tid) {
  constexpr int ITEMS = 8;
  const LKeys a{a_, 0}, b{b_, 0};
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
    for (int i = 0; i < ITEMS; ++i) if (g0 + i < n) a.set(g0 + i, k[i]);
  }
  __syncthreads();
  SORT_STAMP(3);
  LKeys src = a, dst = b;
  int lvl_ = 0;
  for (int run = ITEMS; run < n; run <<= 1, ++lvl_) {
    if (g0 < n) {
      const int lo = (g0 / (2 * run)) * (2 * run);
      const int mid = min(lo + run, n), hi = min(lo + 2 * run, n);
      const LKeys A = src + lo;
      const LKeys B = src + mid;
      const int na = mid - lo, nb = hi - mid;
      int ia = merge_split(A, na, B, nb, g0 - lo);
      int ib = g0 - lo - ia;
      uint64_t va = ia < na ? A[ia] : ~0ull, vb = ib < nb ? B[ib] : ~0ull;
      const int cnt = min(ITEMS, hi - g0);
#pragma unroll
      for (int o = 0; o < ITEMS; ++o) {
        if (o < cnt) {
          const bool takeA = va <= vb;                // keys are unique and < ~0: an exhausted run never wins
          dst.set(g0 + o, takeA ? va : vb);
          if (takeA) { ++ia; va = ia < na ? A[ia] : ~0ull; } else { ++ib; vb = ib < nb ? B[ib] : ~0ull; }
        }
      }
    }
    __syncthreads();
    SORT_STAMP(4 + lvl_);
    const LKeys t = src; src = dst; dst = t;
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
#ifndef GSR_SORT_GRID
#define GSR_SORT_GRID 768
#endif
#ifndef GSR_MERGE_GRID
#define GSR_MERGE_GRID 256
#endif
constexpr int SORT_GRID = GSR_SORT_GRID;
constexpr int CHUNK_WG = SORT_CHUNK / 8;  // chunk sort: 8 keys per thread

// chunk (gridDim.z - 1 - blockIdx.z) of the tiles of rank blockIdx.x, + SORT_GRID, ...: merge-sorted in LDS (merge_sort_lds); a
// single-chunk list goes straight to point_list, otherwise the sorted run replaces the chunk in pair_key
__global__ void __launch_bounds__(CHUNK_WG)
tile_sort_chunk_kernel(int T, int ordered, int64_t max_pairs, const uint32_t* __restrict__ tile_order,
                       const uint32_t* __restrict__ tile_offset, uint64_t* __restrict__ pair_key,
                       uint64_t* __restrict__ pair_tmp, uint32_t* __restrict__ point_list,
                       const int32_t* __restrict__ status, size_t ws_stride) {
  __shared__ uint64_t s_key[2][SORT_CHUNK];
  GSR_FRAME_PTRS();
  status = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(status) + (size_t)blockIdx.y * ws_stride);
  const int tid = threadIdx.x;
  SORT_STAMP(0);
  // blockIdx.z = chunk, dispatched LAST CHUNK FIRST (a launch's workgroups start in x, y, z order: with the first chunks in
  // front, the later chunks of the long lists — 17 us each, tools/sort_trace.py — started only when the first round's
  // workgroups left their LDS and the launch took two rounds for one round of work). Only the first status[6] / status[7]
  // ranks of the size order have a second / third chunk (tile_scan_kernel): every other workgroup of those slices leaves
  // after one load.
  const int chunk0 = (int)(gridDim.z - 1 - blockIdx.z);
  const int rank_end = chunk0 == 0 ? T : min(T, (int)status[chunk0 == 1 ? 6 : 7]);
  for (int rank = blockIdx.x; rank < rank_end; rank += gridDim.x) {
    const TileSpan ts = tile_span(tile_order, tile_offset, max_pairs, rank);
    if (ts.n <= 0) { if (ordered) break; continue; }        // every later list is empty too
    SORT_STAMP(1);
    // chunks chunk0, + SORT_MAX_CHUNKS, ...: one pass for the lists the merge launch stages whole (<= MERGE_KEYS keys),
    // every chunk of a longer list too (it used to be sorted from scratch by its merge workgroup)
    for (int c0 = chunk0 * SORT_CHUNK; c0 < ts.n; c0 += SORT_MAX_CHUNKS * SORT_CHUNK) {
      const int m = min(SORT_CHUNK, ts.n - c0);
      uint64_t* keys = pair_key + ts.start + c0;
      __syncthreads();                                      // the previous chunk's LDS image is dead
      for (int i = tid; i < m; i += CHUNK_WG) s_key[0][ksl(i)] = keys[i];
      __syncthreads();
      SORT_STAMP(2);
      const LKeys sorted = merge_sort_lds<CHUNK_WG>(s_key[0], s_key[1], m, tid);
      if (tid == 0) { SORT_STAMP(20); }
#ifdef GSR_SORT_TRACE
      if (tid == 0 && blockIdx.x == GSR_SORT_TRACE && blockIdx.y == 0 && blockIdx.z == gridDim.z - 1) g_sort_trace[30] = (unsigned long long)m;
#endif
      if (ts.n <= SORT_CHUNK) {
        for (int i = tid; i < m; i += CHUNK_WG) point_list[ts.start + i] = (uint32_t)sorted[i];
      } else {
        for (int i = tid; i < m; i += CHUNK_WG) keys[i] = sorted[i];
      }
      SORT_STAMP(21);
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
__device__ __forceinline__ PKeys lds_merge_levels(uint64_t* src_, uint64_t* dst_, int n, int tid, uint32_t* out32) {
  PKeys src{src_, 0}, dst{dst_, 0};
  const int g0 = tid * 8;
  for (int W = SORT_CHUNK; W < n; W <<= 1) {
    const bool last = 2 * W >= n;                             // this level leaves one run = the sorted list
    if (g0 < n) {
      const int lo = (g0 / (2 * W)) * (2 * W);
      const int mid = min(lo + W, n), hi = min(lo + 2 * W, n);
      const PKeys A = src + lo;
      const PKeys B = src + mid;
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
          else dst.set(g0 + o, v);
        }
      }
    }
    __syncthreads();
    const PKeys t = src; src = dst; dst = t;
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
    const PKeys r = lds_merge_levels(s_merge, s_merge + MERGE_KEYS, m, tid, nullptr);
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

// ---- lists beyond MERGE_KEYS keys (round 6): sorted chunks -> independent output buckets.
// Centimetre-sized Gaussians (the first few hundred iterations of a from-scratch training: the reference's scale warm-up)
// put 10-60 k keys on a tile. Rounds 1-5 merged such a list level by level through HBM in ONE workgroup (2.5 ms of tile
// sort at 4 M pairs per frame, profiles/r05_dsweep.txt). Now every ~4096-key piece of the OUTPUT is its own work item
// (tile_scan_kernel lists them): parallel sorting by regular sampling over the list's sorted chunks ("runs").
//   1. samples: the last key of every group of g consecutive keys of every run (g = psrs_group(runs): runs are
//      SORT_CHUNK-aligned and g divides SORT_CHUNK, so sample q is list position (q + 1) g - 1), S = n / g of them, sorted
//      in LDS; splitter q of p = ceil(n / GSR_SORT_BUCKET) buckets = the sample of rank floor(q S / p)
//   2. a run's keys below a splitter: between g t and g (t + 1) where t = its samples below the splitter, so the rank of
//      splitter q lies in [g r_q, g r_q + runs g) and a bucket holds < n / p + (runs + 1) g + 1 <= MERGE_KEYS keys:
//      it always fits the LDS buffer — by construction, not by luck (psrs_group)
//   3. the bucket's piece of every run by a wave-wide two-round search (64 probes, then 32 per splitter), gathered into
//      LDS back to back, merged pairwise in log2(runs) levels (lds_merge_runs), written to its place in point_list
// Every workgroup recomputes the splitters it needs from the samples (<= 4096 keys, usually a few hundred): no
// dependency between work items, no extra launch, no grid-wide barrier. Same total order as every other path.
static_assert(SORT_CHUNK <= 2048 && (SORT_CHUNK & (SORT_CHUNK - 1)) == 0, "psrs_bucket: a run is searched as <= 64 groups of 32");
__host__ __device__ inline int psrs_group(int runs) {
  const int lim = GSR_SORT_BUCKET / (runs + 2);      // (runs + 1) g < GSR_SORT_BUCKET
  int g = 512;
  while (g > lim) g >>= 1;
  return g;
}

// pairwise merge, level by level, of the nruns sorted runs [bnd[r], bnd[r + 1]) of src (LDS; runs may be empty) until one
// is left; MERGE_WG threads, 8 consecutive outputs per thread and level (merge path); the last level writes the low
// words to out32. nruns >= 2.
__device__ __forceinline__ void lds_merge_runs(uint64_t* src_, uint64_t* dst_, const int* bnd, int nruns, int m, int tid,
                                               uint32_t* out32) {
  PKeys src{src_, 0}, dst{dst_, 0};
  const int g0 = tid * 8;
  for (int stride = 1; stride < nruns; stride <<= 1) {
    const bool last = 2 * stride >= nruns;
    if (g0 < m) {
      // the pair of runs this thread's first output lies in: the largest u with bnd[2 u stride] <= g0
      const int npairs = (nruns + 2 * stride - 1) / (2 * stride);
      int u = 0;
      for (int lo = 0, hi = npairs - 1; ; ) {
        if (lo >= hi) { u = lo; break; }
        const int mid = (lo + hi + 1) >> 1;
        if (bnd[2 * mid * stride] <= g0) lo = mid; else hi = mid - 1;
      }
      int o = 0;
      while (o < 8 && g0 + o < m) {
        const int l0 = bnd[2 * u * stride], l1 = bnd[min(nruns, 2 * u * stride + stride)],
                  l2 = bnd[min(nruns, 2 * (u + 1) * stride)];
        const PKeys A = src + l0;
        const PKeys B = src + l1;
        const int na = l1 - l0, nb = l2 - l1, diag = g0 + o - l0;
        int ia = merge_split(A, na, B, nb, diag);
        int ib = diag - ia;
        const int cnt = min(8 - o, l2 - (g0 + o));
        for (int c = 0; c < cnt; ++c) {
          const bool takeA = (ib >= nb) || (ia < na && A[ia] <= B[ib]);
          const uint64_t v = takeA ? A[ia++] : B[ib++];
          if (last) out32[g0 + o + c] = (uint32_t)v;
          else dst.set(g0 + o + c, v);
        }
        o += cnt;
        ++u;                                                  // the next pair starts where this one ends
      }
    }
    __syncthreads();
    const PKeys t = src; src = dst; dst = t;
  }
}

// bucket b of the list keys[0, n) (its SORT_CHUNK-key chunks sorted), MERGE_KEYS < n <= PSRS_MAX_KEYS -> out[...]
__device__ __forceinline__ void psrs_bucket(const uint64_t* keys, uint32_t* out, int n, int b, uint64_t* s_a, uint64_t* s_b,
                                            int tid) {
  __shared__ int s_lo[PSRS_MAX_RUNS], s_hi[PSRS_MAX_RUNS], s_bnd[PSRS_MAX_RUNS + 1], s_first;
  const int lane = tid & (GSR_WAVE - 1), wave = tid / GSR_WAVE;
  const int runs = (n + SORT_CHUNK - 1) / SORT_CHUNK;
  const int g = psrs_group(runs), S = n / g, p = (n + GSR_SORT_BUCKET - 1) / GSR_SORT_BUCKET;
  __syncthreads();                                            // the previous item's LDS image is dead
  for (int i = tid; i < S; i += MERGE_WG) s_a[ksl(i)] = keys[(int64_t)(i + 1) * g - 1];
  __syncthreads();
  const LKeys ss = merge_sort_lds<MERGE_WG>(s_a, s_b, S, tid);
  const uint64_t klo = b > 0 ? ss[(int)((int64_t)b * S / p)] : 0ull;
  const uint64_t khi = b + 1 < p ? ss[(int)((int64_t)(b + 1) * S / p)] : ~0ull;
  __syncthreads();                                            // every thread holds the splitters: the buffers are free
  // keys of run j below each splitter: round 1 probes the last key of every group of 32 (the groups entirely below the
  // splitter), round 2 the 32 keys of the group the splitter falls in — lanes 0..31 for klo, 32..63 for khi
  for (int j = wave; j < runs; j += MERGE_WG / GSR_WAVE) {
    const uint64_t* run = keys + (int64_t)j * SORT_CHUNK;
    const int L = min(SORT_CHUNK, n - j * SORT_CHUNK);
    const uint64_t v = lane * 32 < L ? run[min(lane * 32 + 31, L - 1)] : ~0ull;
    const int c_lo = __popcll(__ballot(v < klo)), c_hi = __popcll(__ballot(v < khi));
    const int pos = (lane < 32 ? c_lo : c_hi) * 32 + (lane & 31);
    const uint64_t v2 = pos < L ? run[pos] : ~0ull;
    const unsigned long long m2 = __ballot(v2 < (lane < 32 ? klo : khi));
    if (lane == 0) {
      s_lo[j] = min(L, c_lo * 32 + __popc((unsigned)m2));
      s_hi[j] = min(L, c_hi * 32 + __popc((unsigned)(m2 >> 32)));
    }
  }
  __syncthreads();
  if (tid < GSR_WAVE) {                                       // runs <= 64: one lane per run
    int len = tid < runs ? s_hi[tid] - s_lo[tid] : 0, first = tid < runs ? s_lo[tid] : 0;
#pragma unroll
    for (int off = 1; off < GSR_WAVE; off <<= 1) {
      const int t = __shfl_up(len, off);
      if (lane >= off) len += t;
    }
#pragma unroll
    for (int off = GSR_WAVE / 2; off > 0; off >>= 1) first += __shfl_xor(first, off);
    if (tid < runs) s_bnd[tid + 1] = len;
    if (tid == 0) { s_bnd[0] = 0; s_first = first; }
  }
  __syncthreads();
  const int m = s_bnd[runs];
  if (m > MERGE_KEYS) __builtin_trap();                       // (excluded by psrs_group's bound)
  for (int i = tid; i < m; i += MERGE_WG) {
    int r = 0;                                                // the run position i of the bucket comes from
#pragma unroll
    for (int step = PSRS_MAX_RUNS / 2; step > 0; step >>= 1)
      if (r + step <= runs && s_bnd[r + step] <= i) r += step;
    s_a[i] = keys[(int64_t)r * SORT_CHUNK + s_lo[r] + (i - s_bnd[r])];
  }
  __syncthreads();
  lds_merge_runs(s_a, s_b, s_bnd, runs, m, tid, out + s_first);
}

__global__ void __launch_bounds__(MERGE_WG)
tile_merge_all_kernel(int T, int ordered, int64_t max_pairs, const uint32_t* __restrict__ tile_order,
                      const uint32_t* __restrict__ tile_offset, uint64_t* __restrict__ pair_key,
                      uint64_t* __restrict__ pair_tmp, uint32_t* __restrict__ point_list,
                      const uint32_t* __restrict__ sort_work, const int32_t* __restrict__ status, size_t ws_stride) {
  extern __shared__ uint64_t s_merge[];                      // [2][MERGE_KEYS]
  GSR_FRAME_PTRS();
  {
    const size_t off = (size_t)blockIdx.y * ws_stride;
    sort_work = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(sort_work) + off);
    status = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(status) + off);
  }
  const int tid = threadIdx.x;
  // the long lists' output buckets first (tile_scan_kernel's work list): the launch's longest units
  const int nwork = status[5];
  for (int it = blockIdx.x; it < nwork; it += gridDim.x) {
    const uint32_t w = sort_work[it];
    const int tile = (int)(w >> 8);
    const int64_t start = min((int64_t)tile_offset[tile], max_pairs);
    const int n = (int)(min((int64_t)tile_offset[tile + 1], max_pairs) - start);
    psrs_bucket(pair_key + start, point_list + start, n, (int)(w & 255u), s_merge, s_merge + MERGE_KEYS, tid);
  }
  for (int rank = blockIdx.x; rank < T; rank += gridDim.x) {
    const TileSpan ts = tile_span(tile_order, tile_offset, max_pairs, rank);
    // one run: nothing to merge. The order is by size CLASS (1 + floor(log2 n), tile_scan_kernel): a list of exactly
    // SORT_CHUNK keys shares its class with lists that do need merging, so only a list BELOW that class ends the walk
    if (ts.n <= SORT_CHUNK) { if (ordered && ts.n < SORT_CHUNK) break; continue; }
    if (ts.n > MERGE_KEYS) {
      if (ts.n <= PSRS_MAX_KEYS) continue;                    // done bucket by bucket above
      __syncthreads();
      merge_long_list(pair_key + ts.start, pair_tmp + ts.start, point_list + ts.start, ts.n, s_merge, tid);
      continue;
    }
    __syncthreads();                                          // the previous list's LDS image is dead
    const int n = ts.n;
    for (int i = tid; i < n; i += MERGE_WG) s_merge[i] = pair_key[ts.start + i];
    __syncthreads();
    lds_merge_levels(s_merge, s_merge + MERGE_KEYS, n, tid, point_list + ts.start);
  }
}

}  // namespace

#ifdef GSR_SORT_TRACE
extern "C" int gsr_dev_sort_trace(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sort_trace), sizeof(g_sort_trace)); }
#endif

hipError_t launch_binning(const Dims& d, const Workspace& ws, const Batch& bt, hipStream_t stream) {
  {
    ProfScope prof_(K_SCAN, stream);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1, bt.frames), dim3(SCAN_THREADS), 0, stream, d.T, d.max_pairs,
                     ws.tile_count, ws.tile_offset, ws.tile_cursor, ws.status, ws.sort_work,
                     (int)sort_work_capacity(d.max_pairs), bt.ws_stride);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  trace_sync(stream, "tile_scan");
  if (d.P > 0) {
    {
      ProfScope prof_(K_SCATTER, stream);
      hipLaunchKernelGGL(scatter_kernel, dim3((d.P + 255) / 256, bt.frames), dim3(256), win_lds_bytes(d.T), stream, d.P, d.gx,
                       d.max_pairs, ws.rect, ws.depth, ws.tile_cursor, ws.pair_key, bt.ws_stride,
                       d.T < GSR_WIN_CELLS ? d.T : GSR_WIN_CELLS);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    trace_sync(stream, "scatter");
    {
      ProfScope prof_(K_SORT, stream);
      const int gx = min(d.T, SORT_GRID);
      const int ordered = tile_order_is_sorted(d.T);                          // tile_scan_kernel: MAXPER
      hipLaunchKernelGGL(tile_sort_chunk_kernel, dim3(gx, bt.frames, SORT_MAX_CHUNKS), dim3(CHUNK_WG), 0, stream, d.T,
                         ordered, d.max_pairs, ws.tile_count, ws.tile_offset, ws.pair_key, ws.pair_tmp, ws.point_list,
                         ws.status, bt.ws_stride);
      trace_sync(stream, "tile_sort_chunk");
      static PerDeviceFlag attr_set;
      constexpr size_t merge_lds = (size_t)2 * MERGE_KEYS * sizeof(uint64_t);
      if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_merge_all_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)merge_lds);
        if (e != hipSuccess) return e;
        attr_set = true;
      }
      hipLaunchKernelGGL(tile_merge_all_kernel, dim3(min(d.T, GSR_MERGE_GRID), bt.frames), dim3(MERGE_WG), merge_lds, stream, d.T, ordered,
                         d.max_pairs, ws.tile_count, ws.tile_offset, ws.pair_key, ws.pair_tmp, ws.point_list, ws.sort_work,
                         ws.status, bt.ws_stride);
    }
    e = hipGetLastError();
  }
  return e;
}

}  // namespace gsr
