// gsr_render.hip — K5 render_fwd and K6 render_bwd (per-tile alpha compositing).
//
// Shape of the problem on MI355X (measured, 200k avatar Gaussians at 1024^2, per frame): ~590 of 4096
// tiles are occupied, their depth-sorted lists hold ~1000 (up to ~3600) entries; a Gaussian reaches
// alpha >= 1/255 on ~90 pixels, so only ~28 % of a tile's list can touch a given 4x4 pixel block, and of
// the (pixel, entry) slots of those survivors ~36 % really blend. One lane per pixel (the textbook
// mapping) leaves the chip idle and serialises every pixel's chain; round 1's quad-per-pixel kernel paid
// ~170 instructions + an LDS round trip per 4 entries to sum gradients over pixels. This version turns
// the mapping around:
//
//   * ENTRY-parallel, PIXEL-serial. A wave owns a 4x4 pixel block (GSR_SUB; 8x8: see "Block size" below). It culls the tile's list against the
//     block (exact-conservative alpha_min-ellipse box, ballot + mbcnt compaction into a per-wave LDS
//     ring) and consumes the survivors in SEGMENTS of 64: lane l holds survivor l, and the wave walks the
//     block's 16 pixels. For one pixel, transmittance in front of each of the 64 entries is an
//     exclusive prefix product over the lanes — ONE 6-step DPP wave scan (row_shr 1/2/4/8, row_bcast
//     15/31) — the stop test is a ballot, colour sums are wave reductions. Per-pixel state (T, C, last
//     contributor) lives in lane p of a few registers (v_readlane with the pixel index in an SGPR).
//   * the forward pass RECORDS what it consumed: per segment the 64 (Gaussian, list position) pairs and
//     the 16 pixels' (T, C) at its start — 776 bytes, coalesced — plus the pixels' accumulated colour.
//   * the backward pass is SEGMENT-parallel: one wave per recorded segment, no culling, no list walk,
//     perfectly balanced units (the longest chain of the launch is 16 pixels), in FORWARD order:
//       dL/dalpha_k = T_k (c_k.g) - [ (C_total - C_<=k).g + T_final (bg.g) ] / (1 - alpha_k)
//     needs only prefix quantities (T_k: product scan; C_<=k.g: ONE sum scan of the scalar (c_j.g) w_j),
//     not the back-to-front recurrence. A lane sums ITS entry's 9 gradient components over the pixels in
//     registers — no cross-lane reduction at all — and issues 9 atomics at the end of the segment.
// Results are those of the sequential definition up to float association (prefix products / sums are
// evaluated as trees): covered by the image / gradient tolerances of tests/test_raster_gpu.py.
//
// Block size (round 4: 8x8 blocks built, measured, not kept; GSR_SUB selects it). The backward pass ends every segment
// with one 64-byte atomic gradient record per entry, and gfx950 retires those at 18.3 G records/s whatever the access
// pattern (tools/ubench/atomic_rate.hip: one line operation per ~13 clocks and L2 channel — the atomics execute
// memory-side). With 4x4 blocks a (tile, Gaussian) pair survives the cull of 2.16 blocks, with 8x8 blocks — lane =
// pixel — of 1.06 (tools/seg_stats.py: 1.19 M vs 0.59 M records per frame at D = 550 k), i.e. half the atomics; but a
// segment then has 64 pixels to visit instead of 16 with half the hit density: twice the (pixel, entry) evaluations,
// in four times fewer, four times longer wave chains. Measured in the training iteration (2 frames, D = 810 k):
// render_bwd 149 us either way (4x4: atomic-bound; 8x8: issue-bound), render_fwd 141 -> 271 us. Atomics and pixel
// work trade one for one, so the block stays 4x4 (profiles/r04_render_block_size.md).
#include <cstdlib>
#include <type_traits>

#include "gsr_common.h"

namespace gsr {

namespace {

constexpr int WAVES = GSR_TILE_PIX / GSR_WAVE;   // 4 waves per workgroup = 4 render blocks
constexpr int SUB = GSR_SUB;                      // the wave's pixel block is SUB x SUB: 8 (lane = pixel) or 4
constexpr int NPIX = SUB * SUB;
constexpr int BPT = GSR_SEG_BLOCKS;               // render blocks per tile: 4 (8x8) or 16 (4x4)
constexpr int WG_PER_TILE = BPT / WAVES;          // workgroups per tile: 1 or 4
constexpr int RING = 2 * GSR_WAVE;                // per-wave staging ring (survivors waiting for a full segment)
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_EPS = 0.0001f;
static_assert(SUB == 4 || SUB == 8, "render block edge");

// Block b of a tile (b = part * WAVES + wave: workgroup `part` of the tile, its wave): pixel offset of its corner.
// 8x8: the four quadrants; 4x4: quadrant = part, the quadrant's four 4x4 blocks = waves.
__device__ __forceinline__ int block_ox(int b) {
  return SUB == 8 ? (b & 1) * 8 : ((b >> 2) & 1) * 8 + (b & 1) * 4;
}
__device__ __forceinline__ int block_oy(int b) {
  return SUB == 8 ? (b >> 1) * 8 : (b >> 3) * 8 + ((b >> 1) & 1) * 4;
}
// inverse: block index from the corner's offset inside the tile
__device__ __forceinline__ int block_of(int ox, int oy) {
  return SUB == 8 ? ((oy >> 3) << 1) | (ox >> 3)
                  : ((oy >> 3) << 3) | ((ox >> 3) << 2) | (((oy >> 2) & 1) << 1) | ((ox >> 2) & 1);
}

// The same expression tree is used by forward and backward so that both take the same
// skip decisions for a given (pixel, Gaussian).
__device__ __forceinline__ float eval_power(float4 co, float dx, float dy) {
  const float q = fmaf(co.x * dx, dx, co.z * dy * dy);
  return fmaf(-0.5f, q, -(co.y * dx) * dy);
}

// Can the Gaussian reach alpha >= 1/255 at any pixel centre of the box [x0,x0+SUB-1] x [y0,y0+SUB-1]?
// xe = (centre, half extents of its alpha >= 1/255 box), written by K1 (gsr_common.h: alpha_extent).
// Exact-conservative: never drops an entry that blends on a pixel of the block; NaN -> keep.
__device__ __forceinline__ bool may_touch(float4 xe, float x0, float y0) {
  const bool outside = (xe.x + xe.z < x0) || (xe.x - xe.z > x0 + (float)(SUB - 1)) ||
                       (xe.y + xe.w < y0) || (xe.y - xe.w > y0 + (float)(SUB - 1));
  return !outside;
}

template <class T>
__device__ __forceinline__ const T* shift(const T* p, size_t bytes) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + bytes);
}
template <class T>
__device__ __forceinline__ T* shift_mut(T* p, size_t bytes) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(p) + bytes);
}

// Number of set bits of `mask` below this lane.
__device__ __forceinline__ int lane_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// ---- wave-wide DPP scans (GFX9 DPP: row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142,
// row_bcast:31 = 0x143). A lane whose source lies outside its row, or whose row is masked off, keeps
// `old` = the operation's identity.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// inclusive prefix product / sum over the 64 lanes (Hillis-Steele inside the rows of 16, then two row broadcasts)
#ifndef GSR_BUILTIN_SCAN
// In-place VOP2-DPP forms: without bound_ctrl a lane whose source is out of range (or whose row is masked
// off) is simply not written, i.e. keeps its own value — no identity operand, one instruction per step.
// A VALU write followed by a DPP read of the same register needs 2 wait states (s_nop 1); the three-way
// interleaved form hides them behind the other two scans.
#define GSR_SCAN_STEPS(OP, R)                                                    \
  "s_nop 1\n\t" OP " " R ", " R ", " R " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"   \
  "s_nop 1\n\t" OP " " R ", " R ", " R " row_shr:2 row_mask:0xf bank_mask:0xf\n\t"   \
  "s_nop 1\n\t" OP " " R ", " R ", " R " row_shr:4 row_mask:0xf bank_mask:0xf\n\t"   \
  "s_nop 1\n\t" OP " " R ", " R ", " R " row_shr:8 row_mask:0xf bank_mask:0xf\n\t"   \
  "s_nop 1\n\t" OP " " R ", " R ", " R " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
  "s_nop 1\n\t" OP " " R ", " R ", " R " row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
  "s_nop 1"
__device__ __forceinline__ float wave_scan_mul(float x) {
  asm volatile(GSR_SCAN_STEPS("v_mul_f32_dpp", "%0") : "+v"(x));
  return x;
}
__device__ __forceinline__ float wave_scan_add(float x) {
  asm volatile(GSR_SCAN_STEPS("v_add_f32_dpp", "%0") : "+v"(x));
  return x;
}
#define GSR_SCAN3_STEP(CTRL)                                                     \
  "v_add_f32_dpp %0, %0, %0 " CTRL " bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %2, %2, %2 " CTRL " bank_mask:0xf\n\t"
__device__ __forceinline__ void wave_scan_add3(float& a, float& b, float& c) {
  asm volatile("s_nop 1\n\t"
               GSR_SCAN3_STEP("row_shr:1 row_mask:0xf") GSR_SCAN3_STEP("row_shr:2 row_mask:0xf")
               GSR_SCAN3_STEP("row_shr:4 row_mask:0xf") GSR_SCAN3_STEP("row_shr:8 row_mask:0xf")
               GSR_SCAN3_STEP("row_bcast:15 row_mask:0xa") GSR_SCAN3_STEP("row_bcast:31 row_mask:0xc")
               "s_nop 1"
               : "+v"(a), "+v"(b), "+v"(c));
}
#else
__device__ __forceinline__ float wave_scan_mul(float x) {
  x *= dpp<0x111>(1.0f, x);
  x *= dpp<0x112>(1.0f, x);
  x *= dpp<0x114>(1.0f, x);
  x *= dpp<0x118>(1.0f, x);
  x *= dpp<0x142, 0xa>(1.0f, x);
  x *= dpp<0x143, 0xc>(1.0f, x);
  return x;
}
__device__ __forceinline__ float wave_scan_add(float x) {
  x += dpp<0x111>(0.0f, x);
  x += dpp<0x112>(0.0f, x);
  x += dpp<0x114>(0.0f, x);
  x += dpp<0x118>(0.0f, x);
  x += dpp<0x142, 0xa>(0.0f, x);
  x += dpp<0x143, 0xc>(0.0f, x);
  return x;
}
__device__ __forceinline__ void wave_scan_add3(float& a, float& b, float& c) {
  a = wave_scan_add(a); b = wave_scan_add(b); c = wave_scan_add(c);
}
#endif
// lane l gets lane l-1's value, lane 0 gets `first`
__device__ __forceinline__ float wave_shr1(float first, float v) { return dpp<0x138>(first, v); }

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_scan_add_i(int x) {
  x += dpp_i<0x111>(x);
  x += dpp_i<0x112>(x);
  x += dpp_i<0x114>(x);
  x += dpp_i<0x118>(x);
  x += dpp_i<0x142, 0xa>(x);
  x += dpp_i<0x143, 0xc>(x);
  return x;
}
__device__ __forceinline__ float read_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// v with lane `lane` := s (s and lane wave-uniform)
__device__ __forceinline__ float write_lane_at(float v, float s, int lane) {
  return ((int)(threadIdx.x & (GSR_WAVE - 1)) == lane) ? s : v;
}

// ------------------------------------------------------------------------------------ forward
// RECORD = false: the forward-only render (gsr_forward_eval*): nothing is left for a backward pass.
// Waves per SIMD the register allocator is asked to make room for (round 6, tools/build_gsr_variant.sh -DGSR_FWD_WAVES=n /
// -DGSR_BWD_WAVES=n; same box, bench_raster.py): the blend kernels are latency-bound on their gathers and their LDS ring.
// Forward at 6 (79 registers + 12 bytes of scratch instead of 87 and 5 waves): 111 -> 106 us at the avatar set, 164 -> 156 on the
// general set, 381 -> 367 at 5 M pairs per frame; 7: the same; 8: slower (72 bytes of scratch). The backward pass is bound by its
// atomics: 6 / 7 / 8 measured 104 -> 108 / 116 / 117 us — it keeps the allocator's own choice (90 registers, 5 waves).
#ifndef GSR_FWD_WAVES
#define GSR_FWD_WAVES 6
#endif
#if GSR_FWD_WAVES > 0
#define GSR_FWD_OCC __attribute__((amdgpu_waves_per_eu(GSR_FWD_WAVES, GSR_FWD_WAVES)))
#else
#define GSR_FWD_OCC
#endif
#ifdef GSR_BWD_WAVES
#define GSR_BWD_OCC __attribute__((amdgpu_waves_per_eu(GSR_BWD_WAVES, GSR_BWD_WAVES)))
#else
#define GSR_BWD_OCC
#endif
template <bool RECORD>
__global__ void __launch_bounds__(GSR_TILE_PIX) GSR_FWD_OCC
render_fwd_kernel(int W, int H, int gx, int T, int64_t max_pairs, int seg_cap, const uint32_t* __restrict__ tile_order,
                  const uint32_t* __restrict__ tile_offset,
                  const uint32_t* __restrict__ point_list, const float4* __restrict__ xyext,
                  const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb,
                  const float* __restrict__ bg, float* __restrict__ out_color,
                  float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                  uint2* __restrict__ seg_entries, float4* __restrict__ seg_ckpt, uint2* __restrict__ seg_info,
                  float4* __restrict__ pix_accum, uint32_t* __restrict__ seg_count, int32_t* __restrict__ seg_heads,
                  uint32_t* __restrict__ seg_list, size_t ws_stride) {
  {   // batched launch: blockIdx.y = frame
    const size_t off = (size_t)blockIdx.y * ws_stride;
    tile_order = shift(tile_order, off); seg_count = shift_mut(seg_count, off);
    tile_offset = shift(tile_offset, off); point_list = shift(point_list, off); xyext = shift(xyext, off);
    conic_opacity = shift(conic_opacity, off); rgb = shift(rgb, off);
    final_T = shift_mut(final_T, off); n_contrib = shift_mut(n_contrib, off);
    seg_entries = shift_mut(seg_entries, off); seg_ckpt = shift_mut(seg_ckpt, off);
    seg_info = shift_mut(seg_info, off); pix_accum = shift_mut(pix_accum, off); seg_heads = shift_mut(seg_heads, off);
    seg_list = shift_mut(seg_list, off);
    out_color += (size_t)blockIdx.y * 3 * H * W;
  }
  __shared__ uint32_t s_idx[WAVES][RING];
  __shared__ int s_k[WAVES][RING];
  // blocks walk the tiles in the binning's size order (longest lists first, tile_scan_kernel): the long
  // chains of the avatar's interior start at once instead of forming the launch's tail. XCD x takes the
  // tiles of rank = x (mod 8), all blocks of a tile on the same XCD (shared list, shared L2).
  const int xcd = blockIdx.x & 7;
  const int jb = blockIdx.x >> 3;
  const int rank = (jb / WG_PER_TILE) * 8 + xcd;
  if (rank >= T) return;
  const int tile = (int)tile_order[rank];
  const int wave = threadIdx.x / GSR_WAVE, lane = threadIdx.x & (GSR_WAVE - 1);
  const int blk = (jb % WG_PER_TILE) * WAVES + wave;             // this wave's block of the tile
  const int bx0 = (tile % gx) * GSR_TILE + block_ox(blk), by0 = (tile / gx) * GSR_TILE + block_oy(blk);
  const int64_t start = min((int64_t)tile_offset[tile], max_pairs);
  const int64_t end = min((int64_t)tile_offset[tile + 1], max_pairs);
  const int n = (int)(end - start);
  // this block's segment slots (gsr_common.h: seg_first_slot): no counter, no overflow
  const int64_t slot0 = seg_first_slot(start, tile) + (int64_t)blk * seg_block_capacity(start, end);
  const float fbx = (float)bx0, fby = (float)by0;
  // lane p < NPIX <-> pixel p of the block (row-major SUB x SUB)
  const int ppx = bx0 + (lane & (SUB - 1)), ppy = by0 + ((lane / SUB) & (SUB - 1));
  const bool pix_lane = lane < NPIX && ppx < W && ppy < H;
  // per-pixel state in lane p: transmittance, accumulated colour, deepest contributor (1-based)
  float vT = 1.0f, vC0 = 0.f, vC1 = 0.f, vC2 = 0.f;
  int vLast = 0;
  unsigned long long alive = __ballot(pix_lane);                 // bit p: pixel p still accumulating
  int recorded = 0;

  if (n > 0 && alive) {
    const uint32_t* plist = point_list + start;
    // Software pipeline. Culling walks the list in batches of 64: indices three batches ahead, their
    // (centre, extent) records two ahead. A segment is FORMED (64 survivors leave the LDS ring, their
    // records are requested) one step before it is BLENDED; the cull of the next batch runs in between,
    // so the gathers are not waited for.
    uint32_t idx0 = plist[min(lane, n - 1)];
    uint32_t idx1 = plist[min(GSR_WAVE + lane, n - 1)];
    uint32_t idx2 = plist[min(2 * GSR_WAVE + lane, n - 1)];
    float4 xe0 = xyext[idx0];
    float4 xe1 = xyext[idx1];
    int b0 = 0, head = 0, count = 0;
    bool more = true, pending = false;
    // the pending segment
    int take = 0, e_k = 0;
    uint32_t e_idx = 0;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f), co = c, col = c, ckpt = c;
    while (true) {
      if (!pending && (count >= GSR_WAVE || (!more && count > 0))) {
        // LDS traffic of one wave is ordered; no workgroup barrier needed for a wave-private slice
        __builtin_amdgcn_wave_barrier();
        take = min(count, GSR_WAVE);
        const int slot_l = (head + lane) & (RING - 1);
        e_idx = lane < take ? s_idx[wave][slot_l] : 0u;
        e_k = lane < take ? s_k[wave][slot_l] : 0x7fffffff;
        c = xyext[e_idx];          // .xy = centre (records of the 64 survivors: L2-resident gathers)
        co = conic_opacity[e_idx];
        col = rgb[e_idx];
        head = (head + take) & (RING - 1);
        count -= take;
        __builtin_amdgcn_wave_barrier();
        ckpt = make_float4(vT, vC0, vC1, vC2);      // the pixels' state at the segment's start
        pending = true;
      }
      if (more && count < GSR_WAVE) {
        const bool keep = (b0 + lane < n) && may_touch(xe0, fbx, fby);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
          const int pos = (head + count + lane_rank(mask)) & (RING - 1);
          s_idx[wave][pos] = idx0;
          s_k[wave][pos] = b0 + lane;
        }
        count += __popcll(mask);
        b0 += GSR_WAVE;
        more = b0 < n;
        idx0 = idx1; xe0 = xe1;
        idx1 = idx2; xe1 = xyext[idx2];
        idx2 = plist[min(b0 + 2 * GSR_WAVE + lane, n - 1)];
      }
      if (!pending) {
        if (!more && count == 0) break;
        continue;
      }
      pending = false;
      const bool valid = lane < take;
      // the block's pixels that are still accumulating, one after the other
      auto pixel = [&](const int p) {
        const float Tin = read_lane(vT, p);
        const float dx = c.x - (fbx + (float)(p & (SUB - 1)));
        const float dy = c.y - (fby + (float)(p / SUB));
        const float power = eval_power(co, dx, dy);
        const float alpha = fminf(ALPHA_MAX, co.w * __expf(power));
        const float a = (valid & (power <= 0.0f) & (alpha >= ALPHA_MIN)) ? alpha : 0.f;
        const unsigned long long hit = __ballot(a > 0.f);
        if (hit == 0ull) return;                                // nothing of this segment reaches the pixel
        const float Pincl = wave_scan_mul(1.0f - a);            // prod_{j<=l} (1 - a_j)
        const float Tincl = Tin * Pincl;
        // the first hit whose blend would push T below 1e-4 ends the pixel BEFORE it is accumulated
        const unsigned long long stopm = __ballot((a > 0.f) & (Tincl < T_EPS));
        const int first = stopm ? (__ffsll((long long)stopm) - 1) : GSR_WAVE;
        const float Texcl = Tin * wave_shr1(1.0f, Pincl);       // T in front of entry l
        const unsigned long long cm = first < GSR_WAVE ? (hit & ((1ull << first) - 1ull)) : hit;
        const bool contrib = (cm >> lane) & 1ull;
        const float w = contrib ? a * Texcl : 0.f;
        float s0 = col.x * w, s1 = col.y * w, s2 = col.z * w;
        wave_scan_add3(s0, s1, s2);
        const float S0 = read_lane(s0, GSR_WAVE - 1), S1 = read_lane(s1, GSR_WAVE - 1), S2 = read_lane(s2, GSR_WAVE - 1);
        const bool me = lane == p;
        vC0 = me ? vC0 + S0 : vC0;
        vC1 = me ? vC1 + S1 : vC1;
        vC2 = me ? vC2 + S2 : vC2;
        if (cm) {
          const int hi = 63 - __clzll((long long)cm);
          const int lastk = __builtin_amdgcn_readlane(e_k, hi) + 1;
          vLast = me ? lastk : vLast;
        }
        const float Tnew = first < GSR_WAVE ? read_lane(Texcl, first) : read_lane(Tincl, GSR_WAVE - 1);
        vT = write_lane_at(vT, Tnew, p);
        if (first < GSR_WAVE) alive &= ~(1ull << p);
      };
      if constexpr (SUB == 4) {
        // 4x4 blocks: the sixteen pixels unrolled — p is a compile-time constant (lane selects and v_readlane with
        // immediate lanes, the pixel's offsets folded into constants). Round 4 had replaced this by the bit-scan loop
        // of the 8x8 experiment for both block sizes: render_fwd 138 -> 143 us, render_bwd 141 -> 154 us (rocprofv3).
#pragma unroll
        for (int p = 0; p < NPIX; ++p)
          if ((alive >> p) & 1ull) pixel(p);
      } else {
        for (unsigned long long pm = alive; pm; pm &= pm - 1ull) pixel(__builtin_ctzll(pm));   // (p wave-uniform: an SGPR)
      }
      // record the segment for the backward pass
      if (RECORD) {
        const size_t slot = (size_t)(slot0 + recorded);
        seg_entries[slot * GSR_WAVE + lane] = make_uint2(e_idx, (uint32_t)e_k);
        if (lane < NPIX) seg_ckpt[slot * NPIX + lane] = ckpt;
        if (lane == 0) seg_info[slot] = make_uint2((uint32_t)bx0 | ((uint32_t)by0 << 16), (uint32_t)take);
        ++recorded;
      }
      if (!alive) break;
    }
  }
  if (pix_lane) {
    const size_t pix = (size_t)ppy * W + ppx;
    const size_t plane = (size_t)H * W;
    final_T[pix] = vT;
    n_contrib[pix] = (uint32_t)vLast;
    out_color[pix] = fmaf(vT, bg[0], vC0);
    out_color[plane + pix] = fmaf(vT, bg[1], vC1);
    out_color[2 * plane + pix] = fmaf(vT, bg[2], vC2);
    if (RECORD && recorded) pix_accum[pix] = make_float4(vC0, vC1, vC2, vT);
  }
  if (!RECORD) return;
  if (lane == 0) seg_count[tile * BPT + blk] = (uint32_t)recorded;
  // The segment-parallel backward pass strides over dense lists of slot ids, one per XCD class (this block ran on XCD
  // `xcd`, so will the waves that take its segments): one returning atomic per block claims a range of the class's list
  // (seg_heads[64 xcd] = the class count, one 256-byte line per counter: ~1 k returning atomics per address and
  // frame, spread over the whole launch; eight counters on ONE line measured +135 us).
  if (recorded > 0) {
    int base = 0;
    if (lane == 0) base = atomicAdd(&seg_heads[64 * xcd], recorded);
    base = __builtin_amdgcn_readfirstlane(base);
    uint32_t* list = seg_list + (size_t)xcd * seg_cap;
    for (int sgm = lane; sgm < recorded; sgm += GSR_WAVE) list[base + sgm] = (uint32_t)(slot0 + sgm);
  }
}

// ------------------------------------------------------------------------------------ backward
constexpr int NCOMP = GSR_PAIR_GRAD;   // dxy2 (scaled by W/2, H/2), dconic3, dopacity1, drgb3
struct SegRec { uint2 info; uint2 ent; float4 ck; };

// What a backward wave loads for one segment, in two dependent levels: the record itself (addressed by the
// slot), then the Gaussians it names and the pixels of its block.

struct SegData { float2 c; float4 co; float4 col; float4 pa; int last; float g0, g1, g2; };

__global__ void __launch_bounds__(GSR_TILE_PIX) GSR_BWD_OCC
render_bwd_kernel(int W, int H, int list_cap, const float2* __restrict__ xy,
                  const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb,
                  const float* __restrict__ bg, const uint32_t* __restrict__ n_contrib,
                  const uint2* __restrict__ seg_entries, const float4* __restrict__ seg_ckpt,
                  const uint2* __restrict__ seg_info, const float4* __restrict__ pix_accum,
                  const uint32_t* __restrict__ seg_list, const int32_t* __restrict__ seg_heads,
                  const int32_t* __restrict__ status, const float* __restrict__ dL_dout,
                  float* __restrict__ grad_acc, size_t ws_stride) {
  {   // batched launch: blockIdx.y = frame
    const size_t off = (size_t)blockIdx.y * ws_stride;
    status = shift(status, off);
    xy = shift(xy, off); conic_opacity = shift(conic_opacity, off); rgb = shift(rgb, off);
    n_contrib = shift(n_contrib, off); seg_entries = shift(seg_entries, off); seg_ckpt = shift(seg_ckpt, off);
    seg_info = shift(seg_info, off); pix_accum = shift(pix_accum, off); seg_list = shift(seg_list, off);
    seg_heads = shift(seg_heads, off);
    grad_acc = shift_mut(grad_acc, off);
    dL_dout += (size_t)blockIdx.y * 3 * H * W;
  }
  // a frame whose pair buffer overflowed yields no gradient at all (preprocess_bwd writes the zeros)
  if (status[1] != 0) return;
  // per-wave transposition buffer for the gradient records: lane-major [entry][component] in, flat out
  __shared__ float s_g[WAVES][GSR_WAVE * NCOMP];
  __shared__ uint32_t s_gi[WAVES][GSR_WAVE];
  const int lane = threadIdx.x & (GSR_WAVE - 1);
  const int wave = threadIdx.x / GSR_WAVE;
  const int nwaves = gridDim.x * WAVES;
  // persistent waves (nwaves is a multiple of 32): the forward pass listed its segments per XCD class (the tiles of
  // rank = x mod 8 ran on XCD x: render_fwd's tail); workgroup b runs on XCD b % 8 and its wave takes the
  // entries j = (its index among that XCD's waves), += (waves per XCD) of list x: a tile's segments fetch their
  // Gaussians' records into ONE L2
  const int xcd = blockIdx.x & 7;
  const uint32_t* my_list = seg_list + (size_t)xcd * list_cap;
  const int mine = min(seg_heads[64 * xcd], list_cap);
  // waves per XCD that take part: all of them for a long list; for a short one only as many as leave every wave
  // GSR_BWD_SEGS segments (a wave's second-level loads overlap with its previous segment's pixel loop: a wave with a single
  // segment hides nothing) — the rest of the grid exits at once
#ifndef GSR_BWD_SEGS
#define GSR_BWD_SEGS 2
#endif
  const int wpx = GSR_BWD_SEGS > 0 ? max(WAVES * 16, min(nwaves / 8, (mine + GSR_BWD_SEGS - 1) / GSR_BWD_SEGS)) : nwaves / 8;
  const int my_first = (blockIdx.x >> 3) * WAVES + wave;
  if (my_first >= wpx) return;
  const float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;
  const size_t plane = (size_t)H * W;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const int p_l = lane & (NPIX - 1);

  auto load_rec = [&](int j) {
    SegRec r;
    const size_t seg = my_list[j];
    r.info = seg_info[seg];
    r.ent = seg_entries[seg * GSR_WAVE + lane];
    r.ck = seg_ckpt[seg * NPIX + p_l];                   // T, C of the block's pixels at the segment's start
    return r;
  };
  auto load_data = [&](const SegRec& r) {
    SegData d;
    const uint32_t idx = lane < (int)r.info.y ? r.ent.x : 0u;
    d.c = xy[idx];
    d.co = conic_opacity[idx];
    d.col = rgb[idx];
    const int ppx = (int)(r.info.x & 0xffffu) + (p_l & (SUB - 1)), ppy = (int)(r.info.x >> 16) + (p_l / SUB);
    const bool inside = ppx < W && ppy < H;
    const size_t pix = inside ? (size_t)ppy * W + ppx : 0;
    d.pa = pix_accum[pix];                                        // C_total, T_final
    d.last = inside ? (int)n_contrib[pix] : 0;
    d.g0 = inside ? dL_dout[pix] : 0.f;
    d.g1 = inside ? dL_dout[plane + pix] : 0.f;
    d.g2 = inside ? dL_dout[2 * plane + pix] : 0.f;
    return d;
  };

  // Two-level software pipeline: a wave's memory latency (record -> Gaussians/pixels, two dependent round
  // trips of microseconds under load) overlaps with the pixel loop of its previous segment.
  int seg = my_first;
  if (seg >= mine) return;
  SegRec rec = load_rec(seg);
  SegData dat = load_data(rec);
  bool have_next = seg + wpx < mine;
  SegRec rec_n = rec;
  if (have_next) rec_n = load_rec(seg + wpx);
  while (true) {
    const int bx0 = (int)(rec.info.x & 0xffffu), by0 = (int)(rec.info.x >> 16), cnt = (int)rec.info.y;
    const bool valid = lane < cnt;
    const uint32_t idx = valid ? rec.ent.x : 0u;
    const int k = valid ? (int)rec.ent.y : 0x7fffffff;
    const float2 c = dat.c;
    const float4 co = dat.co;
    const float4 col = dat.col;
    const float4 pa = dat.pa, ck = rec.ck;
    const int vLast = dat.last;
    const float vg0 = dat.g0, vg1 = dat.g1, vg2 = dat.g2;
    const float vTs = ck.x;
    // R = (C_total - C_start).g + T_final (bg.g): what lies behind the segment's first entry
    const float vR = (pa.x - ck.y) * vg0 + (pa.y - ck.z) * vg1 + (pa.z - ck.w) * vg2 +
                     pa.w * (bg0 * vg0 + bg1 * vg1 + bg2 * vg2);
    const float fbx = (float)bx0, fby = (float)by0;
    const int kmin = __builtin_amdgcn_readfirstlane(k);            // entries are in list order
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f;
    {
      // the block's pixels that did not saturate in front of this segment, one after the other
      auto pixel = [&](const int p) {
        const int lastp = __builtin_amdgcn_readlane(vLast, p);
        if (SUB == 4 && lastp <= kmin) return;                      // (the static loop tests here; the bit scan below never gets such a pixel)
        const float dx = c.x - (fbx + (float)(p & (SUB - 1)));
        const float dy = c.y - (fby + (float)(p / SUB));
        const float power = eval_power(co, dx, dy);
        const float G = __expf(power);
        const float alpha = fminf(ALPHA_MAX, co.w * G);
        const float a = ((k < lastp) & (power <= 0.0f) & (alpha >= ALPHA_MIN)) ? alpha : 0.f;
        if (__ballot(a > 0.f) == 0ull) return;
        const float g0 = read_lane(vg0, p), g1 = read_lane(vg1, p), g2 = read_lane(vg2, p);
        const float om = 1.0f - a;
        const float T = read_lane(vTs, p) * wave_shr1(1.0f, wave_scan_mul(om));   // in front of this entry
        const float cg = col.x * g0 + col.y * g1 + col.z * g2;
        const float w = a * T;
        const float Sincl = wave_scan_add(cg * w);                   // (C_<=k - C_start).g
        const float rc = __builtin_amdgcn_rcpf(om);                  // a <= 0.99
        float dL_dalpha = fmaf(T, cg, -rc * (read_lane(vR, p) - Sincl));
        dL_dalpha = (a > 0.f) ? dL_dalpha : 0.f;
        const float dL_dG = co.w * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddx = -gdx * co.x - gdy * co.y;
        const float dG_ddy = -gdy * co.z - gdx * co.y;
        v0 = fmaf(dL_dG, dG_ddx, v0);
        v1 = fmaf(dL_dG, dG_ddy, v1);
        v2 = fmaf(-0.5f * gdx * dx, dL_dG, v2);
        v3 = fmaf(-gdx * dy, dL_dG, v3);
        v4 = fmaf(-0.5f * gdy * dy, dL_dG, v4);
        v5 = fmaf(G, dL_dalpha, v5);
        v6 = fmaf(w, g0, v6);
        v7 = fmaf(w, g1, v7);
        v8 = fmaf(w, g2, v8);
      };
      if constexpr (SUB == 4) {        // sixteen pixels unrolled, p a compile-time constant (see render_fwd_kernel)
#pragma unroll
        for (int p = 0; p < NPIX; ++p) pixel(p);
      } else {
        for (unsigned long long pm = __ballot(lane < NPIX && vLast > kmin); pm; pm &= pm - 1ull) pixel(__builtin_ctzll(pm));
      }
    }
    // the next segment's record has arrived by now: start its second-level loads
    SegData dat_n = dat;
    if (have_next) dat_n = load_data(rec_n);
    // One gradient record = 9 floats in one 64-byte line (GSR_GRAD_STRIDE). Issued lane-per-entry, an atomic
    // instruction would touch 64 lines (measured: 334 of 383 us per frame); transposed through LDS each
    // instruction covers 7 whole records, i.e. ~8 lines: 9x fewer memory-side transactions.
    float* sg = s_g[wave];
    sg[lane * NCOMP + 0] = v0 * half_w; sg[lane * NCOMP + 1] = v1 * half_h; sg[lane * NCOMP + 2] = v2;
    sg[lane * NCOMP + 3] = v3; sg[lane * NCOMP + 4] = v4; sg[lane * NCOMP + 5] = v5;
    sg[lane * NCOMP + 6] = v6; sg[lane * NCOMP + 7] = v7; sg[lane * NCOMP + 8] = v8;
    s_gi[wave][lane] = valid ? idx : 0xffffffffu;
    __builtin_amdgcn_wave_barrier();
    {
#pragma unroll 3
      for (int r = 0; r < NCOMP; ++r) {
        const int fl = r * GSR_WAVE + lane;
        const int e = (fl * 7282) >> 16;                 // fl / 9 for fl < 576
        const int q = fl - e * NCOMP;
        const uint32_t gi = s_gi[wave][e];
        const float val = sg[fl];
        if (gi != 0xffffffffu && val != 0.f) unsafeAtomicAdd(&grad_acc[(size_t)gi * GSR_GRAD_STRIDE + q], val);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (!have_next) break;
    seg += wpx;
    rec = rec_n;
    dat = dat_n;
    have_next = seg + wpx < mine;
    if (have_next) rec_n = load_rec(seg + wpx);
  }
}


constexpr int BWD_WG = 1024;           // 16 waves: one workgroup per CU (LDS), all of a tile's segments on it
constexpr int NCAP = 3072;             // list entries whose gradient rows fit the LDS table (9 floats each: 108 KiB)

// Tile-grouped variant of K6 (measurements below: round 3, 4x4 blocks): one workgroup per TILE walks the tile's recorded segments, sums each list entry's
// gradients in an LDS table indexed by list position and flushes the table once per tile (one record per (tile,
// Gaussian) pair). Two uses:
//   * DET (settings.debug, the reference's debug knob /root/reference/gaussian_renderer/__init__.py:33): ONE wave per
//     tile takes the segments in their fixed order, the rows go to per-pair records instead of atomics, and
//     gather_pair_grads_kernel sums a Gaussian's pairs in tile order: bitwise repeatable, several times slower;
//   * !DET with 16 waves per tile is the layout round 2's review asked for (per-tile LDS staging, one atomic record
//     per pair). Built and measured (profiles/r03_render_bwd_tile.md): 290-320 us per 2-frame launch against 150 for
//     the segment-parallel kernel above — ds_add_f32 retires about one lane per clock and CU, so the 14.7 M
//     per-entry adds of a launch cost as much in LDS as they do as memory-side atomics, the flush is a second,
//     serialised round of atomics per tile, and a tile is a much coarser unit than a segment (the largest tile is
//     the launch's tail). Kept compiled for the record; the default path is the segment-parallel kernel.
// Lists longer than NCAP entries are handled in passes over position ranges (a segment is replayed in every pass its
// entries reach into).
template <bool DET>
__global__ void __launch_bounds__(DET ? GSR_WAVE : BWD_WG)
render_bwd_tile_kernel(int W, int H, int gx, int T, int ordered, int64_t max_pairs, const uint32_t* __restrict__ tile_order,
                       const uint32_t* __restrict__ tile_offset, const uint32_t* __restrict__ point_list,
                       const float2* __restrict__ xy, const float4* __restrict__ conic_opacity,
                       const float4* __restrict__ rgb, const float* __restrict__ bg,
                       const uint32_t* __restrict__ n_contrib, const uint2* __restrict__ seg_entries,
                       const float4* __restrict__ seg_ckpt, const uint2* __restrict__ seg_info,
                       const float4* __restrict__ pix_accum, const uint32_t* __restrict__ seg_count,
                       const float* __restrict__ dL_dout, float* __restrict__ grad_acc,
                       float* __restrict__ pair_grad, size_t ws_stride) {
  {   // batched launch: blockIdx.y = frame
    const size_t off = (size_t)blockIdx.y * ws_stride;
    tile_order = shift(tile_order, off); tile_offset = shift(tile_offset, off); point_list = shift(point_list, off);
    xy = shift(xy, off); conic_opacity = shift(conic_opacity, off); rgb = shift(rgb, off);
    n_contrib = shift(n_contrib, off); seg_entries = shift(seg_entries, off); seg_ckpt = shift(seg_ckpt, off);
    seg_info = shift(seg_info, off); pix_accum = shift(pix_accum, off); seg_count = shift(seg_count, off);
    grad_acc = shift_mut(grad_acc, off); pair_grad = shift_mut(pair_grad, off);
    dL_dout += (size_t)blockIdx.y * 3 * H * W;
  }
  extern __shared__ float s_dyn[];                 // gradient table [rows][NCOMP]
  __shared__ float4 s_pa[GSR_TILE_PIX];            // C_total.rgb, T_final per pixel of the tile
  __shared__ float4 s_g[GSR_TILE_PIX];             // dL/dpixel.rgb, .w = last contributor (as float bits)
  __shared__ int s_pre[GSR_SEG_BLOCKS + 1];
  __shared__ int s_kmax;
  const int tid = threadIdx.x;
  const int lane = tid & (GSR_WAVE - 1);
  const int wave = tid / GSR_WAVE;
  const int nwaves = blockDim.x / GSR_WAVE;
  const float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const int p_l = lane & (NPIX - 1);
  const size_t plane = (size_t)H * W;
  // persistent workgroups: rank = blockIdx.x, += gridDim.x over the size-ordered tiles (gridDim.x is a multiple of 8:
  // the XCD of a rank is kept); the first empty tile ends the walk — launching one workgroup per tile would dispatch
  // thousands of 110 KiB-LDS workgroups that find their tile empty. Above 8192 tiles tile_scan_kernel leaves the
  // identity order (`ordered` = 0): an empty tile is then skipped, not the end of the walk.
  for (int rank = blockIdx.x; rank < T; rank += gridDim.x) {
  const int tile = (int)tile_order[rank];
  const int64_t start = min((int64_t)tile_offset[tile], max_pairs);
  const int64_t end = min((int64_t)tile_offset[tile + 1], max_pairs);
  const int n = (int)(end - start);
  if (n == 0) { if (ordered) break; continue; }
  __syncthreads();                                 // the previous tile's LDS is free
  // segments per block -> exclusive prefix (wave 0)
  if (tid < GSR_WAVE) {
    const int c = lane < GSR_SEG_BLOCKS ? (int)seg_count[tile * GSR_SEG_BLOCKS + lane] : 0;
    const int incl = wave_scan_add_i(c);
    if (lane < GSR_SEG_BLOCKS) s_pre[lane + 1] = incl;
    if (lane == 0) { s_pre[0] = 0; s_kmax = 0; }
  }
  __syncthreads();
  const int total = s_pre[GSR_SEG_BLOCKS];
  if (total == 0) {
    if (DET) {      // no segment was recorded: every pair of the tile has a zero record
      float* out = pair_grad + (size_t)start * NCOMP;
      for (int e = tid; e < n * NCOMP; e += blockDim.x) out[e] = 0.f;
    }
    continue;
  }
  const int capb = seg_block_capacity(start, end);
  const int64_t slot_tile = seg_first_slot(start, tile);
  const int tx0 = (tile % gx) * GSR_TILE, ty0 = (tile / gx) * GSR_TILE;
  // the tile's pixels, block-major: entry NPIX b + p = pixel p of render block b (the forward pass's block index)
  for (int e = tid; e < GSR_TILE_PIX; e += blockDim.x) {
    const int b = e / NPIX, p = e & (NPIX - 1);
    const int px = tx0 + block_ox(b) + (p & (SUB - 1));
    const int py = ty0 + block_oy(b) + (p / SUB);
    const bool inside = px < W && py < H;
    const size_t pix = inside ? (size_t)py * W + px : 0;
    const int last = inside ? (int)n_contrib[pix] : 0;
    s_pa[e] = pix_accum[pix];
    s_g[e] = make_float4(inside ? dL_dout[pix] : 0.f, inside ? dL_dout[plane + pix] : 0.f,
                         inside ? dL_dout[2 * plane + pix] : 0.f, __int_as_float(last));
    if (last > 0) atomicMax(&s_kmax, last);
  }
  __syncthreads();
  const int kmax = min(s_kmax, n);                 // entries at positions >= kmax were never blended

  auto slot_of = [&](int i) -> size_t {            // i-th segment of the tile: block by the prefix, then its index
    int b = 0;
#pragma unroll
    for (int j = 1; j < GSR_SEG_BLOCKS; ++j) b += (i >= s_pre[j]) ? 1 : 0;
    return (size_t)(slot_tile + (int64_t)b * capb + (i - s_pre[b]));
  };
  auto load_rec = [&](int i) {
    SegRec r;
    const size_t seg = slot_of(i);
    r.info = seg_info[seg];
    r.ent = seg_entries[seg * GSR_WAVE + lane];
    r.ck = seg_ckpt[seg * NPIX + p_l];             // T, C of the block's pixels at the segment's start
    return r;
  };

  for (int k0 = 0; k0 < kmax; k0 += NCAP) {
    const int rows = min(NCAP, kmax - k0);
    for (int e = tid; e < rows * NCOMP; e += blockDim.x) s_dyn[e] = 0.f;
    __syncthreads();
    // two-level software pipeline, as in the forward pass: the record of segment i + 2 nw and the Gaussians of
    // segment i + nw are in flight while segment i is evaluated
    struct SegGauss { float2 c; float4 co, col; };
    auto load_gauss = [&](const SegRec& r) {
      SegGauss d;
      const uint32_t idx = lane < (int)r.info.y ? r.ent.x : 0u;
      d.c = xy[idx]; d.co = conic_opacity[idx]; d.col = rgb[idx];
      return d;
    };
    int i = wave;
    SegRec rec_c{}, rec_n{};
    SegGauss dat_c{};
    if (i < total) { rec_c = load_rec(i); dat_c = load_gauss(rec_c); }
    if (i + nwaves < total) rec_n = load_rec(i + nwaves);
    for (; i < total; i += nwaves) {
      const SegRec rec = rec_c;
      const SegGauss dat = dat_c;
      rec_c = rec_n;
      if (i + nwaves < total) dat_c = load_gauss(rec_c);
      if (i + 2 * nwaves < total) rec_n = load_rec(i + 2 * nwaves);
      const int cnt = (int)rec.info.y;
      const bool valid = lane < cnt;
      const int k = valid ? (int)rec.ent.y : 0x7fffffff;
      const int kfirst = __builtin_amdgcn_readfirstlane(k);                       // entries are in list order
      const int klast = __builtin_amdgcn_readlane(k, cnt - 1);
      if (klast < k0 || kfirst >= k0 + rows) continue;                            // not in this pass
      const float2 c = dat.c;
      const float4 co = dat.co;
      const float4 col = dat.col;
      const int bx0 = (int)(rec.info.x & 0xffffu), by0 = (int)(rec.info.x >> 16);
      const int b = block_of(bx0 - tx0, by0 - ty0);
      const float4 pa = s_pa[NPIX * b + p_l];
      const float4 gp = s_g[NPIX * b + p_l];
      const float4 ck = rec.ck;
      const int vLast = __float_as_int(gp.w);
      const float vg0 = gp.x, vg1 = gp.y, vg2 = gp.z;
      const float vTs = ck.x;
      // R = (C_total - C_start).g + T_final (bg.g): what lies behind the segment's first entry
      const float vR = (pa.x - ck.y) * vg0 + (pa.y - ck.z) * vg1 + (pa.z - ck.w) * vg2 +
                       pa.w * (bg0 * vg0 + bg1 * vg1 + bg2 * vg2);
      const float fbx = (float)bx0, fby = (float)by0;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f;
      for (unsigned long long pm = __ballot(lane < NPIX && vLast > kfirst); pm; pm &= pm - 1ull) {
        const int p = __builtin_ctzll(pm);                          // a pixel that had not saturated in front of this segment
        const int lastp = __builtin_amdgcn_readlane(vLast, p);
        const float dx = c.x - (fbx + (float)(p & (SUB - 1)));
        const float dy = c.y - (fby + (float)(p / SUB));
        const float power = eval_power(co, dx, dy);
        const float G = __expf(power);
        const float alpha = fminf(ALPHA_MAX, co.w * G);
        const float a = ((k < lastp) & (power <= 0.0f) & (alpha >= ALPHA_MIN)) ? alpha : 0.f;
        if (__ballot(a > 0.f) == 0ull) continue;
        const float g0 = read_lane(vg0, p), g1 = read_lane(vg1, p), g2 = read_lane(vg2, p);
        const float om = 1.0f - a;
        const float Tk = read_lane(vTs, p) * wave_shr1(1.0f, wave_scan_mul(om));   // in front of this entry
        const float cg = col.x * g0 + col.y * g1 + col.z * g2;
        const float w = a * Tk;
        const float Sincl = wave_scan_add(cg * w);                   // (C_<=k - C_start).g
        const float rc = __builtin_amdgcn_rcpf(om);                  // a <= 0.99
        float dL_dalpha = fmaf(Tk, cg, -rc * (read_lane(vR, p) - Sincl));
        dL_dalpha = (a > 0.f) ? dL_dalpha : 0.f;
        const float dL_dG = co.w * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddx = -gdx * co.x - gdy * co.y;
        const float dG_ddy = -gdy * co.z - gdx * co.y;
        v0 = fmaf(dL_dG, dG_ddx, v0);
        v1 = fmaf(dL_dG, dG_ddy, v1);
        v2 = fmaf(-0.5f * gdx * dx, dL_dG, v2);
        v3 = fmaf(-gdx * dy, dL_dG, v3);
        v4 = fmaf(-0.5f * gdy * dy, dL_dG, v4);
        v5 = fmaf(G, dL_dalpha, v5);
        v6 = fmaf(w, g0, v6);
        v7 = fmaf(w, g1, v7);
        v8 = fmaf(w, g2, v8);
      }
      // the entry's sums over the block's pixels -> its row of the tile's table
      if (valid && k >= k0 && k < k0 + rows) {
        float* row = s_dyn + (size_t)(k - k0) * NCOMP;
        atomicAdd(row + 0, v0 * half_w); atomicAdd(row + 1, v1 * half_h); atomicAdd(row + 2, v2);
        atomicAdd(row + 3, v3); atomicAdd(row + 4, v4); atomicAdd(row + 5, v5);
        atomicAdd(row + 6, v6); atomicAdd(row + 7, v7); atomicAdd(row + 8, v8);
      }
    }
    __syncthreads();
    // flush the pass's rows: consecutive threads take consecutive floats of the table (a wave's atomic instruction
    // covers 7 whole 64-byte gradient records)
    const uint32_t* pl = point_list + start + k0;
    if (DET) {
      float* out = pair_grad + (size_t)(start + k0) * NCOMP;
      for (int e = tid; e < rows * NCOMP; e += blockDim.x) out[e] = s_dyn[e];
    } else {
      for (int e = tid; e < rows * NCOMP; e += blockDim.x) {
        const float val = s_dyn[e];
        if (val != 0.f) {
          const int kk = (int)(((unsigned)e * 7282u) >> 16);         // e / 9 for e < 3072 * 9
          unsafeAtomicAdd(&grad_acc[(size_t)pl[kk] * GSR_GRAD_STRIDE + (e - kk * NCOMP)], val);
        }
      }
    }
    __syncthreads();
  }
  if (DET) {      // rows beyond kmax: nothing was blended there
    float* out = pair_grad + (size_t)start * NCOMP;
    for (int e = kmax * NCOMP + tid; e < n * NCOMP; e += blockDim.x) out[e] = 0.f;
  }
  }   // persistent walk over the tiles
}

// DET: a Gaussian's gradient record = the sum of its pairs' records in tile order (row-major over its rectangle).
// Its position in a tile's list is found by binary search on the sort key (depth bits, index).
__global__ void __launch_bounds__(256)
gather_pair_grads_kernel(int P, int gx, int64_t max_pairs, const int4* __restrict__ rect,
                         const float* __restrict__ depth, const uint32_t* __restrict__ tile_offset,
                         const uint32_t* __restrict__ point_list, const float* __restrict__ pair_grad,
                         float* __restrict__ grad_acc, size_t ws_stride) {
  {
    const size_t off = (size_t)blockIdx.y * ws_stride;
    rect = shift(rect, off); depth = shift(depth, off); tile_offset = shift(tile_offset, off);
    point_list = shift(point_list, off); pair_grad = shift(pair_grad, off); grad_acc = shift_mut(grad_acc, off);
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int4 rc = rect[i];
  if ((rc.z - rc.x) * (rc.w - rc.y) <= 0) return;
  const uint64_t key = ((uint64_t)__float_as_uint(depth[i]) << 32) | (uint32_t)i;
  float acc[NCOMP];
#pragma unroll
  for (int q = 0; q < NCOMP; ++q) acc[q] = 0.f;
  for (int ty = rc.y; ty < rc.w; ++ty)
    for (int tx = rc.x; tx < rc.z; ++tx) {
      const int t = ty * gx + tx;
      int64_t lo = min((int64_t)tile_offset[t], max_pairs), hi = min((int64_t)tile_offset[t + 1], max_pairs);
      const int64_t e = hi;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const uint32_t j = point_list[mid];
        const uint64_t kj = ((uint64_t)__float_as_uint(depth[j]) << 32) | j;
        if (kj < key) lo = mid + 1; else hi = mid;
      }
      if (lo < e && point_list[lo] == (uint32_t)i) {
#pragma unroll
        for (int q = 0; q < NCOMP; ++q) acc[q] += pair_grad[(size_t)lo * NCOMP + q];
      }
    }
#pragma unroll
  for (int q = 0; q < NCOMP; ++q) grad_acc[(size_t)i * GSR_GRAD_STRIDE + q] = acc[q];
}

}  // namespace

hipError_t launch_render_fwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             float* out_color, bool record, const Batch& bt, hipStream_t stream) {
  if (d.T == 0) return hipSuccess;
  {
    ProfScope prof_(K_RENDER_FWD, stream);
    const dim3 grid(8 * WG_PER_TILE * ((d.T + 7) / 8), bt.frames);
#define GSR_FWD_ARGS d.W, d.H, d.gx, d.T, d.max_pairs, d.seg_cap, ws.tile_count, ws.tile_offset, ws.point_list, ws.xyext, \
                     ws.conic_opacity, ws.rgb, s.bg, out_color, ws.final_T, ws.n_contrib, ws.seg_entries, ws.seg_ckpt,    \
                     ws.seg_info, ws.pix_accum, ws.seg_count, ws.seg_heads, ws.seg_list, bt.ws_stride
    if (record) hipLaunchKernelGGL(render_fwd_kernel<true>, grid, dim3(GSR_TILE_PIX), 0, stream, GSR_FWD_ARGS);
    else hipLaunchKernelGGL(render_fwd_kernel<false>, grid, dim3(GSR_TILE_PIX), 0, stream, GSR_FWD_ARGS);
#undef GSR_FWD_ARGS
  }
  return hipGetLastError();
}

hipError_t launch_render_bwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             const float* dL_dout, const Batch& bt, hipStream_t stream) {
  if (d.T == 0 || d.P == 0) return hipSuccess;
  ProfScope prof_(K_RENDER_BWD, stream);
  if (s.debug) {
    // deterministic: one wave per tile, per-pair records, per-Gaussian gather in tile order
    const size_t lds = (size_t)NCAP * NCOMP * sizeof(float);
    static PerDeviceFlag attr_set;       
    if (!attr_set) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(render_bwd_tile_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
    hipLaunchKernelGGL(render_bwd_tile_kernel<true>, dim3(d.T < 2048 ? d.T : 2048, bt.frames), dim3(GSR_WAVE), lds, stream,
                       d.W, d.H, d.gx, d.T, tile_order_is_sorted(d.T), d.max_pairs, ws.tile_count, ws.tile_offset,
                       ws.point_list, ws.xy, ws.conic_opacity, ws.rgb, s.bg, ws.n_contrib, ws.seg_entries, ws.seg_ckpt,
                       ws.seg_info, ws.pix_accum, ws.seg_count, dL_dout, ws.grad_acc, ws.pair_grad, bt.ws_stride);
    hipLaunchKernelGGL(gather_pair_grads_kernel, dim3((d.P + 255) / 256, bt.frames), dim3(256), 0, stream, d.P, d.gx,
                       d.max_pairs, ws.rect, ws.depth, ws.tile_offset, ws.point_list, ws.pair_grad, ws.grad_acc,
                       bt.ws_stride);
    return hipGetLastError();
  }
  // one persistent wave per few segments of the per-XCD lists the forward pass left
  const int per_frame = max(16, (GSR_BWD_BLOCKS / bt.frames) & ~15);      // waves per frame: a multiple of 64
  hipLaunchKernelGGL(render_bwd_kernel, dim3(per_frame, bt.frames), dim3(GSR_TILE_PIX), 0, stream, d.W, d.H, d.seg_cap, ws.xy,
                     ws.conic_opacity, ws.rgb, s.bg, ws.n_contrib, ws.seg_entries, ws.seg_ckpt, ws.seg_info,
                     ws.pix_accum, ws.seg_list, ws.seg_heads, ws.status, dL_dout, ws.grad_acc, bt.ws_stride);
  return hipGetLastError();
}

}  // namespace gsr
