// gsr_render.hip — K5 render_fwd and K6 render_bwd (per-tile alpha compositing).
//
// Shape of the problem on MI355X (measured, 200k avatar Gaussians at 1024^2): ~590 of 4096 tiles are
// occupied, their depth-sorted lists hold ~1000 (up to ~3600) entries, and every pixel has to walk
// its list serially until it saturates. With one lane per pixel that is ~2300 long-running waves on
// a chip with 1024 SIMDs: 85 % of the issue slots idle and the kernel time is the longest chain.
// So the work is cut the other way:
//
//   * a wave owns a 4x4 pixel block and spends FOUR lanes on every pixel (quad = pixel, lane&3 =
//     which of 4 consecutive list entries it evaluates). Front-to-back compositing is a product of
//     (1 - alpha) terms, i.e. an associative scan: the transmittance in front of each of the 4
//     entries is an exclusive prefix product across the quad (two DPP quad_perm steps), the
//     termination test, the colour sums and — in the backward pass — the "colour behind"
//     recurrence (an affine map per entry) compose the same way. 4x the waves, chains 4x shorter.
//   * waves are independent (no __syncthreads): each walks the tile's list 64 entries at a time —
//     lane l gathers entry l (index -> 16-byte SoA records), tests the Gaussian's exact-conservative
//     bounding box against the wave's 4x4 block, survivors are compacted (ballot + mbcnt) into the
//     wave's LDS slice; the gathers of the next batch fly while the current one is blended.
//     The culling never drops an entry with alpha >= 1/255 on any pixel of the block, so results
//     are unchanged.
//   * backward: per-pixel gradient contributions are summed over the 16 pixels of the block through
//     LDS columns (lane-private slots, no atomics), then one global atomic per (wave, Gaussian,
//     component) — 36 of them per instruction.
// grid = (4 * tiles, frames): blockIdx.x = tile * 4 + quadrant, the 4 waves of a block take the
// 4x4 sub-blocks of the 8x8 quadrant. Spec: SURVEY.md Appendix A.3 (forward) and A.4 (backward).
#include <cstdlib>
#include <type_traits>

#include "gsr_common.h"

namespace gsr {

namespace {

constexpr int WAVES = GSR_TILE_PIX / GSR_WAVE;   // 4 waves per block
constexpr int SUB = 4;                            // the wave's pixel block is SUB x SUB
constexpr int LPP = 4;                            // lanes per pixel (= one DPP quad)
constexpr int UNR = 2;                            // list steps (of LPP entries) evaluated together
constexpr int PAD = LPP * UNR;                    // staged lists are padded to a multiple of this
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_EPS = 0.0001f;

// The same expression tree is used by forward and backward so that both take the same
// skip decisions for a given (pixel, Gaussian).
__device__ __forceinline__ float eval_power(float4 co, float dx, float dy) {
  const float q = fmaf(co.x * dx, dx, co.z * dy * dy);
  return fmaf(-0.5f, q, -(co.y * dx) * dy);
}

// Can Gaussian (xy, conic A,B,C, opacity o) reach alpha >= 1/255 at any pixel centre of the
// box [x0,x0+SUB-1] x [y0,y0+SUB-1]?  alpha >= 1/255  <=>  d^T Q d <= tau, tau = 2 ln(255 o); the
// axis-aligned bounding box of that ellipse has half extents sqrt(tau * Sigma_xx), sqrt(tau *
// Sigma_yy) with Sigma = Q^-1. The test is conservative (slightly inflated, NaN -> keep).
__device__ __forceinline__ bool may_touch(float2 c, float4 co, float x0, float y0) {
  const float tau = 2.0f * __logf(255.0f * co.w) + 1e-3f;   // margin for the fast log
  const float det = co.x * co.z - co.y * co.y;
  const float inv = 1.0f / det;
  const float hx = sqrtf(tau * co.z * inv) * 1.001f + 0.01f;
  const float hy = sqrtf(tau * co.x * inv) * 1.001f + 0.01f;
  const bool outside = (c.x + hx < x0) || (c.x - hx > x0 + (float)(SUB - 1)) ||
                       (c.y + hy < y0) || (c.y - hy > y0 + (float)(SUB - 1)) || (tau < 0.0f);
  return !outside;
}

template <class T>
__device__ __forceinline__ const T* shift(const T* p, size_t bytes) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + bytes);
}

// Number of set bits of `mask` below this lane.
__device__ __forceinline__ int lane_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// DPP quad permutes (a quad = the 4 lanes of one pixel). quad_perm [a,b,c,d] = a | b<<2 | c<<4 | d<<6.
template <int CTRL>
__device__ __forceinline__ float qperm(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int qperm_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
constexpr int Q_SHR1 = 0x90;   // [0,0,1,2]  lane s reads lane s-1
constexpr int Q_SHR2 = 0x44;   // [0,1,0,1]  lane s reads lane s-2
constexpr int Q_BC3 = 0xFF;    // [3,3,3,3]
constexpr int Q_XOR1 = 0xB1;   // [1,0,3,2]
constexpr int Q_XOR2 = 0x4E;   // [2,3,0,1]

__device__ __forceinline__ float quad_sum(float v) {
  v += qperm<Q_XOR1>(v);
  v += qperm<Q_XOR2>(v);
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, qperm<Q_XOR1>(v));
  v = fmaxf(v, qperm<Q_XOR2>(v));
  return v;
}
__device__ __forceinline__ int quad_max_i(int v) {
  v = max(v, qperm_i<Q_XOR1>(v));
  v = max(v, qperm_i<Q_XOR2>(v));
  return v;
}
// inclusive prefix product over the quad (lane s gets x_0 * ... * x_s)
__device__ __forceinline__ float quad_scan_mul(float x, int sub) {
  const float p1 = qperm<Q_SHR1>(x);
  x *= (sub >= 1) ? p1 : 1.0f;
  const float p2 = qperm<Q_SHR2>(x);
  x *= (sub >= 2) ? p2 : 1.0f;
  return x;
}

// Development aid, compiled in ONLY with -DGSR_ABLATE_BUILD (tools/build_variant.sh): the environment
// variable GSR_ABLATE=<bits> then disables parts of the render kernels to attribute time (1 no global
// atomics, 2 no cross-lane reduction, 4 no culling, 8 no blend loop; results are wrong with any bit
// set). The product library ignores the variable: the flags are the constant 0.
#ifdef GSR_ABLATE_BUILD
inline int ablate_flags() {
  static const int v = [] { const char* e = getenv("GSR_ABLATE"); return e ? atoi(e) : 0; }();
  return v;
}
#else
constexpr int ablate_flags() { return 0; }
#endif

struct Entry {
  float2 xy;
  float4 co;
  float4 rgb;
};

// Branch-free gathers (indices are clamped by the caller): straight-line loads let the
// compiler wait with counted vmcnt instead of draining the prefetch.
__device__ __forceinline__ Entry load_records(uint32_t idx, const float2* __restrict__ xy,
                                              const float4* __restrict__ conic_opacity,
                                              const float4* __restrict__ rgb) {
  Entry e;
  e.xy = xy[idx];
  e.co = conic_opacity[idx];
  e.rgb = rgb[idx];
  return e;
}

struct WaveGeom {
  int wave, lane, sub, pix;
  int bx0, by0;      // the wave's pixel block
  int px, py;
  bool inside;
};

__device__ __forceinline__ WaveGeom wave_geometry(int gx, int W, int H, int tile) {
  WaveGeom g;
  const int quadrant = blockIdx.x & 3;
  g.wave = threadIdx.x / GSR_WAVE;
  g.lane = threadIdx.x & (GSR_WAVE - 1);
  g.sub = g.lane & (LPP - 1);
  g.pix = g.lane >> 2;
  g.bx0 = (tile % gx) * GSR_TILE + (quadrant & 1) * 8 + (g.wave & 1) * SUB;
  g.by0 = (tile / gx) * GSR_TILE + (quadrant >> 1) * 8 + (g.wave >> 1) * SUB;
  g.px = g.bx0 + (g.pix & (SUB - 1));
  g.py = g.by0 + (g.pix >> 2);
  g.inside = (g.px < W) && (g.py < H);
  return g;
}

__global__ void __launch_bounds__(GSR_TILE_PIX)
render_fwd_kernel(int W, int H, int gx, int64_t max_pairs, const uint32_t* __restrict__ tile_order,
                  const uint32_t* __restrict__ tile_offset,
                  const uint32_t* __restrict__ point_list, const float2* __restrict__ xy,
                  const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb,
                  const float* __restrict__ bg, float* __restrict__ out_color,
                  float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int flags,
                  size_t ws_stride) {
  {   // batched launch: blockIdx.y = frame
    const size_t off = (size_t)blockIdx.y * ws_stride;
    tile_order = shift(tile_order, off);
    tile_offset = shift(tile_offset, off); point_list = shift(point_list, off); xy = shift(xy, off);
    conic_opacity = shift(conic_opacity, off); rgb = shift(rgb, off);
    final_T = reinterpret_cast<float*>(reinterpret_cast<char*>(final_T) + off);
    n_contrib = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(n_contrib) + off);
    out_color += (size_t)blockIdx.y * 3 * H * W;
  }
  __shared__ float2 s_xy[WAVES][GSR_WAVE + PAD];
  __shared__ float4 s_co[WAVES][GSR_WAVE + PAD];
  __shared__ float4 s_rgb[WAVES][GSR_WAVE + PAD];
  __shared__ int s_k[WAVES][GSR_WAVE + PAD];
  // blocks walk the tiles in the binning's size order (longest lists first, tile_scan_kernel): the long
  // per-pixel chains of the avatar's interior start at once instead of forming the launch's tail
  const int tile = (int)tile_order[blockIdx.x >> 2];
  const WaveGeom g = wave_geometry(gx, W, H, tile);
  const int wave = g.wave, lane = g.lane, sub = g.sub;
  const int64_t start = min((int64_t)tile_offset[tile], max_pairs);
  const int64_t end = min((int64_t)tile_offset[tile + 1], max_pairs);
  const int n = (int)(end - start);
  const float fpx = (float)g.px, fpy = (float)g.py;
  const float fbx = (float)g.bx0, fby = (float)g.by0;

  // per pixel (replicated over the quad): still accumulating? transmittance after the steps so far
  bool alive = g.inside;
  float Tstep = 1.0f;
  // per lane: partial colour over "its" entries, deepest contributor, T in front of a stopping entry
  float C0 = 0.f, C1 = 0.f, C2 = 0.f;
  int last = 0;
  float Tstop = -1.0f;
  if (n > 0) {
    // Software pipeline over batches of 64 list entries:
    //   iteration b: (1) cull + compact batch b into LDS (its records were requested during
    //   iteration b-1), (2) request the records of batch b+1 (its indices were requested during
    //   iteration b-1) and the indices of batch b+2, (3) blend batch b from LDS while (2) flies.
    const uint32_t* plist = point_list + start;
    uint32_t idx_cur = plist[min(lane, n - 1)];
    Entry cur = load_records(idx_cur, xy, conic_opacity, rgb);
    uint32_t idx_nxt = plist[min(GSR_WAVE + lane, n - 1)];
    for (int b0 = 0; b0 < n; b0 += GSR_WAVE) {
      if (__ballot(alive) == 0ull) break;            // every pixel of the block has saturated
      const bool keep = (b0 + lane < n) && ((flags & 4) || may_touch(cur.xy, cur.co, fbx, fby));
      const unsigned long long mask = __ballot(keep);
      const int cnt = __popcll(mask);
      if (keep) {
        const int pos = lane_rank(mask);
        s_xy[wave][pos] = cur.xy;
        s_co[wave][pos] = cur.co;
        s_rgb[wave][pos] = cur.rgb;
        s_k[wave][pos] = b0 + lane;
      }
      if (lane < PAD) {     // null entries (opacity 0) pad the list to a multiple of PAD
        s_xy[wave][cnt + lane] = make_float2(0.f, 0.f);
        s_co[wave][cnt + lane] = make_float4(1.f, 0.f, 1.f, 0.f);
        s_rgb[wave][cnt + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_k[wave][cnt + lane] = 0;
      }
      cur = load_records(idx_nxt, xy, conic_opacity, rgb);
      idx_nxt = plist[min(b0 + 2 * GSR_WAVE + lane, n - 1)];
      // LDS traffic of one wave is ordered; no workgroup barrier needed for a wave-private slice
      __builtin_amdgcn_wave_barrier();
      for (int t0 = 0; t0 < cnt && !(flags & 8); t0 += PAD) {
        if (__ballot(alive) == 0ull) break;
        float a[UNR];
        float4 col[UNR];
        int kk[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {          // independent loads / exp chains
          const int e = t0 + u * LPP + sub;
          const float2 c = s_xy[wave][e];
          const float4 co = s_co[wave][e];
          col[u] = s_rgb[wave][e];
          kk[u] = s_k[wave][e];
          const float power = eval_power(co, c.x - fpx, c.y - fpy);
          const float alpha = fminf(ALPHA_MAX, co.w * __expf(power));
          a[u] = ((power <= 0.0f) & (alpha >= ALPHA_MIN)) ? alpha : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {          // list order: step u, then lane order in the quad
          const float au = alive ? a[u] : 0.f;
          const float inc = quad_scan_mul(1.0f - au, sub);       // prod_{r<=sub} (1 - a_r)
          const float inc_prev = qperm<Q_SHR1>(inc);   // (DPP must run with the whole quad active)
          const float exc = (sub >= 1) ? inc_prev : 1.0f;
          const float Tbefore = Tstep * exc;
          const bool stop = (au > 0.f) & (Tstep * inc < T_EPS);
          const unsigned qmask = (unsigned)(__ballot(stop) >> (lane & ~(LPP - 1))) & 0xFu;
          const int first = qmask ? (__ffs((int)qmask) - 1) : LPP;      // first stopping entry
          const bool upd = (au > 0.f) & (sub < first);
          const float w = upd ? au * Tbefore : 0.f;
          C0 = fmaf(col[u].x, w, C0);
          C1 = fmaf(col[u].y, w, C1);
          C2 = fmaf(col[u].z, w, C2);
          last = upd ? kk[u] + 1 : last;
          Tstop = (sub == first) ? Tbefore : Tstop;               // T in front of the stopping entry
          Tstep *= qperm<Q_BC3>(inc);
          alive = alive & (first == LPP);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // combine the quad: colour = sum, deepest contributor = max, final T
  C0 = quad_sum(C0); C1 = quad_sum(C1); C2 = quad_sum(C2);
  last = quad_max_i(last);
  const float tstop = quad_max(Tstop);
  const float T = tstop >= 0.f ? tstop : Tstep;
  if (g.inside && sub == 0) {
    const size_t pix = (size_t)g.py * W + g.px;
    const size_t plane = (size_t)H * W;
    final_T[pix] = T;
    n_contrib[pix] = (uint32_t)last;
    out_color[pix] = fmaf(T, bg[0], C0);
    out_color[plane + pix] = fmaf(T, bg[1], C1);
    out_color[2 * plane + pix] = fmaf(T, bg[2], C2);
  }
}

// ------------------------------------------------------------------------------------ backward
constexpr int ACC_ROWS = LPP * 9;    // one step parks 4 entries x 9 gradient components
constexpr int ACC_ROW = 17;          // 16 pixels + 1 pad: column reads by 36 lanes are conflict-free

__global__ void __launch_bounds__(GSR_TILE_PIX)
render_bwd_kernel(int W, int H, int gx, int64_t max_pairs, const uint32_t* __restrict__ tile_order,
                  const uint32_t* __restrict__ tile_offset,
                  const uint32_t* __restrict__ point_list, const float2* __restrict__ xy,
                  const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb,
                  const float* __restrict__ bg, const float* __restrict__ final_T,
                  const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dout,
                  float* __restrict__ grad_acc, int flags, size_t ws_stride) {
  {   // batched launch: blockIdx.y = frame
    const size_t off = (size_t)blockIdx.y * ws_stride;
    tile_order = shift(tile_order, off);
    tile_offset = shift(tile_offset, off); point_list = shift(point_list, off); xy = shift(xy, off);
    conic_opacity = shift(conic_opacity, off); rgb = shift(rgb, off); final_T = shift(final_T, off);
    n_contrib = shift(n_contrib, off);
    grad_acc = reinterpret_cast<float*>(reinterpret_cast<char*>(grad_acc) + off);
    dL_dout += (size_t)blockIdx.y * 3 * H * W;
  }
  __shared__ float2 s_xy[WAVES][GSR_WAVE + PAD];
  __shared__ float4 s_co[WAVES][GSR_WAVE + PAD];
  __shared__ float4 s_rgb[WAVES][GSR_WAVE + PAD];
  __shared__ int s_k[WAVES][GSR_WAVE + PAD];
  __shared__ uint32_t s_idx[WAVES][GSR_WAVE + PAD];
  __shared__ float s_acc[WAVES][ACC_ROWS][ACC_ROW];
  const int tile = (int)tile_order[blockIdx.x >> 2];      // size order, as in the forward kernel
  const WaveGeom g = wave_geometry(gx, W, H, tile);
  const int wave = g.wave, lane = g.lane, sub = g.sub, pixi = g.pix;
  const int64_t start = min((int64_t)tile_offset[tile], max_pairs);
  const int64_t end = min((int64_t)tile_offset[tile + 1], max_pairs);
  const int n = (int)(end - start);
  if (n <= 0) return;
  const size_t pix = (size_t)g.py * W + g.px;
  const size_t plane = (size_t)H * W;
  const int last = g.inside ? (int)n_contrib[pix] : 0;
  // entries beyond the deepest contributor of any pixel of the block are never needed
  int wmax = last;
#pragma unroll
  for (int off = GSR_WAVE / 2; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off));
  wmax = min(wmax, n);
  if (wmax == 0) return;
  const float Tf = g.inside ? final_T[pix] : 0.f;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (g.inside) {
    g0 = dL_dout[pix];
    g1 = dL_dout[plane + pix];
    g2 = dL_dout[2 * plane + pix];
  }
  const float bg_dot_g = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
  const float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;
  const float fpx = (float)g.px, fpy = (float)g.py;
  const float fbx = (float)g.bx0, fby = (float)g.by0;

  // per pixel (replicated over the quad), walking the list back to front:
  //   Tstep  transmittance in front of the entries processed so far (starts at final_T)
  //   S0..2  colour accumulated behind: S <- alpha c + (1 - alpha) S for every contributing entry
  float Tstep = Tf;
  float S0 = 0.f, S1 = 0.f, S2 = 0.f;
  const int nbatch = (wmax + GSR_WAVE - 1) / GSR_WAVE;
  // same software pipeline as the forward pass, walking the batches back to front
  const uint32_t* plist = point_list + start;
  uint32_t idx_cur = plist[min((nbatch - 1) * GSR_WAVE + lane, wmax - 1)];
  Entry cur = load_records(idx_cur, xy, conic_opacity, rgb);
  uint32_t idx_nxt = plist[max(min((nbatch - 2) * GSR_WAVE + lane, wmax - 1), 0)];
  for (int b = nbatch - 1; b >= 0; --b) {
    const int b0 = b * GSR_WAVE;
    const bool keep = (b0 + lane < wmax) && ((flags & 4) || may_touch(cur.xy, cur.co, fbx, fby));
    const unsigned long long mask = __ballot(keep);
    const int cnt = __popcll(mask);
    // stage the survivors in REVERSE list order (position 0 = deepest), so that the blend loop
    // below walks forward through LDS exactly like the forward pass does
    if (keep) {
      const int pos = cnt - 1 - lane_rank(mask);
      s_xy[wave][pos] = cur.xy;
      s_co[wave][pos] = cur.co;
      s_rgb[wave][pos] = cur.rgb;
      s_k[wave][pos] = b0 + lane;
      s_idx[wave][pos] = idx_cur;
    }
    if (lane < PAD) {     // null entries (opacity 0, beyond every pixel's range)
      s_xy[wave][cnt + lane] = make_float2(0.f, 0.f);
      s_co[wave][cnt + lane] = make_float4(1.f, 0.f, 1.f, 0.f);
      s_rgb[wave][cnt + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
      s_k[wave][cnt + lane] = 0x7fffffff;
      s_idx[wave][cnt + lane] = 0;
    }
    idx_cur = idx_nxt;
    cur = load_records(idx_nxt, xy, conic_opacity, rgb);
    idx_nxt = plist[max(min(b0 - 2 * GSR_WAVE + lane, wmax - 1), 0)];
    __builtin_amdgcn_wave_barrier();
    for (int t0 = 0; t0 < cnt && !(flags & 8); t0 += PAD) {
      float2 c[UNR];
      float4 co[UNR], col[UNR];
      float G[UNR], a[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int e = t0 + u * LPP + sub;
        c[u] = s_xy[wave][e];
        co[u] = s_co[wave][e];
        col[u] = s_rgb[wave][e];
        const int k = s_k[wave][e];   // 0-based list position; forward counted it as contributor k+1
        const float power = eval_power(co[u], c[u].x - fpx, c[u].y - fpy);
        G[u] = __expf(power);
        const float alpha = fminf(ALPHA_MAX, co[u].w * G[u]);
        a[u] = ((k < last) & (power <= 0.0f) & (alpha >= ALPHA_MIN)) ? alpha : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (__ballot(a[u] > 0.f) == 0ull) continue;   // nobody in the wave is touched by these 4
        const float au = a[u];
        const bool h = au > 0.f;
        // transmittance in front of this entry: Tstep * prod_{r<=sub} 1/(1 - a_r)
        const float rc = __builtin_amdgcn_rcpf(1.0f - au);          // a <= 0.99
        const float incr = quad_scan_mul(rc, sub);
        const float T = Tstep * incr;
        // colour behind this entry: exclusive composition of the affine maps S -> A S + B of the
        // quad's earlier (deeper) entries, applied to the pixel's S
        float A = 1.0f - au, B0 = au * col[u].x, B1 = au * col[u].y, B2 = au * col[u].z;
        {
          const float pA = qperm<Q_SHR1>(A), p0 = qperm<Q_SHR1>(B0), p1 = qperm<Q_SHR1>(B1),
                      p2 = qperm<Q_SHR1>(B2);
          if (sub >= 1) { B0 = fmaf(A, p0, B0); B1 = fmaf(A, p1, B1); B2 = fmaf(A, p2, B2); A *= pA; }
        }
        {
          const float pA = qperm<Q_SHR2>(A), p0 = qperm<Q_SHR2>(B0), p1 = qperm<Q_SHR2>(B1),
                      p2 = qperm<Q_SHR2>(B2);
          if (sub >= 2) { B0 = fmaf(A, p0, B0); B1 = fmaf(A, p1, B1); B2 = fmaf(A, p2, B2); A *= pA; }
        }
        // exclusive = inclusive of the lane before
        const float xA = qperm<Q_SHR1>(A), x0 = qperm<Q_SHR1>(B0), x1 = qperm<Q_SHR1>(B1),
                    x2 = qperm<Q_SHR1>(B2);
        const float eA = (sub >= 1) ? xA : 1.0f;
        const float e0 = (sub >= 1) ? x0 : 0.0f;
        const float e1 = (sub >= 1) ? x1 : 0.0f;
        const float e2 = (sub >= 1) ? x2 : 0.0f;
        const float acc0 = fmaf(eA, S0, e0), acc1 = fmaf(eA, S1, e1), acc2 = fmaf(eA, S2, e2);
        // advance the pixel state past the whole quad step
        const float tA = qperm<Q_BC3>(A);
        S0 = fmaf(tA, S0, qperm<Q_BC3>(B0));
        S1 = fmaf(tA, S1, qperm<Q_BC3>(B1));
        S2 = fmaf(tA, S2, qperm<Q_BC3>(B2));
        Tstep *= qperm<Q_BC3>(incr);
        // this lane's (pixel, entry) gradient contributions — zero when not hit
        const float dx = c[u].x - fpx, dy = c[u].y - fpy;
        const float w = au * T;
        float dL_dalpha = ((col[u].x - acc0) * g0 + (col[u].y - acc1) * g1 + (col[u].z - acc2) * g2) * T;
        dL_dalpha += (-Tf * rc) * bg_dot_g;
        dL_dalpha = h ? dL_dalpha : 0.f;
        const float dL_dG = co[u].w * dL_dalpha;
        const float gdx = G[u] * dx, gdy = G[u] * dy;
        const float dG_ddx = -gdx * co[u].x - gdy * co[u].y;
        const float dG_ddy = -gdy * co[u].z - gdx * co[u].y;
        float v[9];
        v[0] = dL_dG * dG_ddx * half_w;
        v[1] = dL_dG * dG_ddy * half_h;
        v[2] = -0.5f * gdx * dx * dL_dG;
        v[3] = -gdx * dy * dL_dG;
        v[4] = -0.5f * gdy * dy * dL_dG;
        v[5] = G[u] * dL_dalpha;
        v[6] = w * g0;
        v[7] = w * g1;
        v[8] = w * g2;
        if (!(flags & 2)) {
          // sum over the 16 pixels of the block: lane (pixel p, entry s) parks component q at
          // row s*9+q, column p; lane r < 36 then sums row r and issues the atomic for it
#pragma unroll
          for (int q = 0; q < 9; ++q) s_acc[wave][sub * 9 + q][pixi] = v[q];
          __builtin_amdgcn_wave_barrier();
          if (lane < ACC_ROWS) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              s0 += s_acc[wave][lane][i];
              s1 += s_acc[wave][lane][i + 1];
            }
            const int es = lane / 9, q = lane - es * 9;
            const uint32_t gi = s_idx[wave][t0 + u * LPP + es];
            const float tot = s0 + s1;
            if (tot != 0.f && !(flags & 1))
              unsafeAtomicAdd(&grad_acc[(size_t)gi * GSR_GRAD_STRIDE + q], tot);
          }
          __builtin_amdgcn_wave_barrier();
        } else {
          float mine = 0.f;
#pragma unroll
          for (int q = 0; q < 9; ++q) mine += v[q];
          if (mine == 123.456f) grad_acc[0] = mine;   // keep the values alive
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

hipError_t launch_render_fwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             float* out_color, const Batch& bt, hipStream_t stream) {
  if (d.T == 0) return hipSuccess;
  {
    ProfScope prof_(K_RENDER_FWD, stream);
    hipLaunchKernelGGL(render_fwd_kernel, dim3(4 * d.T, bt.frames), dim3(GSR_TILE_PIX), 0, stream, d.W,
                       d.H, d.gx, d.max_pairs, ws.tile_count, ws.tile_offset, ws.point_list, ws.xy,
                       ws.conic_opacity, ws.rgb, s.bg, out_color, ws.final_T, ws.n_contrib, ablate_flags(),
                       bt.ws_stride);
  }
  return hipGetLastError();
}

hipError_t launch_render_bwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             const float* dL_dout, const Batch& bt, hipStream_t stream) {
  if (d.T == 0 || d.P == 0) return hipSuccess;
  {
    ProfScope prof_(K_RENDER_BWD, stream);
    hipLaunchKernelGGL(render_bwd_kernel, dim3(4 * d.T, bt.frames), dim3(GSR_TILE_PIX), 0, stream, d.W,
                       d.H, d.gx, d.max_pairs, ws.tile_count, ws.tile_offset, ws.point_list, ws.xy,
                       ws.conic_opacity, ws.rgb, s.bg, ws.final_T, ws.n_contrib, dL_dout, ws.grad_acc,
                       ablate_flags(), bt.ws_stride);
  }
  return hipGetLastError();
}

}  // namespace gsr
