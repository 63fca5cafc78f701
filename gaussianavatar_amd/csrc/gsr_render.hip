// gsr_render.hip — K5 render_fwd and K6 render_bwd (per-tile alpha compositing).
//
// Work decomposition (gfx950: wave64, few active tiles, long per-tile lists):
//   one 256-thread workgroup per 16x16 tile, but each of its 4 waves is an INDEPENDENT unit that
//   owns one 8x8 pixel quadrant — no __syncthreads anywhere. A wave walks the tile's depth-sorted
//   list 64 entries at a time: lane l gathers entry l (index -> 16-byte SoA records), tests the
//   Gaussian's exact-conservative screen-space bounding box against the quadrant, and the
//   survivors are compacted (ballot + mbcnt) into the wave's private LDS slice. All 64 lanes
//   (= the 64 pixels) then walk the compacted entries in lock-step with broadcast LDS reads.
//   The next 64 entries are gathered into registers while the current ones are blended.
//   Compared with the 256-entry cooperative batches of the textbook design this removes the
//   barriers, shortens every pixel's serial chain to the entries that can touch its quadrant
//   (the culling never drops an entry with alpha >= 1/255 anywhere in the quadrant, so results
//   are unchanged), and lets a finished quadrant retire without waiting for its neighbours.
// Spec: SURVEY.md Appendix A.3 (forward) and A.4 (backward).
#include <cstdlib>

#include "gsr_common.h"

namespace gsr {

namespace {

constexpr int QUAD = 8;                        // quadrant edge in pixels (one wave64)
constexpr int WAVES = GSR_TILE_PIX / GSR_WAVE;  // 4
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
// box [x0,x0+7] x [y0,y0+7]?  alpha >= 1/255  <=>  d^T Q d <= tau, tau = 2 ln(255 o); the
// axis-aligned bounding box of that ellipse has half extents sqrt(tau * Sigma_xx), sqrt(tau *
// Sigma_yy) with Sigma = Q^-1. The test is conservative (slightly inflated, NaN -> keep).
__device__ __forceinline__ bool may_touch(float2 c, float4 co, float x0, float y0) {
  const float tau = 2.0f * __logf(255.0f * co.w) + 1e-3f;   // margin for the fast log
  const float det = co.x * co.z - co.y * co.y;
  const float inv = 1.0f / det;
  const float hx = sqrtf(tau * co.z * inv) * 1.001f + 0.01f;
  const float hy = sqrtf(tau * co.x * inv) * 1.001f + 0.01f;
  const bool outside = (c.x + hx < x0) || (c.x - hx > x0 + (float)(QUAD - 1)) ||
                       (c.y + hy < y0) || (c.y - hy > y0 + (float)(QUAD - 1)) || (tau < 0.0f);
  return !outside;
}

// Number of set bits of `mask` below this lane.
__device__ __forceinline__ int lane_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// Sum over the 64 lanes of a wave with DPP row operations; the total lands in lane 63.
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, BOUND);
  return v + __int_as_float(moved);
}

__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0xb1, 0xf, true>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4e, 0xf, true>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x114, 0xf, true>(v);   // row_shr:4
  v = dpp_add<0x118, 0xf, true>(v);   // row_shr:8
  v = dpp_add<0x142, 0xa, false>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xc, false>(v);  // row_bcast:31 -> rows 2,3
  return v;
}

// Development aid: GSR_ABLATE=<bits> disables parts of the render kernels to attribute time
// (1 no global atomics, 2 no cross-lane reduction, 4 no quadrant culling, 8 no blend loop).
// Results are wrong with any bit set; never set in production.
inline int ablate_flags() {
  static const int v = [] { const char* e = getenv("GSR_ABLATE"); return e ? atoi(e) : 0; }();
  return v;
}

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

constexpr int ILP = 4;   // entries evaluated together (independent LDS reads / exp chains)
constexpr int ACC_SLOTS = 7;    // entries between gradient flushes: 7 x 9 = 63 rows <= 64 lanes
constexpr int ACC_ROW = 34;     // 32 pair-sums + pad: row stride 34 floats is conflict-free for
                                // 64-bit column reads (17 r mod 32 is a bijection)

// Lane r sums row r (32 floats) of the wave's accumulator and adds it to the Gaussian's
// screen-space gradient record: row r = (slot r / 9, component r % 9).
__device__ __forceinline__ void flush_rows(float (*acc)[ACC_ROW], const uint32_t* slot_idx,
                                           int slots, int lane, float* __restrict__ grad_acc,
                                           int flags) {
  __builtin_amdgcn_wave_barrier();
  if (lane < slots * 9) {
    const float2* row = reinterpret_cast<const float2*>(acc[lane]);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float2 p = row[i];
      s0 += p.x;
      s1 += p.y;
    }
    const int e = lane / 9, q = lane - e * 9;
    if (!(flags & 1))
      unsafeAtomicAdd(&grad_acc[(size_t)slot_idx[e] * GSR_GRAD_STRIDE + q], s0 + s1);
  }
  __builtin_amdgcn_wave_barrier();
}

__global__ void __launch_bounds__(GSR_TILE_PIX)
render_fwd_kernel(int W, int H, int gx, int64_t max_pairs,
                  const uint32_t* __restrict__ tile_offset,
                  const uint32_t* __restrict__ point_list, const float2* __restrict__ xy,
                  const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb,
                  const float* __restrict__ bg, float* __restrict__ out_color,
                  float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int flags) {
  __shared__ float2 s_xy[WAVES][GSR_WAVE + ILP];
  __shared__ float4 s_co[WAVES][GSR_WAVE + ILP];
  __shared__ float4 s_rgb[WAVES][GSR_WAVE + ILP];
  __shared__ int s_k[WAVES][GSR_WAVE + ILP];
  const int tile = blockIdx.x;
  const int wave = threadIdx.x / GSR_WAVE;
  const int lane = threadIdx.x & (GSR_WAVE - 1);
  const int qx0 = (tile % gx) * GSR_TILE + (wave & 1) * QUAD;
  const int qy0 = (tile / gx) * GSR_TILE + (wave >> 1) * QUAD;
  const int px = qx0 + (lane & (QUAD - 1));
  const int py = qy0 + (lane >> 3);
  const bool inside = (px < W) && (py < H);
  const int64_t start = min((int64_t)tile_offset[tile], max_pairs);
  const int64_t end = min((int64_t)tile_offset[tile + 1], max_pairs);
  const int n = (int)(end - start);
  const float fpx = (float)px, fpy = (float)py;
  const float fqx = (float)qx0, fqy = (float)qy0;

  bool done = !inside;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last = 0;
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
      if (__ballot(!done) == 0ull) break;            // the whole quadrant has saturated
      const bool keep = (b0 + lane < n) && ((flags & 4) || may_touch(cur.xy, cur.co, fqx, fqy));
      const unsigned long long mask = __ballot(keep);
      const int cnt = __popcll(mask);
      if (keep) {
        const int pos = lane_rank(mask);
        s_xy[wave][pos] = cur.xy;
        s_co[wave][pos] = cur.co;
        s_rgb[wave][pos] = cur.rgb;
        s_k[wave][pos] = b0 + lane;
      }
      if (lane < ILP) {     // null entries (opacity 0) pad the list to a multiple of ILP
        s_xy[wave][cnt + lane] = make_float2(0.f, 0.f);
        s_co[wave][cnt + lane] = make_float4(1.f, 0.f, 1.f, 0.f);
        s_rgb[wave][cnt + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_k[wave][cnt + lane] = 0;
      }
      cur = load_records(idx_nxt, xy, conic_opacity, rgb);
      idx_nxt = plist[min(b0 + 2 * GSR_WAVE + lane, n - 1)];
      // LDS traffic of one wave is ordered; no workgroup barrier needed for a wave-private slice
      __builtin_amdgcn_wave_barrier();
      for (int t0 = 0; t0 < cnt && !(flags & 8); t0 += ILP) {
        if (__ballot(!done) == 0ull) break;
        // evaluate ILP entries together (independent LDS reads and exp chains), then apply
        // them in list order with selects — no divergent branches in this loop
        float power[ILP], alpha[ILP];
        float4 col[ILP];
        int kk[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
          const float2 c = s_xy[wave][t0 + u];
          const float4 co = s_co[wave][t0 + u];
          col[u] = s_rgb[wave][t0 + u];
          kk[u] = s_k[wave][t0 + u];
          power[u] = eval_power(co, c.x - fpx, c.y - fpy);
          alpha[u] = fminf(ALPHA_MAX, co.w * __expf(power[u]));
        }
        // transmittance before each entry as a short multiply chain (an entry that does not
        // contribute has a = 0); everything else hangs off it with selects
        float a[ILP], Tpre[ILP + 1];
        Tpre[0] = T;
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
          a[u] = ((power[u] <= 0.0f) & (alpha[u] >= ALPHA_MIN)) ? alpha[u] : 0.f;
          Tpre[u + 1] = Tpre[u] * (1.0f - a[u]);
        }
        bool alive = !done;
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
          const bool contributes = a[u] > 0.f;
          const bool stop = contributes & (Tpre[u + 1] < T_EPS);
          const bool upd = alive & contributes & !stop;
          const float w = upd ? a[u] * Tpre[u] : 0.f;
          C0 = fmaf(col[u].x, w, C0);
          C1 = fmaf(col[u].y, w, C1);
          C2 = fmaf(col[u].z, w, C2);
          T = upd ? Tpre[u + 1] : T;
          last = upd ? (uint32_t)kk[u] + 1u : last;
          alive = alive & !stop;
        }
        done = !alive;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (inside) {
    const size_t pix = (size_t)py * W + px;
    const size_t plane = (size_t)H * W;
    final_T[pix] = T;
    n_contrib[pix] = last;
    out_color[pix] = fmaf(T, bg[0], C0);
    out_color[plane + pix] = fmaf(T, bg[1], C1);
    out_color[2 * plane + pix] = fmaf(T, bg[2], C2);
  }
}

__global__ void __launch_bounds__(GSR_TILE_PIX)
render_bwd_kernel(int W, int H, int gx, int64_t max_pairs,
                  const uint32_t* __restrict__ tile_offset,
                  const uint32_t* __restrict__ point_list, const float2* __restrict__ xy,
                  const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb,
                  const float* __restrict__ bg, const float* __restrict__ final_T,
                  const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dout,
                  float* __restrict__ grad_acc, int flags) {
  __shared__ float2 s_xy[WAVES][GSR_WAVE + ILP];
  __shared__ float4 s_co[WAVES][GSR_WAVE + ILP];
  __shared__ float4 s_rgb[WAVES][GSR_WAVE + ILP];
  __shared__ int s_k[WAVES][GSR_WAVE + ILP];
  __shared__ uint32_t s_idx[WAVES][GSR_WAVE + ILP];
  __shared__ float s_acc[WAVES][ACC_SLOTS * 9][ACC_ROW];
  __shared__ uint32_t s_slot_idx[WAVES][ACC_SLOTS + 1];
  const int tile = blockIdx.x;
  const int wave = threadIdx.x / GSR_WAVE;
  const int lane = threadIdx.x & (GSR_WAVE - 1);
  const int qx0 = (tile % gx) * GSR_TILE + (wave & 1) * QUAD;
  const int qy0 = (tile / gx) * GSR_TILE + (wave >> 1) * QUAD;
  const int px = qx0 + (lane & (QUAD - 1));
  const int py = qy0 + (lane >> 3);
  const bool inside = (px < W) && (py < H);
  const int64_t start = min((int64_t)tile_offset[tile], max_pairs);
  const int64_t end = min((int64_t)tile_offset[tile + 1], max_pairs);
  const int n = (int)(end - start);
  if (n <= 0) return;
  const size_t pix = (size_t)py * W + px;
  const size_t plane = (size_t)H * W;
  const int last = inside ? (int)n_contrib[pix] : 0;
  // entries beyond the deepest contributor of any pixel of the quadrant are never needed
  int wmax = last;
#pragma unroll
  for (int off = GSR_WAVE / 2; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off));
  wmax = min(wmax, n);
  if (wmax == 0) return;
  const float Tf = inside ? final_T[pix] : 0.f;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (inside) {
    g0 = dL_dout[pix];
    g1 = dL_dout[plane + pix];
    g2 = dL_dout[2 * plane + pix];
  }
  const float bg_dot_g = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
  const float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;
  const float fpx = (float)px, fpy = (float)py;
  const float fqx = (float)qx0, fqy = (float)qy0;
  float T = Tf;
  int slot = 0;   // entries parked in s_acc since the last flush (wave-uniform)
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
  const int nbatch = (wmax + GSR_WAVE - 1) / GSR_WAVE;
  // same software pipeline as the forward pass, walking the batches back to front
  const uint32_t* plist = point_list + start;
  uint32_t idx_cur = plist[min((nbatch - 1) * GSR_WAVE + lane, wmax - 1)];
  Entry cur = load_records(idx_cur, xy, conic_opacity, rgb);
  uint32_t idx_nxt = plist[max(min((nbatch - 2) * GSR_WAVE + lane, wmax - 1), 0)];
  for (int b = nbatch - 1; b >= 0; --b) {
    const int b0 = b * GSR_WAVE;
    const bool keep = (b0 + lane < wmax) && ((flags & 4) || may_touch(cur.xy, cur.co, fqx, fqy));
    const unsigned long long mask = __ballot(keep);
    const int cnt = __popcll(mask);
    if (keep) {
      const int pos = lane_rank(mask);
      s_xy[wave][pos] = cur.xy;
      s_co[wave][pos] = cur.co;
      s_rgb[wave][pos] = cur.rgb;
      s_k[wave][pos] = b0 + lane;
      s_idx[wave][pos] = idx_cur;
    }
    if (lane < ILP) {     // null entries (opacity 0) pad the list to a multiple of ILP
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
    for (int t0 = (cnt - 1) & ~(ILP - 1); t0 >= 0 && cnt > 0 && !(flags & 8); t0 -= ILP) {
      float2 c[ILP];
      float4 co[ILP], col[ILP];
      float power[ILP], G[ILP], alpha[ILP];
      bool hit[ILP];
#pragma unroll
      for (int u = 0; u < ILP; ++u) {
        c[u] = s_xy[wave][t0 + u];
        co[u] = s_co[wave][t0 + u];
        col[u] = s_rgb[wave][t0 + u];
        const int k = s_k[wave][t0 + u];   // 0-based list position; forward counted it as k+1
        power[u] = eval_power(co[u], c[u].x - fpx, c[u].y - fpy);
        G[u] = __expf(power[u]);
        alpha[u] = fminf(ALPHA_MAX, co[u].w * G[u]);
        hit[u] = (k < last) & (power[u] <= 0.0f) & (alpha[u] >= ALPHA_MIN);
      }
#pragma unroll
      for (int uu = 0; uu < ILP; ++uu) {
        const int u = ILP - 1 - uu;           // back to front inside the group
        const int t = t0 + u;
        if (__ballot(hit[u]) == 0ull) continue;   // wave-uniform
        // predicated update: a lane that is not hit runs with alpha = 0, which leaves T and
        // every accumulator unchanged and yields zero gradient contributions
        const bool h = hit[u];
        const float a = h ? alpha[u] : 0.f;
        const float dx = c[u].x - fpx, dy = c[u].y - fpy;
        const float rcp_1ma = __builtin_amdgcn_rcpf(1.0f - a);   // a <= 0.99
        T = T * rcp_1ma;
        const float w = a * T;
        const float n0 = last_alpha * lc0 + (1.0f - last_alpha) * acc0;
        const float n1 = last_alpha * lc1 + (1.0f - last_alpha) * acc1;
        const float n2 = last_alpha * lc2 + (1.0f - last_alpha) * acc2;
        acc0 = h ? n0 : acc0; acc1 = h ? n1 : acc1; acc2 = h ? n2 : acc2;
        lc0 = h ? col[u].x : lc0; lc1 = h ? col[u].y : lc1; lc2 = h ? col[u].z : lc2;
        float dL_dalpha = (col[u].x - acc0) * g0 + (col[u].y - acc1) * g1 + (col[u].z - acc2) * g2;
        dL_dalpha *= T;
        last_alpha = h ? a : last_alpha;
        dL_dalpha += (-Tf * rcp_1ma) * bg_dot_g;
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
        // Cross-lane sum without a VALU butterfly: lanes add pairwise once (DPP), the even lanes
        // park the 9 partials in LDS rows (slot, component) x 32, and every ACC_SLOTS entries
        // lane r sums row r and issues the atomics for up to 63 (entry, component) pairs at once.
        if (!(flags & 2)) {
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const float pr = dpp_add<0xb1, 0xf, true>(v[q]);   // lane ^ 1
            if ((lane & 1) == 0) s_acc[wave][slot * 9 + q][lane >> 1] = pr;
          }
          if (lane == 0) s_slot_idx[wave][slot] = s_idx[wave][t];
          if (++slot == ACC_SLOTS) {
            flush_rows(s_acc[wave], s_slot_idx[wave], slot, lane, grad_acc, flags);
            slot = 0;
          }
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
  if (slot > 0) flush_rows(s_acc[wave], s_slot_idx[wave], slot, lane, grad_acc, flags);
}

}  // namespace

hipError_t launch_render_fwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             float* out_color, hipStream_t stream) {
  if (d.T == 0) return hipSuccess;
  {
    ProfScope prof_(K_RENDER_FWD, stream);
    hipLaunchKernelGGL(render_fwd_kernel, dim3(d.T), dim3(GSR_TILE_PIX), 0, stream, d.W, d.H, d.gx,
                       d.max_pairs, ws.tile_offset, ws.point_list, ws.xy, ws.conic_opacity, ws.rgb,
                       s.bg, out_color, ws.final_T, ws.n_contrib, ablate_flags());
  }
  return hipGetLastError();
}

hipError_t launch_render_bwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             const float* dL_dout, hipStream_t stream) {
  if (d.T == 0 || d.P == 0) return hipSuccess;
  {
    ProfScope prof_(K_RENDER_BWD, stream);
    hipLaunchKernelGGL(render_bwd_kernel, dim3(d.T), dim3(GSR_TILE_PIX), 0, stream, d.W, d.H, d.gx,
                       d.max_pairs, ws.tile_offset, ws.point_list, ws.xy, ws.conic_opacity, ws.rgb,
                       s.bg, ws.final_T, ws.n_contrib, dL_dout, ws.grad_acc, ablate_flags());
  }
  return hipGetLastError();
}

}  // namespace gsr
