// gsr_render.hip — K5 render_fwd and K6 render_bwd (per-tile alpha compositing).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; wave w owns the 16x4 pixel strip
// rows 4w..4w+3. The tile's depth-sorted list is staged through LDS 256 entries at a time
// (coalesced index load + 16-byte gathers of the projected-Gaussian SoA records), then all
// lanes walk the staged entries in lock-step reading LDS with broadcast reads.
// Spec: SURVEY.md Appendix A.3 (forward) and A.4 (backward).
#include "gsr_common.h"

namespace gsr {

namespace {

constexpr int BATCH = GSR_TILE_PIX;   // entries staged per round = threads per block
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_EPS = 0.0001f;

// The same expression tree is used by forward and backward so that both take the same
// skip decisions for a given (pixel, Gaussian).
__device__ __forceinline__ float eval_power(float4 co, float dx, float dy) {
  const float q = fmaf(co.x * dx, dx, co.z * dy * dy);
  return fmaf(-0.5f, q, -(co.y * dx) * dy);
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

__global__ void __launch_bounds__(BATCH)
render_fwd_kernel(int W, int H, int gx, int64_t max_pairs,
                  const uint32_t* __restrict__ tile_offset,
                  const uint32_t* __restrict__ point_list, const float2* __restrict__ xy,
                  const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb,
                  const float* __restrict__ bg, float* __restrict__ out_color,
                  float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
  __shared__ float2 s_xy[BATCH];
  __shared__ float4 s_co[BATCH];
  __shared__ float4 s_rgb[BATCH];
  const int tile = blockIdx.x;
  const int tid = threadIdx.x;
  const int px = (tile % gx) * GSR_TILE + (tid & (GSR_TILE - 1));
  const int py = (tile / gx) * GSR_TILE + (tid >> 4);
  const bool inside = (px < W) && (py < H);
  const int64_t start = min((int64_t)tile_offset[tile], max_pairs);
  const int64_t end = min((int64_t)tile_offset[tile + 1], max_pairs);
  const int n = (int)(end - start);
  const float fpx = (float)px, fpy = (float)py;

  bool done = !inside;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t contributor = 0, last = 0;
  for (int b0 = 0; b0 < n; b0 += BATCH) {
    if (__syncthreads_count(done) == BATCH) break;
    const int k = b0 + tid;
    if (k < n) {
      const uint32_t idx = point_list[start + k];
      s_xy[tid] = xy[idx];
      s_co[tid] = conic_opacity[idx];
      s_rgb[tid] = rgb[idx];
    }
    __syncthreads();
    const int m = min(BATCH, n - b0);
    for (int j = 0; !done && j < m; ++j) {
      contributor++;
      const float2 c = s_xy[j];
      const float4 co = s_co[j];
      const float dx = c.x - fpx, dy = c.y - fpy;
      const float power = eval_power(co, dx, dy);
      if (power > 0.0f) continue;
      const float alpha = fminf(ALPHA_MAX, co.w * __expf(power));
      if (alpha < ALPHA_MIN) continue;
      const float Tn = T * (1.0f - alpha);
      if (Tn < T_EPS) { done = true; continue; }
      const float4 col = s_rgb[j];
      const float w = alpha * T;
      C0 = fmaf(col.x, w, C0);
      C1 = fmaf(col.y, w, C1);
      C2 = fmaf(col.z, w, C2);
      T = Tn;
      last = contributor;
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

__global__ void __launch_bounds__(BATCH)
render_bwd_kernel(int W, int H, int gx, int64_t max_pairs,
                  const uint32_t* __restrict__ tile_offset,
                  const uint32_t* __restrict__ point_list, const float2* __restrict__ xy,
                  const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb,
                  const float* __restrict__ bg, const float* __restrict__ final_T,
                  const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dout,
                  float* __restrict__ grad_acc) {
  __shared__ float2 s_xy[BATCH];
  __shared__ float4 s_co[BATCH];
  __shared__ float4 s_rgb[BATCH];
  __shared__ uint32_t s_idx[BATCH];
  __shared__ int s_max[BATCH / GSR_WAVE];
  const int tile = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (GSR_WAVE - 1);
  const int px = (tile % gx) * GSR_TILE + (tid & (GSR_TILE - 1));
  const int py = (tile / gx) * GSR_TILE + (tid >> 4);
  const bool inside = (px < W) && (py < H);
  const int64_t start = min((int64_t)tile_offset[tile], max_pairs);
  const int64_t end = min((int64_t)tile_offset[tile + 1], max_pairs);
  const int n = (int)(end - start);
  if (n <= 0) return;
  const size_t pix = (size_t)py * W + px;
  const size_t plane = (size_t)H * W;
  const int last = inside ? (int)n_contrib[pix] : 0;
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

  // entries beyond the deepest contributor of any pixel of the tile are never needed
  int wmax = last;
#pragma unroll
  for (int off = GSR_WAVE / 2; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off));
  if (lane == 0) s_max[tid / GSR_WAVE] = wmax;
  __syncthreads();
  int max_last = 0;
#pragma unroll
  for (int w = 0; w < BATCH / GSR_WAVE; ++w) max_last = max(max_last, s_max[w]);
  max_last = min(max_last, n);
  if (max_last == 0) return;

  float T = Tf;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
  const int nbatch = (max_last + BATCH - 1) / BATCH;
  for (int b = nbatch - 1; b >= 0; --b) {
    const int b0 = b * BATCH;
    const int m = min(BATCH, max_last - b0);
    __syncthreads();
    if (tid < m) {
      const uint32_t idx = point_list[start + b0 + tid];
      s_idx[tid] = idx;
      s_xy[tid] = xy[idx];
      s_co[tid] = conic_opacity[idx];
      s_rgb[tid] = rgb[idx];
    }
    __syncthreads();
    for (int j = m - 1; j >= 0; --j) {
      const int k = b0 + j;   // 0-based list position; forward counted it as contributor k+1
      const float2 c = s_xy[j];
      const float4 co = s_co[j];
      const float dx = c.x - fpx, dy = c.y - fpy;
      const float power = eval_power(co, dx, dy);
      const float G = __expf(power);
      const float alpha = fminf(ALPHA_MAX, co.w * G);
      const bool hit = (k < last) && (power <= 0.0f) && (alpha >= ALPHA_MIN);
      if (__ballot(hit) == 0ull) continue;   // wave-uniform
      float v_dx = 0.f, v_dy = 0.f, v_a = 0.f, v_b = 0.f, v_c = 0.f, v_o = 0.f;
      float v_r = 0.f, v_g = 0.f, v_bl = 0.f;
      if (hit) {
        const float4 col = s_rgb[j];
        T = T / (1.0f - alpha);
        const float w = alpha * T;
        acc0 = last_alpha * lc0 + (1.0f - last_alpha) * acc0;
        acc1 = last_alpha * lc1 + (1.0f - last_alpha) * acc1;
        acc2 = last_alpha * lc2 + (1.0f - last_alpha) * acc2;
        lc0 = col.x; lc1 = col.y; lc2 = col.z;
        float dL_dalpha = (col.x - acc0) * g0 + (col.y - acc1) * g1 + (col.z - acc2) * g2;
        v_r = w * g0; v_g = w * g1; v_bl = w * g2;
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-Tf / (1.0f - alpha)) * bg_dot_g;
        const float dL_dG = co.w * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddx = -gdx * co.x - gdy * co.y;
        const float dG_ddy = -gdy * co.z - gdx * co.y;
        v_dx = dL_dG * dG_ddx * half_w;
        v_dy = dL_dG * dG_ddy * half_h;
        v_a = -0.5f * gdx * dx * dL_dG;
        v_b = -gdx * dy * dL_dG;
        v_c = -0.5f * gdy * dy * dL_dG;
        v_o = G * dL_dalpha;
      }
      // wave-level reduction, then one 9-lane atomic instruction per (wave, Gaussian)
      float s[9];
      s[0] = wave_sum_to_lane63(v_dx);
      s[1] = wave_sum_to_lane63(v_dy);
      s[2] = wave_sum_to_lane63(v_a);
      s[3] = wave_sum_to_lane63(v_b);
      s[4] = wave_sum_to_lane63(v_c);
      s[5] = wave_sum_to_lane63(v_o);
      s[6] = wave_sum_to_lane63(v_r);
      s[7] = wave_sum_to_lane63(v_g);
      s[8] = wave_sum_to_lane63(v_bl);
      float mine = 0.f;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s[q]), 63));
        mine = (lane == q) ? tot : mine;
      }
      if (lane < 9) unsafeAtomicAdd(&grad_acc[(size_t)s_idx[j] * GSR_GRAD_STRIDE + lane], mine);
    }
  }
}

}  // namespace

hipError_t launch_render_fwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             float* out_color, hipStream_t stream) {
  if (d.T == 0) return hipSuccess;
  {
    ProfScope prof_(K_RENDER_FWD, stream);
    hipLaunchKernelGGL(render_fwd_kernel, dim3(d.T), dim3(BATCH), 0, stream, d.W, d.H, d.gx,
                     d.max_pairs, ws.tile_offset, ws.point_list, ws.xy, ws.conic_opacity, ws.rgb,
                     s.bg, out_color, ws.final_T, ws.n_contrib);
  }
  return hipGetLastError();
}

hipError_t launch_render_bwd(const GsrSettings& s, const Dims& d, const Workspace& ws,
                             const float* dL_dout, hipStream_t stream) {
  if (d.T == 0 || d.P == 0) return hipSuccess;
  {
    ProfScope prof_(K_RENDER_BWD, stream);
    hipLaunchKernelGGL(render_bwd_kernel, dim3(d.T), dim3(BATCH), 0, stream, d.W, d.H, d.gx,
                     d.max_pairs, ws.tile_offset, ws.point_list, ws.xy, ws.conic_opacity, ws.rgb,
                     s.bg, ws.final_T, ws.n_contrib, dL_dout, ws.grad_acc);
  }
  return hipGetLastError();
}

}  // namespace gsr
