// gsr_preprocess.hip — K1 (per-Gaussian forward) and K7 (per-Gaussian backward).
//
// This translation unit is compiled with -ffp-contract=off: the integer outputs (radii,
// tile rects, tiles_touched) are decided by float32 expressions whose operation order is
// the contract shared with the CPU oracle (oracle/gsr_oracle.c); a fused multiply-add would
// flip a ceil()/trunc() now and then. Spec: SURVEY.md Appendix A.1 and A.5.
#include "gsr_common.h"

namespace gsr {

namespace {

// Element (r,c) of the column-vector-form matrix stored as the reference's transposed
// row-major tensor (scene/dataset_mono.py:248-255).
__device__ __forceinline__ float vm(const float* m, int r, int c) { return m[c * 4 + r]; }

__device__ __forceinline__ int trunc_clamp(float v, int lo, int hi) {
  if (!(v > (float)lo)) return lo;  // also NaN
  if (v >= (float)hi) return hi;
  return (int)v;
}

__device__ __forceinline__ void quat_to_rot(const float* q, float Rm[3][3]) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];  // NOT normalised (A.1 step 3)
  Rm[0][0] = 1.0f - 2.0f * (y * y + z * z);
  Rm[0][1] = 2.0f * (x * y - r * z);
  Rm[0][2] = 2.0f * (x * z + r * y);
  Rm[1][0] = 2.0f * (x * y + r * z);
  Rm[1][1] = 1.0f - 2.0f * (x * x + z * z);
  Rm[1][2] = 2.0f * (y * z - r * x);
  Rm[2][0] = 2.0f * (x * z - r * y);
  Rm[2][1] = 2.0f * (y * z + r * x);
  Rm[2][2] = 1.0f - 2.0f * (x * x + y * y);
}

// View-dependent intermediates shared by forward and backward.
struct Ewa {
  float tx, ty, tz, u, v;
  float T0[3], T1[3];
  float s0[3], s1[3];   // Sigma * T0^T, Sigma * T1^T
  float a, b, c;        // dilated 2D covariance
  bool clx, cly;
};

__device__ __forceinline__ void ewa_project(const float* V, float fx, float fy, float limx,
                                            float limy, float tx, float ty, float tz,
                                            const float* c6, Ewa& e) {
  e.tx = tx; e.ty = ty; e.tz = tz;
  const float txtz = tx / tz, tytz = ty / tz;
  e.clx = (txtz < -limx) || (txtz > limx);
  e.cly = (tytz < -limy) || (tytz > limy);
  e.u = fminf(limx, fmaxf(-limx, txtz)) * tz;
  e.v = fminf(limy, fmaxf(-limy, tytz)) * tz;
  const float J00 = fx / tz, J02 = -(fx * e.u) / (tz * tz);
  const float J11 = fy / tz, J12 = -(fy * e.v) / (tz * tz);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    e.T0[j] = J00 * vm(V, 0, j) + J02 * vm(V, 2, j);
    e.T1[j] = J11 * vm(V, 1, j) + J12 * vm(V, 2, j);
  }
  const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    e.s0[p] = S[p][0] * e.T0[0] + S[p][1] * e.T0[1] + S[p][2] * e.T0[2];
    e.s1[p] = S[p][0] * e.T1[0] + S[p][1] * e.T1[1] + S[p][2] * e.T1[2];
  }
  e.a = (e.T0[0] * e.s0[0] + e.T0[1] * e.s0[1] + e.T0[2] * e.s0[2]) + 0.3f;
  e.b = e.T1[0] * e.s0[0] + e.T1[1] * e.s0[1] + e.T1[2] * e.s0[2];
  e.c = (e.T1[0] * e.s1[0] + e.T1[1] * e.s1[1] + e.T1[2] * e.s1[2]) + 0.3f;
}

__global__ void __launch_bounds__(256)
preprocess_kernel(int P, int W, int H, int gx, int gy, float tanfovx, float tanfovy,
                  float scale_modifier, const float* __restrict__ view,
                  const float* __restrict__ proj, const float* __restrict__ means3D,
                  const float* __restrict__ colors, const float* __restrict__ opacities,
                  const float* __restrict__ scales, const float* __restrict__ rotations,
                  const float* __restrict__ cov3D_precomp, Workspace ws,
                  int32_t* __restrict__ radii, Batch bt, int win_cells) {
  extern __shared__ int s_dyn[];    // the workgroup's tile window (gsr_common.h), or the fallback's hash table
  __shared__ int s_box[4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = i < P;      // (every thread stays for the workgroup's histogram)
  {   // batched launch: select this frame's inputs, workspace and outputs
    const int64_t f = blockIdx.y;
    view += f * bt.view; proj += f * bt.proj; means3D += f * bt.means;
    if (colors) colors += f * bt.colors;
    opacities += f * bt.opacities;
    if (scales) { scales += f * bt.scales; rotations += f * bt.rotations; }
    if (cov3D_precomp) cov3D_precomp += f * bt.cov3d;
    ws = frame_ws(ws, (size_t)f * bt.ws_stride);
    radii += f * P;
  }
  int4 rc = make_int4(0, 0, 0, 0);
  if (in_range) {
    float V[16], PV[16];
  #pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = view[k]; PV[k] = proj[k]; }

    // defaults for a Gaussian that is not rendered
    int rad_i = 0;
    uint32_t ntiles = 0;
    float depth = 0.f;
    float2 xy = make_float2(0.f, 0.f);
    float4 co = make_float4(0.f, 0.f, 0.f, 0.f);

    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    float c6[6];
    if (cov3D_precomp) {
  #pragma unroll
      for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
    } else {
      float Rm[3][3], M[3][3];
      const float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2],
                          rotations[4 * i + 3]};
      quat_to_rot(q, Rm);
  #pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float sk = scale_modifier * scales[3 * i + k];
  #pragma unroll
        for (int a = 0; a < 3; ++a) M[k][a] = sk * Rm[a][k];
      }
      int o = 0;
  #pragma unroll
      for (int a = 0; a < 3; ++a)
  #pragma unroll
        for (int b = a; b < 3; ++b)
          c6[o++] = M[0][a] * M[0][b] + M[1][a] * M[1][b] + M[2][a] * M[2][b];
    }
  #pragma unroll
    for (int k = 0; k < 6; ++k) ws.cov3d[6 * i + k] = c6[k];
    // with SH input the colour is filled in by sh_color_kernel (gsr_sh.hip) right after this kernel
    if (colors) ws.rgb[i] = make_float4(colors[3 * i], colors[3 * i + 1], colors[3 * i + 2], 0.f);

    const float tx = vm(V, 0, 0) * px + vm(V, 0, 1) * py + vm(V, 0, 2) * pz + vm(V, 0, 3);
    const float ty = vm(V, 1, 0) * px + vm(V, 1, 1) * py + vm(V, 1, 2) * pz + vm(V, 1, 3);
    const float tz = vm(V, 2, 0) * px + vm(V, 2, 1) * py + vm(V, 2, 2) * pz + vm(V, 2, 3);
    if (tz > 0.2f) {
      const float fx = (float)W / (2.0f * tanfovx);
      const float fy = (float)H / (2.0f * tanfovy);
      const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
      const float hx = vm(PV, 0, 0) * px + vm(PV, 0, 1) * py + vm(PV, 0, 2) * pz + vm(PV, 0, 3);
      const float hy = vm(PV, 1, 0) * px + vm(PV, 1, 1) * py + vm(PV, 1, 2) * pz + vm(PV, 1, 3);
      const float hw = vm(PV, 3, 0) * px + vm(PV, 3, 1) * py + vm(PV, 3, 2) * pz + vm(PV, 3, 3);
      const float winv = 1.0f / (hw + 0.0000001f);
      const float ndcx = hx * winv, ndcy = hy * winv;
      Ewa e;
      ewa_project(V, fx, fy, limx, limy, tx, ty, tz, c6, e);
      const float det = e.a * e.c - e.b * e.b;
      if (det != 0.0f) {
        const float det_inv = 1.0f / det;
        const float mid = 0.5f * (e.a + e.c);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float l1 = mid + sq, l2 = mid - sq;
        const float radf = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
        const float pxx = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
        const float pyy = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
        const int x0 = trunc_clamp((pxx - radf) / (float)GSR_TILE, 0, gx);
        const int y0 = trunc_clamp((pyy - radf) / (float)GSR_TILE, 0, gy);
        const int x1 = trunc_clamp((pxx + radf + (float)(GSR_TILE - 1)) / (float)GSR_TILE, 0, gx);
        const int y1 = trunc_clamp((pyy + radf + (float)(GSR_TILE - 1)) / (float)GSR_TILE, 0, gy);
        const int nt = (x1 - x0) * (y1 - y0);
        if (nt > 0) {
          depth = tz;
          rad_i = (radf < 2147483520.0f) ? (int)radf : 2147483647;
          xy = make_float2(pxx, pyy);
          co = make_float4(e.c * det_inv, -e.b * det_inv, e.a * det_inv, opacities[i]);
          rc = make_int4(x0, y0, x1, y1);
          ntiles = (uint32_t)nt;
        }
      }
    }
    radii[i] = rad_i;
    ws.depth[i] = depth;
    ws.xy[i] = xy;
    ws.conic_opacity[i] = co;
    {
      const float2 ext = alpha_extent(co);
      ws.xyext[i] = make_float4(xy.x, xy.y, ext.x, ext.y);
    }
    ws.rect[i] = rc;
    ws.tiles_touched[i] = ntiles;
  }
  // per-tile histogram of pairs (consumed by K2/K3): the workgroup's rectangles as a 2-D difference array over their
  // bounding window in LDS, one prefix sum, then one global atomic per (workgroup, tile) (gsr_common.h: TileWin)
  const bool has_rect = (rc.z - rc.x) * (rc.w - rc.y) > 0;
  const TileWin wn = wg_tile_window(rc, has_rect, s_box, win_cells);
  if (wn.w == 0) return;                                    // nothing of this workgroup is rendered
  if (wn.w > 0) {
    const int ncell = wn.w * wn.h;
    for (int e = threadIdx.x; e < ncell; e += blockDim.x) s_dyn[e] = 0;
    __syncthreads();
    if (has_rect) win_mark(s_dyn, wn, rc);
    __syncthreads();
    win_prefix(s_dyn, wn);
    for (int e = threadIdx.x; e < ncell; e += blockDim.x) {
      const int c = s_dyn[e];
      if (c > 0) atomicAdd(&ws.tile_count[(wn.y0 + e / wn.w) * gx + wn.x0 + e % wn.w], (uint32_t)c);
    }
    return;
  }
  // the window does not fit (more than GSR_WIN_CELLS tiles in the frame): hash table keyed by tile id, a lane walks
  // its own rectangle (rectangles above GSR_BIG_RECT tiles: the whole wave, direct atomics)
  TileAgg& s_agg = *reinterpret_cast<TileAgg*>(s_dyn);
  agg_clear(s_agg);
  __syncthreads();
  if (!rect_is_big(rc))
    for (int cy = rc.y; cy < rc.w; ++cy)
      for (int cx = rc.x; cx < rc.z; ++cx) {
        const int tile = cy * gx + cx;
        const int slot = agg_claim(s_agg, tile);
        if (slot >= 0) atomicAdd(&s_agg.cnt[slot], 1u);
        else atomicAdd(&ws.tile_count[tile], 1u);        // table full along the probe sequence: direct
      }
  for_big_rects(rc, gx, [](int) { return 0; }, [&](int tile, int) { atomicAdd(&ws.tile_count[tile], 1u); });
  __syncthreads();
  for (int sl = threadIdx.x; sl < GSR_AGG_SLOTS; sl += blockDim.x)
    if (s_agg.key[sl] >= 0) atomicAdd(&ws.tile_count[s_agg.key[sl]], s_agg.cnt[sl]);
}

// K7 — Appendix A.5. One thread per Gaussian; reads the screen-space gradient accumulators
// written by K6 (grad_acc: dxy2 (already scaled by 0.5W, 0.5H), dconic3, dopac1, drgb3).
#ifndef GSR_PBWD_WAVES
#define GSR_PBWD_WAVES 0
#endif
#if GSR_PBWD_WAVES > 0
#define GSR_PBWD_OCC __attribute__((amdgpu_waves_per_eu(GSR_PBWD_WAVES, GSR_PBWD_WAVES)))
#else
#define GSR_PBWD_OCC
#endif
__global__ void __launch_bounds__(256) GSR_PBWD_OCC
preprocess_bwd_kernel(int P, int W, int H, float tanfovx, float tanfovy, float scale_modifier,
                      const float* __restrict__ view, const float* __restrict__ proj,
                      const float* __restrict__ means3D, const float* __restrict__ scales,
                      const float* __restrict__ rotations, const int32_t* __restrict__ radii,
                      Workspace ws, float* __restrict__ dL_dmeans3D,
                      float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dcolors,
                      float* __restrict__ dL_dopacity, float* __restrict__ dL_dscales,
                      float* __restrict__ dL_drotations, float* __restrict__ dL_dcov3D,
                      int32_t* __restrict__ overflow_flag, Batch bt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  {
    const int64_t f = blockIdx.y;
    view += f * bt.view; proj += f * bt.proj; means3D += f * bt.means;
    if (scales) { scales += f * bt.scales; rotations += f * bt.rotations; }
    ws = frame_ws(ws, (size_t)f * bt.ws_stride);
    radii += f * P;
    if (dL_dmeans3D) dL_dmeans3D += f * P * 3;
    if (dL_dmeans2D) dL_dmeans2D += f * P * 3;
    if (dL_dcolors) dL_dcolors += f * P * 3;
    if (dL_dopacity) dL_dopacity += f * P;
    if (dL_dscales) dL_dscales += f * P * 3;
    if (dL_drotations) dL_drotations += f * P * 4;
    if (dL_dcov3D) dL_dcov3D += f * P * 6;
  }
  // A frame whose pair buffer overflowed was rendered from truncated tile lists: it yields NO gradient (all zeros)
  // and raises the caller's flag, which the optimiser reads to skip the step (include/gsr.h: gsr_backward)
  const bool overflowed = ws.status[1] != 0;
  if (overflowed && overflow_flag && i == 0) *overflow_flag = 1;
  const bool live = radii[i] > 0 && !overflowed;
  float g[GSR_GRAD_STRIDE];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = live ? ws.grad_acc[(size_t)i * GSR_GRAD_STRIDE + k] : 0.f;
  if (dL_dmeans2D) {
    dL_dmeans2D[3 * i] = g[0];
    dL_dmeans2D[3 * i + 1] = g[1];
    dL_dmeans2D[3 * i + 2] = 0.f;
  }
  if (dL_dopacity) dL_dopacity[i] = g[5];
  if (dL_dcolors) {
    dL_dcolors[3 * i] = g[6];
    dL_dcolors[3 * i + 1] = g[7];
    dL_dcolors[3 * i + 2] = g[8];
  }
  float dmean[3] = {0.f, 0.f, 0.f};
  float dS[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float dscale[3] = {0.f, 0.f, 0.f};
  float drot[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    float V[16], PV[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = view[k]; PV[k] = proj[k]; }
    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    const float tx = vm(V, 0, 0) * px + vm(V, 0, 1) * py + vm(V, 0, 2) * pz + vm(V, 0, 3);
    const float ty = vm(V, 1, 0) * px + vm(V, 1, 1) * py + vm(V, 1, 2) * pz + vm(V, 1, 3);
    const float tz = vm(V, 2, 0) * px + vm(V, 2, 1) * py + vm(V, 2, 2) * pz + vm(V, 2, 3);
    const float fx = (float)W / (2.0f * tanfovx);
    const float fy = (float)H / (2.0f * tanfovy);
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    float c6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = ws.cov3d[6 * i + k];
    Ewa e;
    ewa_project(V, fx, fy, limx, limy, tx, ty, tz, c6, e);
    // (a) conic -> Sigma2D
    const float a = e.a, b = e.b, c = e.c;
    const float den = a * c - b * b;
    const float k2 = 1.0f / (den * den + 0.0000001f);
    const float gA = g[2], gB = g[3], gC = g[4];
    float da = 0.f, db = 0.f, dc = 0.f;
    if (den != 0.0f) {
      da = k2 * (-c * c * gA + b * c * gB + (den - a * c) * gC);
      dc = k2 * (-a * a * gC + a * b * gB + (den - a * c) * gA);
      db = k2 * (2.0f * b * c * gA - (den + 2.0f * b * b) * gB + 2.0f * a * b * gC);
    }
    // (b) Sigma2D -> Sigma3D (off-diagonals hold the sum over both symmetric entries)
    const float* T0 = e.T0;
    const float* T1 = e.T1;
    dS[0] = T0[0] * T0[0] * da + T0[0] * T1[0] * db + T1[0] * T1[0] * dc;
    dS[3] = T0[1] * T0[1] * da + T0[1] * T1[1] * db + T1[1] * T1[1] * dc;
    dS[5] = T0[2] * T0[2] * da + T0[2] * T1[2] * db + T1[2] * T1[2] * dc;
    dS[1] = 2.0f * T0[0] * T0[1] * da + (T0[0] * T1[1] + T0[1] * T1[0]) * db +
            2.0f * T1[0] * T1[1] * dc;
    dS[2] = 2.0f * T0[0] * T0[2] * da + (T0[0] * T1[2] + T0[2] * T1[0]) * db +
            2.0f * T1[0] * T1[2] * dc;
    dS[4] = 2.0f * T0[1] * T0[2] * da + (T0[1] * T1[2] + T0[2] * T1[1]) * db +
            2.0f * T1[1] * T1[2] * dc;
    // (c) Sigma2D -> T -> J -> t -> mean
    float dT0[3], dT1[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      dT0[p] = 2.0f * e.s0[p] * da + e.s1[p] * db;
      dT1[p] = 2.0f * e.s1[p] * dc + e.s0[p] * db;
    }
    const float dJ00 = dT0[0] * vm(V, 0, 0) + dT0[1] * vm(V, 0, 1) + dT0[2] * vm(V, 0, 2);
    const float dJ02 = dT0[0] * vm(V, 2, 0) + dT0[1] * vm(V, 2, 1) + dT0[2] * vm(V, 2, 2);
    const float dJ11 = dT1[0] * vm(V, 1, 0) + dT1[1] * vm(V, 1, 1) + dT1[2] * vm(V, 1, 2);
    const float dJ12 = dT1[0] * vm(V, 2, 0) + dT1[1] * vm(V, 2, 1) + dT1[2] * vm(V, 2, 2);
    const float tz_inv = 1.0f / tz, tz2 = tz_inv * tz_inv, tz3 = tz2 * tz_inv;
    const float dtx = e.clx ? 0.0f : -fx * tz2 * dJ02;
    const float dty = e.cly ? 0.0f : -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + 2.0f * fx * e.u * tz3 * dJ02 +
                      2.0f * fy * e.v * tz3 * dJ12;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      dmean[k] = vm(V, 0, k) * dtx + vm(V, 1, k) * dty + vm(V, 2, k) * dtz;
    // (d) screen position -> mean
    const float hx = vm(PV, 0, 0) * px + vm(PV, 0, 1) * py + vm(PV, 0, 2) * pz + vm(PV, 0, 3);
    const float hy = vm(PV, 1, 0) * px + vm(PV, 1, 1) * py + vm(PV, 1, 2) * pz + vm(PV, 1, 3);
    const float hw = vm(PV, 3, 0) * px + vm(PV, 3, 1) * py + vm(PV, 3, 2) * pz + vm(PV, 3, 3);
    const float winv = 1.0f / (hw + 0.0000001f);
    const float m1 = hx * winv * winv, m2 = hy * winv * winv;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      dmean[k] += (vm(PV, 0, k) * winv - vm(PV, 3, k) * m1) * g[0] +
                  (vm(PV, 1, k) * winv - vm(PV, 3, k) * m2) * g[1];
    // (e) Sigma3D -> scale, rotation
    if (scales) {
      float Rm[3][3], M[3][3], sk[3];
      const float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2],
                          rotations[4 * i + 3]};
      quat_to_rot(q, Rm);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        sk[k] = scale_modifier * scales[3 * i + k];
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2) M[k][a2] = sk[k] * Rm[a2][k];
      }
      const float Dm[3][3] = {{dS[0], 0.5f * dS[1], 0.5f * dS[2]},
                              {0.5f * dS[1], dS[3], 0.5f * dS[4]},
                              {0.5f * dS[2], 0.5f * dS[4], dS[5]}};
      float dM[3][3], dR[3][3];
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2)
          dM[k][a2] = 2.0f * (M[k][0] * Dm[0][a2] + M[k][1] * Dm[1][a2] + M[k][2] * Dm[2][a2]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        // no scale_modifier factor here: upstream quirk kept for parity
        dscale[k] = Rm[0][k] * dM[k][0] + Rm[1][k] * dM[k][1] + Rm[2][k] * dM[k][2];
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2) dR[a2][k] = dM[k][a2] * sk[k];
      }
      const float r = q[0], x = q[1], y = q[2], z = q[3];
      drot[0] = 2.0f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] -
                        y * dR[2][0] + x * dR[2][1]);
      drot[1] = 2.0f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.0f * x * dR[1][1] -
                        r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.0f * x * dR[2][2]);
      drot[2] = 2.0f * (-2.0f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] +
                        z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.0f * y * dR[2][2]);
      drot[3] = 2.0f * (-2.0f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] -
                        2.0f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
    }
  }
  if (dL_dmeans3D) {
#pragma unroll
    for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = dmean[k];
  }
  if (dL_dcov3D) {
#pragma unroll
    for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = dS[k];
  }
  if (dL_dscales) {
#pragma unroll
    for (int k = 0; k < 3; ++k) dL_dscales[3 * i + k] = dscale[k];
  }
  if (dL_drotations) {
#pragma unroll
    for (int k = 0; k < 4; ++k) dL_drotations[4 * i + k] = drot[k];
  }
}

__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                    uint8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
  const float tz = vm(view, 2, 0) * px + vm(view, 2, 1) * py + vm(view, 2, 2) * pz + vm(view, 2, 3);
  out[i] = tz > 0.2f ? 1 : 0;
}

}  // namespace

hipError_t launch_preprocess(const GsrSettings& s, const Dims& d, const float* means3D,
                             const float* colors_precomp, const float* opacities,
                             const float* scales, const float* rotations,
                             const float* cov3D_precomp, const Workspace& ws, int32_t* radii,
                             const Batch& bt, hipStream_t stream) {
  if (d.P == 0) return hipSuccess;
  const int block = 256;
  const int grid = (d.P + block - 1) / block;
  {
    ProfScope prof_(K_PREPROCESS, stream);
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid, bt.frames), dim3(block), win_lds_bytes(d.T), stream, d.P, d.W, d.H,
                     d.gx, d.gy, s.tanfovx, s.tanfovy, s.scale_modifier, s.viewmatrix, s.projmatrix,
                     means3D, colors_precomp, opacities, scales, rotations, cov3D_precomp, ws,
                     radii, bt, d.T < GSR_WIN_CELLS ? d.T : GSR_WIN_CELLS);
  }
  return hipGetLastError();
}

hipError_t launch_preprocess_bwd(const GsrSettings& s, const Dims& d, const float* means3D,
                                 const float* scales, const float* rotations,
                                 const int32_t* radii, const Workspace& ws,
                                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors,
                                 float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                                 float* dL_dcov3D, int32_t* overflow_flag, const Batch& bt, hipStream_t stream) {
  if (d.P == 0) return hipSuccess;
  const int block = 256;
  const int grid = (d.P + block - 1) / block;
  {
    ProfScope prof_(K_PREPROCESS_BWD, stream);
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(grid, bt.frames), dim3(block), 0, stream, d.P, d.W, d.H,
                     s.tanfovx, s.tanfovy, s.scale_modifier, s.viewmatrix, s.projmatrix,
                     means3D, scales, rotations, radii, ws, dL_dmeans3D, dL_dmeans2D,
                     dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, overflow_flag, bt);
  }
  return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix,
                               uint8_t* out, hipStream_t stream) {
  if (P == 0) return hipSuccess;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P,
                     means3D, viewmatrix, out);
  return hipGetLastError();
}

}  // namespace gsr
