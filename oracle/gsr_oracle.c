/*
 * gsr_oracle.c — CPU ORACLE (test infrastructure, NOT product code) for the differentiable
 * Gaussian-splatting rasterizer on the GaussianAvatar render-and-fit hot path.
 *
 * PARITY UNPINNED: the arithmetic this restates lives in the third-party package
 * `diff_gaussian_rasterization` (graphdeco-inria/diff-gaussian-rasterization, original
 * 3DGS-era revision: 12-field settings tuple, (color, radii) return), which the reference
 * imports at /root/reference/gaussian_renderer/__init__.py:6 but does not vendor and does
 * not pin (/root/reference/README.md:37). The reference holds no tests, golden images or
 * known-answer vectors for it. This file therefore restates the published algorithm from
 * the behavioural specification in SURVEY.md Appendix A (A.1 per-Gaussian forward, A.2
 * binning order, A.3 per-pixel forward, A.4 per-pixel backward, A.5 per-Gaussian backward)
 * and anchors on the reference's own call site (gaussian_renderer/__init__.py:21-48) and
 * camera conventions (scene/dataset_mono.py:248-255, utils/graphics_utils.py:41-72).
 * It is cross-checked by an independently written vectorised torch restatement
 * (oracle/raster_torch.py) whose gradients come from autograd, and by finite differences.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Arithmetic: `real` is float (default) — every operation is written out in a fixed order
 * and the file is compiled with -ffp-contract=off so that the integer outputs (radii, tile
 * rects, per-tile order) are a reproducible function of the inputs; the HIP kernels follow
 * the same order for the integer-determining stage. With -DGSRO_F64 `real` is double and
 * the library serves as a high-precision gradient reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef GSRO_F64
typedef double real;
#define R(x) x
#define SQRT sqrt
#define EXP exp
#define CEIL ceil
#else
typedef float real;
#define R(x) x##f
#define SQRT sqrtf
#define EXP expf
#define CEIL ceilf
#endif

#define TILE 16

typedef struct {
  int W, H;
  real tanfovx, tanfovy;
  real scale_modifier;
  const real* bg;    /* [3]  */
  const real* view;  /* [16] element (r,c) of the column-vector-form matrix = view[c*4+r] */
  const real* proj;  /* [16] same convention */
} GsroCam;

int gsro_real_bytes(void) { return (int)sizeof(real); }

static inline real vm(const real* m, int r, int c) { return m[c * 4 + r]; }

/* clamp(trunc(v), lo, hi) without ever converting an out-of-range float (A.1 step 8). */
static inline int trunc_clamp(real v, int lo, int hi) {
  if (!(v > (real)lo)) return lo; /* also NaN */
  if (v >= (real)hi) return hi;
  return (int)v;
}

static inline uint32_t depth_bits(real d) {
  float f = (float)d;
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

/* World covariance from scale and UN-normalised quaternion (A.1 step 3). */
static void cov3d_from_scale_rot(const real* s3, const real* q4, real mod, real* cov6) {
  real r = q4[0], x = q4[1], y = q4[2], z = q4[3];
  real Rm[3][3];
  Rm[0][0] = R(1.0) - R(2.0) * (y * y + z * z);
  Rm[0][1] = R(2.0) * (x * y - r * z);
  Rm[0][2] = R(2.0) * (x * z + r * y);
  Rm[1][0] = R(2.0) * (x * y + r * z);
  Rm[1][1] = R(1.0) - R(2.0) * (x * x + z * z);
  Rm[1][2] = R(2.0) * (y * z - r * x);
  Rm[2][0] = R(2.0) * (x * z - r * y);
  Rm[2][1] = R(2.0) * (y * z + r * x);
  Rm[2][2] = R(1.0) - R(2.0) * (x * x + y * y);
  real M[3][3]; /* M[k][a] = s_k * R[a][k]  (M = S R^T, Sigma = M^T M) */
  for (int k = 0; k < 3; ++k) {
    real sk = mod * s3[k];
    for (int a = 0; a < 3; ++a) M[k][a] = sk * Rm[a][k];
  }
  int o = 0;
  for (int a = 0; a < 3; ++a)
    for (int b = a; b < 3; ++b)
      cov6[o++] = M[0][a] * M[0][b] + M[1][a] * M[1][b] + M[2][a] * M[2][b];
}

/*
 * A.1 — per-Gaussian forward. Outputs (all caller-allocated):
 *   depth[P], xy[P,2], conic_opacity[P,4], cov3d[P,6], radii[P], rect[P,4], tiles_touched[P]
 */
void gsro_preprocess(int P, const real* means3D, const real* scales, const real* rots,
                     const real* cov3D_precomp, const real* opacities, const GsroCam* cam,
                     real* depth, real* xy, real* conic_opacity, real* cov3d, int32_t* radii,
                     int32_t* rect, uint32_t* tiles_touched) {
  const int W = cam->W, H = cam->H;
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const real fx = (real)W / (R(2.0) * cam->tanfovx);
  const real fy = (real)H / (R(2.0) * cam->tanfovy);
  const real limx = R(1.3) * cam->tanfovx, limy = R(1.3) * cam->tanfovy;
  const real* V = cam->view;
  const real* PV = cam->proj;
  for (int i = 0; i < P; ++i) {
    radii[i] = 0;
    tiles_touched[i] = 0;
    rect[4 * i + 0] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
    depth[i] = 0;
    xy[2 * i] = xy[2 * i + 1] = 0;
    for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = 0;
    const real px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    if (cov3D_precomp) {
      for (int k = 0; k < 6; ++k) cov3d[6 * i + k] = cov3D_precomp[6 * i + k];
    } else {
      cov3d_from_scale_rot(scales + 3 * i, rots + 4 * i, cam->scale_modifier, cov3d + 6 * i);
    }
    /* step 1: view space, near cull */
    const real tx = vm(V, 0, 0) * px + vm(V, 0, 1) * py + vm(V, 0, 2) * pz + vm(V, 0, 3);
    const real ty = vm(V, 1, 0) * px + vm(V, 1, 1) * py + vm(V, 1, 2) * pz + vm(V, 1, 3);
    const real tz = vm(V, 2, 0) * px + vm(V, 2, 1) * py + vm(V, 2, 2) * pz + vm(V, 2, 3);
    if (!(tz > R(0.2))) continue;
    /* step 2: clip space */
    const real hx = vm(PV, 0, 0) * px + vm(PV, 0, 1) * py + vm(PV, 0, 2) * pz + vm(PV, 0, 3);
    const real hy = vm(PV, 1, 0) * px + vm(PV, 1, 1) * py + vm(PV, 1, 2) * pz + vm(PV, 1, 3);
    const real hw = vm(PV, 3, 0) * px + vm(PV, 3, 1) * py + vm(PV, 3, 2) * pz + vm(PV, 3, 3);
    const real winv = R(1.0) / (hw + R(0.0000001));
    const real ndcx = hx * winv, ndcy = hy * winv;
    /* step 4: EWA projection of the covariance */
    const real txtz = tx / tz, tytz = ty / tz;
    const real u = fmin(limx, fmax(-limx, txtz)) * tz;
    const real v = fmin(limy, fmax(-limy, tytz)) * tz;
    const real J00 = fx / tz, J02 = -(fx * u) / (tz * tz);
    const real J11 = fy / tz, J12 = -(fy * v) / (tz * tz);
    real T0[3], T1[3];
    for (int j = 0; j < 3; ++j) {
      T0[j] = J00 * vm(V, 0, j) + J02 * vm(V, 2, j);
      T1[j] = J11 * vm(V, 1, j) + J12 * vm(V, 2, j);
    }
    const real* c6 = cov3d + 6 * i;
    const real S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    real s0[3], s1[3];
    for (int p = 0; p < 3; ++p) {
      s0[p] = S[p][0] * T0[0] + S[p][1] * T0[1] + S[p][2] * T0[2];
      s1[p] = S[p][0] * T1[0] + S[p][1] * T1[1] + S[p][2] * T1[2];
    }
    const real a = (T0[0] * s0[0] + T0[1] * s0[1] + T0[2] * s0[2]) + R(0.3);
    const real b = T1[0] * s0[0] + T1[1] * s0[1] + T1[2] * s0[2];
    const real c = (T1[0] * s1[0] + T1[1] * s1[1] + T1[2] * s1[2]) + R(0.3);
    /* step 5 */
    const real det = a * c - b * b;
    if (det == R(0.0)) continue;
    const real det_inv = R(1.0) / det;
    const real cA = c * det_inv, cB = -b * det_inv, cC = a * det_inv;
    /* step 6 */
    const real mid = R(0.5) * (a + c);
    const real sq = SQRT(fmax(R(0.1), mid * mid - det));
    const real l1 = mid + sq, l2 = mid - sq;
    const real radf = CEIL(R(3.0) * SQRT(fmax(l1, l2)));
    /* step 7 */
    const real pxx = ((ndcx + R(1.0)) * (real)W - R(1.0)) * R(0.5);
    const real pyy = ((ndcy + R(1.0)) * (real)H - R(1.0)) * R(0.5);
    /* step 8 */
    const int x0 = trunc_clamp((pxx - radf) / (real)TILE, 0, gx);
    const int y0 = trunc_clamp((pyy - radf) / (real)TILE, 0, gy);
    const int x1 = trunc_clamp((pxx + radf + (real)(TILE - 1)) / (real)TILE, 0, gx);
    const int y1 = trunc_clamp((pyy + radf + (real)(TILE - 1)) / (real)TILE, 0, gy);
    const int nt = (x1 - x0) * (y1 - y0);
    if (nt <= 0) continue;
    /* step 9 */
    depth[i] = tz;
    radii[i] = (radf < R(2147483520.0)) ? (int32_t)radf : INT32_MAX;
    xy[2 * i] = pxx;
    xy[2 * i + 1] = pyy;
    conic_opacity[4 * i + 0] = cA;
    conic_opacity[4 * i + 1] = cB;
    conic_opacity[4 * i + 2] = cC;
    conic_opacity[4 * i + 3] = opacities[i];
    rect[4 * i + 0] = x0;
    rect[4 * i + 1] = y0;
    rect[4 * i + 2] = x1;
    rect[4 * i + 3] = y1;
    tiles_touched[i] = (uint32_t)nt;
  }
}

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return (x > y) - (x < y);
}

/*
 * A.2 — binning. Per tile, the list of Gaussian indices ordered by
 * (float_bits(depth), index). ranges[T,2] = [start,end) into point_list.
 * Returns D (number of pairs); if D > capacity nothing is written and -D is returned.
 */
int64_t gsro_bin(int P, const int32_t* rect, const uint32_t* tiles_touched, const real* depth,
                 int W, int H, uint32_t* ranges, uint32_t* point_list, int64_t capacity) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int T = gx * gy;
  int64_t D = 0;
  for (int i = 0; i < P; ++i) D += tiles_touched[i];
  if (D > capacity) return -D;
  uint32_t* count = (uint32_t*)calloc((size_t)T + 1, sizeof(uint32_t));
  for (int i = 0; i < P; ++i) {
    if (!tiles_touched[i]) continue;
    for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
      for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; ++x) count[y * gx + x]++;
  }
  uint32_t run = 0;
  for (int t = 0; t < T; ++t) {
    ranges[2 * t] = run;
    run += count[t];
    ranges[2 * t + 1] = run;
    count[t] = ranges[2 * t]; /* becomes the write cursor */
  }
  uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(D > 0 ? D : 1));
  for (int i = 0; i < P; ++i) {
    if (!tiles_touched[i]) continue;
    const uint64_t key = ((uint64_t)depth_bits(depth[i]) << 32) | (uint32_t)i;
    for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
      for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; ++x) keys[count[y * gx + x]++] = key;
  }
  for (int t = 0; t < T; ++t) {
    const uint32_t s = ranges[2 * t], e = ranges[2 * t + 1];
    if (e - s > 1) qsort(keys + s, e - s, sizeof(uint64_t), cmp_u64);
  }
  for (int64_t k = 0; k < D; ++k) point_list[k] = (uint32_t)(keys[k] & 0xffffffffu);
  free(keys);
  free(count);
  return D;
}

/* A.3 — per-pixel forward. out_color [3,H,W]; final_T [H*W]; n_contrib [H*W]. */
void gsro_render(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                 const real* xy, const real* conic_opacity, const real* rgb, const real* bg,
                 real* out_color, real* final_T, uint32_t* n_contrib) {
  const int gx = (W + TILE - 1) / TILE;
  for (int py = 0; py < H; ++py) {
    for (int px = 0; px < W; ++px) {
      const int tile = (py / TILE) * gx + (px / TILE);
      const uint32_t s = ranges[2 * tile], e = ranges[2 * tile + 1];
      real T = R(1.0), C[3] = {0, 0, 0};
      uint32_t k = 0, last = 0;
      for (uint32_t n = s; n < e; ++n) {
        k++;
        const uint32_t j = point_list[n];
        const real dx = xy[2 * j] - (real)px, dy = xy[2 * j + 1] - (real)py;
        const real A = conic_opacity[4 * j], B = conic_opacity[4 * j + 1];
        const real Cc = conic_opacity[4 * j + 2], o = conic_opacity[4 * j + 3];
        const real power = R(-0.5) * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
        if (power > R(0.0)) continue;
        const real alpha = fmin(R(0.99), o * EXP(power));
        if (alpha < R(1.0) / R(255.0)) continue;
        const real Tn = T * (R(1.0) - alpha);
        if (Tn < R(0.0001)) break;
        for (int ch = 0; ch < 3; ++ch) C[ch] += rgb[3 * j + ch] * alpha * T;
        T = Tn;
        last = k;
      }
      const size_t pix = (size_t)py * W + px;
      final_T[pix] = T;
      n_contrib[pix] = last;
      for (int ch = 0; ch < 3; ++ch) out_color[(size_t)ch * H * W + pix] = C[ch] + T * bg[ch];
    }
  }
}

/*
 * A.4 — per-pixel backward. Accumulates (all zero-initialised here):
 *   dL_dmean2D [P,2]  gradient w.r.t. the NDC-scaled screen position (includes the 0.5W / 0.5H
 *                     factors) — this is what the reference API returns as the means2D grad
 *   dL_dconic  [P,3]  w.r.t. (A, B, C) with B the single off-diagonal variable of `power`
 *   dL_dopacity[P], dL_dcolor [P,3]
 */
void gsro_render_backward(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                          const real* xy, const real* conic_opacity, const real* rgb,
                          const real* bg, const real* final_T, const uint32_t* n_contrib,
                          const real* dL_dout, real* dL_dmean2D, real* dL_dconic,
                          real* dL_dopacity, real* dL_dcolor) {
  const int gx = (W + TILE - 1) / TILE;
  memset(dL_dmean2D, 0, sizeof(real) * 2 * (size_t)P);
  memset(dL_dconic, 0, sizeof(real) * 3 * (size_t)P);
  memset(dL_dopacity, 0, sizeof(real) * (size_t)P);
  memset(dL_dcolor, 0, sizeof(real) * 3 * (size_t)P);
  const real half_w = R(0.5) * (real)W, half_h = R(0.5) * (real)H;
  for (int py = 0; py < H; ++py) {
    for (int px = 0; px < W; ++px) {
      const size_t pix = (size_t)py * W + px;
      const int tile = (py / TILE) * gx + (px / TILE);
      const uint32_t s = ranges[2 * tile];
      const uint32_t last = n_contrib[pix];
      const real Tf = final_T[pix];
      real g[3];
      for (int ch = 0; ch < 3; ++ch) g[ch] = dL_dout[(size_t)ch * H * W + pix];
      const real bg_dot_g = bg[0] * g[0] + bg[1] * g[1] + bg[2] * g[2];
      real T = Tf, acc[3] = {0, 0, 0}, last_alpha = 0, last_color[3] = {0, 0, 0};
      for (uint32_t k = last; k-- > 0;) {
        const uint32_t j = point_list[s + k];
        const real dx = xy[2 * j] - (real)px, dy = xy[2 * j + 1] - (real)py;
        const real A = conic_opacity[4 * j], B = conic_opacity[4 * j + 1];
        const real Cc = conic_opacity[4 * j + 2], o = conic_opacity[4 * j + 3];
        const real power = R(-0.5) * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
        if (power > R(0.0)) continue;
        const real G = EXP(power);
        const real alpha = fmin(R(0.99), o * G);
        if (alpha < R(1.0) / R(255.0)) continue;
        T = T / (R(1.0) - alpha);
        const real w = alpha * T;
        real dL_dalpha = 0;
        for (int ch = 0; ch < 3; ++ch) {
          const real c = rgb[3 * j + ch];
          acc[ch] = last_alpha * last_color[ch] + (R(1.0) - last_alpha) * acc[ch];
          last_color[ch] = c;
          dL_dalpha += (c - acc[ch]) * g[ch];
          dL_dcolor[3 * j + ch] += w * g[ch];
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-Tf / (R(1.0) - alpha)) * bg_dot_g;
        const real dL_dG = o * dL_dalpha;
        const real gdx = G * dx, gdy = G * dy;
        const real dG_ddx = -gdx * A - gdy * B;
        const real dG_ddy = -gdy * Cc - gdx * B;
        dL_dmean2D[2 * j] += dL_dG * dG_ddx * half_w;
        dL_dmean2D[2 * j + 1] += dL_dG * dG_ddy * half_h;
        dL_dconic[3 * j] += R(-0.5) * gdx * dx * dL_dG;
        dL_dconic[3 * j + 1] += -gdx * dy * dL_dG;
        dL_dconic[3 * j + 2] += R(-0.5) * gdy * dy * dL_dG;
        dL_dopacity[j] += G * dL_dalpha;
      }
    }
  }
}

/*
 * A.5 — per-Gaussian backward. Inputs: the screen-space gradients above. Outputs (fully
 * overwritten; zeros where radii == 0): dL_dmeans3D [P,3], dL_dcov3D [P,6] (off-diagonal
 * entries hold the SUM over both symmetric positions), dL_dscales [P,3], dL_drots [P,4].
 * scales/rots may be NULL (cov3D_precomp path): dL_dscales/dL_drots are then untouched.
 */
void gsro_preprocess_backward(int P, const real* means3D, const real* scales, const real* rots,
                              const real* cov3d, const int32_t* radii, const GsroCam* cam,
                              const real* dL_dmean2D, const real* dL_dconic,
                              real* dL_dmeans3D, real* dL_dcov3D, real* dL_dscales,
                              real* dL_drots) {
  const int W = cam->W, H = cam->H;
  const real fx = (real)W / (R(2.0) * cam->tanfovx);
  const real fy = (real)H / (R(2.0) * cam->tanfovy);
  const real limx = R(1.3) * cam->tanfovx, limy = R(1.3) * cam->tanfovy;
  const real* V = cam->view;
  const real* PV = cam->proj;
  for (int i = 0; i < P; ++i) {
    for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = 0;
    for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0;
    if (scales) {
      for (int k = 0; k < 3; ++k) dL_dscales[3 * i + k] = 0;
      for (int k = 0; k < 4; ++k) dL_drots[4 * i + k] = 0;
    }
    if (radii[i] <= 0) continue;
    const real px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    const real tx = vm(V, 0, 0) * px + vm(V, 0, 1) * py + vm(V, 0, 2) * pz + vm(V, 0, 3);
    const real ty = vm(V, 1, 0) * px + vm(V, 1, 1) * py + vm(V, 1, 2) * pz + vm(V, 1, 3);
    const real tz = vm(V, 2, 0) * px + vm(V, 2, 1) * py + vm(V, 2, 2) * pz + vm(V, 2, 3);
    const real txtz = tx / tz, tytz = ty / tz;
    const int clx = (txtz < -limx) || (txtz > limx);
    const int cly = (tytz < -limy) || (tytz > limy);
    const real u = fmin(limx, fmax(-limx, txtz)) * tz;
    const real v = fmin(limy, fmax(-limy, tytz)) * tz;
    const real J00 = fx / tz, J02 = -(fx * u) / (tz * tz);
    const real J11 = fy / tz, J12 = -(fy * v) / (tz * tz);
    real T0[3], T1[3];
    for (int j = 0; j < 3; ++j) {
      T0[j] = J00 * vm(V, 0, j) + J02 * vm(V, 2, j);
      T1[j] = J11 * vm(V, 1, j) + J12 * vm(V, 2, j);
    }
    const real* c6 = cov3d + 6 * i;
    const real S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    real s0[3], s1[3];
    for (int p = 0; p < 3; ++p) {
      s0[p] = S[p][0] * T0[0] + S[p][1] * T0[1] + S[p][2] * T0[2];
      s1[p] = S[p][0] * T1[0] + S[p][1] * T1[1] + S[p][2] * T1[2];
    }
    const real a = (T0[0] * s0[0] + T0[1] * s0[1] + T0[2] * s0[2]) + R(0.3);
    const real b = T1[0] * s0[0] + T1[1] * s0[1] + T1[2] * s0[2];
    const real c = (T1[0] * s1[0] + T1[1] * s1[1] + T1[2] * s1[2]) + R(0.3);
    /* (a) conic -> Sigma2D */
    const real den = a * c - b * b;
    const real k2 = R(1.0) / (den * den + R(0.0000001));
    const real gA = dL_dconic[3 * i], gB = dL_dconic[3 * i + 1], gC = dL_dconic[3 * i + 2];
    real da = 0, db = 0, dc = 0;
    if (den != R(0.0)) {
      da = k2 * (-c * c * gA + b * c * gB + (den - a * c) * gC);
      dc = k2 * (-a * a * gC + a * b * gB + (den - a * c) * gA);
      db = k2 * (R(2.0) * b * c * gA - (den + R(2.0) * b * b) * gB + R(2.0) * a * b * gC);
    }
    /* (b) Sigma2D -> Sigma3D */
    real dS[6];
    dS[0] = T0[0] * T0[0] * da + T0[0] * T1[0] * db + T1[0] * T1[0] * dc;
    dS[3] = T0[1] * T0[1] * da + T0[1] * T1[1] * db + T1[1] * T1[1] * dc;
    dS[5] = T0[2] * T0[2] * da + T0[2] * T1[2] * db + T1[2] * T1[2] * dc;
    dS[1] = R(2.0) * T0[0] * T0[1] * da + (T0[0] * T1[1] + T0[1] * T1[0]) * db +
            R(2.0) * T1[0] * T1[1] * dc;
    dS[2] = R(2.0) * T0[0] * T0[2] * da + (T0[0] * T1[2] + T0[2] * T1[0]) * db +
            R(2.0) * T1[0] * T1[2] * dc;
    dS[4] = R(2.0) * T0[1] * T0[2] * da + (T0[1] * T1[2] + T0[2] * T1[1]) * db +
            R(2.0) * T1[1] * T1[2] * dc;
    for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = dS[k];
    /* (c) Sigma2D -> T -> J -> t -> mean */
    real dT0[3], dT1[3];
    for (int p = 0; p < 3; ++p) {
      dT0[p] = R(2.0) * s0[p] * da + s1[p] * db;
      dT1[p] = R(2.0) * s1[p] * dc + s0[p] * db;
    }
    const real dJ00 = dT0[0] * vm(V, 0, 0) + dT0[1] * vm(V, 0, 1) + dT0[2] * vm(V, 0, 2);
    const real dJ02 = dT0[0] * vm(V, 2, 0) + dT0[1] * vm(V, 2, 1) + dT0[2] * vm(V, 2, 2);
    const real dJ11 = dT1[0] * vm(V, 1, 0) + dT1[1] * vm(V, 1, 1) + dT1[2] * vm(V, 1, 2);
    const real dJ12 = dT1[0] * vm(V, 2, 0) + dT1[1] * vm(V, 2, 1) + dT1[2] * vm(V, 2, 2);
    const real tz_inv = R(1.0) / tz, tz2 = tz_inv * tz_inv, tz3 = tz2 * tz_inv;
    const real dtx = clx ? R(0.0) : -fx * tz2 * dJ02;
    const real dty = cly ? R(0.0) : -fy * tz2 * dJ12;
    const real dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + R(2.0) * fx * u * tz3 * dJ02 +
                     R(2.0) * fy * v * tz3 * dJ12;
    real dmean[3];
    for (int k = 0; k < 3; ++k)
      dmean[k] = vm(V, 0, k) * dtx + vm(V, 1, k) * dty + vm(V, 2, k) * dtz;
    /* (d) screen position -> mean (perspective projection) */
    const real hx = vm(PV, 0, 0) * px + vm(PV, 0, 1) * py + vm(PV, 0, 2) * pz + vm(PV, 0, 3);
    const real hy = vm(PV, 1, 0) * px + vm(PV, 1, 1) * py + vm(PV, 1, 2) * pz + vm(PV, 1, 3);
    const real hw = vm(PV, 3, 0) * px + vm(PV, 3, 1) * py + vm(PV, 3, 2) * pz + vm(PV, 3, 3);
    const real winv = R(1.0) / (hw + R(0.0000001));
    const real m1 = hx * winv * winv, m2 = hy * winv * winv;
    const real gx_ = dL_dmean2D[2 * i], gy_ = dL_dmean2D[2 * i + 1];
    for (int k = 0; k < 3; ++k)
      dmean[k] += (vm(PV, 0, k) * winv - vm(PV, 3, k) * m1) * gx_ +
                  (vm(PV, 1, k) * winv - vm(PV, 3, k) * m2) * gy_;
    for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = dmean[k];
    /* (e) Sigma3D -> scale, rotation */
    if (scales) {
      const real* q = rots + 4 * i;
      const real r = q[0], x = q[1], y = q[2], z = q[3];
      real Rm[3][3];
      Rm[0][0] = R(1.0) - R(2.0) * (y * y + z * z);
      Rm[0][1] = R(2.0) * (x * y - r * z);
      Rm[0][2] = R(2.0) * (x * z + r * y);
      Rm[1][0] = R(2.0) * (x * y + r * z);
      Rm[1][1] = R(1.0) - R(2.0) * (x * x + z * z);
      Rm[1][2] = R(2.0) * (y * z - r * x);
      Rm[2][0] = R(2.0) * (x * z - r * y);
      Rm[2][1] = R(2.0) * (y * z + r * x);
      Rm[2][2] = R(1.0) - R(2.0) * (x * x + y * y);
      real sk[3], M[3][3];
      for (int k = 0; k < 3; ++k) {
        sk[k] = cam->scale_modifier * scales[3 * i + k];
        for (int a2 = 0; a2 < 3; ++a2) M[k][a2] = sk[k] * Rm[a2][k];
      }
      /* symmetric matrix: diagonal grads, half of each summed off-diagonal grad */
      const real Dm[3][3] = {{dS[0], R(0.5) * dS[1], R(0.5) * dS[2]},
                             {R(0.5) * dS[1], dS[3], R(0.5) * dS[4]},
                             {R(0.5) * dS[2], R(0.5) * dS[4], dS[5]}};
      real dM[3][3]; /* dL/dM[k][a] = 2 sum_b M[k][b] D[b][a] */
      for (int k = 0; k < 3; ++k)
        for (int a2 = 0; a2 < 3; ++a2)
          dM[k][a2] = R(2.0) * (M[k][0] * Dm[0][a2] + M[k][1] * Dm[1][a2] + M[k][2] * Dm[2][a2]);
      real dR[3][3]; /* dL/dR[a][k] = dM[k][a] * s_k */
      for (int k = 0; k < 3; ++k) {
        /* upstream quirk: no scale_modifier factor on dL/dscale */
        dL_dscales[3 * i + k] = Rm[0][k] * dM[k][0] + Rm[1][k] * dM[k][1] + Rm[2][k] * dM[k][2];
        for (int a2 = 0; a2 < 3; ++a2) dR[a2][k] = dM[k][a2] * sk[k];
      }
      dL_drots[4 * i + 0] = R(2.0) * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] -
                                      y * dR[2][0] + x * dR[2][1]);
      dL_drots[4 * i + 1] =
          R(2.0) * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - R(2.0) * x * dR[1][1] -
                    r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - R(2.0) * x * dR[2][2]);
      dL_drots[4 * i + 2] =
          R(2.0) * (-R(2.0) * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] +
                    z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - R(2.0) * y * dR[2][2]);
      dL_drots[4 * i + 3] =
          R(2.0) * (-R(2.0) * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] -
                    R(2.0) * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
    }
  }
}

/*
 * A.1 step 9 / A.5(f) — view-dependent colour from real spherical harmonics (only used when the
 * caller passes `shs` instead of `colors_precomp`; the reference never does —
 * /root/reference/model/avatar_model.py:350-351 — but the rasterizer API offers it,
 * gaussian_renderer/__init__.py:40-47). Published 3DGS convention: real SH basis up to degree
 * 3 with the Condon–Shortley signs folded into the constants, evaluated in the direction
 * normalize(mean - campos), then +0.5 and clamped at 0 (flag kept per channel for backward).
 *   shs [P,M,3] (M >= (deg+1)^2), colors [P,3], clamped [P,3] (uint8)
 */
static const real SH0 = R(0.28209479177387814);
static const real SH1 = R(0.4886025119029199);
static const real SH2[5] = {R(1.0925484305920792), R(-1.0925484305920792), R(0.31539156525252005),
                            R(-1.0925484305920792), R(0.5462742152960396)};
static const real SH3[7] = {R(-0.5900435899266435), R(2.890611442640554), R(-0.4570457994644658),
                            R(0.3731763325901154),  R(-0.4570457994644658), R(1.445305721320277),
                            R(-0.5900435899266435)};

void gsro_sh_forward(int P, int M, int deg, const real* means3D, const real* campos,
                     const real* shs, real* colors, uint8_t* clamped) {
  for (int i = 0; i < P; ++i) {
    const real* sh = shs + (size_t)i * M * 3;
    real dx = means3D[3 * i] - campos[0], dy = means3D[3 * i + 1] - campos[1],
         dz = means3D[3 * i + 2] - campos[2];
    const real len = SQRT(dx * dx + dy * dy + dz * dz);
    const real x = dx / len, y = dy / len, z = dz / len;
    for (int c = 0; c < 3; ++c) {
      real v = SH0 * sh[0 * 3 + c];
      if (deg > 0) {
        v = v - SH1 * y * sh[1 * 3 + c] + SH1 * z * sh[2 * 3 + c] - SH1 * x * sh[3 * 3 + c];
        if (deg > 1) {
          const real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          v = v + SH2[0] * xy * sh[4 * 3 + c] + SH2[1] * yz * sh[5 * 3 + c] +
              SH2[2] * (R(2.0) * zz - xx - yy) * sh[6 * 3 + c] + SH2[3] * xz * sh[7 * 3 + c] +
              SH2[4] * (xx - yy) * sh[8 * 3 + c];
          if (deg > 2) {
            v = v + SH3[0] * y * (R(3.0) * xx - yy) * sh[9 * 3 + c] +
                SH3[1] * xy * z * sh[10 * 3 + c] +
                SH3[2] * y * (R(4.0) * zz - xx - yy) * sh[11 * 3 + c] +
                SH3[3] * z * (R(2.0) * zz - R(3.0) * xx - R(3.0) * yy) * sh[12 * 3 + c] +
                SH3[4] * x * (R(4.0) * zz - xx - yy) * sh[13 * 3 + c] +
                SH3[5] * z * (xx - yy) * sh[14 * 3 + c] +
                SH3[6] * x * (xx - R(3.0) * yy) * sh[15 * 3 + c];
          }
        }
      }
      v += R(0.5);
      clamped[3 * i + c] = v < 0;
      colors[3 * i + c] = v < 0 ? 0 : v;
    }
  }
}

/* Backward of the above: dL_dsh [P,M,3] fully overwritten (zeros beyond the active degree);
 * the view-direction term is ADDED to dL_dmeans3D [P,3]. */
void gsro_sh_backward(int P, int M, int deg, const real* means3D, const real* campos,
                      const real* shs, const uint8_t* clamped, const real* dL_dcolors,
                      real* dL_dsh, real* dL_dmeans3D) {
  for (int i = 0; i < P; ++i) {
    const real* sh = shs + (size_t)i * M * 3;
    real* dsh = dL_dsh + (size_t)i * M * 3;
    for (int k = 0; k < M * 3; ++k) dsh[k] = 0;
    const real dx = means3D[3 * i] - campos[0], dy = means3D[3 * i + 1] - campos[1],
               dz = means3D[3 * i + 2] - campos[2];
    const real sum2 = dx * dx + dy * dy + dz * dz;
    const real len = SQRT(sum2);
    const real x = dx / len, y = dy / len, z = dz / len;
    real gdir[3] = {0, 0, 0}; /* dL/d(x,y,z) of the unit direction */
    for (int c = 0; c < 3; ++c) {
      const real g = clamped[3 * i + c] ? 0 : dL_dcolors[3 * i + c];
      real gx = 0, gy = 0, gz = 0;
      dsh[0 * 3 + c] = SH0 * g;
      if (deg > 0) {
        dsh[1 * 3 + c] = -SH1 * y * g;
        dsh[2 * 3 + c] = SH1 * z * g;
        dsh[3 * 3 + c] = -SH1 * x * g;
        gx = -SH1 * sh[3 * 3 + c];
        gy = -SH1 * sh[1 * 3 + c];
        gz = SH1 * sh[2 * 3 + c];
        if (deg > 1) {
          const real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          dsh[4 * 3 + c] = SH2[0] * xy * g;
          dsh[5 * 3 + c] = SH2[1] * yz * g;
          dsh[6 * 3 + c] = SH2[2] * (R(2.0) * zz - xx - yy) * g;
          dsh[7 * 3 + c] = SH2[3] * xz * g;
          dsh[8 * 3 + c] = SH2[4] * (xx - yy) * g;
          gx += SH2[0] * y * sh[4 * 3 + c] + SH2[2] * R(2.0) * -x * sh[6 * 3 + c] +
                SH2[3] * z * sh[7 * 3 + c] + SH2[4] * R(2.0) * x * sh[8 * 3 + c];
          gy += SH2[0] * x * sh[4 * 3 + c] + SH2[1] * z * sh[5 * 3 + c] +
                SH2[2] * R(2.0) * -y * sh[6 * 3 + c] + SH2[4] * R(2.0) * -y * sh[8 * 3 + c];
          gz += SH2[1] * y * sh[5 * 3 + c] + SH2[2] * R(4.0) * z * sh[6 * 3 + c] +
                SH2[3] * x * sh[7 * 3 + c];
          if (deg > 2) {
            dsh[9 * 3 + c] = SH3[0] * y * (R(3.0) * xx - yy) * g;
            dsh[10 * 3 + c] = SH3[1] * xy * z * g;
            dsh[11 * 3 + c] = SH3[2] * y * (R(4.0) * zz - xx - yy) * g;
            dsh[12 * 3 + c] = SH3[3] * z * (R(2.0) * zz - R(3.0) * xx - R(3.0) * yy) * g;
            dsh[13 * 3 + c] = SH3[4] * x * (R(4.0) * zz - xx - yy) * g;
            dsh[14 * 3 + c] = SH3[5] * z * (xx - yy) * g;
            dsh[15 * 3 + c] = SH3[6] * x * (xx - R(3.0) * yy) * g;
            gx += SH3[0] * sh[9 * 3 + c] * R(6.0) * xy + SH3[1] * sh[10 * 3 + c] * yz +
                  SH3[2] * sh[11 * 3 + c] * -R(2.0) * xy + SH3[3] * sh[12 * 3 + c] * -R(6.0) * xz +
                  SH3[4] * sh[13 * 3 + c] * (R(4.0) * zz - R(3.0) * xx - yy) +
                  SH3[5] * sh[14 * 3 + c] * R(2.0) * xz +
                  SH3[6] * sh[15 * 3 + c] * (R(3.0) * xx - R(3.0) * yy);
            gy += SH3[0] * sh[9 * 3 + c] * (R(3.0) * xx - R(3.0) * yy) +
                  SH3[1] * sh[10 * 3 + c] * xz +
                  SH3[2] * sh[11 * 3 + c] * (R(4.0) * zz - xx - R(3.0) * yy) +
                  SH3[3] * sh[12 * 3 + c] * -R(6.0) * yz + SH3[4] * sh[13 * 3 + c] * -R(2.0) * xy +
                  SH3[5] * sh[14 * 3 + c] * -R(2.0) * yz + SH3[6] * sh[15 * 3 + c] * -R(6.0) * xy;
            gz += SH3[1] * sh[10 * 3 + c] * xy + SH3[2] * sh[11 * 3 + c] * R(8.0) * yz +
                  SH3[3] * sh[12 * 3 + c] * (R(6.0) * zz - R(3.0) * xx - R(3.0) * yy) +
                  SH3[4] * sh[13 * 3 + c] * R(8.0) * xz + SH3[5] * sh[14 * 3 + c] * (xx - yy);
          }
        }
      }
      gdir[0] += gx * g;
      gdir[1] += gy * g;
      gdir[2] += gz * g;
    }
    /* through dir = d / |d| */
    const real inv32 = R(1.0) / SQRT(sum2 * sum2 * sum2);
    dL_dmeans3D[3 * i + 0] += ((sum2 - dx * dx) * gdir[0] - dy * dx * gdir[1] - dz * dx * gdir[2]) * inv32;
    dL_dmeans3D[3 * i + 1] += (-dx * dy * gdir[0] + (sum2 - dy * dy) * gdir[1] - dz * dy * gdir[2]) * inv32;
    dL_dmeans3D[3 * i + 2] += (-dx * dz * gdir[0] - dy * dz * gdir[1] + (sum2 - dz * dz) * gdir[2]) * inv32;
  }
}
