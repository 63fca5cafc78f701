// gsr_api.hip — the extern "C" boundary declared in include/gsr.h.
//
// Replaces the pybind entry points of the un-vendored diff_gaussian_rasterization._C module
// (rasterize_gaussians / rasterize_gaussians_backward / mark_visible) reached from
// /root/reference/gaussian_renderer/__init__.py:40. Plain pointers in, error codes out.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "gsr_common.h"

namespace gsr {

namespace {
thread_local char g_err[512] = "";
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- opt-in profiler
// A caller-owned object (include/gsr.h: GsrProfile) bound to the calling thread; nothing process-global.
}  // namespace gsr

struct GsrProfile {
  struct Rec { hipEvent_t start, stop; int id; };
  std::mutex mu;                      // read may come from another thread than the one that launches
  std::vector<Rec> recs;              // recorded, not yet read
  std::vector<Rec> free_;             // recycled event pairs
  double ms[gsr::K_COUNT] = {0};
  int64_t n[gsr::K_COUNT] = {0};
};

namespace gsr {

namespace {
thread_local GsrProfile* t_prof = nullptr;
thread_local unsigned t_prof_mask = 0;     // bit k: time kernel id k
}  // namespace

ProfScope::ProfScope(KernelId id, hipStream_t s) : slot(-1), stream(s) {
  GsrProfile* p = t_prof;
  if (!p || !((t_prof_mask >> id) & 1u)) return;
  std::lock_guard<std::mutex> lk(p->mu);
  GsrProfile::Rec r;
  if (!p->free_.empty()) { r = p->free_.back(); p->free_.pop_back(); }
  else { if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return; }
  r.id = id;
  if (hipEventRecord(r.start, s) != hipSuccess) { p->free_.push_back(r); return; }
  p->recs.push_back(r);
  slot = (int)p->recs.size() - 1;
}

ProfScope::~ProfScope() {
  GsrProfile* p = t_prof;
  if (slot < 0 || !p) return;
  std::lock_guard<std::mutex> lk(p->mu);
  if (slot < (int)p->recs.size()) (void)hipEventRecord(p->recs[slot].stop, stream);
}

static inline uint64_t align_up(uint64_t v) { return (v + 255u) & ~(uint64_t)255u; }

int compute_layout(int P, int W, int H, int64_t max_pairs, GsrLayout* out) {
  if (P < 0 || W <= 0 || H <= 0 || max_pairs < 0 || max_pairs > 0xfffffff0ll) return 1;
  const Dims d = make_dims(P, W, H, max_pairs);
  const uint64_t Pn = (uint64_t)(P > 0 ? P : 1);
  const uint64_t cap = (uint64_t)(max_pairs > 0 ? max_pairs : 1);
  const uint64_t npix = (uint64_t)W * H;
  uint64_t off = 0;
  auto take = [&](uint64_t bytes) { uint64_t o = off; off = align_up(off + bytes); return o; };
  // ---- what every forward pass touches (GSR_WS_EVAL)
  out->depth = take(Pn * 4);
  out->xy = take(Pn * 8);
  out->conic_opacity = take(Pn * 16);
  out->rgb = take(Pn * 16);
  out->cov3d = take(Pn * 24);
  out->rect = take(Pn * 16);
  out->tiles_touched = take(Pn * 4);
  out->clamped = take(Pn * 4);
  out->tile_count = take((uint64_t)d.T * 4);
  out->tile_offset = take(((uint64_t)d.T + 1) * 4);
  out->tile_cursor = take((uint64_t)d.T * 4);
  out->pair_key = take(cap * 8);
  out->point_list = take(cap * 4);
  out->pair_tmp = take(cap * 8);
  out->final_T = take(npix * 4);
  out->n_contrib = take(npix * 4);
  out->status = take(8 * 4);
  out->seg_heads = take(8 * 64 * 4);            // directly behind status: one clear
  out->seg_count = take((uint64_t)d.T * GSR_SEG_BLOCKS * 4);
  out->xyext = take(Pn * 16);
  out->sort_work = take((uint64_t)sort_work_capacity(max_pairs) * 4);
  out->eval_bytes = off;
  // ---- what the forward pass leaves for the backward pass (GSR_WS_TRAIN)
  out->grad_acc = take(Pn * GSR_GRAD_STRIDE * 4);
  out->seg_entries = take((uint64_t)d.seg_cap * GSR_WAVE * 8);
  out->seg_ckpt = take((uint64_t)d.seg_cap * GSR_SEG_PIX * 16);
  out->seg_info = take((uint64_t)d.seg_cap * 8);
  out->pix_accum = take(npix * 16);
  out->seg_list = take(8 * (uint64_t)d.seg_cap * 4);
  out->train_bytes = off;
  // ---- per-pair records of the deterministic backward (GSR_WS_DEBUG)
  out->pair_grad = take(cap * GSR_PAIR_GRAD * 4);
  out->total_bytes = off;
  return 0;
}

uint64_t mode_bytes(const GsrLayout& L, int mode) {
  return mode == GSR_WS_EVAL ? L.eval_bytes : (mode == GSR_WS_TRAIN ? L.train_bytes : L.total_bytes);
}

Workspace resolve(void* base, const GsrLayout& L) {
  char* b = static_cast<char*>(base);
  Workspace w;
  w.depth = reinterpret_cast<float*>(b + L.depth);
  w.xy = reinterpret_cast<float2*>(b + L.xy);
  w.conic_opacity = reinterpret_cast<float4*>(b + L.conic_opacity);
  w.rgb = reinterpret_cast<float4*>(b + L.rgb);
  w.cov3d = reinterpret_cast<float*>(b + L.cov3d);
  w.rect = reinterpret_cast<int4*>(b + L.rect);
  w.tiles_touched = reinterpret_cast<uint32_t*>(b + L.tiles_touched);
  w.clamped = reinterpret_cast<uint8_t*>(b + L.clamped);
  w.tile_count = reinterpret_cast<uint32_t*>(b + L.tile_count);
  w.tile_offset = reinterpret_cast<uint32_t*>(b + L.tile_offset);
  w.tile_cursor = reinterpret_cast<uint32_t*>(b + L.tile_cursor);
  w.pair_key = reinterpret_cast<uint64_t*>(b + L.pair_key);
  w.point_list = reinterpret_cast<uint32_t*>(b + L.point_list);
  w.pair_tmp = reinterpret_cast<uint64_t*>(b + L.pair_tmp);
  w.final_T = reinterpret_cast<float*>(b + L.final_T);
  w.n_contrib = reinterpret_cast<uint32_t*>(b + L.n_contrib);
  w.grad_acc = reinterpret_cast<float*>(b + L.grad_acc);
  w.status = reinterpret_cast<int32_t*>(b + L.status);
  w.seg_heads = reinterpret_cast<int32_t*>(b + L.seg_heads);
  w.seg_count = reinterpret_cast<uint32_t*>(b + L.seg_count);
  w.xyext = reinterpret_cast<float4*>(b + L.xyext);
  w.seg_entries = reinterpret_cast<uint2*>(b + L.seg_entries);
  w.seg_ckpt = reinterpret_cast<float4*>(b + L.seg_ckpt);
  w.seg_info = reinterpret_cast<uint2*>(b + L.seg_info);
  w.pix_accum = reinterpret_cast<float4*>(b + L.pix_accum);
  w.pair_grad = reinterpret_cast<float*>(b + L.pair_grad);
  w.seg_list = reinterpret_cast<uint32_t*>(b + L.seg_list);
  w.sort_work = reinterpret_cast<uint32_t*>(b + L.sort_work);
  return w;
}

static int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return GSR_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return GSR_ERR_LAUNCH;
}

// gsr_set_trace(1): every stage of every call is announced on stderr and waited for (development: which kernel faulted)
static std::atomic<int> g_trace{0};

void trace_sync(hipStream_t stream, const char* what) {
  if (!g_trace.load(std::memory_order_relaxed)) return;
  fprintf(stderr, "[gsr] %s ...", what); fflush(stderr);
  const hipError_t e = hipStreamSynchronize(stream);
  fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); fflush(stderr);
}

static int debug_sync(const GsrSettings* s, hipStream_t stream, const char* what) {
  if (g_trace.load(std::memory_order_relaxed)) {
    fprintf(stderr, "[gsr] %s ...", what); fflush(stderr);
    const hipError_t e = hipStreamSynchronize(stream);
    fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); fflush(stderr);
    return check_hip(e, what);
  }
  if (!s->debug) return GSR_OK;
  return check_hip(hipStreamSynchronize(stream), what);
}

static int validate(const GsrSettings* s, int32_t P, const float* means3D,
                    const float* colors_precomp, const float* shs, int32_t sh_coeffs,
                    const float* opacities, const float* scales, const float* rotations,
                    const float* cov3D_precomp, void* workspace, size_t workspace_bytes,
                    int64_t max_pairs, int mode, GsrLayout* L) {
  if (!s) { set_error("settings is NULL"); return GSR_ERR_INVALID_ARGUMENT; }
  if (P < 0 || s->image_width <= 0 || s->image_height <= 0) {
    set_error("bad sizes: P=%d W=%d H=%d", P, s->image_width, s->image_height);
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (!s->bg || !s->viewmatrix || !s->projmatrix) {
    set_error("bg, viewmatrix and projmatrix must be device pointers");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (P == 0) {   // empty scene: torch hands out NULL data pointers for empty tensors
    if (compute_layout(P, s->image_width, s->image_height, max_pairs, L)) {
      set_error("bad workspace arguments (max_pairs=%lld)", (long long)max_pairs);
      return GSR_ERR_INVALID_ARGUMENT;
    }
    if (!workspace || workspace_bytes < mode_bytes(*L, mode)) {
      set_error("workspace too small: have %zu bytes, need %llu", workspace_bytes,
                (unsigned long long)mode_bytes(*L, mode));
      return GSR_ERR_WORKSPACE_TOO_SMALL;
    }
    return GSR_OK;
  }
  if ((colors_precomp == nullptr) == (shs == nullptr)) {
    set_error("Please provide excatly one of either SHs or precomputed colors!");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  const bool has_sr = scales != nullptr || rotations != nullptr;
  if ((has_sr && (!scales || !rotations)) || (has_sr == (cov3D_precomp != nullptr))) {
    set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (shs) {
    const int deg = s->sh_degree;
    if (deg < 0 || deg > 3 || sh_coeffs < (deg + 1) * (deg + 1)) {
      set_error("SH colours: sh_degree=%d needs 0 <= degree <= 3 and sh_coeffs >= (degree+1)^2 "
                "(got %d)", deg, sh_coeffs);
      return GSR_ERR_INVALID_ARGUMENT;
    }
    if (!s->campos) {
      set_error("SH colours need settings->campos");
      return GSR_ERR_INVALID_ARGUMENT;
    }
  }
  if (P > 0 && (!means3D || !opacities)) {
    set_error("means3D / opacities are NULL");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (compute_layout(P, s->image_width, s->image_height, max_pairs, L)) {
    set_error("bad workspace arguments (max_pairs=%lld)", (long long)max_pairs);
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (!workspace || workspace_bytes < mode_bytes(*L, mode)) {
    set_error("workspace too small: have %zu bytes, need %llu (gsr_workspace_bytes_for, mode %d)", workspace_bytes,
              (unsigned long long)mode_bytes(*L, mode), mode);
    return GSR_ERR_WORKSPACE_TOO_SMALL;
  }
  return GSR_OK;
}

}  // namespace gsr

using namespace gsr;

extern "C" {

size_t gsr_workspace_bytes(int32_t P, int32_t W, int32_t H, int64_t max_pairs) {
  GsrLayout L;
  if (compute_layout(P, W, H, max_pairs, &L)) return 0;
  return (size_t)L.total_bytes;
}

size_t gsr_workspace_bytes_for(int32_t P, int32_t W, int32_t H, int64_t max_pairs, int32_t mode) {
  GsrLayout L;
  if (mode < GSR_WS_EVAL || mode > GSR_WS_DEBUG || compute_layout(P, W, H, max_pairs, &L)) return 0;
  return (size_t)mode_bytes(L, mode);
}

int gsr_workspace_layout(int32_t P, int32_t W, int32_t H, int64_t max_pairs, GsrLayout* out) {
  if (!out || compute_layout(P, W, H, max_pairs, out)) {
    set_error("gsr_workspace_layout: invalid arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  return GSR_OK;
}

// Clears up to two regions of every frame's workspace slice in ONE launch (hipMemset2DAsync is slow
// for large rows, 65-200 us measured; a 1-D fill per frame and region is a launch each).
// Regions are 16-byte aligned multiples of 4 bytes; blockIdx.y = frame.
__global__ void __launch_bounds__(256)
clear_frames_kernel(char* base, size_t stride, size_t off_a, size_t words_a, size_t off_b, size_t words_b) {
  char* frame = base + (size_t)blockIdx.y * stride;
  uint32_t* a = reinterpret_cast<uint32_t*>(frame + off_a);
  uint32_t* b = reinterpret_cast<uint32_t*>(frame + off_b);
  const size_t step = (size_t)gridDim.x * blockDim.x;
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t quads = words_a >> 2;
  uint4* a4 = reinterpret_cast<uint4*>(a);
  for (size_t i = t0; i < quads; i += step) a4[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = (quads << 2) + t0; i < words_a; i += step) a[i] = 0u;
  for (size_t i = t0; i < words_b; i += step) b[i] = 0u;
}

static hipError_t clear_frames(void* base, size_t stride, int frames, hipStream_t stream, void* region_a,
                               size_t bytes_a, void* region_b = nullptr, size_t bytes_b = 0) {
  const size_t off_a = static_cast<char*>(region_a) - static_cast<char*>(base);
  const size_t off_b = region_b ? static_cast<char*>(region_b) - static_cast<char*>(base) : 0;
  const size_t quads = bytes_a / 16;
  const int blocks = (int)((quads + 255) / 256 < 1024 ? (quads + 255) / 256 : 1024);
  hipLaunchKernelGGL(clear_frames_kernel, dim3(blocks > 0 ? blocks : 1, frames), dim3(256), 0, stream,
                     static_cast<char*>(base), stride, off_a, bytes_a / 4, off_b, bytes_b / 4);
  return hipGetLastError();
}

static int make_batch(const GsrBatch* b, int32_t P, const GsrLayout& L, int mode, size_t workspace_bytes,
                      Batch* out) {
  if (!b || b->frames < 1 || b->frames > 65535) {
    set_error("batch descriptor: frames must be in [1, 65535]");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  const int64_t strides[] = {b->means3D_stride, b->colors_stride, b->opacities_stride,
                             b->scales_stride, b->rotations_stride, b->cov3D_stride,
                             b->viewmatrix_stride, b->projmatrix_stride, b->shs_stride,
                             b->campos_stride};
  for (int64_t v : strides)
    if (v < 0) { set_error("batch descriptor: negative stride"); return GSR_ERR_INVALID_ARGUMENT; }
  if (b->frames > 1 && b->means3D_stride < (int64_t)P * 3) {
    set_error("batch descriptor: means3D_stride must be >= 3*P (frames do not share positions)");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (workspace_bytes < (size_t)b->frames * mode_bytes(L, mode)) {
    set_error("workspace too small: have %zu bytes, need %d x %llu", workspace_bytes, b->frames,
              (unsigned long long)mode_bytes(L, mode));
    return GSR_ERR_WORKSPACE_TOO_SMALL;
  }
  out->frames = b->frames;
  out->ws_stride = (size_t)mode_bytes(L, mode);
  out->means = b->means3D_stride; out->colors = b->colors_stride; out->opacities = b->opacities_stride;
  out->scales = b->scales_stride; out->rotations = b->rotations_stride; out->cov3d = b->cov3D_stride;
  out->view = b->viewmatrix_stride; out->proj = b->projmatrix_stride;
  out->shs = b->shs_stride; out->campos = b->campos_stride;
  return GSR_OK;
}

// the workspace mode of a differentiable render: forward and backward derive it from settings.debug
static int train_mode(const GsrSettings* s) { return s && s->debug ? GSR_WS_DEBUG : GSR_WS_TRAIN; }

static int forward_impl(bool record, const GsrSettings* s, const GsrBatch* batch, int32_t P, const float* means3D,
                        const float* colors_precomp, const float* shs, int32_t sh_coeffs,
                        const float* opacities, const float* scales, const float* rotations,
                        const float* cov3D_precomp, void* workspace, size_t workspace_bytes,
                        int64_t max_pairs, float* out_color, int32_t* out_radii, void* stream_) {
  GsrLayout L;
  const int mode = record ? train_mode(s) : GSR_WS_EVAL;
  int rc = validate(s, P, means3D, colors_precomp, shs, sh_coeffs, opacities, scales, rotations,
                    cov3D_precomp, workspace, workspace_bytes, max_pairs, mode, &L);
  if (rc) return rc;
  if (!out_color || (P > 0 && !out_radii)) {
    set_error("out_color / out_radii are NULL");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  Batch bt;
  if ((rc = make_batch(batch, P, L, mode, workspace_bytes, &bt))) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const Dims d = make_dims(P, s->image_width, s->image_height, max_pairs);
  const Workspace ws = resolve(workspace, L);
  // tile_count .. tile_cursor are contiguous in the layout: one launch clears the histogram and the
  // status words of every frame
  if ((rc = check_hip(clear_frames(workspace, bt.ws_stride, bt.frames, stream, ws.tile_count,
                                   L.tile_cursor - L.tile_count, ws.status, (size_t)(L.seg_count - L.status)),
                      "clear tile_count/status")))
    return rc;
  if ((rc = check_hip(launch_preprocess(*s, d, means3D, colors_precomp, opacities, scales,
                                        rotations, cov3D_precomp, ws, out_radii, bt, stream),
                      "preprocess")))
    return rc;
  if (shs && (rc = check_hip(launch_sh_color(*s, d, means3D, shs, sh_coeffs, ws, bt, stream),
                             "sh_color")))
    return rc;
  if ((rc = debug_sync(s, stream, "preprocess (sync)"))) return rc;
  if ((rc = check_hip(launch_binning(d, ws, bt, stream), "binning"))) return rc;
  if ((rc = debug_sync(s, stream, "binning (sync)"))) return rc;
  if ((rc = check_hip(launch_render_fwd(*s, d, ws, out_color, record, bt, stream), "render_fwd"))) return rc;
  if ((rc = debug_sync(s, stream, "render_fwd (sync)"))) return rc;
  return GSR_OK;
}

int gsr_forward_batch(const GsrSettings* s, const GsrBatch* batch, int32_t P, const float* means3D,
                      const float* colors_precomp, const float* shs, int32_t sh_coeffs,
                      const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp, void* workspace, size_t workspace_bytes,
                      int64_t max_pairs, float* out_color, int32_t* out_radii, void* stream_) {
  return forward_impl(true, s, batch, P, means3D, colors_precomp, shs, sh_coeffs, opacities, scales, rotations,
                      cov3D_precomp, workspace, workspace_bytes, max_pairs, out_color, out_radii, stream_);
}

int gsr_forward_eval_batch(const GsrSettings* s, const GsrBatch* batch, int32_t P, const float* means3D,
                           const float* colors_precomp, const float* shs, int32_t sh_coeffs,
                           const float* opacities, const float* scales, const float* rotations,
                           const float* cov3D_precomp, void* workspace, size_t workspace_bytes,
                           int64_t max_pairs, float* out_color, int32_t* out_radii, void* stream_) {
  return forward_impl(false, s, batch, P, means3D, colors_precomp, shs, sh_coeffs, opacities, scales, rotations,
                      cov3D_precomp, workspace, workspace_bytes, max_pairs, out_color, out_radii, stream_);
}

int gsr_backward_batch(const GsrSettings* s, const GsrBatch* batch, int32_t P, const float* means3D,
                       const float* colors_precomp, const float* shs, int32_t sh_coeffs,
                       const float* opacities, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const int32_t* radii, void* workspace,
                       size_t workspace_bytes, int64_t max_pairs, const float* dL_dout_color,
                       float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors, float* dL_dsh,
                       float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                       float* dL_dcov3D, int32_t* overflow_flag, void* stream_) {
  GsrLayout L;
  const int mode = train_mode(s);
  int rc = validate(s, P, means3D, colors_precomp, shs, sh_coeffs, opacities, scales, rotations,
                    cov3D_precomp, workspace, workspace_bytes, max_pairs, mode, &L);
  if (rc) return rc;
  if (!dL_dout_color || (P > 0 && !radii)) {
    set_error("dL_dout_color / radii are NULL");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  Batch bt;
  if ((rc = make_batch(batch, P, L, mode, workspace_bytes, &bt))) return rc;
  if (P == 0) return GSR_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const Dims d = make_dims(P, s->image_width, s->image_height, max_pairs);
  const Workspace ws = resolve(workspace, L);
  if ((rc = check_hip(clear_frames(workspace, bt.ws_stride, bt.frames, stream, ws.grad_acc,
                                   (size_t)P * GSR_GRAD_STRIDE * sizeof(float)),
                      "clear grad_acc")))
    return rc;
  if ((rc = check_hip(launch_render_bwd(*s, d, ws, dL_dout_color, bt, stream), "render_bwd")))
    return rc;
  if ((rc = debug_sync(s, stream, "render_bwd (sync)"))) return rc;
  if ((rc = check_hip(launch_preprocess_bwd(*s, d, means3D, scales, rotations, radii, ws,
                                            dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity,
                                            dL_dscales, dL_drotations, dL_dcov3D, overflow_flag, bt, stream),
                      "preprocess_bwd")))
    return rc;
  if (shs && (dL_dsh || dL_dmeans3D) &&
      (rc = check_hip(launch_sh_bwd(*s, d, means3D, shs, sh_coeffs, ws, dL_dsh, dL_dmeans3D, bt,
                                    stream), "sh_bwd")))
    return rc;
  if ((rc = debug_sync(s, stream, "preprocess_bwd (sync)"))) return rc;
  return GSR_OK;
}

static const GsrBatch kSingleFrame = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

int gsr_forward(const GsrSettings* s, int32_t P, const float* means3D,
                const float* colors_precomp, const float* shs, int32_t sh_coeffs,
                const float* opacities, const float* scales, const float* rotations,
                const float* cov3D_precomp, void* workspace, size_t workspace_bytes,
                int64_t max_pairs, float* out_color, int32_t* out_radii, void* stream_) {
  return gsr_forward_batch(s, &kSingleFrame, P, means3D, colors_precomp, shs, sh_coeffs, opacities,
                           scales, rotations, cov3D_precomp, workspace, workspace_bytes, max_pairs,
                           out_color, out_radii, stream_);
}

int gsr_backward(const GsrSettings* s, int32_t P, const float* means3D,
                 const float* colors_precomp, const float* shs, int32_t sh_coeffs,
                 const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp, const int32_t* radii, void* workspace,
                 size_t workspace_bytes, int64_t max_pairs, const float* dL_dout_color,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors, float* dL_dsh,
                 float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                 int32_t* overflow_flag, void* stream_) {
  return gsr_backward_batch(s, &kSingleFrame, P, means3D, colors_precomp, shs, sh_coeffs, opacities,
                            scales, rotations, cov3D_precomp, radii, workspace, workspace_bytes,
                            max_pairs, dL_dout_color, dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dsh,
                            dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, overflow_flag, stream_);
}

int gsr_forward_eval(const GsrSettings* s, int32_t P, const float* means3D,
                     const float* colors_precomp, const float* shs, int32_t sh_coeffs,
                     const float* opacities, const float* scales, const float* rotations,
                     const float* cov3D_precomp, void* workspace, size_t workspace_bytes,
                     int64_t max_pairs, float* out_color, int32_t* out_radii, void* stream_) {
  return gsr_forward_eval_batch(s, &kSingleFrame, P, means3D, colors_precomp, shs, sh_coeffs, opacities,
                                scales, rotations, cov3D_precomp, workspace, workspace_bytes, max_pairs,
                                out_color, out_radii, stream_);
}

int gsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* out_visible, void* stream_) {
  (void)projmatrix;
  if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !out_visible))) {
    set_error("gsr_mark_visible: invalid arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  return check_hip(launch_mark_visible(P, means3D, viewmatrix, out_visible,
                                       static_cast<hipStream_t>(stream_)), "mark_visible");
}

namespace {
// [max pairs of a frame, any overflow, max segments of a frame, longest tile list, total pairs, frames, max 6, max 7]
__global__ void batch_status_kernel(const char* ws, size_t status_off, size_t ws_stride, int frames,
                                    int32_t* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t total = 0;
  int32_t mx0 = 0, mx1 = 0, mx2 = 0, mx3 = 0, mx6 = 0, mx7 = 0;
  for (int f = 0; f < frames; ++f) {
    const int32_t* st = reinterpret_cast<const int32_t*>(ws + (size_t)f * ws_stride + status_off);
    mx0 = max(mx0, st[0]); mx1 = max(mx1, st[1]); mx2 = max(mx2, st[2]); mx3 = max(mx3, st[3]);
    total += st[0]; mx6 = max(mx6, st[6]); mx7 = max(mx7, st[7]);
  }
  out[0] = mx0; out[1] = mx1; out[2] = mx2; out[3] = mx3;
  out[4] = (int32_t)(total > 0x7fffffff ? 0x7fffffff : total); out[5] = frames; out[6] = mx6; out[7] = mx7;
}
}  // namespace

int gsr_batch_status(const void* workspace, int32_t frames, int32_t P, int32_t W, int32_t H,
                     int64_t max_pairs, int32_t mode, int32_t* status_dev, void* stream_) {
  GsrLayout L;
  if (!workspace || !status_dev || frames < 1 || mode < GSR_WS_EVAL || mode > GSR_WS_DEBUG ||
      compute_layout(P, W, H, max_pairs, &L)) {
    set_error("gsr_batch_status: invalid arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL(batch_status_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_),
                     static_cast<const char*>(workspace), (size_t)L.status, (size_t)mode_bytes(L, mode), frames,
                     status_dev);
  return check_hip(hipGetLastError(), "batch_status_kernel");
}

int gsr_read_status(const void* workspace, int32_t P, int32_t W, int32_t H, int64_t max_pairs,
                    int32_t* status_host, void* stream_) {
  GsrLayout L;
  if (!workspace || !status_host || compute_layout(P, W, H, max_pairs, &L)) {
    set_error("gsr_read_status: invalid arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_hip(hipMemcpyAsync(status_host, static_cast<const char*>(workspace) + L.status,
                                    8 * sizeof(int32_t), hipMemcpyDeviceToHost, stream),
                     "status copy");
  if (rc) return rc;
  return check_hip(hipStreamSynchronize(stream), "status sync");
}

GsrProfile* gsr_profile_create(void) { return new (std::nothrow) GsrProfile(); }

void gsr_profile_destroy(GsrProfile* p) {
  if (!p) return;
  if (t_prof == p) { t_prof = nullptr; t_prof_mask = 0; }
  for (auto* v : {&p->recs, &p->free_})
    for (auto& r : *v) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  delete p;
}

int gsr_profile_bind(GsrProfile* p, int mask) {
  t_prof = (p && mask) ? p : nullptr;
  t_prof_mask = p ? (unsigned)mask : 0u;
  return GSR_OK;
}

int gsr_profile_read(GsrProfile* p, double* ms_sum, int64_t* launches, int reset) {
  if (!p) { set_error("gsr_profile_read: profile is NULL"); return GSR_ERR_INVALID_ARGUMENT; }
  std::lock_guard<std::mutex> lk(p->mu);
  for (auto& r : p->recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.stop) == hipSuccess &&
        hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      p->ms[r.id] += ms;
      p->n[r.id] += 1;
    }
    p->free_.push_back(r);
  }
  p->recs.clear();
  for (int k = 0; k < K_COUNT; ++k) {
    if (ms_sum) ms_sum[k] = p->ms[k];
    if (launches) launches[k] = p->n[k];
    if (reset) { p->ms[k] = 0; p->n[k] = 0; }
  }
  return GSR_OK;
}

const char* gsr_profile_kernel_name(int id) {
  static const char* names[K_COUNT] = {"preprocess", "tile_scan", "scatter", "tile_sort",
                                       "render_fwd", "render_bwd", "preprocess_bwd"};
  return (id >= 0 && id < K_COUNT) ? names[id] : "";
}

int gsr_render_block_edge(void) { return GSR_SUB; }

void gsr_set_trace(int on) { g_trace.store(on ? 1 : 0, std::memory_order_relaxed); }

const char* gsr_last_error(void) { return g_err; }

int gsr_abi_version(void) { return GSR_ABI_VERSION; }

}  // extern "C"
