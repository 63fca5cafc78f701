/*
 * gsr.h — C ABI of the MI355X (gfx950) differentiable Gaussian-splatting rasterizer.
 *
 * This is the drop-in boundary for the native side of the reference's renderer:
 *
 *   reference call site            /root/reference/gaussian_renderer/__init__.py:6,21-48
 *                                  (`from diff_gaussian_rasterization import
 *                                    GaussianRasterizationSettings, GaussianRasterizer`)
 *   native entry points replaced   the un-vendored `diff_gaussian_rasterization._C`
 *                                  pybind module: `rasterize_gaussians`,
 *                                  `rasterize_gaussians_backward`, `mark_visible`
 *                                  (SURVEY.md §2.1 / §8b — behavioural spec, Appendix A).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *     the caller owns every buffer (in the Python binding they are torch tensors,
 *     so the caching allocator and the autograd ctx manage lifetime);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing
 *     blocks the host, there is no device->host read-back on the hot path;
 *   - matrices are the 16 floats of the torch tensors the reference passes
 *     (`world_view_transform`, `full_proj_transform`): row-major storage of the
 *     transposed matrix, i.e. element (row i, col j) of the column-vector-form
 *     matrix is m[j*4+i]  (/root/reference/scene/dataset_mono.py:248-255);
 *   - return value 0 = success; non-zero = error, text via gsr_last_error()
 *     (thread-local). The library has no process-global mutable state and is re-entrant: the
 *     opt-in profiler records into a caller-owned object bound to the calling thread
 *     (gsr_profile_*).
 *
 * Plain C; no HIP or torch types appear in any signature.
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 6
#define GSR_TILE 16              /* tile edge in pixels (BLOCK_X = BLOCK_Y = 16)       */
#define GSR_NUM_CHANNELS 3

/* Error codes */
#define GSR_OK 0
#define GSR_ERR_INVALID_ARGUMENT 1
#define GSR_ERR_WORKSPACE_TOO_SMALL 2
#define GSR_ERR_LAUNCH 3
#define GSR_ERR_UNSUPPORTED 4

/* Per-call render settings. Mirrors the 12-field GaussianRasterizationSettings tuple the
 * reference builds at gaussian_renderer/__init__.py:21-34 (bg/viewmatrix/projmatrix/campos
 * are device tensors there and stay device pointers here). */
typedef struct GsrSettings {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  float scale_modifier;
  int32_t sh_degree;        /* active SH degree (0..3); ignored when colors_precomp is given */
  int32_t prefiltered;      /* accepted for API parity; no effect (as upstream)              */
  int32_t debug;            /* 1: synchronise + check after every launch                      */
  const float* bg;          /* [3]  device                                                    */
  const float* viewmatrix;  /* [16] device                                                    */
  const float* projmatrix;  /* [16] device                                                    */
  const float* campos;      /* [3]  device (only read for SH colours)                         */
} GsrSettings;

/* Byte offsets of the named sub-arrays inside the opaque workspace (all 256-B aligned).
 * The workspace is the state the forward pass leaves for the backward pass (the
 * reference's geomBuffer / binningBuffer / imgBuffer rolled into one caller-owned
 * allocation) — published so that parity tests can read the integer outputs. */
typedef struct GsrLayout {
  uint64_t total_bytes;
  /* per Gaussian, P entries */
  uint64_t depth;          /* float   [P]    view-space z                                   */
  uint64_t xy;             /* float2  [P]    pixel-space centre                             */
  uint64_t conic_opacity;  /* float4  [P]    (A, B, C, opacity)                             */
  uint64_t rgb;            /* float4  [P]    colour used for blending (.w unused)           */
  uint64_t cov3d;          /* float   [P,6]  world-space covariance (xx,xy,xz,yy,yz,zz)     */
  uint64_t rect;           /* int32   [P,4]  tile rect x0,y0,x1,y1 (half-open)              */
  uint64_t tiles_touched;  /* uint32  [P]                                                   */
  uint64_t clamped;        /* uint8   [P,4]  SH clamp flags rgb, pad                        */
  /* per tile, T = ceil(W/16)*ceil(H/16) */
  uint64_t tile_count;     /* uint32  [T]    pairs per tile (histogram); after the scan: tile ids, largest lists first */
  uint64_t tile_offset;    /* uint32  [T+1]  exclusive scan; [T] = total pairs D            */
  uint64_t tile_cursor;    /* uint32  [T]    scatter cursors (scratch)                      */
  /* per (tile,Gaussian) pair, capacity max_pairs */
  uint64_t pair_key;       /* uint64  [cap]  (depth_bits << 32) | gaussian index            */
  uint64_t point_list;     /* uint32  [cap]  per-tile lists, depth-sorted Gaussian indices  */
  uint64_t pair_tmp;       /* uint64  [cap]  merge ping-pong buffer for oversized tiles     */
  /* per pixel */
  uint64_t final_T;        /* float   [H*W]                                                 */
  uint64_t n_contrib;      /* uint32  [H*W]                                                 */
  /* backward scratch: per-Gaussian screen-space gradient accumulators */
  uint64_t grad_acc;       /* float   [P,16] (dxy2, dconic3, dopac1, drgb3, pad7): one 64-byte line each */
  /* status words */
  uint64_t status;         /* int32   [8]  [0]=pairs needed (D) [1]=overflow flag (pair buffer)
                                           [2]=unused  [3]=max pairs in one tile
                                           [5]=long-list sort buckets (sort_work items)       */
  uint64_t seg_heads;      /* int32   [8,64] word 0 of row x = recorded segments of XCD class x (entries of
                                           seg_list[x]); one 256-byte line per counter         */
  uint64_t seg_count;      /* uint32  [T,B]  segments the forward pass recorded per (tile, render block); B = 16 blocks of 4x4 pixels */
  uint64_t xyext;          /* float4  [P]    (pixel-space centre, half extents of the alpha>=1/255 box) */
  /* what the forward pass consumed, for the backward pass: segments of up to 64 list entries that survived the
   * cull of one 4x4 pixel block. Slots are addressed without counters: the B = 16 blocks of tile t own the slots
   * B (tile_offset[t] / 64 + t) + b c + s  (block b, its s-th segment, c = the tile's per-block capacity) — an
   * exact bound, S = max_pairs B / 64 + B T + B slots, so recording never overflows */
  uint64_t seg_entries;    /* uint32  [S,64,2] (Gaussian index, position in the tile's list)  */
  uint64_t seg_ckpt;       /* float   [S,16,4] (T, C.rgb) of the block's pixels at the segment's start */
  uint64_t seg_info;       /* uint32  [S,2]    (block x0 | y0<<16, entries in the segment)    */
  uint64_t pix_accum;      /* float   [H*W,4]  (C.rgb without background, final T); written for blocks
                                              with at least one segment                      */
  uint64_t pair_grad;      /* float   [cap,9]  per (tile, Gaussian) pair gradient records of the deterministic
                                              (settings.debug) backward pass                 */
  uint64_t seg_list;       /* uint32  [8,S]  the recorded segments as dense lists of slot ids, one per XCD class
                                              (tiles of rank = x mod 8), what the segment-parallel backward strides over */
  /* The sub-arrays are ordered by who needs them, so that a workspace may stop early:
   *   [0, eval_bytes)    everything a forward-only render touches (gsr_forward_eval*)
   *   [0, train_bytes)   + what the forward pass leaves for the backward pass (grad_acc, seg_*, pix_accum)
   *   [0, total_bytes)   + pair_grad, the per-pair records of the deterministic backward (settings.debug) */
  uint64_t eval_bytes;
  uint64_t train_bytes;
  /* ABI 6: work list of the tile sort for lists beyond 8192 keys — one item (tile << 8 | bucket) per ~4096-key output
   * bucket of such a list, written by the tile scan, consumed by the merge launch; 3 max_pairs / 8192 + 16 entries
   * (inside the eval region). status[5] = items of the frame. */
  uint64_t sort_work;      /* uint32  [3 cap / 8192 + 16] */
} GsrLayout;

/* workspace modes (gsr_workspace_bytes_for): per (tile, Gaussian) pair of capacity a forward-only workspace holds
 * 20 bytes, a training workspace ~240 bytes (segment records), the deterministic backward 36 more */
#define GSR_WS_EVAL 0
#define GSR_WS_TRAIN 1
#define GSR_WS_DEBUG 2

/* Size in bytes of the workspace for P Gaussians, a W x H image and room for
 * `max_pairs` (tile,Gaussian) pairs, large enough for every mode (= GSR_WS_DEBUG). Returns 0 on
 * invalid arguments. */
size_t gsr_workspace_bytes(int32_t P, int32_t W, int32_t H, int64_t max_pairs);

/* The same for one mode: GSR_WS_EVAL (gsr_forward_eval*), GSR_WS_TRAIN (gsr_forward + gsr_backward with
 * settings.debug = 0), GSR_WS_DEBUG (settings.debug = 1). In the batched calls the frames' workspaces lie
 * gsr_workspace_bytes_for(mode of the call) apart: eval for gsr_forward_eval_batch, debug when settings.debug is
 * set, train otherwise — forward and backward of one render must therefore agree on settings.debug. */
size_t gsr_workspace_bytes_for(int32_t P, int32_t W, int32_t H, int64_t max_pairs, int32_t mode);

/* Fill `out` with the sub-array offsets for the same arguments. */
int gsr_workspace_layout(int32_t P, int32_t W, int32_t H, int64_t max_pairs, GsrLayout* out);

/*
 * Forward: replaces `_C.rasterize_gaussians`.
 *
 *   means3D        [P,3]
 *   colors_precomp [P,3] or NULL      exactly one of colors_precomp / shs
 *   shs            [P,M,3] or NULL    M = sh_coeffs
 *   opacities      [P] (the reference passes [P,1])
 *   scales         [P,3] and rotations [P,4] (r,x,y,z; NOT normalised), or cov3D_precomp [P,6]
 *   workspace      gsr_workspace_bytes(P,W,H,max_pairs) bytes; kept for backward
 *   out_color      [3,H,W]  (CHW)
 *   out_radii      [P] int32; 0 = not rendered
 *
 * If the scene needs more than max_pairs pairs the call still returns 0 (it cannot know
 * without a host sync): status[1] is set to 1, status[0] holds the required count, the
 * surplus pairs are dropped. Callers poll status (gsr_read_status) after the stream has
 * drained, or asynchronously — see INTEGRATION.md.
 */
int gsr_forward(const GsrSettings* settings, int32_t P,
                const float* means3D, const float* colors_precomp,
                const float* shs, int32_t sh_coeffs,
                const float* opacities, const float* scales, const float* rotations,
                const float* cov3D_precomp,
                void* workspace, size_t workspace_bytes, int64_t max_pairs,
                float* out_color, int32_t* out_radii, void* stream);

/*
 * Forward without a backward pass to follow (the reference's evaluation / novel-pose renders under
 * torch.no_grad(), /root/reference/eval.py:42,65, /root/reference/render_novel_pose.py:30-32): same arguments and
 * results as gsr_forward, but the blend kernel records nothing for a backward pass (no segment records, no
 * per-pixel accumulators: ~40 bytes per (tile, Gaussian) pair and 16 per pixel less written) and the workspace
 * may be the small one (GSR_WS_EVAL). gsr_backward must not be called on such a workspace.
 */
int gsr_forward_eval(const GsrSettings* settings, int32_t P,
                     const float* means3D, const float* colors_precomp,
                     const float* shs, int32_t sh_coeffs,
                     const float* opacities, const float* scales, const float* rotations,
                     const float* cov3D_precomp,
                     void* workspace, size_t workspace_bytes, int64_t max_pairs,
                     float* out_color, int32_t* out_radii, void* stream);

/*
 * Backward: replaces `_C.rasterize_gaussians_backward` (SURVEY.md §2.1 rows 7-9,
 * Appendix A.4-A.5). Same inputs as forward plus the forward's workspace and
 *   dL_dout_color [3,H,W].
 * Outputs (any of them may be NULL = "not needed", the work is skipped where possible):
 *   dL_dmeans3D [P,3], dL_dmeans2D [P,3] (z column 0), dL_dcolors [P,3], dL_dsh [P,M,3],
 *   dL_dopacity [P], dL_dscales [P,3], dL_drotations [P,4], dL_dcov3D [P,6].
 * Every non-NULL output is fully overwritten (zeros for Gaussians with radii == 0).
 * The workspace is not consumed: backward may be called repeatedly (retain_graph).
 * Overflowed forward passes (status[1] == 1: the tile lists were truncated) yield NO gradient: every output of
 * that frame is written as zeros, and if `overflow_flag` (device int32[1], may be NULL) is given it is set to 1
 * — without a host sync. An optimiser that reads the flag (ganet_adam_step) skips its step, so a step computed
 * from truncated tile lists is never applied; the caller clears the flag when it starts a new step.
 */
int gsr_backward(const GsrSettings* settings, int32_t P,
                 const float* means3D, const float* colors_precomp,
                 const float* shs, int32_t sh_coeffs,
                 const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp,
                 const int32_t* radii,
                 void* workspace, size_t workspace_bytes, int64_t max_pairs,
                 const float* dL_dout_color,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors, float* dL_dsh,
                 float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                 float* dL_dcov3D, int32_t* overflow_flag, void* stream);

/*
 * Batched variants: `frames` frames that share P, image size, FoV and background are rendered by
 * ONE launch of every kernel (grid.y = frame). The reference renders the frames of a batch one
 * after the other (/root/reference/model/avatar_model.py:332-365); the per-tile work of one
 * frame leaves most of the chip idle, so batching the frames is what fills it.
 * Strides are in elements between consecutive frames; 0 = the array is shared by all frames
 * (stage 1 shares colours, scales, opacities and rotations). settings->viewmatrix/projmatrix
 * point at frame 0. The workspace is frames x gsr_workspace_bytes(...) (frame f at offset
 * f * gsr_workspace_bytes); out_color [frames,3,H,W], radii [frames,P] and every gradient
 * [frames,P,...] are contiguous.
 */
typedef struct GsrBatch {
  int32_t frames;
  int64_t means3D_stride;
  int64_t colors_stride;
  int64_t opacities_stride;
  int64_t scales_stride;
  int64_t rotations_stride;
  int64_t cov3D_stride;
  int64_t viewmatrix_stride;
  int64_t projmatrix_stride;
  int64_t shs_stride;        /* only read when shs != NULL                                     */
  int64_t campos_stride;     /* only read when shs != NULL                                     */
} GsrBatch;

int gsr_forward_batch(const GsrSettings* settings, const GsrBatch* batch, int32_t P,
                      const float* means3D, const float* colors_precomp,
                      const float* shs, int32_t sh_coeffs,
                      const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp,
                      void* workspace, size_t workspace_bytes, int64_t max_pairs,
                      float* out_color, int32_t* out_radii, void* stream);

int gsr_forward_eval_batch(const GsrSettings* settings, const GsrBatch* batch, int32_t P,
                           const float* means3D, const float* colors_precomp,
                           const float* shs, int32_t sh_coeffs,
                           const float* opacities, const float* scales, const float* rotations,
                           const float* cov3D_precomp,
                           void* workspace, size_t workspace_bytes, int64_t max_pairs,
                           float* out_color, int32_t* out_radii, void* stream);

int gsr_backward_batch(const GsrSettings* settings, const GsrBatch* batch, int32_t P,
                       const float* means3D, const float* colors_precomp,
                       const float* shs, int32_t sh_coeffs,
                       const float* opacities, const float* scales, const float* rotations,
                       const float* cov3D_precomp,
                       const int32_t* radii,
                       void* workspace, size_t workspace_bytes, int64_t max_pairs,
                       const float* dL_dout_color,
                       float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors, float* dL_dsh,
                       float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                       float* dL_dcov3D, int32_t* overflow_flag, void* stream);

/* Replaces `_C.mark_visible`: out_visible[i] = 1 if Gaussian i passes the near-plane
 * test of the forward pass (view-space z > 0.2). out_visible is uint8 [P]. */
int gsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* out_visible, void* stream);

/* Batched launches: combine the per-frame status words of `frames` consecutive workspaces (`mode`: the
 * GSR_WS_* mode of the call that filled them, i.e. their spacing) into ONE device record status_dev[8] = [max
 * pairs needed by a frame, any overflow, sum of word 2, longest tile list, total pairs of all frames, frames, max
 * word 6, max word 7] (asynchronous, on `stream`). */
int gsr_batch_status(const void* workspace, int32_t frames, int32_t P, int32_t W, int32_t H,
                     int64_t max_pairs, int32_t mode, int32_t* status_dev, void* stream);

/* Blocking helper: waits for `stream`, copies the 8 status words to status_host. */
int gsr_read_status(const void* workspace, int32_t P, int32_t W, int32_t H,
                    int64_t max_pairs, int32_t* status_host, void* stream);

/*
 * Optional per-kernel timing (bench/profiling only). The caller owns a GsrProfile object and binds it to the
 * calling thread; from then on every kernel this thread launches through the library whose id is in `mask` is
 * bracketed by hipEvents recorded on the launch stream and accounted to that object (bind NULL or mask 0 to stop).
 * gsr_profile_read waits for the recorded events and returns, per kernel id, the summed GPU time in milliseconds and
 * the number of launches since the last reset. Kernel ids: 0 preprocess, 1 tile_scan, 2 scatter, 3 tile_sort,
 * 4 render_fwd, 5 render_bwd, 6 preprocess_bwd. No process-global state: the binding is thread-local, like the
 * error text.
 */
#define GSR_NUM_KERNELS 7
typedef struct GsrProfile GsrProfile;
GsrProfile* gsr_profile_create(void);
void gsr_profile_destroy(GsrProfile* profile);       /* unbinds it from the calling thread if bound */
int gsr_profile_bind(GsrProfile* profile, int mask); /* bit k = time kernel id k; 0x7f = all */
int gsr_profile_read(GsrProfile* profile, double* ms_sum, int64_t* launches, int reset);
const char* gsr_profile_kernel_name(int id);

/* Edge in pixels of the square pixel block one wave renders (4): a tile has (16 / edge)^2 blocks, the shapes of
 * seg_count [T, blocks], seg_ckpt [S, edge^2, 4] and S = max_pairs * blocks / 64 + blocks * (T + 1) follow from it. */
int gsr_render_block_edge(void);

/* Text of the last error raised on the calling thread ("" if none). */
const char* gsr_last_error(void);

/* Development aid (ABI 6): with on != 0 every stage of every call announces itself on stderr and is waited for, so that
 * a device fault can be attributed to a kernel. Process-wide; off by default. */
void gsr_set_trace(int on);

/* ABI version of the loaded library (== GSR_ABI_VERSION of the header it was built from). */
int gsr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
