// ganet_decoder.hip — the whole decoder MLP (forward, backward) as ONE native call each: the fixed launch sequence of
// the fused layer kernels that gaussianavatar_amd/fused.py::_DecoderFn issues one ctypes call at a time (which stays
// as the path for evaluation mode, multi-rank BatchNorm statistics and ragged row counts, and as this file's
// specification: tests/test_fused_gpu.py compares the two). ~25 launches forward, ~45 backward: from Python that is
// ~1.3 ms of host time per training iteration against ~4 ms of GPU time; from here a launch is a few microseconds.
//
// Architecture (the reference's ShapeDecoder, /root/reference/model/modules.py:508-582): layers 0..4 = conv1..conv5
// (conv5 takes [x | act(conv4)]), then per head h = 0..2: layer 5 + 2 h = conv6*, 6 + 2 h = conv7*, and conv8* (3 / 1 / 3
// columns). Every hidden layer is Conv1d(k = 1) -> BatchNorm1d (batch statistics) -> Softplus.
#include <cstdint>
#include <cstring>

#include "ganet.h"
#include "ganet_common.h"
#include "ganet_mlp_common.h"

namespace ganet {

namespace {

constexpr int NL = GANET_DEC_LAYERS;      // 11 BatchNorm layers
constexpr int H = 128;
constexpr int XP = 72;                    // decoder input padded to 8-float blocks

// W1 [128, cin] -> w1p [128, 72] (zero padded); W5 [128, cin + 128] -> w5p [128, 72 + 128] (pad between the halves)
__global__ void pad_weights_kernel(int cin, const float* __restrict__ W1, const float* __restrict__ W5,
                                   float* __restrict__ w1p, float* __restrict__ w5p) {
  const int n = blockIdx.x;
  for (int k = threadIdx.x; k < XP + H; k += blockDim.x) {
    if (k < XP) {
      w1p[n * XP + k] = k < cin ? W1[n * cin + k] : 0.f;
      w5p[n * (XP + H) + k] = k < cin ? W5[n * (cin + H) + k] : 0.f;
    } else {
      w5p[n * (XP + H) + k] = W5[n * (cin + H) + cin + (k - XP)];
    }
  }
}

// dW1 [128, cin] <- dW0p [128, ldp][:, :cin];  dW5 [128, cin + 128] <- [ dWx [128, ldp][:, :cin] | dWy [128, 128] ]
// (ldp = 72 from the separate weight-gradient kernel, 128 from the one-pass kernel's 128 x 128 tile)
__global__ void assemble_wgrads_kernel(int cin, int ldp, const float* __restrict__ dW0p, const float* __restrict__ dWx,
                                       const float* __restrict__ dWy, float* __restrict__ dW1,
                                       float* __restrict__ dW5) {
  const int n = blockIdx.x;
  for (int k = threadIdx.x; k < cin + H; k += blockDim.x) {
    if (k < cin) {
      dW1[n * cin + k] = dW0p[n * ldp + k];
      dW5[n * (cin + H) + k] = dWx[n * ldp + k];
    } else {
      dW5[n * (cin + H) + k] = dWy[n * H + (k - cin)];
    }
  }
}

struct Sweep {      // alternate the row sweep of consecutive big launches (include/ganet.h: row_order)
  int k = 0;
  int next() { ++k; return (k & 1) ? GANET_ROWS_UP : GANET_ROWS_DOWN; }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct SavedView {
  float* z[NL];
  float* stat[NL];      // mean | rstd | scale | shift, 128 floats each
};
SavedView view_saved(float* saved, int64_t M) {
  SavedView v;
  for (int i = 0; i < NL; ++i) v.z[i] = saved + (size_t)i * M * H;
  float* st = saved + (size_t)NL * M * H;
  for (int i = 0; i < NL; ++i) v.stat[i] = st + (size_t)i * 4 * H;
  return v;
}

bool params_ok(const GanetDecoderParams* p) {
  if (!p || p->cin <= 0 || p->cin > XP) return false;
  for (int i = 0; i < NL; ++i)
    if (!p->W[i] || !p->bias[i] || !p->gamma[i] || !p->beta[i] ||
        ((p->running_mean[i] == nullptr) != (p->running_var[i] == nullptr))) return false;
  for (int j = 0; j < 3; ++j)
    if (!p->W8[j] || !p->b8[j] || p->n8[j] <= 0 || p->n8[j] > 4) return false;
  return true;
}

}  // namespace

}  // namespace ganet

using namespace ganet;

#define GA_TRY(call)            \
  do {                          \
    const int rc_ = (call);     \
    if (rc_) return rc_;        \
  } while (0)

extern "C" {

size_t ganet_decoder_saved_floats(int64_t M) { return M <= 0 ? 0 : (size_t)NL * M * H + (size_t)NL * 4 * H; }

size_t ganet_decoder_fwd_workspace(void) {
  return (3 * ganet_mlp_stats_floats(H) + (size_t)H * XP + (size_t)H * (XP + H)) * sizeof(float);
}

static int decoder_fwd_impl(int64_t M, const float* x, const GanetDecoderParams* p, float* saved, float* const* out,
                            void* workspace, hipStream_t stream) {
  const SavedView sv = view_saved(saved, M);
  float* col_part = static_cast<float*>(workspace);      // three sets: the conv6 branches' statistics in one launch
  float* w1p = col_part + 3 * ganet_mlp_stats_floats(H);
  float* w5p = w1p + (size_t)H * XP;
  hipLaunchKernelGGL(pad_weights_kernel, dim3(H), dim3(256), 0, stream, p->cin, p->W[0], p->W[4], w1p, w5p);
  GA_TRY(check_hip(hipGetLastError(), "pad_weights_kernel"));
  Sweep sweep;
  // layer i: z_i = [x1 | act(bn(z_src))] W^T + b with the column sums of its BatchNorm statistics (about the running
  // mean) into `cp`; the statistics kernel is a separate ~5 us launch that takes up to three layers at once
  auto layer = [&](int i, const float* x1, const float* W, int src, float* cp) -> int {
    const float* x2 = src >= 0 ? sv.z[src] : nullptr;
    const float* sc = src >= 0 ? sv.stat[src] + 2 * H : nullptr;
    const float* sh = src >= 0 ? sv.stat[src] + 3 * H : nullptr;
    return ganet_mlp_fwd(M, H, x1 ? XP : 0, x2 ? H : 0, x1, XP, x2, H, sc, sh, W, p->bias[i], sv.z[i], H, cp,
                         p->running_mean[i], sweep.next(), stream);
  };
  auto stats_job = [&](int i, const float* cp) -> FwdStatsJob {
    float* st = sv.stat[i];
    return FwdStatsJob{cp, p->gamma[i], p->beta[i], p->eps[i], st, st + H, st + 2 * H, st + 3 * H, p->running_mean[i],
                       p->running_var[i], p->momentum[i], reinterpret_cast<long long*>(p->num_batches_tracked[i]),
                       p->running_mean[i]};
  };
  auto hidden = [&](int i, const float* x1, const float* W, int src) -> int {
    GA_TRY(layer(i, x1, W, src, col_part));
    const FwdStatsJob job = stats_job(i, col_part);
    return mlp_stats_launch(1, &job, M, H, stream);
  };
  GA_TRY(hidden(0, x, w1p, -1));
  for (int i = 1; i <= 3; ++i) GA_TRY(hidden(i, nullptr, p->W[i], i - 1));
  GA_TRY(hidden(4, x, w5p, 3));
  // (the heads level by level with batched statistics launches, as the backward pass runs them, measured 0.4 % slower
  // here than head after head: 274.0 vs 275.0 it/s)
  // conv6 of the three heads: ONE launch (they share their input z5: the branch workgroups of a slab sit on the same
  // XCD, the slab leaves HBM once — ganet_layer_fwd.hip) and one statistics launch; shapes it does not take: head by head
  bool conv6_done = false;
  {
    const float* Ws[3] = {p->W[5], p->W[7], p->W[9]};
    const float* bs[3] = {p->bias[5], p->bias[7], p->bias[9]};
    float* zs3[3] = {sv.z[5], sv.z[7], sv.z[9]};
    float* cps[3] = {col_part, col_part + ganet_mlp_stats_floats(H), col_part + 2 * ganet_mlp_stats_floats(H)};
    const float* shifts[3] = {p->running_mean[5], p->running_mean[7], p->running_mean[9]};
    const int rc = layer_fwd_spec3(M, sv.z[4], sv.stat[4] + 2 * H, sv.stat[4] + 3 * H, Ws, bs, zs3, cps, shifts,
                                   sweep.next() == GANET_ROWS_DOWN ? 1 : 0, stream);
    if (rc > 0) return rc;
    if (rc == 0) {
      const FwdStatsJob jobs[3] = {stats_job(5, cps[0]), stats_job(7, cps[1]), stats_job(9, cps[2])};
      GA_TRY(mlp_stats_launch(3, jobs, M, H, stream));
      conv6_done = true;
    }
  }
  for (int j = 0; j < 3; ++j) {
    const int i6 = 5 + 2 * j, i7 = 6 + 2 * j;
    if (!conv6_done) GA_TRY(hidden(i6, nullptr, p->W[i6], 4));
    GA_TRY(hidden(i7, nullptr, p->W[i7], i6));
    GA_TRY(ganet_mlp_fwd(M, p->n8[j], 0, H, nullptr, 0, sv.z[i7], H, sv.stat[i7] + 2 * H, sv.stat[i7] + 3 * H,
                         p->W8[j], p->b8[j], out[j], p->n8[j], nullptr, nullptr, sweep.next(), stream));
  }
  return 0;
}

int ganet_decoder_fwd(int64_t M, const float* x, const GanetDecoderParams* p, float* saved, float* const* out,
                      void* workspace, size_t workspace_bytes, void* stream_) {
  if (M <= 0 || !x || !params_ok(p) || !saved || !out || !out[0] || !out[1] || !out[2] || !workspace) {
    set_error("ganet_decoder_fwd: invalid arguments");
    return 1;
  }
  if (workspace_bytes < ganet_decoder_fwd_workspace()) {
    set_error("ganet_decoder_fwd: workspace too small");
    return 2;
  }
  return decoder_fwd_impl(M, x, p, saved, out, workspace, static_cast<hipStream_t>(stream_));
}

static size_t bwd_wg_slot(int64_t M) {
  return align_up(ganet_wgrad_act_workspace(M, H, H) > ganet_mlp_bwd_fused_workspace()
                      ? ganet_wgrad_act_workspace(M, H, H) : ganet_mlp_bwd_fused_workspace(), 256);
}

size_t ganet_decoder_bwd_workspace(int64_t M) {
  if (M <= 0) return 0;
  const size_t wg = bwd_wg_slot(M);
  const int parts = ganet_mlp_bwd_fused_parts() > ganet_mlp_bwd_data_parts()
                        ? ganet_mlp_bwd_fused_parts() : ganet_mlp_bwd_data_parts();
  const int parts2 = parts > ganet_mlp_head_bwd_parts() ? parts : ganet_mlp_head_bwd_parts();
  size_t b = 0;
  b += (size_t)4 * M * H * sizeof(float);            // four rotating G buffers (the three heads run level by level)
  b += (size_t)GANET_MAX_WGRAD_JOBS * wg;            // partial tiles of every weight gradient
  b += 3 * align_up((size_t)parts2 * 256 * sizeof(float), 256);   // column-sum partials, one set per head
  b += (size_t)NL * 3 * H * sizeof(float);           // (A, q, p) per layer
  b += (size_t)3 * H * H * sizeof(float);            // dW0p, dWx (128 x 72 or the left part of 128 x 128), dWy
  b += 2 * H * sizeof(float);                        // discarded bias gradients of the split conv5 / conv1 halves
  return b;
}

static int decoder_bwd_impl(int64_t M, const float* x, const GanetDecoderParams* p, const float* saved_,
                            const float* const* d_out, const GanetDecoderGrads* g, void* workspace, void* stream_,
                            void* side_stream_) {
  for (int i = 0; i < NL; ++i)
    if (!g->dW[i] || !g->db[i] || !g->dgamma[i] || !g->dbeta[i]) { set_error("ganet_decoder_bwd: missing gradient buffer"); return 1; }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipStream_t side = static_cast<hipStream_t>(side_stream_);
  const SavedView sv = view_saved(const_cast<float*>(saved_), M);
  const int cin = p->cin;
  // ---- carve the workspace
  char* w = static_cast<char*>(workspace);
  float* Gbuf[4];
  for (int i = 0; i < 4; ++i) { Gbuf[i] = reinterpret_cast<float*>(w); w += (size_t)M * H * sizeof(float); }
  const size_t wg = bwd_wg_slot(M);
  char* wg_ws = w; w += (size_t)GANET_MAX_WGRAD_JOBS * wg;
  const int n_data = ganet_mlp_bwd_data_parts(), n_head = ganet_mlp_head_bwd_parts(), n_fused = ganet_mlp_bwd_fused_parts();
  const int parts = n_fused > n_data ? (n_fused > n_head ? n_fused : n_head) : (n_data > n_head ? n_data : n_head);
  float* col_parts[3];
  for (int j = 0; j < 3; ++j) { col_parts[j] = reinterpret_cast<float*>(w); w += align_up((size_t)parts * 256 * sizeof(float), 256); }
  float* col_part = col_parts[0];
  float* coef[NL];
  for (int i = 0; i < NL; ++i) { coef[i] = reinterpret_cast<float*>(w); w += 3 * H * sizeof(float); }
  float* dW0p = reinterpret_cast<float*>(w); w += (size_t)H * H * sizeof(float);
  float* dWx = reinterpret_cast<float*>(w); w += (size_t)H * H * sizeof(float);
  float* dWy = reinterpret_cast<float*>(w); w += (size_t)H * H * sizeof(float);
  float* db_dump = reinterpret_cast<float*>(w); w += 2 * H * sizeof(float);

  GanetWgradJob jobs[GANET_MAX_WGRAD_JOBS];
  int njobs = 0;
  Sweep sweep;
  // the weight-gradient launches that remain separate are off the dependency chain: side stream, ordered by events
  // (one pair per host thread and device: an event belongs to the device that was current when it was created)
  static thread_local hipEvent_t ev_main_dev[64] = {}, ev_side_dev[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  hipEvent_t& ev_main = ev_main_dev[dev_ & 63];
  hipEvent_t& ev_side = ev_side_dev[dev_ & 63];
  if (side && !ev_main) {
    GA_TRY(check_hip(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming), "hipEventCreate"));
    GA_TRY(check_hip(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming), "hipEventCreate"));
  }
  bool side_used = false;
  auto add_job = [&](int N, int K, float* dW, float* db, int nblocks) -> void* {
    void* ws = wg_ws + (size_t)njobs * wg;
    GanetWgradJob& j = jobs[njobs++];
    j.workspace = ws; j.M = M; j.N = N; j.K = K; j.dW = dW; j.db = db; j.nblocks = nblocks;
    return ws;
  };
  // dW [N,K], db [N]: g operand raw (gi < 0: `graw` [M,N]) or layer gi's (G, z, coef); x = act(bn(z_src)) or the input
  auto wgrad = [&](const float* graw, int N, int gi, const float* G, int src, int K, float* dW, float* db) -> int {
    void* ws = add_job(N, K, dW, db, 0);
    hipStream_t st = stream;
    if (side) {
      GA_TRY(check_hip(hipEventRecord(ev_main, stream), "hipEventRecord"));
      GA_TRY(check_hip(hipStreamWaitEvent(side, ev_main, 0), "hipStreamWaitEvent"));
      st = side;
      side_used = true;
    }
    const float* gt = gi < 0 ? graw : G;
    const float* gz = gi < 0 ? nullptr : sv.z[gi];
    const float* cf = gi < 0 ? nullptr : coef[gi];
    const float* xx = src < 0 ? x : sv.z[src];
    const float* sc = src < 0 ? nullptr : sv.stat[src] + 2 * H;
    const float* sh = src < 0 ? nullptr : sv.stat[src] + 3 * H;
    return ganet_wgrad_act(M, N, K, gt, N, gz, gz ? H : 0, cf, xx, src < 0 ? XP : H, sc, sh, nullptr, nullptr, ws, wg,
                           sweep.next(), st);
  };
  // column sums of (G_i, G_i z_i) -> (A, q, p), d gamma, d beta: a ~5 us launch that takes up to three layers
  auto bstats_job = [&](int i, const float* cp, int nparts) -> BwdStatsJob {
    return BwdStatsJob{cp, nparts, sv.stat[i], sv.stat[i] + H, sv.stat[i] + 2 * H, coef[i], g->dgamma[i], g->dbeta[i]};
  };
  auto finish = [&](int i, int nparts) -> int {
    const BwdStatsJob job = bstats_job(i, col_part, nparts);
    return bwd_stats_launch(1, &job, M, stream);
  };
  // hidden layer i with a 128-column activated input z_src: one pass (data + weight gradient)
  auto layer_bwd = [&](int i, const float* G, int src, const float* W, int64_t ldw, float* out, int accumulate, int act,
                       float* dW, float* db, float* cp) -> int {
    void* ws = add_job(H, H, dW, db, n_fused);
    return ganet_mlp_bwd_fused(M, G, sv.z[i], coef[i], W, ldw, out, accumulate, sv.z[src], sv.stat[src] + 2 * H,
                               sv.stat[src] + 3 * H, act, cp, ws, wg, sweep.next(), stream);
  };

  // all three heads carry a gradient in the training loop (a caller with an unused head takes the per-layer path)
  if (!d_out[0] || !d_out[1] || !d_out[2] || !g->dW8[0] || !g->dW8[1] || !g->dW8[2] || !g->db8[0] || !g->db8[1] || !g->db8[2]) {
    set_error("ganet_decoder_bwd: the gradients of all three heads are required");
    return 1;
  }
  // The three heads level by level (their chains are independent until they meet in G_5): one statistics launch per
  // level instead of three, and the kernels of a level read the same tensors back to back. G buffers: G7_j = buffer j;
  // G6_0 = buffer 3, G6_1 = buffer 0 (G7_0 is dead by then), G6_2 = buffer 1; G5 = buffer 2.
  BwdStatsJob bjobs[3];
  for (int j = 0; j < 3; ++j) {
    const int i7 = 6 + 2 * j, N8 = p->n8[j];
    // the head's own weight gradient rides on head_bwd (same g and z rows, the activation shares its exponential
    // with softplus'): its per-workgroup partials join the batched reduction. (Round 1 measured this slower — head_bwd
    // turned VALU-bound beside fp32-MFMA kernels; with the separate pass at 43-55 us per head it now wins: +3 % it/s.)
    float* hw = static_cast<float*>(add_job(N8, H, g->dW8[j], g->db8[j], n_head));
    GA_TRY(ganet_mlp_head_bwd(M, N8, d_out[j], p->W8[j], sv.z[i7], H, sv.stat[i7] + 2 * H, sv.stat[i7] + 3 * H, Gbuf[j], H,
                              col_parts[j], hw, stream));
    bjobs[j] = bstats_job(i7, col_parts[j], n_head);
  }
  GA_TRY(bwd_stats_launch(3, bjobs, M, stream));
  float* const G6[3] = {Gbuf[3], Gbuf[0], Gbuf[1]};
  for (int j = 0; j < 3; ++j) {
    const int i6 = 5 + 2 * j, i7 = 6 + 2 * j;
    GA_TRY(layer_bwd(i7, Gbuf[j], i6, p->W[i7], H, G6[j], 0, 1, g->dW[i7], g->db[i7], col_parts[j]));
    bjobs[j] = bstats_job(i6, col_parts[j], n_fused);
  }
  GA_TRY(bwd_stats_launch(3, bjobs, M, stream));
  float* G5 = Gbuf[2];
  for (int j = 0; j < 3; ++j) {
    const int i6 = 5 + 2 * j;
    GA_TRY(layer_bwd(i6, G6[j], 4, p->W[i6], H, G5, j > 0, j == 2 ? 1 : 0, g->dW[i6], g->db[i6], col_part));
  }
  const int nparts5 = n_fused;
  GA_TRY(finish(4, nparts5));
  const float* W5 = p->W[4];
  const int64_t ld5 = cin + H;
  // the layers fed by the raw input (conv5's input half, conv1): weight gradient + input gradient in one pass when the
  // caller's input (and its gradient) is the zero-padded [M,72] layout, else the two separate kernels
  const bool input_one_pass = g->dx && g->x_cols == XP;
  const int ldp = input_one_pass ? H : XP;
  auto input_bwd = [&](int i, const float* G, const float* W, int64_t ldw, int accumulate, float* dWp, float* db) -> int {
    if (input_one_pass) {
      void* ws = add_job(H, H, dWp, db, n_fused);
      return ganet_mlp_bwd_fused_input(M, G, sv.z[i], coef[i], W, ldw, cin, g->dx, XP, accumulate, x, ws, wg,
                                       sweep.next(), stream);
    }
    GA_TRY(wgrad(nullptr, H, i, G, -1, XP, dWp, db));
    if (g->dx) GA_TRY(ganet_mlp_bwd_data(M, cin, G, H, sv.z[i], H, coef[i], W, ldw, g->dx, g->x_cols, accumulate, nullptr,
                                          0, nullptr, nullptr, nullptr, sweep.next(), stream));
    return 0;
  };
  GA_TRY(input_bwd(4, G5, W5, ld5, 0, dWx, db_dump));
  float* Gcur = Gbuf[0];
  GA_TRY(layer_bwd(4, G5, 3, W5 + cin, ld5, Gcur, 0, 1, dWy, g->db[4], col_part));
  float* Gnext = Gbuf[1];
  for (int i = 3; i >= 1; --i) {
    GA_TRY(finish(i, n_fused));
    GA_TRY(layer_bwd(i, Gcur, i - 1, p->W[i], H, Gnext, 0, 1, g->dW[i], g->db[i], col_part));
    float* t = Gcur; Gcur = Gnext; Gnext = t;
  }
  GA_TRY(finish(0, n_fused));
  GA_TRY(input_bwd(0, Gcur, p->W[0], cin, 1, dW0p, g->db[0]));
  if (side_used) {
    GA_TRY(check_hip(hipEventRecord(ev_side, side), "hipEventRecord"));
    GA_TRY(check_hip(hipStreamWaitEvent(stream, ev_side, 0), "hipStreamWaitEvent"));
  }
  GA_TRY(ganet_wgrad_reduce_batch(njobs, jobs, stream));
  hipLaunchKernelGGL(assemble_wgrads_kernel, dim3(H), dim3(256), 0, stream, cin, ldp, dW0p, dWx, dWy, g->dW[0], g->dW[4]);
  return check_hip(hipGetLastError(), "assemble_wgrads_kernel");
}

int ganet_decoder_bwd(int64_t M, const float* x, const GanetDecoderParams* p, const float* saved_,
                      const float* const* d_out, const GanetDecoderGrads* g, void* workspace, size_t workspace_bytes,
                      void* stream_, void* side_stream_) {
  if (M <= 0 || (M % 32) || !x || !params_ok(p) || !saved_ || !d_out || !g || !workspace) {
    set_error("ganet_decoder_bwd: invalid arguments (M must be a multiple of 32)");
    return 1;
  }
  if (workspace_bytes < ganet_decoder_bwd_workspace(M)) {
    set_error("ganet_decoder_bwd: workspace too small");
    return 2;
  }
  return decoder_bwd_impl(M, x, p, saved_, d_out, g, workspace, stream_, side_stream_);
}

}  // extern "C"
