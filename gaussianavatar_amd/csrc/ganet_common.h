// ganet_common.h — shared helpers of the ganet_* translation units (internal).
#pragma once
#include <atomic>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace ganet {

// "done once per device" flag for per-function attributes (hipFuncSetAttribute is a per-device setting; a process may
// drive several GPUs): bit d = done on device d. Used as `static PerDeviceFlag f; if (!f) { ...; f = true; }`.
struct PerDeviceFlag {
  std::atomic<uint64_t> mask{0};
  static uint64_t bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
  bool operator!() const { return !(mask.load(std::memory_order_acquire) & bit()); }
  PerDeviceFlag& operator=(bool v) { if (v) mask.fetch_or(bit(), std::memory_order_release); return *this; }
};
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// Opt-in per-kernel timing (ganet_profile_* in ganet.h): every instrumented launch is bracketed by
// hipEvents recorded on the launch stream.
enum KernelId { K_MLP_FWD = 0, K_MLP_STATS, K_WGRAD, K_WGRAD_REDUCE, K_BWD_DATA, K_HEAD_BWD,
                K_BWD_STATS, K_SSIM_FWD, K_SSIM_BWD, K_LAYER_BWD, K_ROWGEMM, K_UPZ_FWD, K_DZ_UPT, K_COUNT };
struct ProfScope {
  ProfScope(KernelId id, hipStream_t stream);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

// ganet_mlp_split.hip / ganet_wgrad_split.hip: return -1 when the shape has no split kernel
// BatchNorm statistics of up to three layers in ONE launch (the decoder's three heads run level by level:
// ganet_decoder.hip). Jobs are independent; a launch costs ~5 us whatever it does.
constexpr int kMaxStatsJobs = 3;
struct FwdStatsJob {
  const float* col_part; const float* gamma; const float* beta; float eps;
  float *mean, *rstd, *scale, *shift, *running_mean, *running_var;
  float momentum; long long* num_batches_tracked; const float* stat_shift;
};
struct BwdStatsJob {
  const float* col_part; int nparts; const float *mean, *rstd, *scale; float *coef, *dgamma, *dbeta;
};
// The separable texel grid of the bilinear up-sampling (include/ganet.h: GanetUpGrid, forward tap lists) as the kernels
// take it by value.
struct UpGrid {
  int frames, S, R;
  const int32_t* row_idx; const float* row_w;      // [S,2]
  const int32_t* col_idx; const float* col_w;      // [S,2]
  const float* uv; int64_t uv_frame_stride;        // [frames or 1][S*S][2]
};
// Additive term of a layer's output (ganet_upz.hip): bilinear(P)[m, 0:128] + Wuv . uv[m] — the skip layer's input half.
struct FwdAddend { UpGrid g; const float* P; int64_t ldp; const float* Wuv; };
int mlp_stats_launch(int njobs, const FwdStatsJob* jobs, int64_t M, int N, hipStream_t stream);
int bwd_stats_launch(int njobs, const BwdStatsJob* jobs, int64_t M, hipStream_t stream);
int layer_fwd_spec(int64_t M, const float* x, int64_t ldx, const float* in_scale, const float* in_shift, const float* W,
                   const float* bias, float* z, int64_t ldz, float* col_part, const float* stat_shift, int reverse,
                   hipStream_t stream);
// the same with `add` gathered into the output tile before the statistics (S a multiple of 32: a slab = one texel row)
int layer_fwd_spec_add(int64_t M, const float* x, const float* in_scale, const float* in_shift, const float* W,
                       const float* bias, float* z, float* col_part, const float* stat_shift, const FwdAddend& add,
                       int reverse, hipStream_t stream);
int layer_fwd_spec3(int64_t M, const float* x, const float* in_scale, const float* in_shift, const float* const* W,
                    const float* const* bias, float* const* z, float* const* col_part, const float* const* stat_shift,
                    int reverse, hipStream_t stream);
int mlp_fwd_split(int64_t M, int N, int K1, int K2, const float* x1, int64_t ld1, const float* x2, int64_t ld2,
                  const float* in_scale, const float* in_shift, const float* W, const float* bias, float* z,
                  int64_t ldz, float* col_part, const float* stat_shift, int reverse, hipStream_t stream);
int mlp_bwd_split(int64_t M, int O, const float* g, int64_t ldg, const float* gz, int64_t ldgz,
                  const float* gcoef, const float* W, int64_t ldw, float* out, int64_t ldo, bool accumulate,
                  const float* src_z, int64_t ld_src, const float* src_scale, const float* src_shift,
                  float* col_part, int reverse, hipStream_t stream);
int wgrad_split(int64_t M, int N, int K, const float* g, int64_t ldg, const float* gz, int64_t ldgz,
                const float* gcoef, const float* x, int64_t ldx, const float* in_scale, const float* in_shift,
                float* partial, int blocks, int order, hipStream_t stream);
// ganet_upz.hip
}  // namespace ganet
struct GanetUpGrid;
namespace ganet {
UpGrid up_grid_of(const ::GanetUpGrid* t);
int rowgemm_launch(int64_t M, int N, int K, const float* A, int64_t lda, const float* Bt, int64_t ldb, float* C,
                   int64_t ldc, int accumulate, hipStream_t stream);
int upsample_z_fwd_launch(const UpGrid& g, const float* P, int64_t ldp, const float* Wuv, const float* bias,
                          const float* stat_shift, float* z, float* col_part, hipStream_t stream);
int dz_upsample_t_blocks(const UpGrid& g);
int dz_upsample_t_launch(const UpGrid& g, const ::GanetUpGrid& t, const float* G, const float* Z, const float* coef,
                         float* dP, int64_t ldp, float* partial, hipStream_t stream);
}  // namespace ganet
