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
                K_BWD_STATS, K_SSIM_FWD, K_SSIM_BWD, K_LAYER_BWD, K_COUNT };
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
int mlp_stats_launch(int njobs, const FwdStatsJob* jobs, int64_t M, int N, hipStream_t stream);
int bwd_stats_launch(int njobs, const BwdStatsJob* jobs, int64_t M, hipStream_t stream);
int layer_fwd_spec(int64_t M, const float* x, int64_t ldx, const float* in_scale, const float* in_shift, const float* W,
                   const float* bias, float* z, int64_t ldz, float* col_part, const float* stat_shift, int reverse,
                   hipStream_t stream);
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
}  // namespace ganet
