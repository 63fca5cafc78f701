// ganet_common.h — shared helpers of the ganet_* translation units (internal).
#pragma once
#include <hip/hip_runtime.h>

namespace ganet {
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// Opt-in per-kernel timing (ganet_profile_* in ganet.h): every instrumented launch is bracketed by
// hipEvents recorded on the launch stream.
enum KernelId { K_MLP_FWD = 0, K_MLP_STATS, K_WGRAD, K_WGRAD_REDUCE, K_BWD_DATA, K_HEAD_BWD,
                K_BWD_STATS, K_SSIM_FWD, K_SSIM_BWD, K_COUNT };
struct ProfScope {
  ProfScope(KernelId id, hipStream_t stream);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};
}  // namespace ganet
