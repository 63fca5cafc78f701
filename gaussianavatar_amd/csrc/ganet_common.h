// ganet_common.h — shared helpers of the ganet_* translation units (internal).
#pragma once
#include <hip/hip_runtime.h>

namespace ganet {
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
}  // namespace ganet
