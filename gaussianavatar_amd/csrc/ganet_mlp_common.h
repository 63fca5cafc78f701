// ganet_mlp_common.h — helpers shared by the fused decoder-layer kernels (internal).
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace ganet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// llvm.amdgcn.sched.barrier mask: VALU | SALU | DS | transcendental may cross; MFMA and VMEM may not
constexpr int kSchedMask = 0x2 | 0x4 | 0x80 | 0x100 | 0x200 | 0x400;

// softplus(u) = log1p(exp(u)) (torch.nn.Softplus: beta 1, threshold 20); the series keeps full
// relative precision where exp(u) vanishes against the 1 in 1 + e. Straight-line code on the
// hardware exp2/log2 (no branches: this runs between MFMAs).
__device__ __forceinline__ float softplus_f(float u) {
  const float e = __builtin_amdgcn_exp2f(u * 1.4426950408889634f);
  const float lg = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
  const float ser = e * (1.0f - 0.5f * e);
  const float sp = e < 1e-3f ? ser : lg;
  return u > 20.0f ? u : sp;
}

// d softplus(u) / du = sigmoid(u) (1 where torch's threshold makes softplus the identity)
__device__ __forceinline__ float softplus_grad_f(float u) {
  const float e = __builtin_amdgcn_exp2f(-u * 1.4426950408889634f);
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  return u > 20.0f ? 1.0f : s;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace ganet
