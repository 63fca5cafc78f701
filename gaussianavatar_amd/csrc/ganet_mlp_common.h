// ganet_mlp_common.h — helpers shared by the fused decoder-layer kernels (internal).
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace ganet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// llvm.amdgcn.sched.barrier mask: VALU | SALU | DS | transcendental may cross; MFMA and VMEM may not
constexpr int kSchedMask = 0x2 | 0x4 | 0x80 | 0x100 | 0x200 | 0x400;

// fp32-input MFMA runs at the vector-ALU rate on gfx950 and does NOT overlap with other VALU work of
// the SIMD (measured: tools/ubench/mfma_lds — 4 softplus per 16 MFMAs cost 20 % of the MFMA rate), so
// the activation math next to the MFMAs is written for minimum instruction count:
//   softplus(u) = max(u, 0) + log1p(exp(-|u|))     (no threshold / small-argument branches needed:
//   t = exp(-|u|) is in (0, 1], so 1 + t never overflows and the absolute error is <= 1 ulp of 1)
// in log2 units: kLog2e is folded into the argument by the caller where it can be, the final ln 2 too.
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// log2-unit softplus: returns softplus(u) / ln 2 for u2 = u * log2(e).   5 instructions (2 transcendental)
__device__ __forceinline__ float softplus_log2(float u2) {
  const float t = __builtin_amdgcn_exp2f(-__builtin_fabsf(u2));
  return __builtin_fmaxf(u2, 0.0f) + __builtin_amdgcn_logf(1.0f + t);
}

// softplus(u) (torch.nn.Softplus: beta 1; its threshold 20 is reproduced exactly by rounding)
__device__ __forceinline__ float softplus_f(float u) { return kLn2 * softplus_log2(u * kLog2e); }

// d softplus(u) / du = sigmoid(u) for u2 = u * log2(e)                      3 instructions
__device__ __forceinline__ float sigmoid_log2(float u2) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u2));
}
__device__ __forceinline__ float softplus_grad_f(float u) { return sigmoid_log2(u * kLog2e); }

// Sum of the per-workgroup partials of ONE column (and of its companion sum `second` floats further on) by ONE wave, in
// double, fixed order: every lane adds the parts lane, lane + 64, ... (independent loads, one round trip), then a
// butterfly over the lanes. No LDS, no workgroup barrier: the statistics kernels are launch-latency bound (5.3 -> ~4 us).
__device__ __forceinline__ void column_sums_wave(const float* __restrict__ col_part, int nparts, int stride, int second,
                                                 int n, double& s_out, double& q_out) {
  const int lane = threadIdx.x & 63;
  double s = 0.0, q = 0.0;
  for (int p = lane; p < nparts; p += 64) {
    s += (double)col_part[(size_t)p * stride + n];
    q += (double)col_part[(size_t)p * stride + second + n];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  s_out = s;
  q_out = q;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace ganet
