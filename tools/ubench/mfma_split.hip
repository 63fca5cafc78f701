// Does VALU work overlap with bf16 MFMAs on a SIMD? (dev tool) Variants of one k-step of the split-bf16 decoder loop:
// 24 MFMAs (4 accumulators x 6) and the VALU that produces the next step's fragments. Prints cycles per step per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned f2u(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float u2f(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// MODE bits: 1 = MFMAs, 2 = split VALU (44 / 8 values), 4 = softplus (8 x 6), 8 = sched_group interleave,
//            16 = B fragments from LDS (12 ds_read_b128 per step), 32 = plain independent v_fma filler (100) instead
template <int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, int steps, float a0) {
  extern __shared__ u32x4 s_w[];
  for (int i = threadIdx.x; i < 3 * 128 * 16; i += 64 * WAVES) s_w[i] = u32x4{(unsigned)i * 2654435761u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 + lane * 0.01f + i;
  u32x4 a1 = {f2u(v[0]), f2u(v[1]), f2u(v[2]), f2u(v[3])}, a2 = a1, a3 = a1;
  u32x4 b1 = a1, b2 = a1, b3 = a1;
  float fill[10];
  for (int i = 0; i < 10; ++i) fill[i] = a0 * i;
  int woff = lane & 31;
  for (int s = 0; s < steps; ++s) {
    asm volatile("" : "+v"(woff));
    u32x4 n1 = a1, n2 = a2, n3 = a3;
    if (MODE & 6) {
      float w[8];
      for (int i = 0; i < 8; ++i) {
        float x = v[i] + 0.001f * s;
        if (MODE & 4) {
          const float u = fmaf(x, 1.01f, 0.3f);
          x = __builtin_fmaxf(u, 0.f) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-__builtin_fabsf(u)));
        }
        w[i] = x;
      }
      if (MODE & 2) {
        float r[8], q[8];
        for (int i = 0; i < 8; ++i) r[i] = w[i] - u2f(f2u(w[i]) & 0xffff0000u);
        for (int i = 0; i < 8; ++i) q[i] = r[i] - u2f(f2u(r[i]) & 0xffff0000u);
        for (int j = 0; j < 4; ++j) {
          n1[j] = __builtin_amdgcn_perm(f2u(w[2 * j + 1]), f2u(w[2 * j]), 0x07060302u);
          n2[j] = __builtin_amdgcn_perm(f2u(r[2 * j + 1]), f2u(r[2 * j]), 0x07060302u);
          n3[j] = __builtin_amdgcn_perm(f2u(q[2 * j + 1]), f2u(q[2 * j]), 0x07060302u);
        }
      } else {
        n1 = u32x4{f2u(w[0]), f2u(w[1]), f2u(w[2]), f2u(w[3])}; n2 = u32x4{f2u(w[4]), f2u(w[5]), f2u(w[6]), f2u(w[7])};
      }
    }
    if (MODE & 32) {
#pragma unroll
      for (int j = 0; j < 10; ++j)
#pragma unroll
        for (int i = 0; i < 10; ++i) fill[i] = fmaf(fill[i], 1.0001f, 0.5f);
    }
    if ((MODE & 1) && (MODE & 64)) {
      u32x4 bb1[4], bb2[4], bb3[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bb1[t] = b1; bb2[t] = b2; bb3[t] = b3;
        if (MODE & 16) {
          const int at = (woff + t * 32) * 16 + ((2 * (s & 7) + (lane >> 5)) ^ (woff & 15));
          bb1[t] = s_w[at]; bb2[t] = s_w[128 * 16 + at]; bb3[t] = s_w[2 * 128 * 16 + at];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mf(a3, bb1[t], acc[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mf(a2, bb2[t], acc[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mf(a1, bb3[t], acc[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mf(a2, bb1[t], acc[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mf(a1, bb2[t], acc[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mf(a1, bb1[t], acc[t]);
    } else if (MODE & 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (MODE & 16) {
          const int at = (woff + t * 32) * 16 + ((2 * (s & 7) + (lane >> 5)) ^ (woff & 15));
          b1 = s_w[at]; b2 = s_w[128 * 16 + at]; b3 = s_w[2 * 128 * 16 + at];
        }
        acc[t] = mf(a3, b1, acc[t]); acc[t] = mf(a2, b2, acc[t]); acc[t] = mf(a1, b3, acc[t]);
        acc[t] = mf(a2, b1, acc[t]); acc[t] = mf(a1, b2, acc[t]); acc[t] = mf(a1, b1, acc[t]);
      }
    }
    if (MODE & 8) {
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, 5, 0);
      }
    }
    a1 = n1; a2 = n2; a3 = n3;
  }
  float sum = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
  for (int i = 0; i < 10; ++i) sum += fill[i];
  sum += u2f(a1[0] ^ a2[1] ^ a3[2] ^ a1[3] ^ a2[0] ^ a3[1] ^ a1[2] ^ a2[3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <int MODE, int WAVES>
void run(const char* name, int steps = 4096) {
  const int blocks = 256;
  float* out; hipMalloc(&out, blocks * 64 * WAVES * 4);
  const size_t lds = 3 * 128 * 16 * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, 0, out, steps, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  // ns per step per wave; cycles at 2.4 GHz per step per SIMD (WAVES / 4 waves share a SIMD)
  const double ns = best * 1e6 / steps;
  printf("%-44s waves/SIMD %d  %8.1f ns/step/wave  = %7.0f cyc@2.4GHz per step per SIMD-wave-slot\n", name, WAVES / 4, ns, ns * 2.4 / (WAVES / 4));
  hipFree(out);
}
int main() {
  run<1, 4>("mfma only (tile-major)");
  run<1, 8>("mfma only (tile-major)");
  run<1 | 64, 4>("mfma only, product-major");
  run<1 | 64, 8>("mfma only, product-major");
  run<3 | 64, 8>("mfma + split, product-major");
  run<7 | 64, 8>("mfma + softplus + split, product-major");
  run<7 | 64 | 8, 8>("mfma + softplus + split, product-major, sched_group");
  run<7 | 64 | 16, 8>("mfma + softplus + split + LDS B, product-major");
  run<7 | 64 | 16 | 8, 8>("mfma + softplus + split + LDS B, product-major, sched_group");
  run<7 | 64 | 16, 4>("mfma + softplus + split + LDS B, product-major");
  run<7 | 16, 8>("mfma + softplus + split + LDS B (tile-major)");
  return 0;
}
