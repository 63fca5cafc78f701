// MFMA issue rate with the B operand streamed from LDS the way ganet_mlp's main loop does (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDW4 = 33;
template <int MODE>   // 0: LDS reads feed MFMA; 1: + opaque offset per slab; 2: a operand via v_fma chain
__global__ void __launch_bounds__(512) k(float* out, int slabs, float a0) {
  extern __shared__ float4 s_w[];
  for (int i = threadIdx.x; i < 128 * LDW4; i += 512) s_w[i] = make_float4(i * 1e-6f, 1.f, 2.f, 3.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5, col = lane & 31;
  f32x16 acc[4];
  float s = 0.f;
  float a = a0 + lane;
  for (int sl = 0; sl < slabs; ++sl) {
    int woff = col * LDW4 + h;
    asm volatile("" : "+v"(woff));
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      float4 bw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bw[t] = s_w[woff + t * 32 * LDW4 + 2 * b];
      float av0 = a, av1 = a, av2 = a, av3 = a;
      if (MODE == 2) { av0 = fmaf(a, 1.01f, b); av1 = fmaf(a, 1.02f, b); av2 = fmaf(a, 1.03f, b); av3 = fmaf(a, 1.04f, b); }
      if (MODE == 3) {
        auto sp = [](float u) {
          const float e = __builtin_amdgcn_exp2f(u * 1.4426950408889634f);
          const float lg = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
          const float ser = e * (1.0f - 0.5f * e);
          const float r = e < 1e-3f ? ser : lg;
          return u > 20.0f ? u : r;
        };
        const float x0 = a + 0.01f * (b + sl);
        av0 = sp(fmaf(x0, 1.01f, 0.1f)); av1 = sp(fmaf(x0, 1.02f, 0.2f)); av2 = sp(fmaf(x0, 1.03f, 0.3f)); av3 = sp(fmaf(x0, 1.04f, 0.4f));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bw[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bw[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av2, bw[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av3, bw[t].w, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(int blocks, const char* name, int slabs = 64) {
  float* out; hipMalloc(&out, blocks * 512 * 4);
  const size_t lds = 128 * LDW4 * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), lds, 0, out, slabs, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 8 * slabs * 256;
    if (rep == 2) printf("%-40s slabs=%3d %8.1f us  %7.1f TFLOP/s\n", name, slabs, ms * 1e3, mfma * 4096 / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  for (int sl : {4, 64}) run<0>(256, "LDS-fed B, 2 waves/SIMD", sl);
  for (int sl : {4, 64}) run<3>(256, "LDS-fed B + softplus A", sl);
  return 0;
}
