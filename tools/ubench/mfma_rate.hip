// Measures the sustained issue rate of v_mfma_f32_32x32x2_f32 (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int threads, const char* name) {
  float* out; hipMalloc(&out, blocks * threads * 4);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * (threads / 64) * iters * 4 * NACC;
    if (rep == 2)
      printf("%-28s %8.3f ms  %7.1f TFLOP/s   (%.1f ns per MFMA per wave)\n", name, ms, mfma * 4096 / ms / 1e9,
             ms * 1e6 / (iters * 4.0 * NACC));
  }
  hipFree(out);
}
int main() {
  run<4>(256, 256, "4 acc, 1 wave/SIMD");
  run<4>(512, 256, "4 acc, 2 waves/SIMD");
  run<1>(256, 256, "1 acc (dependent), 1 w/SIMD");
  run<2>(256, 256, "2 acc, 1 wave/SIMD");
  run<4>(1024, 256, "4 acc, 4 waves/SIMD");
  return 0;
}
