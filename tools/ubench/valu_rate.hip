// valu_rate.hip — issue rate of fp32 VALU forms on gfx950: v_fma_f32, v_pk_fma_f32 (VGPR operands), v_pk_fma_f32 with an
// SGPR-pair operand, and v_pk_mul_f32, with 1 / 2 / 4 waves per SIMD and 8 independent accumulator chains per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate && tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s0) {
  f32x2 a[8];
  const float l = (float)threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f32x2{l + i, l - i};
  f32x2 b = f32x2{1.0001f + l * 1e-9f, 0.9999f};
  f32x2 sp = f32x2{s0, s0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) {        // two scalar FMAs
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(b.y));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(b.x), "v"(b.y));
        } else if (MODE == 1) { // packed, VGPR operands
          asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        } else if (MODE == 2) { // packed, SGPR pair operand
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sp), "v"(b));
        } else {                // packed multiply
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
  if (r == 123.456f) out[0] = r;
}

template <int MODE>
int run(const char* name, float* out, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_wave = (double)iters * 64 * (MODE == 0 ? 2 : 1);
  const double waves_per_simd = blocks / 256.0;
  printf("%-28s %4.0f wave/SIMD  %8.1f us  %6.2f ns per instr per SIMD  (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd,
         ms * 1e3, ms * 1e6 / (instr_per_wave * waves_per_simd), ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4);
  return 0;
}

int main() {
  float* out; CK(hipMalloc(&out, 64));
  for (int blocks : {256, 512, 1024}) {
    if (run<0>("v_fma_f32 x2", out, blocks)) return 1;
    if (run<1>("v_pk_fma_f32 vgpr", out, blocks)) return 1;
    if (run<2>("v_pk_fma_f32 sgpr pair", out, blocks)) return 1;
    if (run<3>("v_pk_mul_f32", out, blocks)) return 1;
  }
  return 0;
}
