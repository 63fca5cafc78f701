// Streaming-read bandwidth of the decoder kernels' A-operand pattern vs a fully coalesced one (dev tool).
//   MODE 0: MFMA fragment pattern: lane (row = lane & 31, kg = lane >> 5) reads 2 x 16 B at [row][16 s + 8 kg] of a
//           32-row x 128-float slab, 8 steps per slab (each instruction touches 32 rows)
//   MODE 1: coalesced: instruction j of a slab reads bytes [1024 j, 1024 j + 1024) of the slab (16 B per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int WAVES, int D>
__global__ void __launch_bounds__(64 * WAVES) k(const float* __restrict__ x, float* out, long M) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long nslab = M / 32;
  const long wg = (long)blockIdx.x * WAVES + wave, stride = (long)gridDim.x * WAVES;
  float4 ring[D][2];
  float acc = 0.f;
  auto ptr = [&](long slab, int s) -> const float* {
    const float* base = x + slab * 32 * 128;
    if (MODE == 0) return base + (lane & 31) * 128 + 16 * s + 8 * (lane >> 5);
    return base + (2 * s) * 256 + lane * 4;     // two consecutive 1 KB pieces per step
  };
  auto load = [&](long slab, int s, float4 (&r)[2]) {
    const float* p = ptr(slab < nslab ? slab : nslab - 1, s);
    r[0] = *reinterpret_cast<const float4*>(p);
    r[1] = *reinterpret_cast<const float4*>(p + (MODE == 0 ? 4 : 256));
  };
  for (int s = 0; s < D; ++s) load(wg, s, ring[s]);
  for (long slab = wg; slab < nslab; slab += stride) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float4 a = ring[s % D][0], b = ring[s % D][1];
      acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
      __builtin_amdgcn_sched_barrier(0);
      if (s + D < 8) load(slab, s + D, ring[s % D]); else load(slab + stride, s + D - 8, ring[s % D]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = acc;
}
template <int MODE, int WAVES, int D>
void run(const char* name, int blocks, const float* x, long M) {
  float* out; hipMalloc(&out, blocks * 64 * WAVES * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WAVES, D>), dim3(blocks), dim3(64 * WAVES), 0, 0, x, out, M);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-40s blocks %4d waves/CU %2d ring %d: %7.1f us  %6.2f TB/s\n", name, blocks, WAVES * (blocks / 256), D, best * 1e3, M * 512.0 / (best * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  const long M = 262144 * 4;    // 537 MB: larger than the Infinity Cache
  float* x; hipMalloc(&x, M * 512); hipMemset(x, 0, M * 512);
  run<0, 8, 4>("fragment pattern", 256, x, M);
  run<0, 8, 8>("fragment pattern", 256, x, M);
  run<1, 8, 4>("coalesced", 256, x, M);
  run<1, 8, 8>("coalesced", 256, x, M);
  run<0, 8, 4>("fragment pattern", 512, x, M);
  run<1, 8, 4>("coalesced", 512, x, M);
  run<0, 4, 4>("fragment pattern", 1024, x, M);
  run<1, 4, 4>("coalesced", 1024, x, M);
  run<0, 4, 4>("fragment pattern", 2048, x, M);
  run<1, 4, 4>("coalesced", 2048, x, M);
  return 0;
}
