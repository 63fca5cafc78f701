// stream4.hip — what the memory system gives the one-pass layer backward's traffic pattern: 3 [M,128] fp32 tensors read,
// 1 written (134 MB each at M = 262,144), 256 workgroups of 512 threads, every workgroup one 32-row slab (16 KiB per tensor)
// per round. Variants of the slab -> workgroup mapping and of the number of 16-byte loads a lane keeps in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream4.hip -o tools/ubench/stream4 && tools/ubench/stream4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MAP, int PF>
__global__ void __launch_bounds__(512) k(const float4* a, const float4* b, const float4* c, float4* o, long nslab) {
  const int rounds = (int)(nslab / gridDim.x);
  const int t = threadIdx.x;                       // 512 threads x 2 float4 = 16 KiB per tensor and slab
  auto slab_of = [&](int r) -> long {
    if (MAP == 0) return (long)r * gridDim.x + blockIdx.x;                       // common front
    if (MAP == 1) return (long)blockIdx.x * rounds + r;                          // one contiguous range per workgroup
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;     // XCD x streams its own eighth
    return (long)x * (nslab >> 3) + (long)r * per + j;
  };
  float4 va[PF][2], vb[PF][2], vc[PF][2];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    const long s = slab_of(p) * 1024;
#pragma unroll
    for (int h = 0; h < 2; ++h) { va[p][h] = a[s + t + 512 * h]; vb[p][h] = b[s + t + 512 * h]; vc[p][h] = c[s + t + 512 * h]; }
  }
  for (int r = 0; r < rounds; r += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const long s = slab_of(r + p) * 1024;
      float4 w[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) w[h] = make_float4(va[p][h].x + vb[p][h].y * vc[p][h].z, va[p][h].y, vb[p][h].z, vc[p][h].w);
      const long sn = slab_of(r + p + PF < rounds ? r + p + PF : r + p) * 1024;
#pragma unroll
      for (int h = 0; h < 2; ++h) { va[p][h] = a[sn + t + 512 * h]; vb[p][h] = b[sn + t + 512 * h]; vc[p][h] = c[sn + t + 512 * h]; }
#pragma unroll
      for (int h = 0; h < 2; ++h) o[s + t + 512 * h] = w[h];
    }
  }
}

template <int MAP, int PF>
int run(const char* name, float4* a, float4* b, float4* c, float4* o, long M, size_t pad) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long nslab = M / 32;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MAP, PF>), dim3(256), dim3(512), 0, 0, a, b, c, o, nslab);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MAP, PF>), dim3(256), dim3(512), 0, 0, a, b, c, o, nslab);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = 4.0 * M * 128 * 4;
  printf("%-44s pad %6zu B  %7.1f us  %6.2f TB/s\n", name, pad, us, bytes / us / 1e6);
  return 0;
}

int main() {
  const long M = 262144;
  const size_t n = (size_t)M * 128 * 4;
  for (size_t pad : {(size_t)0, (size_t)4096 + 256, (size_t)(1 << 20) + 8192}) {
    char* base; CK(hipMalloc(&base, 4 * (n + pad) + 4096));
    CK(hipMemset(base, 0, 4 * (n + pad)));
    float4 *a = (float4*)base, *b = (float4*)(base + n + pad), *c = (float4*)(base + 2 * (n + pad)), *o = (float4*)(base + 3 * (n + pad));
    if (run<0, 1>("common front, 1 slab in flight", a, b, c, o, M, pad)) return 1;
    if (run<0, 2>("common front, 2 slabs in flight", a, b, c, o, M, pad)) return 1;
    if (run<1, 2>("contiguous range per workgroup, 2 slabs", a, b, c, o, M, pad)) return 1;
    if (run<2, 2>("one eighth per XCD, 2 slabs", a, b, c, o, M, pad)) return 1;
    CK(hipFree(base));
  }
  return 0;
}
