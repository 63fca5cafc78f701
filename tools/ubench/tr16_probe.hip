// tr16_probe.hip — what ds_read_b64_tr_b16 delivers on gfx950, observed. LDS holds element e = its own index
// (16-bit); every lane passes the address of "its" 4-element group of a [4 rows][16 cols] block per 16-lane group
// (row stride RS elements) and the kernel dumps the four elements each lane gets back.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/tr16_probe.hip -o tools/ubench/tr16_probe && tools/ubench/tr16_probe
// Expected (csrc/ganet_layer_bwd.hip relies on it): lane l of a group receives column (l & 15) of the block,
// elements j = 0..3 = rows 0..3, i.e. value = base + j * RS + (l & 15).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int RS) {
  __shared__ unsigned short s[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
  const unsigned short* p = s + g * 1024 + (l >> 2) * RS + (l & 3) * 4;
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  int bad = 0;
  for (int RS : {16, 128}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, RS);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("row stride %d elements\n", RS);
    for (int t = 0; t < 64; ++t) {
      printf("lane %2d:", t);
      for (int j = 0; j < 4; ++j) {
        const int expect = (t >> 4) * 1024 + j * RS + (t & 15);
        printf(" %5d%s", h[t * 4 + j], h[t * 4 + j] == expect ? "" : "*");
        bad += h[t * 4 + j] != expect;
      }
      printf("\n");
    }
  }
  printf(bad ? "MISMATCH: %d elements differ from the assumed layout\n" : "layout as assumed (%d mismatches)\n", bad);
  return bad != 0;
}
