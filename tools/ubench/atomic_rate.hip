// atomic_rate.hip — the rate at which gfx950 retires render_bwd's gradient records.
//
// render_bwd (gaussianavatar_amd/csrc/gsr_render.hip) ends every 64-entry segment with one 64-byte gradient record
// per entry (9 floats used of 16): the wave transposes its [entry][9] sums through LDS and issues 9 float-atomic
// instructions over the flattened array, so that an instruction covers ~7 whole records (8 lines of 64 B). This
// benchmark issues exactly that pattern with nothing around it: P = 200,000 records, S segments of 64 entries per
// launch (S = 37,500 = the 2.4 M surviving (4x4 block, Gaussian) pairs of a 2-frame launch of the bench scene).
//   index models   "tile"   : a segment's entries are distinct records of a 1,024-record neighbourhood (a tile's list
//                             holds ~1,000 Gaussians of one body region); segments of a workgroup share the region
//                  "random" : entries uniform over all P records
//   variants       flat9    : the kernel's pattern (9 instructions over the flattened [64][9] array)
//                  lane9    : lane = entry, 9 instructions each touching 64 different lines (round 1's pattern)
//                  store    : the flat pattern with plain stores (what the write path alone would take)
//                  flat9+ld : flat9 while every wave also streams 16-byte loads (512 B per entry: roughly what a
//                             segment fetches per entry) — atomics under the kernel's concurrent read traffic
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/atomic_rate.hip -o tools/ubench/atomic_rate && tools/ubench/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int NCOMP = 9, STRIDE = 16, WAVES = 4;

// XCD = true (round 6): the segments of a "tile" (32 consecutive segments = one 1,024-record neighbourhood) are all taken
// by workgroups of ONE XCD class (blockIdx.x % 8 == tile % 8), so a record is only ever updated from one XCD — does the
// atomic rate depend on who else touches the line?
template <int MODE, bool LOADS, bool XCD = false>
__global__ void __launch_bounds__(256) k(int S, const uint32_t* __restrict__ idx, float* __restrict__ grad,
                                         const float4* __restrict__ stream, size_t stream_n, float* __restrict__ sink) {
  __shared__ float s_g[WAVES][64 * NCOMP];
  __shared__ uint32_t s_gi[WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = XCD ? (gridDim.x / 8) * WAVES : gridDim.x * WAVES;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int first = XCD ? (blockIdx.x / 8) * WAVES + wave : blockIdx.x * WAVES + wave;
  for (int i = first; i < (XCD ? S / 8 : S); i += nw) {
    const int s = XCD ? ((i / 32) * 8 + (int)(blockIdx.x % 8)) * 32 + i % 32 : i;
    if (s >= S) continue;
    const uint32_t gi = idx[(size_t)s * 64 + lane];
    if (LOADS) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {       // 32 x 16 B per lane = 512 B per entry
        const float4 v = stream[((size_t)s * 64 * 32 + (size_t)j * 64 + lane) % stream_n];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    const float v = 1.0f + (float)lane * 1e-3f;
    if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < NCOMP; ++q) unsafeAtomicAdd(&grad[(size_t)gi * STRIDE + q], v);
      continue;
    }
    float* sg = s_g[wave];
#pragma unroll
    for (int q = 0; q < NCOMP; ++q) sg[lane * NCOMP + q] = v + (float)q;
    s_gi[wave][lane] = gi;
    __builtin_amdgcn_wave_barrier();
#pragma unroll 3
    for (int r = 0; r < NCOMP; ++r) {
      const int fl = r * 64 + lane;
      const int e = (fl * 7282) >> 16;
      const int q = fl - e * NCOMP;
      const uint32_t g2 = s_gi[wave][e];
      const float val = sg[fl];
      if (MODE == 0) unsafeAtomicAdd(&grad[(size_t)g2 * STRIDE + q], val);
      else grad[(size_t)g2 * STRIDE + q] = val;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (LOADS && acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int MODE, bool LOADS, bool XCD = false>
int run(const char* name, int S, const uint32_t* idx, float* grad, const float4* stream, size_t stream_n, float* sink, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, LOADS, XCD>), dim3(blocks), dim3(256), 0, 0, S, idx, grad, stream, stream_n, sink);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, LOADS, XCD>), dim3(blocks), dim3(256), 0, 0, S, idx, grad, stream, stream_n, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, recs = (double)S * 64;
  printf("%-34s blocks %5d  %8.1f us  %7.2f G records/s  (%5.1f ns per 1000 records)%s\n", name, blocks, us, recs / us / 1e3,
         us * 1e3 / (recs / 1e3), LOADS ? "  [+ 512 B of streaming loads per entry]" : "");
  return 0;
}

int main() {
  const int P = 200000, S = 37500;
  std::vector<uint32_t> tile((size_t)S * 64), rnd((size_t)S * 64);
  uint64_t st = 0x9e3779b97f4a7c15ull;
  auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (int s = 0; s < S; ++s) {
    // a run of 32 consecutive segments (one tile's worth) shares a 1,024-record neighbourhood
    const uint32_t base = (uint32_t)(((uint64_t)(s / 32) * 2654435761ull) % (uint64_t)(P - 1024));
    const uint32_t off = (uint32_t)(next() % 1024);
    for (int l = 0; l < 64; ++l) {
      tile[(size_t)s * 64 + l] = base + (off + 16 * l + (uint32_t)(next() % 16)) % 1024;   // 64 distinct records
      rnd[(size_t)s * 64 + l] = (uint32_t)(next() % P);
    }
  }
  uint32_t *d_tile, *d_rnd; float *grad, *sink; float4* stream;
  const size_t stream_n = (size_t)64 << 20;     // 1 GiB of float4: never cached
  CK(hipMalloc(&d_tile, tile.size() * 4)); CK(hipMalloc(&d_rnd, rnd.size() * 4));
  CK(hipMalloc(&grad, (size_t)P * STRIDE * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&stream, stream_n * 16));
  CK(hipMemcpy(d_tile, tile.data(), tile.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rnd, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(grad, 0, (size_t)P * STRIDE * 4)); CK(hipMemset(stream, 0, stream_n * 16));
  printf("P = %d records of 64 B, %d segments x 64 entries = %.2f M records per launch\n", P, S, S * 64 / 1e6);
  for (int blocks : {512, 2048}) {
    for (int m = 0; m < 2; ++m) {
      const uint32_t* ix = m ? d_rnd : d_tile;
      const char* nm = m ? "random" : "tile";
      char buf[64];
      snprintf(buf, sizeof buf, "flat9 atomics, %s", nm); if (run<0, false>(buf, S, ix, grad, stream, stream_n, sink, blocks)) return 1;
      if (!m) { snprintf(buf, sizeof buf, "flat9 atomics, tile, one XCD each"); if (run<0, false, true>(buf, S, ix, grad, stream, stream_n, sink, blocks)) return 1; }
      if (!m) { snprintf(buf, sizeof buf, "flat9 stores, tile, one XCD each"); if (run<2, false, true>(buf, S, ix, grad, stream, stream_n, sink, blocks)) return 1; }
      snprintf(buf, sizeof buf, "lane9 atomics, %s", nm); if (run<1, false>(buf, S, ix, grad, stream, stream_n, sink, blocks)) return 1;
      snprintf(buf, sizeof buf, "flat9 plain stores, %s", nm); if (run<2, false>(buf, S, ix, grad, stream, stream_n, sink, blocks)) return 1;
      snprintf(buf, sizeof buf, "flat9 atomics + loads, %s", nm); if (run<0, true>(buf, S, ix, grad, stream, stream_n, sink, blocks)) return 1;
      snprintf(buf, sizeof buf, "loads only (stores), %s", nm); if (run<2, true>(buf, S, ix, grad, stream, stream_n, sink, blocks)) return 1;
    }
  }
  return 0;
}
