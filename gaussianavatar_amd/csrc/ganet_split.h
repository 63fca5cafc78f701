// ganet_split.h — fp32 GEMM operands as three bf16 pieces for the bf16 matrix pipe (internal).
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at the VECTOR rate (157 TFLOP/s) but v_mfma_f32_32x32x16_bf16 sixteen
// times faster, and has no TF32-like form. The decoder GEMMs therefore feed the bf16 pipe with an EXACT
// three-way split of every fp32 operand,
//
//     a = a1 + a2 + a3,   a1 = trunc_bf16(a),  a2 = trunc_bf16(a - a1),  a3 = a - a1 - a2
//
// (8 + 8 + 8 significand bits: the subtractions are exact and a3 needs no rounding — for normal fp32 values; of a
// subnormal only its upper 7 mantissa bits survive, an error below 2^-132), and accumulate in fp32
// the six products whose weight is above 2^-24 of |a b|:
//
//     a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)        dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b|
//
// Each bf16 x bf16 product is exact in fp32 and an MFMA step adds 16 of them into the fp32 accumulator, so a
// K = 128 dot product goes through 8 x 6 accumulator roundings instead of the 128 of an fp32 FMA chain: measured
// against float64 the result is as close as (slightly closer than) the fp32-MFMA kernels it replaces
// (tests/test_fused_gpu.py::test_split_mfma_is_fp32_accurate) — this is an fp32 GEMM, not a bf16 one — at
// 6/16 of the matrix-pipe time, which moves the decoder kernels from MFMA-issue-bound to HBM-bound.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

#include "ganet_mlp_common.h"

namespace ganet {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f2u(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float u2f(unsigned v) { return __builtin_bit_cast(float, v); }

// (high half of `even`) | (high half of `odd`) << 16: two truncated bf16 in k order (v_perm_b32)
__device__ __forceinline__ unsigned pack_hi(float even, float odd) {
  return __builtin_amdgcn_perm(f2u(odd), f2u(even), 0x07060302u);
}

// eight consecutive k of one row -> the three bf16x8 fragments. 32 exact VALU ops + 12 packs.
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
  float r[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = v[i] - u2f(f2u(v[i]) & 0xffff0000u);
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = r[i] - u2f(f2u(r[i]) & 0xffff0000u);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p1[j] = pack_hi(v[2 * j], v[2 * j + 1]);
    p2[j] = pack_hi(r[2 * j], r[2 * j + 1]);
    p3[j] = pack_hi(q[2 * j], q[2 * j + 1]);
  }
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The six products of one 32 x 32 x 16 step, smallest terms first.
#define GANET_SPLIT_PRODUCTS(ACC, A1, A2, A3, B1, B2, B3) \
  do {                                                    \
    ACC = mfma_bf16(A3, B1, ACC);                         \
    ACC = mfma_bf16(A2, B2, ACC);                         \
    ACC = mfma_bf16(A1, B3, ACC);                         \
    ACC = mfma_bf16(A2, B1, ACC);                         \
    ACC = mfma_bf16(A1, B2, ACC);                         \
    ACC = mfma_bf16(A1, B1, ACC);                         \
  } while (0)

// LDS image of a weight matrix for the streamed-M kernels: three planes [rows][RU units of 16 bytes], a unit =
// 8 consecutive k of one row as bf16. The MFMA B fragment of lane (col, kg) for k-step s is unit 2 s + kg of row
// `col`: one ds_read_b128. 16 consecutive rows must land on 16 different 16-byte bank groups, hence the XOR.
template <int RU>
__device__ __forceinline__ int split_swizzle(int row) {
  static_assert(RU % 16 == 0 || RU % 16 == 10, "swizzle not derived for this row length");
  return (RU % 16 == 0) ? (row & 15) : ((row >> 3) & 1);
}
template <int RU>
__device__ __forceinline__ int split_unit(int row, int u) { return row * RU + (u ^ split_swizzle<RU>(row)); }


}  // namespace ganet
