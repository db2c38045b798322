// x3.h -- the split-bf16 compute type ("bf16x3", RECNN_BF16X3): fp32-grade contractions on the bf16 matrix cores.
//
// A value x is held as TWO bfloat16 numbers, hi = bf16(x) and lo = bf16(x - hi) (x - hi is exact in fp32; hi + lo carries
// 16-17 significand bits), and a product a * b is evaluated as
//     a_hi * b_hi + a_hi * b_lo + a_lo * b_hi          (the dropped a_lo * b_lo is <= 2^-16 |a b|)
// by three v_mfma_f32_16x16x32_bf16 into the same fp32 accumulator.  Same bytes per element as fp32, 5x the fp32 MFMA
// ceiling; measured against the CPU oracle the 200-step DDPG loss curve stays within 5e-5 (bf16: 2e-4 .. 2e-3; the 1e-4
// of north_star holds), see tests/x3_numerics.py and tests/test_gpu_x3.py.
//
// MEMORY FORMAT.  A logical row of C values is a bf16 row of 2 * roundup(C, 32) elements: logical columns are taken in
// groups of 32, and group g occupies physical columns [64 g, 64 g + 64) as [hi(32) | lo(32)]:
//     hi(r, c) at r * ld + x3_col(c),   lo(r, c) at r * ld + x3_col(c) + 32,    x3_col(c) = 2 (c & ~31) + (c & 31)
// so that
//   * as a CONTRACTION dimension (k contiguous) a 64-element physical step is one MFMA k step of hi and one of lo: the
//     inner loops of the bf16 kernels run unchanged over physical k, only the pairing of the fragments changes;
//   * as a TILE dimension (k strided: dX's weights, both operands of dW) a 16-column MFMA block is all-hi or all-lo;
//   * any aligned run of 4 / 8 logical columns is one 8 / 16-byte access for hi and one for lo.
// Leading dimensions and contraction lengths handed to the kernels are PHYSICAL (2x logical); output extents M / N,
// bias / mask / column indices are LOGICAL.
#pragma once
#include "common.h"

__host__ __device__ inline int x3_col(int c) { return ((c & ~31) << 1) | (c & 31); }
__host__ __device__ inline int64_t x3_ld(int64_t logical_cols) { return 2 * ((logical_cols + 31) / 32 * 32); }

#if defined(__HIPCC__)
__device__ inline void x3_split(float v, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}
// four consecutive logical values (c % 4 == 0) -> packed hi pair words / lo pair words
__device__ inline void x3_split4(const float (&v)[4], uint2& hi, uint2& lo) {
  hi = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
  const float r0 = v[0] - bf2f((bf16_t)(hi.x & 0xFFFFu)), r1 = v[1] - bf2f((bf16_t)(hi.x >> 16));
  const float r2 = v[2] - bf2f((bf16_t)(hi.y & 0xFFFFu)), r3 = v[3] - bf2f((bf16_t)(hi.y >> 16));
  lo = make_uint2(pack_bf2(r0, r1), pack_bf2(r2, r3));
}
__device__ inline float x3_load(const bf16_t* row, int c) {
  const bf16_t* p = row + x3_col(c);
  return bf2f(p[0]) + bf2f(p[32]);
}
__device__ inline void x3_store(bf16_t* row, int c, float v) {
  bf16_t hi, lo;
  x3_split(v, hi, lo);
  bf16_t* p = row + x3_col(c);
  p[0] = hi;
  p[32] = lo;
}
// 3-product accumulate of one 32-k logical step: fragments as the bf16 kernels read them (8 bf16 per lane)
__device__ __forceinline__ f32x4 x3_mfma(const bf16x8 a_hi, const bf16x8 a_lo, const bf16x8 b_hi, const bf16x8 b_lo, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, acc, 0, 0, 0);
}
#endif

struct GemmLaunch;
int x3_gemm_launch(GemmLaunch* L, hipStream_t stream);   // RECNN_BF16X3 problems of gemm_launch (gemm.hip hands them over)
int x3_fwd_launch(GemmLaunch* L, hipStream_t stream);    // defined in gemm.hip (the LDS-DMA forward kernel with the x3 pairing)
int x3_init();
int rows_to_x3_launch(const float* xs, const float* xn, bf16_t* hs, bf16_t* hn, int rows, int cols, int64_t ld32, int64_t ldh, hipStream_t s);
