// mlp_panel.h -- pieces shared by the fused row-panel forward kernels (mlp.hip, mlps.hip): the LDS-DMA primitive, the
// 32 x 256 bf16 activation panel (the next layer's A operand) and the hidden-layer epilogue that fills it.
#pragma once
#include <type_traits>
#include "mlp.h"

namespace {
constexpr int BM = 32;            // rows per workgroup
constexpr int HP = 256;           // hidden width handled
constexpr int PANEL_HALF = BM * 256;          // one k half (128 columns) of the 32 x 256 activation panel

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst_uniform))   // (under SGPR pressure the compiler may hold it in a VGPR)
      : "memory");
}

// hidden-layer epilogue: bias + relu + dropout -> bf16 into the LDS panel (the next layer's A operand).  Kept lean on purpose
// (in-kernel trace + ISA count: the first version spent 630 instructions per wave here, 80 per element, mostly on per-element
// exec-masked 2-byte global stores and address arithmetic -- 7k ticks per epilogue, a third of the workgroup's time): the
// activations now reach global memory from the finished panel as whole rows (panel_to_global), and the four rows of an
// accumulator register group differ only by r in the swizzled chunk position ((c ^ (rb | r)) = (c ^ rb) ^ r).
template <int TNH>
__device__ __forceinline__ void hidden_epilogue(f32x4 (&acc)[2][TNH], const f32x4 (&bias_v)[TNH], int H, int rows, int m0, int wave, int fr,
                                                int fg, int mask_mode, const uint8_t* mask, int64_t ld_mask, uint32_t key,
                                                unsigned char* panel, uint32_t* gbits = nullptr, const bool two = true) {   // two = false: 16-row panel
  uint32_t bits = 0;
#pragma unroll
  for (int tn = 0; tn < TNH; ++tn) {
    const int n0 = wave * (16 * TNH) + tn * 16 + fg * 4;      // this lane's four columns n0 .. n0 + 3
    // panel image: k half (n / 128), row, chunk ((n % 128) / 8) ^ (row & 15), element n % 8
    unsigned char* col = panel + (n0 >> 7) * PANEL_HALF + (n0 & 7) * 2;
    const int c = (n0 & 127) >> 3;
    // (row blocks as compile-time constants; the second one under a uniform branch: a loop that breaks on the run-time flag would not
    //  be unrolled and its accumulator indexing would turn into select chains)
    auto row_block = [&](auto TMc) {
      constexpr int tm = decltype(TMc)::value;
      const int row = tm * 16 + fr, m = m0 + row;
      uint32_t word = 0;
      if (mask_mode == RECNN_MASK_HASH) word = mask_word(key, (uint32_t)(m >> 2), (uint32_t)(n0 >> 2));
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = fmaxf(acc[tm][tn][r] + bias_v[tn][r], 0.f);
        if (mask_mode == RECNN_MASK_EXTERNAL) v[r] = (m < rows && n0 + r < H && mask[(int64_t)m * ld_mask + n0 + r]) ? v[r] * 2.f : 0.f;
        else if (mask_mode == RECNN_MASK_HASH) v[r] = mask_keep(word, m & 3, r) ? v[r] * 2.f : 0.f;
        if (n0 + r >= H) v[r] = 0.f;
      }
      const uint32_t lo = pack_bf2(v[0], v[1]), hi = pack_bf2(v[2], v[3]);
      // gate bit = the ROUNDED activation is positive (what the backward kernels test on the stored bf16 value)
      if (lo & 0x7FFFu) bits |= 1u << (tn * 8 + tm * 4 + 0);
      if (lo & 0x7FFF0000u) bits |= 1u << (tn * 8 + tm * 4 + 1);
      if (hi & 0x7FFFu) bits |= 1u << (tn * 8 + tm * 4 + 2);
      if (hi & 0x7FFF0000u) bits |= 1u << (tn * 8 + tm * 4 + 3);
      *(uint2*)(col + row * 256 + ((c ^ fr) << 4)) = make_uint2(lo, hi);   // row & 15 == fr
    };
    row_block(std::integral_constant<int, 0>{});
    if (two) row_block(std::integral_constant<int, 1>{});
  }
  if (gbits) *gbits = bits;
}

// the finished 32 x 256 panel -> global [rows, ldg] bf16: whole 512-byte rows, one 16-byte chunk per thread and pass
template <int NW>
__device__ __forceinline__ void panel_to_global(const unsigned char* panel, bf16_t* gout, int64_t ldg, int m0, int rows, int tid) {
#pragma unroll
  for (int j = 0; j < (BM * 32) / (NW * 64); ++j) {
    const int idx = tid + j * NW * 64, row = idx >> 5, cc = idx & 31;
    const uint4 v = *(const uint4*)(panel + (cc >> 4) * PANEL_HALF + row * 256 + (((cc & 15) ^ (row & 15)) << 4));
    if (m0 + row < rows) *(uint4*)(gout + (int64_t)(m0 + row) * ldg + cc * 8) = v;
  }
}
}  // namespace
