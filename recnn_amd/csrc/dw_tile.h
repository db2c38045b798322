// dw_tile.h -- the 64 x 64 output tile of a weight-gradient GEMM dW[m][n] = sum_b dZ[b][m] * X[b][n] (bf16, gfx950) of
// gemm_dw_dma_kernel (gemm.hip: split-batch slabs; NWK > 1 = k-groups inside the workgroup, the round-3 experiment).  dwadam.hip
// (round 6: 32 x 64 tiles, whole batch per workgroup, optimizer in the epilogue) shares the LDS image and the transpose reads.
//
// BOTH operands are k-strided (k = batch row), i.e. stored with the tile dimension contiguous.  The rows go global -> LDS
// untouched (`global_load_lds_dwordx4`) and the MFMA fragments are read with gfx950's `ds_read_b64_tr_b16`, which hands lane i
// of a 16-lane group column i of a [4 k][16 cols] block.  A k stage = SUB batch rows x 128 bytes per operand; NS ring slots,
// all filled before the first MFMA.
// LDS image of a stage: row b of an operand is 8 chunks of 16 bytes; chunk pair p of row b sits at pair position
// p ^ f(b), f(b) = ((b>>1)&1) | (((b>>3)&1)<<1), so the 8 rows x 32 bytes one half-wave transpose read touches
// (rows {0..3} and {8..11} of a k step, same 16 columns) fall on 8 different 32-byte bank groups.
#pragma once
#include "gemm.h"

typedef short v4s16 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dw_dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst_uniform)
      : "memory");
}

// acc += dZ[kbeg..kend, m0..m0+63]^T * X[kbeg..kend, n0..n0+63] for the calling workgroup of 256 * NWK threads: 4 * NWK waves as
// NWK k-groups x (2 x 2) wave tiles of 32 x 32.  k-group kq takes the 32-row k steps ks with ks % NWK == kq of every stage, so
// with NWK > 1 the accumulators are PARTIAL sums (the caller adds the NWK groups in a fixed order); NWK = 4 puts four waves on
// every SIMD of a one-workgroup-per-CU launch, which is what hides the LDS-read -> scale -> MFMA chain of a k step (measured:
// the whole batch per tile with one wave per SIMD ran 43 us for what the DMA rate allows in 6).
// dsmem: NS * SUB * 256 bytes of ring + scale_cap floats for the per-row scales (P.a_row_scale) of the range.
// acc[tm][tn][r] = dW[m0 + wm0 + 16 tm + 4 fg + r][n0 + wn0 + 16 tn + fr].
// after_prologue(): called once, right after the ring's first NS stages have been requested (work placed there -- e.g. the
// optimizer's per-launch scalars -- runs under the first memory latency); anything it writes to LDS outside the ring and the
// scale area is visible to every wave after the k loop's barriers (the loop must run at least once, or the caller syncs).
template <int SUB, int NS, int NWK, class F>
__device__ __forceinline__ void dw_tile_accumulate(const GemmProb& P, const int m0, const int n0, const int kbeg, const int kend,
                                                   unsigned char* dsmem, const int scale_cap, f32x4 (&acc)[2][2], F&& after_prologue,
                                                   const int probe = 0) {   // timing probes: 1 no row scale, 2 no LDS reads / MFMA, 4 no DMA
  constexpr int OP_BYTES = SUB * 128;          // one operand of a stage
  constexpr int STAGE_BYTES = 2 * OP_BYTES;
  constexpr int NI = SUB / (16 * NWK);         // DMA instructions per wave and stage
  constexpr int NT = 256 * NWK;                // threads of the workgroup
  static_assert(SUB % (16 * NWK) == 0 && (SUB / 32) % NWK == 0, "stage rows must split over the waves and k-groups");
  constexpr int RG = SUB / 8;                  // 8-row groups per operand and stage
  const unsigned lds0 = (unsigned)(size_t)dsmem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = wave >> 2;                    // k-group of this wave
  const int wm0 = ((wave >> 1) & 1) * 32, wn0 = (wave & 1) * 32;
  const int fr = lane & 15, fg = lane >> 4;
  const GemmSeg& G = P.seg[0];
  const int Kc = G.K;
  const int nt = kend > kbeg ? (kend - kbeg + SUB - 1) / SUB : 0;

  // DMA geometry: wave instruction g = wave*NI + i covers rows 8*(g % RG) .. +7 of operand g / RG (64 lanes x 16 bytes)
  const int d_row = lane >> 3, d_slot = lane & 7;
  auto issue = [&](int t, int stage) {
    const int k0 = kbeg + t * SUB;
    const unsigned sbase = lds0 + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int g = wave * NI + i;
      const int op = g / RG, rg = g % RG;
      const int f = ((d_row >> 1) & 1) | ((rg & 1) << 1);
      const int c = (((d_slot >> 1) ^ f) << 1) | (d_slot & 1);
      const int row = min(k0 + rg * 8 + d_row, kend - 1);  // clamped rows are masked out of the A fragments below
      const char* src = op == 0 ? (const char*)G.A + ((int64_t)row * G.lda + m0) * 2 + c * 16
                                : (const char*)G.B + ((int64_t)row * G.ldb + n0) * 2 + c * 16;
      if (!(probe & 4)) dw_dma16(src, sbase + op * OP_BYTES + rg * 1024);
    }
  };

#pragma unroll
  for (int i = 0; i < NS; ++i)
    if (i < nt) issue(i, i);
  after_prologue();
  // per-row scales of this workgroup's k range -> LDS (read back in the k loop; a global load there would stall every
  // k step for a memory latency)
  float* sds = (float*)(dsmem + NS * STAGE_BYTES);
  const bool lds_scale = P.a_row_scale && (kend - kbeg) <= scale_cap;
  if (lds_scale) {
    for (int i = tid; i < kend - kbeg; i += NT) sds[i] = P.a_row_scale[kbeg + i];
    for (int i = kend - kbeg + tid; i < ((kend - kbeg + 63) & ~63); i += NT) sds[i] = 0.f;
    __syncthreads();
  }
  for (int t = 0; t < nt; ++t) {
    // stages issued so far: the NS of the prologue plus one per iteration 1..t-1
    const int younger = min(nt - 1, NS - 1 + max(t - 1, 0)) - t;
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // stage t is in LDS; every wave is done reading stage t-1
    if (t >= 1 && t - 1 + NS < nt) issue(t - 1 + NS, (t - 1) % NS);
    const unsigned char* sa = dsmem + (t % NS) * STAGE_BYTES;
    const unsigned char* sb = sa + OP_BYTES;
    const int k0 = kbeg + t * SUB;
    const bool tail = k0 + SUB > kend;
#pragma unroll
    for (int kj = 0; kj < SUB / (32 * NWK); ++kj) {
      if (probe & 2) break;
      const int ks = kj * NWK + kq;
      v4s16 a[2][2], b[2][2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = ks * 32 + fg * 8 + half * 4 + (fr >> 2);
        const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
        const int rbyte = row * 128 + (fr & 1) * 8;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const int c = ((wm0 + tm * 16) >> 3) + ((fr & 3) >> 1);  // 16-byte chunk holding the lane's 4 columns
          const int slot = (((c >> 1) ^ f) << 1) | (c & 1);
          a[tm][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4s16*)(sa + rbyte + slot * 16));
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int c = ((wn0 + tn * 16) >> 3) + ((fr & 3) >> 1);
          const int slot = (((c >> 1) ^ f) << 1) | (c & 1);
          b[tn][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4s16*)(sb + rbyte + slot * 16));
        }
      }
      if (tail) {  // batch rows past the end of this range: zero the A side (the B side holds finite, clamped rows)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int kk = k0 + ks * 32 + fg * 8 + half * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kk + j >= kend) { a[0][half][j] = 0; a[1][half][j] = 0; }
        }
      }
      if (P.a_row_scale && !(probe & 1)) {  // uniform: A rows are unit backward tensors, multiply batch row k by its loss seed d_k
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int kk = k0 + ks * 32 + fg * 8 + half * 4;
          float dv[4];
          if (lds_scale) {
            const float4 d4 = *(const float4*)(sds + (kk - kbeg));
            dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
          } else if (kk + 3 < Kc) {   // kk is a multiple of 4 and the scale array is 16-byte aligned
            const float4 d4 = *(const float4*)(P.a_row_scale + kk);
            dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) dv[j] = P.a_row_scale[min(kk + j, Kc - 1)];
          }
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) {
            const uint2 raw = __builtin_bit_cast(uint2, a[tm][half]);
            const uint2 sc = make_uint2(pack_bf2(bf2f((bf16_t)(raw.x & 0xFFFF)) * dv[0], bf2f((bf16_t)(raw.x >> 16)) * dv[1]),
                                        pack_bf2(bf2f((bf16_t)(raw.y & 0xFFFF)) * dv[2], bf2f((bf16_t)(raw.y >> 16)) * dv[3]));
            a[tm][half] = __builtin_bit_cast(v4s16, sc);
          }
        }
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          struct { v4s16 lo, hi; } av = {a[tm][0], a[tm][1]}, bv = {b[tn][0], b[tn][1]};
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                                acc[tm][tn], 0, 0, 0);
        }
    }
  }
}
