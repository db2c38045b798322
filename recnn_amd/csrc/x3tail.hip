// x3tail.hip -- layers 2 and 3 of an Actor / Critic for a 32-row panel, split-bf16 compute type (x3.h), ONE launch for all networks
// of a step whose layer-1 activations exist:
//     h2  = dropout(relu(h1 W2^T + b2))                      recnn/nn/models.py:68-71, :209-211
//     out = h2 W3^T + b3 (+ clamp(noise))   (actor, :72)     |     q = h2 . w3 + b3   (critic, :212)
// Replaces, on the split-bf16 step, the grouped layer-2 launch, the actors' layer-3 launch and the target critic's layer-2
// launch (three launches of 6-11 us each whose k = 256 contractions are all start-up and epilogue) by two launches that keep h2 on
// chip.  Same schedule as the bf16 fused forward (mlps.hip): every weight slab the workgroup will multiply -- eight 64-physical-k
// slabs of W2 (32 logical k each: [hi 32 | lo 32]), then four double slabs of W3 -- is one sequence pushed through a 4-stage LDS
// ring by global_load_lds, three slabs ahead of the consumer, one barrier per slab; the h1 panel (32 rows x 512 physical columns)
// arrives by DMA too and is overwritten by h2.  The x3 pairing turns a slab into three MFMAs per 16 x 16 block.
//
// LDS: 4 x 32 KB ring + 32 KB panel = 160 KB.  Slab rows are 128 bytes (chunk c of row r at c ^ ((r >> 1) & 7)); the panel is four
// 128-column blocks of 32 rows x 256 bytes (chunk c of row r at c ^ (r & 15)): the layouts of mlps.hip / mlp_panel.h.
#include "x3tail.h"
#include "x3.h"

namespace {
constexpr int NW = 16, BM = 32;
constexpr int SLAB = 256 * 128;              // one ring stage: 256 weight rows x 64 physical k
constexpr int NST = 4;
constexpr int PBLK = BM * 256;               // one 128-physical-column block of the panel (8 KB)
constexpr int PANEL_OFF = NST * SLAB;        // 128 KB
constexpr int LDS_TOTAL = PANEL_OFF + 4 * PBLK;   // 160 KB
constexpr int OW = 8;                        // waves of the actor's 128-column output layer
constexpr int RW = BM / NW;                  // critic rows per wave

__device__ __forceinline__ void dma_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "m0");
}

// acc[tm] += A(panel slab q: physical k 64 q .. 64 q + 63 = logical k 32 q .. + 31 as [hi | lo]) * W(rows wrow0 + fr of the ring
// slab)^T, three MFMAs; weights first: acc[tm][r] = C[row 16 tm + fr][column .. + 4 fg + r]
__device__ __forceinline__ void mma_slab(const unsigned char* panel, int q, const unsigned char* sb, f32x4 (&acc)[2], int wrow0, int fr, int fg) {
  const int sw = (fr >> 1) & 7;
  const unsigned char* sa = panel + (q >> 1) * PBLK;
  const int pah = ((((q & 1) * 8) + fg) ^ fr) * 16, pal = ((((q & 1) * 8) + 4 + fg) ^ fr) * 16;
  const int pbh = (fg ^ sw) * 16, pbl = ((4 + fg) ^ sw) * 16;
  const bf16x8 bh = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + (wrow0 + fr) * 128 + pbh));
  const bf16x8 bl = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + (wrow0 + fr) * 128 + pbl));
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)(sa + (tm * 16 + fr) * 256 + pah));
    const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(sa + (tm * 16 + fr) * 256 + pal));
    acc[tm] = x3_mfma(bh, bl, ah, al, acc[tm]);
  }
}

// four logical values of row `row`, columns n0 .. n0 + 3 (n0 % 4 == 0) -> the panel as split halves
__device__ __forceinline__ void panel_put4(unsigned char* panel, int row, int n0, const float (&v)[4]) {
  uint2 hi, lo;
  x3_split4(v, hi, lo);
  const int X = x3_col(n0);
  unsigned char* base = panel + (X >> 7) * PBLK + row * 256 + (X & 7) * 2;
  const int c = (X & 127) >> 3;
  *(uint2*)(base + ((c ^ (row & 15)) << 4)) = hi;
  *(uint2*)(base + (((c + 4) ^ (row & 15)) << 4)) = lo;
}
__device__ __forceinline__ void panel_get4(const unsigned char* panel, int row, int n0, float (&v)[4]) {
  const int X = x3_col(n0);
  const unsigned char* base = panel + (X >> 7) * PBLK + row * 256 + (X & 7) * 2;
  const int c = (X & 127) >> 3;
  const uint2 hi = *(const uint2*)(base + ((c ^ (row & 15)) << 4)), lo = *(const uint2*)(base + (((c + 4) ^ (row & 15)) << 4));
  v[0] = bf2f((bf16_t)(hi.x & 0xFFFFu)) + bf2f((bf16_t)(lo.x & 0xFFFFu)); v[1] = bf2f((bf16_t)(hi.x >> 16)) + bf2f((bf16_t)(lo.x >> 16));
  v[2] = bf2f((bf16_t)(hi.y & 0xFFFFu)) + bf2f((bf16_t)(lo.y & 0xFFFFu)); v[3] = bf2f((bf16_t)(hi.y >> 16)) + bf2f((bf16_t)(lo.y >> 16));
}
}  // namespace

__global__ __launch_bounds__(NW * 64) void x3_tail_kernel(const X3TailBatch batch) {
  kernarg_prefetch<(int)sizeof(X3TailProb)>((int)(blockIdx.y * sizeof(X3TailProb)));
  const int by = __builtin_amdgcn_readfirstlane((int)blockIdx.y), bx = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  const X3TailProb& P = batch.p[by];
  const int m0 = bx * BM;
  if (m0 >= P.rows) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  unsigned char* panel = lds + PANEL_OFF;
  const bool actor = P.W3 != nullptr;

  // ---- the h1 panel: 32 rows x 1 KB.  One DMA instruction = one 128-column block of 4 rows (1 KB, contiguous in the image);
  // wave w: block w & 3, rows 8 (w >> 2) .. + 7 as two instructions.  Source chunk of image position p of row r: p ^ (r & 15).
  {
    const int blk = wave & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r4 = (wave >> 2) * 2 + j;                       // group of 4 rows
      const int row = r4 * 4 + (lane >> 4), p = lane & 15;
      const int gr = min(m0 + row, P.rows - 1);
      const unsigned voff = (unsigned)((gr * (int)P.ldh + blk * 128) * 2 + ((p ^ (row & 15)) << 4));
      dma_s(voff, P.h1, lds0 + PANEL_OFF + blk * PBLK + r4 * 1024);
    }
  }
  // ---- weight slabs: per-lane DMA geometry as mlps.hip (one wave instruction = 8 rows x 128 B)
  const int q_row = lane >> 3, q_pos = lane & 7;
  const int l_row = wave * 8 + q_row;
  const int l_c = (q_pos ^ ((l_row >> 1) & 7)) * 16;
  const unsigned voff_sq = (unsigned)(l_row * (int)P.ldw2 * 2 + l_c);       // W2 / W3 share the pitch (x3tail_launch)
  const unsigned wave_kb = wave * 1024;
  const int nslab = 8 + (actor ? 4 : 0);
  int issued = 0, consumed = 0;
  auto issue = [&]() {
    const int k = issued;
    if (k >= nslab) return;
    const unsigned wb = lds0 + (k & (NST - 1)) * SLAB + wave_kb;
    ++issued;
    if (k < 8) {                                                 // W2 k-slab k: rows l_row and l_row + 128
      const char* b0 = (const char*)P.W2 + k * 128;
      dma_s(voff_sq, b0, wb);
      dma_s(voff_sq, b0 + 256 * P.ldw2, wb + NW * 1024);
    } else {                                                     // W3 (128 rows): k-slabs 2 p and 2 p + 1 as image rows 0..127 / 128..255
      const char* b0 = (const char*)P.W3 + 2 * (k - 8) * 128;
      dma_s(voff_sq, b0, wb);
      dma_s(voff_sq, b0 + 128, wb + NW * 1024);
    }
  };
  auto next = [&]() -> const unsigned char* {
    const int c = consumed++;
    const int y = issued - c - 1;                                // younger slabs in flight (2 instructions each)
    if (y >= 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if (y == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();    // slab c (and everything older: the panel) landed for every wave; slab c - 1's stage is free
    issue();
    return lds + (c & (NST - 1)) * SLAB;
  };
  issue(); issue(); issue();

  // ---- everything the epilogues read from global memory (the compiler's wait for these also drains the first DMAs: they have
  // to land anyway)
  const int n0 = wave * 16 + fg * 4;                             // this lane's four hidden columns
  f32x4 b2v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (n0 + 3 < P.H) b2v = *(const f32x4*)(P.b2 + n0);
  const int no = (wave & (OW - 1)) * 16 + fg * 4;                // this lane's four output columns (actor, waves 0..7)
  float v3[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    v3[r] = actor ? (no + r < P.out_dim ? P.b3[no + r] : 0.f) : ((lane * 4 + r < P.H) ? P.w3row[lane * 4 + r] : 0.f);
  const float b3s = actor ? 0.f : P.b3[0];
  uint32_t key2 = 0;
  if (P.mask_mode == RECNN_MASK_HASH) key2 = mask_key(P.seed, (P.step_ptr ? *P.step_ptr : 0) + P.step_add, P.stream2);
  asm volatile("" : "+v"(b2v));
#pragma unroll
  for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(v3[r]));

  // ------------------------------------------------------------------ layer 2
  f32x4 acc[2];
  acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < 8; ++q) {
    const unsigned char* st = next();
    mma_slab(panel, q, st, acc, wave * 16, fr, fg);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();      // everyone is done reading the h1 panel: h2 takes its place
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int row = tm * 16 + fr, m = m0 + row;
    uint32_t word = 0;
    if (P.mask_mode == RECNN_MASK_HASH) word = mask_word(key2, (uint32_t)(m >> 2), (uint32_t)(n0 >> 2));
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = fmaxf(acc[tm][r] + b2v[r], 0.f);
      if (P.mask_mode == RECNN_MASK_EXTERNAL) v[r] = (m < P.rows && n0 + r < P.H && P.mask2[(int64_t)m * P.ld_mask + n0 + r]) ? v[r] * 2.f : 0.f;
      else if (P.mask_mode == RECNN_MASK_HASH) v[r] = mask_keep(word, m & 3, r) ? v[r] * 2.f : 0.f;
      if (n0 + r >= P.H) v[r] = 0.f;
    }
    panel_put4(panel, row, n0, v);
  }

  if (actor) {
    // ---------------------------------------------------------------- layer 3: 32 x 128 outputs on waves 0..7
    f32x4 o[2];
    o[0] = o[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < 4; ++p) {
      const unsigned char* st = next();                          // (its barrier completes the h2 panel for p = 0)
      if (p == 0 && P.h2) {
#pragma unroll
        for (int j = 0; j < (BM * 64) / (NW * 64); ++j) {         // the finished panel -> global, whole 1 KB rows
          const int idx = tid + j * NW * 64, row = idx >> 6, cc = idx & 63;
          const uint4 val = *(const uint4*)(panel + (cc >> 4) * PBLK + row * 256 + (((cc & 15) ^ (row & 15)) << 4));
          if (m0 + row < P.rows) *(uint4*)((bf16_t*)P.h2 + (int64_t)(m0 + row) * P.ldh + cc * 8) = val;
        }
      }
      if (wave < OW) {
        mma_slab(panel, 2 * p, st, o, wave * 16, fr, fg);
        mma_slab(panel, 2 * p + 1, st, o, 128 + wave * 16, fr, fg);
      }
    }
    if (wave < OW) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int m = m0 + tm * 16 + fr;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ncol = no + r < P.out_dim;
          v[r] = o[tm][r] + v3[r];
          if (P.addend && ncol && m < P.rows) {
            const float z = P.addend[(int64_t)m * P.ld_add + no + r];
            v[r] += fminf(fmaxf(z, -P.add_clip), P.add_clip);
          }
          if (!ncol) v[r] = 0.f;
        }
        if (m < P.rows && no < P.out_dim) {
          uint2 hi, lo;
          x3_split4(v, hi, lo);
          bf16_t* dst = (bf16_t*)P.out + (int64_t)m * P.ldo + x3_col(no);
          if (no + 3 < P.out_dim) {
            *(uint2*)dst = hi;
            *(uint2*)(dst + 32) = lo;
          } else {
            for (int r = 0; r < 4; ++r)
              if (no + r < P.out_dim) {
                dst[r] = (bf16_t)((r < 2 ? hi.x : hi.y) >> ((r & 1) * 16));
                dst[32 + r] = (bf16_t)((r < 2 ? lo.x : lo.y) >> ((r & 1) * 16));
              }
          }
        }
      }
    }
  } else {
    // ---------------------------------------------------------------- critic: q[m] = h2[m, :] . w3 + b3
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();    // h2 panel complete
    if (P.h2) {
#pragma unroll
      for (int j = 0; j < (BM * 64) / (NW * 64); ++j) {
        const int idx = tid + j * NW * 64, row = idx >> 6, cc = idx & 63;
        const uint4 val = *(const uint4*)(panel + (cc >> 4) * PBLK + row * 256 + (((cc & 15) ^ (row & 15)) << 4));
        if (m0 + row < P.rows) *(uint4*)((bf16_t*)P.h2 + (int64_t)(m0 + row) * P.ldh + cc * 8) = val;
      }
    }
    float qsum = 0.f;
    for (int i = 0; i < RW; ++i) {
      const int row = wave * RW + i;
      float hv[4];
      panel_get4(panel, row, lane * 4, hv);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += lane * 4 + j < P.H ? hv[j] * v3[j] : 0.f;
      s = wave_sum(s);
      const float qv = s + b3s;
      if (m0 + row < P.rows) {
        qsum += qv;
        if (lane == 0 && P.q) P.q[m0 + row] = qv;
      }
    }
    if (P.q_part && lane == 0) P.q_part[(int64_t)bx * NW + wave] = qsum;   // sum of Q over this wave's rows (policy loss partials)
  }
}

int x3tail_init() {
  return recnn_check_hip(hipFuncSetAttribute((const void*)x3_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL), "x3 tail attr");
}

int x3tail_parts_per_panel() { return NW; }

int x3tail_launch(const X3TailBatch& b, int nprob, hipStream_t s) {
  int rows = 0;
  for (int i = 0; i < nprob; ++i) {
    const X3TailProb& p = b.p[i];
    if (p.rows > rows) rows = p.rows;
    if (p.H != 256 || (p.W3 && p.out_dim != 128)) { recnn_set_error("x3 tail: hidden must be 256 (and the actor's output 128)"); return RECNN_E_UNSUPPORTED; }
    if (!p.h1 || !p.W2 || !p.b2 || (p.W3 ? (!p.b3 || !p.out) : (!p.w3row || !p.b3))) { recnn_set_error("x3 tail: null operand"); return RECNN_E_INVALID; }
    if (p.W3 && p.ldw3 != p.ldw2) { recnn_set_error("x3 tail: W2 / W3 shadows must share one pitch"); return RECNN_E_INVALID; }
    if (p.ldh < 512 || (p.ldh & 7) || p.ldw2 < 512 || (p.ldw2 & 7) || (((uintptr_t)p.h1 | (uintptr_t)p.W2 | (uintptr_t)p.W3 | (uintptr_t)p.h2) & 15)) {
      recnn_set_error("x3 tail: operands must be 16-byte aligned split rows of >= 512 physical columns");
      return RECNN_E_INVALID;
    }
  }
  if (rows <= 0 || nprob <= 0) return 0;
  if (nprob > X3TAIL_MAX_GROUP) { recnn_set_error("x3 tail: group too large"); return RECNN_E_INVALID; }
  hipLaunchKernelGGL(x3_tail_kernel, dim3((rows + BM - 1) / BM, nprob), dim3(NW * 64), LDS_TOTAL, s, b);
  return recnn_check_hip(hipGetLastError(), "x3_tail_kernel");
}
