// mlpt.hip -- the "tail" of an Actor / Critic application on a 32-row panel: layers 2 and 3, the TD head and the learning
// critic's layer-2 backward, starting from the layer-1 activations a tiled GEMM produced (l1gemm.hip).  bf16, gfx950.
//
//   Actor            h2 = drop(relu(h1 W2^T + b2));  out = h2 W3^T + b3 (+ clip(noise))          recnn/nn/models.py:70-73
//   Critic (Q only)  h2 as above;  q = h2 . w3 + b3                                               recnn/nn/models.py:211-213
//   Critic (learn)   ... plus  y = clamp(r + (1 - done) gamma min_t Q'_t),  d = 2 (q - y) / B,  sum (q - y)^2
//                    (recnn/nn/update/misc.py:6-7,33-39, td3.py:83-93) and the backward of linear3 / linear2:
//                    dz2 = d w3 s [h2 > 0],  dz1 = d ((u2 W2) s [h1 > 0]),  per-panel sums for dw3, db3, db2, db1
//
// Same panel layout and arithmetic as mlps.hip's phases after layer 1 (one 16-wave workgroup, 32 rows x 256 hidden columns,
// activations in a swizzled LDS panel that is the next layer's A operand, weights streamed as 32 KB k-slabs) -- but a
// workgroup here streams 128 KB (critic) or 192 KB (actor) instead of 1.0-1.2 MB, ALL of its weights are requested in the
// first instructions (the ring holds four slabs: everything a critic needs), and because the frozen networks were applied
// before (engine.hip: their Q' arrive as per-row scalars) the TD head needs no cross-workgroup hand-off: the critic's own
// workgroup evaluates it, so the backward tensors leave already multiplied by the per-row loss seed -- the dW GEMM no longer
// rescales its A operand in its k loop (that VALU work was 30 % of it, round-3 trace).
// Rounding matches the path it replaces (unit tensor rounded to bf16, times d, rounded again), so the weight gradients of
// W1 / W2 are bit-identical to mlps.hip + the scaling dW kernel (tests/test_gpu_split.py).
#include <cstddef>
#include <type_traits>
#include "mlp_panel.h"
#include "split.h"

namespace {
constexpr int NW = 16;
constexpr int KB1 = 64;                       // k elements per slab row (128 bytes)
constexpr int W_BYTES = HP * 128;             // 32 KB: one k-slab of a 256-row matrix
constexpr int NST = 4;
constexpr int PANEL_OFF = NST * W_BYTES;      // 128 KB
constexpr int SCR_OFF = PANEL_OFF + 2 * PANEL_HALF;   // per-row scalars: q, e, d, y (fp32 [32] each)
constexpr int COL_OFF = SCR_OFF + 512;        // column-sum partials: fp32 [2][4][256]
constexpr int LDS_TOTAL = COL_OFF + 8192;     // 152.5 KB
constexpr int OW = 8;                         // waves of the actor's 128-column output layer
constexpr int RW = BM / NW;                   // critic head rows per wave

__device__ __forceinline__ void dma_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "m0");
}

// acc += panel(k quarter q: columns 64 q .. 64 q + 63 of the 32 x 256 activation panel) * W(rows wrow0 + fr, 64 k)^T; operands
// swapped as in mlps.hip: acc[tm][0][r] = C[row 16 tm + fr][column wrow0 + 4 fg + r]
__device__ __forceinline__ void mma_panel(const unsigned char* panel, int q, const unsigned char* sb, f32x4 (&acc)[2][1], int wrow0, int fr, int fg,
                                          const bool two = true) {   // two = false: 16-row panel, only the first row block exists
  const int sw = (fr >> 1) & 7;
  const unsigned char* sa = panel + (q >> 1) * PANEL_HALF;
#pragma unroll
  for (int ks = 0; ks < KB1 / 32; ++ks) {
    const int posa = ((((q & 1) * 8) + ks * 4 + fg) ^ fr) * 16;
    const int posb = ((ks * 4 + fg) ^ sw) * 16;
    uint4 a[2], b;
    a[0] = *(const uint4*)(sa + fr * 256 + posa);
    if (two) a[1] = *(const uint4*)(sa + (16 + fr) * 256 + posa);
    b = *(const uint4*)(sb + (wrow0 + fr) * 128 + posb);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a[0]), acc[0][0], 0, 0, 0);
    if (two) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a[1]), acc[1][0], 0, 0, 0);
  }
}
typedef short v4s16 __attribute__((ext_vector_type(4)));

// The critics' layer 2 once ALL four W2 slabs have landed: the same eight 32-k steps in the same order per accumulator as four
// mma_panel calls, but the fragment reads of four steps are issued together, ahead of their MFMAs (round 6: the step-by-step form read,
// waited for the LDS latency and multiplied eight times in a row -- 1.7-2.1k cycles for 8-16 MFMAs; straight-line code, no branch
// inside: hipcc then counts lgkmcnt down instead of waiting for every read in front of every MFMA).
template <bool TWO>
__device__ __forceinline__ void mma_panel_all(const unsigned char* panel, const unsigned char* w, f32x4 (&acc)[2][1], int wrow0, int fr, int fg) {
  const int sw = (fr >> 1) & 7;
#pragma unroll
  for (int h = 0; h < 2; ++h) {                  // k steps 4 h .. 4 h + 3 = slabs 2 h, 2 h + 1
    uint4 a0[4], a1[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = 2 * h + (j >> 1), ks = j & 1;
      const unsigned char* sa = panel + (q >> 1) * PANEL_HALF;
      const int posa = ((((q & 1) * 8) + ks * 4 + fg) ^ fr) * 16;
      const int posb = ((ks * 4 + fg) ^ sw) * 16;
      a0[j] = *(const uint4*)(sa + fr * 256 + posa);
      if (TWO) a1[j] = *(const uint4*)(sa + (16 + fr) * 256 + posa);
      b[j] = *(const uint4*)(w + q * W_BYTES + (wrow0 + fr) * 128 + posb);
    }
    __builtin_amdgcn_sched_barrier(0);           // (without it the scheduler sinks every read to just in front of its first use)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, a0[j]), acc[0][0], 0, 0, 0);
      if (TWO) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, a1[j]), acc[1][0], 0, 0, 0);
    }
  }
}

// U = u2 W2 for the wave's 16 in-columns: k = the 256 out indices, W2's slabs k-strided (transpose reads); same batching
template <bool TWO>
__device__ __forceinline__ void mma_u_all(const unsigned char* panel, const unsigned char* wslab, int cpair, f32x4 (&dacc)[2], int fr, int fg) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint4 a0[4], a1[4];
    v4s16 b[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ks = 4 * h + j;
      const unsigned char* sa = panel + (ks >> 2) * PANEL_HALF;
      const int pos = ((((ks & 3) * 4) + fg) ^ fr) * 16;
      a0[j] = *(const uint4*)(sa + fr * 256 + pos);
      if (TWO) a1[j] = *(const uint4*)(sa + (16 + fr) * 256 + pos);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int k = ks * 32 + fg * 8 + half * 4 + (fr >> 2);
        b[j][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s16*)(wslab + k * 128 + (((cpair + ((fr & 3) >> 1)) ^ ((k >> 1) & 7)) * 16) + (fr & 1) * 8));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      struct { v4s16 lo, hi; } bv = {b[j][0], b[j][1]};
      dacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv), __builtin_bit_cast(bf16x8, a0[j]), dacc[0], 0, 0, 0);
      if (TWO) dacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv), __builtin_bit_cast(bf16x8, a1[j]), dacc[1], 0, 0, 0);
    }
  }
}

// element (row, column n) of the panel image as float
__device__ __forceinline__ float panel_at(const unsigned char* panel, int row, int n) {
  const bf16_t v = *(const bf16_t*)(panel + (n >> 7) * PANEL_HALF + row * 256 + ((((n & 127) >> 3) ^ (row & 15)) << 4) + (n & 7) * 2);
  return bf2f(v);
}
}  // namespace

#define MLPT_STAMP(i) do { if (trow) trow[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
__global__ __launch_bounds__(NW * 64) void mlp_tail_kernel(const TailBatch batch, unsigned long long* trace) {
  const TailProb& P = batch.p[blockIdx.y];
  // rows per workgroup: 32, or 16 for a learning critic with P.half_panels (round 6: every phase behind layer 2 is bound by the VALU
  // issue of the CU's 16 waves, and 64 learning panels of 32 rows leave half the machine idle -- 128 half panels halve each phase)
  const bool two = !P.half_panels;
  const int PR = two ? BM : BM / 2;
  const int m0 = blockIdx.x * PR;
  if (m0 >= P.rows) return;
  // (bit 0 of the trace pointer: stamps from lane 0 of EVERY wave -- [workgroup][16 waves][16] -- to see the skew the barriers absorb)
  const bool tr_waves = ((uintptr_t)trace & 1) != 0;
  unsigned long long* const tbase = (unsigned long long*)((uintptr_t)trace & ~(uintptr_t)1);
  unsigned long long* trow = nullptr;
  if (tbase) {
    const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (tr_waves) { if ((threadIdx.x & 63) == 0) trow = tbase + (wg * 16 + (threadIdx.x >> 6)) * 16; }
    else if (threadIdx.x == 0) trow = tbase + wg * 16;
  }
  asm volatile("" : "+v"(trow));
  MLPT_STAMP(0);
  // pull the kernel-argument cache lines of this workgroup's problem into the scalar cache NOW, all in flight together (a first
  // touch costs a scalar-cache miss of ~0.5 us and the fields are otherwise fetched one dependent batch after the other)
  unsigned touch = 0;
  {
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    const char __attribute__((address_space(4)))* pa = ka + blockIdx.y * sizeof(TailProb);
#pragma unroll
    for (int i = 0; i < (int)((sizeof(TailProb) + 63) / 64); ++i) asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"(i * 64));
    asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"((int)sizeof(TailProb) - 4));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(touch));
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const bool actor = P.kind == TAIL_ACTOR;
  const bool learn = P.kind == TAIL_CRITIC_LEARN;
  unsigned char* panel = lds + PANEL_OFF;
  float* qs = (float*)(lds + SCR_OFF);          // per row: Q, TD error e, loss seed d, TD target y
  float* es = qs + BM;
  float* ds = es + BM;
  float* ys = ds + BM;
  float* colp = (float*)(lds + COL_OFF);        // column-sum partials [2 sums][4 row chunks][256 columns]

  // ---- everything the epilogues and the head read from global memory is requested FIRST: these loads are then the oldest
  // entries of the wave's vector-memory queue, so the counted waits of the operand stream below cover them too, and their
  // first uses come after the stream has landed (a compiler-visible load issued AFTER a DMA makes hipcc wait vmcnt(0) --
  // for every DMA in flight -- at its first use)
  const int n0 = wave * 16 + fg * 4;            // this lane's four hidden columns
  f32x4 b2v[1];
  b2v[0] = (n0 + 3 < P.H) ? *(const f32x4*)(P.b2 + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
  const int no = (wave & (OW - 1)) * 16 + fg * 4;   // this lane's four output columns (actor, waves 0..7)
  float v3[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    v3[r] = actor ? (no + r < P.out_dim ? P.b3[no + r] : 0.f) : ((lane * 4 + r < P.H) ? P.w3row[lane * 4 + r] : 0.f);
  const float b3s = actor ? 0.f : P.b3[0];
  // (raw loads only up here: arithmetic on a loaded value would make the compiler wait for it before the stream is issued)
  float h_rew = 0.f, h_done = 0.f, h_tq0 = 0.f, h_tq1 = 0.f;
  if (learn && wave == 0) {
    const int mc = min(m0 + (lane & (PR - 1)), P.rows - 1);
    h_rew = P.reward[mc];
    h_done = P.done[mc];
    h_tq0 = P.tq[0][mc];
    h_tq1 = P.n_target > 1 ? P.tq[1][mc] : 0.f;
  }
  // the learning critic's w3 for this thread's u2 cell (columns n8 .. n8 + 7 of row lane & 31)
  // (32-row panel: row lane & 31, 32 cells of 8 columns per row over the 16 waves; 16-row panel: row lane & 15, waves 0..7)
  const int crow = two ? (lane & 31) : (lane & 15);
  const int n8 = two ? (2 * wave + (lane >> 5)) * 8 : ((4 * wave + (lane >> 4)) & 31) * 8;
  const bool cell_on = two || wave < NW / 2;
  float4 w3a = make_float4(0.f, 0.f, 0.f, 0.f), w3b = w3a;
  if (learn) {
    const int nb = min(n8, P.H - 8);
    w3a = *(const float4*)(P.w3row + nb);
    w3b = *(const float4*)(P.w3row + nb + 4);
  }
  // ... and for its column of the panel's column sums (dw3 / db2: thread = (row chunk, column)); raw load, used after the head
  const int ck = tid & 255, cc = tid >> 8;
  float w3c = 0.f;
  if (learn && P.dw3_part && ck < P.H) w3c = P.w3row[ck];
  int mrow0 = m0, mset = 0;                     // first row of the panel inside its batch, the batch's index
  if (P.rows_per_set > 0) { mset = m0 / P.rows_per_set; mrow0 = m0 - mset * P.rows_per_set; }
  // the device step counter through the SCALAR cache (hipcc turns a plain load of this uniform global into a vector load + an
  // immediate vmcnt(0)): requested here, waited for where the dropout key is hashed
  int32_t step_now = 0;
  if (P.mask_mode == RECNN_MASK_HASH && P.step_ptr) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(step_now) : "s"(P.step_ptr));

  // ---- request EVERYTHING the workgroup will multiply: the h1 panel (one instruction per wave: 4 rows x 256 bytes of one
  // k half; LDS position p of row r holds source chunk p ^ (r & 15)) and W2's four k-slabs (ring stages 0..3; rows l_row and
  // l_row + 128 per lane, chunk c of row r at position c ^ ((r >> 1) & 7)); the actor's two W3 slabs follow into stages 0 / 1
  // once W2's first two slabs have been multiplied
  {
    const int half = wave >> 3, r = (wave & 7) * 4 + (lane >> 4), pos = lane & 15;
    const int gr = min(m0 + r, P.rows - 1);
    const unsigned voff = (unsigned)((gr * (int)P.ldh + half * 128) * 2 + ((pos ^ (r & 15)) << 4));
    if (two || (wave & 7) < 4) dma_s(voff, P.h1, lds0 + PANEL_OFF + half * PANEL_HALF + (wave & 7) * 1024);
  }
  const int l_row = wave * 8 + (lane >> 3);
  const int l_c = ((lane & 7) ^ ((l_row >> 1) & 7)) * 16;
  const unsigned voff_sq = (unsigned)(l_row * (int)P.ldw2 * 2 + l_c);   // W2 / W3 share the pitch (mlpt_launch)
  const unsigned wave_kb = wave * 1024;
  auto issue_w2 = [&](int q) {
    const char* b0 = (const char*)P.W2 + q * (2 * KB1);
    const unsigned wb = lds0 + q * W_BYTES + wave_kb;
    dma_s(voff_sq, b0, wb);
    dma_s(voff_sq, b0 + 256 * P.ldw2, wb + NW * 1024);
  };
  auto issue_w3 = [&](int p) {     // k-slabs 2 p and 2 p + 1 of W3's 128 rows as image rows 0..127 / 128..255 of stage p
    const char* b0 = (const char*)P.W3 + 2 * p * (2 * KB1);
    const unsigned wb = lds0 + p * W_BYTES + wave_kb;
    dma_s(voff_sq, b0, wb);
    dma_s(voff_sq, b0 + 2 * KB1, wb + NW * 1024);
  };
#pragma unroll
  for (int q = 0; q < 4; ++q) issue_w2(q);

  // ------------------------------------------------------------------ layer 2, slab by slab as the stream lands
  // (per wave: 1 panel + 8 slab instructions in flight; slab q has landed once at most the 2 (3 - q) younger ones are outstanding)
  f32x4 acc[2][1];
  acc[0][0] = acc[1][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t gate1 = 0;   // relu/dropout gate of h1 for this lane's accumulator elements (bit tm * 4 + r): the backward's gate of U
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // A critic's whole stream (144 KB) arrives within ~1k cycles of its first slab (every wave issued its nine instructions
    // back to back), so waiting slab by slab buys nothing and costs three more rendezvous (trace: +0.9k cycles): ONE wait, one
    // barrier.  The actor keeps the per-slab rendezvous: W3's slabs take over the stages of W2's first two.
    if (actor || q == 0) {
      // actor: W3's slab p was issued after W2's slab p + 1 was waited for, so the younger set at slab q is {W2 q+1.., W3 ..}
      if (!actor) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (q == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();             // slab q (and the h1 panel) landed for every wave; everybody is done with slab q - 1
    }
    if (q == 0) MLPT_STAMP(1);
    if (actor && q >= 1 && q <= 2) issue_w3(q - 1);   // W2's slab q - 1 is done with: W3's slab q - 1 takes its stage
    if (q == 0 && learn) {
      // (row blocks as compile-time constants: a loop that BREAKS on a run-time flag is not unrolled, and indexing registers with its
      //  counter becomes select chains -- the U epilogue below spent 2-3k clocks that way until it was written like this)
      auto gate_block = [&](auto TMc) {
        constexpr int tm = decltype(TMc)::value;
        const int row = tm * 16 + fr;
        const uint2 hv = *(const uint2*)(panel + (n0 >> 7) * PANEL_HALF + row * 256 + ((((n0 & 127) >> 3) ^ (row & 15)) << 4) + (n0 & 7) * 2);
        if (hv.x & 0x7FFFu) gate1 |= 1u << (tm * 4 + 0);
        if (hv.x & 0x7FFF0000u) gate1 |= 1u << (tm * 4 + 1);
        if (hv.y & 0x7FFFu) gate1 |= 1u << (tm * 4 + 2);
        if (hv.y & 0x7FFF0000u) gate1 |= 1u << (tm * 4 + 3);
      };
      gate_block(std::integral_constant<int, 0>{});
      if (two) gate_block(std::integral_constant<int, 1>{});
    }
    if (actor) mma_panel(panel, q, lds + q * W_BYTES, acc, wave * 16, fr, fg, true);
  }
  if (!actor) {
    if (two) mma_panel_all<true>(panel, lds, acc, wave * 16, fr, fg);
    else mma_panel_all<false>(panel, lds, acc, wave * 16, fr, fg);
  }
  MLPT_STAMP(2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // everyone is done reading the h1 panel
  // (hidden_epilogue hashes the dropout word from the row's index INSIDE its batch: m0 -> mrow0)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(step_now));
  const uint32_t key2 = P.mask_mode == RECNN_MASK_HASH ? mask_key(P.seed, step_now + P.step_add + mset, P.stream2) : 0u;
  hidden_epilogue<1>(acc, b2v, P.H, P.rows - (m0 - mrow0), mrow0, wave, fr, fg, P.mask_mode,
                     P.mask2 ? P.mask2 + (int64_t)(m0 - mrow0) * P.ld_mask : nullptr, P.ld_mask, key2, panel, nullptr, two);

  if (actor) {
    // ---------------------------------------------------------------- layer 3 (actor): 32 x 128 outputs on waves 0..7
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // h2 panel complete, both W3 slabs landed
    f32x4 o[2][1];
    o[0][0] = o[1][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave < OW) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        mma_panel(panel, 2 * p, lds + p * W_BYTES, o, wave * 16, fr, fg);
        mma_panel(panel, 2 * p + 1, lds + p * W_BYTES, o, 128 + wave * 16, fr, fg);
      }
    }
    MLPT_STAMP(3);
    if (P.h2) panel_to_global<NW>(panel, (bf16_t*)P.h2, P.ldh, m0, P.rows, tid);
    if (wave < OW) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int m = m0 + tm * 16 + fr;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ncol = no + r < P.out_dim;
          v[r] = o[tm][0][r] + v3[r];
          if (P.addend && ncol && m < P.rows) {
            const float z = P.addend[(int64_t)m * P.ld_add + no + r];
            v[r] += fminf(fmaxf(z, -P.add_clip), P.add_clip);
          }
          if (!ncol) v[r] = 0.f;
        }
        uint2 packed = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        if (no + 3 >= P.out_dim) {                             // (padded columns hold bf16 +0, not -0)
          if (no + 0 >= P.out_dim) packed.x &= 0xFFFF0000u;
          if (no + 1 >= P.out_dim) packed.x &= 0x0000FFFFu;
          if (no + 2 >= P.out_dim) packed.y &= 0xFFFF0000u;
          if (no + 3 >= P.out_dim) packed.y &= 0x0000FFFFu;
        }
        if (m < P.rows) {
          if (no + 3 < P.out_dim) {
            *(uint2*)((bf16_t*)P.out + (int64_t)m * P.ldo + no) = packed;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (no + r < P.out_dim) ((bf16_t*)P.out)[(int64_t)m * P.ldo + no + r] = (bf16_t)((r < 2 ? packed.x : packed.y) >> ((r & 1) * 16));
          }
        }
      }
    }
    MLPT_STAMP(10);
    return;
  }

  // ---- the problem's fields this path reads from here on, ONCE, into scalar registers (round 6: read at their use sites each was an
  // s_load + s_waitcnt lgkmcnt(0) in the middle of a phase -- and lgkmcnt also counts the phase's LDS reads: 9 loads / 15 waits in the
  // column-sum phase alone)
  const int kH = P.H, kRows = P.rows, kNTarget = P.n_target;
  const float kScale = P.scale, kGamma = P.gamma, kLo = P.lo, kHi = P.hi;
  const int64_t kLdh = P.ldh;
  float* const kQ = P.q; float* const kDelta = P.delta_out; float* const kExpected = P.expected; float* const kTargetQ = P.target_q;
  float* const kLossPart = P.loss_part; float* const kDb3 = P.db3_part; float* const kDw3 = P.dw3_part; float* const kDb2 = P.db2_part;
  float* const kDb1 = P.db1_part;
  void* const kDz2 = P.dz2; void* const kDz1 = P.dz1; void* const kH2 = P.h2;
  // ------------------------------------------------------------------ critic head: q[m] = h2[m, :] . w3 + b3
  // The TD target needs nothing this workgroup computes (Q' came with the kernel's inputs): wave 0 evaluates it while the
  // h2 panel is being completed, so a row's TD error and loss seed are one subtraction away from its q dot.
  const float h_tq = kNTarget > 1 ? fminf(h_tq0, h_tq1) : h_tq0;
  if (learn && wave == 0 && lane < PR) {
    float y = h_rew + (1.0f - h_done) * kGamma * h_tq;
    y = fminf(fmaxf(y, kLo), kHi);
    ys[lane] = y;
  }
  MLPT_STAMP(3);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // h2 panel complete (and the TD targets are in LDS)
  if (kH2) panel_to_global<NW>(panel, (bf16_t*)kH2, kLdh, m0, min(kRows, m0 + PR), tid);
  // ---- everything that reads h2 happens HERE, in one burst of LDS reads behind one barrier (round 6; before: q dots -> barrier -> column
  // sums and u2 cells -> barrier, each phase a latency chain of its own): the q dots' operands, the 8 values of this thread's column for the
  // d-weighted column sums (kept in registers until d exists), and the thread's u2 cell, which needs no d at all -- u2 = w3 scale [h2 > 0]
  // and U = (u2 W2) scale gate(h1) are UNIT tensors, only their d multiples and the column sums wait for the head.
  // both rows of the wave side by side (two independent load -> dot -> reduction chains); Q and the loss seed reach global
  // memory after the rendezvous below, as two 128-byte stores, instead of one 4-byte store per row from here
  // (16-row panel: one row per wave)
  auto qdot_row = [&](const int row) -> float {
    const int c = ((((lane * 4) & 127) >> 3) ^ (row & 15));
    const uint2 hv = *(const uint2*)(panel + ((lane * 4) >> 7) * PANEL_HALF + row * 256 + c * 16 + ((lane * 4) & 7) * 2);
    const float hf[4] = {bf2f((bf16_t)(hv.x & 0xFFFFu)), bf2f((bf16_t)(hv.x >> 16)), bf2f((bf16_t)(hv.y & 0xFFFFu)), bf2f((bf16_t)(hv.y >> 16))};
    float sd = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) sd += lane * 4 + j < kH ? hf[j] * v3[j] : 0.f;
    return sd;
  };
  float sdot0 = qdot_row(two ? wave * RW : wave), sdot1 = 0.f;
  if (two) sdot1 = qdot_row(wave * RW + 1);
  float hc[8];                                  // h2[rows 8 cc .. 8 cc + 7][column ck]
  uint4 packed = make_uint4(0u, 0u, 0u, 0u);    // this thread's u2 cell (row lane & 31, columns n8 .. n8 + 7), rounded to bf16
  unsigned char* cell = panel + (n8 >> 7) * PANEL_HALF + crow * 256 + ((((n8 & 127) >> 3) ^ (crow & 15)) * 16);
  const bool col_on = two || cc < 2;            // the panel's row chunks of 8: four, or two
  if (learn && col_on) {
#pragma unroll
    for (int j = 0; j < 8; ++j) hc[j] = panel_at(panel, cc * 8 + j, ck);
  }
  if (learn && cell_on) {
    const uint4 raw = *(const uint4*)cell;
    const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
    const float wsc = n8 < kH ? kScale : 0.f;
    const float w3s[8] = {w3a.x * wsc, w3a.y * wsc, w3a.z * wsc, w3a.w * wsc, w3b.x * wsc, w3b.y * wsc, w3b.z * wsc, w3b.w * wsc};
    float uz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float hv = bf2f((bf16_t)((u[j >> 1] >> ((j & 1) * 16)) & 0xFFFF));
      uz[j] = hv > 0.f ? w3s[j] : 0.f;
    }
    packed = make_uint4(pack_bf2(uz[0], uz[1]), pack_bf2(uz[2], uz[3]), pack_bf2(uz[4], uz[5]), pack_bf2(uz[6], uz[7]));
  }
  sdot0 = wave_sum(sdot0);
  if (two) sdot1 = wave_sum(sdot1);
  if (lane == 0) {
    auto head_row = [&](const int row, const float sd) {
      const float qv = sd + b3s;
      const bool valid = m0 + row < kRows;
      qs[row] = qv;
      if (learn) {
        const float e = valid ? qv - ys[row] : 0.f;
        es[row] = e;
        ds[row] = e * (2.0f / (float)kRows);
      }
    };
    head_row(two ? wave * RW : wave, sdot0);
    if (two) head_row(wave * RW + 1, sdot1);
  }
  if (!learn) {
    __syncthreads();
    if (tid < PR && m0 + tid < kRows && kQ) kQ[m0 + tid] = qs[tid];
    MLPT_STAMP(4);
    return;
  }
  MLPT_STAMP(4);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // e and d of all 32 rows are in LDS; every thread has read the h2 values it needs
  MLPT_STAMP(5);
  if (cell_on) *(uint4*)cell = packed;          // the u2 panel (the A operand of the next product) takes h2's place
  if (wave == 1 && lane < PR && m0 + lane < kRows) {       // (wave 0 has the loss sums)
    if (kQ) kQ[m0 + lane] = qs[lane];
    if (kDelta) kDelta[m0 + lane] = ds[lane];
  }

  // ---- loss partial sums (wave 0, off everybody else's path): sum (q - y)^2 and sum d over the panel, lanes 0..31 = rows
  if (wave == 0) {
    const int r = lane & 31, m = m0 + r;
    const bool valid = lane < PR && m < kRows;
    const float e = lane < PR ? es[r] : 0.f;
    const float d = lane < PR ? ds[r] : 0.f;
    if (valid) {
      if (kExpected) kExpected[m] = ys[r];
      if (kTargetQ) kTargetQ[m] = h_tq;
    }
    const float tot = wave_sum(e * e);
    const float dsum = wave_sum(d);
    if (lane == 0) {
      if (kLossPart) kLossPart[blockIdx.x] = tot;
      if (kDb3) kDb3[blockIdx.x] = dsum;
    }
  }
  // ---- dz2 = d * u2 to global (the ROUNDED unit value times d)
  if (cell_on) {
    const int row = crow, m = m0 + row;
    const float d = ds[row];
    const uint32_t pu[4] = {packed.x, packed.y, packed.z, packed.w};
    float dz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dz[j] = bf2f((bf16_t)((pu[j >> 1] >> ((j & 1) * 16)) & 0xFFFF)) * d;
    if (m < kRows)
      *(uint4*)((bf16_t*)kDz2 + (int64_t)m * kLdh + n8) = make_uint4(pack_bf2(dz[0], dz[1]), pack_bf2(dz[2], dz[3]), pack_bf2(dz[4], dz[5]), pack_bf2(dz[6], dz[7]));
  }
  // ---- column sums over the panel's rows, four row chunks of 8 per column (thread = (chunk, column)), fma chains upwards:
  //   dw3[k] = sum_r d_r h2[r][k]         db2[k] = sum_r d_r u2[r][k],  u2 = bf16(w3[k] scale) where h2 > 0
  if (kDw3 && ck < kH && col_on) {
    const float u = bf2f(f2bf(w3c * kScale));
    float s3 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = ds[cc * 8 + j];
      s3 = fmaf(d, hc[j], s3);
      s2 = fmaf(d, hc[j] > 0.f ? u : 0.f, s2);
    }
    colp[cc * 256 + ck] = s3;
    colp[1024 + cc * 256 + ck] = s2;
  }
  MLPT_STAMP(6);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // u2 panel complete, the column partials are in LDS
  MLPT_STAMP(7);
  if (kDw3 && tid < kH) {
    // (16-row panel: the sum of ITS two chunks; the consumer adds the two halves of a 32-row panel first -- optim_dev.h slab_grads, pair)
    kDw3[(int64_t)blockIdx.x * kH + tid] = two ? (colp[tid] + colp[256 + tid]) + (colp[512 + tid] + colp[768 + tid]) : colp[tid] + colp[256 + tid];
    kDb2[(int64_t)blockIdx.x * kH + tid] = two ? (colp[1024 + tid] + colp[1280 + tid]) + (colp[1536 + tid] + colp[1792 + tid]) : colp[1024 + tid] + colp[1280 + tid];
  }

  // ---- U = (u2 W2) * scale * gate(h1): W2's four k-slabs are still in stages 0..3: k-slab q holds in-columns 64 q .. 64 q + 63,
  // rows = out index = the contraction index here, chunk c of row r at c ^ ((r >> 1) & 7); B fragments by transpose reads
  f32x4 dacc[2];
  dacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  dacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const unsigned char* wslab = lds + (wave >> 2) * W_BYTES;     // in-columns 16 wave .. 16 wave + 15
    const int cpair = (wave & 3) * 2;
    if (two) mma_u_all<true>(panel, wslab, cpair, dacc, fr, fg);
    else mma_u_all<false>(panel, wslab, cpair, dacc, fr, fg);
  }
  // (operands swapped: dacc[tm][r] = U[row 16 tm + fr][column 16 wave + 4 fg + r], the layout of gate1)
  MLPT_STAMP(8);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // every wave is done with the u2 panel (MFMA A operand): U takes its place
  {
    unsigned char* col = panel + (n0 >> 7) * PANEL_HALF + (n0 & 7) * 2;
    const int c = (n0 & 127) >> 3;
    const float uscale = kScale;
    const int Hc = kH, nrows = kRows;
    bf16_t* const dz1 = (bf16_t*)kDz1;
    const int64_t ldh = kLdh;
    auto u_block = [&](auto TMc) {
      constexpr int tm = decltype(TMc)::value;
      const int row = tm * 16 + fr, mm = m0 + row;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (((gate1 >> (tm * 4 + r)) & 1u) && n0 + r < Hc) ? dacc[tm][r] * uscale : 0.f;
      const uint32_t lo = pack_bf2(v[0], v[1]), hi = pack_bf2(v[2], v[3]);          // unit U, rounded to bf16
      *(uint2*)(col + row * 256 + ((c ^ fr) << 4)) = make_uint2(lo, hi);             // panel image (db1 sums)
      const float d = ds[row];
      const float z[4] = {bf2f((bf16_t)(lo & 0xFFFFu)) * d, bf2f((bf16_t)(lo >> 16)) * d, bf2f((bf16_t)(hi & 0xFFFFu)) * d, bf2f((bf16_t)(hi >> 16)) * d};
      if (mm < nrows) *(uint2*)(dz1 + (int64_t)mm * ldh + n0) = make_uint2(pack_bf2(z[0], z[1]), pack_bf2(z[2], z[3]));
    };
    u_block(std::integral_constant<int, 0>{});
    if (two) u_block(std::integral_constant<int, 1>{});
  }
  MLPT_STAMP(9);
  if (kDb1) {                             // db1[k] = sum_r d_r U[r][k], same four-chunk order
    __syncthreads();                            // U panel complete
    if (ck < kH && col_on) {
      float s1 = 0.f;
#pragma unroll
      for (int r = cc * 8; r < cc * 8 + 8; ++r) s1 = fmaf(ds[r], panel_at(panel, r, ck), s1);
      colp[cc * 256 + ck] = s1;
    }
    __syncthreads();
    if (tid < kH) kDb1[(int64_t)blockIdx.x * kH + tid] = two ? (colp[tid] + colp[256 + tid]) + (colp[512 + tid] + colp[768 + tid]) : colp[tid] + colp[256 + tid];
  }
  MLPT_STAMP(10);
}

static unsigned long long* g_mlpt_trace = nullptr;
extern "C" void recnn_debug_tail_trace(void* p) { g_mlpt_trace = (unsigned long long*)p; }   // [workgroup][16] uint64 shader-clock stamps

// ---- the critic head alone: q[m] = h2[m, :] . w3 + b3, one wave per row -- the q-dot loop of mlp_tail_kernel, same lane ->
// column map (lane owns columns 4 lane .. 4 lane + 3), same products, same wave_sum tree
__global__ __launch_bounds__(NW * 64) void qdot_kernel(const bf16_t* __restrict__ h2, int64_t ldh, const float* __restrict__ w3row,
                                                       const float* __restrict__ b3, int H, int rows, float* __restrict__ q) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * NW + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v3[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v3[r] = (lane * 4 + r < H) ? w3row[lane * 4 + r] : 0.f;
  const float b3s = b3[0];
  const uint2 hv = *(const uint2*)(h2 + (int64_t)row * ldh + lane * 4);
  const float hf[4] = {bf2f((bf16_t)(hv.x & 0xFFFFu)), bf2f((bf16_t)(hv.x >> 16)), bf2f((bf16_t)(hv.y & 0xFFFFu)), bf2f((bf16_t)(hv.y >> 16))};
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += lane * 4 + j < H ? hf[j] * v3[j] : 0.f;
  s = wave_sum(s);
  if (lane == 0) q[row] = s + b3s;
}

int qdot_launch(const void* h2, int64_t ldh, const float* w3row, const float* b3, int H, int rows, float* q, hipStream_t s) {
  RECNN_REQUIRE(h2 && w3row && b3 && q && rows > 0 && H > 0 && H <= 256 && ldh >= 256 && ldh % 4 == 0, "qdot: bad arguments");
  hipLaunchKernelGGL(qdot_kernel, dim3((rows + NW - 1) / NW), dim3(NW * 64), 0, s, (const bf16_t*)h2, ldh, w3row, b3, H, rows, q);
  return recnn_check_hip(hipGetLastError(), "qdot_kernel");
}

int mlpt_init() {
  return recnn_check_hip(hipFuncSetAttribute((const void*)mlp_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL), "mlp_tail attr");
}

int mlpt_launch(const TailBatch& b, int nprob, hipStream_t s) {
  RECNN_REQUIRE(nprob >= 1 && nprob <= TAIL_MAX_GROUP, "mlp_tail: 1..%d problems per launch", TAIL_MAX_GROUP);
  int panels = 0;
  for (int i = 0; i < nprob; ++i) {
    const TailProb& p = b.p[i];
    const int pr = p.half_panels ? BM / 2 : BM;
    if ((p.rows + pr - 1) / pr > panels) panels = (p.rows + pr - 1) / pr;
    RECNN_REQUIRE(!p.half_panels || (p.kind == TAIL_CRITIC_LEARN && p.rows % BM == 0), "mlp_tail: 16-row panels are the learning critic's, whole 32-row pairs only");
    RECNN_REQUIRE(p.rows > 0 && p.H >= 8 && p.H <= HP && (p.H & 7) == 0 && p.out_dim <= 128, "mlp_tail: hidden <= 256 (multiple of 8), out_dim <= 128");
    RECNN_REQUIRE(p.h1 && p.W2 && p.b2 && p.b3 && p.ldh % 8 == 0 && p.ldw2 % 8 == 0 && (((uintptr_t)p.h1 | (uintptr_t)p.W2) & 15) == 0, "mlp_tail: bad operands");
    RECNN_REQUIRE(p.rows_per_set == 0 || p.rows_per_set % BM == 0, "mlp_tail: batches must be multiples of %d rows", BM);
    if (p.kind == TAIL_ACTOR) {
      RECNN_REQUIRE(p.W3 && p.out && p.ldw3 == p.ldw2 && (((uintptr_t)p.W3) & 15) == 0, "mlp_tail: actor needs W3 (same pitch as W2) and an output");
    } else {
      RECNN_REQUIRE(p.w3row && (((uintptr_t)p.w3row) & 15) == 0, "mlp_tail: critic needs its last layer's row");
      if (p.kind == TAIL_CRITIC_LEARN)
        RECNN_REQUIRE(p.n_target >= 1 && p.n_target <= 2 && p.tq[0] && (p.n_target < 2 || p.tq[1]) && p.reward && p.done && p.dz2 && p.dz1,
                      "mlp_tail: the learning critic needs Q', reward, done and its backward buffers");
    }
  }
  hipLaunchKernelGGL(mlp_tail_kernel, dim3(panels, nprob), dim3(NW * 64), LDS_TOTAL, s, b, g_mlpt_trace);
  return recnn_check_hip(hipGetLastError(), "mlp_tail_kernel");
}
