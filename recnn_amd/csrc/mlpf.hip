// mlpf.hip -- a whole FROZEN network (target actor, target critic, the actor between its optimizer steps) on 128-row panels
// of a policy cycle's batches: all three layers in one workgroup, weights streamed once per 128 rows.  bf16, gfx950.
//
//   Actor   h1 = drop(relu(x W1^T + b1)); h2 = drop(relu(h1 W2^T + b2)); out = h2 W3^T + b3 (+ clip(noise))   recnn/nn/models.py:66-73
//   Critic  h1 = relu([state | action] W1^T + b1); h2 = relu(h1 W2^T + b2); q = h2 . w3 + b3               recnn/nn/models.py:207-213
//
// Cycle mode (engine.hip) applies these networks to M = cycle x rows rows at once.  Doing that with the per-step kernels (a
// tiled layer-1 GEMM, then 32-row tail panels that each restart a 128-192 KB weight stream) spent as long on the 16 % of
// the FLOPs after layer 1 as on layer 1 itself.  Here a 16-wave workgroup owns 128 rows x all 256 hidden columns (waves as
// 4 row groups x 4 column groups, wave tile 32 x 64: six fragment reads per eight MFMAs):
//   layer 1   64-k slabs of A (16 KB) + W1 (32 KB) through a 3-slot ring over the whole 144 KB, two slabs ahead;
//   then      the activation panel (4 sub-panels of 32 rows in mlp_panel.h's swizzled layout: 64 KB) takes the ring's first
//             64 KB, W2's four and W3's two 32 KB slabs stream through a 3-slot ring in the other 96 KB;
//   critic    q dots from the h2 panel (8 rows per wave).
// Per output element the arithmetic is the per-step kernels' (l1gemm.hip + mlpt.hip = mlps.hip): k ascending in MFMA steps of
// 32 from a zero accumulator, segment 0 before segment 1, weights as the first MFMA operand, the same epilogues
// (mlp_panel.h), the same q-dot lane map and reduction tree -- so `run()` in cycle mode equals the eager step loop bit for bit.
#include <cstddef>
#include "mlp_panel.h"
#include "split.h"

namespace {
constexpr int NW = 16;
constexpr int W_BYTES = HP * 128;                // 32 KB: one 64-k slab of a 256-row weight matrix
constexpr int NST1 = 3;
constexpr int LDS_TOTAL = 160 * 1024;
// FR = rows per workgroup: 128 (waves as 4 row groups x 4 column groups, wave tile 32 x 64), or 64 (2 x 8, wave tile 32 x 32) for the
// launches of SHORT cycle segments (round 6): a launch lasts as long as one workgroup does (one workgroup per CU, <= 256 of them), and a
// request that starts or ends inside a policy cycle -- the driver's 20 steps are segments of 6 + 10 + 4 -- pays two launches per
// segment whatever its length; half the rows per workgroup is 0.7 of the time (tools/frozen_trace.py: 48.5k against 69.6k clocks) while
// the launch still fits one round.
template <int FR> struct FrozenPlan {
  static constexpr int WMG = FR / 32, WNG = NW / WMG;            // row groups x column groups of the 16 waves
  static constexpr int TNH = HP / WNG / 16;                      // 16-column blocks per wave in the hidden layers: 4 | 2
  static constexpr int TN3 = 128 / WNG / 16;                     // ... in the actor's output layer: 2 | 1
  static constexpr int A_BYTES = FR * 128;                       // one 64-k slab of the rows: 16 | 8 KB
  static constexpr int STAGE1 = A_BYTES + W_BYTES;               // 48 | 40 KB
  static constexpr int PANEL_BYTES = WMG * 2 * PANEL_HALF;       // 32-row sub-panels: 64 | 32 KB
  static constexpr int WR_OFF = PANEL_BYTES;                     // the later layers' ring: 3 x 32 KB behind the panel
  static constexpr int CONST_OFF = NST1 * STAGE1;                // 144 | 120 KB: b1 | b2 | b3 or w3 (fp32, 1 KB each) while layer 1 runs --
  static constexpr int A_WAVES = FR / 8;                         //   inside the later ring's third stage in both plans
  static_assert(CONST_OFF + 3 * 1024 <= LDS_TOTAL && PANEL_BYTES + 3 * W_BYTES <= LDS_TOTAL, "the two LDS plans share one allocation");
  static_assert(CONST_OFF >= WR_OFF + 2 * W_BYTES, "the constants must survive the later layers' first two slabs");
};

__device__ __forceinline__ void dma_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "m0");
}
}  // namespace

template <int FR>
__global__ __launch_bounds__(NW * 64) void mlp_frozen_kernel(const FrozenBatch batch) {
  using PL = FrozenPlan<FR>;
  constexpr int WNG = PL::WNG, TNH = PL::TNH, TN3 = PL::TN3, A_BYTES = PL::A_BYTES, STAGE1 = PL::STAGE1, WR_OFF = PL::WR_OFF, CONST_OFF = PL::CONST_OFF;
  const FrozenProb& P = batch.p[blockIdx.y];
  const int m0 = blockIdx.x * FR;
  if (m0 >= P.rows) return;
  // pull this problem's kernel-argument lines into the scalar cache, all in flight together
  unsigned touch = 0;
  {
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    const char __attribute__((address_space(4)))* pa = ka + blockIdx.y * sizeof(FrozenProb);
#pragma unroll
    for (int i = 0; i < (int)((sizeof(FrozenProb) + 63) / 64); ++i) asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"(i * 64));
    asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"((int)sizeof(FrozenProb) - 4));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(touch));
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  // (debug: phase stamps by thread 0 -- tools/frozen_trace.py)
  unsigned long long* const trw = batch.trace ? batch.trace + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 : nullptr;
  auto stamp = [&](int k) { if (trw && tid == 0) trw[k] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNG, wn = wave % WNG;    // rows 32 wm .. + 31, hidden columns 16 TNH wn .. + 16 TNH - 1
  const int fr = lane & 15, fg = lane >> 4;
  const bool actor = P.W3 != nullptr;
  const int sw = (fr >> 1) & 7;

  // ---- the biases (and a critic's last-layer row) through the stream itself: three 1 KB transfers (waves 0..2) into the 16 KB
  // that layer 1's ring leaves free; the epilogues read them from LDS.  (Held in registers across the k loop they cost 28
  // VGPRs the kernel does not have -- the first version spilled; fetched by ordinary loads after layer 1 they cost a memory
  // latency on every workgroup's critical path.)
  if (wave < 3) {
    const float* src = wave == 0 ? P.b1 : (wave == 1 ? P.b2 : (actor ? P.b3 : P.w3row));
    const int nvalid = wave < 2 ? P.H : (actor ? P.out_dim : P.H);
    dma_s((unsigned)(min(lane * 4, nvalid - 4) * 4), src, lds0 + CONST_OFF + wave * 1024);
  }
  float b3s = 0.f;                                // critic: b3 through the scalar cache
  if (!actor) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(b3s) : "s"(P.b3));
  int32_t step_now = 0;
  if (P.mask_mode == RECNN_MASK_HASH && P.step_ptr) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(step_now) : "s"(P.step_ptr));

  // ------------------------------------------------------------------ layer 1
  // slab = A rows (one instruction per wave: rows 8 wave .. + 7) + 256 W1 rows (two per wave: rows l_row, l_row + 128); chunk c
  // of slab row r at position c ^ ((r >> 1) & 7)
  const int nt0 = P.K[0] / 64;
  const int nt = nt0 + (P.nseg > 1 ? P.K[1] / 64 : 0);
  const int l_row = wave * 8 + (lane >> 3);
  const int l_c = ((lane & 7) ^ ((l_row >> 1) & 7)) * 16;
  const int gr_a = min(m0 + l_row, P.rows - 1);
  unsigned voff_a = (unsigned)(gr_a * (int)P.lda[0] * 2 + l_c);
  unsigned voff_w = (unsigned)(l_row * (int)P.ldw1 * 2 + l_c);
  const char* a_base = (const char*)P.A[0];
  const char* w_base = (const char*)P.W1 + (int64_t)P.w1_col[0] * 2;
  const int64_t w_half = (int64_t)128 * P.ldw1 * 2;
  const unsigned wave_kb = wave * 1024;
  int issued = 0;
  auto issue_l1 = [&]() {
    if (issued == nt0) {                          // second contraction segment
      voff_a = (unsigned)(gr_a * (int)P.lda[1] * 2 + l_c);
      a_base = (const char*)P.A[1];
      w_base = (const char*)P.W1 + (int64_t)P.w1_col[1] * 2;
    }
    const unsigned sb = lds0 + (issued % NST1) * STAGE1 + wave_kb;
    ++issued;
    if (FR == 128 || wave < PL::A_WAVES) dma_s(voff_a, a_base, sb);      // (64 rows: one instruction each from waves 0..7)
    dma_s(voff_w, w_base, sb + A_BYTES);
    dma_s(voff_w, w_base + w_half, sb + A_BYTES + NW * 1024);
    a_base += 128;
    w_base += 128;
  };
  issue_l1();
  issue_l1();                                     // (layer 1 has at least 2 slabs)

  f32x4 acc[2][TNH];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int t = 0; t < nt; ++t) {
    if (t + 1 >= nt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (FR == 128 || wave < PL::A_WAVES) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // (waves 8..15 of the 64-row plan issue two instructions per slab)
    __builtin_amdgcn_s_barrier();                 // slab t landed for every wave; the stage of slab t - 1 (= slab t + 2's) is free
    if (t == 0) stamp(1);
    if (t == 8) stamp(2);
    if (t == 16) stamp(3);
    if (t + 2 < nt) issue_l1();
    const unsigned char* sa = lds + (t % NST1) * STAGE1;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int pos = ((ks * 4 + fg) ^ sw) * 16;
      uint4 a[2], b[TNH];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (wm * 32 + tm * 16 + fr) * 128 + pos);
#pragma unroll
      for (int tn = 0; tn < TNH; ++tn) b[tn] = *(const uint4*)(sb + (wn * (16 * TNH) + tn * 16 + fr) * 128 + pos);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TNH; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[tn]), __builtin_bit_cast(bf16x8, a[tm]), acc[tm][tn], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // every wave is done with layer 1's ring: the second LDS plan takes over
  stamp(4);

  // ---- the later layers' stream: slab c = W2's k-slab c (c < 4), then (actor) W3's two double slabs; ring stage c % 3
  const unsigned voff_sq = (unsigned)(l_row * (int)P.ldw2 * 2 + l_c);   // W2 / W3 share the pitch (mlpf_launch)
  const int npost = actor ? 6 : 4;
  int posted = 0;
  auto issue_post = [&]() {
    if (posted >= npost) return;
    const int c = posted++;
    const unsigned wb = lds0 + WR_OFF + (c % 3) * W_BYTES + wave_kb;
    if (c < 4) {
      const char* b0 = (const char*)P.W2 + c * 128;
      dma_s(voff_sq, b0, wb);
      dma_s(voff_sq, b0 + 256 * P.ldw2, wb + NW * 1024);
    } else {                                      // W3: k-slabs 2 p and 2 p + 1 of its 128 rows as image rows 0..127 / 128..255
      const char* b0 = (const char*)P.W3 + 2 * (c - 4) * 128;
      dma_s(voff_sq, b0, wb);
      dma_s(voff_sq, b0 + 128, wb + NW * 1024);
    }
  };
  issue_post();
  issue_post();
  int consumed = 0;
  // next slab of the later layers: waits until it has landed for every wave (2 instructions per wave and slab: at most one
  // younger slab outstanding), refills the stage the previous slab occupied
  auto next_post = [&]() -> const unsigned char* {
    const int c = consumed++;
    if (posted - c - 1 >= 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_post();
    return lds + WR_OFF + (c % 3) * W_BYTES;
  };

  // ---- h1 = dropout(relu(acc + b1)) into this wave's sub-panel (rows 32 wm ..), columns 64 wn ..
  unsigned char* spanel = lds + wm * (2 * PANEL_HALF);
  const int sm0 = m0 + wm * 32;                   // first row of the sub-panel
  int mset = 0, mrow0 = sm0;
  if (P.rows_per_set > 0) { mset = sm0 / P.rows_per_set; mrow0 = sm0 - mset * P.rows_per_set; }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(step_now));
  const int32_t step0 = step_now + P.step_add + mset;
  const uint32_t key1 = P.mask_mode == RECNN_MASK_HASH ? mask_key(P.seed, step0, P.stream1) : 0u;
  const uint32_t key2 = P.mask_mode == RECNN_MASK_HASH ? mask_key(P.seed, step0, P.stream2) : 0u;
  const float* cst = (const float*)(lds + CONST_OFF);
  {
    f32x4 b1v[TNH];
#pragma unroll
    for (int tn = 0; tn < TNH; ++tn) {
      const int n = wn * (16 * TNH) + tn * 16 + fg * 4;
      b1v[tn] = (n + 3 < P.H) ? *(const f32x4*)(cst + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    hidden_epilogue<TNH>(acc, b1v, P.H, P.rows - (sm0 - mrow0), mrow0, wn, fr, fg, P.mask_mode, nullptr, 0, key1, spanel);
  }
  stamp(5);
  // the later layers' constants (after the epilogue: the accumulators' registers are free again)
  f32x4 b2v[TNH];
#pragma unroll
  for (int tn = 0; tn < TNH; ++tn) {
    const int n = wn * (16 * TNH) + tn * 16 + fg * 4;
    b2v[tn] = (n + 3 < P.H) ? *(const f32x4*)(cst + 256 + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float v3[TN3][4];                               // actor: b3 of this lane's output columns 16 TN3 wn + 16 tn + 4 fg + r
  float w3v[4];                                   // critic: w3 of columns 4 lane .. 4 lane + 3
#pragma unroll
  for (int tn = 0; tn < TN3; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int no = wn * (16 * TN3) + tn * 16 + fg * 4 + r;
      v3[tn][r] = (actor && no < P.out_dim) ? cst[512 + no] : 0.f;
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) w3v[r] = (!actor && lane * 4 + r < P.H) ? cst[512 + lane * 4 + r] : 0.f;
  // (the constants sit in what becomes the later layers' ring stage 2: every wave has read them -- the lgkmcnt(0) of
  // next_post -- before the first rendezvous below lets slab 2 be requested)

  // ------------------------------------------------------------------ layer 2
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < 4; ++q) {
    const unsigned char* st = next_post();        // (its barrier also completes the h1 panel for q = 0)
    if (q == 0 && P.h1) {
#pragma unroll
      for (int sp = 0; sp < PL::WMG; ++sp) panel_to_global<NW>(lds + sp * (2 * PANEL_HALF), (bf16_t*)P.h1, P.ldh, m0 + sp * 32, P.rows, tid);
    }
    const unsigned char* sa = spanel + (q >> 1) * PANEL_HALF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int posa = ((((q & 1) * 8) + ks * 4 + fg) ^ fr) * 16;
      const int posb = ((ks * 4 + fg) ^ sw) * 16;
      uint4 a[2], b[TNH];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * 256 + posa);
#pragma unroll
      for (int tn = 0; tn < TNH; ++tn) b[tn] = *(const uint4*)(st + (wn * (16 * TNH) + tn * 16 + fr) * 128 + posb);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TNH; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[tn]), __builtin_bit_cast(bf16x8, a[tm]), acc[tm][tn], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // everyone is done reading the h1 panel
  stamp(6);
  hidden_epilogue<TNH>(acc, b2v, P.H, P.rows - (sm0 - mrow0), mrow0, wn, fr, fg, P.mask_mode, nullptr, 0, key2, spanel);
  stamp(7);

  if (actor) {
    // ---------------------------------------------------------------- layer 3: 128 x 128 outputs, wave tile 32 x 32
    f32x4 o[2][TN3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN3; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < 2; ++p) {
      const unsigned char* st = next_post();      // (completes the h2 panel for p = 0)
      if (p == 0 && P.h2) {
#pragma unroll
        for (int sp = 0; sp < PL::WMG; ++sp) panel_to_global<NW>(lds + sp * (2 * PANEL_HALF), (bf16_t*)P.h2, P.ldh, m0 + sp * 32, P.rows, tid);
      }
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {            // k quarter q = 2 p + hq of the panel against image rows hq * 128 + output column
        const int q = 2 * p + hq;
        const unsigned char* sa = spanel + (q >> 1) * PANEL_HALF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int posa = ((((q & 1) * 8) + ks * 4 + fg) ^ fr) * 16;
          const int posb = ((ks * 4 + fg) ^ sw) * 16;
          uint4 a[2], b[TN3];
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * 256 + posa);
#pragma unroll
          for (int tn = 0; tn < TN3; ++tn) b[tn] = *(const uint4*)(st + (hq * 128 + wn * (16 * TN3) + tn * 16 + fr) * 128 + posb);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN3; ++tn)
              o[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[tn]), __builtin_bit_cast(bf16x8, a[tm]), o[tm][tn], 0, 0, 0);
        }
      }
    }
    stamp(8);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = sm0 + tm * 16 + fr;
#pragma unroll
      for (int tn = 0; tn < TN3; ++tn) {
        const int no = wn * (16 * TN3) + tn * 16 + fg * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ncol = no + r < P.out_dim;
          v[r] = o[tm][tn][r] + v3[tn][r];
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
    stamp(9);
    return;
  }

  // ------------------------------------------------------------------ critic head: q[m] = h2[m, :] . w3 + b3, 8 rows per wave
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b3s) : : "memory");
  __builtin_amdgcn_s_barrier();                   // h2 panel complete
  if (P.h2) {
#pragma unroll
    for (int sp = 0; sp < PL::WMG; ++sp) panel_to_global<NW>(lds + sp * (2 * PANEL_HALF), (bf16_t*)P.h2, P.ldh, m0 + sp * 32, P.rows, tid);
  }
  for (int i = 0; i < FR / NW; ++i) {
    const int prow = wave * (FR / NW) + i;        // row of the 128-row panel
    const unsigned char* pp = lds + (prow >> 5) * (2 * PANEL_HALF);
    const int row = prow & 31;
    const int c = ((((lane * 4) & 127) >> 3) ^ (row & 15));
    const uint2 hv = *(const uint2*)(pp + ((lane * 4) >> 7) * PANEL_HALF + row * 256 + c * 16 + ((lane * 4) & 7) * 2);
    const float hf[4] = {bf2f((bf16_t)(hv.x & 0xFFFFu)), bf2f((bf16_t)(hv.x >> 16)), bf2f((bf16_t)(hv.y & 0xFFFFu)), bf2f((bf16_t)(hv.y >> 16))};
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += lane * 4 + j < P.H ? hf[j] * w3v[j] : 0.f;
    s = wave_sum(s);
    if (lane == 0 && m0 + prow < P.rows && P.q) P.q[m0 + prow] = s + b3s;
  }
  stamp(9);
}
static unsigned long long* g_frozen_trace = nullptr;
extern "C" void recnn_debug_frozen_trace(void* p) { g_frozen_trace = (unsigned long long*)p; }   // read at launch (= graph capture) time

int mlpf_init() {
  int rc = recnn_check_hip(hipFuncSetAttribute((const void*)mlp_frozen_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL), "mlp_frozen attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)mlp_frozen_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL), "mlp_frozen<64> attr");
  return rc;
}

int mlpf_launch(const FrozenBatch& b, int nprob, hipStream_t s, int panel_rows) {
  RECNN_REQUIRE(nprob >= 1 && nprob <= FROZEN_MAX_GROUP, "mlp_frozen: 1..%d problems per launch", FROZEN_MAX_GROUP);
  RECNN_REQUIRE(panel_rows == 128 || panel_rows == 64, "mlp_frozen: 128 or 64 rows per workgroup");
  int rows = 0;
  for (int i = 0; i < nprob; ++i) {
    const FrozenProb& p = b.p[i];
    if (p.rows > rows) rows = p.rows;
    RECNN_REQUIRE(p.rows > 0 && p.H >= 8 && p.H <= HP && (p.H & 7) == 0 && p.out_dim <= 128, "mlp_frozen: hidden <= 256 (multiple of 8), out_dim <= 128");
    RECNN_REQUIRE(p.nseg >= 1 && p.nseg <= 2 && p.W1 && p.W2 && p.b1 && p.b2 && p.b3, "mlp_frozen: bad problem");
    int k64 = 0;
    for (int g = 0; g < p.nseg; ++g) {
      RECNN_REQUIRE(p.K[g] > 0 && p.K[g] % 64 == 0 && p.lda[g] % 8 == 0 && (((uintptr_t)p.A[g]) & 15) == 0 && (p.w1_col[g] & 7) == 0 &&
                        (int64_t)p.rows * p.lda[g] * 2 < (1ll << 31),
                    "mlp_frozen: segment %d must be 16-byte aligned with K a multiple of 64 (and < 2 GB)", g);
      k64 += p.K[g] / 64;
    }
    RECNN_REQUIRE(k64 >= 2, "mlp_frozen: layer 1 needs at least 128 k");
    RECNN_REQUIRE(p.ldw1 % 8 == 0 && p.ldw2 % 8 == 0 && (((uintptr_t)p.W1 | (uintptr_t)p.W2) & 15) == 0, "mlp_frozen: bad weight pitches");
    RECNN_REQUIRE(p.mask_mode == RECNN_MASK_NONE || p.mask_mode == RECNN_MASK_HASH, "mlp_frozen: hash dropout masks only");
    RECNN_REQUIRE(p.rows_per_set == 0 || p.rows_per_set % 32 == 0, "mlp_frozen: batches must be multiples of 32 rows");
    if (p.W3) RECNN_REQUIRE(p.out && p.ldw3 == p.ldw2 && (((uintptr_t)p.W3) & 15) == 0 && p.ldo % 4 == 0, "mlp_frozen: actor needs W3 (same pitch as W2) and an output");
    else RECNN_REQUIRE(p.w3row && p.q, "mlp_frozen: critic needs its last layer's row and a Q output");
  }
  FrozenBatch bb = b;
  bb.trace = g_frozen_trace;
  if (panel_rows == 64) hipLaunchKernelGGL(mlp_frozen_kernel<64>, dim3((rows + 63) / 64, nprob), dim3(NW * 64), LDS_TOTAL, s, bb);
  else hipLaunchKernelGGL(mlp_frozen_kernel<128>, dim3((rows + 127) / 128, nprob), dim3(NW * 64), LDS_TOTAL, s, bb);
  return recnn_check_hip(hipGetLastError(), "mlp_frozen_kernel");
}
