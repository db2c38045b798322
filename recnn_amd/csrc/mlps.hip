// mlps.hip -- the fused row-panel forward of a whole Actor / Critic MLP as ONE continuous weight stream (bf16, gfx950).
//
// One 16-wave workgroup owns 32 batch rows of one
// network application and all 256 hidden columns; recnn/nn/models.py:66-73 / :207-213 (cat + 3 x (addmm, relu, dropout)),
// the chained target critics, the TD head and the critic's layer-2 backward tail (mlp.h).
//
// What differs is the SCHEDULE.  In-kernel traces of mlp.hip (tools/mlp_trace.py, profiles/NOTES_r01_r05.md 5b) showed a workgroup spending
// 7.6 us of its 30 in the layer-1 k loop and 18 us in the phases after it: every later phase issued its weights (64-128 KB)
// as a burst when it began and then waited for the whole burst (0.6-1.2 us each, six of them on the target actor's chain),
// although the weights of ALL phases are known at launch and a CU streams 102-111 GB/s back to back but only 70 GB/s as
// burst + wait (tools/dma_bw.hip).  Here every operand a workgroup will ever multiply -- layer-1 A and W1 slabs, W2, W3,
// the chained critics' layer-1 parts, action columns and W2 -- is ONE sequence of 36 KB slabs pushed through a 4-stage
// LDS ring by global_load_lds, always three slabs ahead of the consumer; the phases (k loops, epilogues, heads) just
// consume the next slab.  One barrier per slab; epilogues run while the ring keeps filling.
//
// LDS: 4 stages x (4 KB A part + 32 KB W part) + 16 KB activation panel = 160 KB.  A slab row is 128 bytes (64 k); 16-byte
// chunk c of row r sits at position c ^ ((r >> 1) & 7) (applied on the DMA source address): conflict-free ds_read_b128.
#include <cstddef>
#include "mlp.h"
#include "mlp_panel.h"

namespace {
constexpr int NW = 16;
constexpr int KB1 = 64;                       // k elements per slab row (128 bytes)
constexpr int A1_BYTES = BM * 128;            // 4 KB
constexpr int W1_BYTES = HP * 128;            // 32 KB
constexpr int STAGE1 = A1_BYTES + W1_BYTES;   // 36 KB
constexpr int NST = 4;
constexpr int PANEL_OFF = NST * STAGE1;       // 144 KB
constexpr int LDS_TOTAL = PANEL_OFF + 2 * PANEL_HALF;  // 160 KB

constexpr int OW = 8;                         // waves of the actor's 128-column output layer
constexpr int RW = BM / NW;                   // critic head rows per wave

// One LDS-DMA instruction: 64 lanes x 16 bytes from (uniform base + per-lane byte offset) to LDS at lds_dst + 16 lane.
// M0 is written directly (the kernel uses no other M0 consumer: no movrel, no GWS, no LDS-direct loads); no "memory"
// clobber: the ordering points are the waits and barriers of the consumer, and the argument block stays in registers.
__device__ __forceinline__ void dma_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "m0");
}

// wait until at most `n` of this wave's vector-memory operations are outstanding (n is wave-uniform, 0 .. 2 (NIW + 1))
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
  }
}

// acc += A(32 x 64 k) * W(rows wrow0 + 16 tn + fr, 64 k)^T, both operands from a ring stage (rows of 128 bytes).  Operands
// swapped as in mlp.hip: acc[tm][tn][r] = C[row 16 tm + fr][column .. + 4 fg + r].
__device__ __forceinline__ void mma_stage(const unsigned char* sa, const unsigned char* sb, f32x4 (&acc)[2][1], int wrow0, int fr, int fg) {
  const int sw = (fr >> 1) & 7;
#pragma unroll
  for (int ks = 0; ks < KB1 / 32; ++ks) {
    const int pos = ((ks * 4 + fg) ^ sw) * 16;
    uint4 a[2], b;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * 128 + pos);
    b = *(const uint4*)(sb + (wrow0 + fr) * 128 + pos);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
      acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a[tm]), acc[tm][0], 0, 0, 0);
  }
}
// the same with A = k quarter q (columns 64 q .. 64 q + 63) of the activation panel (rows of 256 bytes per 128-column half,
// chunk c of row r at c ^ (r & 15))
__device__ __forceinline__ void mma_panel(const unsigned char* panel, int q, const unsigned char* sb, f32x4 (&acc)[2][1], int wrow0, int fr,
                                          int fg) {
  const int sw = (fr >> 1) & 7;
  const unsigned char* sa = panel + (q >> 1) * PANEL_HALF;
#pragma unroll
  for (int ks = 0; ks < KB1 / 32; ++ks) {
    const int posa = ((((q & 1) * 8) + ks * 4 + fg) ^ fr) * 16;
    const int posb = ((ks * 4 + fg) ^ sw) * 16;
    uint4 a[2], b;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * 256 + posa);
    b = *(const uint4*)(sb + (wrow0 + fr) * 128 + posb);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
      acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a[tm]), acc[tm][0], 0, 0, 0);
  }
}
typedef short v4s16 __attribute__((ext_vector_type(4)));
}  // namespace

// (the row pointer is computed ONCE, before any DMA is in flight: gridDim.x is a load from the dispatch packet, and the compiler's
// vmcnt(0) for it inside a stamp drained the whole DMA queue of wave 0 at every stamp -- the first traces of this kernel charged
// that drain to whatever phase a stamp followed)
#define MLPS_STAMP(i) do { if (trow) trow[(i)] = __builtin_amdgcn_s_memtime(); } while (0)

// PROBE (timing experiments, recnn_debug_mlp_probe): 1 = no MFMA work, 2 = no DMA
template <int PROBE>
__global__ __launch_bounds__(NW * 64) void mlps_fwd_kernel(const MlpBatch batch, unsigned long long* trace) {
  // (2-D grid (panel, problem): workgroup ids go round the 8 XCDs, so every XCD meets every network and fetches all weights into
  // its own L2 -- 8 copies, the 2x over-fetch of the PMC numbers.  An XCD-affine map (a network's panels on one XCD pair) was
  // measured in round 3: 54.4 -> 43.5 MB of HBM traffic per launch, 26.1 -> 32.2 us: two L2s feeding a network's 64 workgroups
  // are slower than eight; removed.)
  const int by = __builtin_amdgcn_readfirstlane((int)blockIdx.y), bx = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  const int panels = gridDim.x;
  const MlpProb& P = batch.p[by];
  const int m0 = bx * BM;
  if (m0 >= P.rows) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int row_max = P.rows - 1;
  unsigned long long* trow = (trace && threadIdx.x == 0) ? trace + ((int64_t)by * panels + bx) * 32 : nullptr;
  asm volatile("" : "+v"(trow));
  MLPS_STAMP(0);

  // ---- pull every kernel-argument cache line this workgroup will read into the scalar cache NOW.  The argument block is
  // 3.7 KB; a first touch of one of its 64-byte lines costs a scalar-cache miss (in-kernel trace: 1.1-1.4k ticks, ~0.6 us, on
  // the first DMA issue of W3 / of a chained critic's weights, against 260 for a touched line) -- a dozen of those sat on
  // the dependent chain after layer 1.  One dword per line, all in flight together, waited for once (below, where the
  // workgroup waits for its first slabs anyway).
  unsigned touch = 0;   // (one SGPR, threaded through every load so that it stays allocated until the wait)
  {
    // (through the kernarg segment pointer: taking the address of the by-value argument would make the compiler copy it to scratch)
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    const char __attribute__((address_space(4)))* pa = ka + by * sizeof(MlpProb);       // batch.p[by]
    const char __attribute__((address_space(4)))* pb = ka + offsetof(MlpBatch, tail);            // tails, head, critic tails, err
#pragma unroll
    for (int i = 0; i < 5; ++i) asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"(i * 64));
    asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"(300));   // (the block is not 64-byte aligned)
#pragma unroll
    for (int i = 0; i < 7; ++i) asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pb), "n"(i * 64));
  }

  // ------------------------------------------------------------------ the slab sequence of this workgroup
  //   [layer 1: nt slabs (A + W1)] [W2: 4] ( actor: [W3: 2 (two k slabs of 128 rows per stage)]
  //                                          per chained critic: [part: 1 (fp32 32 x 256)] [W1 action columns: 2] [W2: 4] )
  const int nt0 = P.K[0] / KB1;
  const int nt = nt0 + (P.nseg > 1 ? P.K[1] / KB1 : 0);
  const int n_tail = P.W3 ? P.n_tail : 0;

  // per-lane DMA geometry: one wave instruction moves 8 rows x 128 B; instruction i of a slab covers image rows 8 i .. 8 i + 7.
  // Addresses are (uniform 64-bit base) + (per-lane 32-bit offset): the saddr form of global_load_lds keeps the address
  // arithmetic of a slab to one VALU add per stream (16 waves issue every instruction of this bookkeeping: at one SALU / VALU
  // issue slot per SIMD and 4 cycles, an instruction per wave costs 16 cycles of the CU -- the phases after layer 1 were
  // bound by exactly that, not by bytes or flops).
  const int q_row = lane >> 3, q_pos = lane & 7;
  const int l_row = wave * 8 + q_row;                          // image row of this wave's first instruction (+ 128 for the second)
  const int l_c = (q_pos ^ ((l_row >> 1) & 7)) * 16;           // source chunk behind LDS position q_pos (same for row + 128)
  const bool a_wave = wave * 8 < BM;                           // waves 0..3 also carry the layer-1 A panel
  const int gr_a = min(m0 + l_row, row_max);
  unsigned voff_a = (unsigned)(gr_a * (int)P.lda[0] * 2 + l_c);                       // into A[0] (then A[1])
  unsigned voff_w = (unsigned)((l_row * (int)P.ldw1 + P.w1_col[0]) * 2 + l_c);         // into W1, rows l_row / l_row + 128
  const unsigned voff_a1 = (unsigned)(gr_a * (int)P.lda[P.nseg > 1 ? 1 : 0] * 2 + l_c);
  const unsigned voff_w1 = (unsigned)((l_row * (int)P.ldw1 + (P.nseg > 1 ? P.w1_col[1] : 0)) * 2 + l_c);
  const unsigned voff_sq = (unsigned)(l_row * (int)P.ldw2 * 2 + l_c);                  // into W2 / W3 / a chained critic's W2 (same pitch: mlp_launch)
  const char* a_base = (const char*)P.A[0];
  const char* w_base0 = (const char*)P.W1;
  const char* w_base1 = w_base0 + (int64_t)128 * P.ldw1 * 2;
  const unsigned wave_kb = wave * 1024;

  int issued = 0;     // slabs issued so far (stream index of the next one)
  // layer-1 slab: A rows (waves 0..3) + 256 W1 rows; the lane offsets run along k
  auto issue_l1 = [&]() {
    const int i = issued++;
    if (i == nt0) { voff_w = voff_w1; voff_a = voff_a1; a_base = (const char*)P.A[P.nseg > 1 ? 1 : 0]; }   // second contraction segment
    const unsigned sb = lds0 + (i & (NST - 1)) * STAGE1 + wave_kb;
    if constexpr (!(PROBE & 2)) {
      if (a_wave) dma_s(voff_a, a_base, sb);
      dma_s(voff_w, w_base0, sb + A1_BYTES);
      dma_s(voff_w, w_base1, sb + A1_BYTES + NW * 1024);
    }
    voff_a += 2 * KB1;
    voff_w += 2 * KB1;
  };
  // k-slab q of a bf16 matrix [256 rows, ld]: rows l_row and l_row + 128 of this lane (voff = lane offset into k-slab 0)
  auto issue_mat = [&](const void* base, int64_t ld, unsigned voff, int q) {
    const unsigned wb = lds0 + (issued++ & (NST - 1)) * STAGE1 + A1_BYTES + wave_kb;
    const char* b0 = (const char*)base + q * (2 * KB1);
    if constexpr (!(PROBE & 2)) {
      dma_s(voff, b0, wb);
      dma_s(voff, b0 + 256 * ld, wb + NW * 1024);
    }
  };
  // the producers' parts must be complete before their DMA is issued: one wave-0 acquire per chained critic, taken when its
  // part slab comes up for issue (three slabs before it is consumed: the producers -- lower workgroup ids, layer 1 only --
  // finished long before).  Bounded spin: a broken launch order becomes a REPORTED error (batch.err), not a hang.
  auto acquire_part = [&](int ti) {
    if (tid == 0) {
      const int limit = batch.spin_limit > 0 ? batch.spin_limit : (1 << 22);
      int spins = 0;
      bool ok;
      while (!(ok = __hip_atomic_load(batch.tail[ti].flag + bx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0) && ++spins < limit)
        __builtin_amdgcn_s_sleep(2);
      if (!ok && batch.err) __hip_atomic_fetch_or(batch.err, MLP_ERR_PART_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(batch.tail[ti].flag + bx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_barrier();  // (raw: the other waves keep their DMAs in flight; only wave 0 paid the acquire's drain)
  };
  // slab k of the sequence AFTER layer 1:  0..3 W2 | actor: 4, 5 W3 (k-slabs 2 p and 2 p + 1 as image rows 0..127 / 128..255)
  //                                        | chained critic ti: 6 + 7 ti part, + 1, + 2 action columns of W1, + 3 .. + 6 its W2
  const int npost = P.part_out ? 0 : 4 + (P.W3 ? 2 + 7 * n_tail : 0);
  auto issue_post = [&](int k) {
    if (k >= npost) return;
    if (k < 4) { issue_mat(P.W2, P.ldw2, voff_sq, k); return; }
    if (k < 6) {
      const unsigned wb = lds0 + (issued++ & (NST - 1)) * STAGE1 + A1_BYTES + wave_kb;
      const char* b0 = (const char*)P.W3 + 2 * (k - 4) * (2 * KB1);
      if constexpr (!(PROBE & 2)) {
        dma_s(voff_sq, b0, wb);
        dma_s(voff_sq, b0 + 2 * KB1, wb + NW * 1024);
      }
      return;
    }
    const int ti = k >= 13 ? 1 : 0, r = k - 6 - 7 * ti;
    const MlpTail& T = batch.tail[ti];
    if (r == 0) {                                              // the producer's fp32 layer-1 part of these 32 rows, row i = 1 KB
      acquire_part(ti);
      const unsigned sb = lds0 + (issued++ & (NST - 1)) * STAGE1 + wave_kb;
      const char* src = (const char*)(T.part + (int64_t)m0 * HP) + wave_kb;
      if constexpr (!(PROBE & 2)) {
        dma_s(lane * 16, src, sb + A1_BYTES);
        dma_s(lane * 16, src + NW * 1024, sb + A1_BYTES + NW * 1024);
        // ... and the critic's b1 | b2 | w3 (fp32, one KB each) into the idle A part of the same stage: the epilogues and the
        // q dots read them from LDS (kept in registers from the start they cost 24 VGPRs the kernel does not have).  One
        // extra DMA on waves 0..2: their vmcnt waits get one instruction stricter while this slab is among the younger ones.
        if (wave < 3) dma_s(min(lane * 4, P.H - 4) * 4, wave == 0 ? T.b1 : (wave == 1 ? T.b2 : T.w3row), sb);
      }
    } else if (r < 3) {
      issue_mat(T.W1a, T.ldw1, (unsigned)(l_row * (int)T.ldw1 * 2 + l_c), r - 1);
    } else {
      issue_mat(T.W2, T.ldw2, voff_sq, r - 3);
    }
  };

  // A parts after layer 1: the chained critics' constants land in the A parts of their part slabs' stages, ps0 and ps0 + 3
  // (bytes 0 .. 3071); the action slabs take the A parts of the two other stages, the Q' scratch the last KB of stage ps0's.
  // (Issuing the part later than three slabs ahead -- to take the producers' acquire off the early chain -- was tried with
  // placeholder transfers in its slot: 25.3 -> 26.8 us, the late transfer is exposed; kept as is.)
  const int ps0 = (nt + 6) & (NST - 1);
  int consumed = 0;
  // The next slab of the sequence after layer 1: waits until it has landed for every wave (every slab after layer 1 is two
  // DMA instructions per wave, so "at most 2 y outstanding" leaves the y younger slabs in flight; global stores issued
  // meanwhile only make the wait stricter), refills the stage the previous slab occupied, returns the slab's stage.
  auto next_post = [&](int stamp = -1) -> const unsigned char* {
    const int c = consumed++;
    const int y = issued - c - 1;
    if (y >= 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if (y == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (stamp >= 0) MLPS_STAMP(stamp);
    __builtin_amdgcn_s_barrier();  // slab c landed for every wave; everybody is done with slab c - 1 and with its LDS writes so far
    if (stamp >= 0) MLPS_STAMP(stamp + 1);
    issue_post(issued - nt);
    return lds + (c & (NST - 1)) * STAGE1;
  };

  f32x4 acc[2][1];
  acc[0][0] = acc[1][0] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- start the stream, THEN fetch everything the epilogues and heads will read from global memory: the compiler waits for
  // such a load with vmcnt(0), which also drains the DMAs in flight (it cannot count the asm DMAs) -- harmless here, where the
  // first slabs have to land anyway, and the reason why no compiler-visible load may sit inside the stream
  issue_l1();
  issue_l1();
  issue_l1();                                                  // (layer 1 has at least 3 slabs: K >= 192)

  const int n0 = wave * 16 + fg * 4;                           // this lane's four hidden columns n0 .. n0 + 3
  f32x4 b1v[1], b2v[1];                                        // (hidden width is a multiple of 4 here: mlp_launch)
  {
    const bool in = n0 + 3 < P.H && !P.part_out;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    b1v[0] = in ? *(const f32x4*)(P.b1 + n0) : z4;
    b2v[0] = in ? *(const f32x4*)(P.b2 + n0) : z4;
  }
  // layer-3 operands: the actor's four output-column biases, or a critic's w3 entries of columns 4 lane .. 4 lane + 3
  const int no = (wave & (OW - 1)) * 16 + fg * 4;              // this lane's four output columns (actor, waves 0..7)
  float v3[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    v3[r] = P.W3 ? (no + r < P.out_dim ? P.b3[no + r] : 0.f) : ((P.q && lane * 4 + r < P.H) ? P.w3row[lane * 4 + r] : 0.f);
  // (wave-uniform values -- the critics' b3, the dropout step -- come through the scalar cache: no vmcnt involved)
  const float b3s = (!P.W3 && P.q) ? P.b3[0] : 0.f;
  float tb3[MLP_MAX_TAIL];
#pragma unroll
  for (int ti = 0; ti < MLP_MAX_TAIL; ++ti) tb3[ti] = ti < n_tail ? batch.tail[ti].b3[0] : 0.f;
  uint32_t key1 = 0, key2 = 0;
  if (P.mask_mode == RECNN_MASK_HASH && !P.part_out) {
    const int32_t st = (P.step_ptr ? *P.step_ptr : 0) + P.step_add;
    key1 = mask_key(P.seed, st, P.stream1);
    key2 = mask_key(P.seed, st, P.stream2);
  }
  // head inputs of rows m0 .. m0 + 31 (wave 0 evaluates the head)
  float h_rew = 0.f, h_done = 0.f, h_tq = 0.f;
  const bool self_head = batch.head.self_tq[0] != nullptr && P.cbwd_idx >= 0 && !P.W3;   // this critic evaluates its own head
  if (batch.head.n_critic > 0 && (n_tail > 0 || self_head) && wave == 0) {
    const int mc = min(m0 + (lane & 31), P.rows - 1);
    h_rew = batch.head.reward[mc];
    h_done = batch.head.done[mc];
    if (self_head) {
      h_tq = batch.head.self_tq[0][mc];
      if (batch.head.n_target > 1) h_tq = fminf(h_tq, batch.head.self_tq[1][mc]);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(touch));
  asm volatile("" : "+v"(b1v[0]), "+v"(b2v[0]), "+v"(h_rew), "+v"(h_done), "+v"(h_tq));
#pragma unroll
  for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(v3[r]));
  MLPS_STAMP(1);

  // ------------------------------------------------------------------ layer 1
  // slab t has landed for this wave once only the DMAs of the two younger slabs are outstanding (3 instructions per slab on
  // the A waves, 2 on the others); the last three iterations start the stream of the later layers (W2 k-slabs 0..2)
  for (int t = 0; t < nt - 3; ++t) {
    if (a_wave) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // slab t landed for every wave; the stage of slab t - 1 (= slab t + 3's) is free
    issue_l1();
    const unsigned char* st = lds + (consumed++ & (NST - 1)) * STAGE1;
    if constexpr (!(PROBE & 1)) mma_stage(st, st + A1_BYTES, acc, wave * 16, fr, fg);
  }
#pragma unroll
  for (int u = 0; u < 3; ++u) {                                // t = nt - 3 + u: younger slabs = (2 - u) of layer 1 + min(u, npost) later ones
    const int later = min(u, npost);
    const int pend = (2 - u) * (a_wave ? 3 : 2) + 2 * later;
    wait_vm(pend);
    __builtin_amdgcn_s_barrier();
    issue_post(u);
    const unsigned char* st = lds + (consumed++ & (NST - 1)) * STAGE1;
    if constexpr (!(PROBE & 1)) mma_stage(st, st + A1_BYTES, acc, wave * 16, fr, fg);
  }
  MLPS_STAMP(2);
  if (P.part_out) {
    // producer of a chained critic: hand the raw pre-activation part to the consumer workgroup of this panel
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) *(f32x4*)(P.part_out + (int64_t)(m0 + tm * 16 + fr) * HP + n0) = acc[tm][0];
    __syncthreads();  // every thread's stores have completed (the barrier is preceded by s_waitcnt vmcnt(0))
    if (tid == 0 && (batch.fault & 3) != 1) __hip_atomic_store(P.part_flag + bx, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    MLPS_STAMP(9);
    return;
  }
  unsigned char* panel = lds + PANEL_OFF;
  uint32_t gate1 = 0;  // relu/dropout gate of h1 for this lane's accumulator elements (bit tm*4 + r)
  hidden_epilogue<1>(acc, b1v, P.H, P.rows, m0, wave, fr, fg, P.mask_mode, P.mask1, P.ld_mask, key1, panel, &gate1);
  MLPS_STAMP(3);

  // ------------------------------------------------------------------ layer 2
  acc[0][0] = acc[1][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int w2_first = consumed;                               // stream index of W2's k-slab 0 (= nt)
  for (int q = 0; q < 4; ++q) {
    const unsigned char* st = next_post(q == 1 ? 22 : -1);     // (its barrier also completes the h1 panel for q = 0)
    MLPS_STAMP(10 + q);
    if (q == 0 && P.h1) panel_to_global<NW>(panel, (bf16_t*)P.h1, P.ldh, m0, P.rows, tid);
    if (q == 0) MLPS_STAMP(24);
    if constexpr (!(PROBE & 1)) mma_panel(panel, q, st + A1_BYTES, acc, wave * 16, fr, fg);
    if (q == 0) MLPS_STAMP(25);
  }
  MLPS_STAMP(4);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // everyone is done reading the h1 panel
  hidden_epilogue<1>(acc, b2v, P.H, P.rows, m0, wave, fr, fg, P.mask_mode, P.mask2, P.ld_mask, key2, panel);
  MLPS_STAMP(5);

  if (P.W3) {
    // ---------------------------------------------------------------- layer 3 (actor): 32 x 128 outputs on waves 0..7
    f32x4 o[2][1];
    o[0][0] = o[1][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < 2; ++p) {
      const unsigned char* st = next_post();                   // (completes the h2 panel for p = 0)
      MLPS_STAMP(14 + p);
      if (p == 0 && P.h2) panel_to_global<NW>(panel, (bf16_t*)P.h2, P.ldh, m0, P.rows, tid);
      if (wave < OW) {
        if constexpr (!(PROBE & 1)) mma_panel(panel, 2 * p, st + A1_BYTES, o, wave * 16, fr, fg);
        if constexpr (!(PROBE & 1)) mma_panel(panel, 2 * p + 1, st + A1_BYTES, o, 128 + wave * 16, fr, fg);
      }
    }
    MLPS_STAMP(16);
    if (wave < OW) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int row = tm * 16 + fr, m = m0 + row;
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
        // the chained critics read the action as two 64-k A slabs: quarter q in the (idle since layer 1) A part of stage ps0 + 1 + q
        if (n_tail)
          *(uint2*)(lds + ((ps0 + 1 + (no >> 6)) & (NST - 1)) * STAGE1 + row * 128 + (((((no & 63) >> 3)) ^ ((fr >> 1) & 7)) << 4) + (no & 7) * 2) = packed;
      }
    }
    MLPS_STAMP(6);
    // ---------------------------------------------------------------- chained critics (target critic on the new action)
    float* qscratch = (float*)(lds + ps0 * STAGE1 + 3072);     // last KB of an A part the constants never touch
#pragma unroll
    for (int ti = 0; ti < MLP_MAX_TAIL; ++ti) {
      if (ti >= n_tail) break;
      const MlpTail& T = batch.tail[ti];
      const unsigned char* tc = next_post();                   // the producer's layer-1 part (fp32 rows of 1 KB) + b1 | b2 | w3
      if (ti == 0) MLPS_STAMP(17);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) acc[tm][0] = *(const f32x4*)(tc + A1_BYTES + (tm * 16 + fr) * 1024 + n0 * 4);
      f32x4 tbv[1];
      for (int q = 0; q < 2; ++q) {                            // + action x W1[:, action columns]
        const unsigned char* st = next_post();                 // (the action slabs are complete after the part slab's barrier)
        if constexpr (!(PROBE & 1)) mma_stage(lds + ((ps0 + 1 + q) & (NST - 1)) * STAGE1, st + A1_BYTES, acc, wave * 16, fr, fg);
      }
      if (ti == 0) MLPS_STAMP(18);
      if (ti > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the previous critic's q dots are done with the panel
      }
      tbv[0] = *(const f32x4*)(tc + n0 * 4);
      hidden_epilogue<1>(acc, tbv, P.H, P.rows, m0, wave, fr, fg, RECNN_MASK_NONE, nullptr, 0, 0u, panel);
      if (ti == 0) MLPS_STAMP(19);
      acc[0][0] = acc[1][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < 4; ++q) {
        const unsigned char* st = next_post();
        if constexpr (!(PROBE & 1)) mma_panel(panel, q, st + A1_BYTES, acc, wave * 16, fr, fg);
      }
      if (ti == 0) MLPS_STAMP(20);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everyone is done reading the h1 panel
      tbv[0] = *(const f32x4*)(tc + 1024 + n0 * 4);
      hidden_epilogue<1>(acc, tbv, P.H, P.rows, m0, wave, fr, fg, RECNN_MASK_NONE, nullptr, 0, 0u, panel);
      if (ti == 0) MLPS_STAMP(21);
      const f32x4 tw = *(const f32x4*)(tc + 2048 + lane * 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // h2 panel complete
      for (int i = 0; i < RW; ++i) {
        const int row = wave * RW + i;
        const int c = ((((lane * 4) & 127) >> 3) ^ (row & 15));
        const uint2 hv = *(const uint2*)(panel + ((lane * 4) >> 7) * PANEL_HALF + row * 256 + c * 16 + ((lane * 4) & 7) * 2);
        const float hf[4] = {bf2f((bf16_t)(hv.x & 0xFFFFu)), bf2f((bf16_t)(hv.x >> 16)), bf2f((bf16_t)(hv.y & 0xFFFFu)), bf2f((bf16_t)(hv.y >> 16))};
        float sdot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sdot += lane * 4 + j < P.H ? hf[j] * tw[j] : 0.f;
        sdot = wave_sum(sdot);
        if (lane == 0) {
          const float qv = sdot + tb3[ti];
          if (m0 + row < P.rows) T.q[m0 + row] = qv;
          qscratch[ti * BM + row] = qv;
        }
      }
    }
    MLPS_STAMP(7);
    // ---------------------------------------------------------------- head of the learning critic(s)
    if (batch.head.n_critic > 0 && n_tail > 0) {
      const MlpHead& Hd = batch.head;
      __syncthreads();   // Q' of all 32 rows (every tail) is in LDS
      if (wave == 0) {
        const int r = lane & 31, m = m0 + r;
        const bool valid = lane < 32 && m < P.rows;
        float tqv = qscratch[r];
        if (n_tail > 1) tqv = fminf(tqv, qscratch[BM + r]);
        float y = h_rew + (1.0f - h_done) * Hd.gamma * tqv;
        y = fminf(fmaxf(y, Hd.lo), Hd.hi);
        if (valid) {
          if (Hd.expected) Hd.expected[m] = y;
          if (Hd.target_q) Hd.target_q[m] = tqv;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (c < Hd.n_critic) {
            // Q(s, a) from the critic workgroup of the same rows: value-as-flag slot, bounded spin, slot put back to rest
            float q = 0.f;
            if (valid) {
              uint32_t* slot = (uint32_t*)Hd.q_slot[c] + m;
              uint32_t bits = MLP_TQ_EMPTY;
              int spins = 0;
              const int limit = batch.spin_limit > 0 ? batch.spin_limit : (1 << 22);
              while ((bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == MLP_TQ_EMPTY && ++spins < limit)
                __builtin_amdgcn_s_sleep(1);
              if (bits == MLP_TQ_EMPTY && batch.err) __hip_atomic_fetch_or(batch.err, MLP_ERR_Q_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(slot, MLP_TQ_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              q = __builtin_bit_cast(float, bits);
            }
            const float e = valid ? q - y : 0.f;
            const float d = e * (2.0f / (float)P.rows);
            if (valid && Hd.delta_out[c]) Hd.delta_out[c][m] = d;
            const float tot = wave_sum(e * e);
            const float dsum = wave_sum(d);
            if (lane == 0) {
              if (Hd.loss_part[c]) Hd.loss_part[c][bx] = tot;
              if (Hd.db3_part[c]) Hd.db3_part[c][bx] = dsum;
            }
          }
        }
      }
    }
  } else if (P.q) {
    // ---------------------------------------------------------------- critic head: q[m] = h2[m, :] . w3 + b3
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // h2 panel complete; all four W2 k-slabs are (still) in the ring
    if (P.h2) panel_to_global<NW>(panel, (bf16_t*)P.h2, P.ldh, m0, P.rows, tid);
    for (int i = 0; i < RW; ++i) {
      const int row = wave * RW + i;
      const int c = ((((lane * 4) & 127) >> 3) ^ (row & 15));
      const uint2 hv = *(const uint2*)(panel + ((lane * 4) >> 7) * PANEL_HALF + row * 256 + c * 16 + ((lane * 4) & 7) * 2);
      const float hf[4] = {bf2f((bf16_t)(hv.x & 0xFFFFu)), bf2f((bf16_t)(hv.x >> 16)), bf2f((bf16_t)(hv.y & 0xFFFFu)), bf2f((bf16_t)(hv.y >> 16))};
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += lane * 4 + j < P.H ? hf[j] * v3[j] : 0.f;
      s = wave_sum(s);
      if (lane == 0 && self_head) ((float*)(lds + ps0 * STAGE1 + 3072))[row] = s + b3s;   // (an A part: idle since layer 1)
      if (lane == 0 && m0 + row < P.rows) {
        const float qv = s + b3s;
        P.q[m0 + row] = qv;
        if (P.cbwd_idx >= 0 && batch.cbwd[P.cbwd_idx].q_slot && (batch.fault & 3) != 2)   // hand Q(s, a) to the workgroup that evaluates the head
          __hip_atomic_store((uint32_t*)batch.cbwd[P.cbwd_idx].q_slot + m0 + row, __builtin_bit_cast(uint32_t, qv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    MLPS_STAMP(6);
    if (P.cbwd_idx >= 0) {
      const MlpCriticBwd& B = batch.cbwd[P.cbwd_idx];
      // ---- u2 = w3 * scale * [h2 > 0], in place in the panel (it becomes the A operand) and to global
      {
        const int row = lane & 31, m = m0 + row;
        const int n8 = (2 * wave + (lane >> 5)) * 8;
        const int nb = min(n8, P.H - 8);
        const float4 w3a = *(const float4*)(P.w3row + nb), w3b = *(const float4*)(P.w3row + nb + 4);
        const float wsc = n8 < P.H ? B.scale : 0.f;
        const float w3s[8] = {w3a.x * wsc, w3a.y * wsc, w3a.z * wsc, w3a.w * wsc, w3b.x * wsc, w3b.y * wsc, w3b.z * wsc, w3b.w * wsc};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the Q values of a self-evaluated head are in LDS behind this barrier)
        __builtin_amdgcn_s_barrier();   // every wave is done reading h2 rows for its q dots
        unsigned char* cell = panel + (n8 >> 7) * PANEL_HALF + row * 256 + ((((n8 & 127) >> 3) ^ (row & 15)) * 16);
        const uint4 raw = *(const uint4*)cell;
        const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
        float uz[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float hv = bf2f((bf16_t)((u[j >> 1] >> ((j & 1) * 16)) & 0xFFFF));
          uz[j] = hv > 0.f ? w3s[j] : 0.f;
        }
        const uint4 packed = make_uint4(pack_bf2(uz[0], uz[1]), pack_bf2(uz[2], uz[3]), pack_bf2(uz[4], uz[5]), pack_bf2(uz[6], uz[7]));
        *(uint4*)cell = packed;
        if (m < P.rows) *(uint4*)((bf16_t*)B.dz2 + (int64_t)m * P.ldh + n8) = packed;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // u2 panel complete
      // ---- U = (u2 W2) * scale * gate(h1): the four W2 k-slabs are the LAST slabs of this workgroup's stream, so they are
      // all still in the ring: k-slab q (in-columns 64 q .. 64 q + 63) in stage (w2_first + q) & 3, rows = out index = k here,
      // chunk c of row r at c ^ ((r >> 1) & 7); B fragments by transpose reads
      f32x4 dacc[2];
      dacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      dacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* wslab = lds + ((w2_first + (wave >> 2)) & (NST - 1)) * STAGE1 + A1_BYTES;   // in-columns 16 wave .. +15
      const int cpair = (wave & 3) * 2;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const unsigned char* sa = panel + (ks >> 2) * PANEL_HALF;
        const int pos = ((((ks & 3) * 4) + fg) ^ fr) * 16;
        uint4 a[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * 256 + pos);
        v4s16 b[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int k = ks * 32 + fg * 8 + half * 4 + (fr >> 2);
          b[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4s16*)(wslab + k * 128 + (((cpair + ((fr & 3) >> 1)) ^ ((k >> 1) & 7)) * 16) + (fr & 1) * 8));
        }
        struct { v4s16 lo, hi; } bv = {b[0], b[1]};
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          dacc[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv), __builtin_bit_cast(bf16x8, a[tm]), dacc[tm], 0, 0, 0);
      }
      // (operands swapped: dacc[tm][r] = U[row 16 tm + fr][column 16 wave + 4 fg + r], the layout of gate1)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int mm = m0 + tm * 16 + fr;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (((gate1 >> (tm * 4 + r)) & 1u) && n0 + r < P.H) ? dacc[tm][r] * B.scale : 0.f;
        if (mm < P.rows) *(uint2*)((bf16_t*)B.dz1 + (int64_t)mm * P.ldh + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
      }
      // ---- the head of THIS critic on its own rows (cycle mode: Q' given), last so that wave 0 never holds a barrier up:
      // the arithmetic and the reduction tree of the head above (recnn/nn/update/misc.py:6-7,33-39, td3.py:83-93)
      if (self_head && wave == 0) {
        const MlpHead& Hd = batch.head;
        const int c = P.cbwd_idx;
        const int r = lane & 31, m = m0 + r;
        const bool valid = lane < 32 && m < P.rows;
        float y = h_rew + (1.0f - h_done) * Hd.gamma * h_tq;
        y = fminf(fmaxf(y, Hd.lo), Hd.hi);
        if (valid && c == 0) {
          if (Hd.expected) Hd.expected[m] = y;
          if (Hd.target_q) Hd.target_q[m] = h_tq;
        }
        const float q = ((const float*)(lds + ps0 * STAGE1 + 3072))[r];
        const float e = valid ? q - y : 0.f;
        const float d = e * (2.0f / (float)P.rows);
        if (valid && Hd.delta_out[c]) Hd.delta_out[c][m] = d;
        const float tot = wave_sum(e * e);
        const float dsum = wave_sum(d);
        if (lane == 0) {
          if (Hd.loss_part[c]) Hd.loss_part[c][bx] = tot;
          if (Hd.db3_part[c]) Hd.db3_part[c][bx] = dsum;
        }
      }
    }
  }
  MLPS_STAMP(9);
}

// ---- debug hooks (recnn_hip_debug.h: not part of the public ABI; process-wide by nature -- a trace buffer, a fault to inject)
static unsigned long long* g_mlps_trace = nullptr;
extern "C" void recnn_debug_mlp_trace(void* device_u64_wg32) { g_mlps_trace = (unsigned long long*)device_u64_wg32; }
static int g_mlp_fault = 0;
// test hook: break a hand-off on purpose (1: layer-1 part flags, 2: Q slots) with a short spin bound, to exercise the
// error path (tests/test_gpu_engine.py::test_broken_handoff_is_reported); bits 0x100 / 0x200 select the timing probes
extern "C" void recnn_debug_mlp_fault(int mode) { g_mlp_fault = mode; }
extern "C" void recnn_debug_mlp_probe(int bits) { g_mlp_fault = (g_mlp_fault & 0xFF) | ((bits & 3) << 8); }

int mlp_init() {
  int rc = recnn_check_hip(hipFuncSetAttribute((const void*)mlps_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL), "mlps attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)mlps_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL), "mlps attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)mlps_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL), "mlps attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)mlps_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL), "mlps attr");
  return rc;
}

int mlp_waves() { return NW; }

// Round 3: this is the only fused row-panel forward left.  mlp.hip (burst-and-wait schedule), mlp64.hip (64-row panels) and
// mlpr.hip (weights straight into MFMA registers) were measured slower at every shape the engine produces (47 and 70 us vs
// 26 at DDPG / 2048 rows) and are gone; their bit-for-bit agreement with this kernel was tested up to their removal
// (tests/test_gpu_kernels.py history, profiles/r02_gpu_tests_v7.log).
int mlp_launch(const MlpBatch& b_in, int nprob, hipStream_t s) {
  MlpBatch b = b_in;
  b.fault = g_mlp_fault;
  if (g_mlp_fault & 3) b.spin_limit = 1 << 12;
  int rows = 0;
  for (int i = 0; i < nprob; ++i) {
    const MlpProb& p = b.p[i];
    if (p.rows > rows) rows = p.rows;
    if (p.H > HP || p.out_dim > 128) { recnn_set_error("mlp_fwd: hidden > 256 or out_dim > 128"); return RECNN_E_UNSUPPORTED; }
    if ((p.H & 3) || p.H < 4) { recnn_set_error("mlp_fwd: hidden width must be a multiple of 4"); return RECNN_E_UNSUPPORTED; }
    if (p.cbwd_idx >= 2 || (p.cbwd_idx >= 0 && (p.W3 || !p.q || !b.cbwd[p.cbwd_idx].dz2 || !b.cbwd[p.cbwd_idx].dz1))) {
      recnn_set_error("mlp_fwd: critic backward tail needs a critic problem and its buffers");
      return RECNN_E_INVALID;
    }
    if (p.n_tail < 0 || p.n_tail > MLP_MAX_TAIL || (p.n_tail && !p.W3) || (p.part_out && !p.part_flag)) {
      recnn_set_error("mlp_fwd: bad chained-critic description");
      return RECNN_E_INVALID;
    }
    // one lane offset serves every 256-pitch matrix of a workgroup's stream
    if (p.W3 && p.ldw3 != p.ldw2) { recnn_set_error("mlp_fwd: W2 / W3 shadows must share one pitch"); return RECNN_E_INVALID; }
    for (int t = 0; t < (p.W3 ? p.n_tail : 0); ++t)
      if (b.tail[t].ldw2 != p.ldw2) { recnn_set_error("mlp_fwd: chained critics must share the W2 pitch"); return RECNN_E_INVALID; }
    for (int g = 0; g < p.nseg; ++g)
      if (p.K[g] % KB1 || (p.lda[g] % 8) || ((uintptr_t)p.A[g] & 15)) { recnn_set_error("mlp_fwd: bad segment"); return RECNN_E_INVALID; }
  }
  if (rows <= 0 || nprob <= 0) return 0;
  const int panels = (rows + BM - 1) / BM;
  dim3 grid(panels, nprob);
  const dim3 block(NW * 64);
  b.nprob = nprob;
  switch ((b.fault >> 8) & 3) {
    case 1: hipLaunchKernelGGL(mlps_fwd_kernel<1>, grid, block, LDS_TOTAL, s, b, g_mlps_trace); break;
    case 2: hipLaunchKernelGGL(mlps_fwd_kernel<2>, grid, block, LDS_TOTAL, s, b, g_mlps_trace); break;
    case 3: hipLaunchKernelGGL(mlps_fwd_kernel<3>, grid, block, LDS_TOTAL, s, b, g_mlps_trace); break;
    default: hipLaunchKernelGGL(mlps_fwd_kernel<0>, grid, block, LDS_TOTAL, s, b, g_mlps_trace);
  }
  return recnn_check_hip(hipGetLastError(), "mlps_fwd_kernel");
}
