// mlp.hip -- fused row-panel forward of a whole Actor / Critic MLP (bf16 compute, gfx950).
//
// Replaces, in ONE launch per group of network applications, what recnn/nn/models.py:66-73 / :207-213 do with
// cat + 3 x (addmm, relu, dropout): the three dependent GEMMs of a network are chained inside a workgroup, so the
// hidden activations never leave the CU between layers and two kernel boundaries per chain disappear
// (at batch 2048 every boundary costs more than the arithmetic of layers 2 and 3).
//
// One workgroup (4 waves) owns 32 batch rows of one network and ALL 256 hidden columns (wave w: columns 64w..64w+63):
//   layer 1  k loop over the packed input rows: A panel (32 x 128 k) and the full W1 k-slab (256 x 128 k) stream
//            global -> LDS by global_load_lds (2-stage ring, 72 KB per stage), MFMA 16x16x32 bf16, fp32 accumulate
//   epilogue bias + relu + dropout -> bf16 panel in LDS (the next layer's A operand) and, if requested, global h1
//   layer 2  A = LDS panel, W2 (2 k-slabs) by DMA;  same epilogue -> panel, global h2
//   layer 3  actor: W3 by DMA, 32 x 128 outputs (+ TD3 noise) -> global;   critic: per-row dot with fp32 w3 -> q
//   chained critics (mlp.h: MlpTail): the target critic of [next_action | next_state] needs the target actor's
//            output of the same 32 rows.  Its layer-1 state part (1290 of 1418 k) is computed meanwhile by a producer
//            workgroup of the same launch (part_out mode: layer 1 only, raw fp32 out, release flag); the actor's
//            workgroup acquires the flag, starts from that part, adds action x W1[:, :128] and runs layers 2 / 3 on
//            chip -> q.  Removes two launches and the re-read of next rows; measured 36.7 -> 30.6 us for the
//            target side of a DDPG step.
// LDS image of every k-slab row is 256 bytes; 16-byte chunk c of row r sits at chunk position c ^ (r & 15)
// (applied on the DMA source address / the panel write address), so MFMA fragment reads are bank-conflict free.
// Whole 160 KiB of LDS per workgroup (1 workgroup per CU): 2 x (8 KB A + 64 KB W) + 16 KB panel.
#include "mlp.h"
#include "mlp_panel.h"

namespace {
constexpr int KB = 128;           // bf16 k elements per stage row (256 bytes)
constexpr int A_BYTES = BM * 256;             // 8 KB
constexpr int W_BYTES = HP * 256;             // 64 KB
constexpr int STAGE = A_BYTES + W_BYTES;      // 72 KB
constexpr int PANEL_OFF = 2 * STAGE;          // 144 KB
constexpr int LDS_TOTAL = PANEL_OFF + 2 * PANEL_HALF;  // 160 KB
// layer 1 streams through its OWN ring geometry over the same 144 KB: 4 stages of 64-k slabs (rows of 128 bytes), so that two
// slabs are in flight while a third is multiplied.  Micro-benchmark (tools/dma_bw.hip, 256 CUs streaming the same L2-resident
// 720 KB matrix): back-to-back slabs reach 102-111 GB/s per CU, a lone 72 KB burst followed by its wait 70 GB/s -- the
// 2-stage ring paid that latency bubble on every slab (in-kernel trace: 1,290 of 2,150 ticks per slab spent waiting).
constexpr int KB1 = 64;                       // bf16 k elements per layer-1 slab row (128 bytes)
constexpr int A1_BYTES = BM * 128;            // 4 KB
constexpr int W1_BYTES = HP * 128;            // 32 KB
constexpr int STAGE1 = A1_BYTES + W1_BYTES;   // 36 KB
constexpr int NSTAGE1 = 4;                    // 144 KB = the two 72 KB stages of the later layers

// DMA `nrows` rows x 256 bytes (k slab [k0, k0+128) of a bf16 matrix with row pitch ld elements) into LDS at
// lds_dst; rows beyond row_max are clamped (their results are never stored).  4 rows per wave instruction.
template <int NW>
__device__ __forceinline__ void dma_rows(const void* base, int64_t ld, int row0, int row_max, int k0, int nrows,
                                         unsigned lds_dst, int wave, int lane) {
  const int q_row = lane >> 4, q_pos = lane & 15;
  for (int j = 0; (j * NW + wave) * 4 < nrows; ++j) {   // 4 rows per wave instruction, waves interleaved
    const int row = (j * NW + wave) * 4 + q_row;
    const int c = q_pos ^ (row & 15);
    const int gr = min(row0 + row, row_max);
    const char* src = (const char*)base + ((int64_t)gr * ld + k0) * 2 + c * 16;
    dma16(src, lds_dst + (j * NW + wave) * 1024);
  }
}

// acc[tm][tn] += A(32 x 128) * B(rows wn0.. x 128)^T for one k slab already in LDS.  The MFMA operands are SWAPPED (weights
// first): a lane then holds four consecutive output COLUMNS of one batch row --
//     acc[tm][tn][r] = C[row 16 tm + fr][column wn0 + 16 tn + 4 fg + r]
// -- so the epilogues convert and store four neighbours at once (one 8-byte LDS / global access instead of four 2-byte
// ones) and one dropout word serves them.  Same products, same k order: bit-identical to the unswapped form.
template <int TN>
__device__ __forceinline__ void mma_slab(const unsigned char* sa, const unsigned char* sb, f32x4 (&acc)[2][TN], int wn0, int fr,
                                         int fg) {
#pragma unroll
  for (int ks = 0; ks < KB / 32; ++ks) {
    const int pos = ((ks * 4 + fg) ^ fr) * 16;
    uint4 a[2], b[TN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * 256 + pos);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *(const uint4*)(sb + (wn0 + tn * 16 + fr) * 256 + pos);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[tn]), __builtin_bit_cast(bf16x8, a[tm]),
                                                              acc[tm][tn], 0, 0, 0);
  }
}

// the same for a layer-1 slab of the 4-stage ring: 64 k per slab, rows of 128 bytes, chunk c of row r at position
// c ^ ((r >> 1) & 7) (conflict-free for the 4 x 16 lane groups of ds_read_b128).  k order unchanged: bit-identical sums.
template <int TN>
__device__ __forceinline__ void mma_slab64(const unsigned char* sa, const unsigned char* sb, f32x4 (&acc)[2][TN], int wn0, int fr,
                                           int fg) {
  const int sw = (fr >> 1) & 7;               // (rows 16 apart share it)
#pragma unroll
  for (int ks = 0; ks < KB1 / 32; ++ks) {
    const int pos = ((ks * 4 + fg) ^ sw) * 16;
    uint4 a[2], b[TN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * 128 + pos);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *(const uint4*)(sb + (wn0 + tn * 16 + fr) * 128 + pos);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[tn]), __builtin_bit_cast(bf16x8, a[tm]),
                                                              acc[tm][tn], 0, 0, 0);
  }
}

typedef short v4s16 __attribute__((ext_vector_type(4)));
}  // namespace

// trace (tools/mlp_trace.py): lane 0 of wave 0 stamps the shader clock at the phase boundaries into trace[workgroup][16]
#define MLP_STAMP(i) do { if (trace && threadIdx.x == 0) trace[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
template <int NW>
__global__ __launch_bounds__(NW * 64) void mlp_fwd_kernel(const MlpBatch batch, unsigned long long* trace) {
  constexpr int TNH = HP / (16 * NW);   // 16-column MFMA tiles per wave in the hidden layers (4 waves: 4, 8 waves: 2)
  constexpr int OW = NW > 8 ? 8 : NW;   // waves that take part in the actor's 128-column output layer
  constexpr int TNO = 128 / (16 * OW);  // 16-column tiles per participating wave there
  constexpr int RW = BM / NW;           // critic head rows per wave
  const MlpProb& P = batch.p[blockIdx.y];
  const int m0 = blockIdx.x * BM;
  if (m0 >= P.rows) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int row_max = P.rows - 1;
  MLP_STAMP(0);

  f32x4 acc[2][TNH];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- biases of every epilogue of this workgroup, fetched BEFORE the first DMA: a compiler-visible load that is waited
  // for while DMAs are in flight drains them (the compiler's s_waitcnt cannot count the asm DMAs), which used to serialise
  // the W2 / W3 transfers with the epilogues they were meant to overlap (in-kernel trace: 7k of 8.7k ticks per epilogue)
  f32x4 b1v[TNH], b2v[TNH], tb1[MLP_MAX_TAIL][TNH], tb2[MLP_MAX_TAIL][TNH];
#pragma unroll
  for (int tn = 0; tn < TNH; ++tn) {
    const int n0 = wave * (16 * TNH) + tn * 16 + fg * 4;
    // (bias vectors start on 16-byte boundaries of the 16-byte aligned arenas whenever H % 4 == 0: one load per vector)
    const bool in = n0 + 3 < P.H && !P.part_out && !(P.H & 3);
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    b1v[tn] = in ? *(const f32x4*)(P.b1 + n0) : z4;
    b2v[tn] = in ? *(const f32x4*)(P.b2 + n0) : z4;
#pragma unroll
    for (int ti = 0; ti < MLP_MAX_TAIL; ++ti) {
      tb1[ti][tn] = (in && ti < P.n_tail) ? *(const f32x4*)(batch.tail[ti].b1 + n0) : z4;
      tb2[ti][tn] = (in && ti < P.n_tail) ? *(const f32x4*)(batch.tail[ti].b2 + n0) : z4;
    }
    if (!in && !P.part_out) {                                  // ragged hidden width: element by element
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = n0 + r < P.H;
        b1v[tn][r] = ok ? P.b1[n0 + r] : 0.f;
        b2v[tn][r] = ok ? P.b2[n0 + r] : 0.f;
#pragma unroll
        for (int ti = 0; ti < MLP_MAX_TAIL; ++ti) {
          tb1[ti][tn][r] = (ok && ti < P.n_tail) ? batch.tail[ti].b1[n0 + r] : 0.f;
          tb2[ti][tn][r] = (ok && ti < P.n_tail) ? batch.tail[ti].b2[n0 + r] : 0.f;
        }
      }
    }
  }
#pragma unroll
  for (int tn = 0; tn < TNH; ++tn) {       // the empty asm makes the compiler wait for the loads here
    asm volatile("" : "+v"(b1v[tn]), "+v"(b2v[tn]));
#pragma unroll
    for (int ti = 0; ti < MLP_MAX_TAIL; ++ti) asm volatile("" : "+v"(tb1[ti][tn]), "+v"(tb2[ti][tn]));
  }

  // ------------------------------------------------------------------ layer 1
  const int nt0 = P.K[0] / KB1;
  const int nt = nt0 + (P.nseg > 1 ? P.K[1] / KB1 : 0);
  // per-lane source pointers of the layer-1 DMA, advanced by one k slab (128 B) per issue: the DMA statements clobber
  // "memory", so anything read from P.* inside the loop would be re-fetched from the kernel-argument segment every time.
  // One wave instruction moves 8 rows x 128 B; instruction i of a slab covers image rows 8 i .. 8 i + 7.
  const int q_row = lane >> 3, q_pos = lane & 7;
  const int l_row = wave * 8 + q_row;                          // image row of DMA instruction `wave` (+ 8 NW per further one)
  const int l_c = (q_pos ^ ((l_row >> 1) & 7)) * 16;           // (rows 8 NW apart share (row >> 1) & 7)
  const bool a_wave = wave * 8 < BM;                           // waves 0..3 also carry the A panel (one instruction each)
  const int64_t gr_a = min(m0 + l_row, row_max);               // (only meaningful on the A waves)
  const char* a_ptr = (const char*)P.A[0] + gr_a * P.lda[0] * 2 + l_c;
  const char* a_ptr1 = P.nseg > 1 ? (const char*)P.A[1] + gr_a * P.lda[1] * 2 + l_c : a_ptr;
  constexpr int NIW = HP / (8 * NW);                           // W instructions per wave and slab
  const int64_t w_step = (int64_t)8 * NW * P.ldw1 * 2;         // bytes between the rows of consecutive instructions of a wave
  const char* w_ptr = (const char*)P.W1 + ((int64_t)l_row * P.ldw1 + P.w1_col[0]) * 2 + l_c;
  const char* w_ptr1 = (const char*)P.W1 + ((int64_t)l_row * P.ldw1 + (P.nseg > 1 ? P.w1_col[1] : 0)) * 2 + l_c;
  auto issue1 = [&](int t) {
    if (t == nt0) {                                            // second contraction segment
      w_ptr = w_ptr1;
      a_ptr = a_ptr1;
    }
    const unsigned sb = lds0 + (t & (NSTAGE1 - 1)) * STAGE1;
    if (a_wave) dma16(a_ptr, sb + wave * 1024);
    a_ptr += 2 * KB1;
#pragma unroll
    for (int j = 0; j < NIW; ++j) dma16(w_ptr + j * w_step, sb + A1_BYTES + (j * NW + wave) * 1024);
    w_ptr += 2 * KB1;
  };
  issue1(0);
  if (nt > 1) issue1(1);
  if (nt > 2) issue1(2);
  MLP_STAMP(1);
  for (int t = 0; t < nt; ++t) {
    // slab t of THIS wave has landed once at most the DMAs of the younger slabs in flight are outstanding (the loop holds
    // no other vector-memory operation; A waves issue one instruction more per slab)
    const int younger = min(NSTAGE1 - 2, nt - 1 - t);
    if (a_wave) {
      if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NIW + 1)) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIW) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // slab t landed for every wave; the stage of slab t-1 (= slab t+3's) is no longer being read
    if (t + NSTAGE1 - 1 < nt && !(batch.fault & 0x200)) issue1(t + NSTAGE1 - 1);
    const unsigned char* st = lds + (t & (NSTAGE1 - 1)) * STAGE1;
    if (!(batch.fault & 0x100)) mma_slab64<TNH>(st, st + A1_BYTES, acc, wave * (16 * TNH), fr, fg);
  }
  MLP_STAMP(2);
  if (P.part_out) {
    // producer of a chained critic: hand the raw pre-activation part to the consumer workgroup of this panel
#pragma unroll
    for (int tn = 0; tn < TNH; ++tn) {
      const int n0 = wave * (16 * TNH) + tn * 16 + fg * 4;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) *(f32x4*)(P.part_out + (int64_t)(m0 + tm * 16 + fr) * HP + n0) = acc[tm][tn];
    }
    __syncthreads();  // every thread's stores have completed (the barrier is preceded by s_waitcnt vmcnt(0))
    if (tid == 0 && (batch.fault & 3) != 1) __hip_atomic_store(P.part_flag + blockIdx.x, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // one L2 write-back
    MLP_STAMP(9);
    return;
  }
  __builtin_amdgcn_s_barrier();  // ring free
  // W2: both k slabs straight away (they overlap the epilogue below)
  dma_rows<NW>(P.W2, P.ldw2, 0, HP - 1, 0, HP, lds0 + A_BYTES, wave, lane);
  dma_rows<NW>(P.W2, P.ldw2, 0, HP - 1, KB, HP, lds0 + STAGE + A_BYTES, wave, lane);

  uint32_t key1 = 0, key2 = 0;
  if (P.mask_mode == RECNN_MASK_HASH) {
    const int32_t st = (P.step_ptr ? *P.step_ptr : 0) + P.step_add;
    key1 = mask_key(P.seed, st, P.stream1);
    key2 = mask_key(P.seed, st, P.stream2);
  }
  unsigned char* panel = lds + PANEL_OFF;
  uint32_t gate1 = 0;  // relu/dropout gate of h1 for this lane's accumulator elements (bit tn*8 + tm*4 + r)
  hidden_epilogue<TNH>(acc, b1v, P.H, P.rows, m0, wave, fr, fg, P.mask_mode, P.mask1, P.ld_mask, key1, panel, &gate1);

  MLP_STAMP(3);
  // ------------------------------------------------------------------ layer 2
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // W2 landed, h1 panel complete (LDS writes drained before the raw barrier)
  if (P.h1) panel_to_global<NW>(panel, (bf16_t*)P.h1, P.ldh, m0, P.rows, tid);
  mma_slab<TNH>(panel, lds + A_BYTES, acc, wave * (16 * TNH), fr, fg);
  mma_slab<TNH>(panel + PANEL_HALF, lds + STAGE + A_BYTES, acc, wave * (16 * TNH), fr, fg);
  __builtin_amdgcn_s_barrier();  // everyone is done with W2 and the h1 panel
  MLP_STAMP(4);
  f32x4 pacc[MLP_MAX_TAIL][2][TNH];  // chained critics: layer-1 state parts handed over by the producer workgroups
  if (P.W3) {                    // actor: both k slabs of W3 (128 rows each, 32 KB) into W slot 0 ...
    dma_rows<NW>(P.W3, P.ldw3, 0, 127, 0, 128, lds0 + A_BYTES, wave, lane);
    dma_rows<NW>(P.W3, P.ldw3, 0, 127, KB, 128, lds0 + A_BYTES + 128 * 256, wave, lane);
    if (P.n_tail) {              // ... which leaves slot 1 for the first chained critic's action-column slab of W1
      dma_rows<NW>(batch.tail[0].W1a, batch.tail[0].ldw1, 0, HP - 1, 0, HP, lds0 + STAGE + A_BYTES, wave, lane);
      if (tid == 0) {
        // bounded spin (~0.2 s): a producer has a lower workgroup id, so it was dispatched before this workgroup and
        // never waits itself (by now it normally finished long ago); the bound turns a broken launch order into a
        // REPORTED error (batch.err -> RECNN_E_STATE at the next loss / counter read) instead of a hang
        const int limit = batch.spin_limit > 0 ? batch.spin_limit : (1 << 22);
        for (int ti = 0; ti < P.n_tail; ++ti) {
          int spins = 0;
          bool ok;
          while (!(ok = __hip_atomic_load(batch.tail[ti].flag + blockIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0) && ++spins < limit)
            __builtin_amdgcn_s_sleep(2);
          if (!ok && batch.err) __hip_atomic_fetch_or(batch.err, MLP_ERR_PART_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(batch.tail[ti].flag + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
#pragma unroll
      for (int ti = 0; ti < MLP_MAX_TAIL; ++ti)
        if (ti < P.n_tail) {
#pragma unroll
          for (int tn = 0; tn < TNH; ++tn) {
            const int n0 = wave * (16 * TNH) + tn * 16 + fg * 4;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) pacc[ti][tm][tn] = *(const f32x4*)(batch.tail[ti].part + (int64_t)(m0 + tm * 16 + fr) * HP + n0);
          }
        }
    }
  }
  hidden_epilogue<TNH>(acc, b2v, P.H, P.rows, m0, wave, fr, fg, P.mask_mode, P.mask2, P.ld_mask, key2, panel);
  MLP_STAMP(5);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // h2 panel complete (and W3 landed)
  if (P.h2) panel_to_global<NW>(panel, (bf16_t*)P.h2, P.ldh, m0, P.rows, tid);

  // ------------------------------------------------------------------ layer 3
  if (P.W3) {
    if (wave < OW) {
    f32x4 o[2][TNO];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TNO; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_slab<TNO>(panel, lds + A_BYTES, o, wave * (16 * TNO), fr, fg);
    mma_slab<TNO>(panel + PANEL_HALF, lds + A_BYTES + 128 * 256, o, wave * (16 * TNO), fr, fg);
#pragma unroll
    for (int tn = 0; tn < TNO; ++tn) {
      const int n0 = wave * (16 * TNO) + tn * 16 + fg * 4;    // this lane's four output columns
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = n0 + r < P.out_dim ? P.b3[n0 + r] : 0.f;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int row = tm * 16 + fr, m = m0 + row;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ncol = n0 + r < P.out_dim;
          v[r] = o[tm][tn][r] + bv[r];
          if (P.addend && ncol && m < P.rows) {
            const float z = P.addend[(int64_t)m * P.ld_add + n0 + r];
            v[r] += fminf(fmaxf(z, -P.add_clip), P.add_clip);
          }
          if (!ncol) v[r] = 0.f;
        }
        uint2 packed = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        if (n0 + 3 >= P.out_dim) {                             // (padded columns hold bf16 +0, not -0)
          if (n0 + 0 >= P.out_dim) packed.x &= 0xFFFF0000u;
          if (n0 + 1 >= P.out_dim) packed.x &= 0x0000FFFFu;
          if (n0 + 2 >= P.out_dim) packed.y &= 0xFFFF0000u;
          if (n0 + 3 >= P.out_dim) packed.y &= 0x0000FFFFu;
        }
        if (m < P.rows) {
          if (n0 + 3 < P.out_dim) {
            *(uint2*)((bf16_t*)P.out + (int64_t)m * P.ldo + n0) = packed;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n0 + r < P.out_dim) ((bf16_t*)P.out)[(int64_t)m * P.ldo + n0 + r] = (bf16_t)((r < 2 ? packed.x : packed.y) >> ((r & 1) * 16));
          }
        }
        // chained critics read the action panel from the (idle) A slot of ring stage 0, same image as a k slab
        if (P.n_tail) *(uint2*)(lds + row * 256 + (((n0 >> 3) ^ fr) << 4) + (n0 & 7) * 2) = packed;
      }
    }
    }
    MLP_STAMP(6);
    // ---------------------------------------------------------------- chained critics (target critic on the new action)
#pragma unroll
    for (int ti = 0; ti < MLP_MAX_TAIL; ++ti) {
      if (ti >= P.n_tail) break;
      const MlpTail& T = batch.tail[ti];
      constexpr int NI = HP / (4 * NW);  // DMA instructions per wave for one 256-row slab
      // layer-3 MMAs (ti = 0) / the previous critic's head are done; action panel written; W1a and the parts landed
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      dma_rows<NW>(T.W2, T.ldw2, 0, HP - 1, 0, HP, lds0 + A_BYTES, wave, lane);  // W2 slab 0 -> slot 0
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TNH; ++j) acc[i][j] = pacc[ti][i][j];
      mma_slab<TNH>(lds, lds + STAGE + A_BYTES, acc, wave * (16 * TNH), fr, fg);  // + action panel x W1a
      __builtin_amdgcn_s_barrier();  // slot 1 free
      dma_rows<NW>(T.W2, T.ldw2, 0, HP - 1, KB, HP, lds0 + STAGE + A_BYTES, wave, lane);  // W2 slab 1 -> slot 1
      hidden_epilogue<TNH>(acc, tb1[ti], P.H, P.rows, m0, wave, fr, fg, RECNN_MASK_NONE, nullptr, 0, 0u, panel);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TNH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NI) : "memory");
      __builtin_amdgcn_s_barrier();  // slab 0 landed (slab 1 may still be in flight), h1 panel complete
      mma_slab<TNH>(panel, lds + A_BYTES, acc, wave * (16 * TNH), fr, fg);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      mma_slab<TNH>(panel + PANEL_HALF, lds + STAGE + A_BYTES, acc, wave * (16 * TNH), fr, fg);
      __builtin_amdgcn_s_barrier();  // everyone is done reading the h1 panel and both slots
      if (ti + 1 < P.n_tail)
        dma_rows<NW>(batch.tail[ti + 1].W1a, batch.tail[ti + 1].ldw1, 0, HP - 1, 0, HP, lds0 + STAGE + A_BYTES, wave, lane);
      hidden_epilogue<TNH>(acc, tb2[ti], P.H, P.rows, m0, wave, fr, fg, RECNN_MASK_NONE, nullptr, 0, 0u, panel);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int i = 0; i < RW; ++i) {
        const int row = wave * RW + i;
        float sdot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = lane * 4 + j;
          const int c = ((n & 127) >> 3) ^ (row & 15);
          const bf16_t hv = *(const bf16_t*)(panel + (n >> 7) * PANEL_HALF + row * 256 + c * 16 + (n & 7) * 2);
          sdot += n < P.H ? bf2f(hv) * T.w3row[n] : 0.f;
        }
        sdot = wave_sum(sdot);
        if (lane == 0) {
          const float qv = sdot + T.b3[0];
          if (m0 + row < P.rows) T.q[m0 + row] = qv;
          ((float*)(lds + STAGE))[ti * BM + row] = qv;   // for the head below (A slot of ring stage 1 is idle)
        }
      }
    }
    MLP_STAMP(7);
    // ---------------------------------------------------------------- head of the learning critic(s)
    if (batch.head.n_critic > 0 && P.n_tail > 0) {
      const MlpHead& Hd = batch.head;
      __syncthreads();   // Q' of all 32 rows (every tail) is in LDS
      if (wave == 0) {
        const float* stq = (const float*)(lds + STAGE);
        const int r = lane & 31, m = m0 + r, mc = min(m, P.rows - 1);
        const bool valid = lane < 32 && m < P.rows;
        const float rew = Hd.reward[mc], dn = Hd.done[mc];
        float tqv = stq[r];
        if (P.n_tail > 1) tqv = fminf(tqv, stq[BM + r]);
        float y = rew + (1.0f - dn) * Hd.gamma * tqv;
        y = fminf(fmaxf(y, Hd.lo), Hd.hi);
        if (valid) {
          if (Hd.expected) Hd.expected[m] = y;
          if (Hd.target_q) Hd.target_q[m] = tqv;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (c < Hd.n_critic) {
            // Q(s, a) from the critic workgroup of the same rows: it finished its forward long ago; bounded spin
            // (~0.2 s) on the value-as-flag slot, then put the slot back to rest
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
              if (Hd.loss_part[c]) Hd.loss_part[c][blockIdx.x] = tot;
              if (Hd.db3_part[c]) Hd.db3_part[c][blockIdx.x] = dsum;
            }
          }
        }
      }
    }
  } else if (P.q) {
    // critic head: q[m] = h2[m, :] . w3 + b3   (8 rows per wave, lanes split the 256 columns)
    for (int i = 0; i < RW; ++i) {
      const int row = wave * RW + i;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = lane * 4 + j;
        const int c = ((n & 127) >> 3) ^ (row & 15);
        const bf16_t hv = *(const bf16_t*)(panel + (n >> 7) * PANEL_HALF + row * 256 + c * 16 + (n & 7) * 2);
        s += n < P.H ? bf2f(hv) * P.w3row[n] : 0.f;
      }
      s = wave_sum(s);
      if (lane == 0 && m0 + row < P.rows) {
        const float qv = s + P.b3[0];
        P.q[m0 + row] = qv;
        if (P.cbwd_idx >= 0 && batch.cbwd[P.cbwd_idx].q_slot && (batch.fault & 3) != 2)   // hand Q(s, a) to the workgroup that evaluates the head (value = flag)
          __hip_atomic_store((uint32_t*)batch.cbwd[P.cbwd_idx].q_slot + m0 + row, __builtin_bit_cast(uint32_t, qv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    MLP_STAMP(6);
    if constexpr (NW == 16) {
      if (P.cbwd_idx >= 0) {
        const MlpCriticBwd& B = batch.cbwd[P.cbwd_idx];
        // ---- u2 = w3 * scale * [h2 > 0], in place in the panel (it becomes the A operand) and to global
        {
          const int row = lane & 31, m = m0 + row;
          const int n8 = (2 * wave + (lane >> 5)) * 8;
          const int nb = min(n8, P.H - 8);
          const float4 w3a = *(const float4*)(P.w3row + nb), w3b = *(const float4*)(P.w3row + nb + 4);
          const float wsc = n8 < P.H ? B.scale : 0.f;
          const float w3v[8] = {w3a.x * wsc, w3a.y * wsc, w3a.z * wsc, w3a.w * wsc, w3b.x * wsc, w3b.y * wsc, w3b.z * wsc, w3b.w * wsc};
          __builtin_amdgcn_s_barrier();   // every wave is done reading h2 rows for its q dots
          unsigned char* cell = panel + (n8 >> 7) * PANEL_HALF + row * 256 + ((((n8 & 127) >> 3) ^ (row & 15)) * 16);
          const uint4 raw = *(const uint4*)cell;
          const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
          float uz[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float hv = bf2f((bf16_t)((u[j >> 1] >> ((j & 1) * 16)) & 0xFFFF));
            uz[j] = hv > 0.f ? w3v[j] : 0.f;
          }
          const uint4 packed = make_uint4(pack_bf2(uz[0], uz[1]), pack_bf2(uz[2], uz[3]), pack_bf2(uz[4], uz[5]), pack_bf2(uz[6], uz[7]));
          *(uint4*)cell = packed;
          if (m < P.rows) *(uint4*)((bf16_t*)B.dz2 + (int64_t)m * P.ldh + n8) = packed;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // u2 panel complete
        // ---- U = (u2 W2) * scale * gate(h1): W2 k-slabs are still in the two W slots (rows = out index = k here,
        // 128 in-columns per slab, chunk c of row r at c ^ (r & 15)); B fragments by transpose reads
        f32x4 dacc[2];
        dacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        dacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* wslab = lds + (wave >> 3) * STAGE + A_BYTES;   // in-columns 16 wave .. +15 live in slab wave / 8
        const int cpair = (wave & 7) * 2;
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
                (__attribute__((address_space(3))) v4s16*)(wslab + k * 256 + (((cpair + ((fr & 3) >> 1)) ^ (k & 15)) * 16) + (fr & 1) * 8));
          }
          struct { v4s16 lo, hi; } bv = {b[0], b[1]};
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
            dacc[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv), __builtin_bit_cast(bf16x8, a[tm]), dacc[tm], 0, 0, 0);
        }
        // (operands swapped as in mma_slab: dacc[tm][r] = U[row 16 tm + fr][column 16 wave + 4 fg + r], the layout of gate1)
        const int n0 = wave * 16 + fg * 4;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const int mm = m0 + tm * 16 + fr;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (((gate1 >> (tm * 4 + r)) & 1u) && n0 + r < P.H) ? dacc[tm][r] * B.scale : 0.f;
          if (mm < P.rows) *(uint2*)((bf16_t*)B.dz1 + (int64_t)mm * P.ldh + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        }
      }
    }
  }
  MLP_STAMP(9);
}

int mlp64_init();
int mlp64_launch(const MlpBatch& b, int nprob, int rows, hipStream_t s);
int mlp64_map_mode();
int mlpr_init();
int mlpr_launch(const MlpBatch& b, int nprob, int rows, int map_mode, hipStream_t s);
int mlps_init();
int mlps_launch(const MlpBatch& b, int nprob, int rows, hipStream_t s);
// which fused-forward kernel runs: 3 = mlps.hip (32-row panels, every operand of a workgroup as ONE weight stream through a
// 4-stage ring; the default: 25.3 us for the DDPG forward group at 2048 rows), 0 = mlp.hip (32-row panels, per-phase bursts:
// 30.4 us; also the fallback for ragged hidden widths), 1 = mlp64.hip (64-row panels, 3-deep ring: 47 us), 2 = mlpr.hip
// (64-row panels, weights straight into registers: 70 us).  All four agree bit for bit; DESIGN.md section 5b has the
// in-kernel phase traces that explain the ranking.
static int g_mlp_kernel = 3;
extern "C" void recnn_tune_mlp_kernel(int k) { g_mlp_kernel = (k >= 0 && k <= 3) ? k : 3; }
// rows per workgroup: 32 = the kernel in this file (default: 32.4 us for the DDPG forward group at 2048 rows), 64 =
// mlp64.hip (bit-identical results; faster at TD3 / 4096 rows, 47 us at DDPG / 2048 rows: see DESIGN.md section 5)
extern "C" void recnn_tune_mlp_panel(int rows) { g_mlp_kernel = rows == 64 ? 1 : 3; }
static unsigned long long* g_mlp32_trace = nullptr;
void mlp32_set_trace(void* p) { g_mlp32_trace = (unsigned long long*)p; }
static int g_mlp_waves = 16;
static int g_mlp_fault = 0;
// test hook: break a hand-off on purpose (1: layer-1 part flags, 2: Q slots) with a short spin bound, to exercise the
// error path (tests/test_gpu_engine.py::test_broken_handoff_is_reported)
extern "C" void recnn_tune_mlp_fault(int mode) { g_mlp_fault = mode; }
extern "C" void recnn_tune_mlp_waves(int w) { g_mlp_waves = (w == 4 || w == 16) ? w : 8; }

int mlp_init() {
  int rc = mlp64_init();
  if (rc) return rc;
  if ((rc = mlpr_init())) return rc;
  if ((rc = mlps_init())) return rc;
  rc = recnn_check_hip(hipFuncSetAttribute((const void*)mlp_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL),
                           "mlp_fwd_kernel<4> attr");
  if (rc) return rc;
  rc = recnn_check_hip(hipFuncSetAttribute((const void*)mlp_fwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL),
                       "mlp_fwd_kernel<8> attr");
  if (rc) return rc;
  return recnn_check_hip(hipFuncSetAttribute((const void*)mlp_fwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL),
                         "mlp_fwd_kernel<16> attr");
}

int mlp_waves() { return g_mlp_waves; }

int mlp_launch(const MlpBatch& b_in, int nprob, hipStream_t s) {
  MlpBatch b = b_in;
  b.fault = g_mlp_fault;
  if (g_mlp_fault & 3) b.spin_limit = 1 << 12;
  int rows = 0;
  for (int i = 0; i < nprob; ++i) {
    const MlpProb& p = b.p[i];
    if (p.rows > rows) rows = p.rows;
    if (p.H > HP || p.out_dim > 128) { recnn_set_error("mlp_fwd: hidden > 256 or out_dim > 128"); return RECNN_E_UNSUPPORTED; }
    if (p.cbwd_idx >= 2 || (p.cbwd_idx >= 0 && (g_mlp_waves != 16 || p.W3 || !p.q || !b.cbwd[p.cbwd_idx].dz2 || !b.cbwd[p.cbwd_idx].dz1))) {
      recnn_set_error("mlp_fwd: critic backward tail needs the 16-wave variant, a critic problem and its buffers");
      return RECNN_E_INVALID;
    }
    if (p.n_tail < 0 || p.n_tail > MLP_MAX_TAIL || (p.n_tail && !p.W3) || (p.part_out && !p.part_flag)) {
      recnn_set_error("mlp_fwd: bad chained-critic description");
      return RECNN_E_INVALID;
    }
    for (int g = 0; g < p.nseg; ++g)
      if (p.K[g] % KB1 || (p.lda[g] % 8) || ((uintptr_t)p.A[g] & 15)) { recnn_set_error("mlp_fwd: bad segment"); return RECNN_E_INVALID; }
  }
  if (rows <= 0 || nprob <= 0) return 0;
  if (g_mlp_kernel == 1 && g_mlp_waves == 16) return mlp64_launch(b, nprob, rows, s);
  if (g_mlp_kernel == 2 && g_mlp_waves == 16) return mlpr_launch(b, nprob, rows, mlp64_map_mode(), s);
  if (g_mlp_kernel == 3 && g_mlp_waves == 16) {
    bool ok = true;                                            // (mlps.hip wants hidden widths that are multiples of 4)
    for (int i = 0; i < nprob; ++i) {
      const MlpProb& p = b.p[i];
      ok = ok && !(p.H & 3) && p.H >= 4 && (!p.W3 || p.ldw3 == p.ldw2);
      for (int t = 0; t < (p.W3 ? p.n_tail : 0); ++t) ok = ok && b.tail[t].ldw2 == p.ldw2;   // one lane offset serves every 256-pitch matrix
    }
    if (ok) return mlps_launch(b, nprob, rows, s);
  }
  if (g_mlp_waves == 16)
    hipLaunchKernelGGL(mlp_fwd_kernel<16>, dim3((rows + BM - 1) / BM, nprob), dim3(1024), LDS_TOTAL, s, b, g_mlp32_trace);
  else if (g_mlp_waves == 8)
    hipLaunchKernelGGL(mlp_fwd_kernel<8>, dim3((rows + BM - 1) / BM, nprob), dim3(512), LDS_TOTAL, s, b, g_mlp32_trace);
  else
    hipLaunchKernelGGL(mlp_fwd_kernel<4>, dim3((rows + BM - 1) / BM, nprob), dim3(256), LDS_TOTAL, s, b, g_mlp32_trace);
  return recnn_check_hip(hipGetLastError(), "mlp_fwd_kernel");
}
