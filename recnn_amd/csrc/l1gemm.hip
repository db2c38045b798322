// l1gemm.hip -- layer 1 of the Actor / Critic MLPs as a tiled bf16 MFMA GEMM over the whole machine (gfx950).
//
//   h1 = dropout(relu(cat(segments) W1^T + b1))        recnn/nn/models.py:66-69 (Actor), :207-211 (Critic: cat + linear1)
//
// Both operands are k-contiguous (batch rows / weight-shadow rows), so the tile is the plain LDS-DMA pipeline: a k stage is
// 256 bytes (128 k) of every tile row of A and of W1, copied global -> LDS by `global_load_lds_dwordx4` into an NS-slot ring,
// NS - 1 stages in flight ahead of the MFMAs, counted vmcnt waits, one raw s_barrier per stage.  The 16-byte chunk c of tile
// row r sits at chunk position c ^ (r & 15) of its LDS row (XOR applied to the lane's SOURCE address): conflict-free
// ds_read_b128 fragment reads.  MFMA operands are swapped (weights first) as in mlps.hip: a lane then owns four neighbouring
// COLUMNS of one row, so the epilogue packs them into ONE 8-byte store and one dropout word serves the lane's 4 x 4 block.
// Per output element the arithmetic is exactly mlps.hip's (k ascending in steps of 32, segment 0 before segment 1, fp32
// accumulate, + bias, relu, x2 keep-mask, round to bf16): the two forwards are interchangeable bit for bit
// (tests/test_gpu_split.py).
//
// Tile shapes (16 waves as 4 x 4 each): 64 x 64 (4-slot ring = 128 KB: per-step launches -- at 4096 rows x 256 columns that
// is 256 workgroups of 393 KB each instead of 128 of 1.0 MB) and 128 x 128 / 128 x 64 (2- / 3-slot rings: the cycle-batched
// launches of the frozen networks, M >= 16k rows).
#include "split.h"

namespace {

// One LDS-DMA instruction: 64 lanes x 16 bytes from (uniform base + per-lane 32-bit byte offset) to LDS at lds_dst + 16 lane.
// M0 is written directly (the kernel has no other M0 consumer); no "memory" clobber: the ordering points are the waits and
// barriers of the consumer.  (mlps.hip's lean issue: the saddr form keeps a stage's address arithmetic to scalar adds.)
__device__ __forceinline__ void dma_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "m0");
}

template <int N> __device__ __forceinline__ void wait_vm_const() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// TM x TN 16 x 16 MFMA tiles per wave, WM x WN waves per workgroup, NS ring slots
template <int TM, int TN, int WM, int WN, int NS>
__global__ __launch_bounds__(WM * WN * 64) void l1_gemm_kernel(const L1Batch batch, unsigned long long* trace) {
  constexpr int NW = WM * WN;
  unsigned long long* trow = (trace && threadIdx.x == 0) ? trace + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 : nullptr;
  asm volatile("" : "+v"(trow));
  if (trow) trow[0] = __builtin_amdgcn_s_memtime();
  // pull the kernel-argument cache lines of this workgroup's problem into the scalar cache NOW, all in flight together (a first
  // touch costs a scalar-cache miss of ~0.5 us and the fields are otherwise fetched one dependent batch after the other)
  unsigned touch = 0;
  {
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    const char __attribute__((address_space(4)))* pa = ka + blockIdx.y * sizeof(L1Prob);
#pragma unroll
    for (int i = 0; i < (int)((sizeof(L1Prob) + 63) / 64); ++i) asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"(i * 64));
    asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"((int)sizeof(L1Prob) - 4));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(touch));
  }
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  constexpr int KB = 128;                                  // k elements per stage (256-byte rows)
  constexpr int D = NS - 1;                                // prefetch distance
  constexpr int STAGE_BYTES = (BM + BN) * 256;
  constexpr int NA = BM / (4 * NW), NB = BN / (4 * NW);    // DMA instructions per wave and stage (4 rows each)
  static_assert(BM % (4 * NW) == 0 && BN % (4 * NW) == 0, "tile rows must split over the waves");
  static_assert(D * (NA + NB) <= 60, "vmcnt is a 6-bit counter");
  const L1Prob& P = batch.p[blockIdx.y];
  const int nwg = P.tiles_m * P.tiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int lid = xcd_remap(blockIdx.x, nwg);              // consecutive tiles of one row panel share an XCD's L2
  const int tile_n = lid % P.tiles_n, tile_m = lid / P.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
  const unsigned lds0 = (unsigned)(size_t)dsmem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WN) * 16 * TM, wn0 = (wave % WN) * 16 * TN;
  const int fr = lane & 15, fg = lane >> 4;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt0 = P.K[0] / KB;
  const int nt = nt0 + (P.nseg > 1 ? P.K[1] / KB : 0);
  const int q_row = lane >> 4, q_pos = lane & 15;          // within the 4 rows one wave instruction covers

  // per-lane source offsets of this wave's DMA instructions (constant over a contraction segment): instruction j covers tile
  // rows (j NW + wave) 4 .. + 3, the lane's 16-byte chunk is source chunk q_pos ^ (row & 15) of the stage's 256 bytes
  unsigned voff_a[NA], voff_w[NB];
  auto lane_offsets = [&](int sidx) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int row = (j * NW + wave) * 4 + q_row;
      voff_a[j] = (unsigned)(min(m0 + row, P.rows - 1) * (int)P.lda[sidx] * 2 + ((q_pos ^ (row & 15)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int row = (j * NW + wave) * 4 + q_row;
      voff_w[j] = (unsigned)((n0 + row) * (int)P.ldw1 * 2 + ((q_pos ^ (row & 15)) << 4));
    }
  };
  lane_offsets(0);
  const char* sa_ptr = (const char*)P.A[0];                                   // running scalar bases: + 256 bytes per stage
  const char* sw_ptr = (const char*)P.W1 + (int64_t)P.w1_col[0] * 2;
  int issued = 0;
  auto issue = [&](int stage) {
    if (issued == nt0) {           // second contraction segment
      lane_offsets(1);
      sa_ptr = (const char*)P.A[1];
      sw_ptr = (const char*)P.W1 + (int64_t)P.w1_col[1] * 2;
    }
    ++issued;
    const unsigned sbase = lds0 + stage * STAGE_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < NA; ++j) dma_s(voff_a[j], sa_ptr, sbase + j * NW * 1024);
#pragma unroll
    for (int j = 0; j < NB; ++j) dma_s(voff_w[j], sw_ptr, sbase + BM * 256 + j * NW * 1024);
    sa_ptr += 256;
    sw_ptr += 256;
  };

#pragma unroll
  for (int i = 0; i < D; ++i)
    if (i < nt) issue(i);
  // everything the epilogue reads from global memory is requested now, under the first stages' latency (a compiler-visible
  // load inside the stream would drain the DMA queue with a vmcnt(0))
  f32x4 bias[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 16 + fg * 4;
    bias[tn] = (n + 3 < P.H) ? *(const f32x4*)(P.b1 + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // the device step counter through the SCALAR cache (a plain load of this uniform global becomes a vector load + vmcnt(0))
  int32_t step_now = 0;
  if (P.mask_mode == RECNN_MASK_HASH && P.step_ptr) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(step_now) : "s"(P.step_ptr));

  if (trow) trow[1] = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < nt; ++t) {
    // stage t has landed once at most the loads of the (up to D - 1) younger stages are still outstanding
    const int younger = min(D - 1, nt - 1 - t);
    if (younger >= 2 && D >= 3) wait_vm_const<2 * (NA + NB)>();
    else if (younger == 1 && D >= 2) wait_vm_const<NA + NB>();
    else wait_vm_const<0>();
    __builtin_amdgcn_s_barrier();  // every wave's part of stage t is in LDS; every wave is done reading stage t - 1
    if (t + D < nt) issue((t + D) % NS);
    const unsigned char* sa = dsmem + (t % NS) * STAGE_BYTES;
    const unsigned char* sb = sa + BM * 256;
#pragma unroll
    for (int ks = 0; ks < KB / 32; ++ks) {
      const int pos = ((ks * 4 + fg) ^ fr) * 16;
      uint4 a[TM], b[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = *(const uint4*)(sa + (wm0 + tm * 16 + fr) * 256 + pos);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = *(const uint4*)(sb + (wn0 + tn * 16 + fr) * 256 + pos);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[tn]), __builtin_bit_cast(bf16x8, a[tm]),
                                                                acc[tm][tn], 0, 0, 0);
    }
  }

  if (trow) trow[2] = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(step_now));
  const int32_t step0 = step_now + P.step_add;
  // ---- epilogue: acc[tm][tn][r] = C[row m0 + wm0 + 16 tm + fr][column n0 + wn0 + 16 tn + 4 fg + r]
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m0 + wm0 + tm * 16 + fr;
    int mrow = m;                    // row inside its batch / mask-key step of the batch
    uint32_t key = 0;
    if (P.mask_mode == RECNN_MASK_HASH) {
      int set = 0;
      if (P.rows_per_set > 0) { set = m / P.rows_per_set; mrow = m - set * P.rows_per_set; }
      key = mask_key(P.seed, step0 + set, P.stream);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + wn0 + tn * 16 + fg * 4;
      uint32_t word = 0;
      if (P.mask_mode == RECNN_MASK_HASH) word = mask_word(key, (uint32_t)(mrow >> 2), (uint32_t)(n >> 2));
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[tm][tn][r] + bias[tn][r];
        if (!P.no_relu) v[r] = fmaxf(v[r], 0.f);
        if (P.addend && m < P.rows && n + r < P.H) {
          const float z = P.addend[(int64_t)m * P.ld_add + n + r];
          v[r] += fminf(fmaxf(z, -P.add_clip), P.add_clip);
        }
        if (P.mask_mode == RECNN_MASK_EXTERNAL) v[r] = (m < P.rows && n + r < P.H && P.mask[(int64_t)m * P.ld_mask + n + r]) ? v[r] * 2.f : 0.f;
        else if (P.mask_mode == RECNN_MASK_HASH) v[r] = mask_keep(word, mrow & 3, r) ? v[r] * 2.f : 0.f;
        if (n + r >= P.H) v[r] = 0.f;
      }
      if (m < P.rows) *(uint2*)((bf16_t*)P.h1 + (int64_t)m * P.ldh + n) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    }
  }
  if (trow) trow[3] = __builtin_amdgcn_s_memtime();
}

template <int TM, int TN, int WM, int WN, int NS> constexpr int lds_bytes() { return NS * (16 * TM * WM + 16 * TN * WN) * 256; }

// ---- the 64 x 64 per-step tile with the waves' roles split (round 6; gemm.hip x3_fwd_ws_kernel's scheme): NL = 4 LOADER waves do nothing
// but issue the ring's LDS-DMA (8 instructions of 1 KB per 32 KB stage each, lane offsets precomputed, vmcnt-counted), 8 CONSUMER waves
// (2 x 4, wave tile 32 x 16) read fragments and multiply -- the fragment reads of a whole 128-k stage are issued ahead of its MFMAs.
// One s_barrier per stage for all twelve waves; loaders leave after the last.  Same LDS image, same k order per accumulator, same
// epilogue as l1_gemm_kernel<L1_SMALL>: bit-identical (tests/test_gpu_split.py runs under both).
constexpr int WS_NL = 4, WS_NC = 8, WS_NS = 4, WS_BM = 64, WS_BN = 64, WS_STAGE = (WS_BM + WS_BN) * 256;
__global__ __launch_bounds__((WS_NL + WS_NC) * 64) void l1_gemm_ws_kernel(const L1Batch batch) {
  kernarg_prefetch<(int)sizeof(L1Prob)>((int)(blockIdx.y * sizeof(L1Prob)));
  constexpr int KB = 128, D = WS_NS - 1;
  constexpr int PER = (WS_BM + WS_BN) / 4 / WS_NL;         // DMA instructions per loader wave and stage (4 rows each): 8
  const L1Prob& P = batch.p[blockIdx.y];
  const int nwg = P.tiles_m * P.tiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int tile_n = lid % P.tiles_n, tile_m = lid / P.tiles_n;
  const int m0 = tile_m * WS_BM, n0 = tile_n * WS_BN;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
  const unsigned lds0 = (unsigned)(size_t)dsmem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt0 = P.K[0] / KB;
  const int nt = nt0 + (P.nseg > 1 ? P.K[1] / KB : 0);

  if (wave >= WS_NC) {
    // ------------------------------------------------------------ loader wave lw: stage rows (j * NL + lw) * 4 .. + 3, j < PER
    const int lw = wave - WS_NC;
    const int q_row = lane >> 4, q_pos = lane & 15;
    unsigned voff[PER];
    auto lane_offsets = [&](int sidx) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int srow = (j * WS_NL + lw) * 4 + q_row;     // row of the [A tile | W tile] stage image
        const bool isa = srow < WS_BM;
        const int row = isa ? srow : srow - WS_BM;
        voff[j] = isa ? (unsigned)(min(m0 + row, P.rows - 1) * (int)P.lda[sidx] * 2 + ((q_pos ^ (row & 15)) << 4))
                      : (unsigned)((n0 + row) * (int)P.ldw1 * 2 + ((q_pos ^ (row & 15)) << 4));
      }
    };
    lane_offsets(0);
    const char* sa_ptr = (const char*)P.A[0];
    const char* sw_ptr = (const char*)P.W1 + (int64_t)P.w1_col[0] * 2;
    int issued = 0;
    auto issue = [&](int stage) {
      if (issued == nt0) {
        lane_offsets(1);
        sa_ptr = (const char*)P.A[1];
        sw_ptr = (const char*)P.W1 + (int64_t)P.w1_col[1] * 2;
      }
      ++issued;
      const unsigned sbase = lds0 + stage * WS_STAGE + lw * 1024;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const bool isa = (j * WS_NL + lw) * 4 < WS_BM;      // (uniform per instruction: 4-row groups never straddle the A / W boundary)
        dma_s(voff[j], isa ? sa_ptr : sw_ptr, sbase + j * (WS_NL * 1024));
      }
      sa_ptr += 256;
      sw_ptr += 256;
    };
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < nt) issue(i);
    for (int t = 0; t < nt; ++t) {
      const int younger = min(D - 1, nt - 1 - t);
      if (younger >= 2) wait_vm_const<2 * PER>();
      else if (younger == 1) wait_vm_const<PER>();
      else wait_vm_const<0>();
      __builtin_amdgcn_s_barrier();                // this wave's part of stage t has landed; the consumers are done with stage t - 1
      if (t + D < nt) issue((t + D) % WS_NS);
    }
    return;
  }

  // -------------------------------------------------------------- consumer wave: rows 32 (wave / 4) .. + 31, columns 16 (wave % 4) .. + 15
  const int wm0 = (wave >> 2) * 32, wn0 = (wave & 3) * 16;
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[2];
  acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // everything the epilogue reads from global memory is requested now (consumers issue no DMA: no vmcnt bookkeeping to disturb)
  const int nb = n0 + wn0 + fg * 4;
  const f32x4 bias = (nb + 3 < P.H) ? *(const f32x4*)(P.b1 + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
  int32_t step_now = 0;
  if (P.mask_mode == RECNN_MASK_HASH && P.step_ptr) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(step_now) : "s"(P.step_ptr));
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this wave's reads of stage t - 1 have returned: its slot may be refilled)
    __builtin_amdgcn_s_barrier();
    const unsigned char* sa = dsmem + (t % WS_NS) * WS_STAGE;
    const unsigned char* sb = sa + WS_BM * 256;
    uint4 a0[4], a1[4], b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int pos = ((ks * 4 + fg) ^ fr) * 16;
      a0[ks] = *(const uint4*)(sa + (wm0 + fr) * 256 + pos);
      a1[ks] = *(const uint4*)(sa + (wm0 + 16 + fr) * 256 + pos);
      b[ks] = *(const uint4*)(sb + (wn0 + fr) * 256 + pos);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[ks]), __builtin_bit_cast(bf16x8, a0[ks]), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[ks]), __builtin_bit_cast(bf16x8, a1[ks]), acc[1], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(step_now));
  const int32_t step0 = step_now + P.step_add;
  // ---- epilogue (l1_gemm_kernel's, per 16 x 16 block): acc[tm][r] = C[row m0 + wm0 + 16 tm + fr][column nb + r]
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm0 + tm * 16 + fr;
    int mrow = m;
    uint32_t key = 0;
    if (P.mask_mode == RECNN_MASK_HASH) {
      int set = 0;
      if (P.rows_per_set > 0) { set = m / P.rows_per_set; mrow = m - set * P.rows_per_set; }
      key = mask_key(P.seed, step0 + set, P.stream);
    }
    uint32_t word = 0;
    if (P.mask_mode == RECNN_MASK_HASH) word = mask_word(key, (uint32_t)(mrow >> 2), (uint32_t)(nb >> 2));
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = acc[tm][r] + bias[r];
      if (!P.no_relu) v[r] = fmaxf(v[r], 0.f);
      if (P.addend && m < P.rows && nb + r < P.H) {
        const float z = P.addend[(int64_t)m * P.ld_add + nb + r];
        v[r] += fminf(fmaxf(z, -P.add_clip), P.add_clip);
      }
      if (P.mask_mode == RECNN_MASK_EXTERNAL) v[r] = (m < P.rows && nb + r < P.H && P.mask[(int64_t)m * P.ld_mask + nb + r]) ? v[r] * 2.f : 0.f;
      else if (P.mask_mode == RECNN_MASK_HASH) v[r] = mask_keep(word, mrow & 3, r) ? v[r] * 2.f : 0.f;
      if (nb + r >= P.H) v[r] = 0.f;
    }
    if (m < P.rows) *(uint2*)((bf16_t*)P.h1 + (int64_t)m * P.ldh + nb) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
  }
}
}  // namespace

// 16 waves as 4 x 4 everywhere (the more waves issue the DMAs, the closer a CU gets to its ~45 B/clk: profiles/NOTES_r01_r05.md 5):
//   SMALL  64 x  64, wave tile 16 x 16, 4-slot ring (128 KB)      per-step launches
//   BIG   128 x 128, wave tile 32 x 32, 2-slot ring (128 KB)      cycle-batched launches
//   MID   128 x  64, wave tile 32 x 16, 3-slot ring (144 KB)      cycle-batched launches (tuning.l1_big = 2): no 2.5-round tail
#define L1_SMALL 1, 1, 4, 4, 4
#define L1_BIG 2, 2, 4, 4, 2
#define L1_MID 2, 1, 4, 4, 3

static unsigned long long* g_l1_trace = nullptr;
extern "C" void recnn_debug_l1_trace(void* p) { g_l1_trace = (unsigned long long*)p; }   // [workgroup][16] uint64 shader-clock stamps

int l1gemm_init() {
  int rc = recnn_check_hip(hipFuncSetAttribute((const void*)l1_gemm_kernel<L1_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<L1_SMALL>()), "l1gemm attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)l1_gemm_kernel<L1_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<L1_BIG>()), "l1gemm attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)l1_gemm_kernel<L1_MID>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<L1_MID>()), "l1gemm attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)l1_gemm_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WS_NS * WS_STAGE), "l1gemm ws attr");
  return rc;
}

int l1gemm_launch(L1Batch& b, int nprob, int big, hipStream_t s, int ws) {
  RECNN_REQUIRE(nprob >= 1 && nprob <= L1_MAX_GROUP, "l1gemm: 1..%d problems per launch", L1_MAX_GROUP);
  const int shape = big == 2 ? 2 : (big ? 1 : 0);   // 0: 64 x 64 per-step tiles, 1: 128 x 128, 2: 128 x 64 (cycle-batched launches)
  const int BM = shape ? 128 : 64, BN = shape == 1 ? 128 : 64;
  int maxwg = 0;
  for (int i = 0; i < nprob; ++i) {
    L1Prob& p = b.p[i];
    RECNN_REQUIRE(p.rows > 0 && p.H > 0 && p.H <= 256 && (p.H & 3) == 0, "l1gemm: bad shape");
    RECNN_REQUIRE(p.nseg >= 1 && p.nseg <= 2 && p.h1 && p.W1 && p.b1, "l1gemm: bad problem");
    for (int g = 0; g < p.nseg; ++g)
      RECNN_REQUIRE(p.K[g] > 0 && p.K[g] % 128 == 0 && p.lda[g] % 8 == 0 && (((uintptr_t)p.A[g]) & 15) == 0 && (p.w1_col[g] & 7) == 0 &&
                        (int64_t)p.rows * p.lda[g] * 2 < (1ll << 31),
                    "l1gemm: segment %d must be 16-byte aligned with K a multiple of 128 (and < 2 GB)", g);
    RECNN_REQUIRE(p.ldw1 % 8 == 0 && p.ldh % 4 == 0 && (((uintptr_t)p.W1 | (uintptr_t)p.h1) & 15) == 0, "l1gemm: bad pitches");
    RECNN_REQUIRE(p.rows_per_set == 0 || p.rows_per_set % 16 == 0, "l1gemm: batches must be multiples of 16 rows");
    p.tiles_m = (p.rows + BM - 1) / BM;
    p.tiles_n = ((p.w_rows > 0 ? p.w_rows : 256) + BN - 1) / BN;     // (the hidden layers' weight shadows are zero-padded to 256 rows)
    const int nwg = p.tiles_m * p.tiles_n;
    if (nwg > maxwg) maxwg = nwg;
  }
  const dim3 grid(maxwg, nprob), block(1024);
  if (shape == 1) hipLaunchKernelGGL((l1_gemm_kernel<L1_BIG>), grid, block, (lds_bytes<L1_BIG>()), s, b, g_l1_trace);
  else if (shape == 2) hipLaunchKernelGGL((l1_gemm_kernel<L1_MID>), grid, block, (lds_bytes<L1_MID>()), s, b, g_l1_trace);
  else if (ws && !g_l1_trace) hipLaunchKernelGGL(l1_gemm_ws_kernel, grid, dim3((WS_NL + WS_NC) * 64), WS_NS * WS_STAGE, s, b);
  else hipLaunchKernelGGL((l1_gemm_kernel<L1_SMALL>), grid, block, (lds_bytes<L1_SMALL>()), s, b, g_l1_trace);
  return recnn_check_hip(hipGetLastError(), "l1_gemm_kernel");
}
