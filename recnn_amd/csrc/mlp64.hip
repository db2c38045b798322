// mlp64.hip -- fused row-panel forward of a whole Actor / Critic MLP, 64-row panels (bf16 compute, gfx950).
//
// Same contract and the same arithmetic, element for element, as mlp.hip (MlpBatch / MlpProb / MlpTail / MlpHead /
// MlpCriticBwd in mlp.h; recnn/nn/models.py:66-73, :207-213 + recnn/nn/update/misc.py:6-7,33-39): every output is
// accumulated in the same k order, so the two kernels agree bit for bit (tests/test_gpu_kernels.py).  What changes is
// the traffic: a workgroup streams a network's weights once per 64 rows instead of once per 32 -- the forward of a DDPG
// step moves 143 MB instead of 256 MB from L2 into LDS, which is what bounds it -- and the
// stream never drains:
//   * one workgroup (16 waves, wave grid 2 x 8, wave tile 32 rows x 32 columns) owns 64 batch rows x all 256 hidden
//     columns of one network;
//   * every weight byte of the workgroup -- layer-1 k-slabs, W2, W3, the chained target critics' W1a / W2, the W2 slabs
//     of the unit backward -- goes through ONE 3-deep ring of 40 KB stages (A 64 x 64k + W 256 x 64k, bf16) filled by
//     global_load_lds_dwordx4 two slabs ahead, with counted vmcnt waits and one raw s_barrier per slab; the slab
//     schedule of the whole workgroup is known up front, so the DMA of the next layer's weights is already in flight
//     while the previous layer's epilogue runs;
//   * LDS: 3 x 40 KB ring + 32 KB activation panel = 152 KB.  Slab rows are 128 bytes; 16-byte chunk c of row r sits at
//     chunk position c ^ ((r >> 1) & 7) (applied on the DMA source address / the panel write address), which makes the
//     ds_read_b128 fragment reads of 16 consecutive rows hit 16 different 16-byte bank slots;
//   * hidden activations go to global memory from the finished LDS panel as whole 512-byte rows (16-byte stores),
//     biases are fetched before the first DMA is issued (a compiler-visible load waited for while DMAs are in flight
//     would drain the ring: its s_waitcnt cannot count them).
// Workgroup -> (problem, panel): map_mode 0 = problem-major (producers of a hand-off are dispatched before its
// consumers), 2 = XCD-contiguous chunks (a network's weights stay in one or two XCDs' L2).
#include "mlp.h"

namespace {
constexpr int BM = 64;            // rows per workgroup
constexpr int HP = 256;           // hidden width handled
constexpr int KS = 64;            // bf16 k elements per slab row (128 bytes)
constexpr int NW = 16;            // waves per workgroup
constexpr int ROWB = KS * 2;
constexpr int A_BYTES = BM * ROWB;            // 8 KB
constexpr int W_BYTES = HP * ROWB;            // 32 KB
constexpr int STAGE = A_BYTES + W_BYTES;      // 40 KB
constexpr int NS = 3;
constexpr int PANEL_OFF = NS * STAGE;         // 120 KB
constexpr int PANEL_Q = BM * ROWB;            // 64 columns of the 64 x 256 activation panel, same image as an A slab
constexpr int BIAS_OFF = PANEL_OFF + 4 * PANEL_Q;    // fp32 biases: b1 | b2 | b3 | tail0 b1 | tail0 b2 | tail1 b1 | tail1 b2, 256 floats each
constexpr int LDS_TOTAL = BIAS_OFF + 7 * 1024;        // 159 KB

typedef short v4s16 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst_uniform))
      : "memory");
}

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// One wave instruction of a 128-byte-row image: rows 8j .. 8j+7 (k slab [k0, k0+64) of a bf16 matrix, pitch ld).
__device__ __forceinline__ void dma_img8(const void* base, int64_t ld, int row0, int row_max, int k0, int j, unsigned img, int lane) {
  const int ir = j * 8 + (lane >> 3);
  const int c = (lane & 7) ^ swz(ir);
  const int gr = min(row0 + ir, row_max);
  dma16((const char*)base + ((int64_t)gr * ld + k0) * 2 + c * 16, img + j * 1024);
}

// One wave instruction of the transposed-use image of a W2 k-slab (the unit backward contracts over W2's ROW index):
// [column half h][k row 0..63][256 bytes = 128 columns], chunk c of row k at position c ^ (k & 15); read back with
// ds_read_b64_tr_b16.
__device__ __forceinline__ void dma_imgT(const void* base, int64_t ld, int k0, int j, unsigned img, int lane) {
  const int h = j >> 4, krow = (j & 15) * 4 + (lane >> 4);
  const int c = (lane & 15) ^ (krow & 15);
  dma16((const char*)base + ((int64_t)(k0 + krow) * ld + h * 128) * 2 + c * 16, img + j * 1024);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// acc[tm][tn] += A(rows ra0 + 16 tm + fr, 64 k) * B(rows nb0 + 16 tn + fr, 64 k)^T for one slab pair already in LDS
template <int TN>
__device__ __forceinline__ void mma_slab(const unsigned char* sa, const unsigned char* sb, f32x4 (&acc)[2][TN], int ra0, int nb0, int fr,
                                         int fg) {
  const int sw = (fr >> 1) & 7;   // ra0, nb0 are multiples of 16
#pragma unroll
  for (int ks = 0; ks < KS / 32; ++ks) {
    const int pos = ((ks * 4 + fg) ^ sw) * 16;
    uint4 a[2], b[TN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (ra0 + tm * 16 + fr) * ROWB + pos);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *(const uint4*)(sb + (nb0 + tn * 16 + fr) * ROWB + pos);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[tm]), __builtin_bit_cast(bf16x8, b[tn]),
                                                              acc[tm][tn], 0, 0, 0);
  }
}

// hidden-layer epilogue: bias + relu + dropout -> bf16 into the LDS panel (the next layer's A operand)
__device__ __forceinline__ uint32_t hidden_epilogue(f32x4 (&acc)[2][2], const float* bias_lds, int H, int rows, int m0, int wm, int wn,
                                                    int fr, int fg, int mask_mode, const uint8_t* mask, int64_t ld_mask, uint32_t key,
                                                    unsigned char* panel) {
  uint32_t bits = 0;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = wn * 32 + tn * 16 + fr;
    const float bvn = bias_lds[n];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int rb = wm * 32 + tm * 16 + fg * 4;
      uint32_t word = 0;
      if (mask_mode == RECNN_MASK_HASH) word = mask_word(key, (uint32_t)((m0 + rb) >> 2), (uint32_t)(n >> 2));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb + r, m = m0 + row;
        float v = fmaxf(acc[tm][tn][r] + bvn, 0.f);
        if (mask_mode == RECNN_MASK_EXTERNAL) v = (m < rows && n < H && mask[(int64_t)m * ld_mask + n]) ? v * 2.f : 0.f;
        else if (mask_mode == RECNN_MASK_HASH) v = mask_keep(word, r, n & 3) ? v * 2.f : 0.f;
        if (n >= H) v = 0.f;
        const bf16_t hv = f2bf(v);
        if (bf2f(hv) > 0.f) bits |= 1u << (tn * 8 + tm * 4 + r);
        // panel image: k quarter n / 64, row, chunk ((n % 64) / 8) ^ swz(row), element n % 8
        *(bf16_t*)(panel + (n >> 6) * PANEL_Q + row * ROWB + ((((n & 63) >> 3) ^ swz(row)) * 16) + (n & 7) * 2) = hv;
      }
    }
  }
  return bits;
}

// the finished panel -> global [rows, ldg] bf16, whole 512-byte rows with 16-byte stores
__device__ __forceinline__ void panel_to_global(const unsigned char* panel, bf16_t* gout, int64_t ldg, int m0, int rows, int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + j * 1024, row = idx >> 5, cc = idx & 31;
    const uint4 v = *(const uint4*)(panel + (cc >> 3) * PANEL_Q + row * ROWB + (((cc & 7) ^ swz(row)) * 16));
    if (m0 + row < rows) *(uint4*)(gout + (int64_t)(m0 + row) * ldg + cc * 8) = v;
  }
}

// q[row] = h2[row, :] . w3  for the 4 rows of this wave (lanes split the 256 columns)
__device__ __forceinline__ float row_dot(const unsigned char* panel, int row, int lane, int H, const float* w3) {
  const int n = lane * 4;
  const uint2 raw = *(const uint2*)(panel + (n >> 6) * PANEL_Q + row * ROWB + ((((n & 63) >> 3) ^ swz(row)) * 16) + (n & 7) * 2);
  const uint32_t u[2] = {raw.x, raw.y};
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16_t hv = (bf16_t)((u[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
    s += (n + j) < H ? bf2f(hv) * w3[n + j] : 0.f;
  }
  return wave_sum(s);
}
}  // namespace

// trace (tools / bench.py RECNN_MLP_TRACE): when non-null, lane 0 of wave 0 of every workgroup stamps the shader clock at the
// phase boundaries into trace[logical workgroup id][16]
#define MLP64_STAMP(i) do { if (trace && (tid == 0 || tid == 960)) trace[(int64_t)bid * 32 + (tid ? 16 : 0) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
__global__ __launch_bounds__(NW * 64) void mlp64_kernel(const MlpBatch batch, int npanel, int map_mode, int probe, unsigned long long* trace) {
  int bid = blockIdx.x;
  if (map_mode == 2) bid = xcd_remap(bid, gridDim.x);
  const int prob = bid / npanel, panel_idx = bid - prob * npanel;
  const MlpProb& P = batch.p[prob];
  const int m0 = panel_idx * BM;
  if (m0 >= P.rows) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int wm = wave >> 3, wn = wave & 7;
  const int row_max = P.rows - 1;
  unsigned char* panel = lds + PANEL_OFF;
  MLP64_STAMP(0);

  // ---- slab schedule of this workgroup
  const int nt0 = P.K[0] / KS;
  const int n1 = nt0 + (P.nseg > 1 ? P.K[1] / KS : 0);
  const bool has_w3 = P.W3 != nullptr;
  const bool do_cbwd = !has_w3 && P.cbwd_idx >= 0 && batch.cbwd[P.cbwd_idx < 0 ? 0 : P.cbwd_idx].enabled;
  const int total = P.part_out ? n1 : n1 + 4 + (has_w3 ? 4 + 6 * P.n_tail : (do_cbwd ? 4 : 0));

  // ---- biases -> LDS BEFORE the first DMA is issued: a compiler-visible global load that is waited for while DMAs are in
  // flight drains the ring (the compiler's s_waitcnt cannot count them); the epilogues read them with ds_read
  float* bias_lds = (float*)(lds + BIAS_OFF);
  if (!P.part_out && tid < HP) {
    const int n = tid;
    bias_lds[n] = n < P.H ? P.b1[n] : 0.f;
    bias_lds[HP + n] = n < P.H ? P.b2[n] : 0.f;
    bias_lds[2 * HP + n] = (has_w3 && n < P.out_dim) ? P.b3[n] : 0.f;
#pragma unroll
    for (int ti = 0; ti < MLP_MAX_TAIL; ++ti)
      if (ti < P.n_tail) {
        bias_lds[(3 + 2 * ti) * HP + n] = n < P.H ? batch.tail[ti].b1[n] : 0.f;
        bias_lds[(4 + 2 * ti) * HP + n] = n < P.H ? batch.tail[ti].b2[n] : 0.f;
      }
  }
  uint32_t key1 = 0, key2 = 0;
  if (P.mask_mode == RECNN_MASK_HASH) {
    const int32_t st = (P.step_ptr ? *P.step_ptr : 0) + P.step_add;
    key1 = mask_key(P.seed, st, P.stream1);
    key2 = mask_key(P.seed, st, P.stream2);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the first slab's barrier publishes the bias area)

  // ---- the ring.  Everything the per-slab path needs is hoisted into registers first: the DMA statements clobber "memory",
  // so any P.* / batch.* field read after them would be re-fetched from the kernel-argument segment every slab.
  const int ir8 = lane >> 3, p8 = lane & 7;
  const int w_row = wave * 8 + ir8;                       // image row of this lane in DMA instruction `wave` (second one: +128)
  const int w_chunk = (p8 ^ swz(w_row)) * 16;             // source chunk behind LDS position p8 (swz(row + 128) == swz(row))
  const unsigned dst_a = wave * 1024, dst_w0 = A_BYTES + wave * 1024, dst_w1 = A_BYTES + (wave + 16) * 1024;
  // per-lane source pointers of a [rows x 128 B] image of matrix `base` (pitch ld elements) at k offset k0
  struct Src { const char* p0; const char* p1; };
  auto w_src = [&](const void* base, int64_t ld, int k0) -> Src {
    const char* q = (const char*)base + ((int64_t)w_row * ld + k0) * 2 + w_chunk;
    return Src{q, q + ld * 256};                          // 128 rows further down
  };
  auto a_src = [&](const void* base, int64_t ld) -> const char* {
    return (const char*)base + (int64_t)min(m0 + w_row, row_max) * ld * 2 + w_chunk;
  };
  const int nseg = P.nseg;
  const char* a_ptr = a_src(P.A[0], P.lda[0]);
  const char* a_ptr1 = nseg > 1 ? a_src(P.A[1], P.lda[1]) : nullptr;
  Src w1 = w_src(P.W1, P.ldw1, P.w1_col[0]);
  const Src w1b = nseg > 1 ? w_src(P.W1, P.ldw1, P.w1_col[1]) : w1;
  if (probe & 1) a_ptr = (const char*)P.A[0] + (int64_t)min(w_row, row_max) * P.lda[0] * 2 + w_chunk;
  const void* const W2p = P.W2; const int64_t ldw2 = P.ldw2;
  const void* const W3p = P.W3; const int64_t ldw3 = P.ldw3;
  Src cur = w1;                                           // pointer pair of the weight matrix being streamed after layer 1
  const char* curT = nullptr;                             // ... of the transposed-use W2 image (unit backward)
  int head = 0, issued = 0, st_issue = 0, st_head = 0;
  auto stage_of_issue = [&]() -> unsigned {
    const unsigned st = __builtin_amdgcn_readfirstlane(lds0 + st_issue * STAGE);
    st_issue = st_issue == NS - 1 ? 0 : st_issue + 1;
    ++issued;
    return st;
  };
  auto issue_l1 = [&]() {                                 // layer-1 slab `issued`: A panel + W1 k-slab
    if (issued == nt0) { a_ptr = a_ptr1; w1 = w1b; }      // second contraction segment
    const unsigned st = stage_of_issue();
    if (wave < 8 && !(probe & 4)) dma16(a_ptr, st + dst_a);
    if (!(probe & 2)) { dma16(w1.p0, st + dst_w0); dma16(w1.p1, st + dst_w1); }
    else { dma16(w1.p0, st + dst_w0); }
    a_ptr += ROWB; w1.p0 += ROWB; w1.p1 += ROWB;
  };
  auto issue_w = [&]() {                                  // a 256-row weight slab from `cur`
    const unsigned st = stage_of_issue();
    dma16(cur.p0, st + dst_w0);
    dma16(cur.p1, st + dst_w1);
    cur.p0 += ROWB; cur.p1 += ROWB;
  };
  auto issue_w3 = [&]() {                                 // a 128-row slab (actor output layer): one instruction per wave
    const unsigned st = stage_of_issue();
    dma16(cur.p0, st + dst_w0);
    cur.p0 += ROWB;
  };
  auto issue_t = [&]() {                                  // transposed-use image of 64 W2 rows
    const unsigned st = stage_of_issue();
    dma16(curT, st + dst_w0);
    dma16(curT + 256, st + dst_w1);                       // column half 1
    curT += ldw2 * 2 * KS;
  };
  // slab `issued` of the post-layer-1 sequence: [W2 x4] then actor: [W3 x4] [tail: W1a x2, W2 x4]...; critic: [W2^T x4]
  auto refill = [&]() {
    if (issued >= total) return;
    int j = issued - n1;
    if (j < 0) { issue_l1(); return; }
    if (j < 4) {
      if (j == 0) cur = w_src(W2p, ldw2, 0);
      issue_w();
    } else if (has_w3) {
      j -= 4;
      if (j < 4) {
        if (j == 0) cur = w_src(W3p, ldw3, 0);
        issue_w3();
      } else {
        j -= 4;
        const int ti = j >= 6 ? 1 : 0, r = j - ti * 6;
        if (r == 0) cur = w_src(batch.tail[ti].W1a, batch.tail[ti].ldw1, 0);
        else if (r == 2) cur = w_src(batch.tail[ti].W2, batch.tail[ti].ldw2, 0);
        issue_w();
      }
    } else {
      j -= 4;
      if (j == 0) {
        const int krow = (wave & 15) * 4 + (lane >> 4);   // DMA instruction `wave`: column half 0, k rows 4 wave .. 4 wave + 3
        curT = (const char*)W2p + (int64_t)krow * ldw2 * 2 + (((lane & 15) ^ (krow & 15)) * 16);
      }
      issue_t();
    }
  };
  // DMA instructions this wave issues for slab i
  auto dma_count = [&](int i) -> int {
    if (i < n1) return (wave < 8 && !(probe & 4)) ? ((probe & 2) ? 2 : 3) : ((probe & 2) ? 1 : 2);
    if (has_w3 && i >= n1 + 4 && i < n1 + 8) return 1;
    return 2;
  };
  // wait_slab(): waits for slab `head` (leaving slab head+1 in flight) and passes the workgroup barrier behind which the stage
  // of slab head-1 is free; refill(): issues slab head+2 into that stage.  Compiler-visible global stores of a phase go
  // BETWEEN the two, so that the next counted wait (which cannot count them) finds them older than the slab it leaves in flight.
  auto wait_slab = [&]() -> const unsigned char* {
    const int c = issued - head > 1 ? dma_count(head + 1) : 0;
    if (c == 0) wait_vm<0>();
    else if (c == 1) wait_vm<1>();
    else if (c == 2) wait_vm<2>();
    else wait_vm<3>();
    __builtin_amdgcn_s_barrier();
    const unsigned char* st = lds + st_head * STAGE;
    ++head;
    st_head = st_head == NS - 1 ? 0 : st_head + 1;
    return st;
  };
  MLP64_STAMP(1);
  issue_l1();
  issue_l1();

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ------------------------------------------------------------------ layer 1
  for (int t = 0; t < n1 - 2; ++t) {   // steady state: slab t + 2 is another layer-1 slab
    if (t == 2) MLP64_STAMP(10);
    if (t == 12) MLP64_STAMP(11);
    const unsigned char* st = wait_slab();
    if (t == 12) MLP64_STAMP(12);
    issue_l1();
    mma_slab<2>(st, st + A_BYTES, acc, wm * 32, wn * 32, fr, fg);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {        // last two slabs: the refill is the next layer's first weight slabs (nothing for a producer)
    const unsigned char* st = wait_slab();
    refill();
    mma_slab<2>(st, st + A_BYTES, acc, wm * 32, wn * 32, fr, fg);
  }
  MLP64_STAMP(2);
  if (P.part_out) {
    // producer of a chained critic: hand the raw pre-activation part to the consumer workgroup of this panel
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = wn * 32 + tn * 16 + fr;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) P.part_out[(int64_t)(m0 + wm * 32 + tm * 16 + fg * 4 + r) * HP + n] = acc[tm][tn][r];
    }
    __syncthreads();  // every thread's stores have completed (the barrier is preceded by s_waitcnt vmcnt(0))
    if (tid == 0 && batch.fault != 1) __hip_atomic_store(P.part_flag + panel_idx, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    MLP64_STAMP(9);
    return;
  }
  const uint32_t gate1 = hidden_epilogue(acc, bias_lds, P.H, P.rows, m0, wm, wn, fr, fg, P.mask_mode, P.mask1, P.ld_mask, key1, panel);

  MLP64_STAMP(3);
  // ------------------------------------------------------------------ layer 2
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j == 2) MLP64_STAMP(13);
    const unsigned char* st = wait_slab();   // (first pass: the h1 panel is complete behind this barrier)
    if (j == 2) MLP64_STAMP(14);
    if (j == 0 && P.h1 && !(probe & 8)) panel_to_global(panel, (bf16_t*)P.h1, P.ldh, m0, P.rows, tid);
    refill();
    mma_slab<2>(panel + j * PANEL_Q, st + A_BYTES, acc, wm * 32, wn * 32, fr, fg);
  }
  MLP64_STAMP(4);
  // chained critics: wait for the producers' layer-1 parts (the fetch itself follows epilogue 2, straight into the dead accumulators)
  if (has_w3 && P.n_tail) {
    if (tid == 0) {
      // bounded spin: a producer has a lower logical workgroup id and never waits itself; a wait that runs out is
      // REPORTED (batch.err -> RECNN_E_STATE at the next loss / counter read), never silently computed through
      const int limit = batch.spin_limit > 0 ? batch.spin_limit : (1 << 22);
      for (int ti = 0; ti < P.n_tail; ++ti) {
        int spins = 0;
        bool ok;
        while (!(ok = __hip_atomic_load(batch.tail[ti].flag + panel_idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0) && ++spins < limit)
          __builtin_amdgcn_s_sleep(2);
        if (!ok && batch.err) __hip_atomic_fetch_or(batch.err, MLP_ERR_PART_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(batch.tail[ti].flag + panel_idx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  MLP64_STAMP(15);
  __builtin_amdgcn_s_barrier();   // everyone is done with the h1 panel; flags seen (tid 0's acquire dropped this CU's stale lines)
  hidden_epilogue(acc, bias_lds + HP, P.H, P.rows, m0, wm, wn, fr, fg, P.mask_mode, P.mask2, P.ld_mask, key2, panel);

  MLP64_STAMP(5);
  auto load_part = [&](const float* part) {   // the accumulators start from the producer's fp32 layer-1 state part
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = wn * 32 + tn * 16 + fr;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tm][tn][r] = part[(int64_t)(m0 + wm * 32 + tm * 16 + fg * 4 + r) * HP + n];
    }
  };
  if (has_w3) {
    if (P.n_tail) load_part(batch.tail[0].part);   // in flight under layer 3
    // ---------------------------------------------------------------- layer 3 (actor): 64 x 128 outputs
    f32x4 o[2][1];
    o[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    o[1][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned char* st = wait_slab();   // (first pass: the h2 panel is complete)
      if (j == 0 && P.h2 && !(probe & 8)) panel_to_global(panel, (bf16_t*)P.h2, P.ldh, m0, P.rows, tid);
      refill();
      mma_slab<1>(panel + j * PANEL_Q, st + A_BYTES, o, wm * 32, wn * 16, fr, fg);
    }
    {
      const int n = wn * 16 + fr;
      const bool ncol = n < P.out_dim;
      const float b3n = bias_lds[2 * HP + n];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * 32 + tm * 16 + fg * 4 + r, m = m0 + row;
          float v = o[tm][0][r] + b3n;
          if (P.addend && ncol && m < P.rows) {
            const float z = P.addend[(int64_t)m * P.ld_add + n];
            v += fminf(fmaxf(z, -P.add_clip), P.add_clip);
          }
          const bf16_t hv = ncol ? f2bf(v) : (bf16_t)0;
          if (ncol && m < P.rows && !(probe & 8)) ((bf16_t*)P.out)[(int64_t)m * P.ldo + n] = hv;
          // chained critics read the action panel from the (idle since layer 1) A slots of ring stages 0 and 1: two 64-k slabs
          if (P.n_tail) *(bf16_t*)(lds + (n >> 6) * STAGE + row * ROWB + ((((n & 63) >> 3) ^ swz(row)) * 16) + (n & 7) * 2) = hv;
        }
    }
    MLP64_STAMP(6);
    // ---------------------------------------------------------------- chained critics (target critic on the new action)
#pragma unroll
    for (int ti = 0; ti < MLP_MAX_TAIL; ++ti) {
      if (ti >= P.n_tail) break;
      const MlpTail& T = batch.tail[ti];
      if (ti > 0) load_part(T.part);
#pragma unroll
      for (int j = 0; j < 2; ++j) {        // + action panel x W1a (behind the first barrier: action panel written, panel reads done)
        const unsigned char* st = wait_slab();
        refill();
        mma_slab<2>(lds + j * STAGE, st + A_BYTES, acc, wm * 32, wn * 32, fr, fg);
      }
      hidden_epilogue(acc, bias_lds + (3 + 2 * ti) * HP, P.H, P.rows, m0, wm, wn, fr, fg, RECNN_MASK_NONE, nullptr, 0, 0u, panel);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned char* st = wait_slab();
        refill();
        mma_slab<2>(panel + j * PANEL_Q, st + A_BYTES, acc, wm * 32, wn * 32, fr, fg);
      }
      __builtin_amdgcn_s_barrier();  // everyone is done reading the h1 panel
      hidden_epilogue(acc, bias_lds + (4 + 2 * ti) * HP, P.H, P.rows, m0, wm, wn, fr, fg, RECNN_MASK_NONE, nullptr, 0, 0u, panel);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < BM / NW; ++i) {
        const int row = wave * (BM / NW) + i;
        const float sdot = row_dot(panel, row, lane, P.H, T.w3row);
        if (lane == 0) {
          const float qv = sdot + T.b3[0];
          if (m0 + row < P.rows) T.q[m0 + row] = qv;
          ((float*)(lds + 2 * STAGE))[ti * BM + row] = qv;   // for the head below (A slot of ring stage 2 is idle)
        }
      }
    }
    MLP64_STAMP(7);
    // ---------------------------------------------------------------- head of the learning critic(s)
    if (batch.head.n_critic > 0 && P.n_tail > 0) {
      const MlpHead& Hd = batch.head;
      __syncthreads();   // Q' of all 64 rows (every tail) is in LDS
      if (wave == 0) {
        const float* stq = (const float*)(lds + 2 * STAGE);
        const int r = lane, m = m0 + r, mc = min(m, P.rows - 1);
        const bool valid = m < P.rows;
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
            // Q(s, a) from the critic workgroup of the same rows (value-as-flag slot), then the slot goes back to rest
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
            // one partial per 32 rows (the consumers -- loss_finalize, Adam's b3 slab sum -- count 32-row panels)
            const float tot = half_sum32(e * e);
            const float dsum = half_sum32(d);
            if ((lane == 31 || lane == 63) && m0 + (lane >> 5) * 32 < P.rows) {
              const int pi = 2 * panel_idx + (lane >> 5);
              if (Hd.loss_part[c]) Hd.loss_part[c][pi] = tot;
              if (Hd.db3_part[c]) Hd.db3_part[c][pi] = dsum;
            }
          }
        }
      }
    }
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // h2 panel complete
    if (P.h2 && !(probe & 8)) panel_to_global(panel, (bf16_t*)P.h2, P.ldh, m0, P.rows, tid);
  }
  if (!has_w3 && P.q) {
    // ---------------------------------------------------------------- critic head: q[m] = h2[m, :] . w3 + b3
#pragma unroll
    for (int i = 0; i < BM / NW; ++i) {
      const int row = wave * (BM / NW) + i;
      const float s = row_dot(panel, row, lane, P.H, P.w3row);
      if (lane == 0 && m0 + row < P.rows) {
        const float qv = s + P.b3[0];
        P.q[m0 + row] = qv;
        if (P.cbwd_idx >= 0 && batch.cbwd[P.cbwd_idx].q_slot && batch.fault != 2)   // hand Q(s, a) to the head's workgroup (value = flag)
          __hip_atomic_store((uint32_t*)batch.cbwd[P.cbwd_idx].q_slot + m0 + row, __builtin_bit_cast(uint32_t, qv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    MLP64_STAMP(6);
    if (do_cbwd) {
      const MlpCriticBwd& B = batch.cbwd[P.cbwd_idx];
      __builtin_amdgcn_s_barrier();   // every wave is done reading h2 rows for its q dots
      // ---- u2 = w3 * scale * [h2 > 0], in place in the panel (it becomes the A operand) and to global
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = tid + j * 1024, row = idx >> 5, cc = idx & 31, n8 = cc * 8, m = m0 + row;
        const int nb = min(n8, P.H - 8);
        const float4 w3a = *(const float4*)(P.w3row + nb), w3b = *(const float4*)(P.w3row + nb + 4);
        const float wsc = n8 < P.H ? B.scale : 0.f;
        const float w3v[8] = {w3a.x * wsc, w3a.y * wsc, w3a.z * wsc, w3a.w * wsc, w3b.x * wsc, w3b.y * wsc, w3b.z * wsc, w3b.w * wsc};
        unsigned char* cell = panel + (cc >> 3) * PANEL_Q + row * ROWB + (((cc & 7) ^ swz(row)) * 16);
        const uint4 raw = *(const uint4*)cell;
        const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
        float uz[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float hv = bf2f((bf16_t)((u[e >> 1] >> ((e & 1) * 16)) & 0xFFFF));
          uz[e] = hv > 0.f ? w3v[e] : 0.f;
        }
        const uint4 packed = make_uint4(pack_bf2(uz[0], uz[1]), pack_bf2(uz[2], uz[3]), pack_bf2(uz[4], uz[5]), pack_bf2(uz[6], uz[7]));
        *(uint4*)cell = packed;
        if (m < P.rows && !(probe & 8)) *(uint4*)((bf16_t*)B.dz2 + (int64_t)m * P.ldh + n8) = packed;
      }
      // ---- U = (u2 W2) * scale * gate(h1): W2 k-slabs in the transposed-use image, B fragments by transpose reads
      f32x4 dacc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int sw = (fr >> 1) & 7;
#pragma unroll
      for (int qk = 0; qk < 4; ++qk) {
        const unsigned char* st = wait_slab();   // (first pass: the u2 panel is complete)
        refill();
        const unsigned char* wimg = st + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int pos = ((ks * 4 + fg) ^ sw) * 16;
          uint4 a[2];
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(panel + qk * PANEL_Q + (wm * 32 + tm * 16 + fr) * ROWB + pos);
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            const int n0 = wn * 32 + tn * 16;
            const unsigned char* wh = wimg + (n0 >> 7) * (KS * 256);
            const int cpair = (n0 & 127) >> 3;
            v4s16 b[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const int k = ks * 32 + fg * 8 + half * 4 + (fr >> 2);
              b[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (__attribute__((address_space(3))) v4s16*)(wh + k * 256 + (((cpair + ((fr & 3) >> 1)) ^ (k & 15)) * 16) + (fr & 1) * 8));
            }
            struct { v4s16 lo, hi; } bvv = {b[0], b[1]};
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
              dacc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[tm]), __builtin_bit_cast(bf16x8, bvv),
                                                                     dacc[tm][tn], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = wn * 32 + tn * 16 + fr;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int mm = m0 + wm * 32 + tm * 16 + fg * 4 + r;
            if (mm < P.rows && n < P.H) {
              const float v = ((gate1 >> (tn * 8 + tm * 4 + r)) & 1u) ? dacc[tm][tn][r] * B.scale : 0.f;
              ((bf16_t*)B.dz1)[(int64_t)mm * P.ldh + n] = f2bf(v);
            }
          }
      }
    }
  }
  MLP64_STAMP(9);
}

static int g_mlp_map = 0;
extern "C" void recnn_tune_mlp_map(int mode) { g_mlp_map = mode == 2 ? 2 : 0; }
static unsigned long long* g_mlp_trace = nullptr;
void mlpr_set_trace(void* p);
void mlp32_set_trace(void* p);
void mlps_set_trace(void* p);
extern "C" void recnn_tune_mlp_trace(void* device_u64_wg16) {
  g_mlp_trace = (unsigned long long*)device_u64_wg16;
  mlpr_set_trace(device_u64_wg16);
  mlp32_set_trace(device_u64_wg16);
  mlps_set_trace(device_u64_wg16);
}
int mlp64_map_mode();
int mlp64_map_mode() { return g_mlp_map; }
static int g_mlp_probe = 0;
extern "C" void recnn_tune_mlp_probe(int bits) { g_mlp_probe = bits; }   // timing experiments (see issue() in the kernel)

int mlp64_init() {
  return recnn_check_hip(hipFuncSetAttribute((const void*)mlp64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL),
                         "mlp64_kernel attr");
}

// Preconditions are those of mlp_launch (checked there).
int mlp64_launch(const MlpBatch& b, int nprob, int rows, hipStream_t s) {
  for (int i = 0; i < nprob; ++i) {
    const MlpProb& p = b.p[i];
    for (int g = 0; g < p.nseg; ++g)
      if (p.K[g] % KS) { recnn_set_error("mlp64: k extents must be multiples of 64"); return RECNN_E_INVALID; }
    if (p.ldh != HP && (p.h1 || p.h2 || p.cbwd_idx >= 0)) { recnn_set_error("mlp64: hidden activations must have pitch 256"); return RECNN_E_INVALID; }
  }
  const int npanel = (rows + BM - 1) / BM;
  hipLaunchKernelGGL(mlp64_kernel, dim3(npanel * nprob), dim3(NW * 64), LDS_TOTAL, s, b, npanel, g_mlp_map, g_mlp_probe, g_mlp_trace);
  return recnn_check_hip(hipGetLastError(), "mlp64_kernel");
}
