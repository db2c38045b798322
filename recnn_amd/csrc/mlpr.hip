// mlpr.hip -- fused row-panel forward of a whole Actor / Critic MLP; weights straight from L2 into MFMA registers.
//
// Same contract and the same arithmetic, element for element and in the same k order, as mlp.hip / mlp64.hip (MlpBatch /
// MlpProb / MlpTail / MlpHead / MlpCriticBwd in mlp.h; recnn/nn/models.py:66-73, :207-213 + recnn/nn/update/misc.py:6-7,
// 33-39), so the three kernels agree bit for bit.  What changes is how the operands travel (the in-kernel
// phase trace of mlp64.hip showed its time going to the per-slab rendezvous -- every wave waits for every other wave's DMA
// and fragment reads once per 64 k -- and to DMA latency at every layer boundary, not to bandwidth):
//   * a workgroup = 8 waves owns 64 batch rows x all 256 hidden columns of one network; wave w owns columns 32w .. 32w+31
//     of ALL 64 rows (4 x 2 MFMA tiles, 16x16x32 bf16), so every weight row is needed by exactly ONE wave:
//   * the B operand (weights) never touches LDS: each lane loads its own 16 bytes per MFMA k-step with plain global loads,
//     8 k-slabs (128 VGPRs, 256 KB per CU) ahead of their use; the compiler counts those loads itself (no inline-asm DMA in
//     this kernel, hence no hand-counted vmcnt), and the next layer's whole weight slice is requested before the previous
//     layer's epilogue starts;
//   * the A operand (batch rows, shared by all waves) is staged global -> registers -> LDS one 4-slab group ahead into an
//     8-slab ring: ONE workgroup barrier per 256 k instead of one per 64;
//   * LDS: 64 KB A ring + 32 KB activation panel + 16 KB action panel + 8 KB biases / scalars = 120 KB.  Slab rows are 128
//     bytes, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7): the ds_read_b128 fragment reads of 16 consecutive rows
//     hit 16 different bank slots.
// Layer-1 k extents that are not a multiple of 512 are padded with zero A slabs (they add exact zeros).
#include "mlp.h"

namespace {
constexpr int BM = 64, TM = BM / 16, HP = 256, KS = 64, NW = 8, NT = NW * 64;
constexpr int ROWB = KS * 2;                  // 128-byte slab rows
constexpr int SLAB = BM * ROWB;               // 8 KB: one 64-k slab of the A operand / of the activation panel
constexpr int RING_OFF = 0, RING = 8 * SLAB;  // 64 KB
constexpr int PANEL_OFF = RING_OFF + RING;    // 32 KB
constexpr int ACT_OFF = PANEL_OFF + 4 * SLAB; // 16 KB: the target actor's action panel (A operand of the chained critics)
constexpr int BIAS_OFF = ACT_OFF + 2 * SLAB;  // fp32: b1 | b2 | b3 | tail0 b1 | tail0 b2 | tail1 b1 | tail1 b2 (256 floats each)
constexpr int STQ_OFF = BIAS_OFF + 7 * 1024;  // Q' of the chained critics, [2][BM] floats
constexpr int LDS_TOTAL = STQ_OFF + 1024;     // 120 KB

typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst_uniform))
      : "memory");
}
// workgroup barrier that leaves this wave's global loads in flight (LDS traffic of the wave is drained first)
__device__ __forceinline__ void wg_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// B operand of one k-slab for this wave's TN column tiles: bq[tn][s] = W[n0 + 16 tn + fr][k0 + 32 s + 8 fg .. +7]
template <int TN> struct BSlab { uint4 q[TN][2]; };
template <int TN> __device__ __forceinline__ void load_b(BSlab<TN>& b, const char* const (&p)[TN], int64_t off) {
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    b.q[tn][0] = *(const uint4*)(p[tn] + off);
    b.q[tn][1] = *(const uint4*)(p[tn] + off + 64);
  }
}
// per-lane row pointers of a [rows x K] bf16 weight matrix for this wave's column tiles
template <int TN> __device__ __forceinline__ void b_rows(const char* (&p)[TN], const void* base, int64_t ld, int col0, int n0, int fr, int fg) {
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) p[tn] = (const char*)base + ((int64_t)(n0 + tn * 16 + fr) * ld + col0 + 8 * fg) * 2;
}

// acc[tm][tn] += A(slab image `sa`, rows 16 tm + fr) * B-slab
template <int TN>
__device__ __forceinline__ void mma_slab(const unsigned char* sa, const BSlab<TN>& b, f32x4 (&acc)[TM][TN], int fr, int fg) {
  const int sw = (fr >> 1) & 7;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int pos = ((s * 4 + fg) ^ sw) * 16;
    uint4 a[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * ROWB + pos);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[tm]), __builtin_bit_cast(bf16x8, b.q[tn][s]),
                                                              acc[tm][tn], 0, 0, 0);
  }
}

// hidden-layer epilogue: bias + relu + dropout -> bf16 into the LDS panel (the next layer's A operand); returns the relu/dropout
// gate bits of this lane's elements (bit tn*16 + tm*4 + r)
__device__ __forceinline__ uint32_t hidden_epilogue(f32x4 (&acc)[TM][2], const float* bias_lds, int H, int rows, int m0, int wave, int fr,
                                                    int fg, int mask_mode, const uint8_t* mask, int64_t ld_mask, uint32_t key,
                                                    unsigned char* panel) {
  uint32_t bits = 0;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = wave * 32 + tn * 16 + fr;
    const float bvn = bias_lds[n];
    unsigned char* col = panel + (n >> 6) * SLAB + (n & 7) * 2;
    const int c = (n & 63) >> 3;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int rb = tm * 16 + fg * 4;
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
        if (bf2f(hv) > 0.f) bits |= 1u << (tn * 16 + tm * 4 + r);
        *(bf16_t*)(col + row * ROWB + ((c ^ swz(row)) * 16)) = hv;
      }
    }
  }
  return bits;
}

// the finished panel -> global [rows, ldg] bf16, whole 512-byte rows with 16-byte stores
__device__ __forceinline__ void panel_to_global(const unsigned char* panel, bf16_t* gout, int64_t ldg, int m0, int rows, int tid) {
#pragma unroll
  for (int j = 0; j < BM * 32 / NT; ++j) {
    const int idx = tid + j * NT, row = idx >> 5, cc = idx & 31;
    const uint4 v = *(const uint4*)(panel + (cc >> 3) * SLAB + row * ROWB + (((cc & 7) ^ swz(row)) * 16));
    if (m0 + row < rows) *(uint4*)(gout + (int64_t)(m0 + row) * ldg + cc * 8) = v;
  }
}

__device__ __forceinline__ float row_dot(const unsigned char* panel, int row, int lane, int H, const float* w3) {
  const int n = lane * 4;
  const uint2 raw = *(const uint2*)(panel + (n >> 6) * SLAB + row * ROWB + ((((n & 63) >> 3) ^ swz(row)) * 16) + (n & 7) * 2);
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

#define MLPR_STAMP(i) do { if (trace && tid == 0) trace[(int64_t)bid * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

__global__ __launch_bounds__(NT) void mlpr_kernel(const MlpBatch batch, int npanel, int map_mode, u64* trace) {
  int bid = blockIdx.x;
  if (map_mode == 2) bid = xcd_remap(bid, gridDim.x);
  const int prob = bid / npanel, panel_idx = bid - prob * npanel;
  const MlpProb& P = batch.p[prob];
  const int m0 = panel_idx * BM;
  if (m0 >= P.rows) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int row_max = P.rows - 1;
  unsigned char* const ring = lds + RING_OFF;
  unsigned char* const panel = lds + PANEL_OFF;
  float* const bias_lds = (float*)(lds + BIAS_OFF);
  MLPR_STAMP(0);

  const bool has_w3 = P.W3 != nullptr;
  const bool producer = P.part_out != nullptr;
  const bool do_cbwd = !has_w3 && P.cbwd_idx >= 0 && batch.cbwd[P.cbwd_idx < 0 ? 0 : P.cbwd_idx].enabled;
  const int n_tail = has_w3 ? P.n_tail : 0;
  const int nt0 = P.K[0] / KS;
  const int n1 = nt0 + (P.nseg > 1 ? P.K[1] / KS : 0);
  const int npair = (n1 + 7) / 8;              // layer 1 runs over npair x 8 slabs; slabs >= n1 carry a zero A operand

  // ---- biases -> LDS
  if (!producer && tid < HP) {
    const int n = tid;
    bias_lds[n] = n < P.H ? P.b1[n] : 0.f;
    bias_lds[HP + n] = n < P.H ? P.b2[n] : 0.f;
    bias_lds[2 * HP + n] = (has_w3 && n < P.out_dim) ? P.b3[n] : 0.f;
#pragma unroll
    for (int ti = 0; ti < MLP_MAX_TAIL; ++ti)
      if (ti < n_tail) {
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

  // ---- A operand: LDS-DMA, one 4-slab group ahead.  Wave w issues instruction w of each slab (rows 8w .. 8w+7, 128 B each).
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int a_row = wave * 8 + (lane >> 3);
  const int a_off = ((lane & 7) ^ swz(a_row)) * 16;      // source chunk behind LDS position (lane & 7)
  const char* const a_p0 = (const char*)P.A[0] + (int64_t)min(m0 + a_row, row_max) * P.lda[0] * 2 + a_off;
  const char* const a_p1 = P.nseg > 1 ? (const char*)P.A[1] + (int64_t)min(m0 + a_row, row_max) * P.lda[1] * 2 + a_off : a_p0;
  auto dma_a_group = [&](int g) {                         // slabs 4g .. 4g+3 into ring half g & 1 (slabs >= n1: nothing)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = g * 4 + u;
      if (t < n1) dma16(t < nt0 ? a_p0 + t * ROWB : a_p1 + (t - nt0) * ROWB, lds0 + RING_OFF + ((g & 1) * 4 + u) * SLAB + wave * 1024);
    }
  };

  // ---- B of layer 1: this wave's 32 W1 rows; slab t of segment sg at column w1_col[sg] + 64 t
  const char* w1p0[2];
  const char* w1p1[2];
  b_rows<2>(w1p0, P.W1, P.ldw1, P.w1_col[0], wave * 32, fr, fg);
  b_rows<2>(w1p1, P.W1, P.ldw1, P.nseg > 1 ? P.w1_col[1] : P.w1_col[0], wave * 32, fr, fg);
  const void* const W2p = P.W2; const int64_t ldw2 = P.ldw2;
  const void* const W3p = P.W3; const int64_t ldw3 = P.ldw3;
  const char* wp[2];                                      // W2 rows of this wave (a producer has none: any valid address will do)
  b_rows<2>(wp, producer ? P.W1 : W2p, producer ? P.ldw1 : ldw2, 0, wave * 32, fr, fg);
  // slab v of the B stream: layer-1 slabs 0 .. 8 npair - 1 (clamped to n1 - 1), then the four W2 slabs (twice): the reloads
  // of the last pair of groups fetch the NEXT layer's weights, so every group issues the same number of loads -- the
  // compiler's counted waits and the hand-counted one below rely on that
  auto load_stream = [&](BSlab<2>& b, int v) {
    const int tc = min(v, n1 - 1);
    const bool l1 = v < npair * 8;
    const bool seg0 = tc < nt0;
    const int64_t off = l1 ? (int64_t)(seg0 ? tc : tc - nt0) * ROWB : (int64_t)((v - npair * 8) & 3) * ROWB;
    const char* q[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) q[tn] = l1 ? (seg0 ? w1p0[tn] : w1p1[tn]) : wp[tn];
    load_b<2>(b, q, off);
  };

  BSlab<2> bq[8];
  dma_a_group(0);
#pragma unroll
  for (int u = 0; u < 8; ++u) load_stream(bq[u], u);
  MLPR_STAMP(1);

  f32x4 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ------------------------------------------------------------------ layer 1: pairs of 4-slab groups, B ring = 8 slabs
  asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");   // A(0) landed: 32 B loads were issued after its DMAs
  __builtin_amdgcn_s_barrier();
  for (int gp = 0; gp < npair; ++gp) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int g = gp * 2 + h;
      dma_a_group(g + 1);                    // into the other ring half: its last readers passed the barrier above / below
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (g * 4 + u < n1) mma_slab<2>(ring + (h * 4 + u) * SLAB, bq[h * 4 + u], acc, fr, fg);
        load_stream(bq[h * 4 + u], g * 4 + u + 8);
      }
      asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");   // A(g + 1) landed: 16 B loads were issued after its DMAs
      __builtin_amdgcn_s_barrier();          // ... for every wave; and every wave is done reading A(g)
    }
  }
  MLPR_STAMP(2);
  if (producer) {
    // producer of a chained critic: hand the raw pre-activation part to the consumer workgroup of this panel
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = wave * 32 + tn * 16 + fr;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) P.part_out[(int64_t)(m0 + tm * 16 + fg * 4 + r) * HP + n] = acc[tm][tn][r];
    }
    __syncthreads();  // every thread's stores have completed (the barrier is preceded by s_waitcnt vmcnt(0))
    if (tid == 0 && batch.fault != 1) __hip_atomic_store(P.part_flag + panel_idx, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    MLPR_STAMP(9);
    return;
  }
  // (the W2 slice of this wave, 4 slabs, is already in flight into bq[0..3]: see load_stream)
  const uint32_t gate1 = hidden_epilogue(acc, bias_lds, P.H, P.rows, m0, wave, fr, fg, P.mask_mode, P.mask1, P.ld_mask, key1, panel);
  MLPR_STAMP(3);

  // ------------------------------------------------------------------ layer 2
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  wg_sync();                           // h1 panel complete
  if (P.h1) panel_to_global(panel, (bf16_t*)P.h1, P.ldh, m0, P.rows, tid);
#pragma unroll
  for (int u = 0; u < 4; ++u) mma_slab<2>(panel + u * SLAB, bq[u], acc, fr, fg);
  MLPR_STAMP(4);
  // next weights: actor W3 (16 rows per wave) / nothing for a critic (its unit backward stages W2 through LDS)
  BSlab<1> b3[4];
  const char* w3r[1] = {nullptr};
  if (has_w3) {
    b_rows<1>(w3r, W3p, ldw3, 0, wave * 16, fr, fg);
#pragma unroll
    for (int u = 0; u < 4; ++u) load_b<1>(b3[u], w3r, (int64_t)u * ROWB);
  }
  // chained critics: wait for the producers' layer-1 parts
  if (n_tail) {
    if (tid == 0) {
      // bounded spin: a producer has a lower logical workgroup id and never waits itself; a wait that runs out is
      // REPORTED (batch.err -> RECNN_E_STATE at the next loss / counter read), never silently computed through
      const int limit = batch.spin_limit > 0 ? batch.spin_limit : (1 << 22);
      for (int ti = 0; ti < n_tail; ++ti) {
        int spins = 0;
        bool ok;
        while (!(ok = __hip_atomic_load(batch.tail[ti].flag + panel_idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0) && ++spins < limit)
          __builtin_amdgcn_s_sleep(2);
        if (!ok && batch.err) __hip_atomic_fetch_or(batch.err, MLP_ERR_PART_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(batch.tail[ti].flag + panel_idx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  wg_sync();   // everyone is done with the h1 panel; flags seen (tid 0's acquire dropped this CU's stale lines)
  hidden_epilogue(acc, bias_lds + HP, P.H, P.rows, m0, wave, fr, fg, P.mask_mode, P.mask2, P.ld_mask, key2, panel);
  MLPR_STAMP(5);

  auto load_part = [&](const float* part) {   // the accumulators start from the producer's fp32 layer-1 state part
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = wave * 32 + tn * 16 + fr;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tm][tn][r] = part[(int64_t)(m0 + tm * 16 + fg * 4 + r) * HP + n];
    }
  };
  if (has_w3) {
    if (n_tail) {
      load_part(batch.tail[0].part);                    // in flight under layer 3
      const char* w1a[2];
      b_rows<2>(w1a, batch.tail[0].W1a, batch.tail[0].ldw1, 0, wave * 32, fr, fg);
      load_b<2>(bq[4], w1a, 0);
      load_b<2>(bq[5], w1a, ROWB);
    }
    // ---------------------------------------------------------------- layer 3 (actor): 64 x 128 outputs, 16 columns per wave
    f32x4 o[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i) o[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    wg_sync();                         // h2 panel complete
    if (P.h2) panel_to_global(panel, (bf16_t*)P.h2, P.ldh, m0, P.rows, tid);
#pragma unroll
    for (int u = 0; u < 4; ++u) mma_slab<1>(panel + u * SLAB, b3[u], o, fr, fg);
    {
      const int n = wave * 16 + fr;
      const bool ncol = n < P.out_dim;
      const float b3n = bias_lds[2 * HP + n];
      unsigned char* act = lds + ACT_OFF + (n >> 6) * SLAB + (n & 7) * 2;
      const int c = (n & 63) >> 3;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = tm * 16 + fg * 4 + r, m = m0 + row;
          float v = o[tm][0][r] + b3n;
          if (P.addend && ncol && m < P.rows) {
            const float z = P.addend[(int64_t)m * P.ld_add + n];
            v += fminf(fmaxf(z, -P.add_clip), P.add_clip);
          }
          const bf16_t hv = ncol ? f2bf(v) : (bf16_t)0;
          if (ncol && m < P.rows) ((bf16_t*)P.out)[(int64_t)m * P.ldo + n] = hv;
          if (n_tail) *(bf16_t*)(act + row * ROWB + ((c ^ swz(row)) * 16)) = hv;   // A operand of the chained critics
        }
    }
    MLPR_STAMP(6);
    // ---------------------------------------------------------------- chained critics (target critic on the new action)
#pragma unroll
    for (int ti = 0; ti < MLP_MAX_TAIL; ++ti) {
      if (ti >= n_tail) break;
      const MlpTail& T = batch.tail[ti];
      if (ti > 0) {
        load_part(T.part);
        const char* w1a[2];
        b_rows<2>(w1a, T.W1a, T.ldw1, 0, wave * 32, fr, fg);
        load_b<2>(bq[4], w1a, 0);
        load_b<2>(bq[5], w1a, ROWB);
      }
      const char* w2t[2];
      b_rows<2>(w2t, T.W2, T.ldw2, 0, wave * 32, fr, fg);
#pragma unroll
      for (int u = 0; u < 4; ++u) load_b<2>(bq[u], w2t, (int64_t)u * ROWB);
      wg_sync();                       // action panel written; everyone is done with the previous panel contents
      mma_slab<2>(lds + ACT_OFF, bq[4], acc, fr, fg);             // + action panel x W1a
      mma_slab<2>(lds + ACT_OFF + SLAB, bq[5], acc, fr, fg);
      hidden_epilogue(acc, bias_lds + (3 + 2 * ti) * HP, P.H, P.rows, m0, wave, fr, fg, RECNN_MASK_NONE, nullptr, 0, 0u, panel);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      wg_sync();                       // h1 panel of the critic complete
#pragma unroll
      for (int u = 0; u < 4; ++u) mma_slab<2>(panel + u * SLAB, bq[u], acc, fr, fg);
      wg_sync();                       // everyone is done reading the h1 panel
      hidden_epilogue(acc, bias_lds + (4 + 2 * ti) * HP, P.H, P.rows, m0, wave, fr, fg, RECNN_MASK_NONE, nullptr, 0, 0u, panel);
      wg_sync();
#pragma unroll
      for (int i = 0; i < BM / NW; ++i) {
        const int row = wave * (BM / NW) + i;
        const float sdot = row_dot(panel, row, lane, P.H, T.w3row);
        if (lane == 0) {
          const float qv = sdot + T.b3[0];
          if (m0 + row < P.rows) T.q[m0 + row] = qv;
          ((float*)(lds + STQ_OFF))[ti * BM + row] = qv;   // for the head below
        }
      }
    }
    MLPR_STAMP(7);
    // ---------------------------------------------------------------- head of the learning critic(s)
    if (batch.head.n_critic > 0 && n_tail > 0) {
      const MlpHead& Hd = batch.head;
      wg_sync();   // Q' of all 64 rows (every tail) is in LDS
      if (wave == 0) {
        const float* stq = (const float*)(lds + STQ_OFF);
        const int r = lane, m = m0 + r, mc = min(m, P.rows - 1);
        const bool valid = m < P.rows;
        const float rew = Hd.reward[mc], dn = Hd.done[mc];
        float tqv = stq[r];
        if (n_tail > 1) tqv = fminf(tqv, stq[BM + r]);
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
    wg_sync();                         // h2 panel complete
    if (P.h2) panel_to_global(panel, (bf16_t*)P.h2, P.ldh, m0, P.rows, tid);
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
    MLPR_STAMP(6);
    if (do_cbwd) {
      const MlpCriticBwd& B = batch.cbwd[P.cbwd_idx];
      // transposed-use image of 64 W2 rows (the unit backward contracts over W2's ROW index): [column half][k row][256 B],
      // chunk c of row k at position c ^ (k & 15), read back with ds_read_b64_tr_b16; two 32 KB buffers in the (idle) A ring
      uint4 tw[4];
      auto load_t = [&](int qk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = tid + j * NT, krow = idx >> 5, c32 = idx & 31;
          tw[j] = *(const uint4*)((const char*)W2p + ((int64_t)(qk * KS + krow) * ldw2 + c32 * 8) * 2);
        }
      };
      auto store_t = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = tid + j * NT, krow = idx >> 5, c32 = idx & 31;
          *(uint4*)(ring + buf * (4 * SLAB) + (c32 >> 4) * (KS * 256) + krow * 256 + (((c32 & 15) ^ (krow & 15)) * 16)) = tw[j];
        }
      };
      load_t(0);
      wg_sync();   // every wave is done reading h2 rows for its q dots
      // ---- u2 = w3 * scale * [h2 > 0], in place in the panel (it becomes the A operand) and to global
#pragma unroll
      for (int j = 0; j < BM * 32 / NT; ++j) {
        const int idx = tid + j * NT, row = idx >> 5, cc = idx & 31, n8 = cc * 8, m = m0 + row;
        const int nb = min(n8, P.H - 8);
        const float4 w3a = *(const float4*)(P.w3row + nb), w3b = *(const float4*)(P.w3row + nb + 4);
        const float wsc = n8 < P.H ? B.scale : 0.f;
        const float w3v[8] = {w3a.x * wsc, w3a.y * wsc, w3a.z * wsc, w3a.w * wsc, w3b.x * wsc, w3b.y * wsc, w3b.z * wsc, w3b.w * wsc};
        unsigned char* cell = panel + (cc >> 3) * SLAB + row * ROWB + (((cc & 7) ^ swz(row)) * 16);
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
        if (m < P.rows) *(uint4*)((bf16_t*)B.dz2 + (int64_t)m * P.ldh + n8) = packed;
      }
      store_t(0);
      // ---- U = (u2 W2) * scale * gate(h1)
      f32x4 dacc[TM][2];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int sw = (fr >> 1) & 7;
#pragma unroll
      for (int qk = 0; qk < 4; ++qk) {
        if (qk + 1 < 4) load_t(qk + 1);
        wg_sync();                     // image qk (and, first pass, the u2 panel) complete; image qk - 1 no longer read
        const unsigned char* wimg = ring + (qk & 1) * (4 * SLAB);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int pos = ((ks * 4 + fg) ^ sw) * 16;
          uint4 a[TM];
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) a[tm] = *(const uint4*)(panel + qk * SLAB + (tm * 16 + fr) * ROWB + pos);
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            const int n0 = wave * 32 + tn * 16;
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
            for (int tm = 0; tm < TM; ++tm)
              dacc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[tm]), __builtin_bit_cast(bf16x8, bvv),
                                                                     dacc[tm][tn], 0, 0, 0);
          }
        }
        if (qk + 1 < 4) store_t((qk + 1) & 1);   // the other buffer: last read in pass qk - 1, i.e. before this pass's barrier
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = wave * 32 + tn * 16 + fr;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int mm = m0 + tm * 16 + fg * 4 + r;
            if (mm < P.rows && n < P.H) {
              const float v = ((gate1 >> (tn * 16 + tm * 4 + r)) & 1u) ? dacc[tm][tn][r] * B.scale : 0.f;
              ((bf16_t*)B.dz1)[(int64_t)mm * P.ldh + n] = f2bf(v);
            }
          }
      }
    }
  }
  MLPR_STAMP(9);
}

static u64* g_mlpr_trace = nullptr;
void mlpr_set_trace(void* p) { g_mlpr_trace = (u64*)p; }   // called by recnn_tune_mlp_trace (mlp64.hip)

int mlpr_init() {
  return recnn_check_hip(hipFuncSetAttribute((const void*)mlpr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL),
                         "mlpr_kernel attr");
}

// Preconditions are those of mlp_launch (checked there).
int mlpr_launch(const MlpBatch& b, int nprob, int rows, int map_mode, hipStream_t s) {
  for (int i = 0; i < nprob; ++i) {
    const MlpProb& p = b.p[i];
    for (int g = 0; g < p.nseg; ++g)
      if (p.K[g] % KS) { recnn_set_error("mlpr: k extents must be multiples of 64"); return RECNN_E_INVALID; }
    if (p.ldh != HP && (p.h1 || p.h2 || p.cbwd_idx >= 0)) { recnn_set_error("mlpr: hidden activations must have pitch 256"); return RECNN_E_INVALID; }
  }
  const int npanel = (rows + BM - 1) / BM;
  hipLaunchKernelGGL(mlpr_kernel, dim3(npanel * nprob), dim3(NT), LDS_TOTAL, s, b, npanel, map_mode, g_mlpr_trace);
  return recnn_check_hip(hipGetLastError(), "mlpr_kernel");
}
