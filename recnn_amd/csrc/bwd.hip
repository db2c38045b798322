// bwd.hip -- critic head + first backward GEMM in one row-panel kernel (bf16, gfx950).
//
// Replaces, for the fused bf16 path, head_kernel (head.hip) + the dX GEMM launch behind it:
//   recnn/nn/update/misc.py:6-7,33-39   TD target, clamp, MSE          (per-row scalars; Q and Q' arrive from the
//   recnn/nn/update/ddpg.py:79,87       policy_loss = -Q.mean()         fused forward, mlp.hip)
//   autograd through linear3 -> relu/dropout -> linear2 of the critic  (models.py:207-213 backwards)
// One workgroup (16 waves) owns 32 batch rows and all 256 hidden columns:
//   A  per-row seed d = dLoss/dQ (TD error x 2/B, or the constant -1/B), loss partial
//   B  dz2 = d * w3 * scale * [h2 > 0]  -> global (the dW GEMM reads it) and an LDS panel; partial sums for
//      dW3 / db2 / db3 over the 32 rows (shuffle tree inside a wave: a wave holds all 32 rows of its 16 columns)
//   C  dz1 = (dz2 W2) * scale * [h1 > 0]: W2 [k = out][n = in] is k-strided for this product, so its rows go
//      global -> LDS untouched by DMA and the B fragments are read with ds_read_b64_tr_b16 (as in the dW kernel);
//      column sums of dz1 over the panel -> db1 partial
// LDS: 2 x 64 KB W2 k-slabs (128 rows x 512 B) + 16 KB dz2 panel.  Everything is requested up front (both slabs,
// row scalars, the h2 chunk, the h1 gate values): the kernel is one memory latency deep.
// W2 slab image: 32-byte segment s (16 columns) of row k sits at segment position s ^ g(k),
// g(k) = (k & 3) | (((k >> 3) & 1) << 2): the 8 rows x 32 bytes of a half-wave transpose read hit 8 distinct
// 32-byte bank groups.
#include "bwd.h"

namespace {
constexpr int NW = 16;
constexpr int HP = 256;
constexpr int SLAB_BYTES = 128 * 512;       // 128 k rows x 256 columns bf16
constexpr int PANEL_OFF = 2 * SLAB_BYTES;   // dz2 panel: two k halves of 32 rows x 256 B (chunk c of row r at c ^ (r & 15))
constexpr int PANEL_HALF = BWD_ROWS * 256;
constexpr int LDS_TOTAL = PANEL_OFF + 2 * PANEL_HALF;

typedef short v4s16 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst_uniform)
      : "memory");
}

// k rows [k0, k0 + 128) of W2 (row pitch ld elements, 256 columns) -> LDS slab; 2 rows per wave instruction
__device__ __forceinline__ void dma_w2_slab(const void* W2, int64_t ld, int k0, int k_max, unsigned lds_dst, int wave, int lane) {
  const int r_in = lane >> 5, slot = lane & 31;
#pragma unroll
  for (int j = 0; j < 128 / (2 * NW); ++j) {
    const int pair = j * NW + wave;            // row pair index inside the slab
    const int row = pair * 2 + r_in;
    const int g = (row & 3) | (((row >> 3) & 1) << 2);
    const int c = (((slot >> 1) ^ g) << 1) | (slot & 1);
    const int gk = min(k0 + row, k_max);       // rows past H are clamped; they meet zero dz2 columns
    dma16((const char*)W2 + ((int64_t)gk * ld) * 2 + c * 16, lds_dst + pair * 1024);
  }
}
}  // namespace

__global__ __launch_bounds__(NW * 64) void bwd_panel_kernel(const BwdPanelBatch batch) {
  const BwdPanelProb& P = batch.p[blockIdx.y];
  const int m0 = blockIdx.x * BWD_ROWS;
  if (m0 >= P.rows) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // every byte this workgroup needs is requested before the first use: both W2 k-slabs, the row scalars, the
  // thread's h2 chunk and (below) its h1 gate values -- the kernel is one memory latency plus ~1 us of work
  dma_w2_slab(P.W2, P.ldw2, 0, HP - 1, lds0, wave, lane);
  dma_w2_slab(P.W2, P.ldw2, 128, HP - 1, lds0 + SLAB_BYTES, wave, lane);

  // ---------------------------------------------------------------- A + B: seeds, dz2 panel, dW3 / db2 partials
  // wave w owns the 16 hidden columns [16w, 16w+16); lane l: row l & 31, 8-column group 2w + (l >> 5).
  // Every wave evaluates the 32 row seeds itself (4 coalesced loads) so no LDS hand-off or barrier is needed, and
  // the sums over the panel's rows are 5 shuffle steps inside the wave.
  const int row = lane & 31, m = m0 + row;
  const int n8 = (2 * wave + (lane >> 5)) * 8;
  const bool valid = m < P.rows;
  // loads use clamped (always readable) addresses and are selected afterwards: a load inside a divergent branch
  // would make the compiler wait for it -- and with it for the W2 DMAs -- right there
  const int mc = min(m, P.rows - 1);
  uint4 raw = *(const uint4*)((const bf16_t*)P.h2 + (int64_t)mc * P.ldh + n8);
  if (!valid) raw = make_uint4(0, 0, 0, 0);
  const int nb = min(n8, P.H - 8);  // H is a multiple of 8: the thread's 8 columns are all inside H or all outside
  const float4 w3a = *(const float4*)(P.w3 + nb), w3b = *(const float4*)(P.w3 + nb + 4);
  const float wsc = n8 < P.H ? P.scale : 0.f;
  const float w3v[8] = {w3a.x * wsc, w3a.y * wsc, w3a.z * wsc, w3a.w * wsc, w3b.x * wsc, w3b.y * wsc, w3b.z * wsc, w3b.w * wsc};
  // h1 gate values of the epilogue (lane (fr, fg): column 16 wave + fr, rows tm*16 + fg*4 + r)
  const int n = wave * 16 + fr;
  const int ncl = min(n, P.H - 1);
  float gate[2][4];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mm = min(m0 + tm * 16 + fg * 4 + r, P.rows - 1);
      gate[tm][r] = bf2f(((const bf16_t*)P.h1)[(int64_t)mm * P.ldh + ncl]);
    }
  float d = P.delta_const, e2 = 0.f, yv = 0.f, tqv = 0.f;
  if (P.mode == 0) {  // uniform branch
    const float t0 = P.tq[0][mc];
    const float t1 = P.tq[P.n_target > 1 ? 1 : 0][mc];
    const float rew = P.reward[mc], dn = P.done[mc], qv = P.q[mc];
    tqv = fminf(t0, t1);
    yv = rew + (1.0f - dn) * P.gamma * tqv;
    yv = fminf(fmaxf(yv, P.lo), P.hi);
    const float e = qv - yv;
    d = e * (2.0f / (float)P.rows);
    e2 = e * e;
  }
  if (!valid) { d = 0.f; e2 = 0.f; }
  if (P.mode == 0 && wave == 0 && lane < 32 && valid) {
    if (P.expected) P.expected[m] = yv;
    if (P.target_q) P.target_q[m] = tqv;
    if (P.delta_out) P.delta_out[m] = d;
  }
  if (wave == 0) {
    float tot = lane < 32 ? e2 : 0.f, dsum = lane < 32 ? d : 0.f;
    tot = wave_sum(tot);
    dsum = wave_sum(dsum);
    if (lane == 0) {
      if (P.loss_part) P.loss_part[blockIdx.x] = tot;
      if (P.db3_part) P.db3_part[blockIdx.x] = dsum;
    }
  }
  float hv[8], dz[8], sw[8], sb[8];
  {
    const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hv[2 * j] = bf2f((bf16_t)(u[j] & 0xFFFF));
      hv[2 * j + 1] = bf2f((bf16_t)(u[j] >> 16));
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    dz[j] = hv[j] > 0.f ? d * w3v[j] : 0.f;
    sw[j] = d * hv[j];
    sb[j] = dz[j];
  }
  const uint4 packed = make_uint4(pack_bf2(dz[0], dz[1]), pack_bf2(dz[2], dz[3]), pack_bf2(dz[4], dz[5]), pack_bf2(dz[6], dz[7]));
  // panel image (A operand of phase C): k half n / 128, row, 16-byte chunk ((n % 128) / 8) ^ (row & 15)
  *(uint4*)(lds + PANEL_OFF + (n8 >> 7) * PANEL_HALF + row * 256 + ((((n8 & 127) >> 3) ^ (row & 15)) * 16)) = packed;
  if (P.dw3_part) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sw[j] = half_sum32(sw[j]);   // totals over the 32 rows land in lanes 16..31 / 48..63
      sb[j] = half_sum32(sb[j]);
    }
  }

  // ---------------------------------------------------------------- C: dz1 = dz2 x W2
  f32x4 acc[2];
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // both slabs landed for every wave, the dz2 panel is complete
#pragma unroll
  for (int slab = 0; slab < 2; ++slab) {
    const unsigned char* sa = lds + PANEL_OFF + slab * PANEL_HALF;
    const unsigned char* sbm = lds + slab * SLAB_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int pos = ((ks * 4 + fg) ^ fr) * 16;
      uint4 a[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * 256 + pos);
      v4s16 b[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int k = ks * 32 + fg * 8 + half * 4 + (fr >> 2);
        const int g = (k & 3) | (((k >> 3) & 1) << 2);
        b[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s16*)(sbm + k * 512 + ((wave ^ g) * 32) + (fr & 3) * 8));
      }
      struct { v4s16 lo, hi; } bv = {b[0], b[1]};
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
        acc[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[tm]), __builtin_bit_cast(bf16x8, bv), acc[tm], 0, 0, 0);
    }
  }

  // ---------------------------------------------------------------- stores: dz2, partials, gated dz1, column sums
  if (valid) *(uint4*)((bf16_t*)P.dz2 + (int64_t)m * P.ldh + n8) = packed;
  if (P.dw3_part && row == 31) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (n8 + j < P.H) {
        P.dw3_part[(int64_t)blockIdx.x * P.H + n8 + j] = sw[j];
        P.db2_part[(int64_t)blockIdx.x * P.H + n8 + j] = sb[j];
      }
  }
  float cs = 0.f;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mm = m0 + tm * 16 + fg * 4 + r;
      if (mm < P.rows && n < P.H) {
        const float v = gate[tm][r] > 0.f ? acc[tm][r] * P.scale : 0.f;
        cs += v;
        ((bf16_t*)P.dz1)[(int64_t)mm * P.ldh + n] = f2bf(v);
      }
    }
  if (P.colsum) {
    cs += __shfl_xor(cs, 16, 64);
    cs += __shfl_xor(cs, 32, 64);
    if (fg == 0 && n < P.H) P.colsum[(int64_t)blockIdx.x * P.H + n] = cs;
  }
}

// ------------------------------------------------------------------ policy-gradient chain (policy steps)
// Seven weight k-slabs (<= 64 KB each) stream through two LDS slots while four GEMMs run back to back on a 32-row
// panel that never leaves the CU.  Slab s of a B operand with n columns holds 128 k rows x n bf16; 32-byte segment q of
// row k sits at segment position q ^ g(k) (g as above), so transpose reads are conflict free for n = 256 and n = 128.
namespace {
constexpr int CH_SLOT = 64 * 1024;
constexpr int CH_PANEL = 2 * CH_SLOT;           // two ping-pong panels of 32 x 256 bf16 (16 KB each)
constexpr int CH_PANEL_BYTES = 2 * PANEL_HALF;
constexpr int CH_LDS = CH_PANEL + 2 * CH_PANEL_BYTES;   // 160 KB

struct ChSlab { const void* W; int64_t ld; int k0; int krows; int ncols; int col0; };

// k rows [k0, k0 + krows) x ncols columns (from column col0) -> LDS slab; one wave instruction = 1024 bytes
__device__ __forceinline__ void ch_dma_slab(const ChSlab& S, int k_max, unsigned lds_dst, int wave, int lane) {
  const int rpi = S.ncols == 256 ? 2 : 4;            // rows per wave instruction
  const int cpr = S.ncols / 8;                        // 16-byte chunks per row (32 or 16)
  const int r_in = lane / cpr, slot = lane % cpr;
  const int n_instr = S.krows / rpi;
  for (int j = wave; j < n_instr; j += NW) {
    const int row = j * rpi + r_in;
    const int g = (row & 3) | (((row >> 3) & 1) << 2);
    const int c = (((slot >> 1) ^ g) << 1) | (slot & 1);
    const int gk = min(S.k0 + row, k_max);
    dma16((const char*)S.W + ((int64_t)gk * S.ld + S.col0) * 2 + c * 16, lds_dst + j * 1024);
  }
}

// acc += panel(32 x 128 k, half `half` of the panel) x slab (128 k rows x ncols); wave owns columns [16 wave, +16)
__device__ __forceinline__ void ch_mma(const unsigned char* panel, int half, const unsigned char* slab, int row_bytes, int krows,
                                       f32x4 (&acc)[2], int wave, int fr, int fg) {
  const unsigned char* sa = panel + half * PANEL_HALF;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks * 32 >= krows) break;
    const int pos = ((ks * 4 + fg) ^ fr) * 16;
    uint4 a[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) a[tm] = *(const uint4*)(sa + (tm * 16 + fr) * 256 + pos);
    v4s16 b[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int k = ks * 32 + fg * 8 + hh * 4 + (fr >> 2);
      const int g = (k & 3) | (((k >> 3) & 1) << 2);
      b[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) v4s16*)(slab + k * row_bytes + ((wave ^ g) * 32) + (fr & 3) * 8));
    }
    struct { v4s16 lo, hi; } bv = {b[0], b[1]};
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
      acc[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[tm]), __builtin_bit_cast(bf16x8, bv), acc[tm], 0, 0, 0);
  }
}
}  // namespace

__global__ __launch_bounds__(NW * 64) void bwd_chain_kernel(const BwdChainArgs P) {
  const int m0 = blockIdx.x * BWD_ROWS;
  if (m0 >= P.rows) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // the seven slabs in consumption order
  const ChSlab slabs[7] = {
      {P.W2c, P.ldw2c, 0, 128, 256, 0}, {P.W2c, P.ldw2c, 128, 128, 256, 0},   // stage 1: K = 256 (out of critic L2), N = 256
      {P.W1c, P.ldw1c, 0, 128, 128, 0}, {P.W1c, P.ldw1c, 128, 128, 128, 0},   // stage 2: K = 256 (out of critic L1), N = 128 action columns
      {P.W3a, P.ldw3a, 0, 128, 256, 0},                                         // stage 3: K = 128 (actor outputs), N = 256
      {P.W2a, P.ldw2a, 0, 128, 256, 0}, {P.W2a, P.ldw2a, 128, 128, 256, 0}};  // stage 4
  const int kmax[7] = {HP - 1, HP - 1, HP - 1, HP - 1, 127, HP - 1, HP - 1};
  ch_dma_slab(slabs[0], kmax[0], lds0, wave, lane);
  ch_dma_slab(slabs[1], kmax[1], lds0 + CH_SLOT, wave, lane);

  // gates of the three gated stages for this lane's accumulator elements (column 16 wave + fr, rows tm*16 + fg*4 + r)
  const int n = wave * 16 + fr;
  const int ncl = min(n, P.H - 1);
  uint32_t gate_e1 = 0, gate_p2 = 0, gate_p1 = 0;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mm = min(m0 + tm * 16 + fg * 4 + r, P.rows - 1);
      const int64_t off = (int64_t)mm * P.ldh + ncl;
      if (bf2f(((const bf16_t*)P.e1)[off]) > 0.f) gate_e1 |= 1u << (tm * 4 + r);
      if (bf2f(((const bf16_t*)P.p2)[off]) > 0.f) gate_p2 |= 1u << (tm * 4 + r);
      if (bf2f(((const bf16_t*)P.p1)[off]) > 0.f) gate_p1 |= 1u << (tm * 4 + r);
    }

  // ---- stage 0: dz_e2 panel from e2 (thread: row lane & 31, 8 columns)
  unsigned char* panel[2] = {lds + CH_PANEL, lds + CH_PANEL + CH_PANEL_BYTES};
  {
    const int row = lane & 31, m = m0 + row, mc = min(m, P.rows - 1);
    const int n8 = (2 * wave + (lane >> 5)) * 8;
    const int nb = min(n8, P.H - 8);
    uint4 raw = *(const uint4*)((const bf16_t*)P.e2 + (int64_t)mc * P.ldh + n8);
    if (m >= P.rows) raw = make_uint4(0, 0, 0, 0);
    const float4 w3a = *(const float4*)(P.w3c + nb), w3b = *(const float4*)(P.w3c + nb + 4);
    const float wsc = n8 < P.H ? P.scale * P.delta_const : 0.f;
    const float w3v[8] = {w3a.x * wsc, w3a.y * wsc, w3a.z * wsc, w3a.w * wsc, w3b.x * wsc, w3b.y * wsc, w3b.z * wsc, w3b.w * wsc};
    const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
    float dz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float hv = bf2f((bf16_t)((u[j >> 1] >> ((j & 1) * 16)) & 0xFFFF));
      dz[j] = hv > 0.f ? w3v[j] : 0.f;
    }
    *(uint4*)(panel[0] + (n8 >> 7) * PANEL_HALF + row * 256 + ((((n8 & 127) >> 3) ^ (row & 15)) * 16)) =
        make_uint4(pack_bf2(dz[0], dz[1]), pack_bf2(dz[2], dz[3]), pack_bf2(dz[4], dz[5]), pack_bf2(dz[6], dz[7]));
  }

  f32x4 acc[2];
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // stage of each slab, whether it is the stage's last slab, which half of the A panel it multiplies
  const int st_of[7] = {1, 1, 2, 2, 3, 4, 4};
  const int last_of[7] = {0, 1, 0, 1, 1, 0, 1};
  const int half_of[7] = {0, 1, 0, 1, 0, 0, 1};
  int cur = 0;   // panel holding the current stage's A operand
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    // slab i has landed once only the (at most one) younger slab's DMAs are outstanding
    if (i + 1 < 7) {
      // instructions per wave of the younger slab: krows / rows-per-instr / NW
      const int ni_next = (slabs[i + 1].krows / (slabs[i + 1].ncols == 256 ? 2 : 4)) / NW;
      if (ni_next == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const int stage = st_of[i];
    const int ncols = slabs[i].ncols;
    if (wave * 16 < ncols)
      ch_mma(panel[cur], half_of[i], lds + (i & 1) * CH_SLOT, ncols * 2, slabs[i].krows, acc, wave, fr, fg);
    if (last_of[i]) {
      // ---- stage epilogue: gate, column sums, global store, next panel
      const uint32_t gate = stage == 1 ? gate_e1 : (stage == 3 ? gate_p2 : (stage == 4 ? gate_p1 : 0xFFu));
      const float sc = stage == 2 ? 1.0f : P.scale;
      const int nvalid = stage == 2 ? P.A : P.H;
      bf16_t* gout = stage == 2 ? (bf16_t*)P.dact : (stage == 3 ? (bf16_t*)P.dzp2 : (stage == 4 ? (bf16_t*)P.dzp1 : nullptr));
      const int64_t ldg = stage == 2 ? P.ldact : P.ldh;
      float* cpart = stage == 2 ? P.db3_part : (stage == 3 ? P.db2_part : (stage == 4 ? P.db1_part : nullptr));
      unsigned char* nxt = panel[cur ^ 1];
      float cs = 0.f;
      if (wave * 16 < ncols) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = tm * 16 + fg * 4 + r, mm = m0 + row;
            float v = ((gate >> (tm * 4 + r)) & 1u) ? acc[tm][r] * sc : 0.f;
            if (mm >= P.rows || n >= nvalid) v = 0.f;
            cs += v;
            const bf16_t hv = f2bf(v);
            if (gout && mm < P.rows && n < nvalid) gout[(int64_t)mm * ldg + n] = hv;
            if (stage < 4) *(bf16_t*)(nxt + (n >> 7) * PANEL_HALF + row * 256 + ((((n & 127) >> 3) ^ (row & 15)) * 16) + (n & 7) * 2) = hv;
          }
        if (cpart) {
          cs += __shfl_xor(cs, 16, 64);
          cs += __shfl_xor(cs, 32, 64);
          if (fg == 0 && n < nvalid) cpart[(int64_t)blockIdx.x * nvalid + n] = cs;
        }
      }
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      cur ^= 1;
    }
    if (i + 2 < 7) {
      __builtin_amdgcn_s_barrier();   // every wave is done reading slot i & 1
      ch_dma_slab(slabs[i + 2], kmax[i + 2], lds0 + (i & 1) * CH_SLOT, wave, lane);
    }
  }
}

int bwd_chain_launch(const BwdChainArgs& a, hipStream_t s) {
  if (a.rows <= 0) return 0;
  if (a.H > HP || a.A > 128 || (a.H % 8) || (a.A % 8) || a.ldh < HP || (a.ldh % 8) || a.ldact < 128 || (a.ldact % 8)) {
    recnn_set_error("bwd_chain: needs hidden <= 256, action_dim <= 128 (multiples of 8) and padded pitches");
    return RECNN_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL(bwd_chain_kernel, dim3((a.rows + BWD_ROWS - 1) / BWD_ROWS), dim3(NW * 64), CH_LDS, s, a);
  return recnn_check_hip(hipGetLastError(), "bwd_chain_kernel");
}

int bwd_init() {
  int rc = recnn_check_hip(hipFuncSetAttribute((const void*)bwd_panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL),
                           "bwd_panel_kernel attr");
  if (rc) return rc;
  return recnn_check_hip(hipFuncSetAttribute((const void*)bwd_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS),
                         "bwd_chain_kernel attr");
}

int bwd_panel_launch(const BwdPanelBatch& b, int nprob, hipStream_t s) {
  int rows = 0;
  for (int i = 0; i < nprob; ++i) {
    const BwdPanelProb& p = b.p[i];
    if (p.rows > rows) rows = p.rows;
    if (p.H > HP || (p.H % 8) || (p.ldh % 8) || p.ldh < HP || p.ldw2 < HP || (p.ldw2 % 8)) {
      recnn_set_error("bwd_panel: hidden size must be a multiple of 8, <= 256, pitches >= 256");
      return RECNN_E_UNSUPPORTED;
    }
    if (!p.h2 || !p.w3 || !p.dz2 || !p.W2 || !p.h1 || !p.dz1 || (p.mode == 0 && (!p.q || !p.tq[0] || !p.reward || !p.done))) {
      recnn_set_error("bwd_panel: null argument");
      return RECNN_E_INVALID;
    }
    if (p.dw3_part && !p.db2_part) { recnn_set_error("bwd_panel: dw3_part and db2_part go together"); return RECNN_E_INVALID; }
  }
  if (rows <= 0 || nprob <= 0) return 0;
  hipLaunchKernelGGL(bwd_panel_kernel, dim3((rows + BWD_ROWS - 1) / BWD_ROWS, nprob), dim3(NW * 64), LDS_TOTAL, s, b);
  return recnn_check_hip(hipGetLastError(), "bwd_panel_kernel");
}
