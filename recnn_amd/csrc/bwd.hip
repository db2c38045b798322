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

int bwd_init() {
  return recnn_check_hip(hipFuncSetAttribute((const void*)bwd_panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL),
                         "bwd_panel_kernel attr");
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
