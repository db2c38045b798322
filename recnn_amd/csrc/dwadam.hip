// dwadam.hip -- the learning critics' weight gradients AND their optimizer step in ONE launch (bf16, gfx950; round 6).
//
// Replaces, on the single-GPU step whose backward tensors come from mlpt.hip (split forward / cycle mode), two launches and the
// round trip of their hand-off through memory:
//   gemm_dw_dma_kernel   autograd's mm(dZ^T, X) of linear1 / linear2 as 8 / 16 split-batch fp32 slabs   recnn/nn/update/misc.py:42-43
//   apply_kernel         slab sum + torch.optim.Adam.step (+ soft_update on policy steps)                misc.py:44, ddpg.py:95-97
// (10.6 + 8.75 us per step, 27 + 28 MB of traffic for 8.6 + 12 MB of algorithmic bytes: VERDICT r5 weak #5.)
//
// A workgroup owns a 32 x 64 tile of one weight matrix (critic W1: 8 x 23 tiles, W2: 8 x 4 -> 216 workgroups, one per CU) and contracts
// the WHOLE batch for it, split over its eight consumer waves exactly like the slabs of the two-launch path: wave g multiplies batch
// rows [g R, (g + 1) R), R = rows / 8, in 32-row MFMA steps, into one accumulator set per slab of that range (W1: one slab per wave,
// W2: two).  The partial tiles then meet in LDS, every element is summed in apply_kernel's order (slab_grads: 0 + ((s0 + s1) + (s2 + s3))
// + ((s4 + s5) + (s6 + s7)), then the next eight), and the thread that holds the sum applies optim.h's opt_elem to it -- the master
// weight, both moments, the gradient arena, the bf16 compute shadow and, on policy steps, the soft-updated target and its shadow leave
// from there.  Same products in the same order per accumulator, same slab order, same opt_elem: BIT-IDENTICAL to the two launches
// (tests/test_gpu_dwadam.py, and every run == pieces == loop identity of tests/test_gpu_bench_shape.py, whose loop side still takes
// the two-launch path).  Round 3's dwopt.hip had 64 x 64 tiles (112 workgroups, 524 KB each, every wave loading AND multiplying with
// un-pipelined fragment reads: 26.7 us); here the tile is half as large (393 KB per workgroup), four LOADER waves do nothing but issue
// the ring's LDS-DMA (gemm.hip x3_fwd_ws_kernel's split of roles), and the optimizer state of a tile is in registers before the
// contraction starts.
//
// Operand stream: both operands are k-strided (k = batch row).  A stage = the 32 rows every consumer multiplies next: per consumer
// 32 x 128 B of X (64 columns) + 32 x 64 B of dZ (32 columns) = 6 KB, 48 KB per stage, 3-slot ring, two stages in flight.  Rows go
// global -> LDS untouched (`global_load_lds_dwordx4`, the XOR swizzle applied to the lane's SOURCE address) and the MFMA fragments are
// read with `ds_read_b64_tr_b16` (dw_tile.h's layout for the 128-byte rows; for the 64-byte rows of dZ the two 32-byte halves of a row
// swap places in rows 8..15 of every 16, so that the 8 rows x 32 B of a half-wave read fall on 64 different banks).
// The tensors that are not tiled (b1, b2, w3, b3: 769 elements, panel sums from mlpt.hip) are updated by apply_body workgroups of the
// same launch, first in the launch order.
#include "dwadam.h"
#include "optim_dev.h"
#include "x3.h"
#include "recnn_hip_debug.h"

typedef short v4s16 __attribute__((ext_vector_type(4)));

namespace {
constexpr int NC = 8, NL = 4;                    // consumer / loader waves
constexpr int NS = 3, D = NS - 1;                // ring slots, stages in flight ahead of the consumers
constexpr int CH_X = 32 * 128, CH_Z = 32 * 64;   // one consumer's rows of a stage: X 32 x 64 columns, dZ 32 x 32 columns
constexpr int CHUNK = CH_X + CH_Z;               // 6 KB
constexpr int STAGE = NC * CHUNK;                // 48 KB
constexpr int RING = NS * STAGE;                 // 144 KB
constexpr int TP = 68;                           // fp32 pitch of a partial tile image (bank spread)
constexpr int PART = 32 * TP * 4;                // 8704 B per slab
constexpr int SCAL_OFF = RING;                   // the launch's optimizer scalars (OptScalars)
constexpr int LDS_TOTAL = RING + 64;
constexpr int PER = (CH_X + CH_Z) / 1024 * (NC / NL);   // DMA instructions per loader wave and stage (12)
static_assert(16 * PART <= RING, "16 partial tiles must fit the idle ring");
static_assert(sizeof(OptScalars) <= 64, "scalar block");

__device__ __forceinline__ void dma_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "m0");
}

struct Pair { float2 p, m, v, tp; };

#define DWA_STAMP(i) do { if (trow) trow[(i)] = __builtin_amdgcn_s_memtime(); } while (0)

// ---- split bf16 (x3.h): the tile's operands are 64 physical columns of dZ (one [hi 32 | lo 32] group = 32 logical) and 128 of X (two
// groups = 64 logical); three MFMAs per 32-row k step and 16 x 16 block (x3_mfma: lo.hi, hi.lo, hi.hi -- x3.hip x3_dw_kernel's order).
// A stage = 128 batch rows (four k steps) of both operands: X 128 x 256 B + dZ 128 x 128 B = 48 KB, the same 3-slot ring.  Consumer wave w
// owns ONE 16 x 16 logical block (tm = w / 4, tn = w % 4) for ALL slabs (one accumulator per slab: the slabs are consecutive stage
// ranges), so the slab sums need no exchange -- each lane adds its own accumulators in apply_kernel's order and the finished tile goes
// through LDS only to reach the row-contiguous quads of the shared epilogue.
constexpr int X3_ROWS = 128;                     // batch rows per stage
constexpr int X3_XB = X3_ROWS * 256;             // 32 KB of X rows, then 16 KB of dZ rows
__device__ __forceinline__ int swz32(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }   // (x3.hip: 32-byte chunk c of a 256-byte row at c ^ swz32)
struct TrFrag { v4s16 lo, hi; };
// transpose-read fragment of 16 physical columns [col0, col0 + 16) over the 32 rows of k step `ks` of the X image (256-byte rows)
__device__ __forceinline__ bf16x8 x3_frag_x(const unsigned char* s, int col0, int fr, int fg) {
  TrFrag f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = fg * 8 + half * 4 + (fr >> 2);
    const v4s16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) v4s16*)(s + row * 256 + (((col0 >> 4) ^ swz32(row)) << 5) + (fr & 3) * 8));
    if (half == 0) f.lo = v; else f.hi = v;
  }
  return __builtin_bit_cast(bf16x8, f);
}
// ... of the dZ image (128-byte rows, dw_tile.h's pair swizzle p ^ f(row))
__device__ __forceinline__ bf16x8 x3_frag_z(const unsigned char* s, int col0, int fr, int fg) {
  TrFrag f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = fg * 8 + half * 4 + (fr >> 2);
    const int fs = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    const int c = (col0 >> 3) + ((fr & 3) >> 1);
    const int slot = (((c >> 1) ^ fs) << 1) | (c & 1);
    const v4s16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(s + row * 128 + slot * 16 + (fr & 1) * 8));
    if (half == 0) f.lo = v; else f.hi = v;
  }
  return __builtin_bit_cast(bf16x8, f);
}

// SPW = slabs per consumer wave (nslab / 8); X3: split-bf16 operands and shadows
template <int SPW, bool X3>
__device__ __forceinline__ void tile_role(const DwAdamNet& N, const DwAdamProb& P, const int tile, unsigned char* lds, unsigned long long* trow) {
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_n = tile / P.tiles_m, tile_m = tile - tile_n * P.tiles_m;   // the tiles of one X column block side by side (one XCD's L2)
  const int m0 = tile_m * 32, n0 = tile_n * 64;
  const int R = N.rows / NC;                     // batch rows per consumer wave
  const int nstep = R / 32;

  if (X3 && wave >= NC) {
    // ------------------------------------------------------------ loader wave lw (split bf16): instructions q = j * NL + lw, j < 12, of every
    // 48-instruction stage -- q < 32: X rows 4 q .. + 3 (256 B each), else dZ rows 8 (q - 32) .. + 7 (128 B each)
    const int lw = wave - NC;
    const int nst = N.rows / X3_ROWS;
    unsigned voff[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int q = j * NL + lw;
      if (q < 32) {
        const int r = 4 * q + (lane >> 4), ch = lane & 15;
        const int src = ((((ch >> 1) ^ swz32(r)) << 1) | (ch & 1));
        voff[j] = (unsigned)((r * (int)P.ldx + 2 * n0) * 2 + src * 16);
      } else {
        const int r = 8 * (q - 32) + (lane >> 3), ds = lane & 7;
        const int fs = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
        const int c = (((ds >> 1) ^ fs) << 1) | (ds & 1);
        voff[j] = (unsigned)((r * (int)P.ldz + 2 * m0) * 2 + c * 16);
      }
    }
    const int64_t xstep = (int64_t)X3_ROWS * P.ldx * 2, zstep = (int64_t)X3_ROWS * P.ldz * 2;
    auto issue = [&](int t, int slot) {
      const char* xs = (const char*)P.x + t * xstep;
      const char* zs = (const char*)P.dz + t * zstep;
      const unsigned sb = lds0 + slot * STAGE;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int q = j * NL + lw;
        if (q < 32) dma_s(voff[j], xs, sb + q * 1024);
        else dma_s(voff[j], zs, sb + X3_XB + (q - 32) * 1024);
      }
    };
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < nst) issue(i, i);
    int slot = D % NS;
    for (int t = 0; t < nst; ++t) {
      if (t + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + D < nst) issue(t + D, slot);
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
    return;
  }
  if (wave >= NC) {
    // ------------------------------------------------------------ loader wave lw: the chunks of consumers 2 lw, 2 lw + 1 of every stage
    const int lw = wave - NC;
    unsigned voffx[4], voffz[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                // X: 8 rows x 128 B per instruction; pair position p ^ f(row) (dw_tile.h)
      const int d_row = lane >> 3, d_slot = lane & 7;
      const int f = ((d_row >> 1) & 1) | ((i & 1) << 1);
      const int c = (((d_slot >> 1) ^ f) << 1) | (d_slot & 1);
      voffx[i] = (unsigned)(((i * 8 + d_row) * (int)P.ldx + n0) * 2 + c * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {                // dZ: 16 rows x 64 B per instruction; the row's two 32-byte halves swap in rows 8..15
      const int rr = j * 16 + (lane >> 2), q = lane & 3;
      const int c = (((q >> 1) ^ ((rr >> 3) & 1)) << 1) | (q & 1);
      voffz[j] = (unsigned)((rr * (int)P.ldz + m0) * 2 + c * 16);
    }
    const char* xb[2];
    const char* zb[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int g = lw * 2 + gi;
      xb[gi] = (const char*)P.x + (int64_t)g * R * P.ldx * 2;
      zb[gi] = (const char*)P.dz + (int64_t)g * R * P.ldz * 2;
    }
    const int64_t xstep = 32 * P.ldx * 2, zstep = 32 * P.ldz * 2;
    auto issue = [&](int t, int slot) {
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        const unsigned cb = lds0 + slot * STAGE + (lw * 2 + gi) * CHUNK;
        const char* xs = xb[gi] + t * xstep;
        const char* zs = zb[gi] + t * zstep;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_s(voffx[i], xs, cb + i * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma_s(voffz[j], zs, cb + CH_X + j * 1024);
      }
    };
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < nstep) issue(i, i);
    int slot = D % NS;
    for (int t = 0; t < nstep; ++t) {
      if (t + 1 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");   // (D = 2: one younger stage outstanding)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // this wave's part of stage t has landed; the consumers are done with stage t - 1
      if (t + D < nstep) issue(t + D, slot);
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
    return;                                      // (a terminated wave does not take part in the consumers' later barriers)
  }

  // -------------------------------------------------------------- consumer wave cw = slab group cw
  const int cw = wave;
  const int fr = lane & 15, fg = lane >> 4;
  const TensorSeg& T = N.L.t[P.tensor];
  const ApplyArgs& a = N.a;
  // this thread's elements of the finished tile: row em, four neighbouring shadow columns from enq, handled as two PAIRS (a pair never
  // straddles the row end or the rotation wrap: cols, col_rot even -- dwadam_tensor_ok)
  const int em = tid >> 4, enq = (tid & 15) * 4;
  const int row = m0 + em;
  int64_t idx[2];
  bool ok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = n0 + enq + 2 * h;              // shadow column = (col + col_rot) mod cols
    ok[h] = c < T.cols && row < T.rows;
    int col = c - T.col_rot;
    if (col < 0) col += T.cols;
    idx[h] = ok[h] ? T.p_off + (int64_t)row * T.cols + col : T.p_off;   // (a pair outside the tensor loads the tensor's first pair, stores nothing)
  }
  // the optimizer state of those elements: requested NOW, in registers long before the contraction ends.  Straight-line code: a load
  // under a branch makes hipcc wait for it at the join (vmcnt(0) per pair: two serial memory latencies in front of the first stage)
  Pair e[2];
  const float* tsrc = a.tgt_p ? a.tgt_p : a.p;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    e[h].p = *(const float2*)(a.p + idx[h]);
    e[h].m = *(const float2*)(a.m + idx[h]);
    e[h].v = *(const float2*)(a.v + idx[h]);
    e[h].tp = *(const float2*)(tsrc + idx[h]);
  }
  // everything the epilogue reads from the kernel-argument block, ONCE, into registers (read at their use sites these were 17 scalar
  // loads + waits and the optimizer arithmetic's branches waited on them: in-kernel trace, 3.4k clocks for four elements per thread)
  OptK ok_ = {a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.omb1, a.omb2, a.la_alpha};
  asm volatile("" : "+s"(ok_.lr), "+s"(ok_.beta1), "+s"(ok_.beta2), "+s"(ok_.eps), "+s"(ok_.weight_decay), "+s"(ok_.omb1), "+s"(ok_.omb2), "+s"(ok_.la_alpha));
  float* a_p = a.p; float* a_m = a.m; float* a_v = a.v; float* a_g = a.g_out; float* a_tp = a.tgt_p;
  float* a_slow = a.slow;
  bf16_t* a_sh = (bf16_t*)a.shadow; bf16_t* a_tsh = (bf16_t*)a.tgt_shadow;
  float a_tau = a.tau, a_gs = a.grad_scale;
  asm volatile("" : "+s"(a_tau), "+s"(a_gs));
  int64_t t_sh_off = T.sh_off;
  int t_sh_ld = T.sh_ld;
  {  // (pinned: hipcc otherwise re-reads a kernel argument wherever it is used)
    uint64_t q0 = (uint64_t)a_p, q1 = (uint64_t)a_m, q2 = (uint64_t)a_v, q3 = (uint64_t)a_g, q4 = (uint64_t)a_tp, q5 = (uint64_t)a_slow,
             q6 = (uint64_t)a_sh, q7 = (uint64_t)a_tsh;
    asm volatile("" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3), "+s"(q4), "+s"(q5), "+s"(q6), "+s"(q7), "+s"(t_sh_off), "+s"(t_sh_ld));
    a_p = (float*)q0; a_m = (float*)q1; a_v = (float*)q2; a_g = (float*)q3; a_tp = (float*)q4; a_slow = (float*)q5;
    a_sh = (bf16_t*)q6; a_tsh = (bf16_t*)q7;
  }
  // the step's optimizer scalars (an fp64 chain, optim.h): ONE wave evaluates them under the launch's first memory latency
  if (cw == NC - 1) {
    const OptScalars S = opt_scalars(a);
    if (lane == 0) *(OptScalars*)(lds + SCAL_OFF) = S;
  }
  DWA_STAMP(1);

  float g[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (X3) {
    // ---- split bf16: this wave's 16 x 16 logical block over all slabs
    constexpr int NSLAB = NC * SPW;
    const int tm = cw >> 2, tn = cw & 3;
    const int nst = N.rows / X3_ROWS, sps = nst / NSLAB;   // stages, stages per slab
    f32x4 acc[NSLAB];
#pragma unroll
    for (int i = 0; i < NSLAB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int xcol = (tn >> 1) * 64 + (tn & 1) * 16;       // physical column of the block's hi part inside the X image (lo: + 32)
    int slot = 0;
    bool first = true;
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
      for (int st = 0; st < sps; ++st) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (first) { DWA_STAMP(2); first = false; }
        const unsigned char* sx = lds + slot * STAGE;
        const unsigned char* sz = sx + X3_XB;
#pragma unroll
        for (int kk = 0; kk < X3_ROWS / 32; ++kk) {
          const bf16x8 ah = x3_frag_z(sz + kk * 32 * 128, tm * 16, fr, fg), al = x3_frag_z(sz + kk * 32 * 128, 32 + tm * 16, fr, fg);
          const bf16x8 bh = x3_frag_x(sx + kk * 32 * 256, xcol, fr, fg), bl = x3_frag_x(sx + kk * 32 * 256, xcol + 32, fr, fg);
          acc[sl] = x3_mfma(ah, al, bh, bl, acc[sl]);
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
      }
    }
    DWA_STAMP(3);
    // slab sums in apply_kernel's order, per lane: acc[.][r] = dW[m0 + 16 tm + 4 fg + r][n0 + 16 tn + fr]
    f32x4 gsum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < NSLAB; s0 += 8)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        gsum[k] += ((acc[s0][k] + acc[s0 + 1][k]) + (acc[s0 + 2][k] + acc[s0 + 3][k])) + ((acc[s0 + 4][k] + acc[s0 + 5][k]) + (acc[s0 + 6][k] + acc[s0 + 7][k]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // every consumer has read its last stage: the ring is idle
    {
      float* img = (float*)lds;
#pragma unroll
      for (int r = 0; r < 4; ++r) img[(tm * 16 + fg * 4 + r) * TP + tn * 16 + fr] = gsum[r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DWA_STAMP(4);
    const f32x4 gv = *(const f32x4*)((const float*)lds + em * TP + enq);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = gv[k];
  } else {
    f32x4 acc[SPW][2][4];
  #pragma unroll
    for (int s = 0; s < SPW; ++s)
  #pragma unroll
      for (int i = 0; i < 2; ++i)
  #pragma unroll
        for (int j = 0; j < 4; ++j) acc[s][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside a chunk (constant over the stages): k row of the lane = fg * 8 + half * 4 + (fr >> 2)
    int offa[2][2], offb[4][2];
  #pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int kr = fg * 8 + half * 4 + (fr >> 2);
      const int f = ((kr >> 1) & 1) | (((kr >> 3) & 1) << 1);
  #pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int pos = tm ^ ((kr >> 3) & 1);
        offa[tm][half] = CH_X + kr * 64 + (pos * 2 + ((fr & 3) >> 1)) * 16 + (fr & 1) * 8;
      }
  #pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int c = 2 * tn + ((fr & 3) >> 1);
        const int sl = (((c >> 1) ^ f) << 1) | (c & 1);
        offb[tn][half] = kr * 128 + sl * 16 + (fr & 1) * 8;
      }
    }
    const int per_slab = nstep / SPW;
    int slot = 0;
  #pragma unroll
    for (int s = 0; s < SPW; ++s) {
      for (int tt = 0; tt < per_slab; ++tt) {
        __builtin_amdgcn_s_barrier();              // stage t is in LDS (every loader waited for its part)
        if (s == 0 && tt == 0) DWA_STAMP(2);
        const unsigned char* ch = lds + slot * STAGE + cw * CHUNK;
        v4s16 fa[2][2], fb[4][2];
  #pragma unroll
        for (int half = 0; half < 2; ++half) {
  #pragma unroll
          for (int tm = 0; tm < 2; ++tm)
            fa[tm][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(ch + offa[tm][half]));
  #pragma unroll
          for (int tn = 0; tn < 4; ++tn)
            fb[tn][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(ch + offb[tn][half]));
        }
  #pragma unroll
        for (int tm = 0; tm < 2; ++tm)
  #pragma unroll
          for (int tn = 0; tn < 4; ++tn) {
            struct { v4s16 lo, hi; } av = {fa[tm][0], fa[tm][1]}, bv = {fb[tn][0], fb[tn][1]};
            acc[s][tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[s][tm][tn], 0, 0, 0);
          }
        slot = slot + 1 == NS ? 0 : slot + 1;
      }
    }
    DWA_STAMP(3);
    // ---- the partial tiles meet in LDS (the ring is idle once every consumer has read its last stage)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      float* part = (float*)lds;
  #pragma unroll
      for (int s = 0; s < SPW; ++s) {
        float* ps = part + (cw * SPW + s) * (PART / 4);
  #pragma unroll
        for (int tm = 0; tm < 2; ++tm)
  #pragma unroll
          for (int tn = 0; tn < 4; ++tn)
  #pragma unroll
            for (int r = 0; r < 4; ++r) ps[(tm * 16 + fg * 4 + r) * TP + tn * 16 + fr] = acc[s][tm][tn][r];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DWA_STAMP(4);
    constexpr int NG = NC * SPW;
    {
      const float* pe = (const float*)lds + em * TP + enq;
  #pragma unroll
      for (int s0 = 0; s0 < NG; s0 += 8) {         // slab_grads' order: eight slabs as a balanced tree, added to the running sum
        f32x4 v[8];
  #pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(pe + (s0 + j) * (PART / 4));
  #pragma unroll
        for (int k = 0; k < 4; ++k) g[k] += ((v[0][k] + v[1][k]) + (v[2][k] + v[3][k])) + ((v[4][k] + v[5][k]) + (v[6][k] + v[7][k]));
      }
    }
  }
  const OptScalars S = *(const OptScalars*)(lds + SCAL_OFF);
  const float gs = a_gs;
  float pn[4], tn4[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float p[2] = {e[h].p.x, e[h].p.y}, m[2] = {e[h].m.x, e[h].m.y}, v[2] = {e[h].v.x, e[h].v.y}, tp[2] = {e[h].tp.x, e[h].tp.y};
    float sl[2] = {0.f, 0.f};
    if (S.la_sync && ok[h]) { const float2 x = *(const float2*)(a_slow + idx[h]); sl[0] = x.x; sl[1] = x.y; }
    // (the optimizer kind decided ONCE per pair: inside either branch opt_elem's own tests of the step scalars fold away)
    if (S.ranger) {
#pragma unroll
      for (int j = 0; j < 2; ++j) opt_elem(ok_, S, g[2 * h + j], gs, p[j], m[j], v[j], sl[j]);
    } else if (S.adam) {
#pragma unroll
      for (int j = 0; j < 2; ++j) opt_elem(ok_, S, g[2 * h + j], gs, p[j], m[j], v[j], sl[j]);
    }
    if (a_tp) {
#pragma unroll
      for (int j = 0; j < 2; ++j) tp[j] = soft_elem(tp[j], p[j], a_tau);
    }
    pn[2 * h] = p[0]; pn[2 * h + 1] = p[1];
    tn4[2 * h] = tp[0]; tn4[2 * h + 1] = tp[1];
    if (ok[h]) {
      *(float2*)(a_m + idx[h]) = make_float2(m[0], m[1]);
      *(float2*)(a_v + idx[h]) = make_float2(v[0], v[1]);
      *(float2*)(a_p + idx[h]) = make_float2(p[0], p[1]);
      if (S.la_sync) *(float2*)(a_slow + idx[h]) = make_float2(sl[0], sl[1]);
      if (a_g) *(float2*)(a_g + idx[h]) = make_float2(g[2 * h], g[2 * h + 1]);
      if (a_tp) *(float2*)(a_tp + idx[h]) = make_float2(tp[0], tp[1]);
    }
  }
  DWA_STAMP(5);
  if (X3 && t_sh_off >= 0) {                      // split-bf16 shadow(s): hi at the mapped column, lo 32 elements further (x3.h)
    const int c0 = n0 + enq;
    const int64_t se = t_sh_off + (int64_t)row * t_sh_ld;
    if (ok[1]) {
      uint2 hi, lo;
      if (a_sh) { x3_split4(pn, hi, lo); *(uint2*)(a_sh + se + x3_col(c0)) = hi; *(uint2*)(a_sh + se + x3_col(c0) + 32) = lo; }
      if (a_tp && a_tsh) { x3_split4(tn4, hi, lo); *(uint2*)(a_tsh + se + x3_col(c0)) = hi; *(uint2*)(a_tsh + se + x3_col(c0) + 32) = lo; }
    } else if (ok[0]) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (a_sh) x3_store(a_sh + se, c0 + j, pn[j]);
        if (a_tp && a_tsh) x3_store(a_tsh + se, c0 + j, tn4[j]);
      }
    }
  } else if (t_sh_off >= 0) {                    // the bf16 compute shadow(s): 8 bytes per thread (4 when the row ends inside the quad)
    const int64_t se = t_sh_off + (int64_t)row * t_sh_ld + n0 + enq;
    if (ok[1]) {
      if (a_sh) *(uint2*)(a_sh + se) = make_uint2(pack_bf2(pn[0], pn[1]), pack_bf2(pn[2], pn[3]));
      if (a_tp && a_tsh) *(uint2*)(a_tsh + se) = make_uint2(pack_bf2(tn4[0], tn4[1]), pack_bf2(tn4[2], tn4[3]));
    } else if (ok[0]) {
      if (a_sh) *(uint32_t*)(a_sh + se) = pack_bf2(pn[0], pn[1]);
      if (a_tp && a_tsh) *(uint32_t*)(a_tsh + se) = pack_bf2(tn4[0], tn4[1]);
    }
  }
  DWA_STAMP(6);
  if (trow) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trow[7] = __builtin_amdgcn_s_memtime(); }
}

__global__ __launch_bounds__((NC + NL) * 64) void dw_adam_kernel(const DwAdamBatch batch, unsigned long long* trace) {
  kernarg_prefetch<(int)sizeof(DwAdamNet)>((int)(blockIdx.y * sizeof(DwAdamNet)));
  const DwAdamNet& N = batch.n[blockIdx.y];
  const int nwg = N.nsmall + N.ntile[0] + N.ntile[1];
  if ((int)blockIdx.x >= nwg) return;
  unsigned long long* trow = (trace && threadIdx.x == 0) ? trace + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  asm volatile("" : "+v"(trow));
  DWA_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lid = xcd_remap(blockIdx.x, nwg);    // consecutive logical ids share an XCD: the tiles of one X column block read it from one L2
  if (lid < N.nsmall) {
    // ---- the tensors that are not tiled: apply_kernel's workgroups (optim_dev.h), 256 threads
    if (threadIdx.x >= 256) return;
    __shared__ float red[4];
    __shared__ float sp[4][OPT_SMALL_ELEMS];
    const int nb1 = N.L.t[2].blk0 - N.L.t[1].blk0;          // blocks of b1 (tensor 1); the rest follows w2: b2, w3, b3
    const int b = lid < nb1 ? N.L.t[1].blk0 + lid : N.L.t[3].blk0 + (lid - nb1);
    apply_body(N.L, N.a, b, red, sp);
    return;
  }
  const int t0 = lid - N.nsmall;
  const int wi = t0 < N.ntile[0] ? 0 : 1;
  const DwAdamProb& P = N.w[wi];
  const int tile = wi == 0 ? t0 : t0 - N.ntile[0];
  if (N.x3) {
    if (P.nslab == 8) tile_role<1, true>(N, P, tile, lds, trow);
    else tile_role<2, true>(N, P, tile, lds, trow);
  } else if (P.nslab == 8) tile_role<1, false>(N, P, tile, lds, trow);
  else tile_role<2, false>(N, P, tile, lds, trow);
}

unsigned long long* g_dwadam_trace = nullptr;
}  // namespace

extern "C" void recnn_debug_dwadam_trace(void* p) { g_dwadam_trace = (unsigned long long*)p; }   // [workgroup][8] uint64 shader-clock stamps

int dwadam_init() {
  return recnn_check_hip(hipFuncSetAttribute((const void*)dw_adam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL), "dw_adam attr");
}

bool dwadam_tensor_ok(const NetLayout& L, int ti, int rows, int x3) {
  const TensorSeg& T = L.t[ti];
  if (rows < 256 || rows % 256) return false;                       // every consumer wave multiplies whole 32-row steps
  if (T.nslab != 8 && T.nslab != 16) return false;
  if (x3) {
    // split bf16: slabs are whole 128-row stages (x3.hip: chunk = roundup(ceil(rows / nslab), 32))
    const int chunk = (((rows + T.nslab - 1) / T.nslab) + 31) / 32 * 32;
    if (chunk * T.nslab != rows || chunk % 128) return false;
    if (T.sh_ld < 2 * ((T.cols + 63) / 64 * 64)) return false;
  } else {
    // the slabs of the two-launch path must be the waves' ranges (gemm.hip: chunk = roundup(ceil(rows / nslab), 64))
    const int chunk = (((rows + T.nslab - 1) / T.nslab) + 63) / 64 * 64;
    if (chunk * T.nslab != rows || (rows / 8) % chunk) return false;
    if (T.sh_ld < (T.cols + 63) / 64 * 64) return false;            // a 64-column tile is readable inside the shadow / batch row pitch
  }
  if (T.rows % 32 || (T.cols & 1) || (T.col_rot & 1) || (T.p_off & 1) || T.sh_off < 0 || (T.sh_off & 3) || (T.sh_ld & 3)) return false;
  return true;
}

int dwadam_launch(DwAdamBatch& b, int nnet, hipStream_t s) {
  RECNN_REQUIRE(nnet >= 1 && nnet <= DWADAM_MAX_NETS, "dw_adam: 1..%d networks per launch", DWADAM_MAX_NETS);
  int maxwg = 0;
  for (int i = 0; i < nnet; ++i) {
    DwAdamNet& n = b.n[i];
    apply_args_finish(&n.a);
    RECNN_REQUIRE(n.a.do_adam && n.a.from_slabs && !n.a.comm.world && n.a.n_l1 == 0 && (n.a.tc_bf16 == RECNN_BF16 || n.a.tc_bf16 == RECNN_BF16X3) &&
                      n.a.p && n.a.m && n.a.v,
                  "dw_adam: single-GPU bf16 / split-bf16 optimizer steps without the clip quirk only");
    n.x3 = n.a.tc_bf16 == RECNN_BF16X3;
    RECNN_REQUIRE(!(((uintptr_t)n.a.p | (uintptr_t)n.a.m | (uintptr_t)n.a.v | (uintptr_t)n.a.g_out | (uintptr_t)n.a.tgt_p | (uintptr_t)n.a.slow) & 7),
                  "dw_adam: the flat arenas must be 8-byte aligned");
    RECNN_REQUIRE(!(((uintptr_t)n.a.shadow | (uintptr_t)n.a.tgt_shadow) & 7), "dw_adam: the shadow arenas must be 8-byte aligned");
    RECNN_REQUIRE(n.a.opt_kind != RECNN_OPT_RANGER || n.a.slow, "dw_adam: Ranger needs the slow-weight arena");
    for (int w = 0; w < 2; ++w) {
      DwAdamProb& p = n.w[w];
      RECNN_REQUIRE(dwadam_tensor_ok(n.L, p.tensor, n.rows, n.x3), "dw_adam: tensor %d of network %d does not fit the tile plan at %d rows", p.tensor, i, n.rows);
      const TensorSeg& T = n.L.t[p.tensor];
      const int ph = n.x3 ? 2 : 1;                // physical elements per logical column
      RECNN_REQUIRE(p.dz && p.x && p.ldz >= ph * T.rows && p.ldx >= ph * ((T.cols + 63) / 64 * 64) && !(((uintptr_t)p.dz | (uintptr_t)p.x) & 15) &&
                        p.ldz % 8 == 0 && p.ldx % 8 == 0 && (int64_t)n.rows * p.ldx * 2 < (1ll << 31),
                    "dw_adam: bad operands");
      p.nslab = T.nslab;
      p.tiles_m = T.rows / 32;
      p.tiles_n = (T.cols + 63) / 64;
      n.ntile[w] = p.tiles_m * p.tiles_n;
    }
    // the small tensors' optimizer blocks: b1 between w1 and w2, then b2, w3, b3 (tensor order of the layout)
    n.nsmall = (n.L.t[2].blk0 - n.L.t[1].blk0) + (n.L.nblk - n.L.t[3].blk0);
    const int nwg = n.nsmall + n.ntile[0] + n.ntile[1];
    if (nwg > maxwg) maxwg = nwg;
  }
  hipLaunchKernelGGL(dw_adam_kernel, dim3(maxwg, nnet), dim3((NC + NL) * 64), LDS_TOTAL, s, b, g_dwadam_trace);
  return recnn_check_hip(hipGetLastError(), "dw_adam_kernel");
}
