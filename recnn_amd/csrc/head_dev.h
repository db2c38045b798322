// head_dev.h -- device code of the critic head (head.hip), shared with the split-bf16 row-panel tail (x3tail.hip), which runs the
// head of its panel's rows itself once the target critic's Q' exists (no head launch on the chain).
#pragma once
#include <type_traits>
#include "head.h"
#include "x3.h"

// compute-type tags of the templates below: float, bf16_t, or x3_t = split-bf16 rows (x3.h: the column index is mapped, a value is hi + lo)
struct x3_t { bf16_t v; };
template <class TC> struct HeadStore { using type = TC; };
template <> struct HeadStore<x3_t> { using type = bf16_t; };

// ---------------------------------------------------------------- row dots: one wave per row
// h: start of the row, k: first of four consecutive logical columns (k % 4 == 0)
template <class TC> __device__ inline float dot4(const typename HeadStore<TC>::type* __restrict__ hrow, int k, const float4 wv) {
  float x0, x1, x2, x3;
  if constexpr (std::is_same<TC, x3_t>::value) {
    const bf16_t* h = hrow + x3_col(k);
    const uint2 hv = *(const uint2*)h, lv = *(const uint2*)(h + 32);
    x0 = bf2f((bf16_t)(hv.x & 0xFFFF)) + bf2f((bf16_t)(lv.x & 0xFFFF)); x1 = bf2f((bf16_t)(hv.x >> 16)) + bf2f((bf16_t)(lv.x >> 16));
    x2 = bf2f((bf16_t)(hv.y & 0xFFFF)) + bf2f((bf16_t)(lv.y & 0xFFFF)); x3 = bf2f((bf16_t)(hv.y >> 16)) + bf2f((bf16_t)(lv.y >> 16));
    return x0 * wv.x + x1 * wv.y + x2 * wv.z + x3 * wv.w;
  }
  const typename HeadStore<TC>::type* h = hrow + k;
  if constexpr (sizeof(TC) == 4) {
    const float4 hv = *(const float4*)h;
    x0 = hv.x; x1 = hv.y; x2 = hv.z; x3 = hv.w;
  } else {
    const uint2 hv = *(const uint2*)h;
    x0 = bf2f((bf16_t)(hv.x & 0xFFFF)); x1 = bf2f((bf16_t)(hv.x >> 16));
    x2 = bf2f((bf16_t)(hv.y & 0xFFFF)); x3 = bf2f((bf16_t)(hv.y >> 16));
  }
  return x0 * wv.x + x1 * wv.y + x2 * wv.z + x3 * wv.w;
}

// 16 rows per block.  Phase 1: one wave per row, 4 rows per wave; NT target heads and NC critic heads are compile-time
// so the loads of all 4 x (NT + NC) row segments (and of reward / done / biases) are in flight together -- the kernel
// is a chain of memory latencies, not bandwidth.  TD target, Q, dQ, loss partial.  Phase 2 (do_bwd): dz2 and the
// partial sums of dW3 / db2 / db3.  One launch instead of two, dQ never leaves the CU.
// head_block: the work of ONE 16-row block by 256 threads (tid = 0..255) of a workgroup that may be larger (x3tail.hip runs the
// head of its 32-row panel as two blocks on waves 0..7; every wave of the workgroup calls -- the barriers are workgroup-wide --
// with active = false for those that have no block).  sm: this block's scratch; tq_lds (PRE only, optional): Q' of the block's 16
// rows ... see HeadRows.
struct HeadSmem {
  float part[4][HEAD_MAX_CRITIC];
  float sdelta[HEAD_MAX_CRITIC][HEAD_ROWS_PER_BLOCK];
  float red_w[8][256 + 8];
  float red_b[8][256 + 8];
};
// HeadRows (optional): where the block's per-row inputs are read instead of the HeadArgs pointers -- x3tail.hip keeps them in LDS and
// passes pointers to the copies of rows row0 .. (critic 0 / target 0 only).
struct HeadRows {
  const void* ch2;      // rows row0 .. of critic 0's h2, pitch a.ld_h
  const float* tq;      // [row - row0]
  const float* reward;
  const float* done;
  int row0;
};
template <class TC, int NT, int NC, bool PRE>
__device__ __forceinline__ void head_block(const HeadArgs& a, const int blk, const int tid, const bool active, HeadSmem& sm, const HeadRows* ov = nullptr) {
  using ST = typename HeadStore<TC>::type;
  constexpr bool X3 = std::is_same<TC, x3_t>::value;
  float (&part)[4][HEAD_MAX_CRITIC] = sm.part;
  float (&sdelta)[HEAD_MAX_CRITIC][HEAD_ROWS_PER_BLOCK] = sm.sdelta;
  float (&red_w)[8][256 + 8] = sm.red_w;
  float (&red_b)[8][256 + 8] = sm.red_b;
  const int lane = tid & 63, wave = tid >> 6;
  const int r0 = blk * HEAD_ROWS_PER_BLOCK;
  if (active) {
    float acc[HEAD_MAX_CRITIC];
#pragma unroll
    for (int c = 0; c < HEAD_MAX_CRITIC; ++c) acc[c] = 0.f;
    constexpr int ND = PRE ? 0 : NT;  // target heads that still need their row dot
    float st[NT > 0 ? NT : 1][4], sc[NC][4], rew[4], dn[4];
    int64_t roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = min(r0 + wave * 4 + i, a.rows - 1);  // clamp: tail rows recompute the last row, results unused
      roff[i] = (int64_t)r * a.ld_h;
      rew[i] = NT > 0 ? (ov ? ov->reward[r - ov->row0] : a.reward[r]) : 0.f;
      dn[i] = NT > 0 ? (ov ? ov->done[r - ov->row0] : a.done[r]) : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) st[t][i] = PRE ? ((ov && t == 0) ? ov->tq[r - ov->row0] : a.tq_in[t][r]) : 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) sc[c][i] = 0.f;
    }
    float tb[NT > 0 ? NT : 1], cb[NC];
#pragma unroll
    for (int t = 0; t < NT; ++t) tb[t] = PRE ? 0.f : a.tb3[t][0];
#pragma unroll
    for (int c = 0; c < NC; ++c) cb[c] = a.cb3[c][0];
    for (int k = lane * 4; k < a.H; k += 256) {
      float4 wt[NT > 0 ? NT : 1], wc[NC];
#pragma unroll
      for (int t = 0; t < ND; ++t) wt[t] = *(const float4*)(a.tw3[t] + k);
#pragma unroll
      for (int c = 0; c < NC; ++c) wc[c] = *(const float4*)(a.cw3[c] + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int t = 0; t < ND; ++t) st[t][i] += dot4<TC>((const ST*)a.th2[t] + roff[i], k, wt[t]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
          sc[c][i] += (ov && c == 0) ? dot4<TC>((const ST*)ov->ch2 + (roff[i] - (int64_t)ov->row0 * a.ld_h), k, wc[c]) : dot4<TC>((const ST*)a.ch2[c] + roff[i], k, wc[c]);
      }
    }
    float tq[4], qv[HEAD_MAX_CRITIC][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tq[i] = 0.f;
      if constexpr (NT > 0) {
        tq[i] = PRE ? st[0][i] : wave_sum(st[0][i]) + tb[0];
        if constexpr (NT > 1) tq[i] = fminf(tq[i], PRE ? st[1][i] : wave_sum(st[1][i]) + tb[1]);
      }
#pragma unroll
      for (int c = 0; c < HEAD_MAX_CRITIC; ++c) qv[c][i] = c < NC ? wave_sum(sc[c < NC ? c : 0][i]) + cb[c < NC ? c : 0] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + wave * 4 + i;
      const bool valid = r < a.rows;
      float y = 0.f;
      if (NT > 0 && valid) {
        y = rew[i] + (1.0f - dn[i]) * a.gamma * tq[i];
        y = fminf(fmaxf(y, a.lo), a.hi);
        if (lane == 0) {
          if (a.expected) a.expected[r] = y;
          if (a.target_q) a.target_q[r] = tq[i];
        }
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float q = qv[c][i];
        float d;
        if (a.policy_mode) {
          d = a.delta_const;
          if (valid) acc[c] += q;
        } else {
          const float e = q - y;
          d = e * (2.0f / (float)a.rows);
          if (valid) acc[c] += e * e;
          if (valid && lane == 0 && a.delta[c]) a.delta[c][r] = d;
        }
        if (lane == 0) {
          sdelta[c][wave * 4 + i] = valid ? d : 0.f;
          if (valid && a.q[c]) a.q[c][r] = q;
        }
      }
    }
    if (lane == 0)
      for (int c = 0; c < a.n_critic; ++c) part[wave][c] = acc[c];
  }
  __syncthreads();
  if (active && tid < a.n_critic) {
    const int c = tid;
    a.loss_part[c][blk] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
  }
  if (!a.do_bwd) return;
  // Phase 2, vectorised: a thread owns 8 consecutive hidden columns (one 16-byte bf16 load/store per row, two for
  // fp32) of rows {rg, rg+8} of the block (256 threads = 32 column groups x 8 row groups); the per-column sums over the
  // 16 rows are combined across the 8 row groups through LDS in a fixed order (deterministic).
  const int nr = min(HEAD_ROWS_PER_BLOCK, a.rows - r0);
  const float scale = a.train ? 2.0f : 1.0f;
  const int cg = tid & 31, rg = tid >> 5;
  for (int c = 0; c < a.n_critic; ++c) {
    const bool hov = ov && c == 0;
    const ST* h2 = (const ST*)(hov ? ov->ch2 : a.ch2[c]);
    ST* dz2 = (ST*)a.dz2[c];
    for (int n0 = 0; n0 < a.H; n0 += 256) {
      const int n = n0 + cg * 8;
      float w[8], sw[8], sb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        w[j] = (n + j < a.H) ? a.cw3[c][n + j] * scale : 0.f;
        sw[j] = 0.f;
        sb[j] = 0.f;
      }
      if (active && n < a.H) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int i = rg + half * 8;
          if (i < nr) {
            const int64_t off = (int64_t)(r0 + i) * a.ld_h + (X3 ? x3_col(n) : n);     // dz2 (global)
            const int64_t offh = hov ? off - (int64_t)ov->row0 * a.ld_h : off;               // h2 (global, or the caller's copy)
            float hv[8];
            if constexpr (X3) {   // 8 logical columns: one 16-byte load of the hi halves, one of the lo halves
              const uint4 rh = *(const uint4*)(h2 + offh), rl = *(const uint4*)(h2 + offh + 32);
              const uint32_t uh[4] = {rh.x, rh.y, rh.z, rh.w}, ul[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                hv[2 * j] = bf2f((bf16_t)(uh[j] & 0xFFFF)) + bf2f((bf16_t)(ul[j] & 0xFFFF));
                hv[2 * j + 1] = bf2f((bf16_t)(uh[j] >> 16)) + bf2f((bf16_t)(ul[j] >> 16));
              }
            } else if constexpr (sizeof(TC) == 2) {
              const uint4 raw = *(const uint4*)(h2 + offh);
              const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                hv[2 * j] = bf2f((bf16_t)(u[j] & 0xFFFF));
                hv[2 * j + 1] = bf2f((bf16_t)(u[j] >> 16));
              }
            } else {
              const float4 x0 = *(const float4*)(h2 + offh), x1 = *(const float4*)(h2 + offh + 4);
              hv[0] = x0.x; hv[1] = x0.y; hv[2] = x0.z; hv[3] = x0.w; hv[4] = x1.x; hv[5] = x1.y; hv[6] = x1.z; hv[7] = x1.w;
            }
            const float d = sdelta[c][i];
            float dz[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              dz[j] = hv[j] > 0.f ? d * w[j] : 0.f;
              sw[j] += d * hv[j];
              sb[j] += dz[j];
            }
            if constexpr (X3) {
              uint2 h0, l0, h1, l1;
              const float d0[4] = {dz[0], dz[1], dz[2], dz[3]}, d1[4] = {dz[4], dz[5], dz[6], dz[7]};
              x3_split4(d0, h0, l0);
              x3_split4(d1, h1, l1);
              *(uint4*)(dz2 + off) = make_uint4(h0.x, h0.y, h1.x, h1.y);
              *(uint4*)(dz2 + off + 32) = make_uint4(l0.x, l0.y, l1.x, l1.y);
            } else if constexpr (sizeof(TC) == 2) {
              *(uint4*)(dz2 + off) = make_uint4(pack_bf2(dz[0], dz[1]), pack_bf2(dz[2], dz[3]), pack_bf2(dz[4], dz[5]),
                                                pack_bf2(dz[6], dz[7]));
            } else {
              *(float4*)(dz2 + off) = make_float4(dz[0], dz[1], dz[2], dz[3]);
              *(float4*)(dz2 + off + 4) = make_float4(dz[4], dz[5], dz[6], dz[7]);
            }
          }
        }
      }
      if (a.dw3_part[c]) {
        __syncthreads();
        if (active) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            red_w[rg][cg * 8 + j] = sw[j];
            red_b[rg][cg * 8 + j] = sb[j];
          }
        }
        __syncthreads();
        const int col = tid;
        if (active && n0 + col < a.H) {
          float tw = 0.f, tb = 0.f;
#pragma unroll
          for (int g = 0; g < 8; ++g) { tw += red_w[g][col]; tb += red_b[g][col]; }
          a.dw3_part[c][(int64_t)blk * a.H + n0 + col] = tw;
          a.db2_part[c][(int64_t)blk * a.H + n0 + col] = tb;
        }
      }
    }
    if (active && a.db3_part[c] && tid == 0) {
      float s = 0.f;
      for (int i = 0; i < nr; ++i) s += sdelta[c][i];
      a.db3_part[c][blk] = s;
    }
  }
}

