// gather_dev.h -- device body of the replay sampler + embedding gather (see gather.hip for the description);
// shared by frame_gather_kernel (gather.hip) and apply_gather_kernel (optim.hip).
#pragma once
#include "gather.h"
#include "x3.h"

template <int W> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<1> { using type = float; };

template <int W> __device__ __forceinline__ void store_f4(float* p, const float4 v) {
  if constexpr (W == 4) {
    *(float4*)p = v;
  } else if constexpr (W == 2) {
    *(float2*)p = make_float2(v.x, v.y);
    *(float2*)(p + 2) = make_float2(v.z, v.w);
  } else {
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
  }
}

// R rows per workgroup; W = floats per output store (4/2/1 by output alignment).  `block` is the workgroup's index
// among the gather workgroups (the body also runs as extra workgroups of the optimizer launch, optim.hip).
//
// Round 2: no LDS staging.  Consecutive rows of one user share F of their F+1 embedding lines (sliding window) and
// state / next_state / action of one row share all of them; every DISTINCT line of the workgroup's R rows has one owner
// (the first row of the workgroup that contains it) whose threads load its 16-byte chunks once and store them straight
// to every (row, slot) that uses the line -- the LDS round trip (write, barrier, read) of round 1 sat on a dependent chain
// users -> offsets -> items -> table -> stores that is latency, not bandwidth.  LDS only holds the per-row metadata (and
// the inline plan's prefix sums).
template <int R, int W>
__device__ __forceinline__ void frame_gather_body(const GatherArgs& a, const int block, unsigned char* smem_raw) {
  const int F = a.frame, E = a.emb, F1 = F + 1;
  long long* m_src = (long long*)smem_raw;         // CSR offset of the window start, per row
  long long* s_start = m_src + R;                  // inline plan: CSR offset of each batch user's history [n_users]
  const bool scan = a.inline_plan && !a.plan;      // (must match frame_gather_lds_bytes)
  int* m_flag = (int*)(s_start + (scan ? a.n_users : 0));   // per row: bit 0 valid, bit 1 done
  int* s_off = m_flag + R;                         // inline plan: row prefix sums [n_users + 1]
  int* s_sc = s_off + a.n_users + 1;               // inline plan: per-wave totals [4]

  const int tid = threadIdx.x;
  const int row0 = block * R;
  const int32_t* users = a.users;
  int cur = 0;
  if (a.cursor) {  // device-side batch cursor (graph replay); cursor_add looks ahead (batch of the NEXT step)
    cur = *a.cursor + a.cursor_add;
    if (a.cursor_mod > 0 && cur >= a.cursor_mod) cur -= a.cursor_mod;
    users += (int64_t)cur * a.cursor_stride;
  }

  // ---- every thread: the R rows' window starts and flags (bit 0 valid, bit 1 done)
  long long src[R];
  int fl[R];
  if (a.plan) {
    // per-epoch plan table: a row's window start and done flag are ONE load away from the cursor (R entries of one or two
    // cache lines, the same for every thread: no LDS, no barrier)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long p = row0 + r < a.rows ? a.plan[(int64_t)cur * a.plan_stride + row0 + r] : -1;
      src[r] = p >> 1;
      fl[r] = p >= 0 ? (1 | ((int)(p & 1) << 1)) : 0;
    }
  } else {
    const int* row_off = a.row_off;
    if (scan) {
      // exclusive prefix sum of max(L_u - F, 0) over the batch's users, recomputed per workgroup (a few hundred
      // L2-resident loads) instead of a separate single-workgroup plan launch ahead of the gather
      const int n = a.n_users;
      const int per = (n + 255) / 256;
      int lens[4], sum = 0;
      long long starts[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid * per + j;
        int v = 0;
        long long o0 = 0;
        if (j < per && i < n) {
          const int su = users[i];
          o0 = a.user_off[su];
          v = max((int)(a.user_off[su + 1] - o0) - F, 0);
        }
        lens[j] = v;
        starts[j] = o0;
        sum += v;
      }
      // block-wide exclusive scan: shuffles inside a wave, one barrier to combine the four wave totals
      const int lane = tid & 63, wave = tid >> 6;
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      if (lane == 63) s_sc[wave] = incl;
      __syncthreads();
      int run = incl - sum;
      for (int w = 0; w < wave; ++w) run += s_sc[w];
      if (tid == 0) s_off[0] = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid * per + j;
        if (j < per && i < n) { run += lens[j]; s_off[i + 1] = run; s_start[i] = starts[j]; }
      }
      __syncthreads();
      row_off = s_off;
    }
    if (tid < R) {
      const int r = row0 + tid;
      // rows past the planned total (fewer windows than requested) are left untouched
      const int valid = r < a.rows && r < row_off[a.n_users];
      int last = 0;
      long long s0 = 0;
      if (valid) {
        // largest i with row_off[i] <= r   (row_off is non-decreasing, row_off[n_users] > r)
        int lo = 0, hi = a.n_users;
        while (hi - lo > 1) {
          int mid = (lo + hi) >> 1;
          if (row_off[mid] <= r) lo = mid; else hi = mid;
        }
        const int t = r - row_off[lo];
        last = t == row_off[lo + 1] - row_off[lo] - 1;  // the user's final window: done = 1 (utils.py:70-71)
        s0 = (scan ? s_start[lo] : a.user_off[users[lo]]) + t;
      }
      m_src[tid] = s0;
      m_flag[tid] = valid | ((valid && last) << 1);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      src[r] = m_src[r];
      fl[r] = m_flag[r];
    }
  }

  // which rows continue the previous row's window, and the owned-line prefix
  bool valid[R], cont[R];
  int base[R + 1];
  base[0] = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    valid[r] = fl[r] & 1;
    if (valid[r] && tid == r) a.done[row0 + r] = (fl[r] & 2) ? 1.f : 0.f;
    // window shifted by one against the previous row's (same user, previous row not the user's last): shares F lines with it
    cont[r] = r > 0 && valid[r] && valid[r - 1] && !(fl[r - 1] & 2) && src[r] == src[r - 1] + 1;
    base[r + 1] = base[r] + (valid[r] ? (cont[r] ? 1 : F1) : 0);
  }
  const int total = base[R];
  const int FE = F * E, E4 = E >> 2;

  // ---- ratings: one thread per owned (row, slot); state / next_state tails and the reward
  for (int l = tid; l < total; l += 256) {
    int r = 0;
#pragma unroll
    for (int q = 1; q < R; ++q) r = l >= base[q] ? q : r;
    const int j = cont[r] ? F : l - base[r];
    const float v = a.ratings[src[r] + j];
    if (j == F) a.reward[row0 + r] = v;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int rr = r + k, jj = j - k;
      if (rr >= R || jj < 0 || (k > 0 && !cont[rr])) break;
      const int64_t row = row0 + rr;
      if (a.state) {
        if (jj < F) a.state[row * a.ld_state + FE + jj] = v;
        if (jj >= 1) a.next_state[row * a.ld_next + FE + jj - 1] = v;
      }
      if (a.state_h && a.x3) {
        if (jj < F) x3_store(a.state_h + row * a.ld_h, FE + jj, v);
        if (jj >= 1) x3_store(a.next_h + row * a.ld_h, FE + jj - 1, v);
      } else if (a.state_h) {
        if (jj < F) a.state_h[row * a.ld_h + FE + jj] = f2bf(v);
        if (jj >= 1) a.next_h[row * a.ld_h + FE + jj - 1] = f2bf(v);
      }
    }
  }

  // ---- embedding lines: 16-byte chunk `e4` of owned line l; U chunks per thread in flight (ids, then lines, then stores)
  constexpr int U = 4;
  const int nchunk = total * E4;
  for (int c0 = 0; c0 < nchunk; c0 += 256 * U) {
    int rr0[U], jj0[U], ee[U];
    int item[U];
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = c0 + u * 256 + tid;
      rr0[u] = -1;
      if (idx < nchunk) {
        const int l = idx / E4;
        ee[u] = (idx - l * E4) * 4;
        int r = 0;
#pragma unroll
        for (int q = 1; q < R; ++q) r = l >= base[q] ? q : r;
        rr0[u] = r;
        jj0[u] = cont[r] ? F : l - base[r];
        item[u] = a.items[src[r] + jj0[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (rr0[u] >= 0) v[u] = *(const float4*)(a.table + (int64_t)item[u] * E + ee[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (rr0[u] < 0) continue;
      uint2 h = make_uint2(pack_bf2(v[u].x, v[u].y), pack_bf2(v[u].z, v[u].w)), hl = make_uint2(0, 0);
      if (a.x3) {
        const float v4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        x3_split4(v4, h, hl);
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int rr = rr0[u] + k, jj = jj0[u] - k;
        if (rr >= R || jj < 0 || (k > 0 && !cont[rr])) break;
        const int64_t row = row0 + rr;
        if (a.state) {
          if (jj < F) store_f4<W>(a.state + row * a.ld_state + jj * E + ee[u], v[u]);
          if (jj >= 1) store_f4<W>(a.next_state + row * a.ld_next + (jj - 1) * E + ee[u], v[u]);
          if (jj == F) store_f4<W>(a.action + row * a.ld_action + ee[u], v[u]);
        }
        if constexpr (W == 4) {
          if (a.state_h && a.x3) {
            if (jj < F) { bf16_t* d = a.state_h + row * a.ld_h + x3_col(jj * E + ee[u]); *(uint2*)d = h; *(uint2*)(d + 32) = hl; }
            if (jj >= 1) { bf16_t* d = a.next_h + row * a.ld_h + x3_col((jj - 1) * E + ee[u]); *(uint2*)d = h; *(uint2*)(d + 32) = hl; }
            if (jj == F) { bf16_t* d = a.action_h + row * a.ld_h + x3_col(ee[u]); *(uint2*)d = h; *(uint2*)(d + 32) = hl; }
          } else if (a.state_h) {
            if (jj < F) *(uint2*)(a.state_h + row * a.ld_h + jj * E + ee[u]) = h;
            if (jj >= 1) *(uint2*)(a.next_h + row * a.ld_h + (jj - 1) * E + ee[u]) = h;
            if (jj == F) *(uint2*)(a.action_h + row * a.ld_h + ee[u]) = h;
          }
        }
      }
    }
  }
}
