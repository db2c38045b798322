// gather_dev.h -- device body of the replay sampler + embedding gather (see gather.hip for the description);
// shared by frame_gather_kernel (gather.hip) and apply_gather_kernel (optim.hip).
#pragma once
#include "gather.h"

template <int W> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<1> { using type = float; };

// R rows per workgroup; W = floats per output store (4/2/1 by output alignment).  `block` is the workgroup's index
// among the gather workgroups (the body also runs as extra workgroups of the optimizer launch, optim.hip).
template <int R, int W>
__device__ __forceinline__ void frame_gather_body(const GatherArgs& a, const int block, unsigned char* smem_raw) {
  using V = typename VecT<W>::type;
  const int F = a.frame, E = a.emb, F1 = F + 1;
  float* lines = (float*)smem_raw;                 // [R*F1][E]
  float* rat = lines + (size_t)R * F1 * E;         // [R*F1]
  int* meta = (int*)(rat + R * F1);                // per row: base line, cont flag, done flag; + src offset (2 ints)
  int* m_base = meta;
  int* m_cont = meta + R;
  int* m_done = meta + 2 * R;
  int* m_valid = meta + 3 * R;
  long long* m_src = (long long*)(meta + 4 * R);   // CSR offset of the window start
  long long* s_start = m_src + R;                  // inline plan: CSR offset of each batch user's history [n_users]
  int* s_off = (int*)(s_start + (a.inline_plan ? a.n_users : 0));  // inline plan: row prefix sums [n_users + 1]
  int* s_sc = s_off + a.n_users + 1;               // inline plan: per-wave totals [4]

  const int tid = threadIdx.x;
  const int row0 = block * R;
  const int32_t* users = a.users;
  if (a.cursor) {  // device-side batch cursor (graph replay); cursor_add looks ahead (batch of the NEXT step)
    int cur = *a.cursor + a.cursor_add;
    if (a.cursor_mod > 0 && cur >= a.cursor_mod) cur -= a.cursor_mod;
    users += (int64_t)cur * a.cursor_stride;
  }

  const int* row_off = a.row_off;
  if (a.inline_plan) {
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
    int valid = r < a.rows && r < row_off[a.n_users];
    int u = 0, t = 0, last = 0;
    long long src = 0;
    if (valid) {
      // largest i with row_off[i] <= r   (row_off is non-decreasing, row_off[n_users] > r)
      int lo = 0, hi = a.n_users;
      while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (row_off[mid] <= r) lo = mid; else hi = mid;
      }
      u = lo;
      t = r - row_off[lo];
      last = t == row_off[lo + 1] - row_off[lo] - 1;  // the user's final window: done = 1 (utils.py:70-71)
      src = (a.inline_plan ? s_start[u] : a.user_off[users[u]]) + t;
    }
    m_valid[tid] = valid;
    m_src[tid] = src;
    m_done[tid] = valid && last;
    // continuation of the previous row's window (same user => shifted by one)
    m_cont[tid] = 0;
    m_base[tid] = u;  // temporarily the user index
  }
  __syncthreads();
  if (tid == 0) {
    int base = 0, prev_u = -1, prev_valid = 0;
    for (int r = 0; r < R; ++r) {
      int u = m_base[r];
      int cont = r > 0 && prev_valid && m_valid[r] && u == prev_u;
      if (r > 0) base += cont ? 1 : F1;
      m_cont[r] = cont;
      prev_u = u;
      prev_valid = m_valid[r];
      m_base[r] = base;
    }
  }
  __syncthreads();

  // ---- stage the distinct embedding lines + ratings into LDS
  const int E4 = E >> 2;
  for (int idx = tid; idx < R * F1 * E4; idx += 256) {
    const int pair = idx / E4, e4 = idx - pair * E4;
    const int r = pair / F1, j = pair - r * F1;
    if (!m_valid[r] || (m_cont[r] && j < F)) continue;
    const int item = a.items[m_src[r] + j];
    const float4 v = *(const float4*)(a.table + (int64_t)item * E + e4 * 4);
    *(float4*)(lines + (size_t)(m_base[r] + j) * E + e4 * 4) = v;
  }
  for (int pair = tid; pair < R * F1; pair += 256) {
    const int r = pair / F1, j = pair - r * F1;
    if (!m_valid[r] || (m_cont[r] && j < F)) continue;
    rat[m_base[r] + j] = a.ratings[m_src[r] + j];
  }
  __syncthreads();

  // ---- stream out: state / next_state embedding parts and the action
  const int FE = F * E;
  const int per_row = FE / W;
  for (int idx = tid; idx < R * per_row; idx += 256) {
    const int r = idx / per_row, q = idx - r * per_row;
    if (!m_valid[r]) continue;
    const float* src = lines + (size_t)m_base[r] * E + q * W;
    if (a.state) {
      *(V*)(a.state + (int64_t)(row0 + r) * a.ld_state + q * W) = *(const V*)src;
      *(V*)(a.next_state + (int64_t)(row0 + r) * a.ld_next + q * W) = *(const V*)(src + E);
    }
    if constexpr (W == 4) {
      if (a.state_h) {
        *(uint2*)(a.state_h + (int64_t)(row0 + r) * a.ld_h + q * 4) = make_uint2(pack_bf2(src[0], src[1]), pack_bf2(src[2], src[3]));
        *(uint2*)(a.next_h + (int64_t)(row0 + r) * a.ld_h + q * 4) =
            make_uint2(pack_bf2(src[E], src[E + 1]), pack_bf2(src[E + 2], src[E + 3]));
      }
    }
  }
  const int per_act = E / W;
  for (int idx = tid; idx < R * per_act; idx += 256) {
    const int r = idx / per_act, q = idx - r * per_act;
    if (!m_valid[r]) continue;
    const float* src = lines + (size_t)(m_base[r] + F) * E + q * W;
    if (a.action) *(V*)(a.action + (int64_t)(row0 + r) * a.ld_action + q * W) = *(const V*)src;
    if constexpr (W == 4) {
      if (a.action_h)
        *(uint2*)(a.action_h + (int64_t)(row0 + r) * a.ld_h + q * 4) = make_uint2(pack_bf2(src[0], src[1]), pack_bf2(src[2], src[3]));
    }
  }
  // ---- ratings tails, reward, done
  for (int idx = tid; idx < R * F; idx += 256) {
    const int r = idx / F, j = idx - r * F;
    if (!m_valid[r]) continue;
    if (a.state) {
      a.state[(int64_t)(row0 + r) * a.ld_state + FE + j] = rat[m_base[r] + j];
      a.next_state[(int64_t)(row0 + r) * a.ld_next + FE + j] = rat[m_base[r] + 1 + j];
    }
    if (a.state_h) {
      a.state_h[(int64_t)(row0 + r) * a.ld_h + FE + j] = f2bf(rat[m_base[r] + j]);
      a.next_h[(int64_t)(row0 + r) * a.ld_h + FE + j] = f2bf(rat[m_base[r] + 1 + j]);
    }
  }
  if (tid < R && m_valid[tid]) {
    a.reward[row0 + tid] = rat[m_base[tid] + F];
    a.done[row0 + tid] = m_done[tid] ? 1.f : 0.f;
  }
}

