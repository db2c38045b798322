// gather.hip -- replay sampler + embedding gather (gfx950).
//
// Replaces, bit-exactly, the collate of the reference:
//   recnn/data/utils.py:7-10     rolling_window        (sliding windows of length F+1 per user)
//   recnn/data/utils.py:161-187  prepare_batch_static_size (concatenate windows over users)
//   recnn/data/utils.py:51-81    batch_tensor_embeddings   (emb[items], view, cat, done scatter)
// The [B, F+1] item/rating windows are never materialised: a batch row r is located by a binary
// search over the per-user row prefix sum, and the window is read straight from the CSR store.
//
// HBM-bound copy kernel.  One workgroup builds R consecutive batch rows.  Consecutive rows of one
// user share F of their F+1 embedding rows (sliding window) and state / next_state / action of one
// row share all of them, so each distinct (row, slot) embedding line is fetched ONCE into LDS
// ((R+F)..R*(F+1) lines of E floats) with 16-byte coalesced loads and then streamed out to the
// three outputs with the widest store the output alignment allows.
#include "gather.h"

// ------------------------------------------------------------------ plan: row prefix sums
__global__ __launch_bounds__(1024) void frame_plan_kernel(const int64_t* __restrict__ user_off,
                                                          const int32_t* __restrict__ users, int n, int frame,
                                                          int32_t* __restrict__ row_off, const int32_t* __restrict__ cursor,
                                                          int cursor_stride) {
  __shared__ int sc[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (cursor) users += (int64_t)(*cursor) * cursor_stride;  // device-side batch cursor (graph replay)
  if (tid == 0) { carry = 0; row_off[0] = 0; }
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + tid;
    int v = 0;
    if (i < n) {
      int u = users[i];
      int len = (int)(user_off[u + 1] - user_off[u]);
      v = max(len - frame, 0);
    }
    sc[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive Hillis-Steele scan
      int t = tid >= o ? sc[tid - o] : 0;
      __syncthreads();
      sc[tid] += t;
      __syncthreads();
    }
    if (i < n) row_off[i + 1] = carry + sc[tid];
    __syncthreads();
    if (tid == 0) carry += sc[1023];
    __syncthreads();
  }
}

extern "C" int recnn_frame_plan(const int64_t* user_off, const int32_t* batch_users, int n_users, int frame,
                                int32_t* row_off, const int32_t* cursor, int cursor_stride, void* stream) {
  RECNN_REQUIRE(user_off && batch_users && row_off && n_users >= 0 && frame > 0, "frame_plan: bad arguments");
  hipLaunchKernelGGL(frame_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, user_off, batch_users, n_users, frame,
                     row_off, cursor, cursor_stride);
  return recnn_check_hip(hipGetLastError(), "frame_plan");
}

// ------------------------------------------------------------------ gather
template <int W> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<1> { using type = float; };

// R rows per workgroup; W = floats per output store (4/2/1 by output alignment).
template <int R, int W>
__global__ __launch_bounds__(256) void frame_gather_kernel(const GatherArgs a) {
  using V = typename VecT<W>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
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
  const int row0 = blockIdx.x * R;
  const int32_t* users = a.users;
  if (a.cursor) users += (int64_t)(*a.cursor) * a.cursor_stride;

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

template <int R> static int launch_gather(const GatherArgs& a, int W, hipStream_t s) {
  const int F1 = a.frame + 1;
  size_t lds = (size_t)R * F1 * a.emb * 4 + (size_t)R * F1 * 4 + 4 * R * 4 + R * 8 + 16;
  if (a.inline_plan) lds += (size_t)a.n_users * 8 + (size_t)(a.n_users + 1 + 4) * 4;
  if (lds > 160 * 1024) { recnn_set_error("frame_gather: tile does not fit LDS (%zu bytes)", lds); return RECNN_E_UNSUPPORTED; }
  dim3 grid((a.rows + R - 1) / R), block(256);
  hipError_t e = hipSuccess;
  if (W == 4) {
    if (lds > 48 * 1024) e = hipFuncSetAttribute((const void*)frame_gather_kernel<R, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL((frame_gather_kernel<R, 4>), grid, block, lds, s, a);
  } else if (W == 2) {
    if (lds > 48 * 1024) e = hipFuncSetAttribute((const void*)frame_gather_kernel<R, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL((frame_gather_kernel<R, 2>), grid, block, lds, s, a);
  } else {
    if (lds > 48 * 1024) e = hipFuncSetAttribute((const void*)frame_gather_kernel<R, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL((frame_gather_kernel<R, 1>), grid, block, lds, s, a);
  }
  if (e != hipSuccess) return recnn_check_hip(e, "frame_gather attr");
  return recnn_check_hip(hipGetLastError(), "frame_gather");
}

static int g_gather_rows_per_wg = 4;
extern "C" void recnn_tune_gather_rows(int r) { g_gather_rows_per_wg = r; }

extern "C" int recnn_frame_gather(const int32_t* items, const float* ratings, const int64_t* user_off,
                                  const int32_t* batch_users, const int32_t* row_off, int n_users, int rows, int frame,
                                  int emb_dim, const float* table, float* state, int64_t ld_state, float* next_state,
                                  int64_t ld_next, float* action, int64_t ld_action, float* reward, float* done,
                                  const int32_t* cursor, int cursor_stride, void* stream) {
  RECNN_REQUIRE(items && ratings && user_off && batch_users && table, "frame_gather: null input");
  RECNN_REQUIRE(row_off || n_users <= 1024, "frame_gather: more than 1024 users per batch need a recnn_frame_plan row_off");
  RECNN_REQUIRE(state && next_state && action && reward && done, "frame_gather: null output");
  RECNN_REQUIRE(frame > 0 && emb_dim > 0 && (emb_dim % 4) == 0, "frame_gather: emb_dim must be a positive multiple of 4");
  RECNN_REQUIRE(((uintptr_t)table & 15) == 0, "frame_gather: table must be 16-byte aligned");
  RECNN_REQUIRE(n_users >= 0 && rows >= 0, "frame_gather: negative sizes");
  if (rows == 0) return 0;
  RECNN_REQUIRE(n_users > 0, "frame_gather: rows requested from an empty user list");
  GatherArgs a;
  memset(&a, 0, sizeof(a));
  a.items = items; a.ratings = ratings; a.user_off = user_off; a.users = batch_users; a.row_off = row_off;
  a.n_users = n_users; a.rows = rows; a.frame = frame; a.emb = emb_dim; a.table = table;
  a.state = state; a.ld_state = ld_state; a.next_state = next_state; a.ld_next = ld_next;
  a.action = action; a.ld_action = ld_action; a.reward = reward; a.done = done;
  a.cursor = cursor; a.cursor_stride = cursor_stride;
  a.inline_plan = row_off == nullptr;
  return frame_gather_launch(a, (hipStream_t)stream);
}

int frame_gather_launch(GatherArgs a, hipStream_t stream) {
  float* state = a.state; float* next_state = a.next_state; float* action = a.action;
  const int64_t ld_state = a.ld_state, ld_next = a.ld_next, ld_action = a.ld_action;
  const int emb_dim = a.emb;
  // widest store every output row start supports
  auto al = [](const void* p, int64_t ld) {
    uintptr_t x = (uintptr_t)p | (uintptr_t)(ld * 4);
    return (x & 15) == 0 ? 4 : ((x & 7) == 0 ? 2 : 1);
  };
  int W = 4;
  if (state) {  // fp32 rows may be omitted (bf16 twins only) by the engine's own sampler
    W = al(state, ld_state);
    int w2 = al(next_state, ld_next); if (w2 < W) W = w2;
    int w3 = al(action, ld_action); if (w3 < W) W = w3;
  } else if (!a.state_h) {
    recnn_set_error("frame_gather: no output rows");
    return RECNN_E_INVALID;
  }
  if (emb_dim % W) W = 1;
  if (a.state_h && W != 4) { recnn_set_error("frame_gather: bf16 twin rows need 16-byte aligned fp32 rows"); return RECNN_E_INVALID; }
  switch (g_gather_rows_per_wg) {
    case 2: return launch_gather<2>(a, W, stream);
    case 8: return launch_gather<8>(a, W, stream);
    default: return launch_gather<4>(a, W, stream);
  }
}

// ------------------------------------------------------------------ pack a canonical batch
__global__ __launch_bounds__(256) void pack_batch_kernel(const float* __restrict__ state, int64_t ld_state,
                                                         const float* __restrict__ action, int64_t ld_action,
                                                         const float* __restrict__ next_state, int64_t ld_next, int rows, int S,
                                                         int A, float* __restrict__ xs, float* __restrict__ xn, int64_t ld_x) {
  const int r = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;  // column in the packed row
  if (r >= rows || c >= A + S) return;
  if (c < A) {
    xs[(int64_t)r * ld_x + c] = action[(int64_t)r * ld_action + c];
  } else {
    xs[(int64_t)r * ld_x + c] = state[(int64_t)r * ld_state + (c - A)];
    xn[(int64_t)r * ld_x + c] = next_state[(int64_t)r * ld_next + (c - A)];
  }
}

extern "C" int recnn_pack_batch(const float* state, int64_t ld_state, const float* action, int64_t ld_action,
                                const float* next_state, int64_t ld_next, int rows, int state_dim, int action_dim, float* xs,
                                float* xn, int64_t ld_x, void* stream) {
  RECNN_REQUIRE(state && action && next_state && xs && xn, "pack_batch: null pointer");
  RECNN_REQUIRE(rows >= 0 && state_dim > 0 && action_dim > 0 && ld_x >= state_dim + action_dim, "pack_batch: bad sizes");
  if (rows == 0) return 0;
  dim3 grid((state_dim + action_dim + 255) / 256, rows), block(256);
  hipLaunchKernelGGL(pack_batch_kernel, grid, block, 0, (hipStream_t)stream, state, ld_state, action, ld_action, next_state,
                     ld_next, rows, state_dim, action_dim, xs, xn, ld_x);
  return recnn_check_hip(hipGetLastError(), "pack_batch");
}

// ------------------------------------------------------------------ packed fp32 rows -> bf16 twins
// (engine, bf16 mode, batch not produced by the engine's own sampler): one pass over xs and xn.
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ xs, const float* __restrict__ xn,
                                                           bf16_t* __restrict__ hs, bf16_t* __restrict__ hn, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 a = ((const float4*)xs)[i], b = ((const float4*)xn)[i];
    ((uint2*)hs)[i] = make_uint2(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w));
    ((uint2*)hn)[i] = make_uint2(pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
  }
}
int rows_to_bf16_launch(const float* xs, const float* xn, bf16_t* hs, bf16_t* hn, int rows, int64_t ld, hipStream_t s) {
  const int64_t n4 = (int64_t)rows * ld / 4;
  if (n4 <= 0) return 0;
  int grid = (int)((n4 + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(rows_to_bf16_kernel, dim3(grid), dim3(256), 0, s, xs, xn, hs, hn, n4);
  return recnn_check_hip(hipGetLastError(), "rows_to_bf16_kernel");
}
