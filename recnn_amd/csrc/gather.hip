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
// row share all of them, so each distinct (row, slot) embedding line is fetched ONCE (16-byte coalesced
// loads, (R+F)..R*(F+1) lines of E floats per workgroup) by the threads that own it and stored straight
// to every output that uses it with the widest store the output alignment allows (gather_dev.h).
#include "gather_dev.h"

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

// ------------------------------------------------------------------ per-epoch plan table
// One workgroup per batch of the epoch permutation: prefix sums of the batch users' window counts (256 users per pass),
// then every batch row looks itself up once: plan[batch][row] = (CSR offset of the window start << 1) | last-window flag,
// -1 past the batch's last row.  The gather of a step then reaches a row's window with ONE load (GatherArgs.plan).
__global__ __launch_bounds__(256) void frame_plan_rows_kernel(const int64_t* __restrict__ user_off, const int32_t* __restrict__ perm, int n,
                                                              int frame, int rows, int64_t* __restrict__ plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char plan_smem[];
  long long* s_start = (long long*)plan_smem;      // [n]
  int* s_off = (int*)(s_start + n);                // [n + 1]
  int* sc = s_off + n + 1;                         // [256]
  __shared__ int carry;
  const int tid = threadIdx.x;
  const int32_t* users = perm + (int64_t)blockIdx.x * n;
  if (tid == 0) { carry = 0; s_off[0] = 0; }
  __syncthreads();
  for (int base = 0; base < n; base += 256) {
    const int i = base + tid;
    int v = 0;
    if (i < n) {
      const int u = users[i];
      const long long o0 = user_off[u];
      v = max((int)(user_off[u + 1] - o0) - frame, 0);
      s_start[i] = o0;
    }
    sc[tid] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // inclusive Hillis-Steele scan
      const int t = tid >= o ? sc[tid - o] : 0;
      __syncthreads();
      sc[tid] += t;
      __syncthreads();
    }
    if (i < n) s_off[i + 1] = carry + sc[tid];
    __syncthreads();
    if (tid == 0) carry += sc[255];
    __syncthreads();
  }
  const int total = s_off[n];
  int64_t* out = plan + (int64_t)blockIdx.x * rows;
  for (int r = tid; r < rows; r += 256) {
    long long p = -1;
    if (r < total) {
      int lo = 0, hi = n;     // largest i with s_off[i] <= r
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= r) lo = mid; else hi = mid;
      }
      const int t = r - s_off[lo];
      p = ((s_start[lo] + t) << 1) | (t == s_off[lo + 1] - s_off[lo] - 1 ? 1 : 0);
    }
    out[r] = p;
  }
}

extern "C" int recnn_frame_plan_rows(const int64_t* user_off, const int32_t* perm, int users_per_batch, int n_batches, int frame, int rows,
                                     int64_t* plan, void* stream) {
  RECNN_REQUIRE(user_off && perm && plan && users_per_batch > 0 && n_batches > 0 && frame > 0 && rows > 0, "frame_plan_rows: bad arguments");
  const size_t lds = (size_t)users_per_batch * 8 + (size_t)(users_per_batch + 1 + 256) * 4;
  RECNN_REQUIRE(lds <= 64 * 1024, "frame_plan_rows: at most ~5000 users per batch (%d given)", users_per_batch);
  hipLaunchKernelGGL(frame_plan_rows_kernel, dim3(n_batches), dim3(256), lds, (hipStream_t)stream, user_off, perm, users_per_batch, frame, rows,
                     plan);
  return recnn_check_hip(hipGetLastError(), "frame_plan_rows");
}

// ------------------------------------------------------------------ dense epoch plan: every window of every user, once
// The reference trains on ALL L - F windows of every user of a batch (recnn/data/utils.py:161-187: rolling_window over each
// user's whole history, concatenated); a fixed-row batch that keeps only the first `rows` rows of its users drops the rest.
// Here the windows of the epoch's user SEQUENCE are concatenated in sequence order and cut into batches of `rows` rows: batch
// b = global rows [b rows, (b + 1) rows), a user's windows continue in the next batch, `done` marks each user's last window
// (utils.py:70-71) wherever it falls.  skip0 = windows of the first sequence entry already consumed by the previous epoch's last
// batch (the leftover of an epoch is carried, so that over consecutive epochs every (user, window) is visited exactly once per
// epoch).  Two launches: prefix sums of the window counts over the sequence, then one thread per plan row.
__global__ __launch_bounds__(1024) void frame_plan_seq_kernel(const int64_t* __restrict__ user_off, const int32_t* __restrict__ seq, int n,
                                                              int frame, int skip0, int32_t* __restrict__ row_off) {
  __shared__ int sc[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (tid == 0) { carry = 0; row_off[0] = 0; }
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    int v = 0;
    if (i < n) {
      const int u = seq[i];
      v = max((int)(user_off[u + 1] - user_off[u]) - frame - (i == 0 ? skip0 : 0), 0);
    }
    sc[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive Hillis-Steele scan
      const int t = tid >= o ? sc[tid - o] : 0;
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
__global__ __launch_bounds__(256) void frame_plan_dense_kernel(const int64_t* __restrict__ user_off, const int32_t* __restrict__ seq,
                                                               const int32_t* __restrict__ row_off, int n, int frame, int skip0, int rows,
                                                               int64_t n_rows, int64_t* __restrict__ plan) {
  const int total = row_off[n];
  const int64_t usable = (int64_t)(total / rows) * rows;      // whole batches of this epoch
  for (int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x; g0 < n_rows; g0 += (int64_t)gridDim.x * 256) {
    long long p = -1;
    // plan slots past the epoch's last whole batch repeat its first batches: a caller that lets the device cursor run past the
    // epoch (raw recnn_engine_graph_run without the host's epoch logic) reads valid rows, not garbage
    const int64_t g = (g0 < usable || usable == 0) ? g0 : (g0 - usable) % usable;
    if (g < total) {
      int lo = 0, hi = n;               // largest i with row_off[i] <= g   (row_off non-decreasing, row_off[n] > g)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (row_off[mid] <= g) lo = mid; else hi = mid;
      }
      const int u = seq[lo];
      const long long o0 = user_off[u];
      const int t = (int)(g - row_off[lo]) + (lo == 0 ? skip0 : 0);
      const int last = (int)(user_off[u + 1] - o0) - frame - 1;
      p = ((o0 + t) << 1) | (t == last ? 1 : 0);
    }
    plan[g0] = p;
  }
}

extern "C" int recnn_frame_plan_dense(const int64_t* user_off, const int32_t* seq, int n_seq, int skip0, int frame, int rows,
                                      int32_t* row_off, int64_t n_rows, int64_t* plan, void* stream) {
  RECNN_REQUIRE(user_off && seq && row_off && plan && n_seq > 0 && skip0 >= 0 && frame > 0 && rows > 0 && n_rows > 0, "frame_plan_dense: bad arguments");
  hipLaunchKernelGGL(frame_plan_seq_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, user_off, seq, n_seq, frame, skip0, row_off);
  int grid = (int)((n_rows + 255) / 256);
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(frame_plan_dense_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, user_off, seq, row_off, n_seq, frame, skip0, rows,
                     n_rows, plan);
  return recnn_check_hip(hipGetLastError(), "frame_plan_dense");
}

// ------------------------------------------------------------------ gather
template <int R, int W>
__global__ __launch_bounds__(256) void frame_gather_kernel(const GatherArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  frame_gather_body<R, W>(a, blockIdx.x, smem_raw);
}

// n consecutive batches of the epoch plan in ONE launch (blockIdx.y = batch j: cursor look-ahead + j, outputs j * rows rows
// further down): the batches of a whole policy cycle for the cycle-batched frozen-network forwards (engine.hip).  bf16 rows only.
__global__ __launch_bounds__(256) void frame_gather_multi_kernel(const GatherArgs a0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  GatherArgs a = a0;
  const int j = blockIdx.y;
  const int64_t r0 = (int64_t)j * a.rows;
  a.cursor_add += j;
  a.state_h += r0 * a.ld_h; a.next_h += r0 * a.ld_h; a.action_h += r0 * a.ld_h;
  a.reward += r0; a.done += r0;
  frame_gather_body<4, 4>(a, blockIdx.x, smem_raw);
}

int frame_gather_multi_launch(const GatherArgs& a, int n_sets, hipStream_t s) {
  const size_t lds = frame_gather_lds_bytes(a, 4);
  if (lds > 48 * 1024 || !a.state_h || a.state || (a.emb % 4) || a.rows <= 0 || n_sets <= 0 || n_sets > 65535) {
    recnn_set_error("frame_gather_multi: needs the bf16-only gather with a tile that fits 48 KB of LDS");
    return RECNN_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL(frame_gather_multi_kernel, dim3((a.rows + 3) / 4, n_sets), dim3(256), lds, s, a);
  return recnn_check_hip(hipGetLastError(), "frame_gather_multi");
}

size_t frame_gather_lds_bytes(const GatherArgs& a, int R) {
  const int F1 = a.frame + 1;
  (void)F1;
  size_t lds = (size_t)R * 8 + (size_t)R * 4 + 16;      // per-row window offsets and flags (no line staging since round 2)
  if (a.inline_plan && !a.plan) lds += (size_t)a.n_users * 8 + (size_t)(a.n_users + 1 + 4) * 4;
  return lds;
}

template <int R> static int launch_gather(const GatherArgs& a, int W, hipStream_t s) {
  const size_t lds = frame_gather_lds_bytes(a, R);
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
  // (4 rows per workgroup: 2 and 8 were measured in round 2 -- 7.0 / 9.6 us against 7.2 -- and are gone)
  return launch_gather<4>(a, W, stream);
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
