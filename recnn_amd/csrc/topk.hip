// topk.hip -- batched exact top-K scoring of action vectors against the item-embedding table (gfx950).
//
// SURVEY.md 8(f2) "next": replaces the three external retrieval paths the reference uses to turn a generated action
// into recommended items -- faiss IndexFlatL2 / IndexFlatIP / IP-on-normalised rows (examples/streamlit_demo.py:190-204),
// the per-item scipy distance loop (examples/streamlit_demo.py:207-231, `rank`) and the Milvus service
// (recnn/data/db_con.py:45-56, `MilvusConnection.search`) -- by one exact-fp32 MFMA scoring GEMM fused with a
// per-query top-K selection.
//
//   score(q, t):  IP  = q.t          (larger is better)
//                 L2  = |q - t|^2    (smaller is better; squared distance, as faiss IndexFlatL2 reports)
//                 COS = q.t / |t|    (larger is better; the demo normalises the table rows, not the query)
//
// Grid = (ceil(B/64) query tiles) x (S item splits).  A workgroup keeps its 64 query rows in LDS, streams its share
// of the table in 64-item chunks (register-staged, fp32), multiplies with v_mfma_f32_16x16x4_f32 (exact fp32: ranking
// by bf16 scores would reorder near ties), parks the 64x64 score tile in LDS and lets each wave maintain the sorted
// top-K lists of 16 query rows: a chunk's scores are compared against the row's current K-th best with one ballot per
// 64 items, and only the (rare) survivors are inserted.  Ties are broken towards the smaller item id.  Partial lists
// [B][S][K] are merged by a second tiny kernel.
#include "common.h"

namespace {
constexpr int QT = 64;     // query rows per workgroup
constexpr int IT = 64;     // items per chunk
constexpr int KMAX = 64;   // largest supported K
constexpr int PITCH = 132; // floats per LDS row of a [rows][128] tile (+4 pad: conflict-light b128 reads)
enum { M_IP = 0, M_L2 = 1, M_COS = 2 };

struct TopkArgs {
  const float* q; int64_t ldq; int B;
  const float* table; int N, E;
  const float* aux;        // L2: |t|^2 per item;  COS: 1/|t| per item;  IP: unused
  int metric, K, splits;
  float* part_score;       // [B][S][K]  internal key (larger = better)
  int32_t* part_id;        // [B][S][K]
};

__device__ inline bool better(float s, int id, float s2, int id2) { return s > s2 || (s == s2 && id < id2); }

template <int E_>
__global__ __launch_bounds__(256) void topk_scores_kernel(const TopkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Qs = (float*)smem;                  // [QT][PITCH]
  float* Ts = Qs + QT * PITCH;               // [IT][PITCH]
  float* Ss = Ts + IT * PITCH;               // [QT][IT + 4]
  float* Ls = Ss + QT * (IT + 4);            // [QT][KMAX] keys
  int* Li = (int*)(Ls + QT * KMAX);          // [QT][KMAX] ids
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * QT;
  const int per = (a.N + a.splits - 1) / a.splits;
  const int n_begin = blockIdx.y * per, n_end = min(a.N, n_begin + per);
  const int K = a.K;

  for (int i = tid; i < QT * KMAX; i += 256) { Ls[i] = -INFINITY; Li[i] = 0x7FFFFFFF; }
  // query tile -> LDS (rows past B are zero)
  for (int c = tid; c < QT * (E_ / 4); c += 256) {
    const int r = c / (E_ / 4), k4 = c % (E_ / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + r < a.B) v = *(const float4*)(a.q + (int64_t)(q0 + r) * a.ldq + k4 * 4);
    *(float4*)&Qs[r * PITCH + k4 * 4] = v;
  }
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;   // wave tile: 32 queries x 32 items
  const int fr = lane & 15, fg = lane >> 4;
  __syncthreads();

  for (int n0 = n_begin; n0 < n_end; n0 += IT) {
    // ---- table chunk -> LDS
    for (int c = tid; c < IT * (E_ / 4); c += 256) {
      const int r = c / (E_ / 4), k4 = c % (E_ / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + r < n_end) v = *(const float4*)(a.table + (int64_t)(n0 + r) * E_ + k4 * 4);
      *(float4*)&Ts[r * PITCH + k4 * 4] = v;
    }
    __syncthreads();
    // ---- scores = Q T^T (exact fp32 MFMA)
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < E_ / 16; ++ks) {
      float4 qa[2], tb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) qa[i] = *(const float4*)&Qs[(wm0 + i * 16 + fr) * PITCH + ks * 16 + fg * 4];
#pragma unroll
      for (int j = 0; j < 2; ++j) tb[j] = *(const float4*)&Ts[(wn0 + j * 16 + fr) * PITCH + ks * 16 + fg * 4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&qa[i])[e], ((const float*)&tb[j])[e], acc[i][j], 0, 0, 0);
    }
    // ---- ranking key (larger = better) into the score tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn0 + j * 16 + fr;
        const int n = n0 + col;
        float ax = 0.f;
        if (a.metric != M_IP && n < n_end) ax = a.aux[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm0 + i * 16 + fg * 4 + r;
          float s = acc[i][j][r];
          if (a.metric == M_L2) s = 2.f * s - ax;       // |q|^2 is constant per row: it does not change the order
          else if (a.metric == M_COS) s = s * ax;
          if (n >= n_end) s = -INFINITY;
          Ss[row * (IT + 4) + col] = s;
        }
      }
    __syncthreads();
    // ---- selection: wave w owns query rows 16w .. 16w+15; lane = item of the chunk
    for (int rr = 0; rr < 16; ++rr) {
      const int row = wave * 16 + rr;
      if (q0 + row >= a.B) break;
      float* ls = Ls + row * KMAX;
      int* li = Li + row * KMAX;
      const float s = Ss[row * (IT + 4) + lane];
      const int id = n0 + lane;
      float thr = ls[K - 1];
      int thr_id = li[K - 1];
      unsigned long long m = __ballot(id < n_end && better(s, id, thr, thr_id));
      while (m) {                                       // rare after the first chunks
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float cs = __shfl(s, src, 64);
        const int cid = n0 + src;
        if (!better(cs, cid, ls[K - 1], li[K - 1])) continue;
        // insert (cs, cid) into the sorted list: lanes shift the tail in parallel
        const float mine = lane < K ? ls[lane] : 0.f;
        const int mine_id = lane < K ? li[lane] : 0;
        const bool before = lane < K && better(mine, mine_id, cs, cid);       // entries that stay in front
        const int pos = __popcll(__ballot(before));                            // insertion position
        const float up = __shfl_up(mine, 1, 64);                               // (all lanes: no divergent shuffles)
        const int up_id = __shfl_up(mine_id, 1, 64);
        if (lane < K) {
          if (lane == pos) { ls[lane] = cs; li[lane] = cid; }
          else if (lane > pos) { ls[lane] = up; li[lane] = up_id; }
        }
      }
    }
    __syncthreads();
  }
  // ---- partial lists out
  for (int i = tid; i < QT * K; i += 256) {
    const int row = i / K, j = i % K;
    if (q0 + row < a.B) {
      const int64_t o = ((int64_t)(q0 + row) * a.splits + blockIdx.y) * K + j;
      a.part_score[o] = Ls[row * KMAX + j];
      a.part_id[o] = Li[row * KMAX + j];
    }
  }
}

// merge S sorted partial lists per query; one wave per query row
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ ps, const int32_t* __restrict__ pi, int B, int S, int K,
                                                         int metric, const float* __restrict__ q, int64_t ldq, int E,
                                                         float* __restrict__ out_d, int64_t* __restrict__ out_i) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= B) return;
  const int total = S * K;
  // each lane caches up to 8 candidates (S*K <= 512)
  float cs[8]; int ci[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = lane + j * 64;
    cs[j] = c < total ? ps[(int64_t)row * total + c] : -INFINITY;
    ci[j] = c < total ? pi[(int64_t)row * total + c] : 0x7FFFFFFF;
  }
  float qn = 0.f;
  if (metric == M_L2) {
    for (int k = lane; k < E; k += 64) { const float v = q[(int64_t)row * ldq + k]; qn += v * v; }
    qn = wave_sum(qn);
  }
  for (int k = 0; k < K; ++k) {
    float bs = -INFINITY; int bi = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (better(cs[j], ci[j], bs, bi)) { bs = cs[j]; bi = ci[j]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (better(os, oi, bs, bi)) { bs = os; bi = oi; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (ci[j] == bi) { cs[j] = -INFINITY; ci[j] = 0x7FFFFFFF; }   // ids are unique across the splits
    if (lane == 0) {
      out_d[(int64_t)row * K + k] = metric == M_L2 ? fmaxf(qn - bs, 0.f) : bs;  // L2: |q|^2 - (2 q.t - |t|^2)
      out_i[(int64_t)row * K + k] = bi == 0x7FFFFFFF ? -1 : bi;
    }
  }
}

// per-item auxiliary term of the metric: L2 -> |t|^2, COS -> 1/|t|
__global__ __launch_bounds__(256) void topk_aux_kernel(const float* __restrict__ table, int N, int E, int metric, float* __restrict__ aux) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  float s = 0.f;
  for (int k = lane; k < E; k += 64) { const float v = table[(int64_t)n * E + k]; s += v * v; }
  s = wave_sum(s);
  if (lane == 0) aux[n] = metric == M_L2 ? s : (s > 0.f ? 1.0f / sqrtf(s) : 0.f);
}
}  // namespace

extern "C" int recnn_topk_workspace_bytes(int n_queries, int k, int64_t* h_bytes) {
  RECNN_REQUIRE(h_bytes && n_queries >= 0 && k > 0 && k <= KMAX, "topk_workspace_bytes: bad arguments (k <= 64)");
  *h_bytes = (int64_t)n_queries * 8 * k * 8;   // 8 splits x (score + id)
  return 0;
}

extern "C" int recnn_topk_item_aux(const float* table, int n_items, int emb_dim, int metric, float* aux, void* stream) {
  RECNN_REQUIRE(table && aux && n_items > 0 && emb_dim > 0 && (metric == M_L2 || metric == M_COS), "topk_item_aux: bad arguments");
  hipLaunchKernelGGL(topk_aux_kernel, dim3((n_items + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, n_items, emb_dim, metric, aux);
  return recnn_check_hip(hipGetLastError(), "topk_aux_kernel");
}

extern "C" int recnn_topk_search(const float* queries, int64_t ld_q, int n_queries, const float* table, int n_items, int emb_dim,
                                 int metric, const float* item_aux, int k, float* out_dist, int64_t* out_ids, void* workspace,
                                 void* stream) {
  RECNN_REQUIRE(queries && table && out_dist && out_ids && workspace, "topk_search: null pointer");
  RECNN_REQUIRE(n_queries >= 0 && n_items > 0 && k > 0 && k <= KMAX && k <= n_items, "topk_search: need 0 < k <= min(64, n_items)");
  RECNN_REQUIRE(emb_dim == 128, "topk_search: emb_dim must be 128 (the reference's embedding width)");
  RECNN_REQUIRE(metric == M_IP || ((metric == M_L2 || metric == M_COS) && item_aux), "topk_search: L2 / COS need the item aux array");
  RECNN_REQUIRE((((uintptr_t)queries | (uintptr_t)table) & 15) == 0 && (ld_q % 4) == 0, "topk_search: 16-byte alignment");
  if (n_queries == 0) return 0;
  TopkArgs a;
  a.q = queries; a.ldq = ld_q; a.B = n_queries; a.table = table; a.N = n_items; a.E = emb_dim; a.aux = item_aux;
  a.metric = metric; a.K = k;
  const int tiles = (n_queries + QT - 1) / QT;
  int splits = 512 / tiles;               // enough workgroups to fill 256 CUs twice
  if (splits < 1) splits = 1;
  if (splits > 8) splits = 8;
  while (splits > 1 && (n_items + splits - 1) / splits < 4 * IT) --splits;
  a.splits = splits;
  a.part_score = (float*)workspace;
  a.part_id = (int32_t*)((char*)workspace + (int64_t)n_queries * 8 * k * 4);
  const size_t lds = (size_t)(QT * PITCH + IT * PITCH + QT * (IT + 4) + QT * KMAX) * 4 + (size_t)QT * KMAX * 4;
  static bool attr = false;
  if (!attr) {
    RECNN_HIP(hipFuncSetAttribute((const void*)topk_scores_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  hipLaunchKernelGGL(topk_scores_kernel<128>, dim3(tiles, splits), dim3(256), lds, (hipStream_t)stream, a);
  hipLaunchKernelGGL(topk_merge_kernel, dim3((n_queries + 3) / 4), dim3(256), 0, (hipStream_t)stream, a.part_score, a.part_id, n_queries,
                     splits, k, metric, queries, ld_q, emb_dim, out_dist, out_ids);
  return recnn_check_hip(hipGetLastError(), "topk_search");
}
