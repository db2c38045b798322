// csr.hip -- device-side replay-store builder: ratings rows -> CSR ordered by (user, time)   
//
// Replaces recnn/data/dataset_functions.py:84-126 (`prepare_dataset`: per-row Python lambdas for the rating transform and
// the movieId -> dense id map, `sort_values(by="timestamp")`, a Python callback per user group) by one pass on the GPU:
//   1. composite key  (user - user_min) << ts_bits | (timestamp - ts_min)  and the row index as payload
//   2. LSD radix sort of (key, index), 8-bit digits, only over the bits the data uses (ML20M: 18 + 30 bits = 6 passes);
//      every pass is stable, so rows with equal (user, timestamp) keep their input order
//   3. gather through the sorted index: dense item id (binary search in the sorted key table), rating 2 (r - 2.5) in
//      fp64, and a first-row-of-user flag; a scan of the flags compacts users / user_off.
// Tie order: pandas' sort_values is numpy's default argsort, which is NOT stable (and on AVX-512 hosts is a different
// algorithm again), so the reference's order of rows with equal (user, timestamp) is host dependent; this builder defines
// it as input order.  Wherever (user, timestamp) pairs are unique the output equals the reference's bit for bit.
//
// All of it is HBM-bound integer work: a radix pass reads and writes 12 B per row twice (histogram + scatter), so
// 6 passes x 20 M rows move ~3 GB -- well under a millisecond of bandwidth at 8 TB/s; the kernels below are simple
// (one wave per 2048-row tile in the scatter, ballot-based stable ranking, tile sorted in LDS before it is written out) and
// finish the sort in ~2 ms, the whole build in ~4 ms.
#include "common.h"

namespace {

constexpr int RB = 8, NB = 1 << RB;    // digit bits, buckets
constexpr int TILE = 2048;             // rows per workgroup in the histogram / scatter kernels
constexpr int ST = 256, SI = 8, SCAN_TILE = ST * SI;   // scan: threads, items per thread

struct MinMax { long long umin, umax, tmin, tmax; };

__global__ __launch_bounds__(256) void minmax_kernel(const int64_t* __restrict__ users, const int64_t* __restrict__ ts, int64_t n,
                                                     MinMax* __restrict__ parts) {
  __shared__ long long red[4][256];
  long long a = INT64_MAX, b = INT64_MIN, c = INT64_MAX, d = INT64_MIN;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const long long u = users[i], t = ts[i];
    a = u < a ? u : a; b = u > b ? u : b; c = t < c ? t : c; d = t > d ? t : d;
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c; red[3][threadIdx.x] = d;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const int j = threadIdx.x + s;
      if (red[0][j] < red[0][threadIdx.x]) red[0][threadIdx.x] = red[0][j];
      if (red[1][j] > red[1][threadIdx.x]) red[1][threadIdx.x] = red[1][j];
      if (red[2][j] < red[2][threadIdx.x]) red[2][threadIdx.x] = red[2][j];
      if (red[3][j] > red[3][threadIdx.x]) red[3][threadIdx.x] = red[3][j];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = MinMax{red[0][0], red[1][0], red[2][0], red[3][0]};
}

__global__ __launch_bounds__(256) void make_keys_kernel(const int64_t* __restrict__ users, const int64_t* __restrict__ ts, int64_t n,
                                                        long long umin, long long tmin, int tbits, uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ idx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    keys[i] = ((uint64_t)(users[i] - umin) << tbits) | (uint64_t)(ts[i] - tmin);
    idx[i] = (uint32_t)i;
  }
}

// hist[d * nblocks + b] = number of rows of tile b whose digit is d   (digit-major: one exclusive scan gives every
// (digit, tile) its global start)
__global__ __launch_bounds__(256) void radix_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ hist,
                                                         int nblocks) {
  __shared__ uint32_t h[NB];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE;
  for (int j = threadIdx.x; j < TILE; j += 256) {
    const int64_t i = base + j;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & (NB - 1)], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// One wave per tile, rows taken 64 at a time in order: a row's rank inside its digit = rows of this tile with the same
// digit seen so far + its rank among the equal-digit lanes below it (8 ballots) -- stable by construction.  The tile is
// first sorted INSIDE LDS (slot = local start of the digit + rank), then written out position by position: neighbouring LDS
// positions of one digit are neighbouring global slots, so the stores leave as whole runs (~16 rows = 128 B of keys per
// digit and tile) instead of 8-byte and 4-byte writes scattered over 256 runs (0.63 -> 0.31 ms per pass at 20 M rows).
// The running per-digit offsets are touched by this one wave only: LDS instructions of a wave execute in program order, so
// the lanes' reads of off[d] are done before the group leaders' updates -- no barrier inside the loop.  The next round's rows
// are loaded while the current round is ranked.
__global__ __launch_bounds__(WAVE) void radix_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, int64_t n,
                                                             int shift, const uint32_t* __restrict__ start, int nblocks,
                                                             uint64_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out) {
  __shared__ uint64_t sk[TILE];
  __shared__ uint32_t sv[TILE];
  __shared__ uint32_t off[NB];      // running local slot of each digit
  __shared__ uint32_t lstart[NB];   // local start of each digit inside the sorted tile
  __shared__ uint32_t gstart[NB];   // global start of (digit, this tile)
  const int lane = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * TILE;
  const int rows = (int)(n - base < TILE ? n - base : TILE);
  // counts of this tile = differences of the scanned digit-major histogram (entry (d, b) is followed by (d, b + 1), the
  // last tile of digit d by the first tile of digit d + 1, the very last entry by n)
  uint32_t cnt[NB / WAVE], run = 0;
#pragma unroll
  for (int j = 0; j < NB / WAVE; ++j) {
    const int d = lane * (NB / WAVE) + j;
    const int64_t e = (int64_t)d * nblocks + blockIdx.x;
    const uint32_t s0 = start[e];
    const uint32_t s1 = e + 1 < (int64_t)NB * nblocks ? start[e + 1] : (uint32_t)n;
    gstart[d] = s0;
    cnt[j] = s1 - s0;
    run += cnt[j];
  }
  uint32_t incl = run;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o, WAVE);
    if (lane >= o) incl += t;
  }
  uint32_t ex = incl - run;
#pragma unroll
  for (int j = 0; j < NB / WAVE; ++j) {
    const int d = lane * (NB / WAVE) + j;
    lstart[d] = ex;
    off[d] = ex;
    ex += cnt[j];
  }
  __syncthreads();
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int rounds = (rows + WAVE - 1) / WAVE;
  uint64_t k_next = 0;
  uint32_t v_next = 0;
  if (lane < rows) { k_next = keys[base + lane]; v_next = idx[base + lane]; }
  for (int r = 0; r < rounds; ++r) {
    const int p = r * WAVE + lane;
    const bool valid = p < rows;
    const uint64_t k = k_next;
    const uint32_t v = v_next;
    if (p + WAVE < rows) { k_next = keys[base + p + WAVE]; v_next = idx[base + p + WAVE]; }
    const uint32_t d = (uint32_t)(k >> shift) & (NB - 1);
    uint64_t same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      same &= bit ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(same & below);
    uint32_t slot = 0;
    if (valid) slot = off[d] + rank;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every lane holds its slot before a leader moves the offset on
    if (valid && rank == 0) off[d] += (uint32_t)__popcll(same);
    if (valid) {
      sk[slot] = k;
      sv[slot] = v;
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  // write-out: LDS position p of digit d -> global slot gstart[d] + (p - lstart[d])
  for (int p = lane; p < rows; p += WAVE) {
    const uint64_t k = sk[p];
    const uint32_t d = (uint32_t)(k >> shift) & (NB - 1);
    const uint32_t g = gstart[d] + ((uint32_t)p - lstart[d]);
    keys_out[g] = k;
    idx_out[g] = sv[p];
  }
}

// ---------------------------------------------------------------- exclusive scan of uint32 (three phases, recursive)
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t* sh, uint32_t* total) {
  // Hillis-Steele over ST threads
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = 1; s < ST; s <<= 1) {
    const uint32_t t = (int)threadIdx.x >= s ? sh[threadIdx.x - s] : 0u;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  const uint32_t incl = sh[threadIdx.x];
  *total = sh[ST - 1];
  __syncthreads();
  return incl - v;
}

__global__ __launch_bounds__(ST) void scan_sums_kernel(const uint32_t* __restrict__ x, int64_t m, uint32_t* __restrict__ sums) {
  __shared__ uint32_t sh[ST];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SI;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SI; ++j) s += base + j < m ? x[base + j] : 0u;
  uint32_t total;
  block_excl_scan(s, sh, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// in place: x[i] <- sum_{j<i} x[j]; offs (scanned block sums) may be NULL for a single block; total (optional) <- sum x
__global__ __launch_bounds__(ST) void scan_apply_kernel(uint32_t* __restrict__ x, int64_t m, const uint32_t* __restrict__ offs,
                                                        uint32_t* __restrict__ total_out) {
  __shared__ uint32_t sh[ST];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SI;
  uint32_t v[SI], s = 0;
#pragma unroll
  for (int j = 0; j < SI; ++j) {
    v[j] = base + j < m ? x[base + j] : 0u;
    s += v[j];
  }
  uint32_t total;
  uint32_t run = block_excl_scan(s, sh, &total) + (offs ? offs[blockIdx.x] : 0u);
#pragma unroll
  for (int j = 0; j < SI; ++j) {
    if (base + j < m) x[base + j] = run;
    run += v[j];
  }
  if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == ST - 1) *total_out = run;
}

// scratch: room for ceil(m / SCAN_TILE) + ceil(that / SCAN_TILE) + ... uint32
void scan_exclusive(uint32_t* x, int64_t m, uint32_t* scratch, uint32_t* total_out, hipStream_t s) {
  const int64_t nb = (m + SCAN_TILE - 1) / SCAN_TILE;
  if (nb <= 1) {
    hipLaunchKernelGGL(scan_apply_kernel, dim3(1), dim3(ST), 0, s, x, m, (const uint32_t*)nullptr, total_out);
    return;
  }
  hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)nb), dim3(ST), 0, s, x, m, scratch);
  scan_exclusive(scratch, nb, scratch + nb, nullptr, s);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(ST), 0, s, x, m, scratch, total_out);
}

int64_t scan_scratch_elems(int64_t m) {
  int64_t t = 0;
  while (m > SCAN_TILE) {
    m = (m + SCAN_TILE - 1) / SCAN_TILE;
    t += m;
  }
  return t + 1;
}

// ---------------------------------------------------------------- gather through the sorted order
// dense id of an item key: binary search in the ascending key table; -1 (and a count) when the key is absent
__device__ inline int64_t map_item(int64_t item, const int64_t* __restrict__ map_keys, const int64_t* __restrict__ map_vals, int n_map,
                                   uint32_t* missing) {
  if (!map_keys) return item;
  int lo = 0, hi = n_map;     // first position with map_keys[pos] >= item
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (map_keys[mid] < item) lo = mid + 1; else hi = mid;
  }
  if (lo < n_map && map_keys[lo] == item) return map_vals[lo];
  if (missing) atomicAdd(missing, 1u);
  return -1;
}

// the mapped id column in INPUT row order (what `df["movieId"].map(key_to_id)` leaves in the frame)
__global__ __launch_bounds__(256) void csr_map_rows_kernel(const int64_t* __restrict__ item_keys, int64_t n, const int64_t* __restrict__ map_keys,
                                                           const int64_t* __restrict__ map_vals, int n_map, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = map_item(item_keys[i], map_keys, map_vals, n_map, nullptr);
}

__global__ __launch_bounds__(256) void csr_gather_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ order, int64_t n,
                                                         int tbits, const int64_t* __restrict__ item_keys,
                                                         const double* __restrict__ ratings, const int64_t* __restrict__ map_keys,
                                                         const int64_t* __restrict__ map_vals, int n_map, int64_t* __restrict__ items_out,
                                                         double* __restrict__ ratings_out, uint32_t* __restrict__ flags,
                                                         uint32_t* __restrict__ missing) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t src = order[i];
    const int64_t item = map_item(item_keys[src], map_keys, map_vals, n_map, missing);
    items_out[i] = item;
    ratings_out[i] = 2.0 * (ratings[src] - 2.5);
    const uint64_t u = keys[i] >> tbits;
    flags[i] = (i == 0 || (keys[i - 1] >> tbits) != u) ? 1u : 0u;
  }
}

// pos = exclusive scan of the first-row flags: user number of every row
__global__ __launch_bounds__(256) void csr_users_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos, int64_t n, int tbits,
                                                        long long umin, int64_t* __restrict__ users_out, int64_t* __restrict__ user_off) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint64_t u = keys[i] >> tbits;
    const bool first = i == 0 || (keys[i - 1] >> tbits) != u;
    if (first) {
      users_out[pos[i]] = (int64_t)u + umin;
      user_off[pos[i]] = i;
    }
    if (i == n - 1) user_off[pos[i] + (first ? 1 : 0)] = n;     // one past the last user
  }
}

__global__ __launch_bounds__(256) void csr_widen_order_kernel(const uint32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}

inline int bits_for(uint64_t range) {  // bits needed to hold values 0..range
  int b = 0;
  while (range) { ++b; range >>= 1; }
  return b < 1 ? 1 : b;
}
inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct Layout {
  int64_t keys_a, keys_b, idx_a, idx_b, hist, scan, flags, mm, total;
  int nblocks;
};
Layout layout_for(int64_t n) {
  Layout L{};
  L.nblocks = (int)((n + TILE - 1) / TILE);
  int64_t o = 0;
  L.keys_a = o; o += align256(8 * n);
  L.keys_b = o; o += align256(8 * n);
  L.idx_a = o; o += align256(4 * n);
  L.idx_b = o; o += align256(4 * n);
  L.hist = o; o += align256(4 * (int64_t)NB * L.nblocks);
  const int64_t sm = scan_scratch_elems(n > (int64_t)NB * L.nblocks ? n : (int64_t)NB * L.nblocks);
  L.scan = o; o += align256(4 * sm);
  L.flags = o; o += align256(4 * n);
  L.mm = o; o += align256((int64_t)sizeof(MinMax) * 1024 + 16);
  L.total = o;
  return L;
}

}  // namespace

extern "C" {

int recnn_csr_workspace_bytes(int64_t n_rows, int64_t* bytes) {
  RECNN_REQUIRE(bytes && n_rows >= 0 && n_rows < ((int64_t)1 << 32) - TILE, "csr_workspace_bytes: bad arguments (rows must fit 32 bits)");
  *bytes = layout_for(n_rows > 0 ? n_rows : 1).total;
  return 0;
}

int recnn_csr_build(const int64_t* user_ids, const int64_t* item_keys, const double* ratings, const int64_t* timestamps, int64_t n_rows,
                    const int64_t* map_keys, const int64_t* map_vals, int n_map, int64_t* items_out, double* ratings_out,
                    int64_t* users_out, int64_t* user_off_out, int64_t* order_out, int64_t* mapped_rows_out, int64_t* host_counts,
                    void* workspace, int64_t workspace_bytes, void* stream) {
  RECNN_REQUIRE(user_ids && item_keys && ratings && timestamps && items_out && ratings_out && users_out && user_off_out && host_counts &&
                workspace, "csr_build: bad arguments");
  RECNN_REQUIRE(n_rows > 0 && n_rows < ((int64_t)1 << 32) - TILE, "csr_build: 1 <= rows < 2^32");
  RECNN_REQUIRE((map_keys == nullptr) == (map_vals == nullptr) && (!map_keys || n_map > 0), "csr_build: the id map needs keys and values");
  const Layout L = layout_for(n_rows);
  RECNN_REQUIRE(workspace_bytes >= L.total, "csr_build: workspace too small (recnn_csr_workspace_bytes)");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  uint64_t* keys[2] = {(uint64_t*)(ws + L.keys_a), (uint64_t*)(ws + L.keys_b)};
  uint32_t* idx[2] = {(uint32_t*)(ws + L.idx_a), (uint32_t*)(ws + L.idx_b)};
  uint32_t* hist = (uint32_t*)(ws + L.hist);
  uint32_t* scan = (uint32_t*)(ws + L.scan);
  uint32_t* flags = (uint32_t*)(ws + L.flags);
  MinMax* mm = (MinMax*)(ws + L.mm);
  uint32_t* counters = (uint32_t*)(ws + L.mm + sizeof(MinMax) * 1024);   // [0] users, [1] rows with an unmapped item key
  const int n = L.nblocks;
  const int grid = (int)((n_rows + 255) / 256 < 4096 ? (n_rows + 255) / 256 : 4096);

  // key ranges (one small round trip: the number of radix passes depends on them)
  const int mmb = grid < 1024 ? grid : 1024;
  hipLaunchKernelGGL(minmax_kernel, dim3(mmb), dim3(256), 0, s, user_ids, timestamps, n_rows, mm);
  MinMax host[1024];
  RECNN_HIP(hipMemcpyAsync(host, mm, sizeof(MinMax) * mmb, hipMemcpyDeviceToHost, s));
  RECNN_HIP(hipStreamSynchronize(s));
  MinMax r = host[0];
  for (int i = 1; i < mmb; ++i) {
    if (host[i].umin < r.umin) r.umin = host[i].umin;
    if (host[i].umax > r.umax) r.umax = host[i].umax;
    if (host[i].tmin < r.tmin) r.tmin = host[i].tmin;
    if (host[i].tmax > r.tmax) r.tmax = host[i].tmax;
  }
  const int tbits = bits_for((uint64_t)(r.tmax - r.tmin)), ubits = bits_for((uint64_t)(r.umax - r.umin));
  RECNN_REQUIRE(tbits + ubits <= 64, "csr_build: user id range (%d bits) + timestamp range (%d bits) exceed a 64-bit key", ubits, tbits);

  hipLaunchKernelGGL(make_keys_kernel, dim3(grid), dim3(256), 0, s, user_ids, timestamps, n_rows, r.umin, r.tmin, tbits, keys[0], idx[0]);
  int cur = 0;
  for (int shift = 0; shift < tbits + ubits; shift += RB) {
    hipLaunchKernelGGL(radix_hist_kernel, dim3(n), dim3(256), 0, s, keys[cur], n_rows, shift, hist, n);
    scan_exclusive(hist, (int64_t)NB * n, scan, nullptr, s);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(n), dim3(WAVE), 0, s, keys[cur], idx[cur], n_rows, shift, hist, n, keys[cur ^ 1], idx[cur ^ 1]);
    cur ^= 1;
  }
  RECNN_HIP(hipMemsetAsync(counters, 0, 8, s));
  hipLaunchKernelGGL(csr_gather_kernel, dim3(grid), dim3(256), 0, s, keys[cur], idx[cur], n_rows, tbits, item_keys, ratings, map_keys, map_vals,
                     n_map, items_out, ratings_out, flags, counters + 1);
  scan_exclusive(flags, n_rows, scan, counters, s);
  hipLaunchKernelGGL(csr_users_kernel, dim3(grid), dim3(256), 0, s, keys[cur], flags, n_rows, tbits, r.umin, users_out, user_off_out);
  if (order_out)     // the permutation itself, for callers that carry further columns along
    hipLaunchKernelGGL(csr_widen_order_kernel, dim3(grid), dim3(256), 0, s, idx[cur], n_rows, order_out);
  if (mapped_rows_out)
    hipLaunchKernelGGL(csr_map_rows_kernel, dim3(grid), dim3(256), 0, s, item_keys, n_rows, map_keys, map_vals, n_map, mapped_rows_out);
  uint32_t hc[2];
  RECNN_HIP(hipMemcpyAsync(hc, counters, 8, hipMemcpyDeviceToHost, s));
  RECNN_HIP(hipStreamSynchronize(s));
  host_counts[0] = hc[0];
  host_counts[1] = hc[1];
  host_counts[2] = tbits + ubits;
  return recnn_check_hip(hipGetLastError(), "csr_build");
}

}  // extern "C"
