// policy.hip -- the categorical head of the REINFORCE policy (gfx950).
//
// Replaces, for `DiscreteActor` (recnn/nn/models.py:76-184):
//   F.softmax(action_scores)                                   models.py:95-99
//   Categorical(probs).sample() / .log_prob(action)            models.py:107-111, :116-141
//   their autograd backward (log -> clamp -> normalise -> softmax)
// and the one-hot action rows of `batch_contstate_discaction` (recnn/data/utils.py:108-109).
//
// All of it is HBM-bound row work over [rows, n_items] fp32 matrices (n_items ~ 1e5): one 1024-thread workgroup per row,
// 16-byte accesses, the row is read from HBM once (online max/sum) and re-read from L2 / Infinity Cache for the
// normalising pass.  The sampler is an inverse-CDF walk in a fixed (thread-major) item order with a counter-based
// uniform per (seed, step, row): reproducible, independent of launch geometry history, no RNG state in memory.
#include "common.h"

namespace {

constexpr int PT = 1024;        // threads per row workgroup
constexpr int NWAVE = PT / WAVE;
constexpr float CAT_EPS = 1.1920928955078125e-07f;  // torch.finfo(float32).eps: Categorical clamps probs to [eps, 1-eps]

__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}

// sum over the workgroup, result in every thread.  `red` holds NWAVE floats.
__device__ inline float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NWAVE; ++i) t += red[i];
  return t;
}
__device__ inline float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NWAVE; ++i) t = fmaxf(t, red[i]);
  return t;
}

__device__ inline float4 load4_tail(const float* row, int i4, int n, float fill) {
  // element 4*i4 + c exists iff < n; rows are 16-byte aligned and padded to a multiple of 4 floats in memory
  float4 v = *(const float4*)(row + 4 * (int64_t)i4);
  const int base = 4 * i4;
  if (base + 3 >= n) {
    if (base + 1 >= n) v.y = fill;
    if (base + 2 >= n) v.z = fill;
    v.w = fill;
  }
  return v;
}

// One workgroup per row.
//   SOFTMAX: x holds logits on entry, probabilities on exit (p = exp(x - max) / sum);  else x holds (unnormalised)
//            probabilities and is left untouched.
//   sample : draw actions[row] ~ p / sum(p);  else actions[row] is read (or ignored when actions == NULL).
// logprob[row] = log(clamp(p[a] / sum(p), eps, 1 - eps)), rowstat[row] = {max, sum exp, sum p, clamped}.
template <bool SOFTMAX>
__global__ __launch_bounds__(PT) void categorical_kernel(float* __restrict__ x, int64_t ld, int n, int sample, uint32_t key,
                                                         int64_t* __restrict__ actions, float* __restrict__ logprob,
                                                         float* __restrict__ rowstat) {
  __shared__ float red[NWAVE];
  __shared__ float part[PT];
  __shared__ int owner_s;
  __shared__ float before_s;
  const int row = blockIdx.x, tid = threadIdx.x;
  float* xr = x + (int64_t)row * ld;
  const int n4 = (n + 3) >> 2;

  float mx = 0.f, denom = 1.f;
  if constexpr (SOFTMAX) {
    // online max / sum: one pass over HBM
    float m = -INFINITY, s = 0.f;
    for (int i = tid; i < n4; i += PT) {
      const float4 v = load4_tail(xr, i, n, -INFINITY);
      const float vm = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
      if (vm > m) { s *= expf(m - vm); m = vm; }
      s += expf(v.x - m) + expf(v.y - m) + expf(v.z - m) + expf(v.w - m);
    }
    mx = block_max(m, red);
    s = (m == -INFINITY) ? 0.f : s * expf(m - mx);
    denom = block_sum(s, red);
  }

  // normalising pass (row now comes from L2): write p, accumulate this thread's share of sum(p)
  float mine = 0.f;
  const float inv = 1.f / denom;
  for (int i = tid; i < n4; i += PT) {
    float4 v = load4_tail(xr, i, n, SOFTMAX ? -INFINITY : 0.f);
    if constexpr (SOFTMAX) {
      v.x = expf(v.x - mx) * inv; v.y = expf(v.y - mx) * inv; v.z = expf(v.z - mx) * inv; v.w = expf(v.w - mx) * inv;
      *(float4*)(xr + 4 * (int64_t)i) = v;   // padding columns become exp(-inf) = 0
    }
    mine += (v.x + v.y) + (v.z + v.w);
  }
  part[tid] = mine;
  const float psum = block_sum(mine, red);   // (also orders part[] and the p stores before what follows)

  int64_t a = -1;
  if (sample) {
    // inverse CDF in thread-major order: thread t owns items {4(t + k PT) + c}
    const uint32_t h = mix32(key ^ mix32((uint32_t)row * 0x9E3779B1u + 0x3C6EF372u));
    const float target = (float)(h >> 8) * (1.0f / 16777216.0f) * psum;
    if (tid < WAVE) {
      // lane l sums partials [16 l, 16 l + 16), wave scan, then the owning lane walks its 16 partials
      float loc = 0.f;
#pragma unroll
      for (int j = 0; j < PT / WAVE; ++j) loc += part[tid * (PT / WAVE) + j];
      float inc = loc;
#pragma unroll
      for (int o = 1; o < WAVE; o <<= 1) {
        const float t = __shfl_up(inc, o, WAVE);
        if (tid >= o) inc += t;
      }
      const float exc = inc - loc;
      const bool hit = (target >= exc) && (target < inc);
      const uint64_t ballot = __ballot(hit);
      int lane = ballot ? (int)__builtin_ctzll(ballot) : WAVE - 1;  // rounding fell off the end: take the tail
      if (tid == lane) {
        float run = exc;
        int own = tid * (PT / WAVE) + (PT / WAVE) - 1;
        float before = 0.f;
        bool found = false;
        for (int j = 0; j < PT / WAVE; ++j) {
          const float pj = part[tid * (PT / WAVE) + j];
          if (!found && target < run + pj) { own = tid * (PT / WAVE) + j; before = run; found = true; }
          run += pj;
        }
        if (!found) before = run - part[own];
        owner_s = own;
        before_s = before;
      }
    }
    __syncthreads();
    if (tid == owner_s) {
      float run = before_s;
      int pick = -1, last_pos = -1;
      for (int i = tid; i < n4 && pick < 0; i += PT) {
        const float4 v = load4_tail(xr, i, n, 0.f);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (pick < 0 && e[c] > 0.f) {
            last_pos = 4 * i + c;
            run += e[c];
            if (target < run) pick = 4 * i + c;
          }
        }
      }
      if (pick < 0) pick = last_pos >= 0 ? last_pos : 0;   // rounding at the very end of the CDF / degenerate row
      actions[row] = pick;
      a = pick;
    }
  } else if (actions && tid == 0) {
    a = actions[row];
    if (a < 0 || a >= n) a = -1;
  }

  if (a >= 0) {   // exactly one thread
    const float q = xr[a] / psum;
    const float qc = fminf(fmaxf(q, CAT_EPS), 1.f - CAT_EPS);
    if (logprob) logprob[row] = logf(qc);
    if (rowstat) rowstat[4 * (int64_t)row + 3] = (q < CAT_EPS || q > 1.f - CAT_EPS) ? 1.f : 0.f;
  } else if (tid == 0 && !sample) {
    if (logprob && actions) logprob[row] = NAN;   // action out of range
    if (rowstat) rowstat[4 * (int64_t)row + 3] = 0.f;
  }
  if (rowstat && tid == 1) {
    rowstat[4 * (int64_t)row + 0] = mx;
    rowstat[4 * (int64_t)row + 1] = denom;
    rowstat[4 * (int64_t)row + 2] = psum;
  }
}

// d logits of  sum_rows g[row] * log(clamp(p[a]/sum p)) :   g (onehot(a) - p / sum p),  zero where the clamp was active.
// Grid: (column groups of 256 float4, slabs of 32 rows).  Also writes the column sums of each slab (bias gradient partials).
constexpr int BWD_ROWS = 32;
__device__ inline float4 load_out4(const float* p) { return *(const float4*)p; }
__device__ inline float4 load_out4(const bf16_t* p) {
  const uint2 u = *(const uint2*)p;
  return make_float4(bf2f((bf16_t)(u.x & 0xFFFFu)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xFFFFu)), bf2f((bf16_t)(u.y >> 16)));
}
__device__ inline void store_out4(float* p, float4 v) { *(float4*)p = v; }
__device__ inline void store_out4(bf16_t* p, float4 v) { *(uint2*)p = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)); }

// OUT = float, or bf16 for the bf16 catalogue GEMMs that consume dlogits (the column sums are taken before the rounding)
template <class OUT>
__global__ __launch_bounds__(256) void logprob_bwd_kernel(const float* __restrict__ p, int64_t ldp, int rows, int n,
                                                          const int64_t* __restrict__ actions, const float* __restrict__ g,
                                                          const float* __restrict__ rowstat, OUT* __restrict__ d, int64_t ldd,
                                                          int accumulate, float* __restrict__ colpart) {
  __shared__ float gs[BWD_ROWS], is[BWD_ROWS];
  __shared__ int as[BWD_ROWS];
  const int r0 = blockIdx.y * BWD_ROWS;
  const int nr = min(BWD_ROWS, rows - r0);
  if (threadIdx.x < BWD_ROWS) {
    const int r = r0 + threadIdx.x;
    float gv = 0.f, iv = 0.f;
    int av = -1;
    if (threadIdx.x < nr) {
      const bool clamped = rowstat[4 * (int64_t)r + 3] != 0.f;
      gv = (g && !clamped) ? g[r] : 0.f;
      iv = 1.f / rowstat[4 * (int64_t)r + 2];
      av = (int)actions[r];
    }
    gs[threadIdx.x] = gv; is[threadIdx.x] = iv; as[threadIdx.x] = av;
  }
  __syncthreads();
  const int i4 = blockIdx.x * 256 + threadIdx.x;
  const int n4 = (n + 3) >> 2;
  if (i4 >= n4) return;
  const int c0 = 4 * i4;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = 0; j < nr; ++j) {
    const int64_t r = r0 + j;
    const float4 pv = *(const float4*)(p + r * ldp + c0);
    const float gv = gs[j], s = -gv * is[j];
    float4 o = make_float4(pv.x * s, pv.y * s, pv.z * s, pv.w * s);
    const int rel = as[j] - c0;
    if (rel == 0) o.x += gv; else if (rel == 1) o.y += gv; else if (rel == 2) o.z += gv; else if (rel == 3) o.w += gv;
    if (accumulate) {
      const float4 old = load_out4(d + r * ldd + c0);
      o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
    }
    if (c0 + 3 >= n) {   // keep the padding columns zero (they are contraction padding of the dX GEMM)
      if (c0 + 1 >= n) o.y = 0.f;
      if (c0 + 2 >= n) o.z = 0.f;
      o.w = 0.f;
    }
    store_out4(d + r * ldd + c0, o);
    cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
  }
  if (colpart) *(float4*)(colpart + (int64_t)blockIdx.y * (4 * (int64_t)n4) + c0) = cs;
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ colpart, int slabs, int n4, int n,
                                                            float* __restrict__ out) {
  const int i4 = blockIdx.x * 256 + threadIdx.x;
  if (i4 >= n4) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < slabs; ++k) {
    const float4 v = *(const float4*)(colpart + (int64_t)k * (4 * (int64_t)n4) + 4 * i4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float e[4] = {s.x, s.y, s.z, s.w};
  for (int c = 0; c < 4; ++c)
    if (4 * i4 + c < n) out[4 * i4 + c] = e[c];
}

// out[c] = sum_r x[r, c], rows added in order (the bias gradient of a catalogue-wide Linear: column sums of d logits [rows, n_items];
// as a dW GEMM against a column of ones it cost a 277 us launch, this reads the 102 MB once)
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ x, int64_t ld, int rows, int n4, int n,
                                                          float* __restrict__ out) {
  const int i4 = blockIdx.x * 256 + threadIdx.x;
  if (i4 >= n4) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < rows; ++r) {
    const float4 v = *(const float4*)(x + (int64_t)r * ld + 4 * i4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float e[4] = {s.x, s.y, s.z, s.w};
  for (int c = 0; c < 4; ++c)
    if (4 * i4 + c < n) out[4 * i4 + c] = e[c];
}

// d logits (+)= p (dp - sum_j dp_j p_j): backward of p = softmax(logits).  One workgroup per row.
__global__ __launch_bounds__(PT) void softmax_bwd_kernel(const float* __restrict__ p, int64_t ldp, int n,
                                                         const float* __restrict__ dp, int64_t lddp, float* __restrict__ d,
                                                         int64_t ldd) {
  __shared__ float red[NWAVE];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* pr = p + (int64_t)row * ldp;
  const float* gr = dp + (int64_t)row * lddp;
  float* dr = d + (int64_t)row * ldd;
  float dot = 0.f;
  for (int j = tid; j < n; j += PT) dot += pr[j] * gr[j];
  dot = block_sum(dot, red);
  const int npad = (n + 3) & ~3;
  for (int j = tid; j < npad; j += PT) dr[j] = j < n ? pr[j] * (gr[j] - dot) : 0.f;
}

// out[r, :] = onehot(idx[r])  (rows of ld floats; columns [n, ld) zeroed too)
// Rows are walked with a grid stride (gridDim.y is capped at 65535); an index outside [0, n) leaves its row all zero -- it
// never lands in the padding columns [n, ld) -- and the host wrapper refuses such indices like the reference's scatter_.
__global__ __launch_bounds__(256) void onehot_kernel(const int64_t* __restrict__ idx, int rows, int n, float* __restrict__ out, int64_t ld) {
  const int i4 = blockIdx.x * 256 + threadIdx.x;
  if (4 * (int64_t)i4 >= ld) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    const int64_t id = idx[r];
    const int64_t rel = id - 4 * (int64_t)i4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (id >= 0 && id < n) {
      if (rel == 0) v.x = 1.f; else if (rel == 1) v.y = 1.f; else if (rel == 2) v.z = 1.f; else if (rel == 3) v.w = 1.f;
    }
    *(float4*)(out + (int64_t)r * ld + 4 * (int64_t)i4) = v;
  }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// dst[c][r] = src[r][c]: fp32 [R, C] -> fp32 or bf16 [C, ldt].  The backward of the policy head contracts d logits [rows, n_items]
// with W2 [n_items, hidden] over the CATALOGUE -- W2 is k-strided for that product; its transpose (made once per weight version)
// puts the contraction on the contiguous axis for the LDS-DMA kernels.  64 x 64 tiles through LDS (pitch 65: no bank conflicts),
// both sides coalesced: 256-byte reads, 128 / 256-byte writes.  (torch's strided copy_ takes 3.4 ms for [100k, 2048]; this 0.3.)
template <class TD>
__global__ __launch_bounds__(256) void transpose_rows_kernel(const float* __restrict__ src, int64_t ld, int R, int C, TD* __restrict__ dst,
                                                             int64_t ldt) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int r = r0 + ty + 4 * j, c = c0 + tx;
    tile[ty + 4 * j][tx] = (r < R && c < C) ? src[(int64_t)r * ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = c0 + ty + 4 * j, r = r0 + tx;
    if (c < C && r < R) tc_store(dst + (int64_t)c * ldt + r, tile[tx][ty + 4 * j]);
  }
}


// ---- the softmax of a row whose columns are SHARDED over ranks (recnn_amd/parallel.py VocabParallelPolicyFunction; round 6: these were
// ATen amax / exp / sum / div / gather calls between the all-reduces).  Three passes, one workgroup per row, the all-reduces of the row
// maxima / sums / chosen probabilities between them stay with torch.distributed:
//   rowmax        m[r] = max_j x[r, j]
//   exp_rowsum    x[r, j] = exp(x[r, j] - m[r]) in place (m: the max over ALL shards), s[r] = sum_j of it
//   norm_pick     x[r, j] /= s[r] (s: the sum over all shards) in place -> this shard's probabilities; pa[r] = x[r, a_r] when this shard
//                 owns the row's action (0 <= a_r < n), else 0
__global__ __launch_bounds__(PT) void shard_rowmax_kernel(const float* __restrict__ x, int64_t ld, int n, float* __restrict__ m) {
  __shared__ float red[NWAVE];
  const float* xr = x + (int64_t)blockIdx.x * ld;
  const int n4 = (n + 3) >> 2;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n4; i += PT) {
    const float4 v = load4_tail(xr, i, n, -INFINITY);
    mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
  }
  mx = block_max(mx, red);
  if (threadIdx.x == 0) m[blockIdx.x] = mx;
}
__global__ __launch_bounds__(PT) void shard_exp_rowsum_kernel(float* __restrict__ x, int64_t ld, int n, const float* __restrict__ m,
                                                              float* __restrict__ ssum) {
  __shared__ float red[NWAVE];
  float* xr = x + (int64_t)blockIdx.x * ld;
  const int n4 = (n + 3) >> 2;
  const float mx = m[blockIdx.x];
  float mine = 0.f;
  for (int i = threadIdx.x; i < n4; i += PT) {
    float4 v = load4_tail(xr, i, n, -INFINITY);
    v.x = expf(v.x - mx); v.y = expf(v.y - mx); v.z = expf(v.z - mx); v.w = expf(v.w - mx);   // padding columns: exp(-inf) = 0
    *(float4*)(xr + 4 * (int64_t)i) = v;
    mine += (v.x + v.y) + (v.z + v.w);
  }
  const float tot = block_sum(mine, red);
  if (threadIdx.x == 0) ssum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PT) void shard_norm_pick_kernel(float* __restrict__ x, int64_t ld, int n, const float* __restrict__ ssum,
                                                             const int64_t* __restrict__ local, float* __restrict__ pa) {
  float* xr = x + (int64_t)blockIdx.x * ld;
  const int n4 = (n + 3) >> 2;
  const float s = ssum[blockIdx.x];
  for (int i = threadIdx.x; i < n4; i += PT) {
    float4 v = *(const float4*)(xr + 4 * (int64_t)i);
    v.x = v.x / s; v.y = v.y / s; v.z = v.z / s; v.w = v.w / s;          // (a true division, like torch's div_)
    *(float4*)(xr + 4 * (int64_t)i) = v;
  }
  __syncthreads();                                                         // the row's stores are visible to thread 0's read below
  if (threadIdx.x == 0) {
    const int64_t a = local[blockIdx.x];
    pa[blockIdx.x] = (a >= 0 && a < n) ? xr[a] : 0.f;
  }
}
// d logits of this shard for d loss / d log(clamp(p_a)) = g:  dlog[r, j] = -g[r] p[r, j]  (+ g[r] at j = a_r when the shard owns it)
__global__ __launch_bounds__(256) void shard_logprob_bwd_kernel(const float* __restrict__ p, int64_t ldp, int rows, int n,
                                                                const int64_t* __restrict__ local, const float* __restrict__ g,
                                                                float* __restrict__ dlog, int64_t ldd) {
  const int n4 = (n + 3) >> 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    const float gr = g[r];
    const int64_t a = local[r];
    float4 v = *(const float4*)(p + (int64_t)r * ldp + 4 * (int64_t)i);
    v.x = v.x * -gr; v.y = v.y * -gr; v.z = v.z * -gr; v.w = v.w * -gr;
    if (a >= 0 && a < n && (a >> 2) == i) {
      const int c = (int)(a & 3);
      if (c == 0) v.x += gr; else if (c == 1) v.y += gr; else if (c == 2) v.z += gr; else v.w += gr;
    }
    *(float4*)(dlog + (int64_t)r * ldd + 4 * (int64_t)i) = v;
  }
}
}  // namespace

extern "C" {

int recnn_categorical_rows(float* x, int64_t ld, int rows, int n, int flags, uint32_t seed, int32_t step, int64_t* actions,
                           float* logprob, float* rowstat, void* stream) {
  RECNN_REQUIRE(x && rows >= 0 && n > 0, "categorical: bad arguments");
  RECNN_REQUIRE(aligned16(x) && ld % 4 == 0 && ld >= ((n + 3) & ~3), "categorical: rows must be 16-byte aligned and padded to 4 floats");
  const int sample = (flags & RECNN_CAT_SAMPLE) ? 1 : 0;
  RECNN_REQUIRE(!sample || actions, "categorical: sampling needs an actions buffer");
  if (rows == 0) return 0;
  const uint32_t key = mask_key(seed, step, 0x51u);
  if (flags & RECNN_CAT_SOFTMAX)
    hipLaunchKernelGGL(categorical_kernel<true>, dim3(rows), dim3(PT), 0, (hipStream_t)stream, x, ld, n, sample, key, actions, logprob, rowstat);
  else
    hipLaunchKernelGGL(categorical_kernel<false>, dim3(rows), dim3(PT), 0, (hipStream_t)stream, x, ld, n, sample, key, actions, logprob, rowstat);
  return recnn_check_hip(hipGetLastError(), "categorical_kernel launch");
}

int recnn_logprob_bwd(const float* p, int64_t ldp, int rows, int n, const int64_t* actions, const float* g, const float* rowstat,
                      void* dlogits, int64_t ldd, int flags, float* colsum, float* scratch, void* stream) {
  RECNN_REQUIRE(p && actions && rowstat && dlogits && rows >= 0 && n > 0, "logprob_bwd: bad arguments");
  RECNN_REQUIRE(aligned16(p) && aligned16(dlogits) && ldp % 4 == 0 && ldd % 4 == 0 && ldp >= ((n + 3) & ~3) && ldd >= ((n + 3) & ~3),
                "logprob_bwd: rows must be 16-byte aligned and padded to 4 floats");
  RECNN_REQUIRE(!colsum || scratch, "logprob_bwd: column sums need the scratch buffer (ceil(rows/32) * round4(n) floats)");
  if (rows == 0) return 0;
  const int n4 = (n + 3) >> 2, slabs = (rows + BWD_ROWS - 1) / BWD_ROWS;
  const int accumulate = (flags & RECNN_LPB_ACCUMULATE) ? 1 : 0;
  if (flags & RECNN_LPB_BF16)
    hipLaunchKernelGGL(logprob_bwd_kernel<bf16_t>, dim3((n4 + 255) / 256, slabs), dim3(256), 0, (hipStream_t)stream, p, ldp, rows, n,
                       actions, g, rowstat, (bf16_t*)dlogits, ldd, accumulate, colsum ? scratch : nullptr);
  else
    hipLaunchKernelGGL(logprob_bwd_kernel<float>, dim3((n4 + 255) / 256, slabs), dim3(256), 0, (hipStream_t)stream, p, ldp, rows, n,
                       actions, g, rowstat, (float*)dlogits, ldd, accumulate, colsum ? scratch : nullptr);
  if (colsum)
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, slabs, n4, n, colsum);
  return recnn_check_hip(hipGetLastError(), "logprob_bwd launch");
}

int recnn_softmax_bwd(const float* p, int64_t ldp, int rows, int n, const float* dprobs, int64_t lddp, float* dlogits, int64_t ldd,
                      void* stream) {
  RECNN_REQUIRE(p && dprobs && dlogits && rows >= 0 && n > 0, "softmax_bwd: bad arguments");
  RECNN_REQUIRE(ldd >= ((n + 3) & ~3), "softmax_bwd: output rows must be padded to 4 floats");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(rows), dim3(PT), 0, (hipStream_t)stream, p, ldp, n, dprobs, lddp, dlogits, ldd);
  return recnn_check_hip(hipGetLastError(), "softmax_bwd launch");
}

int recnn_onehot_rows(const int64_t* idx, int rows, int n, float* out, int64_t ld, void* stream) {
  RECNN_REQUIRE(idx && out && rows >= 0 && n > 0 && ld >= n, "onehot: bad arguments");
  RECNN_REQUIRE(aligned16(out) && ld % 4 == 0, "onehot: rows must be 16-byte aligned");
  if (rows == 0) return 0;
  const int l4 = (int)(ld / 4);
  hipLaunchKernelGGL(onehot_kernel, dim3((l4 + 255) / 256, rows < 65535 ? rows : 65535), dim3(256), 0, (hipStream_t)stream, idx, rows, n, out, ld);
  return recnn_check_hip(hipGetLastError(), "onehot_kernel launch");
}

int recnn_shard_softmax_pass(float* x, int64_t ld, int rows, int n, int pass, float* rowval, const int64_t* local, float* pa, void* stream) {
  RECNN_REQUIRE(x && rowval && rows >= 0 && n > 0 && pass >= 0 && pass <= 2, "shard_softmax_pass: bad arguments");
  RECNN_REQUIRE(aligned16(x) && ld % 4 == 0 && ld >= ((n + 3) & ~3), "shard_softmax_pass: rows must be 16-byte aligned and padded to 4 floats");
  RECNN_REQUIRE(pass != 2 || (local && pa), "shard_softmax_pass: the last pass needs the local action indices and the output");
  if (rows == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (pass == 0) hipLaunchKernelGGL(shard_rowmax_kernel, dim3(rows), dim3(PT), 0, s, x, ld, n, rowval);
  else if (pass == 1) hipLaunchKernelGGL(shard_exp_rowsum_kernel, dim3(rows), dim3(PT), 0, s, x, ld, n, rowval, pa);
  else hipLaunchKernelGGL(shard_norm_pick_kernel, dim3(rows), dim3(PT), 0, s, x, ld, n, rowval, local, pa);
  return recnn_check_hip(hipGetLastError(), "shard_softmax_pass launch");
}

int recnn_shard_logprob_bwd(const float* p, int64_t ldp, int rows, int n, const int64_t* local, const float* g, float* dlogits, int64_t ldd,
                            void* stream) {
  RECNN_REQUIRE(p && local && g && dlogits && rows >= 0 && n > 0, "shard_logprob_bwd: bad arguments");
  RECNN_REQUIRE(aligned16(p) && aligned16(dlogits) && ldp % 4 == 0 && ldd % 4 == 0 && ldp >= ((n + 3) & ~3) && ldd >= ((n + 3) & ~3),
                "shard_logprob_bwd: rows must be 16-byte aligned and padded to 4 floats");
  if (rows == 0) return 0;
  const int n4 = (n + 3) >> 2;
  hipLaunchKernelGGL(shard_logprob_bwd_kernel, dim3((n4 + 255) / 256, rows < 256 ? rows : 256), dim3(256), 0, (hipStream_t)stream, p, ldp, rows, n,
                     local, g, dlogits, ldd);
  return recnn_check_hip(hipGetLastError(), "shard_logprob_bwd launch");
}

int recnn_colsum_rows(const float* x, int64_t ld, int rows, int n, float* out, void* stream) {
  RECNN_REQUIRE(x && out && rows >= 0 && n > 0, "colsum_rows: bad arguments");
  RECNN_REQUIRE(aligned16(x) && ld % 4 == 0 && ld >= ((n + 3) & ~3), "colsum_rows: rows must be 16-byte aligned and padded to 4 floats");
  const int n4 = (n + 3) >> 2;
  hipLaunchKernelGGL(colsum_rows_kernel, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ld, rows, n4, n, out);
  return recnn_check_hip(hipGetLastError(), "colsum_rows_kernel launch");
}

int recnn_transpose_rows(const float* src, int64_t ld, int rows, int cols, void* dst, int64_t ldt, int dst_bf16, void* stream) {
  RECNN_REQUIRE(src && dst && rows >= 0 && cols >= 0 && ld >= cols && ldt >= rows, "transpose_rows: bad arguments");
  if (rows == 0 || cols == 0) return 0;
  RECNN_REQUIRE((rows + 63) / 64 <= 65535, "transpose_rows: more than 4M rows");
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (dst_bf16) hipLaunchKernelGGL(transpose_rows_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, ld, rows, cols, (bf16_t*)dst, ldt);
  else hipLaunchKernelGGL(transpose_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, ld, rows, cols, (float*)dst, ldt);
  return recnn_check_hip(hipGetLastError(), "transpose_rows_kernel launch");
}

}  // extern "C"
