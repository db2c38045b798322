// optim.hip -- gradient reduction, the clip quirk, Adam, soft target update, shadow refresh (gfx950).
//
// Replaces:
//   torch.optim.Adam.step  as injected by the reference's users   recnn/nn/update/misc.py:44, ddpg.py:93,
//                                                                  td3.py:97,101,134
//   torch.nn.utils.clip_grad_norm_(policy.parameters(), -1, 1)     ddpg.py:92, td3.py:133
//   recnn/utils/misc.py:1-5  soft_update   (target*(1-tau) + param*tau, that operand order)
// Flat chunks of each parameter tensor per workgroup (optim.h) over a flat fp32 arena; the same pass writes the compute-type
// "shadow" copy of the weights (zero-padded, 16-byte aligned rows; critic W1 columns rotated to the
// packed [action | state] batch layout) that the MFMA GEMMs read, and optionally the soft-updated target.
#include "optim.h"
#include "x3.h"
#include "comm_dev.h"
#include "gather_dev.h"

double recnn_snap7(float x) {
  char buf[40];
  snprintf(buf, sizeof(buf), "%.7g", (double)x);
  return strtod(buf, nullptr);
}

__device__ inline int find_tensor(const NetLayout& L, int b) {
  int ti = 0;
#pragma unroll
  for (int i = 1; i < 6; ++i)
    if (b >= L.t[i].blk0) ti = i;
  return ti;
}

__device__ inline float block_sum256(float v, float* red /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- which elements of tensor T this thread owns (see optim.h for the mapping) -------------------------------
struct Own {
  int64_t e;   // first element (index inside the tensor)
  int cnt;     // 0..4 consecutive elements
  bool vec;    // 16-byte accesses allowed
};
__device__ inline Own own_elems(const TensorSeg& T, int bt) {
  const int64_t n = (int64_t)T.rows * T.cols;
  Own o;
  if (T.small) {
    o.e = (int64_t)bt * OPT_SMALL_ELEMS + (threadIdx.x & 63);
    o.cnt = (threadIdx.x < 64 && o.e < n) ? 1 : 0;
    o.vec = false;
  } else {
    o.e = (int64_t)bt * OPT_BLOCK_ELEMS + threadIdx.x * 4;
    const int64_t left = n - o.e;
    o.cnt = left >= 4 ? 4 : (left > 0 ? (int)left : 0);
    o.vec = T.vec4 && o.cnt == 4;
  }
  return o;
}
__device__ inline void load_own(const float* __restrict__ src, const Own& o, float out[4]) {
  if (o.vec) {
    const float4 x = *(const float4*)src;
    out[0] = x.x; out[1] = x.y; out[2] = x.z; out[3] = x.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = j < o.cnt ? src[j] : 0.f;
  }
}
__device__ inline void store_own(float* __restrict__ dst, const Own& o, const float v[4]) {
  if (o.vec) {
    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < o.cnt) dst[j] = v[j];
  }
}

// Sum of the gradient partial slabs for the thread's elements; fixed order (deterministic).  All threads of the
// workgroup must call it (the small-tensor path meets at a barrier).
__device__ inline void slab_grads(const TensorSeg& T, int bt, const Own& o, float g[4], float (*sp)[OPT_SMALL_ELEMS]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] = 0.f;
  if (T.small) {
    // wave w sums slabs [w*q, (w+1)*q) of element (lane); up to 32 loads in flight per thread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = (int64_t)bt * OPT_SMALL_ELEMS + lane;
    const bool in = e < (int64_t)T.rows * T.cols;
    const int q = (T.nslab + 3) >> 2;
    int s = wave * q;
    const int s_end = min(s + q, T.nslab);
    float acc = 0.f;
    if (in) {
      for (; s + 32 <= s_end; s += 32) {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = T.gpart[(int64_t)(s + j) * T.slab_stride + e];
#pragma unroll
        for (int w = 16; w > 0; w >>= 1)
#pragma unroll
          for (int j = 0; j < w; ++j) v[j] += v[j + w];
        acc += v[0];
      }
      for (; s + 8 <= s_end; s += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = T.gpart[(int64_t)(s + j) * T.slab_stride + e];
        acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
      for (; s < s_end; ++s) acc += T.gpart[(int64_t)s * T.slab_stride + e];
    }
    sp[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) g[0] = (sp[0][lane] + sp[1][lane]) + (sp[2][lane] + sp[3][lane]);
    return;
  }
  if (o.cnt == 0) return;
  const float* base = T.gpart + o.e;
  int s = 0;
  if (o.vec) {
    for (; s + 8 <= T.nslab; s += 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *(const float4*)(base + (int64_t)(s + j) * T.slab_stride);
      g[0] += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
      g[1] += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
      g[2] += ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
      g[3] += ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
    }
    for (; s < T.nslab; ++s) {
      const float4 v = *(const float4*)(base + (int64_t)s * T.slab_stride);
      g[0] += v.x; g[1] += v.y; g[2] += v.z; g[3] += v.w;
    }
  } else {
    for (; s < T.nslab; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < o.cnt) g[j] += base[(int64_t)s * T.slab_stride + j];
  }
}

// g_flat = sum of partial slabs (split-K slabs of the dW GEMMs, row-tile column sums for biases)
__global__ __launch_bounds__(256) void grad_reduce_kernel(const NetLayout L, float* __restrict__ gflat,
                                                          float* __restrict__ l1part) {
  __shared__ float red[4];
  __shared__ float sp[4][OPT_SMALL_ELEMS];
  const int b = blockIdx.x;
  const TensorSeg& T = L.t[find_tensor(L, b)];
  const int bt = b - T.blk0;
  const Own o = own_elems(T, bt);
  float g[4];
  slab_grads(T, bt, o, g, sp);
  if (o.cnt) store_own(gflat + T.p_off + o.e, o, g);
  if (l1part) {
    float l1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < o.cnt) l1 += fabsf(g[j]);
    const float tot = block_sum256(l1, red);
    if (threadIdx.x == 0) l1part[b] = tot;
  }
}

int grad_reduce_launch(const NetLayout& L, float* gflat, float* l1part, hipStream_t s) {
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(L.nblk), dim3(256), 0, s, L, gflat, l1part);
  return recnn_check_hip(hipGetLastError(), "grad_reduce_kernel");
}

// per-workgroup |g| partial sums of a flat gradient (after any all-reduce), same mapping as grad_reduce_kernel
__global__ __launch_bounds__(256) void l1_blocks_kernel(const NetLayout L, const float* __restrict__ gflat,
                                                        float* __restrict__ l1part) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const TensorSeg& T = L.t[find_tensor(L, b)];
  const Own o = own_elems(T, b - T.blk0);
  float g[4], l1 = 0.f;
  load_own(gflat + T.p_off + o.e, o, g);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < o.cnt) l1 += fabsf(g[j]);
  const float tot = block_sum256(l1, red);
  if (threadIdx.x == 0) l1part[b] = tot;
}

int l1_blocks_launch(const NetLayout& L, const float* gflat, float* l1part, hipStream_t s) {
  hipLaunchKernelGGL(l1_blocks_kernel, dim3(L.nblk), dim3(256), 0, s, L, gflat, l1part);
  return recnn_check_hip(hipGetLastError(), "l1_blocks_kernel");
}

__device__ inline float clip_coef(const float* l1part, int n, float grad_scale, float* red) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += l1part[i];
  const float total = block_sum256(s, red) * grad_scale;
  // clip_grad_norm_(max_norm=-1, norm_type=1): coef = -1/(total+1e-6), clamped to <= 1
  return fminf(-1.0f / (total + 1e-6f), 1.0f);
}

// Data parallel: this thread's slab-summed gradient elements g[0 .. o.cnt) at flat offset e = e0 + ... of the workgroup's range
// [e0, e0 + n_blk) (a multiple of 4 floats from a multiple of 4) -> the sums over the ranks.  All threads of the workgroup call.
__device__ inline void exchange_grads(const ApplyArgs& a, int64_t e, int64_t e0, int n_blk, const Own& o, float g[4], int b) {
  const CommPort& c = a.comm;
  const uint32_t ep = comm_epoch(c);
  char* own = c.peer[c.rank];
  if (o.cnt) {
    float* in = comm_in_of(own) + c.off;
    if (o.vec) comm_st4(comm_rsrc(in), e >> 2, f32x4{g[0], g[1], g[2], g[3]});
    else
      for (int j = 0; j < o.cnt; ++j) comm_st1(in + e + j, g[j]);
  }
  comm_raise(c, false, b, ep);
  comm_wait(c, false, b, ep);
  {  // this rank's share of the workgroup's float4 groups: summed over the ranks in rank order, scattered to every out[]
    const int groups = (n_blk + 3) >> 2, piece = (groups + c.world - 1) / c.world;
    const int lo = piece * c.rank, hi = lo + piece < groups ? lo + piece : groups;
    const int64_t g0 = (c.off + e0) >> 2;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
      f32x4 v[COMM_MAX_WORLD];
#pragma unroll
      for (int p = 0; p < COMM_MAX_WORLD; ++p)
        if (p < c.world) v[p] = comm_ld4(comm_rsrc(comm_in_of(c.peer[p])), g0 + i);
      f32x4 s = v[0];
#pragma unroll
      for (int p = 1; p < COMM_MAX_WORLD; ++p)
        if (p < c.world) s += v[p];
#pragma unroll
      for (int p = 0; p < COMM_MAX_WORLD; ++p)
        if (p < c.world) comm_st4(comm_rsrc(comm_out_of(c.peer[p], c.cap)), g0 + i, s);
    }
  }
  comm_raise(c, true, b, ep);
  comm_wait(c, true, b, ep);
  if (o.cnt) {
    float* out = comm_out_of(own, c.cap) + c.off;
    if (o.vec) {
      const f32x4 v = comm_ld4(comm_rsrc(out), e >> 2);
      g[0] = v[0]; g[1] = v[1]; g[2] = v[2]; g[3] = v[3];
    } else {
      for (int j = 0; j < o.cnt; ++j) g[j] = comm_ld1(out + e + j);
    }
  }
  if (threadIdx.x == 0) comm_leave(c, ep, a.comm_nwg);
}

// Adam (+ clip quirk) + shadow refresh + soft target update: one pass, each element touched by exactly one thread,
// every load of the thread issued before the first use.
__device__ __forceinline__ void apply_body(const NetLayout& L, const ApplyArgs& a, const int b, float* red,
                                           float (*sp)[OPT_SMALL_ELEMS]) {
  const TensorSeg& T = L.t[find_tensor(L, b)];
  const int bt = b - T.blk0;
  const Own o = own_elems(T, bt);
  const int64_t e = T.p_off + o.e;

  float p[4], m[4], v[4], tp[4], g[4], sl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = m[j] = v[j] = tp[j] = g[j] = sl[j] = 0.f;
  const OptScalars S = opt_scalars(a);
  const bool la_sync = S.la_sync != 0;   // Lookahead: slow += alpha (p - slow); p = slow
  if (o.cnt) {
    load_own(a.p + e, o, p);
    if (a.do_adam) { load_own(a.m + e, o, m); load_own(a.v + e, o, v); }
    if (a.tgt_p) load_own(a.tgt_p + e, o, tp);
    if (la_sync) load_own(a.slow + e, o, sl);
  }
  if (a.do_adam) {
    if (a.from_slabs) {
      slab_grads(T, bt, o, g, sp);
      if (a.comm.world) {
        const int64_t nT = (int64_t)T.rows * T.cols, per = T.small ? OPT_SMALL_ELEMS : OPT_BLOCK_ELEMS;
        const int64_t left = nT - (int64_t)bt * per;
        exchange_grads(a, e, T.p_off + (int64_t)bt * per, (int)(left < per ? left : per), o, g, b);
      }
      if (o.cnt && a.g_out) store_own(a.g_out + e, o, g);
    } else if (o.cnt && a.g_sys) {
      // data parallel: the summed gradient is read where the collective's peers wrote it, bypassing the caches (a plain load
      // could hit a line cached from the previous step's sums); the bound arena gets a copy
      if (o.vec) {
        const f32x4 x = comm_ld4(comm_rsrc(a.g), e >> 2);
        g[0] = x[0]; g[1] = x[1]; g[2] = x[2]; g[3] = x[3];
      } else {
        for (int j = 0; j < o.cnt; ++j) g[j] = comm_ld1(a.g + e + j);
      }
      if (a.g_out) store_own(a.g_out + e, o, g);
    } else if (o.cnt) {
      load_own(a.g + e, o, g);
    }
  }
  float gs = a.grad_scale;
  if (a.n_l1 > 0) {
    const float coef = clip_coef(a.l1part, a.n_l1, a.grad_scale, red);
    if (a.coef_out && b == 0 && threadIdx.x == 0) a.coef_out[0] = coef;
    gs *= coef;
  }
  if (o.cnt == 0) return;
  if (a.do_adam) {
#pragma unroll
    for (int j = 0; j < 4; ++j) opt_elem(a, S, g[j], gs, p[j], m[j], v[j], sl[j]);
    store_own(a.m + e, o, m);
    store_own(a.v + e, o, v);
    store_own(a.p + e, o, p);
    if (la_sync) store_own(a.slow + e, o, sl);
  }
  if (a.tgt_p) {
#pragma unroll
    for (int j = 0; j < 4; ++j) tp[j] = soft_elem(tp[j], p[j], a.tau);
    store_own(a.tgt_p + e, o, tp);
  }
  if (T.sh_off >= 0 && (a.shadow || (a.tgt_p && a.tgt_shadow))) {
    int row = (int)(o.e / T.cols);
    int col = (int)(o.e - (int64_t)row * T.cols);
    if (a.tc_bf16 == RECNN_BF16X3) {   // split-bf16 shadow (x3.h): hi at the mapped column, lo 32 elements further
      {
        int c0 = col + T.col_rot;
        if (c0 >= T.cols) c0 -= T.cols;
        // four elements of one row inside one 4-aligned column run (no row end, no rotation wrap): two 8-byte stores per shadow
        if (o.cnt == 4 && col + 3 < T.cols && c0 + 3 < T.cols && !(c0 & 3) && !(T.sh_ld & 3) && !(T.sh_off & 3)) {
          const int64_t se = T.sh_off + (int64_t)row * T.sh_ld + x3_col(c0);
          uint2 hi, lo;
          if (a.shadow) {
            x3_split4(p, hi, lo);
            *(uint2*)((bf16_t*)a.shadow + se) = hi;
            *(uint2*)((bf16_t*)a.shadow + se + 32) = lo;
          }
          if (a.tgt_p && a.tgt_shadow) {
            x3_split4(tp, hi, lo);
            *(uint2*)((bf16_t*)a.tgt_shadow + se) = hi;
            *(uint2*)((bf16_t*)a.tgt_shadow + se + 32) = lo;
          }
          return;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < o.cnt) {
          int cc = col + T.col_rot;
          if (cc >= T.cols) cc -= T.cols;
          const int64_t se = T.sh_off + (int64_t)row * T.sh_ld;
          if (a.shadow) x3_store((bf16_t*)a.shadow + se, cc, p[j]);
          if (a.tgt_p && a.tgt_shadow) x3_store((bf16_t*)a.tgt_shadow + se, cc, tp[j]);
        }
        if (++col == T.cols) { col = 0; ++row; }
      }
      return;
    }
    const bool pairs = a.tc_bf16 && o.vec && !((T.cols | T.col_rot | T.sh_ld) & 1) && !(T.sh_off & 1);
    if (pairs) {  // two 4-byte stores instead of four 2-byte ones: a pair never straddles a row end or the rotation wrap
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        int cc = col + T.col_rot;
        if (cc >= T.cols) cc -= T.cols;
        const int64_t se = T.sh_off + (int64_t)row * T.sh_ld + cc;
        if (a.shadow) *(uint32_t*)((bf16_t*)a.shadow + se) = pack_bf2(p[j], p[j + 1]);
        if (a.tgt_p && a.tgt_shadow) *(uint32_t*)((bf16_t*)a.tgt_shadow + se) = pack_bf2(tp[j], tp[j + 1]);
        col += 2;
        if (col == T.cols) { col = 0; ++row; }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < o.cnt) {
        int cc = col + T.col_rot;
        if (cc >= T.cols) cc -= T.cols;
        const int64_t se = T.sh_off + (int64_t)row * T.sh_ld + cc;
        if (a.shadow) {
          if (a.tc_bf16) ((bf16_t*)a.shadow)[se] = f2bf(p[j]);
          else ((float*)a.shadow)[se] = p[j];
        }
        if (a.tgt_p && a.tgt_shadow) {
          if (a.tc_bf16) ((bf16_t*)a.tgt_shadow)[se] = f2bf(tp[j]);
          else ((float*)a.tgt_shadow)[se] = tp[j];
        }
      }
      if (++col == T.cols) { col = 0; ++row; }
    }
  }
}

__global__ __launch_bounds__(256) void apply_kernel(const NetLayout L, const ApplyArgs a) {
  kernarg_prefetch<(int)(sizeof(NetLayout) + sizeof(ApplyArgs))>();
  __shared__ float red[4];
  __shared__ float sp[4][OPT_SMALL_ELEMS];
  apply_body(L, a, blockIdx.x, red, sp);
}

// The optimizer pass of step t with the replay sampler + embedding gather of step t+1 as extra workgroups of the
// same launch: the gather is a chain of dependent memory latencies with almost no bandwidth or ALU demand, the
// optimizer pass streams -- run together, the gather's ~11 us disappear from the step's critical path.  The two
// roles touch disjoint memory (the gather fills the OTHER batch buffer set, engine.hip).
// The gather workgroups come first in the launch order: their latency chain starts at once and the optimizer
// workgroups stream underneath it.
__global__ __launch_bounds__(256) void apply_gather_kernel(const NetLayout L, const ApplyArgs a, const GatherArgs g, const int n_gather) {
  kernarg_prefetch<(int)(sizeof(NetLayout) + sizeof(ApplyArgs) + sizeof(GatherArgs))>();
  __shared__ float red[4];
  __shared__ float sp[4][OPT_SMALL_ELEMS];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if ((int)blockIdx.x < n_gather) frame_gather_body<4, 4>(g, blockIdx.x, smem_raw);
  else apply_body(L, a, (int)blockIdx.x - n_gather, red, sp);
}

void apply_args_finish(ApplyArgs* a) {
  a->log_beta1 = log(recnn_snap7(a->beta1));
  a->log_beta2 = log(recnn_snap7(a->beta2));
  a->omb1 = (float)(1.0 - recnn_snap7(a->beta1));
  a->omb2 = (float)(1.0 - recnn_snap7(a->beta2));
}

int apply_launch(const NetLayout& L, const ApplyArgs& a0, hipStream_t s, const GatherArgs* pregather) {
  ApplyArgs a = a0;
  apply_args_finish(&a);
  if (pregather) {
    const GatherArgs& g = *pregather;
    const size_t lds = frame_gather_lds_bytes(g, 4);
    if (lds > 48 * 1024 || !g.state_h || g.state || (g.emb % 4) || g.rows <= 0) {
      recnn_set_error("apply+gather: needs the bf16-only gather with a tile that fits 48 KB of LDS");
      return RECNN_E_UNSUPPORTED;
    }
    const int ng = (g.rows + 3) / 4;
    hipLaunchKernelGGL(apply_gather_kernel, dim3(L.nblk + ng), dim3(256), lds, s, L, a, g, ng);
    return recnn_check_hip(hipGetLastError(), "apply_gather_kernel");
  }
  hipLaunchKernelGGL(apply_kernel, dim3(L.nblk), dim3(256), 0, s, L, a);
  return recnn_check_hip(hipGetLastError(), "apply_kernel");
}

// in-place clip for callers that run their own optimizer: g *= grad_scale * coef
__global__ __launch_bounds__(256) void scale_grads_kernel(const NetLayout L, float* __restrict__ g, const float* l1part, int n_l1,
                                                          float grad_scale) {
  __shared__ float red[4];
  float gs = grad_scale;
  if (n_l1 > 0) gs *= clip_coef(l1part, n_l1, grad_scale, red);
  const int b = blockIdx.x;
  const TensorSeg& T = L.t[find_tensor(L, b)];
  const Own o = own_elems(T, b - T.blk0);
  if (o.cnt == 0) return;
  float x[4];
  load_own(g + T.p_off + o.e, o, x);
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] *= gs;
  store_own(g + T.p_off + o.e, o, x);
}

int scale_grads_launch(const NetLayout& L, float* gflat, const float* l1part, int n_l1, float grad_scale, hipStream_t s) {
  hipLaunchKernelGGL(scale_grads_kernel, dim3(L.nblk), dim3(256), 0, s, L, gflat, l1part, n_l1, grad_scale);
  return recnn_check_hip(hipGetLastError(), "scale_grads_kernel");
}

// ---------------------------------------------------------------- flat entry points (C ABI section 3)
__global__ __launch_bounds__(256) void soft_update_flat_kernel(float* __restrict__ t, const float* __restrict__ p, int64_t n,
                                                               float tau) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    t[i] = t[i] * (1.0f - tau) + p[i] * tau;
}
extern "C" int recnn_soft_update_flat(float* target, const float* net, int64_t n, float tau, void* stream) {
  RECNN_REQUIRE(target && net && n >= 0, "soft_update_flat: bad arguments");
  if (n == 0) return 0;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(soft_update_flat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, target, net, n, tau);
  return recnn_check_hip(hipGetLastError(), "soft_update_flat");
}

// A row-padded copy of a 2-D parameter in another leading dimension / type (recnn_shadow_out: what the GEMM kernels read in place of
// a [rows, cols] weight whose rows are not 16-byte aligned or not in the compute type), rewritten by the optimizer pass that has the
// new value in a register anyway: element i of the flat array is (i / cols, i % cols) of the copy.
struct ShadowDst { void* dst; int cols; int64_t ld; int bf16; int quad; };
__device__ __forceinline__ void shadow_put(const ShadowDst& sh, int64_t i, float v) {
  const int64_t r = i / sh.cols;
  const int c = (int)(i - r * sh.cols);
  if (sh.bf16) ((bf16_t*)sh.dst)[r * sh.ld + c] = f2bf(v);
  else ((float*)sh.dst)[r * sh.ld + c] = v;
}
// Four consecutive flat elements i .. i + 3 (i a multiple of 4) with ONE division.  sh.quad (shadow_arg): 4 = the quad lies in one row and
// its destination is 8- / 16-byte aligned (cols and ld multiples of 4): one store; 2 = cols and ld even: a PAIR never straddles a row and is
// 4- / 8-byte aligned (the catalogue-sized weights of REINFORCE: [2048, 101290], [100000, 1290]): two stores; 1 = element by element.
__device__ __forceinline__ void shadow_put4(const ShadowDst& sh, int64_t i, const float (&x)[4]) {
  int64_t r;
  if (i < (int64_t)0x7fffffff) r = (uint32_t)i / (uint32_t)sh.cols;   // (a 64-bit division is ~100 instructions)
  else r = i / sh.cols;
  int c = (int)(i - r * sh.cols);
  if (sh.quad == 4) {
    if (sh.bf16) *(uint2*)((bf16_t*)sh.dst + r * sh.ld + c) = make_uint2(pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]));
    else *(float4*)((float*)sh.dst + r * sh.ld + c) = make_float4(x[0], x[1], x[2], x[3]);
  } else if (sh.quad == 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (sh.bf16) *(uint32_t*)((bf16_t*)sh.dst + r * sh.ld + c) = pack_bf2(x[2 * h], x[2 * h + 1]);
      else *(float2*)((float*)sh.dst + r * sh.ld + c) = make_float2(x[2 * h], x[2 * h + 1]);
      c += 2;
      if (c >= sh.cols) { c = 0; ++r; }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (sh.bf16) ((bf16_t*)sh.dst)[r * sh.ld + c] = f2bf(x[j]);
      else ((float*)sh.dst)[r * sh.ld + c] = x[j];
      if (++c == sh.cols) { c = 0; ++r; }
    }
  }
}
// whether the flat passes may take 16 bytes per lane and instruction (the flat arrays 16-byte aligned; the copy is handled by sh.quad)
static bool flat_vec4(std::initializer_list<const void*> ptrs) {
  for (const void* q : ptrs)
    if ((uintptr_t)q & 15) return false;
  return true;
}
static int shadow_arg(const recnn_shadow_out* h, int64_t n, ShadowDst* out) {
  out->dst = nullptr; out->cols = 1; out->ld = 0; out->bf16 = 0; out->quad = 1;
  if (!h || !h->dst) return 0;
  RECNN_REQUIRE(h->cols > 0 && h->ld >= h->cols && n % h->cols == 0, "optimizer shadow: the flat length %lld is not rows x cols = . x %d (ld %lld)",
                (long long)n, h->cols, (long long)h->ld);
  out->dst = h->dst; out->cols = h->cols; out->ld = h->ld; out->bf16 = h->bf16 ? 1 : 0;
  const uintptr_t d = (uintptr_t)h->dst;
  const int esz = out->bf16 ? 2 : 4;
  if (!(h->cols & 3) && !(h->ld & 3) && !(d & (uintptr_t)(4 * esz - 1))) out->quad = 4;
  else if (!(h->cols & 1) && !(h->ld & 1) && !(d & (uintptr_t)(2 * esz - 1))) out->quad = 2;
  else out->quad = 1;
  return 0;
}

// One element of the flat Adam pass (torch.optim.Adam's arithmetic, fp contraction off: every instantiation rounds alike).
__device__ __forceinline__ void adam_elem(float& pi, float& mi, float& vi, float graw, float beta2, float eps, float wd, float step_size,
                                          float bc2_sqrt, float gs, float omb1, float omb2) {
#pragma clang fp contract(off)
  float gi = graw * gs;
  if (wd != 0.f) gi += wd * pi;
  mi += omb1 * (gi - mi);
  vi = beta2 * vi + omb2 * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= step_size * (mi / denom);
}
// VEC = 4: 16 bytes per lane and load (catalogue-sized tensors -- REINFORCE at 100k items -- are bound by bytes in flight, not by HBM, with
// 4-byte lanes); the n % 4 tail and unaligned callers take the scalar form.  Element by element the same arithmetic.
template <bool SH, int VEC>
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, float lr, float beta1, float beta2,
                                                        float eps, float wd, float step_size, float bc2_sqrt, float gs, float omb1,
                                                        float omb2, const ShadowDst sh) {
#pragma clang fp contract(off)      // (both instantiations must round alike: which products fuse into FMAs is the compiler's choice per body)
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  int64_t done = 0;
  if (VEC == 4) {
    const int64_t n4 = n >> 2;
    for (int64_t q = gid; q < n4; q += stride) {
      const float4 P = ((const float4*)p)[q], G = ((const float4*)g)[q], M = ((const float4*)m)[q], V = ((const float4*)v)[q];
      float x[4] = {P.x, P.y, P.z, P.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
      const float gg[4] = {G.x, G.y, G.z, G.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_elem(x[j], mm[j], vv[j], gg[j], beta2, eps, wd, step_size, bc2_sqrt, gs, omb1, omb2);
      ((float4*)p)[q] = make_float4(x[0], x[1], x[2], x[3]);
      ((float4*)m)[q] = make_float4(mm[0], mm[1], mm[2], mm[3]);
      ((float4*)v)[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (SH) shadow_put4(sh, q << 2, x);
    }
    done = n4 << 2;
  }
  for (int64_t i = done + gid; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_elem(pi, mi, vi, g[i], beta2, eps, wd, step_size, bc2_sqrt, gs, omb1, omb2);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (SH) shadow_put(sh, i, pi);
  }
}
extern "C" int recnn_adam_flat_shadow(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                      float eps, float weight_decay, int step_t, float grad_scale, const recnn_shadow_out* h_shadow,
                                      void* stream) {
  RECNN_REQUIRE(p && g && m && v && n >= 0 && step_t >= 1, "adam_flat: bad arguments");
  if (n == 0) return 0;
  ShadowDst sh;
  int rc = shadow_arg(h_shadow, n, &sh);
  if (rc) return rc;
  const double b1 = recnn_snap7(beta1), b2 = recnn_snap7(beta2);
  const double bc1 = 1.0 - pow(b1, (double)step_t), bc2 = 1.0 - pow(b2, (double)step_t);
  const bool v4 = flat_vec4({p, g, m, v});
  int grid = (int)(((v4 ? (n + 3) / 4 : n) + 255) / 256);
  if (grid > 2048) grid = 2048;
#define ADAM_FLAT_GO(SH, VEC)                                                                                                             \
  hipLaunchKernelGGL((adam_flat_kernel<SH, VEC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,  \
                     weight_decay, (float)((double)lr / bc1), (float)sqrt(bc2), grad_scale, (float)(1.0 - b1), (float)(1.0 - b2), sh)
  if (sh.dst) { if (v4) ADAM_FLAT_GO(true, 4); else ADAM_FLAT_GO(true, 1); }
  else { if (v4) ADAM_FLAT_GO(false, 4); else ADAM_FLAT_GO(false, 1); }
#undef ADAM_FLAT_GO
  return recnn_check_hip(hipGetLastError(), "adam_flat");
}
extern "C" int recnn_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step_t, float grad_scale, void* stream) {
  return recnn_adam_flat_shadow(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step_t, grad_scale, nullptr, stream);
}

// The same step with the step count on the DEVICE (t = *t_ptr + t_add): a captured graph replays it with the count the graph
// itself advances (recnn_amd.optim.Adam(capturable=True)); the bias corrections are evaluated per thread, in double like the host.
__global__ __launch_bounds__(256) void adam_flat_at_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, int64_t n, float lr, float beta2, float eps, float wd,
                                                           double b1, double b2, const int32_t* __restrict__ t_ptr, int t_add, float gs,
                                                           float omb1, float omb2) {
#pragma clang fp contract(off)      // the same roundings as adam_flat_kernel: a captured step and an eager one stay bit-identical
  const double t = (double)(*t_ptr + t_add);
  const float step_size = (float)((double)lr / (1.0 - pow(b1, t)));
  const float bc2_sqrt = (float)sqrt(1.0 - pow(b2, t));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pi = p[i];
    float gi = g[i] * gs;
    if (wd != 0.f) gi += wd * pi;
    float mi = m[i], vi = v[i];
    mi += omb1 * (gi - mi);
    vi = beta2 * vi + omb2 * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
extern "C" int recnn_adam_flat_at(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, const int32_t* step_dev, int step_add, float grad_scale, void* stream) {
  RECNN_REQUIRE(p && g && m && v && n >= 0 && step_dev, "adam_flat_at: bad arguments");
  if (n == 0) return 0;
  const double b1 = recnn_snap7(beta1), b2 = recnn_snap7(beta2);
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(adam_flat_at_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta2, eps, weight_decay, b1,
                     b2, step_dev, step_add, grad_scale, (float)(1.0 - b1), (float)(1.0 - b2));
  return recnn_check_hip(hipGetLastError(), "adam_flat_at");
}

__global__ __launch_bounds__(256) void l1_part_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += fabsf(g[i]);
  float tot = block_sum256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  float tot = block_sum256(s, red);
  if (threadIdx.x == 0) out[0] = tot;
}
extern "C" int recnn_l1_norm_flat(const float* g, int64_t n, float* scratch, float* out, void* stream) {
  RECNN_REQUIRE(g && scratch && out && n >= 0, "l1_norm_flat: bad arguments");
  int grid = (int)((n + 255) / 256);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(l1_part_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, scratch);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, grid, out);
  return recnn_check_hip(hipGetLastError(), "l1_norm_flat");
}

// One element of the flat Ranger pass (RAdam + Lookahead: torch_optimizer.Ranger, the third-party optimizer recnn/nn/algo.py:84-90 constructs;
// its published step restated in oracle/), fp contraction off.
__device__ __forceinline__ void ranger_elem(float& pi, float& mi, float& vi, float graw, float lr, float beta1, float beta2, float eps, float wd,
                                            int rect, float sl_lr, float gs, float omb1, float omb2) {
#pragma clang fp contract(off)
  const float gi = graw * gs;
  vi = beta2 * vi + omb2 * gi * gi;
  mi = beta1 * mi + omb1 * gi;
  if (wd != 0.f) pi += (-wd * lr) * pi;
  if (rect) pi += -sl_lr * (mi / (sqrtf(vi) + eps));
  else pi += -sl_lr * mi;
}
template <bool SH, int VEC>
__global__ __launch_bounds__(256) void ranger_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, float* __restrict__ slow, int64_t n, float lr,
                                                          float beta1, float beta2, float eps, float wd, float la_alpha, int la_sync,
                                                          int rect, float step, float gs, float omb1, float omb2, const ShadowDst sh) {
#pragma clang fp contract(off)
  const float sl_lr = step * lr;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  int64_t done = 0;
  if (VEC == 4) {      // (see adam_flat_kernel)
    const int64_t n4 = n >> 2;
    for (int64_t q = gid; q < n4; q += stride) {
      const float4 P = ((const float4*)p)[q], G = ((const float4*)g)[q], M = ((const float4*)m)[q], V = ((const float4*)v)[q];
      float x[4] = {P.x, P.y, P.z, P.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
      const float gg[4] = {G.x, G.y, G.z, G.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) ranger_elem(x[j], mm[j], vv[j], gg[j], lr, beta1, beta2, eps, wd, rect, sl_lr, gs, omb1, omb2);
      if (la_sync) {
        const float4 S = ((const float4*)slow)[q];
        float ss[4] = {S.x, S.y, S.z, S.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { ss[j] += la_alpha * (x[j] - ss[j]); x[j] = ss[j]; }
        ((float4*)slow)[q] = make_float4(ss[0], ss[1], ss[2], ss[3]);
      }
      ((float4*)p)[q] = make_float4(x[0], x[1], x[2], x[3]);
      ((float4*)m)[q] = make_float4(mm[0], mm[1], mm[2], mm[3]);
      ((float4*)v)[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (SH) shadow_put4(sh, q << 2, x);
    }
    done = n4 << 2;
  }
  for (int64_t i = done + gid; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    ranger_elem(pi, mi, vi, g[i], lr, beta1, beta2, eps, wd, rect, sl_lr, gs, omb1, omb2);
    if (la_sync) {
      float si = slow[i];
      si += la_alpha * (pi - si);
      pi = si;
      slow[i] = si;
    }
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (SH) shadow_put(sh, i, pi);
  }
}
extern "C" int recnn_ranger_flat_shadow(float* p, const float* g, float* m, float* v, float* slow, int64_t n, float lr, float beta1,
                                        float beta2, float eps, float weight_decay, float la_alpha, int la_k, float nsma_threshold,
                                        int step_t, float grad_scale, const recnn_shadow_out* h_shadow, void* stream) {
  RECNN_REQUIRE(p && g && m && v && slow && n >= 0 && step_t >= 1, "ranger_flat: bad arguments");
  if (n == 0) return 0;
  ShadowDst sh;
  int rc = shadow_arg(h_shadow, n, &sh);
  if (rc) return rc;
  const double b1 = recnn_snap7(beta1), b2 = recnn_snap7(beta2);
  const RadamScalars rs = radam_scalars(step_t, log(b1), log(b2), b2, (double)nsma_threshold);
  const bool v4 = flat_vec4({p, g, m, v, slow});
  int grid = (int)(((v4 ? (n + 3) / 4 : n) + 255) / 256);
  if (grid > 2048) grid = 2048;     // (measured: 4096 / 8192 workgroups and non-temporal loads / stores are all 4-10 % slower)
  const int sync = (la_k > 0 && step_t % la_k == 0) ? 1 : 0;
#define RANGER_FLAT_GO(SH, VEC)                                                                                                             \
  hipLaunchKernelGGL((ranger_flat_kernel<SH, VEC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, slow, n, lr, beta1, beta2, \
                     eps, weight_decay, la_alpha, sync, rs.rect, rs.step, grad_scale, (float)(1.0 - b1), (float)(1.0 - b2), sh)
  if (sh.dst) { if (v4) RANGER_FLAT_GO(true, 4); else RANGER_FLAT_GO(true, 1); }
  else { if (v4) RANGER_FLAT_GO(false, 4); else RANGER_FLAT_GO(false, 1); }
#undef RANGER_FLAT_GO
  return recnn_check_hip(hipGetLastError(), "ranger_flat");
}
extern "C" int recnn_ranger_flat(float* p, const float* g, float* m, float* v, float* slow, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, float la_alpha, int la_k, float nsma_threshold,
                                 int step_t, float grad_scale, void* stream) {
  return recnn_ranger_flat_shadow(p, g, m, v, slow, n, lr, beta1, beta2, eps, weight_decay, la_alpha, la_k, nsma_threshold, step_t,
                                  grad_scale, nullptr, stream);
}
