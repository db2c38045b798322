// optim_dev.h -- device-side pieces of the flat optimizer pass shared by optim.hip (apply_kernel) and dwadam.hip (the weight-gradient
// GEMM with the optimizer in its epilogue): workgroup -> element map, fixed-order slab sums, the clip coefficient, the data-parallel
// exchange and apply_body itself.  Include after optim.h; device code only.
#pragma once
#include "optim.h"
#include "x3.h"
#include "comm_dev.h"

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
      // slab s of this element; half slabs (mlpt.hip's 16-row panels, T.pair): a 32-row panel's sum is (its first half) + (its second
      // half) = (c0 + c1) + (c2 + c3), what the 32-row panel kernels write -- then the slabs combine exactly as the whole ones do
      const float* gp = T.gpart + e;
      const int64_t st = T.slab_stride;
      const bool pair = T.pair != 0;
      auto ld = [&](int k) -> float { return pair ? gp[(int64_t)(2 * k) * st] + gp[(int64_t)(2 * k + 1) * st] : gp[(int64_t)k * st]; };
      for (; s + 32 <= s_end; s += 32) {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = ld(s + j);
#pragma unroll
        for (int w = 16; w > 0; w >>= 1)
#pragma unroll
          for (int j = 0; j < w; ++j) v[j] += v[j + w];
        acc += v[0];
      }
      for (; s + 8 <= s_end; s += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ld(s + j);
        acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
      for (; s < s_end; ++s) acc += ld(s);
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

