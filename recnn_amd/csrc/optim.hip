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

#include "optim_dev.h"

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
