// optim.hip -- gradient reduction, the clip quirk, Adam, soft target update, shadow refresh (gfx950).
//
// Replaces:
//   torch.optim.Adam.step  as injected by the reference's users   recnn/nn/update/misc.py:44, ddpg.py:93,
//                                                                  td3.py:97,101,134
//   torch.nn.utils.clip_grad_norm_(policy.parameters(), -1, 1)     ddpg.py:92, td3.py:133
//   recnn/utils/misc.py:1-5  soft_update   (target*(1-tau) + param*tau, that operand order)
// One workgroup per parameter-matrix row over a flat fp32 arena; the same pass writes the compute-type
// "shadow" copy of the weights (zero-padded, 16-byte aligned rows; critic W1 columns rotated to the
// packed [action | state] batch layout) that the MFMA GEMMs read, and optionally the soft-updated target.
#include "optim.h"

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

// fixed summation order (deterministic); 8 independent loads in flight per thread
__device__ inline float sum_slabs(const TensorSeg& T, int64_t e) {
  float g = 0.f;
  int s = 0;
  for (; s + 32 <= T.nslab; s += 32) {  // many small slabs (head / column-sum partials): 32 loads in flight
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = T.gpart[(int64_t)(s + j) * T.slab_stride + e];
#pragma unroll
    for (int w = 16; w > 0; w >>= 1)
#pragma unroll
      for (int j = 0; j < w; ++j) v[j] += v[j + w];
    g += v[0];
  }
  for (; s + 8 <= T.nslab; s += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = T.gpart[(int64_t)(s + j) * T.slab_stride + e];
    g += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (; s < T.nslab; ++s) g += T.gpart[(int64_t)s * T.slab_stride + e];
  return g;
}

// g_flat = sum of partial slabs (split-K slabs of the dW GEMMs, row-tile column sums for biases)
__global__ __launch_bounds__(256) void grad_reduce_kernel(const NetLayout L, float* __restrict__ gflat,
                                                          float* __restrict__ l1part) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const TensorSeg& T = L.t[find_tensor(L, b)];
  const int row = b - T.blk0;
  float l1 = 0.f;
  for (int c = threadIdx.x; c < T.cols; c += 256) {
    const int64_t e = (int64_t)row * T.cols + c;
    const float g = sum_slabs(T, e);
    gflat[T.p_off + e] = g;
    l1 += fabsf(g);
  }
  if (l1part) {
    float tot = block_sum256(l1, red);
    if (threadIdx.x == 0) l1part[b] = tot;
  }
}

int grad_reduce_launch(const NetLayout& L, float* gflat, float* l1part, hipStream_t s) {
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(L.nblk), dim3(256), 0, s, L, gflat, l1part);
  return recnn_check_hip(hipGetLastError(), "grad_reduce_kernel");
}

__device__ inline float clip_coef(const float* l1part, int n, float grad_scale, float* red) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += l1part[i];
  const float total = block_sum256(s, red) * grad_scale;
  // clip_grad_norm_(max_norm=-1, norm_type=1): coef = -1/(total+1e-6), clamped to <= 1
  return fminf(-1.0f / (total + 1e-6f), 1.0f);
}

__global__ __launch_bounds__(256) void apply_kernel(const NetLayout L, const ApplyArgs a) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const TensorSeg& T = L.t[find_tensor(L, b)];
  const int row = b - T.blk0;

  float gs = a.grad_scale;
  if (a.n_l1 > 0) {
    const float coef = clip_coef(a.l1part, a.n_l1, a.grad_scale, red);
    if (a.coef_out && b == 0 && threadIdx.x == 0) a.coef_out[0] = coef;
    gs *= coef;
  }
  float step_size = 0.f, bc2_sqrt = 1.f;
  if (a.do_adam) {
    const int t = *a.t_ptr + 1;
    const double bc1 = 1.0 - pow((double)a.beta1, (double)t);
    const double bc2 = 1.0 - pow((double)a.beta2, (double)t);
    step_size = (float)((double)a.lr / bc1);
    bc2_sqrt = (float)sqrt(bc2);
  }
  for (int c = threadIdx.x; c < T.cols; c += 256) {
    const int64_t e = T.p_off + (int64_t)row * T.cols + c;
    float p = a.p[e];
    if (a.do_adam) {
      float graw;
      if (a.from_slabs) {
        graw = sum_slabs(T, (int64_t)row * T.cols + c);
        if (a.g_out) a.g_out[e] = graw;
      } else {
        graw = a.g[e];
      }
      float g = graw * gs;
      if (a.weight_decay != 0.f) g += a.weight_decay * p;
      float m = a.m[e], v = a.v[e];
      m += (1.0f - a.beta1) * (g - m);
      v = a.beta2 * v + (1.0f - a.beta2) * g * g;
      const float denom = sqrtf(v) / bc2_sqrt + a.eps;
      p -= step_size * (m / denom);
      a.m[e] = m;
      a.v[e] = v;
      a.p[e] = p;
    }
    int64_t se = -1;
    if (T.sh_off >= 0) {
      int cc = c + T.col_rot;
      if (cc >= T.cols) cc -= T.cols;
      se = T.sh_off + (int64_t)row * T.sh_ld + cc;
      if (a.shadow) {
        if (a.tc_bf16) ((bf16_t*)a.shadow)[se] = f2bf(p);
        else ((float*)a.shadow)[se] = p;
      }
    }
    if (a.tgt_p) {
      const float tp = a.tgt_p[e] * (1.0f - a.tau) + p * a.tau;  // utils/misc.py:3-5 operand order
      a.tgt_p[e] = tp;
      if (se >= 0 && a.tgt_shadow) {
        if (a.tc_bf16) ((bf16_t*)a.tgt_shadow)[se] = f2bf(tp);
        else ((float*)a.tgt_shadow)[se] = tp;
      }
    }
  }
}

int apply_launch(const NetLayout& L, const ApplyArgs& a, hipStream_t s) {
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
  const int row = b - T.blk0;
  for (int c = threadIdx.x; c < T.cols; c += 256) g[T.p_off + (int64_t)row * T.cols + c] *= gs;
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

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, float lr, float beta1, float beta2,
                                                        float eps, float wd, float step_size, float bc2_sqrt, float gs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pi = p[i];
    float gi = g[i] * gs;
    if (wd != 0.f) gi += wd * pi;
    float mi = m[i], vi = v[i];
    mi += (1.0f - beta1) * (gi - mi);
    vi = beta2 * vi + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
extern "C" int recnn_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step_t, float grad_scale, void* stream) {
  RECNN_REQUIRE(p && g && m && v && n >= 0 && step_t >= 1, "adam_flat: bad arguments");
  if (n == 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step_t), bc2 = 1.0 - pow((double)beta2, (double)step_t);
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(adam_flat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, (float)((double)lr / bc1), (float)sqrt(bc2), grad_scale);
  return recnn_check_hip(hipGetLastError(), "adam_flat");
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
