// head.hip -- the critic's N=1 output layer, TD target, losses and their backward seed (gfx950).
//
// Replaces:
//   recnn/nn/models.py:212      value = linear3(h2)            (row dot, not an MFMA shape)
//   recnn/nn/update/misc.py:6-7 temporal_difference
//   recnn/nn/update/misc.py:33-39  clamp + mean((V - y)^2)
//   recnn/nn/update/td3.py:83-93   min of twin targets, MSELoss x2
//   recnn/nn/update/ddpg.py:79,87 / td3.py:117-127   policy_loss = -Q.mean()
// and the first step of autograd's backward through linear3 of the critic.
#include "head_dev.h"

template <class TC, int NT, int NC, bool PRE> __global__ __launch_bounds__(256) void head_kernel(const HeadArgs a) {
  __shared__ HeadSmem sm;
  head_block<TC, NT, NC, PRE>(a, (int)blockIdx.x, (int)threadIdx.x, true, sm);
}

int head_launch(const HeadArgs& a, hipStream_t s) {
  if (a.rows <= 0) return 0;
  if (a.H % 8 || a.ld_h % 8) { recnn_set_error("head: hidden size and pitch must be multiples of 8"); return RECNN_E_INVALID; }
  if (a.tc_bf16 == 2 && a.ld_h < x3_ld(a.H)) { recnn_set_error("head (bf16x3): pitch below the split row width"); return RECNN_E_INVALID; }
  dim3 grid((a.rows + HEAD_ROWS_PER_BLOCK - 1) / HEAD_ROWS_PER_BLOCK), block(256);
  if (a.n_target < 0 || a.n_target > 2 || a.n_critic < 1 || a.n_critic > HEAD_MAX_CRITIC) {
    recnn_set_error("head: n_target must be 0..2 and n_critic 1..2");
    return RECNN_E_INVALID;
  }
  const bool pre = a.n_target > 0 && a.tq_in[0] != nullptr;
  if (pre && a.n_target > 1 && !a.tq_in[1]) { recnn_set_error("head: tq_in[1] missing"); return RECNN_E_INVALID; }
#define HEAD_GO(TC, NT, NC) do { if (pre) hipLaunchKernelGGL((head_kernel<TC, NT, NC, true>), grid, block, 0, s, a); \
                                 else hipLaunchKernelGGL((head_kernel<TC, NT, NC, false>), grid, block, 0, s, a); } while (0)
#define HEAD_TC(NT, NC) do { if (a.tc_bf16 == 2) HEAD_GO(x3_t, NT, NC); else if (a.tc_bf16) HEAD_GO(bf16_t, NT, NC); else HEAD_GO(float, NT, NC); } while (0)
  switch (a.n_target * 2 + (a.n_critic - 1)) {
    case 0: HEAD_TC(0, 1); break;
    case 1: HEAD_TC(0, 2); break;
    case 2: HEAD_TC(1, 1); break;
    case 3: HEAD_TC(1, 2); break;
    case 4: HEAD_TC(2, 1); break;
    default: HEAD_TC(2, 2); break;
  }
#undef HEAD_TC
#undef HEAD_GO
  return recnn_check_hip(hipGetLastError(), "head_kernel");
}

// ---------------------------------------------------------------- loss finalize (+ step tick)
// losses[c] = scale[c] * sum(part[c][0..n)) ; then the device counters advance.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const LossFinalizeArgs a) {
  // All 256 threads stride over every loss's partial sums (loads of all losses issued before the first reduction);
  // the counter updates ride on separate lanes of the last wave so that every global read-modify-write of this
  // single-workgroup, latency-only kernel is in flight at the same time.
  __shared__ float red[4][4];
  __shared__ int s_old;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = 255 - (int)threadIdx.x;  // 0..n_tick-1: tick counters, n_tick: sampler cursor
  int32_t* cnt = nullptr;
  int32_t cv = 0;
  if (slot < a.n_tick) cnt = a.tick[slot];
  else if (slot == a.n_tick) cnt = a.wrap_ptr;
  if (cnt) cv = *cnt;
  if (slot == 0) s_old = cv;   // tick[0] is the step counter
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < a.n) {
      const float* __restrict__ part = a.part[c];
      const int n = a.n_part[c];
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int i = threadIdx.x;
      for (; i + 768 < n; i += 1024) {
        s0 += part[i]; s1 += part[i + 256]; s2 += part[i + 512]; s3 += part[i + 768];
      }
      for (; i < n; i += 256) s0 += part[i];
      acc[c] = (s0 + s1) + (s2 + s3);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < a.n) {
      const float t = wave_sum(acc[c]);
      if (lane == 0) red[c][wave] = t;
    }
  }
  if (cnt) {
    if (slot == a.n_tick) {
      cv += a.wrap_inc;
      if (cv >= a.wrap_mod) cv -= a.wrap_mod;  // wrap_inc <= wrap_mod
    } else {
      cv += a.tick_inc[slot];
    }
    *cnt = cv;
  }
  __syncthreads();
  if ((int)threadIdx.x < a.n) {
    const int c = threadIdx.x;
    const float s = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
    const float v = s * a.scale[c] + (a.add_ptr[c] ? a.add_scale[c] * a.add_ptr[c][0] : 0.f);
    a.out[c] = v;
    if (a.ring && a.n_tick > 0) a.ring[(int64_t)((s_old + a.tick_inc[0] - 1) & a.ring_mask) * 4 + c] = v;
  }
}

__global__ __launch_bounds__(256) void loss_history_kernel(const LossHistoryArgs a) {
  __shared__ float red[4][4];
  const int j = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < a.n) {
      const float* __restrict__ part = a.part[c] + (int64_t)j * a.stride[c];
      const int n = c == a.n - 1 ? a.pol_count[j] : a.n_part[c];
      float s0 = 0.f;
      for (int i = threadIdx.x; i < n; i += 256) s0 += part[i];
      acc[c] = s0;
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < a.n) {
      const float t = wave_sum(acc[c]);
      if (lane == 0) red[c][wave] = t;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < a.n) {
    const int c = threadIdx.x;
    float v = ((red[c][0] + red[c][1]) + (red[c][2] + red[c][3])) * a.scale[c];
    if (c == a.n - 1 && a.pol_add[j]) v -= a.b3[0];
    a.ring[(int64_t)((*a.step_ctr + j) & a.ring_mask) * 4 + c] = v;
  }
}

int loss_history_launch(const LossHistoryArgs& a, hipStream_t s) {
  if (a.n_steps <= 0) return 0;
  hipLaunchKernelGGL(loss_history_kernel, dim3(a.n_steps), dim3(256), 0, s, a);
  return recnn_check_hip(hipGetLastError(), "loss_history_kernel");
}

int loss_finalize_launch(const LossFinalizeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, a);
  return recnn_check_hip(hipGetLastError(), "loss_finalize_kernel");
}

// ---------------------------------------------------------------- TD3 target-policy noise (perf mode)
// out[i] ~ N(0, stddev^2), counter-based (Box-Muller over two hashed uniforms), keyed by (seed, step).
// The parity tests bypass it with the reference's own CPU draw (td3.py:74).
__global__ __launch_bounds__(256) void noise_fill_kernel(float* __restrict__ out, int64_t n, float stddev, uint32_t seed,
                                                         const int32_t* __restrict__ step_ptr, int step_add) {
  const uint32_t key = mask_key(seed, (step_ptr ? *step_ptr : 0) + step_add, 0xA511CEu);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t a = mix32((uint32_t)i * 0x9E3779B1u + key);
    const uint32_t b = mix32(a ^ 0x68E31DA4u ^ (uint32_t)(i >> 32));
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);           // [0, 1)
    out[i] = stddev * sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
  }
}
int noise_fill_launch(float* out, int64_t n, float stddev, uint32_t seed, const int32_t* step_ptr, int step_add, hipStream_t s) {
  if (n <= 0) return 0;
  int grid = (int)((n + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(noise_fill_kernel, dim3(grid), dim3(256), 0, s, out, n, stddev, seed, step_ptr, step_add);
  return recnn_check_hip(hipGetLastError(), "noise_fill_kernel");
}
