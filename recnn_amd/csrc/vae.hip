// vae.hip -- the latent layer and the loss of BCQ's conditional VAE.
//
// Replaces, in recnn/nn/models.py:265-282 (bcqGenerator.forward) and recnn/nn/update/bcq.py:78-81, the ATen chain
//   clamp -> exp -> mul -> add            (z = mean + exp(clamp(log_std, -4, 15)) * eps)
//   mse_loss, pow, log, sub, mean, add    (loss = mean((u - a)^2) + 0.5 * -0.5 * mean(1 + log(std^2) - mean^2 - std^2))
// and their autograd backward with four small kernels over [rows, latent] / [rows, action] panels.  The GEMMs either side
// (encoder e1/e2/[mean | log_std], decoder d1/d2/d3) are gemm.hip's.  Rows are independent, so the kernels are plain
// grid-stride element loops; the two loss sums are reduced in a fixed order (per-workgroup partials, then one wave), which
// keeps the loss bit-reproducible from run to run.
#include "common.h"

namespace {

constexpr int VT = 256;          // threads per workgroup
constexpr int MAX_PARTS = 256;   // workgroups of the loss kernel == partial sums

__device__ inline float clamp_log_std(float r) { return fminf(fmaxf(r, RECNN_VAE_LOG_STD_MIN), RECNN_VAE_LOG_STD_MAX); }

// z = mean + std * eps, std = exp(clamp(raw)).  ml: [rows, ld_ml] = [mean | raw]
__global__ __launch_bounds__(VT) void vae_latent_fwd_kernel(const float* __restrict__ ml, int64_t ld_ml, const float* __restrict__ eps,
                                                            int64_t ld_eps, int rows, int L, float* __restrict__ z, int64_t ldz,
                                                            float* __restrict__ std_out, int64_t ld_std) {
  const int64_t total = (int64_t)rows * L;
  for (int64_t i = (int64_t)blockIdx.x * VT + threadIdx.x; i < total; i += (int64_t)gridDim.x * VT) {
    const int r = (int)(i / L), c = (int)(i - (int64_t)r * L);
    const float mean = ml[r * ld_ml + c];
    const float sd = expf(clamp_log_std(ml[r * ld_ml + L + c]));
    z[r * ldz + c] = mean + sd * eps[r * ld_eps + c];
    std_out[r * ld_std + c] = sd;
  }
}

// d[mean | raw] from dz (through z), dmean (on the mean output) and dstd (on the std output); any of them may be NULL.
// clamp passes the gradient on the closed interval [min, max] (ATen clamp_backward).
__global__ __launch_bounds__(VT) void vae_latent_bwd_kernel(const float* __restrict__ ml, int64_t ld_ml, const float* __restrict__ eps,
                                                            int64_t ld_eps, const float* __restrict__ sd, int64_t ld_std,
                                                            const float* __restrict__ dz, int64_t ld_dz,
                                                            const float* __restrict__ dmean, int64_t ld_dmean,
                                                            const float* __restrict__ dstd, int64_t ld_dstd, int rows, int L,
                                                            float* __restrict__ dml, int64_t ld_dml) {
  const int64_t total = (int64_t)rows * L;
  for (int64_t i = (int64_t)blockIdx.x * VT + threadIdx.x; i < total; i += (int64_t)gridDim.x * VT) {
    const int r = (int)(i / L), c = (int)(i - (int64_t)r * L);
    const float g = dz ? dz[r * ld_dz + c] : 0.f;
    float gm = g, gs = g * eps[r * ld_eps + c];
    if (dmean) gm += dmean[r * ld_dmean + c];
    if (dstd) gs += dstd[r * ld_dstd + c];
    const float raw = ml[r * ld_ml + L + c];
    const bool inside = raw >= RECNN_VAE_LOG_STD_MIN && raw <= RECNN_VAE_LOG_STD_MAX;
    dml[r * ld_dml + c] = gm;
    dml[r * ld_dml + L + c] = inside ? gs * sd[r * ld_std + c] : 0.f;
  }
}

__device__ inline float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < VT / WAVE; ++i) t += red[i];
  __syncthreads();
  return t;
}

// parts[b] = (sum (u - a)^2, sum (1 + log(std^2) - mean^2 - std^2)) over workgroup b's elements
__global__ __launch_bounds__(VT) void vae_loss_part_kernel(const float* __restrict__ u, int64_t ldu, const float* __restrict__ a, int64_t lda,
                                                           const float* __restrict__ mean, int64_t ldm, const float* __restrict__ sd,
                                                           int64_t lds_, int rows, int A, int L, float* __restrict__ parts) {
  __shared__ float red[VT / WAVE];
  float se = 0.f, kl = 0.f;
  const int64_t na = (int64_t)rows * A, nl = (int64_t)rows * L;
  for (int64_t i = (int64_t)blockIdx.x * VT + threadIdx.x; i < na; i += (int64_t)gridDim.x * VT) {
    const int r = (int)(i / A), c = (int)(i - (int64_t)r * A);
    const float d = u[r * ldu + c] - a[r * lda + c];
    se += d * d;
  }
  for (int64_t i = (int64_t)blockIdx.x * VT + threadIdx.x; i < nl; i += (int64_t)gridDim.x * VT) {
    const int r = (int)(i / L), c = (int)(i - (int64_t)r * L);
    const float m = mean[r * ldm + c], s = sd[r * lds_ + c];
    kl += 1.f + logf(s * s) - m * m - s * s;
  }
  se = block_sum(se, red);
  kl = block_sum(kl, red);
  if (threadIdx.x == 0) {
    parts[2 * blockIdx.x] = se;
    parts[2 * blockIdx.x + 1] = kl;
  }
}

// out = (recon_loss, kl_loss, recon_loss + kl_weight * kl_loss); one wave, lane-strided partials in a fixed order
__global__ __launch_bounds__(WAVE) void vae_loss_final_kernel(const float* __restrict__ parts, int nparts, float inv_na, float inv_nl,
                                                              float kl_weight, float* __restrict__ out) {
  float se = 0.f, kl = 0.f;
  for (int i = threadIdx.x; i < nparts; i += WAVE) {
    se += parts[2 * i];
    kl += parts[2 * i + 1];
  }
  se = wave_sum(se);
  kl = wave_sum(kl);
  if (threadIdx.x == 0) {
    const float recon = se * inv_na, kld = -0.5f * (kl * inv_nl);
    out[0] = recon;
    out[1] = kld;
    out[2] = recon + kl_weight * kld;
  }
}

// gradients of g[0] * recon + g[1] * kl + g[2] * (recon + kl_weight * kl) w.r.t. u, mean, std (g: the incoming gradient of out3)
__global__ __launch_bounds__(VT) void vae_loss_bwd_kernel(const float* __restrict__ u, int64_t ldu, const float* __restrict__ a, int64_t lda,
                                                          const float* __restrict__ mean, int64_t ldm, const float* __restrict__ sd,
                                                          int64_t lds_, int rows, int A, int L, const float* __restrict__ g,
                                                          float kl_weight, float* __restrict__ du, int64_t ld_du,
                                                          float* __restrict__ dmean, int64_t ld_dmean, float* __restrict__ dstd,
                                                          int64_t ld_dstd) {
  const int64_t na = (int64_t)rows * A, nl = (int64_t)rows * L;
  const float ca = (g[0] + g[2]) * 2.f / (float)na;
  const float cl = (g[1] + g[2] * kl_weight) * -0.5f / (float)nl;
  for (int64_t i = (int64_t)blockIdx.x * VT + threadIdx.x; i < na; i += (int64_t)gridDim.x * VT) {
    const int r = (int)(i / A), c = (int)(i - (int64_t)r * A);
    du[r * ld_du + c] = ca * (u[r * ldu + c] - a[r * lda + c]);
  }
  for (int64_t i = (int64_t)blockIdx.x * VT + threadIdx.x; i < nl; i += (int64_t)gridDim.x * VT) {
    const int r = (int)(i / L), c = (int)(i - (int64_t)r * L);
    const float m = mean[r * ldm + c], s = sd[r * lds_ + c];
    dmean[r * ld_dmean + c] = cl * (-2.f * m);
    dstd[r * ld_dstd + c] = cl * (2.f / s - 2.f * s);
  }
}

inline int grid_for(int64_t elems) {
  int64_t g = (elems + VT - 1) / VT;
  return (int)(g < 1 ? 1 : (g > MAX_PARTS ? MAX_PARTS : g));
}

}  // namespace

extern "C" {

int recnn_vae_latent_fwd(const float* ml, int64_t ld_ml, const float* eps, int64_t ld_eps, int rows, int latent, float* z, int64_t ldz,
                         float* std_out, int64_t ld_std, void* stream) {
  RECNN_REQUIRE(ml && eps && z && std_out && rows >= 0 && latent > 0, "vae_latent_fwd: bad arguments");
  RECNN_REQUIRE(ld_ml >= 2 * latent && ld_eps >= latent && ldz >= latent && ld_std >= latent, "vae_latent_fwd: row strides too small");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(vae_latent_fwd_kernel, dim3(grid_for((int64_t)rows * latent)), dim3(VT), 0, (hipStream_t)stream, ml, ld_ml, eps, ld_eps,
                     rows, latent, z, ldz, std_out, ld_std);
  return recnn_check_hip(hipGetLastError(), "vae_latent_fwd launch");
}

int recnn_vae_latent_bwd(const float* ml, int64_t ld_ml, const float* eps, int64_t ld_eps, const float* std_in, int64_t ld_std,
                         const float* dz, int64_t ld_dz, const float* dmean, int64_t ld_dmean, const float* dstd, int64_t ld_dstd, int rows,
                         int latent, float* dml, int64_t ld_dml, void* stream) {
  RECNN_REQUIRE(ml && eps && std_in && dml && rows >= 0 && latent > 0, "vae_latent_bwd: bad arguments");
  RECNN_REQUIRE(ld_ml >= 2 * latent && ld_dml >= 2 * latent && ld_eps >= latent && ld_std >= latent, "vae_latent_bwd: row strides too small");
  RECNN_REQUIRE((!dz || ld_dz >= latent) && (!dmean || ld_dmean >= latent) && (!dstd || ld_dstd >= latent),
                "vae_latent_bwd: gradient row strides too small");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(vae_latent_bwd_kernel, dim3(grid_for((int64_t)rows * latent)), dim3(VT), 0, (hipStream_t)stream, ml, ld_ml, eps, ld_eps,
                     std_in, ld_std, dz, ld_dz, dmean, ld_dmean, dstd, ld_dstd, rows, latent, dml, ld_dml);
  return recnn_check_hip(hipGetLastError(), "vae_latent_bwd launch");
}

int recnn_vae_loss_fwd(const float* recon, int64_t ld_recon, const float* action, int64_t ld_action, const float* mean, int64_t ld_mean,
                       const float* std_in, int64_t ld_std, int rows, int action_dim, int latent, float kl_weight, float* out3,
                       float* scratch, void* stream) {
  RECNN_REQUIRE(recon && action && mean && std_in && out3 && scratch && rows > 0 && action_dim > 0 && latent > 0,
                "vae_loss_fwd: bad arguments (scratch: 2 * 256 floats)");
  const int64_t na = (int64_t)rows * action_dim, nl = (int64_t)rows * latent;
  const int nparts = grid_for(na > nl ? na : nl);
  hipLaunchKernelGGL(vae_loss_part_kernel, dim3(nparts), dim3(VT), 0, (hipStream_t)stream, recon, ld_recon, action, ld_action, mean, ld_mean,
                     std_in, ld_std, rows, action_dim, latent, scratch);
  hipLaunchKernelGGL(vae_loss_final_kernel, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, scratch, nparts, 1.f / (float)na, 1.f / (float)nl,
                     kl_weight, out3);
  return recnn_check_hip(hipGetLastError(), "vae_loss_fwd launch");
}

int recnn_vae_loss_bwd(const float* recon, int64_t ld_recon, const float* action, int64_t ld_action, const float* mean, int64_t ld_mean,
                       const float* std_in, int64_t ld_std, int rows, int action_dim, int latent, const float* gout, float kl_weight,
                       float* d_recon, int64_t ld_drecon, float* d_mean, int64_t ld_dmean, float* d_std, int64_t ld_dstd, void* stream) {
  RECNN_REQUIRE(recon && action && mean && std_in && gout && d_recon && d_mean && d_std && rows > 0 && action_dim > 0 && latent > 0,
                "vae_loss_bwd: bad arguments");
  const int64_t na = (int64_t)rows * action_dim, nl = (int64_t)rows * latent;
  hipLaunchKernelGGL(vae_loss_bwd_kernel, dim3(grid_for(na > nl ? na : nl)), dim3(VT), 0, (hipStream_t)stream, recon, ld_recon, action,
                     ld_action, mean, ld_mean, std_in, ld_std, rows, action_dim, latent, gout, kl_weight, d_recon, ld_drecon, d_mean, ld_dmean,
                     d_std, ld_dstd);
  return recnn_check_hip(hipGetLastError(), "vae_loss_bwd launch");
}

}  // extern "C"
