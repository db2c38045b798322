// optim.h -- flat-arena optimizer kernels (optim.hip).
#pragma once
#include "common.h"
#include "comm_dev.h"

// One parameter tensor of a network inside the flat canonical arena [w1|b1|w2|b2|w3|b3].
struct TensorSeg {
  int64_t p_off;        // offset (floats) in the canonical arena
  int rows, cols;       // torch [out, in] shape (bias: rows = 1)
  int64_t sh_off;       // offset (tc elements) in the shadow arena, -1 = tensor has no shadow
  int sh_ld;            // shadow row pitch (elements), zero padded to the GEMM K multiple
  int col_rot;          // shadow column = (col + col_rot) mod cols   (critic W1: [action | state])
  const float* gpart;   // gradient partial slabs (slab 0), NULL = none
  int nslab;
  int64_t slab_stride;
  int blk0;             // first workgroup of this tensor
  int small;            // 1: OPT_SMALL_ELEMS elements per workgroup (slabs split over the 4 waves), 0: OPT_BLOCK_ELEMS
  int vec4;             // offsets / sizes allow 16-byte accesses
  int pair;             // small tensors only: gpart holds 2 nslab HALF slabs (mlpt.hip's 16-row panels); slab s = half[2 s] + half[2 s + 1]
};
struct NetLayout {
  TensorSeg t[6];
  int nblk;
  int64_t n_params;
};

// Workgroup -> element mapping of the optimizer kernels.  It depends on tensor shapes only, so every kernel that
// produces or consumes the per-workgroup |g| partial sums (clip quirk) agrees on it.  A large tensor is cut into
// flat chunks of OPT_BLOCK_ELEMS (4 consecutive elements per thread, one pass, everything in flight at once); a
// small one (biases, the critic's last layer: few elements but up to rows/16 partial slabs each) into chunks of
// OPT_SMALL_ELEMS whose slab range is split over the four waves.
constexpr int OPT_BLOCK_ELEMS = 1024;
constexpr int OPT_SMALL_ELEMS = 64;
constexpr int OPT_SMALL_MAX = 1024;
inline int opt_is_small(int64_t n) { return n <= OPT_SMALL_MAX; }
inline int opt_blocks(int64_t n) {
  return (int)(opt_is_small(n) ? (n + OPT_SMALL_ELEMS - 1) / OPT_SMALL_ELEMS : (n + OPT_BLOCK_ELEMS - 1) / OPT_BLOCK_ELEMS);
}

struct ApplyArgs {
  float* p;             // canonical parameters
  const float* g;       // reduced gradient (flat, canonical layout)
  float* m;
  float* v;
  void* shadow;         // tc shadow arena of this net (may be NULL)
  int tc_bf16;
  int do_adam;
  const int32_t* t_ptr; // device: number of optimizer steps already taken ...
  int t_add;            // ... plus this many captured ahead of the counter (run graphs tick once, at their end)
  float lr, beta1, beta2, eps, weight_decay;
  float grad_scale;
  const float* l1part;  // clip quirk: per-workgroup |g| partial sums, n_l1 of them (0 = no clip)
  int n_l1;
  float* tgt_p;         // soft update target (canonical) or NULL
  void* tgt_shadow;
  float tau;
  float* coef_out;      // optional: the clip coefficient (debug)
  int from_slabs;       // gradient = sum of the layout's partial slabs (fused slab reduction); g_out receives it
  float* g_out;
  double log_beta1, log_beta2;  // filled by apply_launch
  float omb1, omb2;             // 1 - beta in the precision torch uses (Python double, then rounded to float): apply_launch
  int opt_kind;         // RECNN_OPT_ADAM | RECNN_OPT_RANGER
  float* slow;          // Ranger: Lookahead slow weights (canonical layout)
  float la_alpha;
  int la_k;
  float nsma_thr;
  // Data parallel, from_slabs only (comm.world > 0): the exchange of the slab-summed gradient runs INSIDE this launch, workgroup by
  // workgroup -- workgroup b publishes ITS elements into the peer buffer, meets workgroup b of every other rank (flags, no
  // grid-wide counter), reduces its 1 / world share of those elements over the ranks, and reads its elements' sums back
  // (optim.hip exchange_grads; protocol: comm.hip).  A data-parallel critic step then has the launches of the single-GPU step.
  CommPort comm;
  int comm_nwg;         // workgroups of the optimizer role (= the layout's nblk <= COMM_MAX_WG)
  int g_sys;            // g lives in peer-written memory (the all-reduce's out[] buffer, comm.hip): read it with system-scope loads
};

// The hyper-parameters cross the C ABI as floats; torch computes 1 - beta and beta^t from the Python double the user wrote
// (0.999, not float(0.999) = 0.99900001287...: 1 - 0.999f is off by 1.3e-5 relative).  A float keeps 7 significant digits,
// so printing it with 7 digits recovers the literal.
double recnn_snap7(float x);

// RAdam step scalars for step t (double, as the Python implementations compute them): rectified?, step size factor
struct RadamScalars { int rect; float step; };
__host__ __device__ inline RadamScalars radam_scalars(int t, double log_beta1, double log_beta2, double beta2, double nsma_thr) {
  const double b2t = exp((double)t * log_beta2), b1t = exp((double)t * log_beta1);
  const double n_max = 2.0 / (1.0 - beta2) - 1.0;
  const double n_sma = n_max - 2.0 * (double)t * b2t / (1.0 - b2t);
  RadamScalars r;
  r.rect = n_sma > nsma_thr;
  r.step = (float)((r.rect ? sqrt((1.0 - b2t) * (n_sma - 4.0) / (n_max - 4.0) * (n_sma - 2.0) / n_sma * n_max / (n_max - 2.0)) : 1.0)
                   / (1.0 - b1t));
  return r;
}

// ---- the optimizer arithmetic of ONE element, shared by every kernel that applies it (apply_kernel, the dW GEMM's fused
// epilogue in gemm.hip): floating-point contraction is OFF inside these functions, so the operation sequence is exactly the
// one written here whatever code surrounds the call -- a fused multiply-add chosen in one kernel and not in the other would
// make "gradient arena + apply_kernel" (data parallel, external optimizers) and "fused epilogue" (single GPU) drift apart.
struct OptScalars {
  int adam, ranger, rect, la_sync;
  float step_size, bc2_sqrt, sl_lr;
  int pad;
};
// The step scalars are double-precision arithmetic (bias corrections 1 - beta^t = -expm1(t ln beta), sqrt, a division: torch
// computes them in Python floats) on a dependent chain of ~500 fp64 instructions (~13k shader cycles, in-kernel trace, round 3).
// Every thread of apply_kernel evaluates them under the latency of its own loads; a per-graph table of precomputed scalars was
// measured (round 3: no change) and removed (round 5).
__host__ __device__ inline OptScalars opt_scalars_at(int do_adam, int opt_kind, int t, float lr, double log_beta1, double log_beta2,
                                                     float nsma_thr, int la_k) {
  OptScalars S;
  S.adam = do_adam && opt_kind != RECNN_OPT_RANGER;
  S.ranger = do_adam && opt_kind == RECNN_OPT_RANGER;
  S.rect = 0; S.la_sync = 0; S.step_size = 0.f; S.bc2_sqrt = 1.f; S.sl_lr = 0.f; S.pad = 0;
  if (S.ranger) {
    // RAdam + Lookahead, the published torch_optimizer.Ranger algorithm (recnn/nn/algo.py:84-89 builds it; the package
    // itself is absent and un-pinned: restated, see recnn_amd/optim.py):
    //   v = b2 v + (1-b2) g^2;  m = b1 m + (1-b1) g;  N_sma = N_max - 2 t b2^t / (1 - b2^t)
    //   p -= wd lr p;  N_sma > thr: p -= step lr m / (sqrt(v) + eps)  else  p -= step lr m
    //   every k-th step: slow += alpha (p - slow); p = slow
    const RadamScalars rs = radam_scalars(t, log_beta1, log_beta2, exp(log_beta2), (double)nsma_thr);
    S.rect = rs.rect;
    S.sl_lr = rs.step * lr;
    S.la_sync = la_k > 0 && (t % la_k) == 0;
  } else if (S.adam) {
    // bias corrections 1 - beta^t = -expm1(t ln beta) in double (torch computes them in Python floats)
    const double bc1 = -expm1((double)t * log_beta1);
    const double bc2 = -expm1((double)t * log_beta2);
    S.step_size = (float)((double)lr / bc1);
    S.bc2_sqrt = (float)sqrt(bc2);
  }
  return S;
}
#if defined(__HIPCC__)
__device__ inline OptScalars opt_scalars(const ApplyArgs& a) {
  const int t = a.do_adam ? *a.t_ptr + 1 + a.t_add : 0;
  return opt_scalars_at(a.do_adam, a.opt_kind, t, a.lr, a.log_beta1, a.log_beta2, a.nsma_thr, a.la_k);
}
// the hyper-parameters opt_elem reads, as a register-resident copy (dwadam.hip: read from the kernel-argument block at their use sites they
// were scalar loads + waits inside the optimizer arithmetic); same field names as ApplyArgs, opt_elem takes either
struct OptK { float lr, beta1, beta2, eps, weight_decay, omb1, omb2, la_alpha; };
// g: raw gradient, gs: gradient scale (1/world, clip coefficient); p, m, v, sl updated in place (sl only on Lookahead syncs)
template <class A>
__device__ inline void opt_elem(const A& a, const OptScalars& S, float g, float gs, float& p, float& m, float& v, float& sl) {
#pragma clang fp contract(off)
  if (S.ranger) {
    const float gj = g * gs;
    v = a.beta2 * v + (a.omb2 * gj) * gj;
    m = a.beta1 * m + a.omb1 * gj;
    if (a.weight_decay != 0.f) p = p + (-a.weight_decay * a.lr) * p;
    if (S.rect) p = p + (-S.sl_lr) * (m / (sqrtf(v) + a.eps));
    else p = p + (-S.sl_lr) * m;
    if (S.la_sync) { sl = sl + a.la_alpha * (p - sl); p = sl; }
  } else if (S.adam) {
    float gj = g * gs;
    if (a.weight_decay != 0.f) gj = gj + a.weight_decay * p;
    m = m + a.omb1 * (gj - m);
    v = a.beta2 * v + (a.omb2 * gj) * gj;
    const float denom = sqrtf(v) / S.bc2_sqrt + a.eps;
    p = p - S.step_size * (m / denom);
  }
}
// recnn/utils/misc.py:3-5, that operand order: target * (1 - tau) + param * tau
__device__ inline float soft_elem(float tp, float p, float tau) {
#pragma clang fp contract(off)
  return tp * (1.0f - tau) + p * tau;
}
#endif

int grad_reduce_launch(const NetLayout& L, float* gflat, float* l1part, hipStream_t s);
// fills the launch-time scalars of an ApplyArgs (log betas in double, 1 - beta as torch rounds them)
void apply_args_finish(ApplyArgs* a);

struct GatherArgs;
// pregather != NULL: the sampler + gather of the NEXT step runs as extra workgroups of this launch
int apply_launch(const NetLayout& L, const ApplyArgs& a, hipStream_t s, const GatherArgs* pregather = nullptr);
int scale_grads_launch(const NetLayout& L, float* gflat, const float* l1part, int n_l1, float grad_scale, hipStream_t s);
int l1_blocks_launch(const NetLayout& L, const float* gflat, float* l1part, hipStream_t s);
