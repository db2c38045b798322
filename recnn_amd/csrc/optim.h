// optim.h -- flat-arena optimizer kernels (optim.hip).
#pragma once
#include "common.h"

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

int grad_reduce_launch(const NetLayout& L, float* gflat, float* l1part, hipStream_t s);
struct GatherArgs;
// pregather != NULL: the sampler + gather of the NEXT step runs as extra workgroups of this launch
int apply_launch(const NetLayout& L, const ApplyArgs& a, hipStream_t s, const GatherArgs* pregather = nullptr);
int scale_grads_launch(const NetLayout& L, float* gflat, const float* l1part, int n_l1, float grad_scale, hipStream_t s);
int l1_blocks_launch(const NetLayout& L, const float* gflat, float* l1part, hipStream_t s);
