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
  int blk0;             // first workgroup of this tensor (one workgroup per row)
};
struct NetLayout {
  TensorSeg t[6];
  int nblk;
  int64_t n_params;
};

struct ApplyArgs {
  float* p;             // canonical parameters
  const float* g;       // reduced gradient (flat, canonical layout)
  float* m;
  float* v;
  void* shadow;         // tc shadow arena of this net (may be NULL)
  int tc_bf16;
  int do_adam;
  const int32_t* t_ptr; // device: number of optimizer steps already taken
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
};

int grad_reduce_launch(const NetLayout& L, float* gflat, float* l1part, hipStream_t s);
int apply_launch(const NetLayout& L, const ApplyArgs& a, hipStream_t s);
int scale_grads_launch(const NetLayout& L, float* gflat, const float* l1part, int n_l1, float grad_scale, hipStream_t s);
