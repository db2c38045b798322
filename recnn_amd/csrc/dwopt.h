// dwopt.h -- the critic's weight-gradient GEMMs with the optimizer in their epilogue (dwopt.hip).
#pragma once
#include "gemm.h"
#include "optim.h"
#include "gather.h"

constexpr int DWOPT_APPLY = 1;   // epilogue: Adam / Ranger (+ soft target update) on the tile's own parameters
constexpr int DWOPT_GRAD = 2;    // epilogue: the finished gradient tile goes to the flat gradient arena (ApplyArgs::g_out)

// One launch = every parameter tensor of up to two critics:
//   W1, W2   64 x 64 output tiles over the WHOLE batch (no split, no partial slabs), GemmProb as for gemm_dw_dma_kernel
//   w3, b1, b2, b3   "vector" workgroups: column sums over all rows of d_r * {h2, u2, U} and sum_r d_r (DwVecProb's inputs)
// and, optionally, the replay sampler + gather of the next step as extra workgroups (as apply_gather_kernel did).
struct DwOpt {
  int mode, n_net;
  int probe;                      // timing probes (recnn_tune_dw_probe), 0 in production
  ApplyArgs a[2];                 // per critic (apply_args_finish()ed by dwopt_launch)
  TensorSeg seg[2][6];            // per critic: [w1|b1|w2|b2|w3|b3]
  signed char prob_net[GEMM_MAX_GROUP], prob_tensor[GEMM_MAX_GROUP];
  DwVecProb v[2];                 // rows, H, delta, h2, u2, U, ldh (the *_part outputs are not used)
};

int dwopt_init();
void dwopt_set_groups(int groups);   // k-groups of 4 waves per workgroup: 4 (default), 2, 1
// L: a GEMM_DW launch description (bf16 operands, dw_splits ignored); pregather may be NULL
int dwopt_launch(GemmLaunch* L, const DwOpt& o, const GatherArgs* pregather, hipStream_t s);
bool dwopt_eligible(const GemmLaunch* L);
