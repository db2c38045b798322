// dwadam.h -- the critics' weight-gradient GEMMs with the optimizer in their epilogue (dwadam.hip).
#pragma once
#include "optim.h"

// One weight tensor of a network: dW[M, N] = dZ[rows, M]^T X[rows, N], M = N.L.t[tensor].rows (a multiple of 32), N = .cols
struct DwAdamProb {
  const void* dz;        // bf16 [rows, ldz]: backward tensor, ALREADY times the per-row loss seed (mlpt.hip)
  const void* x;         // bf16 [rows, ldx]: the layer's input, columns in SHADOW order (critic W1: [action | state]), zero padded to 64
  int64_t ldz, ldx;
  int tensor;            // index into NetLayout.t
  int tiles_m, tiles_n;  // 32 x 64 output tiles
  int nslab;             // 8 | 16: batch slices summed exactly like the slabs of the two-launch path (NetLayout.t[tensor].nslab)
};
struct DwAdamNet {
  NetLayout L;
  ApplyArgs a;           // finished by dwadam_launch (apply_args_finish)
  DwAdamProb w[2];       // launch order: the small contraction (W2) first
  int rows;
  int x3;                // split-bf16 operands and shadows (filled by dwadam_launch from a.tc_bf16)
  int nsmall;            // optimizer workgroups of the tensors that are NOT weight tiles (biases, the last layer)
  int ntile[2];
};
constexpr int DWADAM_MAX_NETS = 2;
struct DwAdamBatch {
  DwAdamNet n[DWADAM_MAX_NETS];
};

int dwadam_init();
// whether tensor `ti` of layout L can be a tile problem of this kernel at `rows` batch rows
bool dwadam_tensor_ok(const NetLayout& L, int ti, int rows, int x3 = 0);
int dwadam_launch(DwAdamBatch& b, int nnet, hipStream_t s);
