// x3tail.h -- layers 2 + 3 of the step's networks for the split-bf16 compute type, one launch (x3tail.hip)
#pragma once
#include "common.h"

constexpr int X3TAIL_MAX_GROUP = 8;

struct X3TailProb {
  const void* h1; int64_t ldh;      // layer-1 activations, split rows [rows, ldh] (512 physical columns used)
  const void* W2; int64_t ldw2;     // split shadow [256 rows, ldw2]
  const void* W3; int64_t ldw3;     // actor: split shadow [128 rows, ldw3]; NULL: critic
  const float* b2;
  const float* b3;                  // actor: [out_dim]; critic: [1]
  const float* w3row;               // critic: canonical fp32 [H]
  int rows, H, out_dim;
  // dropout of the second hidden layer (layer 1's is in h1 already)
  int mask_mode;
  const uint8_t* mask2; int64_t ld_mask;
  uint32_t seed, stream2;
  const int32_t* step_ptr; int step_add;
  void* h2;                         // out (optional): split rows [rows, ldh]
  void* out; int64_t ldo;           // actor output, split rows
  const float* addend; int64_t ld_add; float add_clip;
  float* q;                         // critic output fp32 [rows] (optional)
  float* q_part;                    // optional: [panels * x3tail_parts_per_panel()] sums of q
};
struct X3TailBatch { X3TailProb p[X3TAIL_MAX_GROUP]; };

int x3tail_init();
int x3tail_parts_per_panel();
int x3tail_launch(const X3TailBatch& b, int nprob, hipStream_t s);
