// gather.h -- argument block of frame_gather_kernel (gather.hip), shared with engine.hip.
#pragma once
#include "common.h"

struct GatherArgs {
  const int32_t* items;
  const float* ratings;
  const int64_t* user_off;
  const int32_t* users;
  const int32_t* row_off;
  int n_users, rows, frame, emb;
  const float* table;
  float* state; int64_t ld_state;
  float* next_state; int64_t ld_next;
  float* action; int64_t ld_action;
  float* reward;
  float* done;
  const int32_t* cursor;
  int cursor_stride;
  int cursor_add, cursor_mod;  // batch index = (*cursor + cursor_add) mod cursor_mod (cursor_mod 0: no wrap)
  int inline_plan;  // row_off == NULL: every workgroup scans the batch's history lengths itself (n_users <= 1024)
  // optional per-epoch plan table (recnn_frame_plan_rows): plan[batch * plan_stride + row] = (CSR offset of the row's window
  // start << 1) | done, -1 for rows the batch does not have; replaces users -> offsets -> scan -> search by ONE load
  const int64_t* plan;
  int64_t plan_stride;
  // optional bf16 twins of the packed rows (engine, bf16 compute mode): same columns, row stride ld_h (elements)
  bf16_t* state_h;
  bf16_t* next_h;
  bf16_t* action_h;
  int64_t ld_h;
  int x3;   // the twins are split-bf16 rows (x3.h): ld_h physical, every value stored as hi (mapped column) + lo (32 further)
};


int frame_gather_launch(GatherArgs a, hipStream_t s);
// batches cursor_add .. cursor_add + n_sets - 1 of the plan; batch j lands j * a.rows rows below the given outputs (bf16 rows only)
int frame_gather_multi_launch(const GatherArgs& a, int n_sets, hipStream_t s);
size_t frame_gather_lds_bytes(const GatherArgs& a, int rows_per_wg);
