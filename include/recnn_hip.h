/*
 * recnn_hip.h -- C ABI of librecnn_hip.so: the MI355X (gfx950) implementation of the RecNN
 * DDPG/TD3 inner training step.
 *
 * The reference (awarebayes/RecNN) is pure Python and has no FFI; the "drop-in boundary" is
 * its Python API, mirrored by the `recnn_amd` package.  This header is the native boundary
 * underneath: every entry point below replaces one group of reference Python/ATen calls and
 * is what a maintainer of the reference would bind with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative RECNN_E_* code or a positive hipError_t;
 *     `recnn_last_error()` returns a human-readable message for the calling thread.
 *   - all pointers are DEVICE pointers unless the parameter name starts with `h_`.
 *   - `stream` is a `hipStream_t` passed as void* (NULL = the null stream).  All work is
 *     stream-ordered; nothing synchronises the host unless stated.
 *   - no ownership is transferred: callers allocate every buffer (the Python host uses torch
 *     for device memory); the engine object only records pointers.
 *   - "tc" buffers hold the compute type of the engine: float (RECNN_F32) or bfloat16
 *     (RECNN_BF16, fp32 accumulate, fp32 master weights).
 *
 * Reference lines each group replaces are cited as `recnn/...py:lines` (relative to the
 * upstream repository root).
 */
#ifndef RECNN_HIP_H
#define RECNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): recnn_engine_tuning replaced the recnn_tune_* setters, recnn_gemm_args grew ws / ws_bytes, recnn_shadow_out and the
 * split-bf16 type were added (round 4), recnn_engine_tuning::x3_fwd took a reserved slot. */
#define RECNN_ABI_VERSION 2

enum {
  RECNN_OK = 0,
  RECNN_E_INVALID = -1,   /* bad argument (null pointer, unsupported size, misalignment) */
  RECNN_E_STATE = -2,     /* call order violated (e.g. step before bind) */
  RECNN_E_UNSUPPORTED = -3
};

/* Compute types.  RECNN_BF16X3 ("split bf16"): every value is held as hi = bf16(x), lo = bf16(x - hi) and every product is
 * evaluated as hi*hi + hi*lo + lo*hi on the bf16 matrix cores into an fp32 accumulator: fp32-grade results (the 1e-4 loss-curve
 * parity of the fp32 path) at bf16 MFMA rates.  Tensors of this type are bfloat16 arrays with TWICE the logical columns: logical
 * column c lives at physical column x = 2 (c & ~31) + (c & 31) (hi) and x + 32 (lo); leading dimensions and contraction lengths
 * passed for such tensors are PHYSICAL, output extents / bias / mask indices logical (csrc/x3.h). */
enum { RECNN_F32 = 0, RECNN_BF16 = 1, RECNN_BF16X3 = 2 };

/* dropout mask source for the train-mode networks */
enum {
  RECNN_MASK_NONE = 0,     /* no dropout (eval) */
  RECNN_MASK_HASH = 1,     /* in-kernel counter-based generator keyed by (seed, step, stream, row, col) */
  RECNN_MASK_EXTERNAL = 2  /* uint8 keep-masks supplied by the caller (parity tests) */
};

int recnn_abi_version(void);
const char* recnn_last_error(void);
/* sizeof() of an ABI struct: 0 recnn_gemm_args, 1 recnn_engine_config, 2 recnn_hyper,
 * 3 recnn_engine_sizes, 4 recnn_sampler, 5 recnn_engine_tuning, 6 recnn_shadow_out (lets a binding verify its declarations);
 * -1 if unknown. */
int64_t recnn_abi_sizeof(int which);
/* =====================================================================================
 * 1. Replay sampler + embedding gather
 *    replaces recnn/data/utils.py:161-187 (prepare_batch_static_size: rolling_window +
 *    concatenate) and recnn/data/utils.py:51-81 (batch_tensor_embeddings: emb[items], view,
 *    cat, done scatter).  Integer/copy work: bit-exact.
 *
 *    Replay store (device resident CSR): items int32[sum L] (dense item ids, time-sorted per
 *    user), ratings float[sum L], user_off int64[n_store_users+1].
 * ===================================================================================== */

/* row_off[0..n_users] = exclusive prefix sum of max(L_u - frame, 0) over the batch's users.
 * row_off[n_users] is the total row count of the batch.
 * `cursor` (device int32, may be NULL): the user list actually read is
 * batch_users + (*cursor) * cursor_stride -- lets a replayed hipGraph walk an epoch permutation. */
int recnn_frame_plan(const int64_t* user_off, const int32_t* batch_users, int n_users, int frame,
                     int32_t* row_off, const int32_t* cursor, int cursor_stride, void* stream);

/* Builds rows [0, rows) of the batch.  Output row r of user u at window t:
 *   state[r]      = [emb(i_t .. i_{t+F-1}) | r_t .. r_{t+F-1}]          (F*E + F floats)
 *   next_state[r] = [emb(i_{t+1} .. i_{t+F}) | r_{t+1} .. r_{t+F}]
 *   action[r]     = emb(i_{t+F}),  reward[r] = r_{t+F},  done[r] = (t == L_u - F - 1)
 * ld_* are row strides in floats (>= row width); rows must be 8-byte aligned, 16-byte
 * aligned rows take the vector path.  `rows` may be smaller than row_off[n_users]
 * (fixed-row batches); rows >= row_off[n_users] are left untouched.
 * row_off may be NULL when n_users <= 1024: the kernel then computes the plan itself (per workgroup, in
 * LDS) and no recnn_frame_plan launch is needed. */
int recnn_frame_gather(const int32_t* items, const float* ratings, const int64_t* user_off,
                       const int32_t* batch_users, const int32_t* row_off, int n_users, int rows,
                       int frame, int emb_dim, const float* table,
                       float* state, int64_t ld_state, float* next_state, int64_t ld_next,
                       float* action, int64_t ld_action, float* reward, float* done,
                       const int32_t* cursor, int cursor_stride, void* stream);

/* Packs a caller-made canonical batch (reference layout, utils.py:265-276 get_base_batch)
 * into the engine's packed rows: xs[r] = [action | state | 0-pad], xn[r] = [<next action
 * slot> | next_state | 0-pad].  Used when the batch did not come from recnn_frame_gather. */
int recnn_pack_batch(const float* state, int64_t ld_state, const float* action, int64_t ld_action,
                     const float* next_state, int64_t ld_next, int rows, int state_dim, int action_dim,
                     float* xs, float* xn, int64_t ld_x, void* stream);

/* =====================================================================================
 * 2. Dense layers (MFMA GEMMs) -- single-problem launchers used by the module-level
 *    forward/backward and by the kernel unit tests.  The fused step (section 4) launches
 *    the same kernels through grouped descriptors.
 *    replaces recnn/nn/models.py:66-73 and :207-213 (addmm/relu/dropout) and their autograd
 *    backward (mm, threshold_backward, dropout mul, bias sum).
 * ===================================================================================== */

typedef struct recnn_gemm_args {
  int dtype;              /* RECNN_F32 | RECNN_BF16: compute type tc */
  int M, N;               /* output rows / cols */
  /* contraction segment 0 and optional segment 1 (accumulated into the same tile) */
  const void* A[2];       /* fwd/dx: [M, lda] (k contiguous).  dw: [Kc, lda] (m = row index) */
  const void* B[2];       /* fwd: [N, ldb] (k contiguous).  dx/dw: [Kc, ldb]                 */
  int64_t lda[2], ldb[2];
  int K[2];               /* contraction length per segment; K[1] = 0 if unused.  fwd/dx: a multiple of
                             one 16-byte chunk of the compute type (4 fp32 / 8 bf16); multiples of 64 (128 for
                             bf16) take the LDS-DMA kernels.  dw: the number of valid rows (any value). */
  int a_f32[2];           /* segment's A operand is float in memory although tc is bf16 */
  int b_f32[2];           /* same for B (dw of layer 1: B = packed fp32 batch rows) */
  /* epilogue */
  void* C; int64_t ldc; int c_f32;
  const float* bias;      /* fwd: [N] or NULL */
  int relu;               /* fwd */
  int mask_mode;          /* fwd: RECNN_MASK_* */
  const uint8_t* mask; int64_t ld_mask;      /* external keep mask [M, ld_mask] */
  uint32_t seed, stream_id; const int32_t* step_ptr;   /* hash mask key (step read from device) */
  const float* addend; int64_t ld_add; float add_clip; /* fwd: C += clamp(addend, +-add_clip) (TD3 noise) */
  int add_row_div;        /* fwd: output row m reads addend row m / add_row_div (0 or 1: row m).  One addend row serves a
                             run of consecutive output rows: the state part of layer 1 shared by the n candidate actions
                             scored per state (BCQ, recnn/nn/update/bcq.py:98-104).  Sits in what used to be padding. */
  const void* yref; int64_t ldy; float dx_scale;       /* dx: C = acc * dx_scale * [yref > 0]; yref NULL = plain.
                                                          fwd: same gate applied after bias/relu (a dX computed with
                                                          pre-transposed weights through the k-contiguous kernels) */
  float* colsum;          /* dx: column sums per 32-row slab, float[ceil(M/32)][N] (bias gradients) */
  int dw_splits;          /* dw: number of K splits; slab s written at C + s*dw_slab_stride */
  int64_t dw_slab_stride;
  int dw_valid_cols;      /* dw: columns >= this are not stored */
  int dw_col_rot;         /* dw: stored column = (col + rot) mod valid_cols */
  void* ws; int64_t ws_bytes;   /* fwd, optional: caller-owned scratch.  A one-segment product with a LONG contraction and few output
                             tiles (K >= 32768, at most 64 tiles of 128 x 128: [256, 2048] x K = 100k, the catalogue-wide
                             contractions of REINFORCE) is then split into up to 8 K slices, one workgroup column each; their fp32
                             partial products (slices * M * N floats must fit) are summed in slice order by a second launch that
                             applies the epilogue.  NULL: one workgroup per tile walks the whole K. */
} recnn_gemm_args;

/* C[M,N] = epi( sum_seg A_seg[M,K] * B_seg[N,K]^T ) */
int recnn_gemm_fwd(const recnn_gemm_args* h_args, void* stream);
/* C[M,N] = epi( A[M,Kc] * B[Kc,N] )            (dX = dZ * W) */
int recnn_gemm_dx(const recnn_gemm_args* h_args, void* stream);
/* C[M,N] = A[Kc,M]^T * B[Kc,N], split over Kc  (dW = dZ^T * X) */
int recnn_gemm_dw(const recnn_gemm_args* h_args, void* stream);

/* fp32 [rows, cols] (row stride ld) <-> split-bf16 rows (RECNN_BF16X3; row stride ldx >= 2 * roundup(cols, 32) bfloat16, padding
 * columns of the split rows are left untouched by pack). */
int recnn_x3_pack(const float* src, int64_t ld, int rows, int cols, void* dst, int64_t ldx, void* stream);
int recnn_x3_unpack(const void* src, int64_t ldx, int rows, int cols, float* dst, int64_t ld, void* stream);

/* Writes the keep-mask the RECNN_MASK_HASH generator produces for (seed, step, stream_id)
 * into out[M, N] (uint8).  Lets tests feed the in-kernel masks to the CPU oracle. */
int recnn_hash_mask_dump(uint32_t seed, int32_t step, uint32_t stream_id, int M, int N,
                         uint8_t* out, void* stream);
/* ... with the step in device memory (step = *step_dev + step_add): a captured graph draws fresh masks on every replay */
int recnn_hash_mask_dump_at(uint32_t seed, const int32_t* step_dev, int step_add, uint32_t stream_id, int M, int N, uint8_t* out,
                            void* stream);

/* =====================================================================================
 * 3. Flat-arena optimizer / soft-update kernels
 *    replaces torch.optim.Adam.step (the optimizer the reference's users inject,
 *    recnn/nn/update/misc.py:44, ddpg.py:93, td3.py:97,101,134), the clip quirk
 *    torch.nn.utils.clip_grad_norm_(params, -1, 1) (ddpg.py:92, td3.py:133) and
 *    recnn/utils/misc.py:1-5 (soft_update).
 * ===================================================================================== */

/* target = target*(1-tau) + net*tau over n floats (operand order as the reference). */
int recnn_soft_update_flat(float* target, const float* net, int64_t n, float tau, void* stream);

/* One Adam step over a flat fp32 arena: p, m, v updated in place from g * grad_scale.
 * step_t is the 1-based step count (bias correction). */
int recnn_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step_t, float grad_scale, void* stream);

/* ... and, in the same pass, a row-padded copy of the updated [n / cols, cols] parameter in another leading dimension and
 * optionally in bfloat16 (the layout the GEMM kernels read in place of a weight whose rows are not 16-byte aligned, or the bf16
 * operand of the catalogue-wide products): the optimizer has the new value in a register, a separate conversion pass would
 * read 4 bytes per element again.  Padding columns [cols, ld) are not touched.  h_shadow NULL or dst NULL: plain step. */
typedef struct recnn_shadow_out {
  void* dst;       /* float or bfloat16 [n / cols, ld] */
  int cols;        /* n % cols == 0 */
  int64_t ld;      /* >= cols */
  int bf16;        /* 1: dst is bfloat16 (round to nearest even), 0: float */
} recnn_shadow_out;
int recnn_adam_flat_shadow(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int step_t, float grad_scale, const recnn_shadow_out* h_shadow, void* stream);

/* The same step with the 1-based step count taken from device memory: t = *step_dev + step_add.  For captured graphs
 * (recnn_amd.optim.Adam(capturable=True)): the graph advances *step_dev itself, every replay steps with the right count. */
int recnn_adam_flat_at(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, const int32_t* step_dev, int step_add, float grad_scale, void* stream);

/* out[0] = sum |g| over n floats (deterministic two-pass reduction; scratch >= 1024 floats). */
int recnn_l1_norm_flat(const float* g, int64_t n, float* scratch, float* out, void* stream);

/* =====================================================================================
 * 4. The fused DDPG / TD3 step engine
 *    replaces recnn/nn/update/ddpg.py:8-104, recnn/nn/update/misc.py:10-55,
 *    recnn/nn/update/td3.py:8-150 for nets = Actor/Critic (recnn/nn/models.py) and the
 *    Adam optimizer, including the soft target update.
 * ===================================================================================== */

typedef struct recnn_engine recnn_engine;

enum { RECNN_ALGO_DDPG = 0, RECNN_ALGO_TD3 = 1 };

/* network slots inside an engine */
enum {
  RECNN_NET_POLICY = 0, RECNN_NET_TARGET_POLICY = 1,
  RECNN_NET_VALUE1 = 2, RECNN_NET_TARGET_VALUE1 = 3,
  RECNN_NET_VALUE2 = 4, RECNN_NET_TARGET_VALUE2 = 5,
  RECNN_NET_COUNT = 6
};

typedef struct recnn_engine_config {
  int algo;            /* RECNN_ALGO_* */
  int dtype;           /* RECNN_F32 | RECNN_BF16 */
  int state_dim;       /* S = F*E + F (1290) */
  int action_dim;      /* A (128) */
  int hidden;          /* H (256) */
  int max_rows;        /* batch capacity */
  int mask_mode;       /* RECNN_MASK_* for the learning nets */
  uint32_t seed;
  int device;          /* hip device ordinal */
} recnn_engine_config;

typedef struct recnn_hyper {
  float gamma, min_value, max_value;   /* DDPG clamps the TD target; TD3 passes -inf/+inf */
  float soft_tau;
  int policy_every;                    /* policy_step / policy_update */
  float noise_std, noise_clip;         /* TD3 */
  /* optimizer slots: [0] = policy optimizer, [1] = value optimizer(s) */
  float lr[2], beta1[2], beta2[2], eps[2], weight_decay[2];
  /* opt_kind: RECNN_OPT_ADAM = torch.optim.Adam arithmetic (L2 weight decay folded into the gradient);
   *           RECNN_OPT_RANGER = RAdam (variance-rectified Adam, decoupled lr*wd*p decay) + Lookahead(la_k, la_alpha):
   *           the shape of the reference's default torch_optimizer.Ranger (recnn/nn/algo.py:84-89, 139-147); needs the
   *           slow-weight arena bound with recnn_engine_bind_slow. */
  int opt_kind[2];
  float la_alpha[2];          /* Lookahead interpolation (Ranger default 0.5) */
  int la_k[2];                /* Lookahead period in optimizer steps (Ranger default 6) */
  float nsma_threshold[2];    /* rectification switch: adaptive step iff N_sma > threshold (Ranger default 5) */
} recnn_hyper;
enum { RECNN_OPT_ADAM = 0, RECNN_OPT_RANGER = 1 };

/* Sizes (in bytes) of the buffers the caller must allocate for an engine. */
typedef struct recnn_engine_sizes {
  int64_t master_floats_actor, master_floats_critic;  /* canonical flat parameter arenas */
  int64_t workspace_bytes;                            /* everything else, one allocation */
  int64_t ld_x;                                       /* packed batch row stride (floats) */
  int64_t x_rows;                                     /* rows to allocate for xs / xn */
} recnn_engine_sizes;

int recnn_engine_query(const recnn_engine_config* h_cfg, recnn_engine_sizes* h_out);

/* `workspace` must be zero-initialised, 256-byte aligned, workspace_bytes long. */
int recnn_engine_create(const recnn_engine_config* h_cfg, void* workspace, recnn_engine** h_out);
void recnn_engine_destroy(recnn_engine* e);

/* Bind the canonical fp32 arenas of one network: params / grads / Adam moments, each a flat
 * array laid out [w1 | b1 | w2 | b2 | w3 | b3] in torch's [out,in] row-major layout.
 * grads/m/v may be NULL for target networks. */
int recnn_engine_bind_net(recnn_engine* e, int net, float* params, float* grads, float* adam_m, float* adam_v);
/* Lookahead slow weights of a learning network (flat fp32 arena, canonical layout, initialised by the caller with the
 * parameters as they were when the optimizer state was created); required for RECNN_OPT_RANGER. */
int recnn_engine_bind_slow(recnn_engine* e, int net, float* slow);
/* One optimizer step over a flat fp32 array, RAdam + Lookahead (RECNN_OPT_RANGER arithmetic; step_t = 1-based step). */
int recnn_ranger_flat(float* p, const float* g, float* m, float* v, float* slow, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, float la_alpha, int la_k, float nsma_threshold, int step_t,
                      float grad_scale, void* stream);
int recnn_ranger_flat_shadow(float* p, const float* g, float* m, float* v, float* slow, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float la_alpha, int la_k, float nsma_threshold, int step_t,
                             float grad_scale, const recnn_shadow_out* h_shadow, void* stream);

/* =====================================================================================
 * 3b. Categorical policy head (REINFORCE)
 *    replaces F.softmax + torch.distributions.Categorical(probs).sample() / .log_prob() of
 *    DiscreteActor (recnn/nn/models.py:95-99, :107-111, :116-141), their autograd backward, and the
 *    one-hot action rows of batch_contstate_discaction (recnn/data/utils.py:108-109).
 *    Rows are float[rows, ld] with ld % 4 == 0, 16-byte aligned; columns [n, ld) are padding.
 * ===================================================================================== */
#define RECNN_CAT_SOFTMAX 1 /* x holds logits; overwritten with p = softmax(x) (padding columns become 0) */
#define RECNN_CAT_SAMPLE 2  /* draw actions[row] ~ p / sum p (else actions[row] is read; NULL = no action) */
/* logprob[row] = log(clamp(p[a] / sum p, eps, 1 - eps)) (torch.distributions.Categorical arithmetic);
 * rowstat[row] = {max logit, sum exp, sum p, 1.0 if the clamp was active}.  logprob / rowstat may be NULL.
 * The uniform of row r is a pure function of (seed, step, r); the inverse-CDF walks items in a fixed order. */
int recnn_categorical_rows(float* x, int64_t ld, int rows, int n, int flags, uint32_t seed, int32_t step,
                           int64_t* actions, float* logprob, float* rowstat, void* stream);
#define RECNN_LPB_ACCUMULATE 1 /* dlogits += ... (else =) */
#define RECNN_LPB_BF16 2       /* dlogits is bfloat16[rows, ldd] (operand of the bf16 catalogue GEMMs); else float */
/* dlogits (+)= g[row] * (onehot(a) - p / sum p), zero for rows whose clamp was active (g NULL = zeros);
 * colsum[n] (optional) = column sums of the resulting dlogits (bias gradient), scratch = float[ceil(rows/32) * round4(n)]. */
int recnn_logprob_bwd(const float* p, int64_t ldp, int rows, int n, const int64_t* actions, const float* g,
                      const float* rowstat, void* dlogits, int64_t ldd, int flags, float* colsum, float* scratch,
                      void* stream);
/* dlogits = p * (dprobs - sum_j dprobs_j p_j): backward of p = softmax(logits). */
int recnn_softmax_bwd(const float* p, int64_t ldp, int rows, int n, const float* dprobs, int64_t lddp, float* dlogits,
                      int64_t ldd, void* stream);
/* out[r, :] = onehot(idx[r]) over n columns (columns [n, ld) zeroed). */
int recnn_onehot_rows(const int64_t* idx, int rows, int n, float* out, int64_t ld, void* stream);
/* out[c] = sum over rows of x[r, c], c < n (rows added in order; x rows 16-byte aligned, ld % 4 == 0, padded to 4 floats): the bias
 * gradient of a catalogue-wide Linear from d logits. */
int recnn_colsum_rows(const float* x, int64_t ld, int rows, int n, float* out, void* stream);
/* The softmax of rows whose columns are SHARDED over ranks (the vocabulary-parallel policy head, recnn_amd/parallel.py; replaces the
 * F.softmax of recnn/nn/models.py:95-99 on a shard): three passes over this rank's float[rows, ld] logits x, the all-reduces between
 * them are the caller's.  pass 0: rowval[r] = max_j x[r, j].  pass 1: x = exp(x - rowval[r]) in place (rowval = the max over ALL shards),
 * pa[r] = the row's sum.  pass 2: x /= rowval[r] in place (rowval = the sum over all shards) -> this shard's probabilities, pa[r] =
 * x[r, local[r]] if 0 <= local[r] < n else 0 (local = the row's action minus the shard's first item). */
int recnn_shard_softmax_pass(float* x, int64_t ld, int rows, int n, int pass, float* rowval, const int64_t* local, float* pa, void* stream);
/* dlogits[r, j] = -g[r] p[r, j] (+ g[r] at j = local[r] when the shard owns it): backward of log(clamp(p_a)) through the sharded softmax */
int recnn_shard_logprob_bwd(const float* p, int64_t ldp, int rows, int n, const int64_t* local, const float* g, float* dlogits, int64_t ldd,
                            void* stream);
/* dst[c, r] = src[r, c]: float [rows, cols] (row stride ld) -> float or bfloat16 [cols, ldt] (ldt >= rows; columns [rows, ldt) are
 * not touched).  The transposed copy of the policy head's W2 [n_items, hidden] that puts the catalogue on the contiguous axis
 * for the backward product d logits x W2 (made once per weight version). */
int recnn_transpose_rows(const float* src, int64_t ld, int rows, int cols, void* dst, int64_t ldt, int dst_bf16, void* stream);

/* =====================================================================================
 * 3c. Conditional-VAE latent layer and loss (BCQ)
 *    replaces the ATen chain of bcqGenerator.forward between the encoder and decoder GEMMs
 *    (recnn/nn/models.py:271-277: clamp(log_std, -4, 15), exp, z = mean + std * eps) and the generator loss of
 *    bcq_update (recnn/nn/update/bcq.py:78-81: mse(recon, action) + 0.5 * KL), with their autograd backward.
 *    All operands float[rows, ld]; `ml` holds [mean | raw log_std] side by side (2 * latent columns).
 * ===================================================================================== */
#define RECNN_VAE_LOG_STD_MIN (-4.0f)
#define RECNN_VAE_LOG_STD_MAX (15.0f)
/* std = exp(clamp(ml[:, L:2L])), z = ml[:, :L] + std * eps */
int recnn_vae_latent_fwd(const float* ml, int64_t ld_ml, const float* eps, int64_t ld_eps, int rows, int latent, float* z,
                         int64_t ldz, float* std_out, int64_t ld_std, void* stream);
/* dml[:, :L] = dz + dmean;  dml[:, L:] = (dz * eps + dstd) * std * [min <= raw <= max];  dz / dmean / dstd may be NULL */
int recnn_vae_latent_bwd(const float* ml, int64_t ld_ml, const float* eps, int64_t ld_eps, const float* std_in, int64_t ld_std,
                         const float* dz, int64_t ld_dz, const float* dmean, int64_t ld_dmean, const float* dstd,
                         int64_t ld_dstd, int rows, int latent, float* dml, int64_t ld_dml, void* stream);
/* out3 = { mean((recon - action)^2), -0.5 * mean(1 + log(std^2) - mean^2 - std^2), out3[0] + kl_weight * out3[1] };
 * scratch: 512 floats.  Fixed summation order (bit-reproducible). */
int recnn_vae_loss_fwd(const float* recon, int64_t ld_recon, const float* action, int64_t ld_action, const float* mean,
                       int64_t ld_mean, const float* std_in, int64_t ld_std, int rows, int action_dim, int latent,
                       float kl_weight, float* out3, float* scratch, void* stream);
/* gradients of sum_i gout[i] * out3[i] w.r.t. recon, mean and std (gout: float[3] on the device) */
int recnn_vae_loss_bwd(const float* recon, int64_t ld_recon, const float* action, int64_t ld_action, const float* mean,
                       int64_t ld_mean, const float* std_in, int64_t ld_std, int rows, int action_dim, int latent,
                       const float* gout, float kl_weight, float* d_recon, int64_t ld_drecon, float* d_mean, int64_t ld_dmean,
                       float* d_std, int64_t ld_dstd, void* stream);

/* =====================================================================================
 * 3d. Replay-store builder: ratings rows -> CSR ordered by (user, time)   
 *    replaces recnn/data/dataset_functions.py:84-126 (prepare_dataset: rating transform and movieId -> dense id map as
 *    per-row Python lambdas, sort_values(by="timestamp"), a Python callback per user group).  All pointers are device
 *    memory except host_counts.  Rows with equal (user, timestamp) keep their input order (pandas' unstable sort leaves
 *    that order host-dependent; everything else is bit-identical to the reference).
 * ===================================================================================== */
int recnn_csr_workspace_bytes(int64_t n_rows, int64_t* bytes);
/* in : user_ids / item_keys / timestamps int64[n_rows], ratings double[n_rows] (raw, e.g. 0.5..5);
 *      map_keys (ascending) / map_vals int64[n_map]: item key -> dense id, NULL = keep the keys;
 * out: items_out int64[n_rows] (-1 where the key is not in the map), ratings_out double[n_rows] = 2 (r - 2.5),
 *      users_out int64[<= n_rows] ascending user ids, user_off_out int64[users + 1], order_out int64[n_rows] or NULL
 *      (output row i is input row order_out[i]), mapped_rows_out int64[n_rows] or NULL (dense id of INPUT row i);
 *      host_counts[3] (HOST memory) = { users, rows with an unmapped item key, key bits sorted }.
 * Synchronises the stream twice (key ranges decide the number of radix passes; the user count is returned). */
int recnn_csr_build(const int64_t* user_ids, const int64_t* item_keys, const double* ratings, const int64_t* timestamps,
                    int64_t n_rows, const int64_t* map_keys, const int64_t* map_vals, int n_map, int64_t* items_out,
                    double* ratings_out, int64_t* users_out, int64_t* user_off_out, int64_t* order_out, int64_t* mapped_rows_out,
                    int64_t* host_counts, void* workspace, int64_t workspace_bytes, void* stream);

/* Packed batch buffers (float[x_rows, ld_x]) + reward/done (float[max_rows]). */
int recnn_engine_bind_batch(recnn_engine* e, float* xs, float* xn, float* reward, float* done);

/* Device-resident replay sampler: when bound, every step first builds its batch on the GPU
 * (recnn_frame_plan + recnn_frame_gather straight into the packed rows) from
 * perm[cursor*users_per_batch .. +users_per_batch), cut to the step's `rows`, and then advances
 * the cursor (mod n_batches).  This makes the whole hot path -- sampler, gather, networks,
 * optimizer, soft update -- one launch sequence / one hipGraph.
 *   replaces the DataLoader + collate of recnn/data/env.py:225-252. */
/* Plan table of a whole epoch permutation: plan[b * rows + r] = (CSR offset of the window start of row r of batch b) << 1 |
 * (1 if that window is its user's last: done), -1 where batch b has fewer than r + 1 rows.  One workgroup per batch.
 *   replaces, ahead of time, the per-step users -> offsets -> prefix scan -> row search of recnn_frame_gather. */
int recnn_frame_plan_rows(const int64_t* user_off, const int32_t* perm, int users_per_batch, int n_batches, int frame,
                          int rows, int64_t* plan, void* stream);
/* Plan table of a DENSE epoch: the windows of the users seq[0 .. n_seq) (store slots) concatenated in that order -- every L - F
 * windows of every user, as the reference's collate builds them (recnn/data/utils.py:161-187) -- and cut into batches of `rows`
 * rows: plan[g] for global row g, same encoding as recnn_frame_plan_rows, done = 1 at each user's last window wherever it falls.
 * skip0: windows of seq[0] already consumed (the previous epoch's leftover is carried into this one).  Slots past the epoch's last
 * whole batch repeat its first batches.  row_off: scratch int32[n_seq + 1] (prefix sums of the window counts; the total must fit
 * 31 bits).  Bind the table with recnn_sampler.plan / plan_rows = rows; the sampler's perm / users_per_batch are then unused. */
int recnn_frame_plan_dense(const int64_t* user_off, const int32_t* seq, int n_seq, int skip0, int frame, int rows, int32_t* row_off,
                           int64_t n_rows, int64_t* plan, void* stream);
typedef struct recnn_sampler {
  const int32_t* items; const float* ratings; const int64_t* user_off;  /* CSR replay store */
  const int32_t* perm;        /* epoch permutation of store user slots, int32[n_batches*users_per_batch] */
  int users_per_batch;
  int n_batches;
  int frame, emb_dim;
  const float* table;         /* float[n_items, emb_dim] */
  int32_t* row_off;           /* scratch int32[users_per_batch + 1] */
  int32_t* cursor;            /* device int32: next batch index */
  const int64_t* plan;        /* optional: int64[n_batches * plan_rows] made by recnn_frame_plan_rows for THIS perm (NULL = the
                                 gather plans its rows itself); must be re-made whenever perm changes */
  int plan_rows;              /* rows per batch the plan covers (steps of at most this many rows use it) */
} recnn_sampler;
int recnn_engine_bind_sampler(recnn_engine* e, const recnn_sampler* h_sampler);  /* NULL unbinds */
/* Who draws from a bound sampler: graph replays (recnn_engine_graph_run), the data-parallel phase graphs and
 * recnn_engine_profile always do.  Eager calls (recnn_engine_step / value_grads / finish) run on the BOUND batch --
 * `algo.update(env.test_batch(), learn=False)` between two replays evaluates the batch it was given and leaves the
 * sampler cursor alone -- unless on = 1 is set here (eager data-parallel phases). */
int recnn_engine_sampler_eager(recnn_engine* e, int on);

/* External inputs for parity runs: masks uint8[n_masks][max_rows][hidden] (6 DDPG, 8 TD3, in the
 * reference's consumption order), noise float[max_rows][action_dim] (TD3, unclipped). */
int recnn_engine_bind_external(recnn_engine* e, const uint8_t* masks, const float* noise);

int recnn_engine_set_hyper(recnn_engine* e, const recnn_hyper* h_hyper);
/* switch the dropout-mask source of the learning nets (RECNN_MASK_*); invalidates built graphs. */
int recnn_engine_set_mask_mode(recnn_engine* e, int mask_mode);

/* Re-derive the compute-layout shadow of a network from its canonical arena (call after the
 * canonical parameters were changed by anything other than the engine). */
int recnn_engine_refresh(recnn_engine* e, int net, void* stream);

/* Set optimizer step counters (1-based count of steps already taken). */
int recnn_engine_set_counters(recnn_engine* e, int policy_t, int value1_t, int value2_t, int step);

/* One full update on the bound batch (rows valid rows).
 *   learn      : 0 = losses only (reference learn=False), 1 = update
 *   step       : the caller's step counter (policy step iff step % policy_every == 0)
 *   fused_optim: 1 = run Adam + soft update inside; 0 = stop after gradients (phases below)
 * Losses are written to the device loss buffer; fetch with recnn_engine_read_losses. */
int recnn_engine_step(recnn_engine* e, int rows, int learn, int step, void* stream);

/* Phase API (external optimizers, data-parallel all-reduce between phases):
 *   value_grads  : [sampler gather if bound,] TD target, critic forward/backward -> value grads in the bound
 *                  grad arenas
 *   value_apply  : Adam on the critic(s) (+ fused soft update when `soft`), shadows refreshed
 *   policy_grads : actor forward, critic forward, policy loss; with `backward` also the actor
 *                  gradient (reduced into the bound grad arena, NOT yet clipped)
 *   policy_apply : L1 clip quirk + Adam on the actor (+ soft update when `soft`)
 * grad_scale multiplies gradients inside the apply phases (1/world_size after an all-reduce). */
int recnn_engine_value_grads(recnn_engine* e, int rows, int learn, void* stream);
int recnn_engine_value_apply(recnn_engine* e, int soft, float grad_scale, void* stream);
int recnn_engine_policy_grads(recnn_engine* e, int rows, int backward, void* stream);
int recnn_engine_policy_apply(recnn_engine* e, int soft, float grad_scale, void* stream);
/* the same pieces for callers that run their own optimizer: */
int recnn_engine_clip_policy_grads(recnn_engine* e, float grad_scale, void* stream);  /* g *= coef */
int recnn_engine_soft_update(recnn_engine* e, int net, int target_net, float tau, void* stream);
/* closes a step driven through the phase API: reduces the loss partials into the loss buffer
 * and advances the device step / optimizer counters. */
int recnn_engine_finish(recnn_engine* e, int rows, int value_stepped, int policy_stepped, void* stream);

/* Capture `recnn_engine_step` for a fixed row count into hipGraphs -- one ordinary step, one policy step, and a family
 * of RUN graphs (k ordinary steps; a policy step + k ordinary steps, k < policy_every; whole policy cycles up to 64
 * steps, see tuning.graph_run) -- and replay `n_steps` consecutive steps starting at `first_step`: ANY
 * (first_step, n_steps) is covered with at most n_steps / policy_every + 2 graph launches (policy_every <= 17; longer
 * cycles compose power-of-two stretches).  Inside a run graph the device counters are ticked once at its end, the sampler + gather of step t+1 and the policy-loss forward of step t ride on
 * other launches (tuning.pregather, tuning.defer_policy_fwd), and every step's losses land in the history ring
 * (recnn_engine_read_counters). */
int recnn_engine_graph_build(recnn_engine* e, int rows, void* stream);
int recnn_engine_graph_run(recnn_engine* e, int first_step, int n_steps, void* stream);
/* Optional: a run graph made to order for requests of exactly n_steps (2..64) steps whose first step number is congruent
 * to first_step modulo policy_every; recnn_engine_graph_run then serves such requests with ONE graph launch instead of
 * composing them from the family.  Up to 4 are kept (oldest replaced); dropped with the other graphs. */
int recnn_engine_graph_prepare(recnn_engine* e, int first_step, int n_steps, void* stream);

/* Runs n_steps eager steps with a hipEvent pair around every kernel launch and returns, per
 * launch slot, the average device time in milliseconds (h_ms[i]), its name (h_names[i], static
 * strings) and, for GEMM slots, the algorithmic FLOPs of one launch (h_flops[i], 0 otherwise).
 * Kernels whose relaunch does not change state (everything but Adam and the loss/counter kernel) are issued
 * 8 times back to back inside their event pair and the time divided by 8, which amortises the event overhead.
 * *h_n is in: capacity, out: slots used.  Policy and non-policy steps have different slot
 * lists; only steps with (step % policy_every == 0) == policy_steps are run and averaged.  policy_steps = 2 profiles CYCLE
 * MODE (what run graphs of >= tuning.cycle_min_len steps replay): the gather of one policy cycle's batches, the frozen
 * networks on all of them, then one ordinary step of the cycle on the split forward. */
int recnn_engine_profile(recnn_engine* e, int rows, int policy_steps, int n_steps, void* stream,
                         float* h_ms, double* h_flops, const char** h_names, int* h_n);

/* Data-parallel replay: the step cut into four hipGraphs around the two gradient all-reduces the caller issues
 * (torch.distributed / RCCL) on the same stream:
 *   0: batch, all forwards, critic backward, slab reduction   -> caller all-reduces the critic gradient arena(s)
 *   1: non-policy step: critic Adam, policy loss, finish
 *   2: policy step: critic Adam (+soft update), policy loss, actor backward -> caller all-reduces the actor arena
 *   3: policy step: L1 clip + actor Adam (+soft update), finish
 *   4: (overlap_actor = 1 only) the actor forward alone; graph 0 then omits it, and the caller launches graph 4
 *      right after starting the critic all-reduce so that the collective's latency hides behind it
 *   5: graph 1 of step t followed by graph 0 of step t+1 in ONE graph (one graph launch per step instead of two)
 *   6: graph 3 of step t followed by graph 0 of step t+1
 * `which` = kind + 8 * set, set = batch buffer set holding step t's batch: recnn_engine_dp_sets() returns 2 when
 * consecutive steps alternate between two sets (then the sampler + gather of step t+1 rides on step t's critic
 * optimizer launch inside graph 5), else 1 (always pass set 0).  A run of n steps:
 *   launch 0;  per step: all-reduce critic arena(s); ordinary step: launch 5 (1 on the last step);
 *   policy step: launch 2, all-reduce actor arena, launch 6 (3 on the last step);  set ^= 1 after 5 / 6 if 2 sets.
 * grad_scale (1/world_size) is baked into the graphs. */
int recnn_engine_dp_graph_build(recnn_engine* e, int rows, float grad_scale, int overlap_actor, void* stream);
int recnn_engine_dp_graph_launch(recnn_engine* e, int which, void* stream);
int recnn_engine_dp_sets(recnn_engine* e);

/* ---- device-side gradient exchange (csrc/comm.hip; SURVEY.md 8(b) `dp_allreduce_flat`, 5 "two-shot P2P all-reduce").
 * New functionality (the reference is single-process); what it preserves is the gradient of the GLOBAL batch mean that
 * recnn/nn/update/ddpg.py:74-87 / td3.py:95-123 compute on one concatenated batch: sum over ranks x 1 / world.
 *
 * Every rank (one process per GPU, up to 8 = one node) creates a communicator able to carry `max_floats` floats, exports its
 * peer buffer as an opaque handle of recnn_comm_handle_bytes() bytes (a hipIpcMemHandle_t), exchanges the handles out of band
 * (torch.distributed all_gather_object, MPI, a file: anything) and connects with the `world` handles in rank order.  After
 * that recnn_dp_allreduce_flat sums data[0 .. n) over the ranks IN PLACE with ONE kernel launch on `stream`: reduce-scatter by
 * direct peer reads, all-gather by direct peer writes over xGMI, ranks added in the order 0 .. world-1 (every rank gets the
 * same bits; world 2 == any other order).  The launch can be captured: epochs are advanced on the device.  Every rank must
 * issue the same sequence of collectives.  A peer that does not arrive within 4 s sets an error word instead of hanging the
 * GPU: recnn_comm_status then returns RECNN_E_STATE and names the rank(s).
 *
 * recnn_engine_set_comm attaches a connected communicator to an engine: from then on every step -- eager or inside the run
 * graphs (rebuild them) -- all-reduces the critics' flat gradient arenas each step and the actor's on policy steps in-stream
 * and the optimizers step on grad * grad_scale (1 / world); the L1 clip quirk acts on the reduced actor gradient.  A
 * data-parallel run is then recnn_engine_graph_run on every rank: no host code between the phases of a step.  NULL detaches. */
typedef struct recnn_comm recnn_comm;
int recnn_comm_create(int world, int rank, int64_t max_floats, recnn_comm** out);
int64_t recnn_comm_handle_bytes(void);
int recnn_comm_export(recnn_comm* c, void* handle_out, int64_t bytes);
int recnn_comm_connect(recnn_comm* c, const void* handles, int64_t bytes_each);
int recnn_dp_allreduce_flat(recnn_comm* c, float* data, int64_t n, void* stream);
int recnn_comm_status(recnn_comm* c, int32_t* timed_out_ranks, int32_t* epoch);
/* bound (ms, default 4000) of every peer wait of the collectives launched or captured afterwards; clear a reported time-out */
int recnn_comm_set_timeout_ms(recnn_comm* c, int ms);
int recnn_comm_clear_status(recnn_comm* c);
void recnn_comm_destroy(recnn_comm* c);
int recnn_engine_set_comm(recnn_engine* e, recnn_comm* comm, float grad_scale);
/* Settings of ONE communicator (round 6: no process-wide communicator state is left): recnn_comm_create_ex takes the memory kind of the
 * peer buffers -- 0 fine-grained (default, = recnn_comm_create), 1 uncached, 2 ordinary device memory; recnn_comm_set_workgroups the
 * workgroups per collective launch made or captured afterwards (default 128; ranks that share ONE GPU in tests need every rank's
 * collective resident at once: 32). */
int recnn_comm_create_ex(int world, int rank, int64_t max_floats, int memory_kind, recnn_comm** out);
int recnn_comm_set_workgroups(recnn_comm* c, int n);

/* ---- per-engine tuning.  Every field selects among schedules / kernel tilings that produce the SAME numbers (the GPU suite runs
 * under several of them); nothing here is process-wide: two engines of one process can run different schedules.
 * recnn_engine_tuning_init fills the defaults, recnn_engine_set_tuning copies a (clamped) tuning into an engine and drops its
 * graphs.  In-launch hand-offs of the fused forward wait with a bound: a wait that runs out sets a device error word and
 * recnn_engine_read_losses / read_counters then return RECNN_E_STATE (debug hooks that provoke this: csrc/recnn_hip_debug.h). */
typedef struct recnn_engine_tuning {
  int fused_mlp;            /* 1: bf16 engines with hidden <= 256 run the networks of a step as ONE fused row-panel launch (csrc/mlps.hip)
                               when a launch has >= 3 of them, 2: always, 0: layer-by-layer GEMM launches */
  int chain_target_critic;  /* 1: the target critics run inside the fused launch (producer workgroups + flag hand-off), 0: after it */
  int bwd_panel;            /* where the critic head + first backward GEMM run: 2 inside the fused forward launch (unit backward
                               tensors, per-row seed applied by the dW launch), 1 one row-panel launch (bwd.hip), 0 head + dX launches */
  int policy_chain;         /* 1: policy steps run the gradient chain from the policy loss into the actor as ONE row-panel launch */
  int split_fwd;            /* 0: fused forward everywhere; 1: run graphs of >= cycle_min_len steps run in cycle mode (a policy cycle's
                               batches gathered at once, frozen networks applied to all of them: mlpf.hip; per-step launches = split
                               forward of the learning critics: l1gemm.hip + mlpt.hip); 2: split forward and cycle mode everywhere */
  int cycle_min_len;        /* default 20 (round 5; 30 before) */
  int cycle_min_seg;        /* cycle mode: segments shorter than this step through the fused forward (default 4 since round 6: the per-step launches of cycle mode got cheaper; 5 in round 5, 3 before) */
  int frozen_fused;         /* cycle mode: 1 each frozen network as one launch of 128-row panels, 0 tiled layer 1 + later layers */
  int frozen_gemm;          /* ... whose later layers run as tiled GEMMs (1) or row-panel tails (0) */
  int graph_run;            /* steps per run graph: -1 as many whole policy cycles as fit 64 steps, 0 single-step graphs only */
  int pregather;            /* 1: inside a run graph the sampler + gather of step t+1 ride on step t's critic optimizer launch */
  int defer_policy_fwd;     /* 1: ... and the policy-loss forward of step t on step t+1's forward launch(es) */
  int sampler_f32_rows;     /* 1: an engine that samples its own batches also fills the bound fp32 packed rows (inspection) */
  int dw_splits;            /* batch splits (gradient slabs) of the layer-1 dW GEMM, 1..8 */
  int comm_fused;           /* data parallel: 1 the critics' gradient exchange runs inside their optimizer launch, 0 as its own launches */
  int l1_big;               /* tile of the cycle-batched layer-1 GEMMs: 1 = 128 x 128 (default), 2 = 128 x 64 */
  int gemm_variant;         /* register-staged GEMM tile: -1 per launch, 0 = 64 x 64, 1 = 32 x 64 with a 2x longer k stage */
  int gemm_v0_threshold;    /* ... the per-launch choice takes 64 x 64 from this many tiles on (512) */
  int gemm_dma;             /* 1: forward GEMMs on compute-type operands use the LDS-DMA ring kernel */
  int gemm_dma_depth;       /* 1: 5-stage ring for launches of <= 320 tiles */
  int gemm_dma_waves;       /* 8 | 4 waves per workgroup of the LDS-DMA forward kernel */
  int gemm_waves;           /* 8 | 4 waves per workgroup of the register-staged dX kernel */
  int dw_dma;               /* bf16 dW: 0 register-staged loader, 1..7 LDS-DMA + transpose reads with (rows per stage, ring slots) =
                               (128,2) (64,2) (64,3) (64,4) (32,2) (32,4) (32,3); default 2 */
  int x3_tail;              /* split-bf16 engines (hidden 256, action 128): 1 layers 2 + 3 of a step's networks as row-panel launches
                               that keep h2 on chip (csrc/x3tail.hip), 0 grouped GEMM launches per layer */
  int x3_fwd;               /* split-bf16 forward GEMM kernel: 2 (default) wave-specialised -- loader waves + consumer waves (csrc/gemm.hip
                               x3_fwd_ws_kernel), 11 only its 64 x 128-tile launches, 0 every wave loads and multiplies (round 4) */
  int dw_fuse;              /* 1 (default): on the single-GPU bf16 step whose backward tensors come from the split forward (cycle mode), the critics'
                               weight-gradient GEMMs carry the optimizer in their epilogue -- ONE launch, no gradient slabs (csrc/dwadam.hip);
                               0: dW launch + optimizer launch.  Bit-identical results */
  int tail_half;            /* 1 (default): the learning critic's tail launch (csrc/mlpt.hip) runs 16-row panels -- twice the workgroups, half the
                               per-workgroup epilogue work (its phases are bound by the CU's vector issue); 0: 32-row panels.  Bit-identical */
  int l1_ws;                /* 1: the per-step layer-1 GEMM (csrc/l1gemm.hip, 64 x 64 tiles) runs with 4 loader + 8 consumer waves; 0: all 16 waves
                               load and multiply.  Bit-identical */
  int frozen_half;          /* 1: a cycle segment's frozen-network launch (csrc/mlpf.hip) runs 64-row workgroups instead of 128-row ones while
                               it then still fits one round of workgroups (short segments: a request that starts or ends inside a policy
                               cycle); 0: always 128 rows.  Bit-identical */
  int reserved[3];
} recnn_engine_tuning;
void recnn_engine_tuning_init(recnn_engine_tuning* h_t);
int recnn_engine_set_tuning(recnn_engine* e, const recnn_engine_tuning* h_t);
int recnn_engine_get_tuning(recnn_engine* e, recnn_engine_tuning* h_t);

/* Device counters {steps finalized, actor optimizer steps, critic 1 steps, critic 2 steps} (synchronises the stream).
 * The debug view "loss_ring" ([1024][4] fp32: value1, value2 / policy, policy per recnn_engine_read_losses' layout)
 * holds the losses of the last 1024 steps at index (step counter value of that step) mod 1024 -- including every
 * step replayed inside a run graph, whose per-step partial sums are kept and reduced at the end of the run. */
int recnn_engine_read_counters(recnn_engine* e, int32_t* h_out4, void* stream);
/* Host copy of the last step's losses (synchronises `stream`):
 * DDPG: {value, policy}; TD3: {value1, value2, policy}.  h_out has room for 4 floats. */
int recnn_engine_read_losses(recnn_engine* e, float* h_out, void* stream);

/* 1 if the last step left UNIT backward tensors in the debug views "critic1_dz2" / "critic1_dz1" (dz / d: the per-row
 * loss seed "delta1" is applied by the consumers inside the dW launch), 0 if they hold dz itself. */
int recnn_engine_unit_backward(recnn_engine* e);
/* Debug / test access to intermediate device buffers by name
 * ("next_action", "expected", "q1", "gen_action", ...).  Returns NULL if unknown. */
const void* recnn_engine_buffer(recnn_engine* e, const char* name, int64_t* h_rows, int64_t* h_cols,
                                int64_t* h_ld, int* h_is_f32);

/* =====================================================================================
 * 5. Batched exact top-K action scoring ("next" row)
 *    replaces the retrieval step after the actor: faiss IndexFlatL2 / IndexFlatIP / IP on normalised rows
 *    (examples/streamlit_demo.py:190-204), the per-item scipy loop `rank` (streamlit_demo.py:207-231) and
 *    MilvusConnection.search (recnn/data/db_con.py:45-56).
 *    metric: 0 = IP (q.t, descending), 1 = L2 (|q-t|^2, ascending), 2 = COS (q.t/|t|, descending).
 *    Ties are broken towards the smaller item id.  emb_dim must be 128, k <= 64.
 * ===================================================================================== */
enum { RECNN_METRIC_IP = 0, RECNN_METRIC_L2 = 1, RECNN_METRIC_COS = 2 };
/* per-item auxiliary array float[n_items] needed by L2 (|t|^2) and COS (1/|t|); build once per table */
int recnn_topk_item_aux(const float* table, int n_items, int emb_dim, int metric, float* aux, void* stream);
int recnn_topk_workspace_bytes(int n_queries, int k, int64_t* h_bytes);
/* out_dist float[n_queries, k], out_ids int64[n_queries, k] (faiss / Milvus result layout) */
int recnn_topk_search(const float* queries, int64_t ld_q, int n_queries, const float* table, int n_items, int emb_dim,
                      int metric, const float* item_aux, int k, float* out_dist, int64_t* out_ids, void* workspace,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECNN_HIP_H */
