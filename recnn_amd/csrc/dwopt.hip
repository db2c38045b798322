// dwopt.hip -- critic weight gradients + optimizer in ONE launch (bf16, gfx950).
//
// Replaces, on the fused single-GPU step, three launches and their round trip through HBM (round 2: gemm_dw_dma_kernel wrote 8
// split-batch fp32 slabs = 11.6 MB, apply_kernel read them back: 12.3 + 10.9 us, PMC traffic 28 + 26 MB for 12 MB of
// algorithmic bytes):
//   autograd's mm(dZ^T, X) of the critic's linear1 / linear2            recnn/nn/update/misc.py:42-43 (value_loss.backward())
//   torch.optim.Adam.step (as injected by the reference's users)         recnn/nn/update/misc.py:44
//   recnn/utils/misc.py:1-5 soft_update on policy steps                  recnn/nn/update/ddpg.py:95-97
// A workgroup owns a 64 x 64 tile of dW and contracts the WHOLE batch for it (dw_tile.h; 4-slot ring of 128-row stages: 96 KB
// in flight per CU, one workgroup per CU), so the finished gradient never leaves the chip: the tile goes through LDS (to turn
// the MFMA accumulator layout into row-contiguous pairs) and the same threads read p / m / v, apply optim.h's opt_elem and
// write p / m / v, the bf16 compute shadow (and the soft-updated target + its shadow on policy steps).  Fixed summation order
// (k ascending, one accumulator chain per element): deterministic, and bit-identical to "gradient arena + apply_kernel"
// (mode DWOPT_GRAD, then optim.hip) because both apply the same contraction-pinned opt_elem.
// The small tensors (w3, b1, b2, b3: 769 elements) are column sums over all rows of d_r * {h2, u2, U} and sum_r d_r: "vector"
// workgroups of the same launch (first in the launch order) reduce them over the whole batch in a fixed order and update
// them in place as well -- no partial slabs, no second launch.
#include "dw_tile.h"
#include "dwopt.h"
#include "gather_dev.h"

namespace {
constexpr int SUB = 128, NS = 4;
// NWK = k-groups of 4 waves per workgroup (template parameter of the kernel): 4 -> 16 waves, 4 per SIMD (default)
constexpr int SCALE_CAP = 4096;                              // per-row loss seeds staged in LDS up to this many batch rows
constexpr int RING_BYTES = NS * SUB * 256;                   // 128 KB
constexpr int SCAL_OFF = RING_BYTES + SCALE_CAP * 4;         // the launch's optimizer scalars (OptScalars), one copy per workgroup
constexpr int LDS_BYTES = SCAL_OFF + 64;                     // 144 KB: one workgroup per CU
constexpr int TP = 68;                                       // fp32 pitch of the 64 x 64 tile image in LDS (bank spread)
constexpr int VEC_COLS = 32;                                 // columns per vector workgroup
constexpr int W1i = 0, B1i = 1, W2i = 2, B2i = 3, W3i = 4, B3i = 5;

// the parameters of one element pair / element: everything loaded before the first use
struct Elem2 { float2 p, m, v, tp, sl; };

// Finish `cnt` (1 or 2) consecutive elements starting at canonical arena index e with gradient (g0, g1).
// sh: shadow element index of the first element or -1; vec2: 8-byte accesses allowed (e even, cnt == 2).
__device__ __forceinline__ void finish_elems(const ApplyArgs& a, const OptScalars& S, const int mode, const int64_t e, const int cnt,
                                             const bool vec2, const float g0, const float g1, const Elem2& x, const int64_t sh) {
  if (mode == DWOPT_GRAD) {
    if (a.g_out) {
      if (vec2) *(float2*)(a.g_out + e) = make_float2(g0, g1);
      else { a.g_out[e] = g0; if (cnt > 1) a.g_out[e + 1] = g1; }
    }
    return;
  }
  float p[2] = {x.p.x, x.p.y}, m[2] = {x.m.x, x.m.y}, v[2] = {x.v.x, x.v.y}, sl[2] = {x.sl.x, x.sl.y}, tp[2] = {x.tp.x, x.tp.y};
  const float g[2] = {g0, g1};
#pragma unroll
  for (int j = 0; j < 2; ++j) opt_elem(a, S, g[j], a.grad_scale, p[j], m[j], v[j], sl[j]);
  if (a.tgt_p) {
#pragma unroll
    for (int j = 0; j < 2; ++j) tp[j] = soft_elem(tp[j], p[j], a.tau);
  }
  if (vec2) {
    *(float2*)(a.m + e) = make_float2(m[0], m[1]);
    *(float2*)(a.v + e) = make_float2(v[0], v[1]);
    *(float2*)(a.p + e) = make_float2(p[0], p[1]);
    if (S.la_sync) *(float2*)(a.slow + e) = make_float2(sl[0], sl[1]);
    if (a.g_out) *(float2*)(a.g_out + e) = make_float2(g0, g1);
    if (a.tgt_p) *(float2*)(a.tgt_p + e) = make_float2(tp[0], tp[1]);
    if (sh >= 0) {
      if (a.shadow) *(uint32_t*)((bf16_t*)a.shadow + sh) = pack_bf2(p[0], p[1]);
      if (a.tgt_p && a.tgt_shadow) *(uint32_t*)((bf16_t*)a.tgt_shadow + sh) = pack_bf2(tp[0], tp[1]);
    }
  } else {
    for (int j = 0; j < cnt; ++j) {
      a.m[e + j] = m[j]; a.v[e + j] = v[j]; a.p[e + j] = p[j];
      if (S.la_sync) a.slow[e + j] = sl[j];
      if (a.g_out) a.g_out[e + j] = g[j];
      if (a.tgt_p) a.tgt_p[e + j] = tp[j];
      if (sh >= 0) {
        if (a.shadow) ((bf16_t*)a.shadow)[sh + j] = f2bf(p[j]);
        if (a.tgt_p && a.tgt_shadow) ((bf16_t*)a.tgt_shadow)[sh + j] = f2bf(tp[j]);
      }
    }
  }
}

// (the Lookahead slow weights are fetched whenever the optimizer is Ranger -- whether THIS step synchronises them is one of
// the step scalars, which the early loads do not wait for)
__device__ __forceinline__ Elem2 load_elems(const ApplyArgs& a, const int mode, const int64_t e, const int cnt, const bool vec2) {
  Elem2 x;
  x.p = x.m = x.v = x.tp = x.sl = make_float2(0.f, 0.f);
  if (mode != DWOPT_APPLY || cnt <= 0) return x;
  const bool slow = a.opt_kind == RECNN_OPT_RANGER && a.slow;
  if (vec2) {
    x.p = *(const float2*)(a.p + e);
    x.m = *(const float2*)(a.m + e);
    x.v = *(const float2*)(a.v + e);
    if (a.tgt_p) x.tp = *(const float2*)(a.tgt_p + e);
    if (slow) x.sl = *(const float2*)(a.slow + e);
  } else {
    x.p.x = a.p[e]; x.m.x = a.m[e]; x.v.x = a.v[e];
    if (a.tgt_p) x.tp.x = a.tgt_p[e];
    if (slow) x.sl.x = a.slow[e];
    if (cnt > 1) {
      x.p.y = a.p[e + 1]; x.m.y = a.m[e + 1]; x.v.y = a.v[e + 1];
      if (a.tgt_p) x.tp.y = a.tgt_p[e + 1];
      if (slow) x.sl.y = a.slow[e + 1];
    }
  }
  return x;
}

// ---- vector workgroups: sum over ALL batch rows of d_r * X[r][c] for 32 columns of X in {h2, u2, U}  (-> w3, b2, b1), or
// sum_r d_r (-> b3).  NT threads: 4 per row (16 bytes = 8 columns each), NT / 4 rows per round, every round's load in flight
// before the first use; fixed order: a thread walks its rows upwards, the row groups are added in 8 runs of consecutive
// groups, the runs 0, 1, ... 7.
template <int NT>
__device__ __forceinline__ void vec_role(const DwOpt& o, const int net, const int role, unsigned char* smem, unsigned long long* trow) {
  constexpr int RGN = NT / 4;                     // row groups
  const DwVecProb& V = o.v[net];
  const ApplyArgs& a = o.a[net];
  const int tid = threadIdx.x;
  const int rows = V.rows;
  float* dl = (float*)smem;                       // d_r of every row (when it fits)
  float* red = dl + SCALE_CAP;                    // [RGN row groups][32 columns]
  float* red2 = red + RGN * VEC_COLS;             // [8 runs][32 columns]
  const bool staged = rows <= SCALE_CAP;
  if (staged)
    for (int i = tid; i < rows; i += NT) dl[i] = V.delta[i];
  OptScalars* Sp = (OptScalars*)(smem + SCAL_OFF);
  if (tid == NT - 1) *Sp = opt_scalars(a);        // (a table load when the engine precomputed the step scalars)
  __syncthreads();
  if (trow) trow[1] = __builtin_amdgcn_s_memtime();
  const int nblk = (V.H + VEC_COLS - 1) / VEC_COLS;
  if (role == 3 * nblk) {                         // b3 <- sum_r d_r
    float s = 0.f;
    for (int r = tid; r < rows; r += NT) s += staged ? dl[r] : V.delta[r];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
      float g = 0.f;
      for (int w = 0; w < NT / 64; ++w) g += red[w];
      const OptScalars S = *Sp;
      const int64_t e = o.seg[net][B3i].p_off;
      const Elem2 x = load_elems(a, o.mode, e, 1, false);
      finish_elems(a, S, o.mode, e, 1, false, g, 0.f, x, -1);
    }
    return;
  }
  const int kind = role / nblk, cb = role % nblk;
  const bf16_t* X = (const bf16_t*)(kind == 0 ? V.h2 : (kind == 1 ? V.u2 : V.U));
  const int rg = tid >> 2, c0 = cb * VEC_COLS + (tid & 3) * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c0 < V.H) {                                 // (H is a multiple of 8: 16-byte row segments stay inside the row)
    for (int r0 = rg; r0 < rows; r0 += RGN * 8) {
      uint4 x[8];
      float d[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + RGN * j;
        const int rc = min(r, rows - 1);
        x[j] = *(const uint4*)(X + (int64_t)rc * V.ldh + c0);
        d[j] = r < rows ? (staged ? dl[rc] : V.delta[rc]) : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t w[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[2 * q] = fmaf(d[j], bf2f((bf16_t)(w[q] & 0xFFFFu)), acc[2 * q]);
          acc[2 * q + 1] = fmaf(d[j], bf2f((bf16_t)(w[q] >> 16)), acc[2 * q + 1]);
        }
      }
    }
  }
  if (trow) trow[2] = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rg * VEC_COLS + (tid & 3) * 8 + j] = acc[j];
  __syncthreads();
  if (tid < 8 * VEC_COLS) {                       // run `part` = row groups [part * RGN / 8, (part + 1) * RGN / 8), upwards
    const int c = tid & (VEC_COLS - 1), part = tid / VEC_COLS;
    float g = 0.f;
    for (int q = part * (RGN / 8); q < (part + 1) * (RGN / 8); ++q) g += red[q * VEC_COLS + c];
    red2[part * VEC_COLS + c] = g;
  }
  __syncthreads();
  if (tid < VEC_COLS) {
    const int c = cb * VEC_COLS + tid;
    if (c < V.H) {
      const TensorSeg& T = o.seg[net][kind == 0 ? W3i : (kind == 1 ? B2i : B1i)];
      const int64_t e = T.p_off + c;
      const Elem2 x = load_elems(a, o.mode, e, 1, false);
      float g = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) g += red2[q * VEC_COLS + tid];
      const OptScalars S = *Sp;
      finish_elems(a, S, o.mode, e, 1, false, g, 0.f, x, -1);   // (a critic's w3 / biases have no compute shadow)
    }
  }
}
}  // namespace

#define DW_STAMP(i) do { if (trow) trow[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
template <int NWK> __global__ __launch_bounds__(256 * NWK) void dw_opt_kernel(const GemmBatch batch, const DwOpt o, const GatherArgs ga, const int n_gather,
                                                                                 unsigned long long* trace) {
  constexpr int NT = 256 * NWK;
  unsigned long long* trow = (trace && threadIdx.x == 0) ? trace + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  DW_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
  // launch order = dispatch order: the two latency chains (vector sums, gather) first, the streaming tiles underneath
  int y = blockIdx.y;
  // the gather role is written for 256 threads: the other waves of such a workgroup leave at once (a workgroup barrier only
  // counts the waves that are still alive)
  if (n_gather > 0 && y == 1 && threadIdx.x >= 256) return;
  if (y == 0) {
    const int nblk = (o.v[0].H + VEC_COLS - 1) / VEC_COLS;
    const int per_net = 3 * nblk + 1;
    if ((int)blockIdx.x < o.n_net * per_net) vec_role<NT>(o, blockIdx.x / per_net, blockIdx.x % per_net, dsmem, trow);
    DW_STAMP(7);
    return;
  }
  --y;
  if (n_gather > 0) {
    if (y == 0) {
      if ((int)blockIdx.x < n_gather) frame_gather_body<4, 4>(ga, blockIdx.x, dsmem);
      return;
    }
    --y;
  }
  const GemmProb& P = batch.p[y];
  const int nwg = P.tiles_m * P.tiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int tile_n = lid % P.tiles_n, tile_m = lid / P.tiles_n;
  const int m0 = tile_m * 64, n0 = tile_n * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = ((wave >> 1) & 1) * 32, wn0 = (wave & 1) * 32;
  const int fr = lane & 15, fg = lane >> 4;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int net = o.prob_net[y];
  const ApplyArgs& a = o.a[net];
  const TensorSeg& T = o.seg[net][o.prob_tensor[y]];
  // ---- this thread's share of the tile's parameters, requested BEFORE the operand stream starts: thread -> NP column pairs
  // (32 consecutive pairs of one row per half-wave: 256-byte segments); p / m / v (/ target / slow) were written by the
  // previous step's launch, so they can ride under the whole k loop instead of costing the epilogue a memory latency.
  // (They are the oldest entries of the wave's vector-memory queue: the k loop's counted waits stay valid.)
  const int cols = P.dw_valid_cols;
  const bool even = !((cols | P.dw_col_rot | T.sh_ld) & 1) && !(T.p_off & 1) && !(T.sh_off & 1);
  constexpr int NP = 2048 / NT;                 // pairs per thread
  Elem2 x[NP];
  int64_t ee[NP], sh[NP];
  int cnt[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = tid + NT * i, row = q >> 5, n = n0 + 2 * (q & 31), m = m0 + row;
    cnt[i] = (m < P.M && n < cols) ? (n + 1 < cols ? 2 : 1) : 0;
    int cc = n + P.dw_col_rot;
    if (cc >= cols) cc -= cols;
    if (!even && cnt[i] == 2 && cc + 1 >= cols) cnt[i] = -2;   // pair straddles the rotation wrap: two single elements
    ee[i] = T.p_off + (int64_t)m * cols + cc;
    sh[i] = T.sh_off >= 0 ? T.sh_off + (int64_t)m * T.sh_ld + n : -1;
    x[i] = load_elems(a, o.mode, ee[i], cnt[i] == -2 ? 1 : cnt[i], even && cnt[i] == 2);
  }
  // the launch's optimizer scalars: ONE thread fetches (or, without a table, computes: expm1 / sqrt in double, ~6 us) them
  // right after the ring's first stages are requested
  OptScalars* Sp = (OptScalars*)(dsmem + SCAL_OFF);
  dw_tile_accumulate<SUB, NS, NWK>(P, m0, n0, 0, P.seg[0].K, dsmem, SCALE_CAP, acc, [&] { if (tid == NT - 1) *Sp = opt_scalars(a); }, o.probe);

  DW_STAMP(1);
  // ---- the k-groups' partial accumulators -> LDS tile images [NWK][64][TP] (every wave is done with the ring after the barrier)
  __syncthreads();
  DW_STAMP(2);
  const int kq = wave >> 2;
  float* tile = (float*)dsmem;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[kq * (64 * TP) + (wm0 + tm * 16 + fg * 4 + r) * TP + wn0 + tn * 16 + fr] = acc[tm][tn][r];
  __syncthreads();
  DW_STAMP(3);

  // ---- optimizer on the tile
  const OptScalars S = *Sp;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = tid + NT * i, row = q >> 5, cp = q & 31;
    float2 g = *(const float2*)(tile + row * TP + 2 * cp);
#pragma unroll
    for (int k = 1; k < NWK; ++k) {              // fixed order: ((g0 + g1) + g2) + g3
      const float2 gk = *(const float2*)(tile + k * (64 * TP) + row * TP + 2 * cp);
      g.x += gk.x; g.y += gk.y;
    }
    if (cnt[i] == -2) {            // (odd layouts only: never the critic's)
      finish_elems(a, S, o.mode, ee[i], 1, false, g.x, 0.f, x[i], sh[i]);
      const int64_t e1 = T.p_off + (int64_t)(m0 + row) * cols;    // wraps to canonical column 0
      const Elem2 x1 = load_elems(a, o.mode, e1, 1, false);
      finish_elems(a, S, o.mode, e1, 1, false, g.y, 0.f, x1, sh[i] >= 0 ? sh[i] + 1 : -1);
    } else if (cnt[i] > 0) {
      finish_elems(a, S, o.mode, ee[i], cnt[i], even && cnt[i] == 2, g.x, g.y, x[i], sh[i]);
    }
  }
  DW_STAMP(7);
}

static unsigned long long* g_dwopt_trace = nullptr;   // [workgroup (y * gridDim.x + x)][8] shader-clock stamps (tools/dw_trace.py)
extern "C" void recnn_tune_dw_trace(void* p) { g_dwopt_trace = (unsigned long long*)p; }
static int g_dwopt_groups = 4;
static int g_dwopt_probe = 0;
extern "C" void recnn_tune_dw_probe(int bits) { g_dwopt_probe = bits; }   // timing experiments (results are garbage): dw_tile.h
void dwopt_set_groups(int g) { g_dwopt_groups = (g == 1 || g == 2) ? g : 4; }

int dwopt_init() {
  int rc = recnn_check_hip(hipFuncSetAttribute((const void*)dw_opt_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), "dw_opt attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)dw_opt_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), "dw_opt attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)dw_opt_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), "dw_opt attr");
  return rc;
}

// both operands bf16 in memory, 64-column tiles readable inside the row pitch (padding columns may hold anything: they only
// reach outputs that are never written)
bool dwopt_eligible(const GemmLaunch* L) {
  if (L->mode != GEMM_DW || L->dtype != RECNN_BF16 || L->a_f32 || L->b_f32) return false;
  for (int i = 0; i < L->nprob; ++i) {
    const GemmProb& p = L->batch.p[i];
    if (p.nseg != 1 || p.seg[0].K <= 0) return false;
    if (p.seg[0].lda < (p.M + 63) / 64 * 64 || p.seg[0].ldb < (p.N + 63) / 64 * 64) return false;
    if (((uintptr_t)p.seg[0].A | (uintptr_t)p.seg[0].B) & 15) return false;
  }
  return true;
}

int dwopt_launch(GemmLaunch* L, const DwOpt& o_in, const GatherArgs* pregather, hipStream_t s) {
  RECNN_REQUIRE(dwopt_eligible(L), "dwopt: needs bf16 k-strided operands with 64-column tiles inside the row pitch");
  RECNN_REQUIRE(o_in.mode == DWOPT_APPLY || o_in.mode == DWOPT_GRAD, "dwopt: bad mode");
  RECNN_REQUIRE(o_in.n_net >= 1 && o_in.n_net <= 2, "dwopt: one or two critics per launch");
  DwOpt o = o_in;
  o.probe = g_dwopt_probe;
  int maxwg = 0;
  for (int i = 0; i < L->nprob; ++i) {
    GemmProb& p = L->batch.p[i];
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_n = (p.N + 63) / 64;
    p.dw_splits = 1;
    const int nwg = p.tiles_m * p.tiles_n;
    if (nwg > maxwg) maxwg = nwg;
    RECNN_REQUIRE(o.prob_net[i] >= 0 && o.prob_net[i] < o.n_net && o.seg[(int)o.prob_net[i]][(int)o.prob_tensor[i]].cols == p.dw_valid_cols,
                  "dwopt: problem %d does not match its parameter tensor", i);
  }
  for (int c = 0; c < o.n_net; ++c) {
    apply_args_finish(&o.a[c]);
    RECNN_REQUIRE(o.v[c].H % 8 == 0 && o.v[c].ldh % 8 == 0 && o.v[c].rows == o.v[0].rows && o.v[c].H == o.v[0].H, "dwopt: bad vector problem");
    if (o.mode == DWOPT_APPLY) RECNN_REQUIRE(o.a[c].do_adam && o.a[c].p && o.a[c].m && o.a[c].v && o.a[c].t_ptr, "dwopt: optimizer state missing");
    else RECNN_REQUIRE(o.a[c].g_out, "dwopt: gradient arena missing");
  }
  const int vec_wg = o.n_net * (3 * ((o.v[0].H + VEC_COLS - 1) / VEC_COLS) + 1);
  if (vec_wg > maxwg) maxwg = vec_wg;
  GatherArgs ga;
  memset(&ga, 0, sizeof(ga));
  int ng = 0;
  if (pregather) {
    ga = *pregather;
    const size_t lds = frame_gather_lds_bytes(ga, 4);
    if (lds > 48 * 1024 || !ga.state_h || ga.state || (ga.emb % 4) || ga.rows <= 0) {
      recnn_set_error("dwopt+gather: needs the bf16-only gather with a tile that fits 48 KB of LDS");
      return RECNN_E_UNSUPPORTED;
    }
    ng = (ga.rows + 3) / 4;
    if (ng > maxwg) maxwg = ng;
  }
  const dim3 grid(maxwg, L->nprob + 1 + (ng > 0 ? 1 : 0), 1);
  if (g_dwopt_groups == 1) hipLaunchKernelGGL(dw_opt_kernel<1>, grid, dim3(256, 1, 1), LDS_BYTES, s, L->batch, o, ga, ng, g_dwopt_trace);
  else if (g_dwopt_groups == 2) hipLaunchKernelGGL(dw_opt_kernel<2>, grid, dim3(512, 1, 1), LDS_BYTES, s, L->batch, o, ga, ng, g_dwopt_trace);
  else hipLaunchKernelGGL(dw_opt_kernel<4>, grid, dim3(1024, 1, 1), LDS_BYTES, s, L->batch, o, ga, ng, g_dwopt_trace);
  return recnn_check_hip(hipGetLastError(), "dw_opt_kernel launch");
}
