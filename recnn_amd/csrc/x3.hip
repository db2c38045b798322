// x3.hip -- split-bf16 (RECNN_BF16X3, x3.h) GEMMs whose operands have the split columns along a TILE dimension: the
// backward products of the actor / critic MLPs (autograd's mm calls behind recnn/nn/models.py:66-73, :207-213):
//   dX  C[m, j] = gate( sum_n dZ[m, n] * W[n, j] )        dZ k-contiguous (split along k), W k-strided (split along j)
//   dW  C[i, j] =       sum_b dZ[b, i] * X[b, j]          both k-strided (split along i / j), batch split into fp32 slabs
// plus the fp32 <-> split-row conversions.  The forward product (both operands k-contiguous) is the LDS-DMA kernel of
// gemm.hip with the x3 fragment pairing.
//
// k-strided operands stay in memory order in LDS ([k row][physical columns], rows padded by 16 bytes) and their MFMA
// fragments are read with gfx950's ds_read_b64_tr_b16 (lane i of a 16-lane group receives column i of a [4 k][16 cols]
// block): a 16-column block of physical columns is all-hi or all-lo, so one logical 16 x 16 output block takes the hi and
// the lo block of each operand and three MFMAs.  Tiles are staged through registers (two LDS buffers, one barrier per
// 32-row k step): these launches are a small part of the step (profiles/NOTES_r01_r05.md 5d).
#include "gemm.h"
#include "x3.h"

typedef short v4s16 __attribute__((ext_vector_type(4)));

namespace {

constexpr int X3_LO_LO = 0;   // 1: also add a_lo * b_lo (a fourth MFMA per block; tests/x3_numerics.py "x4")

struct Frag { v4s16 lo, hi; };   // two transpose reads = the 8 k values of a lane (k = 8 fg + 0..3 | 4..7)

// transpose-read fragment of 16 physical columns [col0, col0 + 16) over k rows 0..31 of an LDS image with `pitch` bytes per row
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* s, int pitch, int col0, int fr, int fg) {
  Frag f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = fg * 8 + half * 4 + (fr >> 2);
    const v4s16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) v4s16*)(s + row * pitch + (col0 + (fr & 3) * 4) * 2));
    if (half == 0) f.lo = v; else f.hi = v;
  }
  return __builtin_bit_cast(bf16x8, f);
}

// The same fragment from a SWIZZLED image (pitch 256 bytes, no padding): the 32-byte chunk c of row r lives at chunk c ^ g(r),
// g(r) = (r & 3) | ((r >> 3) & 1) << 2.  A transpose read's lane group touches rows {0..3, 8..11} (+4, +16) and 32 bytes of
// each: with a plain pitch of 256 + 16 (the dW kernel's first layout) neighbouring rows overlap in 4 of their 8 banks -- a third
// of that kernel's LDS cycles were conflicts (SQ_LDS_BANK_CONFLICT, profiles/r04_x3_pmc.txt); g maps the 8 rows to the 8 disjoint
// bank windows.
__device__ __forceinline__ int swz32(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ bf16x8 tr_frag_swz(const unsigned char* s, int col0, int fr, int fg) {
  Frag f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = fg * 8 + half * 4 + (fr >> 2);
    const v4s16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) v4s16*)(s + row * 256 + (((col0 >> 4) ^ swz32(row)) << 5) + (fr & 3) * 8));
    if (half == 0) f.lo = v; else f.hi = v;
  }
  return __builtin_bit_cast(bf16x8, f);
}

__device__ __forceinline__ f32x4 mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x4 acc) {
  if (X3_LO_LO) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, acc, 0, 0, 0);
  return x3_mfma(ah, al, bh, bl, acc);
}

// ------------------------------------------------------------------ dX
// Workgroup: 64 rows x 32 logical columns (one [hi 32 | lo 32] group of W's physical columns); 4 waves as 2 (row halves) x 2
// (16-column blocks).  The contraction is short (n <= 256: a hidden layer), so the WHOLE k range of both operands is staged at
// once -- every global load of the workgroup in flight together, one barrier, then 8 k steps of MFMAs straight from LDS (the
// first version staged 32 n per step through two buffers and paid a memory latency per step: 12.4 us for 0.27 GFLOP).
constexpr int DX_BM = 64, DX_BN = 32, DX_KMAX = 256;                 // logical k staged at most
constexpr int DX_PA = 2 * DX_KMAX * 2 + 16, DX_PB = 144;             // row pitches (bytes): A rows hold 2 K physical k, B rows 64 columns
constexpr int DX_LDS = DX_BM * DX_PA + DX_KMAX * DX_PB;

__global__ __launch_bounds__(256) void x3_dx_kernel(const GemmBatch batch) {
  kernarg_prefetch<(int)sizeof(GemmProb)>((int)(blockIdx.y * sizeof(GemmProb)));
  const GemmProb& P = batch.p[blockIdx.y];
  const int nwg = P.tiles_m * P.tiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int tile_n = blockIdx.x % P.tiles_n, tile_m = blockIdx.x / P.tiles_n;
  const int m0 = tile_m * DX_BM, n0 = tile_n * DX_BN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wm0 = (wave >> 1) * 32, wc0 = (wave & 1) * 16;   // rows of the tile / logical column block of the group
  const GemmSeg& G = P.seg[0];
  const int Kc = G.K >> 1;                  // logical contraction length (n); G.K counts dZ's physical columns
  const int nt = (Kc + 31) / 32;
  const bf16_t* A = (const bf16_t*)G.A;
  const bf16_t* B = (const bf16_t*)G.B;
  unsigned char* sa = smem;
  unsigned char* sb = smem + DX_BM * DX_PA;

  // A: 64 rows x (8 nt) chunks of 16 bytes (<= 16 per thread); B: (32 nt) rows x 8 chunks (<= 8 per thread) -- EVERY load of the
  // thread is requested before its first LDS store (fully unrolled with predicates: a loop with a run-time trip count would wait
  // for each round of loads before issuing the next)
  const int ca = 8 * nt;                    // chunks per A row
  const int na = DX_BM * ca, nb = 32 * nt * 8;
  uint4 ra[16], rb[8];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int c = u * 256 + tid;
    const int row = c / ca, kc = c - row * ca;
    ra[u] = c < na ? *(const uint4*)(A + (int64_t)min(m0 + row, P.M - 1) * G.lda + kc * 8) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = u * 256 + tid;
    const int n = c >> 3, kc = c & 7;
    rb[u] = (c < nb && n < Kc) ? *(const uint4*)(B + (int64_t)n * G.ldb + 2 * n0 + kc * 8) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int c = u * 256 + tid;
    const int row = c / ca, kc = c - row * ca;
    if (c < na) *(uint4*)(sa + row * DX_PA + kc * 16) = ra[u];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = u * 256 + tid;
    if (c < nb) *(uint4*)(sb + (c >> 3) * DX_PB + (c & 7) * 16) = rb[u];
  }
  // the gate operand of this lane's outputs (rows wm0 + 16 tm + fr, columns jb .. jb + 3), requested with the tile loads
  const int jb = n0 + wc0 + fg * 4;
  uint2 yq[2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = min(m0 + wm0 + tm * 16 + fr, P.M - 1);
    yq[tm] = (P.yref && jb < P.N) ? *(const uint2*)((const bf16_t*)P.yref + (int64_t)m * P.ldy + x3_col(jb)) : make_uint2(0x3F803F80u, 0x3F803F80u);
  }
  __syncthreads();

  // (weights first: acc[tm][r] = C[row wm0 + 16 tm + fr][column jb + r] -- a lane owns four neighbouring columns of one row)
  f32x4 acc[2];
  acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < nt; ++t) {
    const unsigned char* sbt = sb + t * 32 * DX_PB;
    const bf16x8 bh = tr_frag(sbt, DX_PB, wc0, fr, fg), bl = tr_frag(sbt, DX_PB, 32 + wc0, fr, fg);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const unsigned char* ar = sa + (wm0 + tm * 16 + fr) * DX_PA + t * 128;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)(ar + fg * 16));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(ar + 64 + fg * 16));
      acc[tm] = mfma3(bh, bl, ah, al, acc[tm]);
    }
  }

  // ---- epilogue: scale, relu/dropout gate from yref, column sums per 32-row slab, split store (8 bytes of hi, 8 of lo per row)
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm0 + tm * 16 + fr;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = acc[tm][r] * P.dx_scale;
      const float y = bf2f((bf16_t)((r < 2 ? yq[tm].x : yq[tm].y) >> ((r & 1) * 16)));
      if (!(y > 0.f) || m >= P.M || jb + r >= P.N) v[r] = 0.f;
      cs[r] += v[r];
    }
    if (m < P.M && jb < P.N) {
      if (P.c_f32) {
        for (int r = 0; r < 4; ++r) if (jb + r < P.N) ((float*)P.C)[(int64_t)m * P.ldc + jb + r] = v[r];
      } else if (jb + 3 < P.N) {
        uint2 hi, lo;
        x3_split4(v, hi, lo);
        bf16_t* dst = (bf16_t*)P.C + (int64_t)m * P.ldc + x3_col(jb);
        *(uint2*)dst = hi;
        *(uint2*)(dst + 32) = lo;
      } else {
        for (int r = 0; r < 4; ++r) if (jb + r < P.N) x3_store((bf16_t*)P.C + (int64_t)m * P.ldc, jb + r, v[r]);
      }
    }
  }
  if (P.colsum) {   // a wave owns one 32-row slab: fixed tree over its rows (tm in the lane, then the 16 fr lanes of the column quad)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float c = cs[r];
      c += __shfl_xor(c, 1, 64);
      c += __shfl_xor(c, 2, 64);
      c += __shfl_xor(c, 4, 64);
      c += __shfl_xor(c, 8, 64);
      if (fr == 0 && jb + r < P.N && m0 + wm0 < P.M) P.colsum[(int64_t)((m0 + wm0) >> 5) * P.N + jb + r] = c;
    }
  }
}

// ------------------------------------------------------------------ dW
// Workgroup: 64 x 64 logical outputs = 128 physical columns of each operand (two [hi 32 | lo 32] groups); 4 waves as 2 x 2, a
// wave owns group wm of dZ x group wn of X = 32 x 32 logical = 2 x 2 blocks x 3 MFMAs per 32-row k step.
// A stage = 32 batch rows (one k step) of both operands, 17 KB: eight workgroups share a CU.  (64-row stages -- a 256-row batch
// slice four memory latencies deep instead of eight, but 70 KB per workgroup -- measured slower: 15.4 vs 13.2 us.)
constexpr int DW_ROWS = 32, DW_PITCH = 256, DW_OP = DW_ROWS * DW_PITCH, DW_STAGE = 2 * DW_OP;   // swizzled rows (tr_frag_swz)

__global__ __launch_bounds__(256) void x3_dw_kernel(const GemmBatch batch) {
  const GemmProb& P = batch.p[blockIdx.y];
  const int nwg = P.tiles_m * P.tiles_n * P.dw_splits;
  if ((int)blockIdx.x >= nwg) return;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int tile_n = lid % P.tiles_n;
  const int tile_m = (lid / P.tiles_n) % P.tiles_m;
  const int split = lid / (P.tiles_n * P.tiles_m);
  const int m0 = tile_m * 64, n0 = tile_n * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const GemmSeg& G = P.seg[0];
  const int Kc = G.K;
  const int chunk = (((Kc + P.dw_splits - 1) / P.dw_splits) + 31) / 32 * 32;   // (a multiple of the 32-row k step)
  const int kbeg = split * chunk;
  const int kend = min(Kc, kbeg + chunk);
  const int nt = kend > kbeg ? (kend - kbeg + DW_ROWS - 1) / DW_ROWS : 0;
  const bf16_t* A = (const bf16_t*)G.A + 2 * m0;
  const bf16_t* B = (const bf16_t*)G.B + 2 * n0;

  // staging: per operand 64 rows x 16 chunks of 16 bytes (4 per thread)
  constexpr int NL = DW_ROWS * 16 / 256;
  uint4 ra[NL], rb[NL];
  auto load = [&](int t) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int c = tid + i * 256, row = c >> 4, cc = c & 15;
      const int k = kbeg + t * DW_ROWS + row;
      const bool in = k < kend;
      ra[i] = in ? *(const uint4*)(A + (int64_t)k * G.lda + cc * 8) : make_uint4(0, 0, 0, 0);
      rb[i] = in ? *(const uint4*)(B + (int64_t)k * G.ldb + cc * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store = [&](unsigned char* st) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int c = tid + i * 256, row = c >> 4, cc = c & 15;
      const int pc = ((((cc >> 1) ^ swz32(row)) << 1) | (cc & 1)) * 16;
      *(uint4*)(st + row * DW_PITCH + pc) = ra[i];
      *(uint4*)(st + DW_OP + row * DW_PITCH + pc) = rb[i];
    }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nt > 0) { load(0); store(smem); }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const unsigned char* sa = smem + (t & 1) * DW_STAGE;
    const unsigned char* sb = sa + DW_OP;
    if (t + 1 < nt) load(t + 1);
#pragma unroll
    for (int ks = 0; ks < DW_ROWS / 32; ++ks) {
      const unsigned char* ka = sa + ks * 32 * DW_PITCH;
      const unsigned char* kb = sb + ks * 32 * DW_PITCH;
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ah[q] = tr_frag_swz(ka, wm * 64 + q * 16, fr, fg);
        al[q] = tr_frag_swz(ka, wm * 64 + 32 + q * 16, fr, fg);
        bh[q] = tr_frag_swz(kb, wn * 64 + q * 16, fr, fg);
        bl[q] = tr_frag_swz(kb, wn * 64 + 32 + q * 16, fr, fg);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = mfma3(ah[tm], al[tm], bh[tn], bl[tn], acc[tm][tn]);
    }
    if (t + 1 < nt) store(smem + ((t + 1) & 1) * DW_STAGE);
    __syncthreads();
  }

  float* Cs = (float*)P.C + (int64_t)split * P.dw_slab_stride;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = n0 + wn * 32 + tn * 16 + fr;
      const int mb = m0 + wm * 32 + tm * 16 + fg * 4;
      if (n < P.dw_valid_cols) {
        int cc = n + P.dw_col_rot;
        if (cc >= P.dw_valid_cols) cc -= P.dw_valid_cols;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mb + r;
          if (m < P.M) Cs[(int64_t)m * P.ldc + cc] = acc[tm][tn][r];
        }
      }
    }
}

// ------------------------------------------------------------------ conversions
// one thread per 4 logical columns
__global__ __launch_bounds__(256) void x3_pack_kernel(const float* __restrict__ src, int64_t ld, int rows, int cols, bf16_t* __restrict__ dst,
                                                      int64_t ldx, const float* __restrict__ src2, bf16_t* __restrict__ dst2) {
  const int c4 = (cols + 3) / 4;
  const int64_t n = (int64_t)rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / c4), c = (int)(i - (int64_t)r * c4) * 4;
    const bool vec = c + 3 < cols && !(ld & 3) && !(((uintptr_t)src) & 15);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const float* s = which ? src2 : src;
      bf16_t* d = which ? dst2 : dst;
      if (!s) continue;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        const float4 x = *(const float4*)(s + (int64_t)r * ld + c);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
      } else {
        for (int j = 0; j < 4; ++j) if (c + j < cols) v[j] = s[(int64_t)r * ld + c + j];
      }
      uint2 hi, lo;
      x3_split4(v, hi, lo);
      bf16_t* p = d + (int64_t)r * ldx + x3_col(c);
      if (c + 3 < cols) {
        *(uint2*)p = hi;
        *(uint2*)(p + 32) = lo;
      } else {
        for (int j = 0; j < 4 && c + j < cols; ++j) {
          p[j] = (bf16_t)((j < 2 ? hi.x : hi.y) >> ((j & 1) * 16));
          p[32 + j] = (bf16_t)((j < 2 ? lo.x : lo.y) >> ((j & 1) * 16));
        }
      }
    }
  }
}
__global__ __launch_bounds__(256) void x3_unpack_kernel(const bf16_t* __restrict__ src, int64_t ldx, int rows, int cols, float* __restrict__ dst,
                                                        int64_t ld) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    dst[(int64_t)r * ld + c] = x3_load(src + (int64_t)r * ldx, c);
  }
}

int launch_dx(GemmLaunch* L, hipStream_t s) {
  int maxwg = 0;
  for (int i = 0; i < L->nprob; ++i) {
    GemmProb& p = L->batch.p[i];
    p.tiles_m = (p.M + DX_BM - 1) / DX_BM;
    p.tiles_n = (p.N + DX_BN - 1) / DX_BN;
    if (p.seg[0].K % 64 || p.seg[0].K > 2 * DX_KMAX) {
      recnn_set_error("gemm dx (bf16x3): physical contraction length %d must be a multiple of 64, at most %d", p.seg[0].K, 2 * DX_KMAX);
      return RECNN_E_UNSUPPORTED;
    }
    if (p.seg[0].lda < p.seg[0].K || p.seg[0].ldb < (int64_t)p.tiles_n * 64) {
      recnn_set_error("gemm dx (bf16x3): operand pitch below the split width (lda=%lld K=%d, ldb=%lld N=%d)", (long long)p.seg[0].lda, p.seg[0].K,
                      (long long)p.seg[0].ldb, p.N);
      return RECNN_E_INVALID;
    }
    if (!p.c_f32 && p.ldc < x3_ld(p.N)) { recnn_set_error("gemm dx (bf16x3): ldc below the split row width"); return RECNN_E_INVALID; }
    if (p.tiles_m * p.tiles_n > maxwg) maxwg = p.tiles_m * p.tiles_n;
  }
  if (maxwg == 0) return 0;
  hipLaunchKernelGGL(x3_dx_kernel, dim3(maxwg, L->nprob, 1), dim3(256), DX_LDS, s, L->batch);
  return recnn_check_hip(hipGetLastError(), "x3_dx_kernel launch");
}

int launch_dw(GemmLaunch* L, hipStream_t s) {
  int maxwg = 0;
  for (int i = 0; i < L->nprob; ++i) {
    GemmProb& p = L->batch.p[i];
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_n = (p.N + 63) / 64;
    if (p.a_row_scale || L->vec) { recnn_set_error("gemm dw (bf16x3): row scaling / vector partials are bf16-only"); return RECNN_E_UNSUPPORTED; }
    if (p.seg[0].lda < (int64_t)p.tiles_m * 128 || p.seg[0].ldb < (int64_t)p.tiles_n * 128) {
      recnn_set_error("gemm dw (bf16x3): operand pitch below the tile width (lda=%lld M=%d, ldb=%lld N=%d)", (long long)p.seg[0].lda, p.M,
                      (long long)p.seg[0].ldb, p.N);
      return RECNN_E_INVALID;
    }
    if (p.tiles_m * p.tiles_n * p.dw_splits > maxwg) maxwg = p.tiles_m * p.tiles_n * p.dw_splits;
  }
  if (maxwg == 0) return 0;
  hipLaunchKernelGGL(x3_dw_kernel, dim3(maxwg, L->nprob, 1), dim3(256), 2 * DW_STAGE, s, L->batch);
  return recnn_check_hip(hipGetLastError(), "x3_dw_kernel launch");
}

}  // namespace

int x3_init() {
  int rc = recnn_check_hip(hipFuncSetAttribute((const void*)x3_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DX_LDS), "x3 dx attr");
  if (!rc) rc = recnn_check_hip(hipFuncSetAttribute((const void*)x3_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DW_STAGE), "x3 dw attr");
  return rc;
}

int x3_gemm_launch(GemmLaunch* L, hipStream_t stream) {
  if (L->a_f32 || L->b_f32) { recnn_set_error("gemm (bf16x3): operands must be split-bf16 rows"); return RECNN_E_UNSUPPORTED; }
  for (int i = 0; i < L->nprob; ++i) {
    const GemmProb& p = L->batch.p[i];
    for (int s = 0; s < p.nseg; ++s) {
      const GemmSeg& g = p.seg[s];
      if (!g.A || !g.B) { recnn_set_error("gemm: null operand"); return RECNN_E_INVALID; }
      if ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) || (g.lda % 8) || (g.ldb % 8)) { recnn_set_error("gemm (bf16x3): operands / pitches must be 16-byte aligned"); return RECNN_E_INVALID; }
    }
    if (!p.C) { recnn_set_error("gemm: null output"); return RECNN_E_INVALID; }
    if (L->mode != GEMM_FWD && p.nseg != 1) { recnn_set_error("gemm (bf16x3): one contraction segment for dx / dw"); return RECNN_E_UNSUPPORTED; }
  }
  switch (L->mode) {
    case GEMM_FWD: return x3_fwd_launch(L, stream);
    case GEMM_DX: return launch_dx(L, stream);
    case GEMM_DW: return launch_dw(L, stream);
  }
  recnn_set_error("gemm (bf16x3): bad mode");
  return RECNN_E_INVALID;
}

// packed fp32 rows (row stride ld32 floats) -> split twins (row stride ldh bf16), both arrays in one launch
int rows_to_x3_launch(const float* xs, const float* xn, bf16_t* hs, bf16_t* hn, int rows, int cols, int64_t ld32, int64_t ldh, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 0;
  const int64_t n = (int64_t)rows * ((cols + 3) / 4);
  int grid = (int)((n + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(x3_pack_kernel, dim3(grid), dim3(256), 0, s, xs, ld32, rows, cols, hs, ldh, xn, hn);
  return recnn_check_hip(hipGetLastError(), "rows_to_x3");
}

extern "C" int recnn_x3_pack(const float* src, int64_t ld, int rows, int cols, void* dst, int64_t ldx, void* stream) {
  RECNN_REQUIRE(src && dst && rows >= 0 && cols > 0 && ld >= cols && ldx >= x3_ld(cols), "x3_pack: bad arguments");
  RECNN_REQUIRE(!((uintptr_t)dst & 15) && !(ldx & 7), "x3_pack: split rows must be 16-byte aligned");
  return rows_to_x3_launch(src, nullptr, (bf16_t*)dst, nullptr, rows, cols, ld, ldx, (hipStream_t)stream);
}
extern "C" int recnn_x3_unpack(const void* src, int64_t ldx, int rows, int cols, float* dst, int64_t ld, void* stream) {
  RECNN_REQUIRE(src && dst && rows >= 0 && cols > 0 && ld >= cols && ldx >= x3_ld(cols), "x3_unpack: bad arguments");
  if (rows == 0) return 0;
  const int64_t n = (int64_t)rows * cols;
  int grid = (int)((n + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(x3_unpack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ldx, rows, cols, dst, ld);
  return recnn_check_hip(hipGetLastError(), "x3_unpack");
}
