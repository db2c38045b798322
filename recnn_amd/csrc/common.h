// common.h -- shared device helpers for librecnn_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/recnn_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef uint16_t bf16_t;  // storage type of a bfloat16

constexpr int WAVE = 64;

// ---------------------------------------------------------------- error plumbing (host)
void recnn_set_error(const char* fmt, ...);
int recnn_check_hip(hipError_t e, const char* what);
#define RECNN_HIP(expr)                                   \
  do {                                                    \
    int _rc = recnn_check_hip((expr), #expr);             \
    if (_rc) return _rc;                                  \
  } while (0)
#define RECNN_REQUIRE(cond, ...)                          \
  do {                                                    \
    if (!(cond)) {                                        \
      recnn_set_error(__VA_ARGS__);                       \
      return RECNN_E_INVALID;                             \
    }                                                     \
  } while (0)

// ---------------------------------------------------------------- bf16 <-> f32
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 pair, round to nearest even: one v_cvt_pk_bf16_f32 on gfx950
__device__ inline uint32_t cvt_pk_bf16(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__host__ __device__ inline bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (bf16_t)(cvt_pk_bf16(f, 0.f) & 0xFFFFu);
#else
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even (inputs are finite)
  return (bf16_t)(u >> 16);
#endif
}
__host__ __device__ inline float bf2f(bf16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  return __builtin_bit_cast(float, u);
}
__device__ inline uint32_t pack_bf2(float lo, float hi) { return cvt_pk_bf16(lo, hi); }

template <class T> struct TcTraits;
template <> struct TcTraits<float> {
  static constexpr int VEC = 4;    // elements per 16-byte chunk
  static constexpr int BK = 32;    // k elements per LDS stage
  static constexpr int KSTEP = 16; // k elements consumed per fragment read
  static constexpr int DT = RECNN_F32;
};
template <> struct TcTraits<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr int BK = 64;
  static constexpr int KSTEP = 32;
  static constexpr int DT = RECNN_BF16;
};

__device__ inline float tc_load(const float* p) { return *p; }
__device__ inline float tc_load(const bf16_t* p) { return bf2f(*p); }
__device__ inline void tc_store(float* p, float v) { *p = v; }
__device__ inline void tc_store(bf16_t* p, float v) { *p = f2bf(v); }

// ---------------------------------------------------------------- dropout keep-mask generator
// Counter-based: the keep bit of element (row, col) of mask stream `stream_id` at step `step` is a
// pure function of (seed, step, stream_id, row, col) -- independent of tiling, so
// recnn_hash_mask_dump reproduces exactly what the GEMM epilogues used.  One 32-bit word serves a 4 x 4 block
// (rows row & ~3 .. +3, columns col & ~3 .. +3): that is what one lane of a 16x16 MFMA tile owns in either operand order
// (four rows of one column, or -- with the operands swapped, mlp.hip -- four columns of one row).
__host__ __device__ inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ inline uint32_t mask_key(uint32_t seed, int32_t step, uint32_t stream_id) {
  return mix32(seed ^ mix32((uint32_t)step * 0x9E3779B1u + stream_id * 0x7F4A7C15u + 0x1234567u));
}
__host__ __device__ inline uint32_t mask_word(uint32_t key, uint32_t row4, uint32_t col4) {
  return mix32((row4 * 0x9E3779B1u) ^ mix32(col4 * 0x85EBCA77u + key));
}
// keep bit for row (row4*4 + r), column (col4*4 + c)
__host__ __device__ inline bool mask_keep(uint32_t word, int r, int c) { return (word >> (8 + 4 * r + c)) & 1u; }

// ---------------------------------------------------------------- wave / block reductions
// Cross-lane sums with DPP row operations (one VALU op per step) instead of ds_bpermute shuffles (LDS crossbar).
template <int CTRL, int ROW_MASK = 0xF> __device__ inline float dpp_take(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over each 32-lane half of the wave; valid in lanes 16..31 (lower half) and 48..63 (upper half)
__device__ inline float half_sum32(float v) {
  v += dpp_take<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_take<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_take<0x141>(v);       // row_half_mirror
  v += dpp_take<0x140>(v);       // row_mirror: every lane of a 16-lane row holds the row total
  v += dpp_take<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  return v;
}
// sum over the wave, returned in every lane
__device__ inline float wave_sum(float v) {
  v = half_sum32(v);
  v += dpp_take<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Pull `BYTES` bytes of the kernel-argument segment, starting `dyn_offset` bytes in, into the scalar cache -- every 64-byte line touched
// by one s_load_dword, all in flight together, waited for once.  A kernel that reads the fields of a by-value argument block one dependent
// batch after the other pays a scalar-cache miss (~0.5 us) per first touch of a line; after this they are hits (measured: l1gemm.hip, round
// 3; x3_fwd_ws_kernel -0.5 us, round 5).
#if defined(__HIPCC__)
template <int BYTES> __device__ __forceinline__ void kernarg_prefetch(int dyn_offset = 0) {
  unsigned touch = 0;
  const char __attribute__((address_space(4)))* pa =
      (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + dyn_offset;
#pragma unroll
  for (int i = 0; i < (BYTES + 63) / 64; ++i) asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"(i * 64));
  asm volatile("s_load_dword %0, %1, %2" : "+s"(touch) : "s"(pa), "n"(BYTES - 4));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(touch));
}
#endif

// XCD-aware bijective remap of a linear workgroup id: consecutive logical ids land on one XCD
// (hardware places block b on XCD b % 8), so tiles that share an operand panel share an L2.
__device__ inline int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  int xcd = bid % NX, idx = bid / NX;
  int q = nwg / NX, r = nwg % NX;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
