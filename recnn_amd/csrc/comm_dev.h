// comm_dev.h -- device side of the two-shot all-reduce (comm.hip): accessors of the peer buffers.  Protocol, memory kinds and
// the reuse argument: see the header of comm.hip.
#pragma once
#include "common.h"

constexpr int COMM_MAX_WORLD = 8;          // one node
constexpr int COMM_MAX_WG = 512;           // workgroups per collective launch (flag slots per rank)
constexpr int64_t COMM_FLAGS = (int64_t)COMM_MAX_WORLD * COMM_MAX_WG * sizeof(uint32_t);   // bytes per flag array
constexpr int64_t COMM_HDR = 2 * COMM_FLAGS;   // in_flag[rank][wg], out_flag[rank][wg] in front of in[]

// what a launch needs to take part in one collective on region [off, off + n) of the peer buffers
struct CommPort {
  char* peer[COMM_MAX_WORLD];
  uint32_t* ctl;        // local: [0] epoch, [3] workgroups that left, [4] error word
  int64_t cap, off;     // floats per in[] / out[]; first float of the region (a multiple of 4)
  int world, rank;
  unsigned long long timeout;   // wall_clock64 ticks
};

#if defined(__HIPCC__)
// Everything another agent reads or writes moves with SYSTEM-scope accesses (sc0 sc1: write-through stores, cache-bypassing
// loads -- what the compiler emits for system-scope atomics, here 16 bytes wide through buffer descriptors) on fine-grained
// memory, ordered by draining the stores (s_waitcnt vmcnt(0)) before a flag goes up.  No release / acquire FENCE anywhere: a
// system-scope fence writes back and invalidates the whole L2 of every XCD it runs on (microseconds each, and the launches that
// follow start on cold caches) -- the first, fenced version of the collective cost 30-40 us per call.
typedef __attribute__((ext_vector_type(4))) uint32_t comm_u32x4;
constexpr int COMM_AUX_SYS = 17;   // sc0 | sc1
__device__ inline __amdgpu_buffer_rsrc_t comm_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7FFFFFFF, 0x00020000); }
__device__ inline f32x4 comm_ld4(__amdgpu_buffer_rsrc_t r, int64_t group) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(group << 4), 0, COMM_AUX_SYS));
}
__device__ inline void comm_st4(__amdgpu_buffer_rsrc_t r, int64_t group, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(comm_u32x4, v), r, (int)(group << 4), 0, COMM_AUX_SYS);
}
__device__ inline float comm_ld1(const float* p) {
  return __builtin_bit_cast(float, __hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ inline void comm_st1(float* p, float v) {
  __hip_atomic_store((uint32_t*)p, __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline uint32_t comm_ld_flag(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void comm_st_flag(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline float* comm_in_of(char* base) { return (float*)(base + COMM_HDR); }
__device__ inline float* comm_out_of(char* base, int64_t cap) { return (float*)(base + COMM_HDR) + cap; }
// flag written by workgroup `wg` of rank `rank`, inside the buffer at `base`
__device__ inline uint32_t* comm_flag(char* base, bool out, int rank, int wg) {
  return (uint32_t*)(base + (out ? COMM_FLAGS : 0)) + rank * COMM_MAX_WG + wg;
}
__device__ inline uint32_t comm_epoch(const CommPort& c) { return __hip_atomic_load(c.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1; }

// Workgroup `wg` waits for ITS counterpart on every rank: lane p polls the flag workgroup `wg` of rank p raised in this rank's
// buffer.  All threads of the workgroup call.
__device__ inline void comm_wait(const CommPort& c, bool out, int wg, uint32_t ep) {
  if ((int)threadIdx.x < c.world) {
    const uint32_t* f = comm_flag(c.peer[c.rank], out, threadIdx.x, wg);
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(comm_ld_flag(f) - ep) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > c.timeout) { atomicOr(c.ctl + 4, 1u << threadIdx.x); break; }
    }
  }
  __syncthreads();
}

// Every wave's system-scope stores are acknowledged, then lane p raises this workgroup's flag in rank p's buffer.  All threads
// of the workgroup call.
__device__ inline void comm_raise(const CommPort& c, bool out, int wg, uint32_t ep) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < c.world) comm_st_flag(comm_flag(c.peer[threadIdx.x], out, c.rank, wg), ep);
}

// the workgroup that leaves last closes the epoch (thread 0 of every workgroup calls; nobody waits for this counter)
__device__ inline void comm_leave(const CommPort& c, uint32_t ep, int nwg) {
  if (__hip_atomic_fetch_add(c.ctl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)nwg - 1) {
    __hip_atomic_store(c.ctl + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(c.ctl, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#endif
