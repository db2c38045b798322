// dma_bw.hip -- micro-benchmark: how fast can ONE CU stream an L2-resident weight matrix?  (decides what bounds layer 1 of
// csrc/mlp.hip).  Every workgroup (1 per CU, 1024 threads) streams the same [256 x K] bf16 matrix slab by slab.
//   mode 0: global_load_lds_dwordx4, 4 rows x 256 B per wave instruction (the kernel's pattern), D slabs in flight
//   mode 1: same, source pre-tiled (1 KiB contiguous per wave instruction)
//   mode 2: global_load_dwordx4 into registers, 4 rows x 256 B per wave instruction
//   mode 3: mode 2 with 1 KiB contiguous per wave instruction
//   mode 4: global_load_lds_dwordx4, 8 rows x 128 B per wave instruction (64-k slabs of a 256-row matrix), row pitch = ld_bytes
//   mode 5: waves 0-7 by LDS-DMA, waves 8-15 through registers (global_load_dwordx4 -> ds_write_b128), half the rows each (round 5)
//   mode 6: all 16 waves through registers + ds_write_b128
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/dma_bw tools/dma_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

template <int MODE, int SLAB_ROWS, int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const char* w, int ld_bytes, int nslab, int reps, unsigned long long* ticks, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NI = SLAB_ROWS / 64;                 // instructions per wave per slab (16 waves x 4 rows)
  constexpr int STAGE = SLAB_ROWS * 256;
  const int q_row = lane >> 4, q_pos = lane & 15;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 5 || MODE == 6) {
    // register path, software-pipelined two slabs deep (pa / pb): the loads of slab t + 1 are issued before slab t is written to LDS
    const bool regw = MODE == 6 || wave >= 8;
    u32x4 pa[NI], pb[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) pa[j] = pb[j] = acc;
    auto src_of = [&](int t, int j) { return w + (int64_t)((j * 16 + wave) * 4 + q_row) * ld_bytes + t * 256 + q_pos * 16; };
    for (int rep = 0; rep < reps; ++rep) {
      if (regw) {
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pa[j]) : "v"(src_of(0, j)) : "memory");
      }
      for (int t = 0; t < nslab; t += 2) {
        if (regw) {
          if (t + 1 < nslab) {
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pb[j]) : "v"(src_of(t + 1, j)) : "memory");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
          } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < NI; ++j) *(u32x4*)(lds + (t % DEPTH) * STAGE + (j * 16 + wave) * 1024 + lane * 16) = pa[j];
          if (t + 2 < nslab) {
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pa[j]) : "v"(src_of(t + 2, j)) : "memory");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
          } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (t + 1 < nslab) {
#pragma unroll
            for (int j = 0; j < NI; ++j) *(u32x4*)(lds + ((t + 1) % DEPTH) * STAGE + (j * 16 + wave) * 1024 + lane * 16) = pb[j];
          }
        } else {
          for (int tt = t; tt < t + 2 && tt < nslab; ++tt) {
#pragma unroll
            for (int j = 0; j < NI; ++j) dma16(src_of(tt, j), lds0 + (tt % DEPTH) * STAGE + (j * 16 + wave) * 1024);
            if (tt >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NI) : "memory");
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    if (pa[0].x == 0x12345678u || pb[0].y == 0x12345678u) sink[0] = 1.f;
    return;
  }
  for (int rep = 0; rep < reps; ++rep) {
    for (int t = 0; t < nslab; ++t) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int inst = j * 16 + wave;              // 4 rows each
        const char* src;
        if (MODE == 4) src = w + (int64_t)(inst * 8 + (lane >> 3)) * ld_bytes + (t % (ld_bytes / 128 > 0 ? ld_bytes / 128 : 1)) * 128 + (lane & 7) * 16;
        else if (MODE == 0 || MODE == 2) src = w + (int64_t)(inst * 4 + q_row) * ld_bytes + t * 256 + q_pos * 16;
        else src = w + ((int64_t)t * SLAB_ROWS * 256) + inst * 1024 + lane * 16;
        if (MODE < 2 || MODE == 4) {
          dma16(src, lds0 + (t % DEPTH) * STAGE + inst * 1024);
        } else {
          u32x4 v;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(src) : "memory");
          acc ^= v;      // (consumed only after the explicit waits below; 16 loads in flight need 64 VGPRs)
        }
      }
      if (MODE < 2 || MODE == 4) {
        // keep DEPTH-1 slabs in flight
        if (t >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NI) : "memory");
      } else {
        if (t >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NI) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  if (acc.x == 0x12345678u) sink[0] = 1.f;
}

template <int MODE, int SLAB_ROWS, int DEPTH>
void run(const char* name, const char* w, int ld_bytes, int nslab, int nwg) {
  unsigned long long* ticks;
  float* sink;
  hipMalloc(&ticks, nwg * sizeof(*ticks));
  hipMalloc(&sink, 4);
  const int reps = 8;
  const int lds_bytes = DEPTH * SLAB_ROWS * 256;
  hipFuncSetAttribute((const void*)stream_kernel<MODE, SLAB_ROWS, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<MODE, SLAB_ROWS, DEPTH>), dim3(nwg), dim3(1024), lds_bytes, 0, w, ld_bytes, nslab, reps, ticks, sink);
    hipEventRecord(e1);
    hipError_t er = hipEventSynchronize(e1);
    if (er != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: FAILED %s\n", name, hipGetErrorString(er)); return; }
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), ticks, nwg * sizeof(*ticks), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= nwg;
  const double bytes = (double)reps * nslab * SLAB_ROWS * 256;
  printf("%-52s wgs %3d  slab %3d KB depth %d: %8.0f ticks/WG, %6.1f B/tick/CU, kernel %7.1f us -> %6.1f GB/s per CU, %6.2f TB/s chip\n", name, nwg,
         SLAB_ROWS / 4, DEPTH, mean, bytes / mean, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * nwg / (ms * 1e-3) / 1e12);
  hipFree(ticks); hipFree(sink);
}

static FILE* g_json = nullptr;
static bool g_first = true;
template <int MODE, int SLAB_ROWS, int DEPTH>
void run_j(const char* name, const char* w, int ld_bytes, int nslab, int nwg) {
  // (run() prints; the same numbers again as one JSON row)
  unsigned long long* ticks;
  float* sink;
  hipMalloc(&ticks, nwg * sizeof(*ticks));
  hipMalloc(&sink, 4);
  const int reps = 8;
  const int lds_bytes = DEPTH * SLAB_ROWS * 256;
  hipFuncSetAttribute((const void*)stream_kernel<MODE, SLAB_ROWS, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<MODE, SLAB_ROWS, DEPTH>), dim3(nwg), dim3(1024), lds_bytes, 0, w, ld_bytes, nslab, reps, ticks, sink);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: FAILED\n", name); return; }
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), ticks, nwg * sizeof(*ticks), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= nwg;
  const double bytes = (double)reps * nslab * SLAB_ROWS * 256;
  // (rates from the kernel's event time; B/clk at the nominal 2.4 GHz)
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("%-64s slab %3d KB x %d in flight: kernel %7.1f us, %6.1f GB/s per CU = %5.1f B/clk/CU, %6.2f TB/s chip\n", name, SLAB_ROWS / 4, DEPTH,
         ms * 1e3, gbs, gbs / 2.4, bytes * nwg / (ms * 1e-3) / 1e12);
  (void)mean;
  if (g_json) {
    fprintf(g_json, "%s\n  {\"variant\": \"%s\", \"slab_kb\": %d, \"slabs_in_flight\": %d, \"workgroups\": %d, \"B_per_clk_per_CU_at_2p4GHz\": %.2f, \"GBs_per_CU\": %.1f, \"TBs_chip\": %.2f, \"kernel_us\": %.1f}",
            g_first ? "" : ",", name, SLAB_ROWS / 4, DEPTH, nwg, gbs / 2.4, gbs, bytes * nwg / (ms * 1e-3) / 1e12, ms * 1e3);
    g_first = false;
  }
  hipFree(ticks); hipFree(sink);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int nwg = argc > 1 ? atoi(argv[1]) : 256;
  if (argc > 2) { g_json = fopen(argv[2], "w"); if (g_json) fprintf(g_json, "{\"what\": \"one workgroup of 16 waves per CU streams an L2-resident [256 x 1408] bf16 matrix into LDS slab by slab (tools/dma_bw.hip): which issue path moves the most bytes per clock per CU\",\n \"rows\": ["); }
  const int K = 1408, rows = 256;
  char* w;
  hipMalloc(&w, (size_t)rows * K * 2 + 4096);
  hipMemset(w, 1, (size_t)rows * K * 2 + 4096);
  const int nslab = K / 128;
  // round 5: the LDS-DMA path alone, the register path alone, and BOTH at once sharing the bytes
  run_j<0, 128, 4>("lds-dma only (16 waves, 4 rows x 256 B per instruction)", w, K * 2, nslab, nwg);
  run_j<6, 128, 4>("registers only (global_load_dwordx4 -> ds_write_b128, 16 waves)", w, K * 2, nslab, nwg);
  run_j<5, 128, 4>("split: waves 0-7 lds-dma + waves 8-15 registers, half the rows each", w, K * 2, nslab, nwg);
  run_j<0, 256, 2>("lds-dma only, 64 KB slabs x 2", w, K * 2, nslab / 2, nwg);
  run_j<5, 256, 2>("split, 64 KB slabs x 2", w, K * 2, nslab / 2, nwg);
  run_j<4, 128, 4>("lds-dma 8 rows x 128 B per instruction (mlps.hip's weight slabs)", w, 2816, 22, nwg);
  if (g_json) { fprintf(g_json, "\n ]}\n"); fclose(g_json); }
  return 0;
}
