// comm.hip -- recnn_dp_allreduce_flat: the data-parallel gradient exchange as ONE kernel launch per collective, so that a
// data-parallel step is the single-GPU run graph plus collective NODES -- no host code between the phases of a step
// (SURVEY.md 8(b) export list, 5: "two-shot P2P all-reduce").  New functionality: the reference has no distributed code;
// the arithmetic it must preserve is "gradient of the global batch mean" = sum over ranks of the per-rank gradients times
// 1 / world (recnn/nn/update/ddpg.py:74-87 run on the concatenated batch).
//
// Messages are tiny (1.7 MB per critic, 1.6 MB for the actor), so the exchange is latency-bound: a ring over 8 GPUs is 14
// dependent hops.  Here every rank owns one peer-visible buffer {flags, in[cap], out[cap]} (fine-grained device memory,
// exported with hipIpcGetMemHandle and mapped by every other rank: on an MI355X node the 8 GPUs are fully connected, 7 xGMI
// links each).  The region is cut into W chunks (rank r reduces chunk r) and every chunk into G slices, G = workgroups of the
// launch (the same on every rank); workgroup j only ever touches slice j of the chunks:
//   A  copy slice j of every chunk of the local gradient into in[] of the own buffer, drain the stores, raise
//      in_flag[rank][j] in EVERY rank's buffer
//   B  wait for in_flag[p][j] of all W ranks; sum slice j of chunk `rank` over the W in[] buffers -- direct peer reads, all 7
//      links at once, ranks added in the order 0 .. W-1 whoever computes, so every rank gets the same bits -- and write the
//      sums into EVERY rank's out[] (direct peer writes); drain, raise out_flag[rank][j] everywhere
//   C  wait for out_flag[p][j] of all W ranks; copy slice j of every chunk of out[] to the destination
// Two dependent link latencies per collective instead of 2 (W - 1), and a workgroup waits for W peer workgroups only: no
// grid-wide arrival counter (128 same-address atomics per phase cost 4 us each on this chip).  Flags carry an epoch number that
// the kernel itself advances (device memory), so a captured graph replays correctly.  Buffer reuse is safe without double
// buffering: workgroup j leaves C of epoch k only after its counterparts finished reading slice j of its in[] (they raised
// out_flag after B), and nobody writes slice j of its out[] for epoch k + 1 before it raised in_flag k + 1 -- which it does
// after the stream-ordered consumer of the destination ran.
//
// Every wait is bounded (wall clock, COMM_TIMEOUT_MS): a peer that never arrives sets an error word (recnn_comm_status) and
// the kernel ends -- a wrong result that is reported, not a hung GPU.
#include "comm.h"
#include "comm_dev.h"

namespace {
constexpr int COMM_THREADS = 256;
constexpr unsigned COMM_TIMEOUT_MS = 4000;

struct CommArgs {
  CommPort port;
  const float* src;     // this rank's contribution (NULL: the producer launch wrote it into in[off ..) itself)
  float* dst;           // where the sums go (ordinary device memory: the consumer launch reads it with plain loads)
  int64_t n;
};

// group index of element k of the concatenation of slice `wg` of all `world` chunks (-1: past the end of that slice)
__device__ inline int64_t slice_group(int64_t k, int64_t slice, int64_t chunk, int64_t g4, int wg) {
  const int64_t q = k / slice, in_slice = k - q * slice;
  const int64_t in_chunk = (int64_t)wg * slice + in_slice;
  const int64_t g = q * chunk + in_chunk;
  return (in_chunk < chunk && g < g4) ? g : -1;
}

__global__ __launch_bounds__(COMM_THREADS) void allreduce_kernel(const CommArgs a) {
  const CommPort& c = a.port;
  const uint32_t ep = comm_epoch(c);
  const int nwg = gridDim.x, wg = blockIdx.x;
  const int64_t n4 = a.n >> 2, g4 = (a.n + 3) >> 2;
  const int64_t chunk = (g4 + c.world - 1) / c.world;      // groups per rank
  const int64_t slice = (chunk + nwg - 1) / nwg;            // groups per workgroup inside a chunk
  const int64_t mine = slice * c.world;                     // groups this workgroup moves in A and in C
  const int tail = (int)(a.n & 3);
  char* own = c.peer[c.rank];
  // ---- A: gradient -> in[] with system-scope stores (a partial last group is padded with zeros)
  if (a.src) {
    const __amdgpu_buffer_rsrc_t rin = comm_rsrc(comm_in_of(own) + c.off);
    for (int64_t k0 = threadIdx.x; k0 < mine; k0 += 4 * COMM_THREADS) {
      f32x4 v[4];
      int64_t g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + u * COMM_THREADS;
        g[u] = k < mine ? slice_group(k, slice, chunk, g4, wg) : -1;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g[u] >= 0 && g[u] < n4) v[u] = ((const f32x4*)a.src)[g[u]];
        else if (g[u] >= 0)
          for (int t = 0; t < tail; ++t) v[u][t] = a.src[(n4 << 2) + t];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (g[u] >= 0) comm_st4(rin, g[u], v[u]);
    }
  }
  comm_raise(c, false, wg, ep);
  // ---- B: slice `wg` of chunk `rank`, summed over the ranks in rank order, scattered to every rank's out[]
  comm_wait(c, false, wg, ep);
  {
    const int64_t lo = chunk * c.rank + (int64_t)wg * slice;
    int64_t hi = lo + slice;
    if (hi > chunk * (c.rank + 1)) hi = chunk * (c.rank + 1);
    if (hi > g4) hi = g4;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 2 * COMM_THREADS) {
      const bool two = i + COMM_THREADS < hi;
      f32x4 v0[COMM_MAX_WORLD], v1[COMM_MAX_WORLD];
#pragma unroll
      for (int p = 0; p < COMM_MAX_WORLD; ++p)
        if (p < c.world) {
          const __amdgpu_buffer_rsrc_t r = comm_rsrc(comm_in_of(c.peer[p]) + c.off);
          v0[p] = comm_ld4(r, i);
          v1[p] = two ? comm_ld4(r, i + COMM_THREADS) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      f32x4 s0 = v0[0], s1 = v1[0];
#pragma unroll
      for (int p = 1; p < COMM_MAX_WORLD; ++p)
        if (p < c.world) { s0 += v0[p]; s1 += v1[p]; }
#pragma unroll
      for (int p = 0; p < COMM_MAX_WORLD; ++p)
        if (p < c.world) {
          const __amdgpu_buffer_rsrc_t r = comm_rsrc(comm_out_of(c.peer[p], c.cap) + c.off);
          comm_st4(r, i, s0);
          if (two) comm_st4(r, i + COMM_THREADS, s1);
        }
    }
  }
  comm_raise(c, true, wg, ep);
  // ---- C: out[] -> the caller's buffer (plain stores: the next launch reads them as usual); without a destination the
  // consumer launch reads out[] itself, with system-scope loads (the optimizer of a critic: optim.hip)
  comm_wait(c, true, wg, ep);
  if (a.dst) {
    const __amdgpu_buffer_rsrc_t rout = comm_rsrc(comm_out_of(own, c.cap) + c.off);
    for (int64_t k0 = threadIdx.x; k0 < mine; k0 += 4 * COMM_THREADS) {
      f32x4 v[4];
      int64_t g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + u * COMM_THREADS;
        g[u] = k < mine ? slice_group(k, slice, chunk, g4, wg) : -1;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g[u] >= 0) v[u] = comm_ld4(rout, g[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (g[u] >= 0 && g[u] < n4) ((f32x4*)a.dst)[g[u]] = v[u];
        else if (g[u] >= 0)
          for (int t = 0; t < tail; ++t) a.dst[(n4 << 2) + t] = v[u][t];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) comm_leave(c, ep, nwg);
}
}  // namespace

struct recnn_comm {
  int world = 1, rank = 0;
  int64_t cap = 0;
  size_t bytes = 0;
  char* base = nullptr;
  char* peer[COMM_MAX_WORLD] = {};
  uint32_t* ctl = nullptr;
  hipIpcMemHandle_t handle;
  bool exported = false, connected = false;
  unsigned long long timeout = 0;
  int khz = 100000;              // wall_clock64 rate
  unsigned timeout_ms = COMM_TIMEOUT_MS;
  int memory = 0;                // 0 fine-grained device memory (default), 1 uncached, 2 ordinary hipMalloc (system-scope fences alone)
  int max_wg = 128;              // workgroups per collective (x 256 threads x 4 groups in flight = 2 MB per pass)
};

int comm_world(const recnn_comm* c) { return c ? c->world : 1; }

// Workgroups per collective launch of THIS communicator, for launches made or captured afterwards (round 6: was a process-wide setting).
// Ranks that share one GPU in tests need every rank's collective resident at once: 32.
extern "C" int recnn_comm_set_workgroups(recnn_comm* c, int n) {
  RECNN_REQUIRE(c, "comm_set_workgroups: null communicator");
  c->max_wg = n < 1 ? 1 : (n > COMM_MAX_WG ? COMM_MAX_WG : n);
  return 0;
}

int comm_port(const recnn_comm* c, int64_t off, CommPort* out) {
  RECNN_REQUIRE(c && c->connected && out && off >= 0 && (off & 3) == 0 && off < c->cap, "comm_port: communicator not connected / bad region offset");
  memset(out, 0, sizeof(*out));
  for (int p = 0; p < c->world; ++p) out->peer[p] = c->peer[p];
  out->ctl = c->ctl; out->cap = c->cap; out->off = off; out->world = c->world; out->rank = c->rank; out->timeout = c->timeout;
  return 0;
}

static int launch(recnn_comm* c, const float* src, float* dst, int64_t off, int64_t n, hipStream_t s) {
  CommArgs a;
  memset(&a, 0, sizeof(a));
  int rc = comm_port(c, off, &a.port);
  if (rc) return rc;
  a.src = src; a.dst = dst; a.n = n;
  int64_t wg = ((n + 3) / 4 + COMM_THREADS * 4 - 1) / (COMM_THREADS * 4);
  if (wg < 1) wg = 1;
  if (wg > c->max_wg) wg = c->max_wg;
  hipLaunchKernelGGL(allreduce_kernel, dim3((unsigned)wg), dim3(COMM_THREADS), 0, s, a);
  return recnn_check_hip(hipGetLastError(), "allreduce_kernel");
}

int comm_allreduce_launch(recnn_comm* c, float* data, int64_t n, hipStream_t s) {
  RECNN_REQUIRE(c && c->connected, "dp_allreduce_flat: the communicator is not connected (recnn_comm_connect)");
  RECNN_REQUIRE(data && n > 0 && n <= c->cap, "dp_allreduce_flat: %lld floats do not fit the communicator's %lld", (long long)n, (long long)c->cap);
  RECNN_REQUIRE(((uintptr_t)data & 15) == 0, "dp_allreduce_flat: the buffer must be 16-byte aligned");
  return launch(c, data, data, 0, n, s);
}

int64_t comm_capacity(const recnn_comm* c) { return c ? c->cap : 0; }
float* comm_out(const recnn_comm* c, int64_t off) { return (float*)(c->base + COMM_HDR) + c->cap + off; }
int comm_allreduce_region(recnn_comm* c, int64_t off, const float* src, float* dst, int64_t n, hipStream_t s) {
  RECNN_REQUIRE(c && c->connected, "dp_allreduce_flat: the communicator is not connected (recnn_comm_connect)");
  RECNN_REQUIRE(src && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0 && n > 0 && off >= 0 && (off & 3) == 0 && off + n <= c->cap,
                "dp_allreduce_flat: region [%lld, +%lld) outside the communicator's %lld floats, or a misaligned buffer",
                (long long)off, (long long)n, (long long)c->cap);
  return launch(c, src, dst, off, n, s);
}

extern "C" int recnn_comm_create_ex(int world, int rank, int64_t max_floats, int memory_kind, recnn_comm** out);
extern "C" int recnn_comm_create(int world, int rank, int64_t max_floats, recnn_comm** out) {
  return recnn_comm_create_ex(world, rank, max_floats, 0, out);
}
// memory_kind of the peer buffers: 0 fine-grained (default), 1 uncached, 2 ordinary device memory
extern "C" int recnn_comm_create_ex(int world, int rank, int64_t max_floats, int memory_kind, recnn_comm** out) {
  RECNN_REQUIRE(out, "comm_create: null output");
  RECNN_REQUIRE(memory_kind >= 0 && memory_kind <= 2, "comm_create: memory kind 0 (fine-grained), 1 (uncached) or 2 (ordinary)");
  *out = nullptr;
  RECNN_REQUIRE(world >= 1 && world <= COMM_MAX_WORLD && rank >= 0 && rank < world, "comm_create: world %d (max %d), rank %d", world, COMM_MAX_WORLD, rank);
  RECNN_REQUIRE(max_floats > 0, "comm_create: capacity must be positive");
  recnn_comm* c = new recnn_comm();
  c->world = world; c->rank = rank; c->memory = memory_kind;
  c->cap = (max_floats + 63) & ~(int64_t)63;
  c->bytes = (size_t)(COMM_HDR + 2 * c->cap * (int64_t)sizeof(float));
  // fine-grained device memory: stores of one agent become visible to the others inside a running kernel (system-scope
  // release / acquire), which ordinary (coarse-grained) device memory only promises at kernel boundaries
  hipError_t e = c->memory == 2 ? hipMalloc((void**)&c->base, c->bytes)
                                : hipExtMallocWithFlags((void**)&c->base, c->bytes, c->memory == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
  if (e != hipSuccess) { (void)hipGetLastError(); delete c; return recnn_check_hip(e, "comm_create: peer buffer allocation"); }
  e = hipMalloc((void**)&c->ctl, 8 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(c->base, 0, c->bytes);
  if (e == hipSuccess) e = hipMemset(c->ctl, 0, 8 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(c->base); if (c->ctl) (void)hipFree(c->ctl); delete c; return recnn_check_hip(e, "comm_create: buffers"); }
  c->peer[rank] = c->base;
  int khz = 100000;   // wall_clock64 runs at the constant 100 MHz reference clock on gfx9
  int dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  if (khz <= 0) khz = 100000;
  c->khz = khz;
  c->timeout = (unsigned long long)khz * COMM_TIMEOUT_MS;
  c->connected = world == 1;
  *out = c;
  return 0;
}

extern "C" int64_t recnn_comm_handle_bytes(void) { return (int64_t)sizeof(hipIpcMemHandle_t); }

extern "C" int recnn_comm_export(recnn_comm* c, void* handle_out, int64_t bytes) {
  RECNN_REQUIRE(c && handle_out && bytes == (int64_t)sizeof(hipIpcMemHandle_t), "comm_export: need a %d-byte handle buffer", (int)sizeof(hipIpcMemHandle_t));
  if (!c->exported) {
    RECNN_HIP(hipIpcGetMemHandle(&c->handle, c->base));
    c->exported = true;
  }
  memcpy(handle_out, &c->handle, sizeof(c->handle));
  return 0;
}

extern "C" int recnn_comm_connect(recnn_comm* c, const void* handles, int64_t bytes_each) {
  RECNN_REQUIRE(c && handles && bytes_each == (int64_t)sizeof(hipIpcMemHandle_t), "comm_connect: need world x %d bytes of handles", (int)sizeof(hipIpcMemHandle_t));
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank || c->peer[p]) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (int64_t)p * bytes_each, sizeof(h));
    void* ptr = nullptr;
    RECNN_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
    c->peer[p] = (char*)ptr;
  }
  c->connected = true;
  return 0;
}

extern "C" int recnn_dp_allreduce_flat(recnn_comm* c, float* data, int64_t n, void* stream) {
  return comm_allreduce_launch(c, data, n, (hipStream_t)stream);
}

extern "C" int recnn_comm_status(recnn_comm* c, int32_t* timed_out_ranks, int32_t* epoch) {
  RECNN_REQUIRE(c, "comm_status: null communicator");
  uint32_t h[8];
  RECNN_HIP(hipMemcpy(h, c->ctl, sizeof(h), hipMemcpyDeviceToHost));
  if (timed_out_ranks) *timed_out_ranks = (int32_t)h[4];
  if (epoch) *epoch = (int32_t)h[0];
  if (h[4]) {
    recnn_set_error("dp_allreduce_flat: a wait for peer rank(s) 0x%x ran out after %u ms (rank %d of %d, epoch %u): the reduced "
                    "gradients since then are invalid", h[4], c->timeout_ms, c->rank, c->world, h[0]);
    return RECNN_E_STATE;
  }
  return 0;
}

// Bound of every peer wait of the collectives launched -- or captured into graphs -- AFTERWARDS (the value travels in the kernel
// arguments).  A first contact across a fabric is better made with a short bound (a pre-flight: 200 ms), training with the default.
extern "C" int recnn_comm_set_timeout_ms(recnn_comm* c, int ms) {
  RECNN_REQUIRE(c && ms > 0 && ms <= 600000, "comm_set_timeout_ms: 1 .. 600000 ms");
  c->timeout_ms = (unsigned)ms;
  c->timeout = (unsigned long long)c->khz * (unsigned)ms;
  return 0;
}
/* clears the timed-out word (after a reported time-out, e.g. between the stages of a pre-flight) */
extern "C" int recnn_comm_clear_status(recnn_comm* c) {
  RECNN_REQUIRE(c, "comm_clear_status: null communicator");
  RECNN_HIP(hipDeviceSynchronize());
  RECNN_HIP(hipMemset(c->ctl + 4, 0, sizeof(uint32_t)));
  return 0;
}

extern "C" void recnn_comm_destroy(recnn_comm* c) {
  if (!c) return;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  if (c->base) (void)hipFree(c->base);
  if (c->ctl) (void)hipFree(c->ctl);
  delete c;
}
