// comm.h -- data-parallel gradient exchange without the host: a two-shot all-reduce over peer-mapped buffers (comm.hip).
#pragma once
#include "common.h"
#include "comm_dev.h"

struct recnn_comm;
// in-place sum over the ranks of data[0 .. n): ONE kernel launch on stream s (capturable: a graph node)
int comm_allreduce_launch(recnn_comm* c, float* data, int64_t n, hipStream_t s);
int comm_world(const recnn_comm* c);
// Region mode (the engine): the producer launch writes this rank's contribution straight into comm_in(c, off)[0 .. n) -- whole
// float4 groups: a partial last group must be zero padded -- and the collective launch delivers the sums to dst (no copy in).
// off: a multiple of 4 floats; regions of different arenas must not overlap while both are in flight.
int64_t comm_capacity(const recnn_comm* c);
float* comm_in(const recnn_comm* c, int64_t off);
// dst == NULL: the sums stay in comm_out(c, off) -- peer-written fine-grained memory: the consumer launch must read it with
// system-scope loads (comm_dev.h comm_ld4 / comm_ld1), plain loads could hit lines cached from the previous collective
float* comm_out(const recnn_comm* c, int64_t off);
int comm_allreduce_region(recnn_comm* c, int64_t off, float* dst, int64_t n, hipStream_t s);
// the description of region `off` a launch needs to run the exchange inside itself (optim.hip)
int comm_port(const recnn_comm* c, int64_t off, CommPort* out);
