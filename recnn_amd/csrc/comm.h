// comm.h -- data-parallel gradient exchange without the host: a two-shot all-reduce over peer-mapped buffers (comm.hip).
#pragma once
#include "common.h"
#include "comm_dev.h"

struct recnn_comm;
// in-place sum over the ranks of data[0 .. n): ONE kernel launch on stream s (capturable: a graph node)
int comm_allreduce_launch(recnn_comm* c, float* data, int64_t n, hipStream_t s);
int comm_world(const recnn_comm* c);
// Region mode (the engine): every trained network has a region [off, off + n) of its own in the peer buffers (off: a multiple
// of 4 floats), so collectives of different networks never meet.  src: this rank's contribution (ordinary device memory, copied
// into in[] with system-scope stores inside the launch).  dst == NULL: the sums stay in comm_out(c, off) -- peer-written
// fine-grained memory: the consumer launch must read it with system-scope loads (comm_dev.h comm_ld4 / comm_ld1), plain loads
// could hit lines cached from the previous collective.
int64_t comm_capacity(const recnn_comm* c);
float* comm_out(const recnn_comm* c, int64_t off);
int comm_allreduce_region(recnn_comm* c, int64_t off, const float* src, float* dst, int64_t n, hipStream_t s);
// the description of region `off` a launch needs to run the exchange inside itself (optim.hip)
int comm_port(const recnn_comm* c, int64_t off, CommPort* out);
