// comm.h -- transports of the sharded evaluation (dist2d.hip).
//
//   RcclTransport      RCCL over xGMI, called from INSIDE the library: grouped
//                      ncclSend / ncclRecv (direct peer sends: on the fully connected
//                      8-GPU xGMI mesh every peer is one hop away, a ring would be bound by
//                      a single link) and ncclAllReduce, all enqueued on the handle's
//                      communication stream and ordered against the compute streams with
//                      events -- no host synchronisation per panel.
//   CallbackTransport  host-synchronous exchange through two callbacks of the host layer
//                      (gloo / MPI / sockets): payloads are staged through pinned host
//                      memory.  Used for rehearsals where RCCL cannot run (several ranks
//                      sharing one GPU) and by hosts without RCCL.
//
// The reference has no counterpart (single process, goroutines only: gp/gp.go:165-213).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/gogp_hip.h"

namespace gogp {

struct XferOp {
  int peer;      // rank
  bool send;     // else receive
  double *ptr;   // device pointer
  int64_t count;  // doubles
};

class Transport {
 public:
  virtual ~Transport() {}
  virtual const char *name() const = 0;
  // true: group()/allreduce() only enqueue on `sc`; false: they return after completion
  virtual bool async() const = 0;
  // ranks of the communicator as the transport itself counts them (RCCL: ncclCommCount)
  virtual int comm_ranks() const = 0;
  virtual bool is_rccl() const { return false; }
  // All transfers of one step.  Every rank calls group() for every step in the same order;
  // within a step each (sender, receiver) pair lists its transfers in the same order.
  virtual int group(hipStream_t sc, const std::vector<XferOp> &ops, std::string *err) = 0;
  // In-place sum over all ranks of `count` doubles in device memory.
  virtual int allreduce(hipStream_t sc, double *dev, int64_t count, std::string *err) = 0;
};

Transport *make_rccl_transport(int rank, int nranks, const void *unique_id128, std::string *err);
Transport *make_callback_transport(int rank, int nranks, gogp_exchange_fn ex, gogp_allreduce_fn ar,
                                   void *user);
int rccl_unique_id(void *id128);
// Measurement only (libgogp_testhooks.so: gogp_test_dist_init_replay): rank `rank` of an nranks-grid ALONE on a GPU.
// Nothing is sent; a receive zero-fills its buffer on the communication stream (the bytes a peer would have written),
// an all-reduce leaves its argument as it is.  Every launch of the rank's own share of the sweep runs with its real
// shape -- the tile kernel's time does not depend on the values -- so the evaluation's wall time is that rank's COMPUTE
// time; the numbers it returns mean nothing.
Transport *make_replay_transport(int rank, int nranks);

}  // namespace gogp
