// comm.hip -- transports of the sharded evaluation: RCCL (in-library, stream-ordered) and
// host callbacks (rehearsal / hosts without RCCL).  See comm.h.
#include "comm.h"

#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

namespace gogp {

static_assert(GOGP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

#define NCCLCHK(call)                                                                         \
  do {                                                                                        \
    ncclResult_t r_ = (call);                                                                 \
    if (r_ != ncclSuccess) {                                                                  \
      if (err) {                                                                              \
        char b_[384];                                                                         \
        snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, \
                 __LINE__);                                                                   \
        *err = b_;                                                                            \
      }                                                                                       \
      return GOGP_EHIP;                                                                       \
    }                                                                                         \
  } while (0)

int rccl_unique_id(void *id128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return GOGP_EHIP;
  memcpy(id128, &id, sizeof id);
  return GOGP_OK;
}

// ---- RCCL ----------------------------------------------------------------------------------
class RcclTransport : public Transport {
 public:
  ncclComm_t comm = nullptr;
  ~RcclTransport() override {
    if (comm) (void)ncclCommDestroy(comm);
  }
  const char *name() const override { return "RCCL grouped ncclSend/ncclRecv + ncclAllReduce on the comm stream"; }
  bool async() const override { return true; }
  bool is_rccl() const override { return true; }
  int comm_ranks() const override {
    int n = 0;
    return (comm && ncclCommCount(comm, &n) == ncclSuccess) ? n : -1;
  }
  int group(hipStream_t sc, const std::vector<XferOp> &ops, std::string *err) override {
    bool any = false;
    for (const XferOp &o : ops) any = any || o.count > 0;
    if (!any) return GOGP_OK;
    NCCLCHK(ncclGroupStart());
    for (const XferOp &o : ops) {
      if (o.count <= 0) continue;
      ncclResult_t r = o.send ? ncclSend(o.ptr, (size_t)o.count, ncclDouble, o.peer, comm, sc)
                              : ncclRecv(o.ptr, (size_t)o.count, ncclDouble, o.peer, comm, sc);
      if (r != ncclSuccess) {
        (void)ncclGroupEnd();
        if (err) *err = std::string("ncclSend/ncclRecv failed: ") + ncclGetErrorString(r);
        return GOGP_EHIP;
      }
    }
    NCCLCHK(ncclGroupEnd());
    return GOGP_OK;
  }
  int allreduce(hipStream_t sc, double *dev, int64_t count, std::string *err) override {
    NCCLCHK(ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, comm, sc));
    return GOGP_OK;
  }
};

Transport *make_rccl_transport(int rank, int nranks, const void *unique_id128, std::string *err) {
  RcclTransport *t = new RcclTransport();
  ncclUniqueId id;
  memcpy(&id, unique_id128, sizeof id);
  ncclResult_t r = ncclCommInitRank(&t->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    if (err) *err = std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r);
    t->comm = nullptr;
    delete t;
    return nullptr;
  }
  return t;
}

// ---- host callbacks ------------------------------------------------------------------------
class CallbackTransport : public Transport {
 public:
  gogp_exchange_fn ex = nullptr;
  gogp_allreduce_fn ar = nullptr;
  void *user = nullptr;
  int nranks = 1;
  char *host = nullptr;  // pinned staging
  size_t host_bytes = 0;
  ~CallbackTransport() override {
    if (host) (void)hipHostFree(host);
  }
  const char *name() const override { return "host callbacks (pinned staging, host-synchronous)"; }
  bool async() const override { return false; }
  int comm_ranks() const override { return nranks; }
  int ensure(size_t bytes, std::string *err) {
    if (bytes <= host_bytes) return GOGP_OK;
    if (host) (void)hipHostFree(host);
    host = nullptr;
    host_bytes = 0;
    bytes += bytes / 4 + 4096;
    if (hipHostMalloc((void **)&host, bytes, hipHostMallocDefault) != hipSuccess) {
      if (err) *err = "callback transport: pinned staging allocation failed";
      (void)hipGetLastError();
      return GOGP_ENOMEM;
    }
    host_bytes = bytes;
    return GOGP_OK;
  }
  int group(hipStream_t sc, const std::vector<XferOp> &ops, std::string *err) override {
    size_t total = 0;
    for (const XferOp &o : ops)
      if (o.count > 0) total += (size_t)o.count * sizeof(double);
    if (total == 0) return GOGP_OK;
    int rc = ensure(total, err);
    if (rc != GOGP_OK) return rc;
    std::vector<gogp_xfer> hx;
    size_t off = 0;
    for (const XferOp &o : ops) {
      if (o.count <= 0) continue;
      const size_t b = (size_t)o.count * sizeof(double);
      gogp_xfer x;
      x.peer = o.peer;
      x.is_send = o.send ? 1 : 0;
      x.buf = host + off;
      x.bytes = (int64_t)b;
      if (o.send && hipMemcpyAsync(host + off, o.ptr, b, hipMemcpyDeviceToHost, sc) != hipSuccess) {
        if (err) *err = "callback transport: D2H copy failed";
        return GOGP_EHIP;
      }
      hx.push_back(x);
      off += b;
    }
    if (hipStreamSynchronize(sc) != hipSuccess) {
      if (err) *err = "callback transport: stream synchronisation failed";
      return GOGP_EHIP;
    }
    if (ex(user, hx.data(), (int32_t)hx.size()) != 0) {
      if (err) *err = "exchange callback failed";
      return GOGP_EHIP;
    }
    off = 0;
    for (const XferOp &o : ops) {
      if (o.count <= 0) continue;
      const size_t b = (size_t)o.count * sizeof(double);
      if (!o.send && hipMemcpyAsync(o.ptr, host + off, b, hipMemcpyHostToDevice, sc) != hipSuccess) {
        if (err) *err = "callback transport: H2D copy failed";
        return GOGP_EHIP;
      }
      off += b;
    }
    // the staging buffer is reused by the next group
    if (hipStreamSynchronize(sc) != hipSuccess) {
      if (err) *err = "callback transport: stream synchronisation failed";
      return GOGP_EHIP;
    }
    return GOGP_OK;
  }
  int allreduce(hipStream_t sc, double *dev, int64_t count, std::string *err) override {
    const size_t b = (size_t)count * sizeof(double);
    int rc = ensure(b, err);
    if (rc != GOGP_OK) return rc;
    if (hipMemcpyAsync(host, dev, b, hipMemcpyDeviceToHost, sc) != hipSuccess ||
        hipStreamSynchronize(sc) != hipSuccess) {
      if (err) *err = "callback transport: D2H copy failed";
      return GOGP_EHIP;
    }
    if (ar(user, (double *)host, count) != 0) {
      if (err) *err = "allreduce callback failed";
      return GOGP_EHIP;
    }
    if (hipMemcpyAsync(dev, host, b, hipMemcpyHostToDevice, sc) != hipSuccess ||
        hipStreamSynchronize(sc) != hipSuccess) {
      if (err) *err = "callback transport: H2D copy failed";
      return GOGP_EHIP;
    }
    return GOGP_OK;
  }
};

Transport *make_callback_transport(int rank, int nranks, gogp_exchange_fn ex, gogp_allreduce_fn ar,
                                   void *user) {
  (void)rank;
  CallbackTransport *t = new CallbackTransport();
  t->nranks = nranks;
  t->ex = ex;
  t->ar = ar;
  t->user = user;
  return t;
}

// ---- replay: one rank of a grid alone on a GPU (measurement only, comm.h) -------------------------------------------
class ReplayTransport : public Transport {
 public:
  int rank = 0, nranks = 1;
  int64_t recv_bytes = 0, send_bytes = 0;
  const char *name() const override { return "replay (one rank of the grid alone: receives zero-fill, nothing is sent)"; }
  bool async() const override { return true; }
  int comm_ranks() const override { return nranks; }
  int group(hipStream_t sc, const std::vector<XferOp> &ops, std::string *err) override {
    for (const XferOp &o : ops) {
      if (o.send) {
        send_bytes += o.count * 8;
        continue;
      }
      recv_bytes += o.count * 8;
      if (hipMemsetAsync(o.ptr, 0, (size_t)o.count * sizeof(double), sc) != hipSuccess) {
        if (err) *err = "replay transport: hipMemsetAsync failed";
        return GOGP_EHIP;
      }
    }
    return GOGP_OK;
  }
  int allreduce(hipStream_t, double *, int64_t, std::string *) override { return GOGP_OK; }
};
Transport *make_replay_transport(int rank, int nranks) {
  ReplayTransport *t = new ReplayTransport();
  t->rank = rank;
  t->nranks = nranks;
  return t;
}

}  // namespace gogp
