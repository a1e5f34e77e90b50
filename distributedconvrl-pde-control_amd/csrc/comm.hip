// comm.hip -- the build's one collective: a sum all-reduce of the flattened policy/critic
// gradient over RCCL (xGMI).  The reference has no collective at all; this
// is the data-parallel addition (one process per GPU, batch of trajectories sharded).
// At ~21 k floats the op is latency-bound, so it is a single fused buffer per network.
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "common.hpp"
#include "mlp.hpp"

namespace pdec {
struct Comm : Object {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
  Comm() : Object(Kind::Comm) {}
  ~Comm() override {
    if (comm) ncclCommDestroy(comm);
  }
};
}  // namespace pdec

using namespace pdec;

#define PDEC_NCCL(call)                                                           \
  do {                                                                            \
    ncclResult_t r__ = (call);                                                    \
    if (r__ != ncclSuccess) {                                                     \
      set_error("%s failed: %s", #call, ncclGetErrorString(r__));                 \
      return PDEC_E_COMM;                                                         \
    }                                                                             \
  } while (0)

extern "C" {

int pdec_comm_unique_id(void* id128) {
  PDEC_REQUIRE(id128, "pdec_comm_unique_id: null");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  PDEC_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, 128);
  return PDEC_OK;
}

int pdec_comm_create(pdec_handle* c, int nranks, int rank, const void* id128) {
  PDEC_REQUIRE(c && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "pdec_comm_create: bad arguments");
  auto C = std::make_unique<Comm>();
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  PDEC_NCCL(ncclCommInitRank(&C->comm, nranks, id, rank));
  C->nranks = nranks;
  C->rank = rank;
  *c = register_object(std::move(C));
  return PDEC_OK;
}

// The rendezvous of ncclCommInitRank blocks until every rank has arrived: one rank that never calls it (it failed earlier, or
// died) leaves the others inside it for ever.  Here the blocking call runs on a helper thread and the caller waits for it with a
// deadline; past the deadline the caller gets PDEC_E_COMM and carries on (the helper stays parked inside RCCL, owning only its
// own shared state, and its communicator -- should the rendezvous complete after all -- is destroyed by the helper itself).  The
// communicator stays an ordinary BLOCKING one: a non-blocking config (ncclCommInitRankConfig, blocking = 0) would make every
// later ncclAllReduce on the update stream an asynchronous call that has to be polled.
int pdec_comm_create_timeout(pdec_handle* c, int nranks, int rank, const void* id128, int timeout_ms) {
  PDEC_REQUIRE(c && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "pdec_comm_create_timeout: bad arguments");
  if (timeout_ms <= 0) return pdec_comm_create(c, nranks, rank, id128);
  struct Shared {
    std::mutex m;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    ncclResult_t res = ncclSuccess;
    ncclComm_t comm = nullptr;
  };
  auto sh = std::make_shared<Shared>();
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  int dev = 0;
  PDEC_HIP(hipGetDevice(&dev));
  std::thread([sh, id, nranks, rank, dev]() {
    ncclComm_t cm = nullptr;
    ncclResult_t r = hipSetDevice(dev) == hipSuccess ? ncclCommInitRank(&cm, nranks, id, rank) : ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lk(sh->m);
    if (sh->abandoned) {                  // the caller gave up: nobody will ever use this communicator
      lk.unlock();
      if (r == ncclSuccess && cm) ncclCommAbort(cm);
      return;
    }
    sh->res = r;
    sh->comm = cm;
    sh->done = true;
    sh->cv.notify_all();
  }).detach();
  {
    std::unique_lock<std::mutex> lk(sh->m);
    if (!sh->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return sh->done; })) {
      sh->abandoned = true;
      set_error("ncclCommInitRank(nranks=%d, rank=%d) did not return within %d ms (a rank missing from the rendezvous?)", nranks, rank,
                timeout_ms);
      return PDEC_E_COMM;
    }
  }
  if (sh->res != ncclSuccess) {
    set_error("ncclCommInitRank(&C->comm, nranks, id, rank) failed: %s", ncclGetErrorString(sh->res));
    return PDEC_E_COMM;
  }
  auto C = std::make_unique<Comm>();
  C->comm = sh->comm;
  C->nranks = nranks;
  C->rank = rank;
  *c = register_object(std::move(C));
  return PDEC_OK;
}

int pdec_allreduce(pdec_handle comm, void* dptr, size_t n, int dtype, void* hip_stream) {
  Comm* C = lookup_as<Comm>(comm, Kind::Comm);
  if (!C) { set_error("pdec_allreduce: not a comm handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(dptr || n == 0, "pdec_allreduce: null");
  if (n == 0) return PDEC_OK;
  PDEC_NCCL(ncclAllReduce(dptr, dptr, n, dtype == PDEC_F64 ? ncclDouble : ncclFloat, ncclSum, C->comm,
                          (hipStream_t)hip_stream));
  return PDEC_OK;
}

int pdec_allreduce_grads_on(pdec_handle comm, pdec_handle mlp, void* hip_stream) {
  Mlp* M = lookup_as<Mlp>(mlp, Kind::Mlp);
  if (!M) { set_error("pdec_allreduce_grads: not an mlp handle"); return PDEC_E_HANDLE; }
  return pdec_allreduce(comm, M->grads.p, (size_t)M->nparams, M->dtype, hip_stream ? hip_stream : (void*)M->stream);
}

int pdec_allreduce_grads(pdec_handle comm, pdec_handle mlp) { return pdec_allreduce_grads_on(comm, mlp, nullptr); }

}  // extern "C"
