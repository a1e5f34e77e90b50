// comm.hip -- the build's one collective: a sum all-reduce of the flattened policy/critic
// gradient over RCCL (xGMI).  The reference has no collective at all; this
// is the data-parallel addition (one process per GPU, batch of trajectories sharded).
// At ~21 k floats the op is latency-bound, so it is a single fused buffer per network.
#include <rccl/rccl.h>

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

int pdec_allreduce(pdec_handle comm, void* dptr, size_t n, int dtype, void* hip_stream) {
  Comm* C = lookup_as<Comm>(comm, Kind::Comm);
  if (!C) { set_error("pdec_allreduce: not a comm handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(dptr || n == 0, "pdec_allreduce: null");
  if (n == 0) return PDEC_OK;
  PDEC_NCCL(ncclAllReduce(dptr, dptr, n, dtype == PDEC_F64 ? ncclDouble : ncclFloat, ncclSum, C->comm,
                          (hipStream_t)hip_stream));
  return PDEC_OK;
}

int pdec_allreduce_grads(pdec_handle comm, pdec_handle mlp) {
  Mlp* M = lookup_as<Mlp>(mlp, Kind::Mlp);
  if (!M) { set_error("pdec_allreduce_grads: not an mlp handle"); return PDEC_E_HANDLE; }
  return pdec_allreduce(comm, M->grads.p, (size_t)M->nparams, M->dtype, (void*)M->stream);
}

}  // extern "C"
