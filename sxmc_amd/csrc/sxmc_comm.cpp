// sxmc_comm.cpp -- the one exchange of the multi-GPU path, straight on RCCL (librccl; xGMI between the GPUs of a
// node): fake experiments shard one-per-rank with no data-path collective (sxmc.cpp:59-145 is a loop of
// independent iterations), and at the end every rank contributes its experiments' intervals
// (point_estimate, lower, upper, coverage per parameter: interval.h:22-27) to ONE all-gather.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/sxmc_hip.h"

// (sxmc_runtime.cpp) launches the calling thread's deferred evaluations and waits for a lazily finished batch unless `s`
// orders with it by itself: the collectives read buffers those evaluations may still be writing
int sx_flush_and_order(hipStream_t s);

struct sxmc_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
};

namespace {
thread_local std::string t_comm_error;
int comm_fail(int code, const std::string& msg) {
  t_comm_error = msg;
  return code;
}
}  // namespace

extern "C" {

const char* sxmc_comm_last_error(void) { return t_comm_error.c_str(); }

int sxmc_comm_init_all(const int* devices, int ndevices, sxmc_comm_t* out) {
  if (!devices || !out || ndevices < 1) return comm_fail(SXMC_ERR_INVALID, "bad arguments");
  std::vector<ncclComm_t> comms((size_t)ndevices);
  ncclResult_t r = ncclCommInitAll(comms.data(), ndevices, devices);
  if (r != ncclSuccess) return comm_fail(SXMC_ERR_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
  for (int i = 0; i < ndevices; i++) {
    sxmc_comm* c = new sxmc_comm;
    c->comm = comms[(size_t)i];
    c->rank = i;
    c->nranks = ndevices;
    c->device = devices[i];
    out[i] = c;
  }
  return SXMC_OK;
}

int sxmc_comm_unique_id(char* id, size_t id_bytes) {
  if (!id || id_bytes < sizeof(ncclUniqueId)) return comm_fail(SXMC_ERR_INVALID, "id buffer too small (128 bytes)");
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) return comm_fail(SXMC_ERR_HIP, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
  std::memcpy(id, &u, sizeof u);
  return SXMC_OK;
}

int sxmc_comm_init_rank(const char* id, size_t id_bytes, int nranks, int rank, sxmc_comm_t* out) {
  if (!id || !out || id_bytes < sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks) {
    return comm_fail(SXMC_ERR_INVALID, "bad arguments");
  }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  sxmc_comm* c = new sxmc_comm;
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return comm_fail(SXMC_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  c->rank = rank;
  c->nranks = nranks;
  (void)hipGetDevice(&c->device);
  *out = c;
  return SXMC_OK;
}

int sxmc_comm_rank(sxmc_comm_t c, int* rank, int* nranks) {
  if (!c || !rank || !nranks) return comm_fail(SXMC_ERR_INVALID, "null argument");
  return sxmc_comm_query(c, rank, nranks, nullptr);
}

int sxmc_comm_query(sxmc_comm_t c, int* rank, int* nranks, int* device) {
  if (!c || !c->comm) return comm_fail(SXMC_ERR_INVALID, "null communicator");
  // asked of the communicator itself, not of what the caller passed in when it was made
  int v = 0;
  ncclResult_t r;
  if (rank) {
    if ((r = ncclCommUserRank(c->comm, &v)) != ncclSuccess)
      return comm_fail(SXMC_ERR_HIP, std::string("ncclCommUserRank: ") + ncclGetErrorString(r));
    *rank = v;
  }
  if (nranks) {
    if ((r = ncclCommCount(c->comm, &v)) != ncclSuccess)
      return comm_fail(SXMC_ERR_HIP, std::string("ncclCommCount: ") + ncclGetErrorString(r));
    *nranks = v;
  }
  if (device) {
    if ((r = ncclCommCuDevice(c->comm, &v)) != ncclSuccess)
      return comm_fail(SXMC_ERR_HIP, std::string("ncclCommCuDevice: ") + ncclGetErrorString(r));
    *device = v;
  }
  return SXMC_OK;
}

int sxmc_comm_async_error(sxmc_comm_t c, int* failed) {
  if (!c || !c->comm || !failed) return comm_fail(SXMC_ERR_INVALID, "null argument");
  ncclResult_t state = ncclSuccess;
  ncclResult_t r = ncclCommGetAsyncError(c->comm, &state);
  if (r != ncclSuccess) return comm_fail(SXMC_ERR_HIP, std::string("ncclCommGetAsyncError: ") + ncclGetErrorString(r));
  *failed = (state != ncclSuccess && state != ncclInProgress) ? 1 : 0;
  if (*failed) t_comm_error = std::string("asynchronous RCCL error: ") + ncclGetErrorString(state);
  return SXMC_OK;
}

int sxmc_comm_abort(sxmc_comm_t c) {
  if (!c) return SXMC_OK;
  // frees the communicator AND ends any collective of it that is still waiting for a peer that will not come
  if (c->comm) (void)ncclCommAbort(c->comm);
  delete c;
  return SXMC_OK;
}

int sxmc_comm_allgather_f32(sxmc_comm_t c, const float* d_send, float* d_recv, size_t count, sxmc_stream_t s) {
  if (!c || !d_send || !d_recv) return comm_fail(SXMC_ERR_INVALID, "null argument");
  if (int rc = sx_flush_and_order((hipStream_t)s)) return comm_fail(rc, sxmc_last_error());
  ncclResult_t r = ncclAllGather(d_send, d_recv, count, ncclFloat32, c->comm, (hipStream_t)s);
  if (r != ncclSuccess) return comm_fail(SXMC_ERR_HIP, std::string("ncclAllGather: ") + ncclGetErrorString(r));
  return SXMC_OK;
}

int sxmc_comm_destroy(sxmc_comm_t c) {
  if (!c) return SXMC_OK;
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
  return SXMC_OK;
}

}  // extern "C"
