// hgs_comm.hip — RCCL side of the sharded loop-closure batch (hgs_comm.h).  The only collective of the path: one ncclAllGather
// of 112-byte hgs_result records per batch, launched on the engine's stream right behind the kernels that produced the
// records; over xGMI it is latency-bound (512 candidates x 112 B = 56 KB in total).
#include "hgs_comm.h"

#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

namespace hgs {

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

static_assert(sizeof(ncclUniqueId) == kCommUniqueIdBytes, "ncclUniqueId is not 128 bytes");

namespace {
int fail(ncclResult_t r, const char* what, char* err, size_t cap) {
  if (err && cap) snprintf(err, cap, "%s failed: %s", what, ncclGetErrorString(r));
  return 1;
}
}  // namespace

int comm_unique_id(void* id_out, char* err, size_t err_cap) {
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail(r, "ncclGetUniqueId", err, err_cap);
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

int comm_create(Comm** out, int rank, int world, const void* id_bytes, int device, char* err, size_t err_cap) {
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) {
    if (err && err_cap) snprintf(err, err_cap, "hipSetDevice(%d) failed", device);
    return 1;
  }
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  Comm* c = new (std::nothrow) Comm();
  if (!c) return 1;
  c->rank = rank, c->world = world;
  const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(r, "ncclCommInitRank", err, err_cap);
  }
  *out = c;
  return 0;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
}

int comm_rank(const Comm* c) { return c->rank; }
int comm_world(const Comm* c) { return c->world; }

int comm_all_gather(Comm* c, const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream, char* err, size_t err_cap) {
  const ncclResult_t r = ncclAllGather(send, recv, bytes_per_rank, ncclChar, c->comm, stream);
  if (r != ncclSuccess) return fail(r, "ncclAllGather", err, err_cap);
  return 0;
}

}  // namespace hgs
