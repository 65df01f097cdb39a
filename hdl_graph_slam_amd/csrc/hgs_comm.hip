// hgs_comm.hip — RCCL side of the sharded loop-closure batch (hgs_comm.h).  The only collectives of the path: per batch one
// ncclAllGather of a 16-byte header per rank (shard size + status) and one of the ranks' 112-byte hgs_result records, launched on
// the engine's stream; over xGMI both are latency-bound (512 candidates x 112 B = 56 KB in total).
//
// RCCL is loaded lazily (dlopen at the first hgs_comm_* call): the registration path itself has no RCCL dependency, a machine
// without librccl can still build, load and run the single-GPU library, and a process that has already loaded an RCCL (torch
// bundles one) gets THAT one — the same SONAME resolves to the loaded object — instead of a second copy.
#include "hgs_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: every call goes through the table below

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

namespace hgs {

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

static_assert(sizeof(ncclUniqueId) == kCommUniqueIdBytes, "ncclUniqueId is not 128 bytes");

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;  // optional: older libraries without it are only covered by the deadline
  char why[256] = "";
};

// nullptr (and `err` filled) when no usable librccl can be loaded
const Rccl* rccl(char* err, size_t cap) {
  static Rccl table;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    // an RCCL the process already holds (RTLD_NOLOAD matches the names and the SONAME it was loaded under), then $ROCM_PATH's, then the search path
    for (int k = 0; k < 2 && !table.lib; k++) table.lib = dlopen(names[k], RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    const char* env = std::getenv("ROCM_PATH");
    char from_env[512] = "";
    if (env) snprintf(from_env, sizeof(from_env), "%s/lib/librccl.so", env);
    if (!table.lib && from_env[0]) table.lib = dlopen(from_env, RTLD_NOW | RTLD_LOCAL);
    for (const char* n : names)
      if (!table.lib) table.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!table.lib) {
      snprintf(table.why, sizeof(table.why), "librccl.so not found (%s)", dlerror());
      return;
    }
    table.GetUniqueId = reinterpret_cast<decltype(table.GetUniqueId)>(dlsym(table.lib, "ncclGetUniqueId"));
    table.CommInitRank = reinterpret_cast<decltype(table.CommInitRank)>(dlsym(table.lib, "ncclCommInitRank"));
    table.CommDestroy = reinterpret_cast<decltype(table.CommDestroy)>(dlsym(table.lib, "ncclCommDestroy"));
    table.CommAbort = reinterpret_cast<decltype(table.CommAbort)>(dlsym(table.lib, "ncclCommAbort"));
    table.AllGather = reinterpret_cast<decltype(table.AllGather)>(dlsym(table.lib, "ncclAllGather"));
    table.GetErrorString = reinterpret_cast<decltype(table.GetErrorString)>(dlsym(table.lib, "ncclGetErrorString"));
    table.CommGetAsyncError = reinterpret_cast<decltype(table.CommGetAsyncError)>(dlsym(table.lib, "ncclCommGetAsyncError"));
    if (!table.GetUniqueId || !table.CommInitRank || !table.CommDestroy || !table.CommAbort || !table.AllGather || !table.GetErrorString) {
      snprintf(table.why, sizeof(table.why), "librccl.so lacks an expected ncclXxx symbol");
      table.lib = nullptr;
    }
  });
  if (!table.lib) {
    if (err && cap) snprintf(err, cap, "RCCL unavailable: %s", table.why);
    return nullptr;
  }
  return &table;
}

int fail(const Rccl* R, ncclResult_t r, const char* what, char* err, size_t cap) {
  if (err && cap) snprintf(err, cap, "%s failed: %s", what, R->GetErrorString(r));
  return 1;
}
}  // namespace

int comm_unique_id(void* id_out, char* err, size_t err_cap) {
  const Rccl* R = rccl(err, err_cap);
  if (!R) return 2;
  ncclUniqueId id;
  const ncclResult_t r = R->GetUniqueId(&id);
  if (r != ncclSuccess) return fail(R, r, "ncclGetUniqueId", err, err_cap);
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

int comm_create(Comm** out, int rank, int world, const void* id_bytes, int device, char* err, size_t err_cap) {
  *out = nullptr;
  const Rccl* R = rccl(err, err_cap);
  if (!R) return 2;
  if (hipSetDevice(device) != hipSuccess) {
    if (err && err_cap) snprintf(err, err_cap, "hipSetDevice(%d) failed", device);
    return 1;
  }
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  Comm* c = new (std::nothrow) Comm();
  if (!c) return 1;
  c->rank = rank, c->world = world;
  const ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(R, r, "ncclCommInitRank", err, err_cap);
  }
  *out = c;
  return 0;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  const Rccl* R = rccl(nullptr, 0);
  if (c->comm && R) (void)R->CommDestroy(c->comm);
  delete c;
}

// Tears the communicator down without the peers' cooperation: their pending and future collectives on it return an error
// instead of blocking.  The last resort of a rank that cannot take part in a collective the others have entered or will enter.
void comm_abort(Comm* c) {
  if (!c) return;
  const Rccl* R = rccl(nullptr, 0);
  if (c->comm && R) (void)R->CommAbort(c->comm);
  c->comm = nullptr;
}

int comm_async_error(Comm* c, char* err, size_t err_cap) {
  if (!c || !c->comm) {
    if (err && err_cap) snprintf(err, err_cap, "the communicator has been aborted");
    return 1;
  }
  const Rccl* R = rccl(nullptr, 0);
  if (!R || !R->CommGetAsyncError) return 0;
  ncclResult_t async = ncclSuccess;
  const ncclResult_t r = R->CommGetAsyncError(c->comm, &async);
  if (r != ncclSuccess) return fail(R, r, "ncclCommGetAsyncError", err, err_cap);
  if (async != ncclSuccess && async != ncclInProgress) return fail(R, async, "a collective on this communicator", err, err_cap);
  return 0;
}

int comm_rank(const Comm* c) { return c->rank; }
int comm_world(const Comm* c) { return c->world; }

int comm_all_gather(Comm* c, const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream, char* err, size_t err_cap) {
  const Rccl* R = rccl(err, err_cap);
  if (!R) return 2;
  if (!c->comm) {
    if (err && err_cap) snprintf(err, err_cap, "the communicator has been aborted");
    return 1;
  }
  const ncclResult_t r = R->AllGather(send, recv, bytes_per_rank, ncclChar, c->comm, stream);
  if (r != ncclSuccess) return fail(R, r, "ncclAllGather", err, err_cap);
  return 0;
}

}  // namespace hgs
