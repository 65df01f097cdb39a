// hgs_comm.h — the one exchange step of the path: an all-gather of fixed-size per-candidate records among the
// ranks of a sharded loop-closure batch (one process per GPU), issued on the engine's own HIP stream.  hgs_comm.hip implements
// it on RCCL (ncclAllGather over xGMI, the library loaded lazily with dlopen); the host emulation of tests/emul supplies a single-rank stand-in.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace hgs {

struct Comm;  // opaque: an RCCL communicator + its rank / world size

constexpr size_t kCommUniqueIdBytes = 128;  // sizeof(ncclUniqueId)

// rank 0 creates the id and hands it to the other ranks out of band (the caller's launcher: MPI, torch.distributed, a file)
int comm_unique_id(void* id_out /* kCommUniqueIdBytes */, char* err, size_t err_cap);
int comm_create(Comm** out, int rank, int world, const void* id /* kCommUniqueIdBytes */, int device, char* err, size_t err_cap);
void comm_destroy(Comm* c);
// tears the communicator down unilaterally: the peers' pending / future collectives on it fail instead of blocking
void comm_abort(Comm* c);
// non-zero (and `err` filled) when the communicator has been aborted or reports an asynchronous error (ncclCommGetAsyncError): polled by a rank
// that waits behind a collective, because a peer's abort is not guaranteed to end this rank's own all-gather kernel
int comm_async_error(Comm* c, char* err, size_t err_cap);
int comm_rank(const Comm* c);
int comm_world(const Comm* c);
// recv[r * bytes_per_rank ...] <- rank r's send[0 .. bytes_per_rank), device pointers, asynchronous on `stream`
int comm_all_gather(Comm* c, const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream, char* err, size_t err_cap);

}  // namespace hgs
