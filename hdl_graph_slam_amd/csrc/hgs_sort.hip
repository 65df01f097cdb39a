// hgs_sort.hip — device-wide key/value radix sort used to order points along the Hilbert curve (search index)
// and by voxel key (Gaussian cell build).  Plain library sort (rocPRIM); everything domain-specific is in
// hgs_kernels.hip.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "hgs_sort.h"

// (Up to a million keys rocPRIM sorts 1024-item blocks and merges them level by level, one ~6 us launch per level: ten launches for a sweep's 119 k voxel keys.
// Round 6 tried 4096-item blocks — two merge levels less — for sorts of at most 256 k keys: the bigger block sort costs what the merges save, KITTI pipeline
// 0.320 + 0.283 vs 0.324 + 0.288 ms; profiles/r06_ab15_sort_blocks.log.  The library's own configuration stays.)
extern "C" int hgs_sort_pairs_u64_u32(void* temp, size_t* temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                                      uint32_t* vals_out, size_t n, int begin_bit, int end_bit, void* stream) {
  hipError_t e = rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, n, (unsigned)begin_bit, (unsigned)end_bit,
                                           (hipStream_t)stream, false);
  return (int)e;
}

extern "C" int hgs_exclusive_scan_u32(void* temp, size_t* temp_bytes, const uint32_t* in, uint32_t* out, size_t n, void* stream) {
  hipError_t e = rocprim::exclusive_scan(temp, *temp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), (hipStream_t)stream, false);
  return (int)e;
}
