#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
// rocPRIM radix_sort_pairs<uint64 keys, uint32 values> over bits [begin_bit, end_bit). With temp == nullptr only
// *temp_bytes is written. Returns the hipError_t as int.
int hgs_sort_pairs_u64_u32(void* temp, size_t* temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                           size_t n, int begin_bit, int end_bit, void* stream);
// rocPRIM exclusive prefix sum of uint32 (out[i] = sum of in[0..i)); same temp protocol.
int hgs_exclusive_scan_u32(void* temp, size_t* temp_bytes, const uint32_t* in, uint32_t* out, size_t n, void* stream);
#ifdef __cplusplus
}
#endif
