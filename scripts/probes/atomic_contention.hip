// What do the flush atomics of a single-cloud k_ndt_pass cost?  465 blocks finish together; each adds 172 64-bit totals to the SAME 172 addresses
// and then takes a ticket on ONE address.  Measures, for N blocks of 256 threads:
//   (a) an empty kernel of N blocks (the floor);
//   (b) every block: one returning atomicAdd on one address (the tile ticket);
//   (c) every block: 172 non-returning 64-bit atomicAdds on 172 addresses, replicated R ways (address set = block & (R - 1)), wait, ticket;
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/atomic_contention.hip -o scripts/probes/atomic_contention_probe && scripts/probes/atomic_contention_probe
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ __launch_bounds__(256) void k_empty(unsigned* sink) {
  if (sink == nullptr) __builtin_trap();
}
__global__ __launch_bounds__(256) void k_ticket(unsigned* counter, unsigned* sink) {
  __shared__ unsigned last;
  if (threadIdx.x == 0) last = atomicAdd(counter, 1u) + 1u == gridDim.x ? 1u : 0u;
  __syncthreads();
  if (last && threadIdx.x == 0) sink[0] = 1u;
}
__global__ __launch_bounds__(256) void k_flush(unsigned long long* totals /* [R][172] */, int replicas, unsigned* counter, unsigned* sink, int with_ticket) {
  __shared__ unsigned last;
  const int t = threadIdx.x;
  if (t < 172) atomicAdd(&totals[(blockIdx.x & (replicas - 1)) * 172 + t], (unsigned long long)(t + 1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (with_ticket) {
    if (t == 0) last = atomicAdd(counter, 1u) + 1u == gridDim.x ? 1u : 0u;
    __syncthreads();
    if (last && t == 0) sink[0] = 1u;
  }
}

template <typename F>
static float time_us(F&& launch, int reps = 200) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  for (int i = 0; i < 10; i++) launch();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; i++) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return 1000.f * ms / reps;
}

int main() {
  unsigned *counter, *sink;
  unsigned long long* totals;
  if (hipMalloc(&counter, 256) != hipSuccess) return puts("no device"), 2;
  (void)hipMalloc(&sink, 256), (void)hipMalloc(&totals, 64 * 172 * 8);
  (void)hipMemset(counter, 0, 256), (void)hipMemset(totals, 0, 64 * 172 * 8);
  for (int nb : {116, 232, 465, 930}) {
    const float t_empty = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(nb), dim3(256), 0, 0, sink); });
    const float t_ticket = time_us([&] { hipLaunchKernelGGL(k_ticket, dim3(nb), dim3(256), 0, 0, counter, sink); });
    printf("%4d blocks: empty %.2f us, one returning ticket atomic per block %.2f us (+%.1f ns per block)\n", nb, t_empty, t_ticket, 1000.f * (t_ticket - t_empty) / nb);
    for (int R : {1, 2, 8, 32}) {
      const float a = time_us([&] { hipLaunchKernelGGL(k_flush, dim3(nb), dim3(256), 0, 0, totals, R, counter, sink, 0); });
      const float b = time_us([&] { hipLaunchKernelGGL(k_flush, dim3(nb), dim3(256), 0, 0, totals, R, counter, sink, 1); });
      printf("      172 totals x %2d replicas: adds only %.2f us, adds + ticket %.2f us\n", R, a, b);
    }
  }
  return 0;
}
