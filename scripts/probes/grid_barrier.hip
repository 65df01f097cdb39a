// What does one grid-wide synchronisation cost on an MI355X?  Decides the shape of the single-registration path (round 4): a persistent
// kernel that runs a whole LM / Newton loop needs ~4 such points per iteration; the launch chain it replaces costs ~6-12 us per boundary.
//   (a) sense-reversing barrier: one device-scope atomic per block on a counter, the last arrival bumps a generation word, everybody else
//       spins on it with device-scope loads;
//   (b) "ticket + one worker": the same arrival, but the last block runs a short serial section (a stand-in for the 6x6 solve) before
//       it releases the others;
//   (c) for reference: back-to-back launches of an empty kernel on one stream (the chain's boundary).
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/grid_barrier.hip -o scripts/probes/grid_barrier_probe && scripts/probes/grid_barrier_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__device__ __forceinline__ void grid_barrier(unsigned* count, unsigned* gen, unsigned nblocks, unsigned& my_gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    my_gen++;
    const unsigned arrived = __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (arrived == nblocks) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gen, my_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != my_gen) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_barriers(unsigned* count, unsigned* gen, int rounds, float* sink) {
  unsigned my_gen = 0;
  float v = (float)threadIdx.x;
  for (int r = 0; r < rounds; r++) {
    v = v * 1.0001f + 1.f;
    grid_barrier(count, gen, gridDim.x, my_gen);
  }
  if (v == -1.f) sink[0] = v;
}

// (b): the last arrival does `serial` dependent fma's (~ a 6x6 LDLT + se3 exp is a few hundred) and writes a result the others read
__global__ __launch_bounds__(256) void k_barriers_worker(unsigned* count, unsigned* gen, int rounds, int serial, double* state, float* sink) {
  unsigned my_gen = 0;
  double acc = 0;
  for (int r = 0; r < rounds; r++) {
    __syncthreads();
    if (threadIdx.x == 0) {
      my_gen++;
      const unsigned arrived = __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      if (arrived == gridDim.x) {
        double x = state[0];
        for (int i = 0; i < serial; i++) x = fma(x, 1.0000001, 1e-9);
        state[0] = x;
        __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gen, my_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != my_gen) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    acc += __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == -1.0) sink[0] = (float)acc;
}

__global__ void k_empty(float* sink) {
  if (sink == nullptr) __builtin_trap();
}

int main() {
  unsigned* d;
  double* st;
  float* sink;
  if (hipMalloc(&d, 256) != hipSuccess) return puts("no device"), 2;
  (void)hipMalloc(&st, 64), (void)hipMalloc(&sink, 64);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  const int rounds = 2000;
  for (int nb : {64, 256, 512, 1024}) {
    for (int pass = 0; pass < 2; pass++) {
      (void)hipMemset(d, 0, 256), (void)hipMemset(st, 0, 64);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_barriers, dim3(nb), dim3(256), 0, 0, d, d + 32, rounds, sink);
      (void)hipEventRecord(e1, 0);
      if (hipEventSynchronize(e1) != hipSuccess) return puts("barrier kernel failed"), 1;
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (pass) printf("grid barrier, %4d blocks x 256 threads: %.2f us per barrier\n", nb, 1000.0 * ms / rounds);
    }
    for (int serial : {0, 300, 1000}) {
      (void)hipMemset(d, 0, 256), (void)hipMemset(st, 0, 64);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_barriers_worker, dim3(nb), dim3(256), 0, 0, d, d + 32, rounds, serial, st, sink);
      (void)hipEventRecord(e1, 0);
      if (hipEventSynchronize(e1) != hipSuccess) return puts("worker kernel failed"), 1;
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("   last block works %4d dependent fma: %.2f us per round\n", serial, 1000.0 * ms / rounds);
    }
  }
  // (c) launch chain
  for (int pass = 0; pass < 2; pass++) {
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < rounds; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (pass) printf("empty-kernel launch chain: %.2f us per launch\n", 1000.0 * ms / rounds);
  }
  return 0;
}
