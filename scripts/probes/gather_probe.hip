// gather_probe.hip — micro-benchmark behind the "packet walk vs per-lane traversal" decision: how many 128-byte
// records per second can the lanes of a wave fetch when every lane chases its OWN chain of records (a per-lane tree descent)
// compared with one record per wave broadcast to all lanes (the packet walk)?  Dependent chains, occupancy as in the product
// kernels (256-thread blocks, up to 8 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_private(const float4* __restrict__ recs, unsigned nrec, unsigned window, int iters, int loads, float* out) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  const unsigned wave_base = ((gid >> 6) * 2654435761u) % nrec;
  unsigned idx = (wave_base + (threadIdx.x & 63) * 7u % window) % nrec;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    const float4* r = recs + 8 * (size_t)idx;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (k < loads) v[k] = r[k];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (k < loads) s += v[k].x + v[k].y + v[k].z + v[k].w;
    acc += s;
    // the next record depends on the data (a descent), stays within `window` records of the wave's region
    const unsigned step = (__float_as_uint(s) >> 3) % window;
    idx = (wave_base + step + (unsigned)it * 17u) % nrec;
  }
  out[gid] = acc;
}

__global__ __launch_bounds__(256) void k_packet(const float4* __restrict__ recs, unsigned nrec, unsigned window, int iters, float* out) {
  __shared__ __attribute__((aligned(128))) float slots[4][32];
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned wave_base = ((gid >> 6) * 2654435761u) % nrec;
  unsigned idx = wave_base;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    const float v = reinterpret_cast<const float*>(recs + 8 * (size_t)idx)[lane & 31];
    __builtin_amdgcn_wave_barrier();
    slots[wave][lane & 31] = v;
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float4 r = reinterpret_cast<const float4*>(slots[wave])[k];
      s += r.x * (lane + 1) + r.y + r.z + r.w;
    }
    acc += s;
    const unsigned step = (unsigned)__builtin_amdgcn_readfirstlane((int)((__float_as_uint(s) >> 3) % window));
    idx = (wave_base + step + (unsigned)it * 17u) % nrec;
  }
  out[gid] = acc;
}

int main() {
  const int iters = 64;
  for (size_t mb : {2, 16, 128}) {
    const unsigned nrec = (unsigned)(mb * 1024 * 1024 / 128);
    std::vector<float> h((size_t)nrec * 32);
    for (auto& x : h) x = (float)(rand() % 1000) * 0.37f;
    float4* d;
    float* out;
    const int blocks = 256 * 8 * 4;  // 8 blocks of 4 waves per CU
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (unsigned window : {16u, 256u, nrec}) {
      for (int loads : {2, 8}) {
        hipLaunchKernelGGL(k_private, dim3(blocks), dim3(256), 0, 0, d, nrec, window, iters, loads, out);
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_private, dim3(blocks), dim3(256), 0, 0, d, nrec, window, iters, loads, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ms /= 5;
        const double recs_per_s = (double)blocks * 256 * iters / (ms * 1e-3);
        printf("private  table %4zu MB window %8u loads/lane %d x16B : %7.3f ms  %8.2f G lane-records/s  %8.1f GB/s gathered\n", mb, window, loads, ms, recs_per_s * 1e-9,
               recs_per_s * loads * 16 * 1e-9);
      }
      hipLaunchKernelGGL(k_packet, dim3(blocks), dim3(256), 0, 0, d, nrec, window, iters, out);
      hipEventRecord(a);
      for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_packet, dim3(blocks), dim3(256), 0, 0, d, nrec, window, iters, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      ms /= 5;
      const double steps_per_s = (double)blocks * 4 * iters / (ms * 1e-3);
      printf("packet   table %4zu MB window %8u (1 record / wave step)  : %7.3f ms  %8.2f G wave-steps/s = %8.2f G lane-records/s equivalent\n", mb, window, ms, steps_per_s * 1e-9,
             steps_per_s * 64 * 1e-9);
    }
    hipFree(d), hipFree(out);
  }
  return 0;
}
