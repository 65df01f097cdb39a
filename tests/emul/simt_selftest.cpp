// Self-test of the SIMT emulation shim (tests/emul/simt/hip/hip_runtime.h + simt_runtime.cpp): small kernels whose results are
// known, exercising what the product kernels rely on — ballot / shuffle / readlane among the LIVE lanes of a wave, lanes that
// have returned, block barriers with shared memory across waves, atomics across blocks, grids in y, 64-bit shuffles.
// Prints "ok" and returns 0, or the first failed check.  Built and run by tests/test_simt_emulator.py.
#include <hip/hip_runtime.h>

#include <vector>

#define CHECK(cond)                                              \
  do {                                                           \
    if (!(cond)) {                                               \
      printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      return 1;                                                  \
    }                                                            \
  } while (0)

__global__ void k_ballot(unsigned long long* out, int n_live) {
  const int lane = threadIdx.x & 63;
  if (lane >= n_live) return;  // returned lanes take no part and contribute 0
  const unsigned long long even = __ballot((lane & 1) == 0);
  if (lane == 0) out[blockIdx.x] = even;
}

__global__ void k_wave_sum(double* out) {
  double v = (double)threadIdx.x;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = v;
}

__global__ void k_readlane(int* out) {
  const int lane = threadIdx.x & 63;
  const int v = __builtin_amdgcn_readlane(lane * 3 + 1, 17);
  out[threadIdx.x] = v;
}

__global__ void k_block_reduce(int* out, int n) {
  __shared__ int part[4];
  const int t = threadIdx.x;
  if (t < 4) part[t] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + t;
  int v = i < n ? i : 0;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((t & 63) == 0) part[t >> 6] = v;
  __syncthreads();
  if (t == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

__global__ void k_barrier_with_early_exit(int* out) {
  __shared__ int flag;
  if (threadIdx.x >= 100) return;  // a whole wave and part of another one leave before the barrier
  if (threadIdx.x == 99) flag = 7;
  __syncthreads();
  out[threadIdx.x] = flag;
}

__global__ void k_grid_y(int* out) { out[blockIdx.y * gridDim.x + blockIdx.x] = (int)(blockIdx.y * 100 + blockIdx.x) + (int)blockDim.x; }

int main() {
  {
    std::vector<unsigned long long> out_v(3, 0);
    unsigned long long* out = out_v.data();  // kernel arguments are captured by value, like on the device
    hipLaunchKernelGGL(k_ballot, dim3(3), dim3(64), 0, nullptr, out, 64);
    CHECK(out[0] == 0x5555555555555555ull && out[2] == out[0]);
    hipLaunchKernelGGL(k_ballot, dim3(1), dim3(64), 0, nullptr, out, 10);
    CHECK(out[0] == 0x155ull);  // lanes 0,2,4,6,8 of the ten live ones
  }
  {
    std::vector<double> out_v(4, 0);
    double* out = out_v.data();
    hipLaunchKernelGGL(k_wave_sum, dim3(1), dim3(256), 0, nullptr, out);
    for (int w = 0; w < 4; w++) CHECK(out[w] == 64.0 * (64 * w) + 2016.0);
  }
  {
    std::vector<int> out_v(128, 0);
    int* out = out_v.data();
    hipLaunchKernelGGL(k_readlane, dim3(1), dim3(128), 0, nullptr, out);
    for (int i = 0; i < 128; i++) CHECK(out[i] == 17 * 3 + 1);
  }
  {
    int total_v = 0;
    int* total = &total_v;
    const int n = 1000;
    hipLaunchKernelGGL(k_block_reduce, dim3((n + 255) / 256), dim3(256), 0, nullptr, total, n);
    CHECK(total_v == n * (n - 1) / 2);
  }
  {
    std::vector<int> out_v(256, -1);
    int* out = out_v.data();
    hipLaunchKernelGGL(k_barrier_with_early_exit, dim3(1), dim3(256), 0, nullptr, out);
    for (int i = 0; i < 100; i++) CHECK(out[i] == 7);
    CHECK(out[100] == -1);
  }
  {
    std::vector<int> out_v(6, 0);
    int* out = out_v.data();
    hipLaunchKernelGGL(k_grid_y, dim3(3, 2), dim3(32), 0, nullptr, out);
    CHECK(out[0] == 32 && out[2] == 34 && out[3] == 132 && out[5] == 134);
  }
  {
    void* p = nullptr;
    CHECK(hipMalloc(&p, 1000) == hipSuccess && p && ((uintptr_t)p % 256) == 0);
    CHECK(hipMemsetAsync(p, 0xff, 1000, nullptr) == hipSuccess && ((unsigned char*)p)[999] == 0xff);
    CHECK(hipFree(p) == hipSuccess);
    CHECK(__clz(1) == 31 && __clz(0) == 32 && __ffsll(8) == 4 && __popcll(0xf0f0ull) == 8);
    CHECK(__builtin_amdgcn_fmed3f(3.f, 1.f, 2.f) == 2.f && __float_as_int(1.0f) == 0x3f800000);
  }
  printf("ok\n");
  return 0;
}
