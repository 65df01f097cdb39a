// hgs_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the scan-matching hot path that
// hdl_graph_slam reaches through select_registration_method (src/hdl_graph_slam/registrations.cpp:22-124):
//   * search index  : Hilbert sort keys, implicit bounding-interval tree build (replaces the kd-trees of
//                     pcl::Registration::tree_ and fast_gicp)
//   * GICP          : k-NN covariance pre-pass, fused 1-NN correspondence + J^T M J accumulation, LM trial error,
//                     on-device 6x6 solve / LM control (fast_gicp::FastGICP, registrations.cpp:27-36)
//   * NDT           : Gaussian voxel table build, DIRECT1/7 derivative pass, on-device Newton control
//                     (pclomp::NormalDistributionsTransform, registrations.cpp:101-120)
//   * VGICP         : Gaussian voxel map of the target, voxel-lookup linearize / trial error on the GICP LM state machine
//                     (fast_gicp::FastVGICP, registrations.cpp:48-56)
//   * fitness score : pcl::Registration::getFitnessScore (information_matrix_calculator.cpp:49-80)
//   * next rows     : prefilter (apps/prefiltering_nodelet.cpp:131-182), map cloud (map_cloud_generator.cpp:13-51)
// All kernels are batched: blockIdx.y selects the problem (loop-closure candidate, loop_detector.hpp:135-154);
// the odometry path is the batch of one.  The NN-based stages are bound by the instruction issue of the exact tree walk
// (hgs_wave_bvh.h), the per-point algebra by HBM: no MFMA (the contraction is 6x6), coalesced float4 loads, wave64
// shuffle reductions, deterministic two-stage sums.
#include <hip/hip_runtime.h>
#include "../../include/hgs_registration.h"
#include "hgs_device.h"
#include "hgs_wave_bvh.h"

namespace hgs {

// ------------------------------------------------------------------------------------------------ helpers
#ifndef HGS_OPAQUE_POINTER  // (the host emulation of tests/emul supplies its own spelling of these two)
#define HGS_OPAQUE_POINTER(p) asm volatile("" : "+v"(p))
#define HGS_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// lane id recomputed where it is needed (2 VALU) instead of kept in a register across a long search: not CSE'd with an earlier one
#define HGS_LANE_ID(dst) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(dst))
#endif
#ifndef HGS_COMPILER_MEMORY_BARRIER
#define HGS_COMPILER_MEMORY_BARRIER() asm volatile("" ::: "memory")
#endif
#ifndef HGS_LOAD_GLOBAL_XYZ  // (the host emulation supplies its own spelling of these as well)
// a float4 behind a pointer read from a descriptor in memory: the compiler cannot tell its address space and issues flat_load, which
// counts on vmcnt AND lgkmcnt (every LDS wait then waits for it too); this is global_load
__device__ __forceinline__ float4 hgs_load_global_xyz(const float4* p) {  // .w = 0: three dwords loaded, no fourth register to wait for
  const __attribute__((address_space(1))) float* g = (const __attribute__((address_space(1))) float*)(p);
  return make_float4(g[0], g[1], g[2], 0.f);
}
#define HGS_LOAD_GLOBAL_XYZ(p) hgs_load_global_xyz(p)
// `off` = 0 the compiler cannot see through, added to a __shared__ address: the object is re-read where it is used (like HGS_OPAQUE_POINTER)
// but stays an LDS address (ds_read instead of flat_load)
#define HGS_OPAQUE_OFFSET(off) asm volatile("" : "+v"(off))
// s_waitcnt vmcnt(0) the compiler's own wait insertion knows about (gfx9 encoding: expcnt and lgkmcnt fields at their maxima)
#define HGS_WAIT_VMEM_TRACKED() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif
#ifndef HGS_LINEARIZE_WAVES
#define HGS_LINEARIZE_WAVES 7  // waves per SIMD k_gicp_linearize is compiled for (A/B knob; 6 / 7 / 8 measured equal: 68 VGPRs, no scratch)
#endif
#ifndef HGS_KNN_WAVES
#define HGS_KNN_WAVES 5  // waves per SIMD k_knn_cov is compiled for.  Measured on the 64-cloud pass: the compiler's own choice (108 VGPRs, 4 waves) 3.98 ms,
#endif                  // 5 waves (96 VGPRs, 64 bytes of spills outside the walk) 3.66 ms, 6 waves (80 VGPRs, spills inside the insertion chains) 6.7 ms
// (the 32- / 64-slot lists and the instantiation with the per-point eigen-decomposition keep the compiler's choice: capped at 96 VGPRs the
// latter spills 536 bytes and its covariance pass takes 5.8 instead of 3.5 ms)
#define HGS_KNN_OCCUPANCY __attribute__((amdgpu_waves_per_eu((KMAX <= 20 && REG != 1) ? HGS_KNN_WAVES : 1)))
// The block-per-problem control kernels (k_gicp_solve: 4 waves, k_gicp_decide: 1 wave) run next to the other lanes' point kernels.  Compiled freely they
// take 108 / 118 VGPRs: a wave of theirs then fits a SIMD only after TWO of k_gicp_linearize's 72-VGPR waves have left it (7 x 72 = 504 of 512 are
// taken), and the 30 k-block launch refills every hole with its own next block first — round 4's lanes profile shows k_gicp_solve at 40.7 us on
// average (13.6 alone, 504 max).  With waves_per_eu(6) they take 80 VGPRs (8 / 100 bytes of scratch in the one-lane LM step) = the hole ONE departing
// linearize wave leaves; s_setprio puts their one working lane in front of the 6-7 point-kernel waves that share its SIMD's issue slots; reduce_tiles
// keeps 8 loads in flight instead of 4 (same additions in the same order).  Same-box A/B on the 64 x 119 k batch (profiles/r05_ab_control_kernels.log):
// 5391 -> 5437 registrations/s FROBENIUS (+0.8 %), PLANE within noise: what the inversion really cost the timed configuration was under 1 %.
#ifndef HGS_CONTROL_WAVES
#define HGS_CONTROL_WAVES 6
#endif
#ifndef HGS_CONTROL_PRIO
#define HGS_CONTROL_PRIO 3
#endif
#ifndef HGS_REDUCE_UNROLL
#define HGS_REDUCE_UNROLL 2
#endif
#ifndef HGS_NDT_FLUSH_ROT
#define HGS_NDT_FLUSH_ROT 1  // ndt_flush: bank-conflict-free rotation (0: round 4's; A/B)
#endif
#ifndef HGS_FITNESS_WAVES
#define HGS_FITNESS_WAVES 8
#endif
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}
__device__ __forceinline__ bool finite3(const float4& p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

// Tile order.  Workgroups are dealt to the 8 XCDs round-robin, so with the identity mapping below spatially adjacent
// tiles run on different XCDs at the same time.  The XCD-aware alternative (each XCD a contiguous slab of tiles, so that
// its private L2 holds one spatial slab of the target tree) was measured on the loop-closure batch and is NOT faster:
// linearize 2.48-2.55 ms per step with slabs vs 2.41-2.42 ms with the identity, covariances and fitness unchanged — all
// XCDs working on the same region at the same moment keeps the shared Infinity Cache footprint small, which matters more
// here than the per-XCD L2 (the tree records are re-read from neighbouring waves within microseconds either way).
#ifdef HGS_XCD_SLABS  // A/B knob (round 3 re-test): launch grids padded to a multiple of 8 in x, XCD x works on the x-th contiguous slab of tiles
__device__ __forceinline__ int xcd_tile(int bid, int /*ntiles*/) { return (bid & 7) * ((int)gridDim.x >> 3) + (bid >> 3); }
#define HGS_GRID_X(n) ((((n) + 7) / 8) * 8)
#else
__device__ __forceinline__ int xcd_tile(int bid, int /*ntiles*/) { return bid; }
#define HGS_GRID_X(n) (n)
#endif

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// End-of-round bookkeeping, called by ONE thread of each of the B per-problem blocks of the last kernel of a round:
// counts finished problems and completed rounds in HBM and mirrors both into host-mapped pinned memory, so that the
// host can keep the queue filled and stop enqueueing rounds without ever synchronising the stream.
// release_done: the caller has stored results into host-mapped memory that the host reads as soon as it sees host_done (k_gicp_linearize<true>'s early result).
__device__ __forceinline__ void progress_tick(Progress p, bool finished_now, bool release_done = false) {
  // the mirror lives in host-mapped memory: system-scope atomic stores reach it without a __threadfence_system(), whose cache
  // write-back + invalidate would hit every block of a kernel that is still computing (k_ndt_pass ticks from inside)
  if (finished_now && atomicAdd(&p.dev[0], 1) + 1 == p.B) {
    if (release_done) __hip_atomic_store(const_cast<int*>(p.host_done), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(const_cast<int*>(p.host_done), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int t = atomicAdd(&p.dev[1], 1) + 1;
  if (t % p.B == 0) __hip_atomic_store(const_cast<int*>(p.host_rounds), t / p.B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Sum N per-thread doubles over a 256-thread block in a fixed order; thread k < N stores result k.
template <int N>
__device__ __forceinline__ void block_reduce_store(const double* acc, double* out, double* lds /* [4*N] */, int lane, int wave) {
#pragma unroll
  for (int k = 0; k < N; k++) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) lds[wave * N + k] = v;
  }
  __syncthreads();
  if (wave == 0 && lane < N) out[lane] = (lds[lane] + lds[N + lane]) + (lds[2 * N + lane] + lds[3 * N + lane]);
}
template <int N>
__device__ __forceinline__ void block_reduce_store(const double* acc, double* out, double* lds /* [4*N] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) lds[wave * N + k] = v;
  }
  __syncthreads();
  if (threadIdx.x < N) out[threadIdx.x] = (lds[threadIdx.x] + lds[N + threadIdx.x]) + (lds[2 * N + threadIdx.x] + lds[3 * N + threadIdx.x]);
}

// N per-lane doubles summed over the wave; lane 0 stores sum k at row[slot[k]].
template <int N>
__device__ __forceinline__ void wave_sums_to(const double (&v)[N], const int (&slot)[N], double* row, int lane) {
#pragma unroll
  for (int k = 0; k < N; k++) {
    const double s = wave_sum(v[k]);
    if (lane == 0) row[slot[k]] = s;
  }
}

// ---- transposing wave sums (round 3) -------------------------------------------------------------------------------------------
// Summing N doubles over the wave one by one costs N x 6 shuffle steps (a 64-bit __shfl_down is two cross-lane moves plus
// an add: ~500 instructions for the 28 sums of k_gicp_linearize, a third of what the kernel ran outside the tree walk).  The swap
// instructions of gfx950 halve the number of VALUES at every step instead: v_permlane32_swap exchanges the upper half of one register
// with the lower half of another, so ONE add folds two values over the wave's halves (lanes < 32 keep the first value's partial
// sums, lanes >= 32 the second's), v_permlane16_swap does the same over 16-lane rows, and only the last 16 -> 1 steps run per
// register (DPP row shifts).  32 values -> 16 -> 8 registers -> 8 row reductions.  The order of the additions is fixed: results are
// bitwise reproducible (they differ in the last bits from the shuffle tree's, which associated differently).
__device__ __forceinline__ double swap_add32(double x, double y) {  // lanes < 32: x[l] + x[l + 32]; lanes >= 32: y[l - 32] + y[l]
  const unsigned long long bx = (unsigned long long)__double_as_longlong(x), by = (unsigned long long)__double_as_longlong(y);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)bx, (unsigned)by, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(bx >> 32), (unsigned)(by >> 32), false, false);
  return __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
__device__ __forceinline__ double swap_add16(double x, double y) {  // rows 0 / 2: x's rows 0+1 / 2+3; rows 1 / 3: y's rows 0+1 / 2+3
  const unsigned long long bx = (unsigned long long)__double_as_longlong(x), by = (unsigned long long)__double_as_longlong(y);
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)bx, (unsigned)by, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(bx >> 32), (unsigned)(by >> 32), false, false);
  return __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
__device__ __forceinline__ double row_sum16(double v) {  // lane 15 of every 16-lane row: the row's sum (row_shr:1, 2, 4, 8, zero fill)
#define HGS_ROW_STEP(ctrl)                                                                                                         \
  {                                                                                                                                \
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);                                                       \
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)b, ctrl, 0xf, 0xf, true);                                         \
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(b >> 32), ctrl, 0xf, 0xf, true);                                 \
    v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));                                                    \
  }
  HGS_ROW_STEP(0x111) HGS_ROW_STEP(0x112) HGS_ROW_STEP(0x114) HGS_ROW_STEP(0x118)
#undef HGS_ROW_STEP
  return v;
}
// even[j] / odd[j] = the lane's values for output slots 2j / 2j + 1, j < 14 (slots 0..27); the wave's sums land in row[0..27].
// Slot p = 4i + r ends up in lane 15 of row r of register i: r = 0 even[2i], 1 odd[2i], 2 even[2i+1], 3 odd[2i+1].
__device__ __forceinline__ void wave_sums28_to(const double (&even)[14], const double (&odd)[14], double* row, int lane) {
#pragma unroll
  for (int i = 0; i < 7; i++) {
    const double a = swap_add32(even[2 * i], even[2 * i + 1]);  // halves: slot 4i | slot 4i + 2
    const double c = swap_add32(odd[2 * i], odd[2 * i + 1]);    // halves: slot 4i + 1 | slot 4i + 3
    const double u = row_sum16(swap_add16(a, c));               // rows: 4i, 4i + 1, 4i + 2, 4i + 3
    if ((lane & 15) == 15) row[4 * i + (lane >> 4)] = u;
  }
}

// Second reduction stage: out[k] = sum over tiles of p[tile*N + k], k < N, for a 256-thread block.
// ROWS x N threads (ROWS = 256 / N: 9 x 28 for GICP, 5 x 43 for NDT) read ROWS*N consecutive doubles per step (fully
// coalesced), each thread owns one (row, column) and walks tiles row, row+ROWS, ... ; the rows are then added in a
// fixed order -> bitwise reproducible.
constexpr int kSolveBlock = 256;     // GICP / VGICP solve
constexpr int kNdtSolveBlock = 512;  // NDT: 43 columns leave only 5 rows at 256 threads (1024 was measured slower)
// nchunks > 0 (k_gicp_linearize's short-packet launches): p holds one row per 64-point CHUNK and the partial of tile t is (c[4t] + c[4t+1]) + (c[4t+2] + c[4t+3]),
// rows past the last chunk counting as zeros — what last_wave_stores makes of the four wave rows of a 256-point block, bit for bit.
template <int N, int THREADS = kSolveBlock>
__device__ __forceinline__ void reduce_tiles(const double* __restrict__ p, int ntiles, double* out /* LDS [N] */, double* scratch /* LDS [THREADS] */, int nchunks = 0) {
  constexpr int ROWS = THREADS / N;
  const int t = threadIdx.x;
  if (t < ROWS * N) {
    const int col = t % N, row = t / N;
    const auto partial = [&](int tile) -> double {
      if (nchunks == 0) return p[(size_t)tile * N + col];
      const int c = 4 * tile;
      const double r0 = p[(size_t)c * N + col];
      const double r1 = c + 1 < nchunks ? p[(size_t)(c + 1) * N + col] : 0.0;
      const double r2 = c + 2 < nchunks ? p[(size_t)(c + 2) * N + col] : 0.0;
      const double r3 = c + 3 < nchunks ? p[(size_t)(c + 3) * N + col] : 0.0;
      return (r0 + r1) + (r2 + r3);
    };
    // four independent partial sums keep four loads in flight per thread (one dependent chain was latency bound)
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int tile = row;
#if HGS_REDUCE_UNROLL == 2  // (a macro as the pragma's operand does not survive a separate preprocessing step: hipcc --save-temps)
#pragma unroll 2
#endif
    for (; tile + 3 * ROWS < ntiles; tile += 4 * ROWS) {
      const double a = partial(tile), b = partial(tile + ROWS);
      const double c = partial(tile + 2 * ROWS), d = partial(tile + 3 * ROWS);
      s0 += a, s1 += b, s2 += c, s3 += d;
    }
    for (; tile < ntiles; tile += ROWS) s0 += partial(tile);
    scratch[row * N + col] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (t < N) {
    double s = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++) s += scratch[r * N + t];
    out[t] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ BvhView view_of(const TargetView& t) {
  BvhView v;
  v.nodes = t.nodes, v.pts = t.pts, v.lpts = t.lpts, v.P = t.P, v.n = t.meta->nvalid;
  return v;
}

// ------------------------------------------------------------------------------------------------ upload
// (round 5) the packing kernel of a new cloud also resets the cloud's meta record — nvalid and the bounding box k_bbox_count accumulates into with atomics
// right behind it in the stream: one dispatch less per uploaded sweep than the separate k_meta_init
__device__ __forceinline__ void meta_reset(CloudMeta* m) {
  m->nvalid = 0;
  for (int d = 0; d < 3; d++) m->bbmin[d] = 0xffffffffu, m->bbmax[d] = 0u;
}
// `staged`: the points as the host packed them on their way up (hgs_engine.hip, upload_points_packed): 16-byte records {x, y, z, intensity}
// (round 6) ... and writes the cloud's RESIDENT descriptor (`desc_out`, inside the cloud's own block): the single-cloud launches of a registration read
// it from there instead of from a descriptor array that a copy kernel has to fill in front of every stage (6 of the ~14 copies of an odometry sweep)
__global__ __launch_bounds__(kBlock) void k_pack_aos(const float4* __restrict__ staged, int n, float4* __restrict__ raw, float* __restrict__ intensity, CloudMeta* meta,
                                                     CloudDesc desc, CloudDesc* desc_out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (meta && i == 0) meta_reset(meta);
  if (desc_out && i == 0) *desc_out = desc;
  if (i >= n) return;
  const float4 p = staged[i];
  raw[i] = make_float4(p.x, p.y, p.z, __int_as_float(i));
  if (intensity) intensity[i] = p.w;
}
void launch_pack_aos(hipStream_t s, const float4* staged, int n, float4* raw, float* intensity, CloudMeta* meta, const CloudDesc* desc, CloudDesc* desc_out) {
  if (n <= 0 && !meta) return;
  hipLaunchKernelGGL(k_pack_aos, dim3(std::max(1, (n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, staged, n, raw, intensity, meta, desc ? *desc : CloudDesc{},
                     desc ? desc_out : nullptr);
}

// ------------------------------------------------------------------------------------------------ search index
__global__ void k_meta_init(const CloudDesc* descs, int ncloud) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncloud) return;
  meta_reset(descs[c].meta);
}
void launch_meta_init(hipStream_t s, const CloudDesc* descs, int ncloud) {
  hipLaunchKernelGGL(k_meta_init, dim3((ncloud + 63) / 64), dim3(64), 0, s, descs, ncloud);
}

constexpr int kBboxPerThread = 8;
__global__ __launch_bounds__(kBlock) void k_bbox_count(const CloudDesc* descs) {
  const CloudDesc d = descs[blockIdx.y];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int cnt = 0;
  // a thread's kBboxPerThread points are loaded together (round 4 walked them in a dependent loop: 22 us for one 119 k-point sweep — eight L2 round trips)
  for (int i0 = blockIdx.x * kBlock * kBboxPerThread + threadIdx.x; i0 < d.n_input; i0 += gridDim.x * kBlock * kBboxPerThread) {
    float4 p[kBboxPerThread];
#pragma unroll
    for (int k = 0; k < kBboxPerThread; k++) {
      const int i = i0 + k * kBlock;
      p[k] = i < d.n_input ? d.raw[i] : make_float4(NAN, NAN, NAN, 0.f);
    }
#pragma unroll
    for (int k = 0; k < kBboxPerThread; k++)
      if (finite3(p[k])) {
        cnt++;
        mn[0] = fminf(mn[0], p[k].x), mn[1] = fminf(mn[1], p[k].y), mn[2] = fminf(mn[2], p[k].z);
        mx[0] = fmaxf(mx[0], p[k].x), mx[1] = fmaxf(mx[1], p[k].y), mx[2] = fmaxf(mx[2], p[k].z);
      }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    cnt += __shfl_down(cnt, off, 64);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      mn[k] = fminf(mn[k], __shfl_down(mn[k], off, 64));
      mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off, 64));
    }
  }
  // the four waves of the block meet in LDS and ONE thread sends the block's seven atomics: every block of every cloud's launch hits the same seven
  // words, and it is their serialisation at the L2, not the loads, that this kernel's time is made of (wave-level atomics: 39 us for a 119 k-point
  // sweep with 4 points per thread, 22 us with 8; per block and 8 points per thread: profiles/r05_upload.md)
  __shared__ float s_mn[kBlock / 64][3], s_mx[kBlock / 64][3];
  __shared__ int s_cnt[kBlock / 64];
  const int wave = (int)(threadIdx.x >> 6);
  if ((threadIdx.x & 63) == 0) {
    s_cnt[wave] = cnt;
    for (int k = 0; k < 3; k++) s_mn[wave][k] = mn[k], s_mx[wave][k] = mx[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; w++) {
      cnt += s_cnt[w];
      for (int k = 0; k < 3; k++) mn[k] = fminf(mn[k], s_mn[w][k]), mx[k] = fmaxf(mx[k], s_mx[w][k]);
    }
    if (cnt > 0) {
      atomicAdd(&d.meta->nvalid, cnt);
      for (int k = 0; k < 3; k++) {
        atomicMin(&d.meta->bbmin[k], f2ord(mn[k]));
        atomicMax(&d.meta->bbmax[k], f2ord(mx[k]));
      }
    }
  }
}
void launch_bbox_count(hipStream_t s, const CloudDesc* descs, int ncloud, int max_n) {
  int gx = (max_n + kBlock * kBboxPerThread - 1) / (kBlock * kBboxPerThread);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_bbox_count, dim3(gx, ncloud), dim3(kBlock), 0, s, descs);
}

// drop_bits: low bits of the 48-bit curve position the sort will not look at (3 per level dropped, hgs_engine.hip): a finite point's position is kept
// below the all-ones value ALSO in the bits that remain, so that no finite point ties with the non-finite ones, which must end up behind nvalid
__global__ __launch_bounds__(kBlock) void k_hilbert_keys(const CloudDesc* descs, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals, int drop_bits) {
  const CloudDesc d = descs[blockIdx.y];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= d.n_input) return;
  const float4 p = d.raw[i];
  unsigned long long code = 0xffffffffffffull;
  if (finite3(p)) {
    const float mnx = ord2f(d.meta->bbmin[0]), mny = ord2f(d.meta->bbmin[1]), mnz = ord2f(d.meta->bbmin[2]);
    const float ex = ord2f(d.meta->bbmax[0]) - mnx, ey = ord2f(d.meta->bbmax[1]) - mny, ez = ord2f(d.meta->bbmax[2]) - mnz;
    const float ext = fmaxf(ex, fmaxf(ey, ez));
    const float sc = ext > 0.f ? 65535.f / ext : 0.f;
    const unsigned qx = min(65535u, (unsigned)((p.x - mnx) * sc));
    const unsigned qy = min(65535u, (unsigned)((p.y - mny) * sc));
    const unsigned qz = min(65535u, (unsigned)((p.z - mnz) * sc));
    code = hilbert48(qx, qy, qz);
    const unsigned long long top = 0xffffffffffffull >> drop_bits << drop_bits;
    if (code >= top) code = top - 1ull;  // reserve all-ones (in the compared bits) for non-finite points
  }
  keys[d.sort_off + i] = ((unsigned long long)blockIdx.y << 48) | code;
  vals[d.sort_off + i] = (unsigned)i;
}
void launch_hilbert_keys(hipStream_t s, const CloudDesc* descs, int ncloud, int max_n, unsigned long long* keys, unsigned* vals, int drop_bits) {
  if (max_n <= 0) return;
  hipLaunchKernelGGL(k_hilbert_keys, dim3((max_n + kBlock - 1) / kBlock, ncloud), dim3(kBlock), 0, s, descs, keys, vals, drop_bits);
}

__global__ __launch_bounds__(kBlock) void k_gather_sorted(const CloudDesc* descs, const unsigned* __restrict__ sorted_vals) {
  const CloudDesc d = descs[blockIdx.y];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= d.P * kLeaf) return;
  float4 p = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(-1));
  if (i < d.meta->nvalid) p = d.raw[sorted_vals[d.sort_off + i]];
  d.pts[i] = p;
  float* leaf = reinterpret_cast<float*>(d.lpts) + 32 * (size_t)(i >> 3) + (i & 7);  // SoA copy for the wave walk
  leaf[0] = p.x, leaf[8] = p.y, leaf[16] = p.z, leaf[24] = p.w;
  if (d.pad) d.corr[i] = -1;  // hgs_cloud_invalidate: the seeds of earlier registrations are forgotten with the index
}
void launch_gather_sorted(hipStream_t s, const CloudDesc* descs, int ncloud, int max_slots, const unsigned* sorted_vals) {
  if (max_slots <= 0) return;
  hipLaunchKernelGGL(k_gather_sorted, dim3((max_slots + kBlock - 1) / kBlock, ncloud), dim3(kBlock), 0, s, descs, sorted_vals);
}

// Slots 0 and 1 of group 0 (node 0 does not exist, node 1 is the root, whose box no search ever tests) hold EMPTY
// boxes: an odd-height tree starts its 4-ary walk at the virtual node 0, whose group is then {empty, empty, 2, 3}.
__device__ __forceinline__ void store_walk_sentinels(float4* nodes) {
  const float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  bvh_store_box(nodes, 0u, mn, mx);
  bvh_store_box(nodes, 1u, mn, mx);
}

// Leaves + the 8 tree levels above them: one block owns 256 consecutive leaves, merges in LDS.  Node 0 and the
// pad floats of the grouped layout are never read as boxes.
__global__ __launch_bounds__(kBlock) void k_build_bottom(const CloudDesc* descs) {
  const CloudDesc d = descs[blockIdx.y];
  const int P = d.P;
  const int leaf0 = blockIdx.x * kBlock;
  if (leaf0 >= P) return;
  __shared__ float smn[2][kBlock][3], smx[2][kBlock][3];
  const int t = threadIdx.x;
  const int nvalid = d.meta->nvalid;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  const int leaf = leaf0 + t;
  if (leaf < P) {
#pragma unroll
    for (int l = 0; l < kLeaf; l++) {
      const int idx = leaf * kLeaf + l;
      if (idx < nvalid) {
        const float4 p = d.pts[idx];
        mn[0] = fminf(mn[0], p.x), mn[1] = fminf(mn[1], p.y), mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x), mx[1] = fmaxf(mx[1], p.y), mx[2] = fmaxf(mx[2], p.z);
      }
    }
    bvh_store_box(d.nodes, (unsigned)(P + leaf), mn, mx);
  }
  for (int k = 0; k < 3; k++) smn[0][t][k] = mn[k], smx[0][t][k] = mx[k];
  int cur = 0, width = P < kBlock ? P : kBlock, level_nodes = P, first = leaf0;
  while (width > 1) {
    __syncthreads();
    const int nw = width >> 1;
    level_nodes >>= 1;
    first >>= 1;
    if (t < nw) {
      float a[3], b[3];
      for (int k = 0; k < 3; k++) {
        a[k] = fminf(smn[cur][2 * t][k], smn[cur][2 * t + 1][k]);
        b[k] = fmaxf(smx[cur][2 * t][k], smx[cur][2 * t + 1][k]);
        smn[cur ^ 1][t][k] = a[k];
        smx[cur ^ 1][t][k] = b[k];
      }
      bvh_store_box(d.nodes, (unsigned)(level_nodes + first + t), a, b);
    }
    cur ^= 1;
    width = nw;
  }
  if (t == 0 && P <= kBlock) store_walk_sentinels(d.nodes);
}
// Remaining top levels (only when P > 256): one block per cloud walks them level by level.
__global__ __launch_bounds__(kBlock) void k_build_top(const CloudDesc* descs) {
  const CloudDesc d = descs[blockIdx.x];
  if (d.P <= kBlock) return;
  for (int level_nodes = (d.P / kBlock) >> 1; level_nodes >= 1; level_nodes >>= 1) {
    for (int t = threadIdx.x; t < level_nodes; t += kBlock) {
      const unsigned id = (unsigned)(level_nodes + t);
      float amn[3], amx[3], bmn[3], bmx[3];
      bvh_load_box(d.nodes, 2 * id, amn, amx);
      bvh_load_box(d.nodes, 2 * id + 1, bmn, bmx);
      const float mn[3] = {fminf(amn[0], bmn[0]), fminf(amn[1], bmn[1]), fminf(amn[2], bmn[2])};
      const float mx[3] = {fmaxf(amx[0], bmx[0]), fmaxf(amx[1], bmx[1]), fmaxf(amx[2], bmx[2])};
      bvh_store_box(d.nodes, id, mn, mx);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) store_walk_sentinels(d.nodes);
}
void launch_build_tree(hipStream_t s, const CloudDesc* descs, int ncloud, int max_P) {
  hipLaunchKernelGGL(k_build_bottom, dim3((max_P + kBlock - 1) / kBlock, ncloud), dim3(kBlock), 0, s, descs);
  if (max_P > kBlock) hipLaunchKernelGGL(k_build_top, dim3(ncloud), dim3(kBlock), 0, s, descs);
}

// ------------------------------------------------------------------------------------------------ GICP covariances
// calculate_covariances of fast_gicp: one thread per (Hilbert-sorted) point, one packet walk per wave, two passes:
//   1. the squared distance r2 of the k-th nearest neighbour — a sorted list of k DISTANCES only (one v_med3_f32 per
//      slot and insertion; keeping the positions sorted alongside costs 4x as many instructions and this kernel is
//      VALU bound, see profiles/),
//   2. a gather bounded by r2 that sums (p - q) and (p - q)(p - q)^T over exactly the k nearest points (those closer than r2
//      plus, of those at exactly r2, as many as the list held, lowest original index first like a kd-tree k-NN orders equal
//      distances) — the covariance needs the set, not its order.  With REPLAY the gather does not walk the tree again: pass 1 logs the
//      leaves it visits, and because it prunes with box_d2 <= bound (never tighter than the final r2) that log plus the wave's own
//      8 leaves holds every leaf with a point within r2 of some lane; their records are fetched by index, the next one in flight
//      while the current one is summed.  The tree is walked a second time only when the log overflows (kKnnLeafLog leaves) or a
//      lane needs several points at exactly its k-th distance.  Without REPLAY (one or two clouds, dense clouds: launch_knn_cov)
//      pass 2 is a second walk bounded by r2.
// Algorithmic bytes per point: 16 (query) + k*16 (neighbours) + 24 (cov).
// qpw = queries per wave (64, or fewer — a multiple of 8 — when the whole launch is too small to fill the chip: shorter
// packets walk fewer nodes, so the dependent-load chain that bounds a small launch gets shorter; lanes >= qpw idle).
// REG_GENERAL = false: FROBENIUS (the mode hdl_graph_slam runs); true: any hgs_regularization (3x3 eigen-decomposition per
// point), a separate instantiation so that the default kernel carries none of it.
// GATHER: how pass 2 finds a lane's neighbours again — 0: a second tree walk bounded by r2; 1: replay of the leaves pass 1 visited (all
// lanes over the union); 2: per-lane leaf lists (every lane over the leaves that gave IT a candidate; own leaves in lock-step).
// (Round 6 measured a mode 3 — the same lists, fp64 sums neighbour by neighbour from per-leaf bit masks instead of leaf by leaf with predicated adds,
// bit-identical sums — at 3.41 vs 3.43 ms per 65-cloud pass: pass 2 is not where this kernel's time is.  Removed; profiles/r06_ab_gather_seed.log.)
// REG: 0 = FROBENIUS inline (the mode hdl_graph_slam runs by SURVEY A.2); 1 = any hgs_regularization inline (3x3 eigen-decomposition per point: the
// compiler then needs > 96 VGPRs for the whole kernel); 2 = the neighbourhood covariance staged in fp64 (`raw`, 48 bytes per point) for
// k_cov_regularize — the search keeps FROBENIUS's register budget and occupancy, the eigen-decompositions run in a streaming kernel of their own.
template <int KMAX, int REG, int GATHER, bool WINDOW = false>
__global__ __launch_bounds__(kBlock) HGS_KNN_OCCUPANCY void k_knn_cov(const CloudDesc* descs, int k, int qpw, int reg_method, double* __restrict__ raw, int raw_stride) {
  constexpr bool REG_GENERAL = REG == 1;
  constexpr bool REPLAY = GATHER == 1, LISTS = GATHER == 2;
  const CloudDesc d = descs[blockIdx.y];
  const int n = d.meta->nvalid;
  const int tile_pts = (kBlock / 64) * qpw;
  const int ntiles = (n + tile_pts - 1) / tile_pts;
  if (xcd_tile(blockIdx.x, ntiles) >= ntiles) return;
  const int lane = (int)(threadIdx.x & 63);
  const int i = xcd_tile(blockIdx.x, ntiles) * tile_pts + (int)(threadIdx.x >> 6) * qpw + lane;
  const bool active = lane < qpw && i < n;
  BvhView tv;
  tv.nodes = d.nodes, tv.pts = d.pts, tv.lpts = d.lpts, tv.P = d.P, tv.n = n;
  const float4 qp = active ? d.pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const F3 q = {qp.x, qp.y, qp.z};
  const int live = k < KMAX ? k : KMAX;
  __shared__ __attribute__((aligned(512))) float walk_slots[kBlock / 64][kParkFloats];
  __shared__ unsigned leaf_log[REPLAY ? kBlock / 64 : 1][REPLAY ? kKnnLeafLog : 1];
  __shared__ unsigned lane_lists[LISTS ? kKnnLaneList : 1][LISTS ? kBlock : 1];  // [entry][thread]: conflict-free for equal counts
  float* slot = walk_slots[threadIdx.x >> 6];  // the quad walk's parking area; its first 128 bytes stage the replayed records
  LeafLog log = {leaf_log[REPLAY ? threadIdx.x >> 6 : 0], REPLAY ? kKnnLeafLog : 0, 0};
  const int i0 = i - lane;
  // the pre-fill window [win0, win0 + win_n): whole leaves that contain the packet's own points; at least 32 points (>= k for every k hdl_graph_slam uses)
  // unless the cloud has fewer, at most 64 (39 when it is not the packet itself)
  // WINDOW = false (packets of 32 or 64 queries): the window is the packet itself, as in rounds 2-5 — the general form below costs the 64-candidate batch
  // 3.7 % of its covariance stage (3.53 -> 3.66 ms) and buys a 32- / 64-query packet nothing: its last packet is no longer the slowest wave, but the packets
  // over sparse far returns are, at 4x the median's leaves (profiles/r06_ab7_knn_window.log).  WINDOW = true: launches with 16- / 8-query packets.
  int win0 = i0, win_n = min(qpw, n - i0), win_leaves = qpw >> 3, order_lane = qpw >> 1;
  F3 wp = q;
  if (WINDOW) {
    win0 = i0 & ~31, win_n = min(max(qpw, 32), n - win0);
    if (win_n < 32 && win0 > 0) win0 = max(0, (n - 32) & ~7), win_n = n - win0;
    if (i0 >= n) win0 = 0, win_n = 0;  // (a wave of idle lanes behind the cloud's end)
    win_leaves = (win_n + kLeaf - 1) / kLeaf;
    order_lane = max(0, min(qpw, n - i0)) >> 1;  // the packet's middle QUERY decides which child is nearest (an idle lane wants none)
    const float4 wpp = lane < win_n ? d.pts[win0 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    wp = F3{wpp.x, wpp.y, wpp.z};
  }
  const bool walk = WINDOW ? n > win_n : n > qpw;  // otherwise the window was the whole cloud
  float r2;
  int ties, list_cnt = 0;
  HGS_PROBE_TIME(0);
  {
    KnnRadiusLane<KMAX, GATHER != 0, LISTS> L;
    L.init(live, active);
    L.list = &lane_lists[0][LISTS ? threadIdx.x : 0], L.stride = kBlock, L.cnt = 0, L.cap = kKnnLaneList;
    // The wave's window — its own 64 points (8 whole leaves of the Hilbert order) — holds every lane's first candidates: all-pairs
    // through v_readlane — the walk then starts with every list full and a bound within ~1.2x of
    // the final radius instead of +inf, which is what keeps it from wandering (3x fewer insertions and leaves).
    // (WINDOW, above: a packet of fewer than 32 points — every one of a launch with 16- / 8-query packets, the last one of a cloud — takes the window of >= k
    // points around it.  A launch of less than one packet per SIMD lasts as long as its slowest wave, and those are the packets over sparse far returns at
    // ~4x the median's leaves (scripts/probes/knn_probe.py): shorter packets split them, but only behind a full window — with 16 own points the lists start
    // unfilled and the walk wanders, which is why round 6's first attempt at 16-query packets measured equal.)
    for (int jj = 0; jj < win_n; jj++) {
      const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wp.x), jj)), py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wp.y), jj)),
                  pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wp.z), jj));
      const float dd = dist2f(q, px, py, pz);
      if (__ballot(dd < L.worst()) != 0ull) L.insert_wave(dd < L.worst() ? dd : FLT_MAX);
    }
    HGS_PROBE_TIME(1);
    if (walk)
      wave_walk_quad(tv, L, q, slot, order_lane, (unsigned)(tv.P + (win0 >> 3)), (unsigned)win_leaves, REPLAY ? &log : nullptr);
    HGS_PROBE_TIME(2);
    r2 = L.worst();
    list_cnt = L.cnt;
    int n_lt = 0;
#pragma unroll
    for (int j = 0; j < KMAX; j++) n_lt += (L.d[j] >= 0.f && L.d[j] < r2) ? 1 : 0;
    ties = live - n_lt;
  }
  // pass 2 (TIES = 1 unless some lane of this wave needs several points at exactly its k-th distance)
  double s1[3];
  Sym3 s2;
  int found;
  if (__ballot(active && ties > 1) == 0ull) {
    KnnGatherLane<1> L;
    L.init(active ? r2 : -1.f, ties, q.x, q.y, q.z);
    if (LISTS && walk && __ballot(active && list_cnt > kKnnLaneList) == 0ull) {
      // Every lane over ITS leaves, entry r of all lanes at once, each lane gathering its own 128-byte leaf record: first the wave's own
      // leaves whose box lies within the lane's radius (their eight boxes are two adjacent group records: one fetch, two box
      // evaluations — 3.9 of the 8 on average), then the leaves on its list.  ~11.5 rounds per wave instead of 8 lock-step visits + ~9
      // list rounds, and every lane's arithmetic is on a leaf that matters to it.
      const hgs_f2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
      const unsigned leaf0 = (unsigned)(win0 >> 3);  // ("own" leaves = the window's: the walk skipped exactly those; <= 5 of them from any leaf on, or 8 from a multiple of 8)
      const int n_own = max(0, min(win_leaves, tv.P - (int)leaf0));
      unsigned own_mask = 0;
      if (tv.P >= 8) {  // the own leaves are two aligned sibling groups of four
        const float v = reinterpret_cast<const float*>(tv.nodes + 8 * (size_t)(((unsigned)tv.P + leaf0) >> 2))[lane];
        __builtin_amdgcn_wave_barrier();
        slot[lane] = v;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 2; g++) {
          const hgs_f16v lo = *reinterpret_cast<const hgs_f16v*>(slot + 32 * g);
          const hgs_f8v hi = *reinterpret_cast<const hgs_f8v*>(slot + 32 * g + 16);
          const hgs_f2 d01 = pk_box_dist2(qx, qy, qz, hgs_f2{lo[0], lo[1]}, hgs_f2{lo[4], lo[5]}, hgs_f2{lo[8], lo[9]}, hgs_f2{lo[12], lo[13]}, hgs_f2{hi[0], hi[1]},
                                          hgs_f2{hi[4], hi[5]});
          const hgs_f2 d23 = pk_box_dist2(qx, qy, qz, hgs_f2{lo[2], lo[3]}, hgs_f2{lo[6], lo[7]}, hgs_f2{lo[10], lo[11]}, hgs_f2{lo[14], lo[15]}, hgs_f2{hi[2], hi[3]},
                                          hgs_f2{hi[6], hi[7]});
          own_mask |= ((d01.x <= r2 ? 1u : 0u) | (d01.y <= r2 ? 2u : 0u) | (d23.x <= r2 ? 4u : 0u) | (d23.y <= r2 ? 8u : 0u)) << (4 * g);
        }
        own_mask = (own_mask >> (leaf0 & 3u)) & ((1u << n_own) - 1u);  // the own leaves that exist (32-query packets start mid-pair of groups; the spare groups behind a tiny tree hold no boxes)
      } else {
        own_mask = (1u << n_own) - 1u;
      }
      if (!active) own_mask = 0;
      int li = 0;
      while (__ballot(active && (own_mask != 0u || li < list_cnt)) != 0ull) {
        if (active && (own_mask != 0u || li < list_cnt)) {
          unsigned leaf;
          if (own_mask) {
            leaf = leaf0 + (unsigned)__builtin_ctz(own_mask);
            own_mask &= own_mask - 1u;
          } else {
            leaf = lane_lists[li++][threadIdx.x];
          }
          const hgs_f16v* rec = reinterpret_cast<const hgs_f16v*>(tv.lpts + 8 * (size_t)leaf);
          const hgs_f16v lo = rec[0], hi = rec[1];
          L.visit_leaf(lo, hi, qx, qy, qz, (int)leaf * kLeaf);
        }
      }
    } else if (REPLAY && walk && log.count <= log.cap) {
      // No second tree walk: every leaf with a point within r2 of some lane (box_d2 <= r2 <= the bound pass 1 had when it met the
      // leaf) is either one of the wave's own 8 leaves or in pass 1's log.  The records are fetched by index, so the next one is
      // in flight while the current one is summed — no dependent chain, no box tests.
      // (the wave's own leaves that exist: a tail wave of a tiny cloud would otherwise read leaf records past the array)
      const int n_own = max(0, min(win_leaves, tv.P - (win0 >> 3))), total = n_own + log.count;
      const hgs_f2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
      const int l32 = lane & 31;
      unsigned leaf = n_own > 0 ? (unsigned)(win0 >> 3) : (total > 0 ? log.ids[0] - (unsigned)tv.P : 0u);  // leaf index (node id - P)
      float v = reinterpret_cast<const float*>(tv.lpts + 8 * (size_t)leaf)[l32];
      for (int j = 0; j < total; j++) {
        __builtin_amdgcn_wave_barrier();  // the previous record's reads are issued before the slot is overwritten
        slot[l32] = v;
        __builtin_amdgcn_wave_barrier();
        const hgs_f16v* r = reinterpret_cast<const hgs_f16v*>(slot);
        const hgs_f16v lo = r[0], hi = r[1];
        const unsigned this_leaf = leaf;
        if (j + 1 < total) {
          leaf = j + 1 < n_own ? (unsigned)(win0 >> 3) + (unsigned)(j + 1) : log.ids[j + 1 - n_own] - (unsigned)tv.P;
          v = reinterpret_cast<const float*>(tv.lpts + 8 * (size_t)leaf)[l32];
        }
        L.visit_leaf(lo, hi, qx, qy, qz, (int)this_leaf * kLeaf);
      }
    } else {
      wave_walk_quad(tv, L, q, slot, order_lane);
    }
    L.finish(d.pts);
    s1[0] = L.s1[0], s1[1] = L.s1[1], s1[2] = L.s1[2], found = L.found;
    s2 = Sym3{L.s2[0], L.s2[1], L.s2[2], L.s2[3], L.s2[4], L.s2[5]};
  } else {
    KnnGatherLane<4> L;
    L.init(active ? r2 : -1.f, ties, q.x, q.y, q.z);
    wave_walk_quad(tv, L, q, slot, order_lane);
    L.finish(d.pts);
    for (int taken = 4; __ballot(active && ties > taken) != 0ull; taken += 4) {  // more than 4 at the k-th distance
      const bool more = active && ties > taken;
      L.rearm_ties(more ? r2 : -1.f, more ? ties - taken : 0);
      wave_walk_quad(tv, L, q, slot, order_lane);
      L.finish(d.pts);
    }
    s1[0] = L.s1[0], s1[1] = L.s1[1], s1[2] = L.s1[2], found = L.found;
    s2 = Sym3{L.s2[0], L.s2[1], L.s2[2], L.s2[3], L.s2[4], L.s2[5]};
  }
  HGS_PROBE_TIME(3);
  if (!active) return;
  if (REG == 2) {  // staged: k_cov_regularize finishes the point
    const Sym3 c = gicp_neighbour_cov(s1, s2, found, k, 0.0);
    double* o = raw + ((size_t)blockIdx.y * (size_t)raw_stride + (size_t)i) * 6;
    o[0] = c.xx, o[1] = c.xy, o[2] = c.xz, o[3] = c.yy, o[4] = c.yz, o[5] = c.zz;
    return;
  }
  const Sym3 c = REG_GENERAL ? gicp_regularized_cov(s1, s2, found, k, reg_method) : gicp_regularized_cov(s1, s2, found, k);
  d.cov[2 * i] = make_float4((float)c.xx, (float)c.xy, (float)c.xz, (float)c.yy);
  d.cov[2 * i + 1] = make_float4((float)c.yz, (float)c.zz, 0.f, 0.f);
}
// The regularisation of the staged covariances (REG == 2): 48 bytes in, 24 out per point, one point per thread, nothing else live — the 3x3 Jacobi
// eigen-decomposition that cost k_knn_cov<.., 1, ..> its 5-waves-per-SIMD register budget for the WHOLE search runs here at full occupancy.
__global__ __launch_bounds__(kBlock) void k_cov_regularize(const CloudDesc* descs, const double* __restrict__ raw, int raw_stride, int reg_method) {
  const CloudDesc d = descs[blockIdx.y];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= d.meta->nvalid) return;
  const double* r = raw + ((size_t)blockIdx.y * (size_t)raw_stride + (size_t)i) * 6;
  const Sym3 c = gicp_regularize_cov(Sym3{r[0], r[1], r[2], r[3], r[4], r[5]}, reg_method);
  d.cov[2 * i] = make_float4((float)c.xx, (float)c.xy, (float)c.xz, (float)c.yy);
  d.cov[2 * i + 1] = make_float4((float)c.yz, (float)c.zz, 0.f, 0.f);
}
#ifdef HGS_KNN_PROBE
extern "C" int hgs_debug_read_knn_probe(void* out, size_t bytes) {  // (measurement builds only) copies the probe out and clears it
  if (bytes > sizeof(unsigned long long) * (1 << 16) * 8) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_probe), bytes) != hipSuccess) return -3;
  void* dev = nullptr;
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_knn_probe)) != hipSuccess || hipMemset(dev, 0, sizeof(unsigned long long) * (1 << 16) * 8) != hipSuccess) return -4;
  return 0;
}
#endif
template <int REG, int REPLAY>
static void launch_knn_cov_t(hipStream_t s, const CloudDesc* descs, int ncloud, int max_n, int k, int qpw, int reg_method, double* raw, int raw_stride) {
  const int tile_pts = (kBlock / 64) * qpw;
  const dim3 grid(HGS_GRID_X((max_n + tile_pts - 1) / tile_pts), ncloud), block(kBlock);
  if (qpw < 32) {  // short packets (hgs_engine.hip asks for them with the per-lane lists only, k <= 20: the window is sized for that)
    if (REPLAY == 2 && k <= 20) hipLaunchKernelGGL((k_knn_cov<20, REG, 2, true>), grid, block, 0, s, descs, k, qpw, reg_method, raw, raw_stride);
    return;
  }
  if (k <= 8) hipLaunchKernelGGL((k_knn_cov<8, REG, REPLAY>), grid, block, 0, s, descs, k, qpw, reg_method, raw, raw_stride);
  else if (k <= 16) hipLaunchKernelGGL((k_knn_cov<16, REG, REPLAY>), grid, block, 0, s, descs, k, qpw, reg_method, raw, raw_stride);
  else if (k <= 20) hipLaunchKernelGGL((k_knn_cov<20, REG, REPLAY>), grid, block, 0, s, descs, k, qpw, reg_method, raw, raw_stride);
  else if (k <= 32) hipLaunchKernelGGL((k_knn_cov<32, REG, REPLAY>), grid, block, 0, s, descs, k, qpw, reg_method, raw, raw_stride);
  else hipLaunchKernelGGL((k_knn_cov<64, REG, REPLAY>), grid, block, 0, s, descs, k < 64 ? k : 64, qpw, reg_method, raw, raw_stride);
}
template <int REG>
static void launch_knn_cov_g(hipStream_t s, const CloudDesc* descs, int ncloud, int max_n, int k, int qpw, int reg_method, int gather, double* raw, int raw_stride) {
  if (gather == 2) launch_knn_cov_t<REG, 2>(s, descs, ncloud, max_n, k, qpw, reg_method, raw, raw_stride);
  else if (gather == 1) launch_knn_cov_t<REG, 1>(s, descs, ncloud, max_n, k, qpw, reg_method, raw, raw_stride);
  else launch_knn_cov_t<REG, 0>(s, descs, ncloud, max_n, k, qpw, reg_method, raw, raw_stride);
}
// replay: the gather pass replays pass 1's leaf log instead of walking the tree again.  Measured (same-box A/B): 64 LiDAR clouds in one
// launch 6.28 -> 6.14 ms; a single HDL-32E pair 0.33 -> 0.41 ms and a dense 1 M-point pair 1.91 -> 2.15 ms (a dense cloud's first walk
// visits far more leaves than the gather needs) — hence the instantiations and a choice per launch (hgs_engine.hip).
// raw_stage (>= ncloud * max_n * 6 doubles, or null): regularisations other than FROBENIUS run as search (REG 2) + k_cov_regularize; without it inline (REG 1).
void launch_knn_cov(hipStream_t s, const CloudDesc* descs, int ncloud, int max_n, int k, int qpw, int reg_method, int gather, double* raw_stage) {
  if (max_n <= 0) return;
  if (reg_method == 0) {
    launch_knn_cov_g<0>(s, descs, ncloud, max_n, k, qpw, reg_method, gather, nullptr, 0);
  } else if (raw_stage) {
    launch_knn_cov_g<2>(s, descs, ncloud, max_n, k, qpw, reg_method, gather, raw_stage, max_n);
    hipLaunchKernelGGL(k_cov_regularize, dim3((max_n + kBlock - 1) / kBlock, ncloud), dim3(kBlock), 0, s, descs, raw_stage, max_n, reg_method);
  } else {
    launch_knn_cov_g<1>(s, descs, ncloud, max_n, k, qpw, reg_method, gather, nullptr, 0);
  }
}

// ------------------------------------------------------------------------------------------------ GICP iteration
__global__ void k_gicp_init(GicpState* states, const float* guesses, int B, Progress prog) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0) prog.dev[0] = 0, prog.dev[1] = 0;
  if (b >= B) return;
  gicp_state_init(states[b], guesses + 16 * b);
}
void launch_gicp_init(hipStream_t s, GicpState* states, const float* guesses, int B, Progress prog) {
  hipLaunchKernelGGL(k_gicp_init, dim3((B + 63) / 64), dim3(64), 0, s, states, guesses, B, prog);
}
// a single registration: the guess travels in the kernel's arguments (no upload in front of the launch)
__global__ void k_gicp_init1(GicpState* states, Guess16 guess, Progress prog) {
  if (threadIdx.x != 0) return;
  prog.dev[0] = 0, prog.dev[1] = 0;
  gicp_state_init(states[0], guess.m);
}
void launch_gicp_init1(hipStream_t s, GicpState* states, const float* guess_host, Progress prog) {
  Guess16 g;
  for (int i = 0; i < 16; i++) g.m[i] = guess_host[i];
  hipLaunchKernelGGL(k_gicp_init1, dim3(1), dim3(64), 0, s, states, g, prog);
}

__device__ __forceinline__ Sym3 load_cov(const float4* cov, int i) {
  const float4 a = cov[2 * i], b = cov[2 * i + 1];
  return sym3_from_floats(a.x, a.y, a.z, a.w, b.x, b.y);
}
// Streamed-once data (the source side of a registration: points, covariances, correspondences) is read and written
// with the non-temporal policy so that it does not evict the target's tree / leaves / covariances, which every wave
// of every candidate re-reads, from the L2 (measured neutral on this workload; kept).
__device__ __forceinline__ float4 load_stream(const float4* p) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ Sym3 load_cov_stream(const float4* cov, int i) {
  const float4 a = load_stream(cov + 2 * i), b = load_stream(cov + 2 * i + 1);
  return sym3_from_floats(a.x, a.y, a.z, a.w, b.x, b.y);
}

// The waves of a block leave without waiting for each other (their walks end 30-40 k cycles apart: the closing __syncthreads of
// round 2 cost ~20 % of a wave's lifetime): each wave has put its N sums into its LDS row, lane 0 takes a ticket, and the wave
// that takes the last one adds the four rows in wave order — the same fixed order as before, so the tile partial is bitwise
// what the block-wide reduction produced.  `arrivals` is zeroed by thread 0 before the block's opening barrier.
template <int N>
__device__ __forceinline__ void last_wave_stores(const double* lds /* [4 * N] */, unsigned* arrivals, double* out, int lane) {
  unsigned ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  ticket = (unsigned)__builtin_amdgcn_readlane((int)ticket, 0);
  if (ticket == kBlock / 64 - 1 && lane < N) out[lane] = (lds[lane] + lds[N + lane]) + (lds[2 * N + lane] + lds[3 * N + lane]);
}

// Exact 1-NN of one packet of queries for the kernels below: the quad walk (hgs_wave_bvh.h), nearest-first only when some lane
// has no seed; position and original index resolved from the lane's best leaf, or — when some lane reached its minimum in two
// leaves — by the exact keyed walk.  pos < 0: no target point within bound2.
// ---- seed grid (round 6) -------------------------------------------------------------------------------------------------------
// What makes a 1-NN packet cheap is a per-lane starting bound close to the answer: a packet seeded with the previous correspondences walks ~2.2x
// fewer steps than an unseeded one (k_fitness 0.68 vs 1.48 ms on the 64 x 119 k batch).  getFitnessScore queries that have no correspondences —
// behind NDT / VGICP, calc_fitness_score — get a bound from a grid over the TARGET: three direct-mapped tables (cells of 0.25 / 1 / 4 m) whose
// entry is the sorted position of SOME target point that hashed there (atomicMin: deterministic).  A lookup reads the three entries of the query's
// cells and keeps the nearest of the (up to) three points.  Nothing about it needs to be exact — any target point is a valid upper bound of the
// nearest-neighbour distance, a hash collision only loosens it — and the search result does not depend on the seed (the walk is exact for every
// bound).  2.6 MB per 119 k-point target, built once per target index (~10 us), shared by all candidates of a batch.  Measured (round 6,
// profiles/r06_ab_seed_grid.log): NDT batch k_fitness 1.52 -> 1.13 ms.  NOT used by k_gicp_linearize: the first linearisation of a registration is
// slow because its true nearest-neighbour distances are large (the guess is off by decimetres), not because it lacks a bound — with grid seeds
// the stage measured 6.40 instead of 6.33 ms.  Refining the grid's point inside its own leaf (the nearest of the eight points that share it: a tighter bound for
// eight distances) does not pay either: NDT batch k_fitness 1.12 -> 1.14 ms (profiles/r06_ab17_seed_refine.log).
#ifndef HGS_SEED_INV0
#define HGS_SEED_INV0 4.0f  // finest cell: 0.25 m
#endif
constexpr float kSeedInv0 = HGS_SEED_INV0;
__device__ __forceinline__ unsigned seed_hash(int cx, int cy, int cz) {
  unsigned h = (unsigned)cx * 73856093u ^ (unsigned)cy * 19349663u ^ (unsigned)cz * 83492791u;
  h ^= h >> 15;
  h *= 0x2c1b3c6du;
  h ^= h >> 12;
  return h;
}
__device__ __forceinline__ void seed_cells(float x, float y, float z, int& cx, int& cy, int& cz) {
  cx = (int)floorf(fminf(fmaxf(x * kSeedInv0, -1.0e9f), 1.0e9f));  // NaN -> -1e9 (fmaxf drops it): any cell will do
  cy = (int)floorf(fminf(fmaxf(y * kSeedInv0, -1.0e9f), 1.0e9f));
  cz = (int)floorf(fminf(fmaxf(z * kSeedInv0, -1.0e9f), 1.0e9f));
}
__device__ __forceinline__ void seed_slots(int cx, int cy, int cz, int bits, unsigned& s0, unsigned& s1, unsigned& s2) {
  const unsigned n0 = 1u << bits, n1 = n0 >> 2, n2 = n0 >> 4;
  s0 = seed_hash(cx, cy, cz) & (n0 - 1u);
  s1 = n0 + (seed_hash(cx >> 2, cy >> 2, cz >> 2) & (n1 - 1u));  // (arithmetic shifts: floor division, the same cell for negative coordinates)
  s2 = n0 + n1 + (seed_hash(cx >> 4, cy >> 4, cz >> 4) & (n2 - 1u));
}
__global__ __launch_bounds__(kBlock) void k_seed_grid_build(const float4* __restrict__ pts, const CloudMeta* meta, unsigned* __restrict__ tab, int bits) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= meta->nvalid) return;
  const float4 p = pts[i];
  int cx, cy, cz;
  seed_cells(p.x, p.y, p.z, cx, cy, cz);
  unsigned s0, s1, s2;
  seed_slots(cx, cy, cz, bits, s0, s1, s2);
  atomicMin(tab + s0, (unsigned)i);
  atomicMin(tab + s1, (unsigned)i);
  atomicMin(tab + s2, (unsigned)i);
}
void launch_seed_grid_build(hipStream_t s, const float4* pts, const CloudMeta* meta, int n_max, unsigned* tab, int bits) {
  if (n_max <= 0) return;
  hipLaunchKernelGGL(k_seed_grid_build, dim3((n_max + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, meta, tab, bits);
}
// the nearest of the (up to three) target points the grid holds for q's cells, or -1
__device__ __forceinline__ int seed_grid_lookup(const TargetView& t, const F3& q) {
  int cx, cy, cz;
  seed_cells(q.x, q.y, q.z, cx, cy, cz);
  unsigned s0, s1, s2;
  seed_slots(cx, cy, cz, t.seed_bits, s0, s1, s2);
  const unsigned e0 = t.seed_tab[s0], e1 = t.seed_tab[s1], e2 = t.seed_tab[s2];
  const unsigned n = (unsigned)t.meta->nvalid;
  int best = -1;
  float best_d = FLT_MAX;
  const unsigned e[3] = {e0, e1, e2};
#pragma unroll
  for (int l = 0; l < 3; l++) {
    if (e[l] < n) {
      const float4 p = t.pts[e[l]];
      const float d = dist2f(q, p.x, p.y, p.z);
      if (d < best_d) best_d = d, best = (int)e[l];
    }
  }
  return best;
}

__device__ __forceinline__ void packet_nn1(const BvhView& tv, float* park, const F3& q, bool active, float bound2, int seed, int qpw, float& d2, int& pos, int& orig) {
  const bool seeded = seed >= 0 && seed < tv.n;
  int leaf;
  bool tie, found;
  if (__ballot(active && !seeded) != 0ull) wave_nn1_quad<true, true>(tv, park, q, active, bound2, seed, qpw >> 1, d2, leaf, tie, found);
  else wave_nn1_quad<false, true>(tv, park, q, active, bound2, seed, qpw >> 1, d2, leaf, tie, found);
  if (__ballot(tie) != 0ull) {
    const F3 qa[1] = {q};
    const bool aa[1] = {active};
    const int sa[1] = {seed};
    float da[1];
    int pa[1], oa[1];
    wave_nn1<1>(tv, park, qa, aa, bound2, sa, da, pa, oa, qpw);
    d2 = da[0], pos = pa[0], orig = oa[0];
  } else {
    nn1_resolve_leaf(tv, q, found ? leaf : -1, d2, pos, orig);
  }
}

// ---- two launches per LM round for single registrations (round 6) ---------------------------------------------------------------------------------
// A single registration (the odometry step, config 2) is a chain of ~4 us kernels: the two per-problem control launches of an LM round (k_gicp_solve,
// k_gicp_decide) cost as much as its two point kernels.  Handing the control step to ONE block inside the point kernel ("fused tails": the last block
// of a problem, found by a ticket) was built twice and loses twice: with a fence per block (L2 write-back + invalidate on a chip with one L2 per XCD)
// config 2 went 0.886 -> 1.139 ms (profiles/r06_ab_fused_tails_seed2.log); fence-free (device-scope stores / loads of the partials, relaxed ticket —
// bitwise identical, profiles/r06_ab10_fused_tails_fence_free.{log,patch}) 0.888 -> 0.905 ms: store -> ticket -> coherent loads is a chain of three
// memory round trips, which is what a kernel boundary costs.  What works without ANY hand-over inside a kernel is to REPLICATE the control step: every
// block of k_gicp_linearize<true> first runs the accept / reject step of the previous round on its own copy of the state (the same code on the same
// partials: every block arrives at the same bits), every block of k_gicp_error<true> the solve of the linearisation just made; block 0 of a problem
// writes the state out — into the OTHER of two state buffers, so that a block starting late never reads a state that has already been advanced.
// Launches of <= kFusedRoundMaxProblems problems below the engine's size limit, and batches of small problems up to a few hundred tiles in total
// (run_batch: fused_rounds_max_blocks), only: a 5 us serial step repeated by 250 blocks is latency nobody waits for, repeated by 30 000 it is throughput.
__device__ __forceinline__ DevResult gicp_result_of(const GicpState& st) {
  DevResult r;
  pose_to_colmajor_f(st.x0, r.T);
  r.converged = st.converged, r.iterations = st.iterations, r.lm_tries = st.lm_tries_total, r.pad = 0;
  r.error = st.y0;
  r.fit_sum = 0, r.fit_count = 0, r.pad2 = 0;
  return r;
}
// A finished single registration hands its result to the host the moment it exists: the record goes into host-mapped memory with system-scope stores in
// front of the progress mirror's `done` flag (released), and hgs_align returns from its poll — no result kernel, no device-to-host copy, no stream
// synchronisation, and the no-op rounds the host had queued ahead drain behind its back (engine: run_batch, `early`).
__device__ __forceinline__ void gicp_early_result(DevResult* mapped, const DevResult& r) {
  static_assert(sizeof(DevResult) % 8 == 0, "stored as 64-bit words");
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(&r);
  unsigned long long* o = reinterpret_cast<unsigned long long*>(mapped);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(DevResult) / 8); k++) __hip_atomic_store(o + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void gicp_state_load(GicpState& st, const GicpState* src) {
  static_assert(sizeof(GicpState) % sizeof(double) == 0 && sizeof(GicpState) / sizeof(double) <= kBlock, "state copied one double per thread");
  if (threadIdx.x < sizeof(GicpState) / sizeof(double)) reinterpret_cast<double*>(&st)[threadIdx.x] = reinterpret_cast<const double*>(src)[threadIdx.x];
}
__device__ __forceinline__ void gicp_state_store(GicpState* dst, const GicpState& st) {
  if (threadIdx.x < sizeof(GicpState) / sizeof(double)) reinterpret_cast<double*>(dst)[threadIdx.x] = reinterpret_cast<const double*>(&st)[threadIdx.x];
}

// The 28 sums of one 64-point wave row: the lane's terms, then the transposing swap-adds.
__device__ __forceinline__ void gicp_wave_row(const GicpPointResidual& r, const Sym3& M, double* row, int lane) {
  double even[14], odd[14];
  {
    double t[kAcc];
    gicp_point_terms_by_slot(r, M, t);
#pragma unroll
    for (int j = 0; j < 14; j++) even[j] = t[2 * j];
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    double t[kAcc];
    gicp_point_terms_by_slot(r, M, t);
#pragma unroll
    for (int j = 0; j < 14; j++) odd[j] = t[2 * j + 1];
  }
  wave_sums28_to(even, odd, row, lane);
}

// update_correspondences + linearize fused: per source point 1-NN in the target tree, Mahalanobis matrix,
// 6x6 normal-equation terms; wave shuffle reduction, one LDS row per wave, the last wave of the block adds the rows.
// Algorithmic bytes per source point: 16 (a_i) + 24 (C_A) + 4 (corr) + 16 (b_j) + 24 (C_B) = 84.
// ROUND2 (k_gicp_linearize<true>): states = the buffer the previous kernel wrote, states_out = the other one; partials_err = the trial errors of the
// previous k_gicp_error<true>; the accept / reject step (gicp_decide_wave's arithmetic, in its order) runs first, in every block.
// SHORT (with ROUND2 only): the short-packet form, below.  Its own instantiation and not a branch: with both forms of the per-point arithmetic in one function
// the compiler shared subexpressions between them and contracted the 64-query form's multiplies and adds differently from k_gicp_linearize<false>'s
// (measured: the last bits of the pose moved; profiles/r06_ab13_short_packets.log).
template <bool ROUND2, bool SHORT = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ROUND2 ? 4 : HGS_LINEARIZE_WAVES))) void k_gicp_linearize(const CloudDesc* descs, TargetView tgt, const GicpState* states, GicpConsts c,
                                                           double* __restrict__ partials, int max_blocks, int qpw, GicpState* states_out,
                                                           const double* __restrict__ partials_err, Progress prog, DevResult* results, DevResult* early_out) {
  const int b = blockIdx.y;
  __shared__ GicpState st2;                         // ROUND2: this block's copy of the problem's state
  __shared__ double ws2[kGicpControlWorkspace];
  if constexpr (!ROUND2) {
    if (states[b].phase != GICP_LINEARIZE) return;
  }
  const CloudDesc d = descs[b];
  const int n = d.meta->nvalid;
  if constexpr (ROUND2) {
    const int phase_in = states[b].phase;  // (block-uniform)
    gicp_state_load(st2, states + b);
    double s = 0;
    if (phase_in == GICP_TRY && threadIdx.x < 64) {
      const int ntiles_err = (n + kBlock - 1) / kBlock;
      const double* pe = partials_err + (size_t)b * max_blocks;
      for (int t = threadIdx.x; t < ntiles_err; t += 64) s += pe[t];
      s = wave_sum(s);
    }
    __syncthreads();
    if (phase_in == GICP_TRY && threadIdx.x == 0) gicp_after_error(st2, s, c, ws2);
    __syncthreads();
    if (blockIdx.x == 0) {
      gicp_state_store(states_out + b, st2);
      if (threadIdx.x == 0) {
        const bool finished_now = phase_in == GICP_TRY && st2.phase == GICP_DONE;
        if (finished_now && early_out) {  // (single registrations: the result record in HBM for whoever reads it on the device, and in host-mapped memory)
          const DevResult r = gicp_result_of(st2);
          results[b] = r;
          gicp_early_result(early_out + b, r);
        }
        progress_tick(prog, finished_now, early_out != nullptr);  // (k_gicp_decide's tick: once per problem and round)
      }
    }
    if (st2.phase != GICP_LINEARIZE) return;
  }
  const int tile_pts = (kBlock / 64) * qpw * kNW;
  const int ntiles = (n + tile_pts - 1) / tile_pts;
  const int tile = xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  __shared__ double lds[4 * kAcc];
  __shared__ unsigned arrivals;
  __shared__ __attribute__((aligned(512))) float park[kBlock / 64][kParkFloats];
  if (threadIdx.x == 0) arrivals = 0u;
  __syncthreads();
  // the wave's index lives in an SGPR and the lane id is recomputed after the search: nothing about the thread's identity is
  // kept in (or spilled from) a VGPR across the walk
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const Pose T = ROUND2 ? st2.x0 : states[b].x0;
  float Tf[12];
  pose_to_float(T, Tf);
  static_assert(kNW == 1, "one packet of 64 consecutive source points per wave");
  const int idx = tile * tile_pts + wave * qpw + (int)(threadIdx.x & 63);
  const bool active = (int)(threadIdx.x & 63) < qpw && idx < n;
  const float4 a = active ? load_stream(d.pts + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  const F3 q = transform_point_f(Tf, a.x, a.y, a.z);
  // seed: the correspondence of the previous linearisation (or of an earlier align; -1 / stale values are harmless)
  const int seed = active ? __builtin_nontemporal_load(d.corr + idx) : -1;
  float d2;
  int j, orig;
  packet_nn1(view_of(tgt), park[wave], q, active, c.search_bound2, seed, qpw, d2, j, orig);
  const double R[9] = {T.m[0], T.m[1], T.m[2], T.m[4], T.m[5], T.m[6], T.m[8], T.m[9], T.m[10]};
  int jj = active ? j : -1;
  if (jj >= 0 && !((double)d2 < c.max_corr2)) jj = -1;
  if (active) __builtin_nontemporal_store(jj, d.corr + idx);
  if constexpr (SHORT) {
    static_assert(ROUND2, "the short-packet form belongs to the two-launch rounds");
    // Short packets (qpw 16 / 32: a single small registration, where a launch lasts as long as ONE packet walk and a shorter packet walks fewer nodes) must
    // not change a bit of the result: the waves of the block only search; the LAST one to finish takes the correspondences of the block's 64-point chunks
    // from LDS and runs the linearisation of each chunk with all 64 lanes — the wave row the 64-query kernel computes for the same points, same code — and
    // writes one row per chunk; the solve adds the four chunk rows of a 256-point tile in last_wave_stores' order (reduce_tiles).
    {
      __shared__ int chunk_corr[kBlock];
      int lane;
      HGS_LANE_ID(lane);
      if (lane < qpw) chunk_corr[wave * qpw + lane] = jj;
      unsigned ticket = 0;
      if (lane == 0) ticket = __hip_atomic_fetch_add(&arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      ticket = (unsigned)__builtin_amdgcn_readlane((int)ticket, 0);
      if (ticket != kBlock / 64 - 1) return;
      const int chunks = tile_pts / 64;  // 1 (qpw 16) or 2 (qpw 32)
      for (int ch = 0; ch < chunks; ch++) {
        const int i2 = tile * tile_pts + ch * 64 + lane;
        if (tile * tile_pts + ch * 64 >= n) break;
        const int j2 = i2 < n ? chunk_corr[ch * 64 + lane] : -1;
        Sym3 M2 = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        GicpPointResidual r2 = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (j2 >= 0) {
          const float4 a2 = load_stream(d.pts + i2);
          M2 = gicp_mahalanobis(R, load_cov_stream(d.cov, i2), load_cov(tgt.cov, j2));
          const float4 bp = tgt.pts[j2];
          r2 = gicp_point_residual(T, M2, a2.x, a2.y, a2.z, bp.x, bp.y, bp.z);
        }
        gicp_wave_row(r2, M2, lds, lane);
        __builtin_amdgcn_wave_barrier();  // (the row is written by lanes 15, 31, 47, 63 and read by lanes 0..27 of the same wave: LDS runs a wave's accesses in order)
        if (lane < kAcc) partials[((size_t)b * max_blocks + (size_t)tile * chunks + ch) * kAcc + lane] = lds[lane];
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }
  } else {
  // a lane without a correspondence carries M = 0, T a = 0: every term below is then an exact zero
  Sym3 M = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  GicpPointResidual r = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (jj >= 0) {
    M = gicp_mahalanobis(R, load_cov_stream(d.cov, idx), load_cov(tgt.cov, jj));
    const float4 bp = tgt.pts[jj];
    r = gicp_point_residual(T, M, a.x, a.y, a.z, bp.x, bp.y, bp.z);
  }
  // The 28 sums of the wave by transposing swap-adds (wave_sums28_to), the even output slots first, then the odd ones: the terms are
  // recomputed for the second half (a few fp64 multiplies each, the unused ones are dead code) so that at most 14 are live.
  int lane;
  HGS_LANE_ID(lane);
  double* row = lds + wave * kAcc;
  gicp_wave_row(r, M, row, lane);
  last_wave_stores<kAcc>(lds, &arrivals, partials + ((size_t)b * max_blocks + tile) * kAcc, lane);
  }
}
void launch_gicp_linearize(hipStream_t s, const CloudDesc* descs, TargetView tgt, const GicpState* states, GicpConsts c, double* partials,
                           int max_blocks, int B, int qpw) {
  hipLaunchKernelGGL(k_gicp_linearize<false>, dim3(HGS_GRID_X(max_blocks), B), dim3(kBlock), 0, s, descs, tgt, states, c, partials, max_blocks, qpw, (GicpState*)nullptr,
                     (const double*)nullptr, Progress{}, (DevResult*)nullptr, (DevResult*)nullptr);
}
void launch_gicp_linearize_round2(hipStream_t s, const CloudDesc* descs, TargetView tgt, const GicpState* states_in, GicpState* states_out, GicpConsts c, double* partials,
                                  const double* partials_err, int max_blocks, int lin_blocks, int B, int qpw, Progress prog, DevResult* results, DevResult* early_out) {
  if (qpw < 64)
    hipLaunchKernelGGL((k_gicp_linearize<true, true>), dim3(HGS_GRID_X(lin_blocks), B), dim3(kBlock), 0, s, descs, tgt, states_in, c, partials, max_blocks, qpw, states_out, partials_err, prog, results, early_out);
  else
    hipLaunchKernelGGL((k_gicp_linearize<true, false>), dim3(HGS_GRID_X(lin_blocks), B), dim3(kBlock), 0, s, descs, tgt, states_in, c, partials, max_blocks, qpw, states_out, partials_err, prog, results, early_out);
}

// The LM control step behind a linearisation, run by a whole 256-thread block: fixed-order tile reduction, then ONE lane factorises and steps.  The
// control step is a chain of dependent loads and stores on the problem's state and on the factorisation's pivoted arrays: both live in LDS for its
// duration (state copied in and out by the block; round 3 ran it on HBM + 592 bytes of scratch).  
__device__ __forceinline__ void gicp_solve_block(int ntiles, GicpState* state, const GicpConsts& c, const double* __restrict__ problem_partials, int nchunks = 0) {
  __shared__ double acc[kAcc];
  __shared__ double scratch[kSolveBlock];
  __shared__ GicpState st;
  __shared__ double ws[kGicpControlWorkspace];
  static_assert(sizeof(GicpState) % sizeof(double) == 0 && sizeof(GicpState) / sizeof(double) <= kSolveBlock, "state copied one double per thread");
  if (threadIdx.x < sizeof(GicpState) / sizeof(double)) reinterpret_cast<double*>(&st)[threadIdx.x] = reinterpret_cast<const double*>(state)[threadIdx.x];
  reduce_tiles<kAcc>(problem_partials, ntiles, acc, scratch, nchunks);  // (ends with a barrier: the state copy is complete)
  if (threadIdx.x == 0) gicp_after_linearize(st, acc, c, ws);
  __syncthreads();
  if (threadIdx.x < sizeof(GicpState) / sizeof(double)) reinterpret_cast<double*>(state)[threadIdx.x] = reinterpret_cast<const double*>(&st)[threadIdx.x];
}
__global__ __launch_bounds__(kSolveBlock) __attribute__((amdgpu_waves_per_eu(HGS_CONTROL_WAVES))) void k_gicp_solve(const CloudDesc* descs, GicpState* states, GicpConsts c, const double* __restrict__ partials,
                                                      int max_blocks, int tile_points) {
  const int b = blockIdx.x;
  if (states[b].phase != GICP_LINEARIZE) return;
  if (HGS_CONTROL_PRIO) __builtin_amdgcn_s_setprio(HGS_CONTROL_PRIO);
  // tiles of the linearize kernel that filled `partials`; its short-packet launches (tile_points < kBlock) leave one row per 64-point chunk (reduce_tiles)
  const int n = descs[b].meta->nvalid;
  const bool chunked = tile_points < kBlock;
  const int ntiles = chunked ? (n + kBlock - 1) / kBlock : (n + tile_points - 1) / tile_points;
  gicp_solve_block(ntiles, states + b, c, partials + (size_t)b * max_blocks * kAcc, chunked ? (n + 63) / 64 : 0);
}
void launch_gicp_solve(hipStream_t s, const CloudDesc* descs, GicpState* states, GicpConsts c, const double* partials, int max_blocks, int B,
                       int tile_points) {
  hipLaunchKernelGGL(k_gicp_solve, dim3(B), dim3(kSolveBlock), 0, s, descs, states, c, partials, max_blocks, tile_points);
}

// compute_error(xi): same correspondences, Mahalanobis matrices of the linearisation pose x0, residuals at xi.
// ROUND2 (k_gicp_error<true>): the LM solve of the linearisation the previous k_gicp_linearize<true> made (gicp_solve_block's arithmetic, in its order)
// runs first, in every block; `partials` / lin_tile_points: that kernel's tile partials and tiling.
template <bool ROUND2>
__global__ __launch_bounds__(kBlock) void k_gicp_error(const CloudDesc* descs, TargetView tgt, const GicpState* states, double* __restrict__ partials_err,
                                                       int max_blocks, GicpConsts c, GicpState* states_out, const double* __restrict__ partials, int lin_tile_points) {
  const int b = blockIdx.y;
  __shared__ GicpState st2;
  __shared__ double ws2[kGicpControlWorkspace];
  __shared__ double acc2[kAcc];
  __shared__ double scratch2[kSolveBlock];
  if constexpr (!ROUND2) {
    if (states[b].phase != GICP_TRY) return;
  }
  const CloudDesc d = descs[b];
  const int n = d.meta->nvalid;
  if constexpr (ROUND2) {
    static_assert(kSolveBlock == kBlock, "reduce_tiles runs on the point kernel's block");
    const int phase_in = states[b].phase;  // (block-uniform)
    gicp_state_load(st2, states + b);
    if (phase_in == GICP_LINEARIZE) {
      const bool chunked = lin_tile_points < kBlock;  // (short-packet launches: one row per 64-point chunk, k_gicp_solve)
      const int ntiles_lin = chunked ? (n + kBlock - 1) / kBlock : (n + lin_tile_points - 1) / lin_tile_points;
      reduce_tiles<kAcc>(partials + (size_t)b * max_blocks * kAcc, ntiles_lin, acc2, scratch2, chunked ? (n + 63) / 64 : 0);  // (ends with a barrier: the state copy is complete)
      if (threadIdx.x == 0) gicp_after_linearize(st2, acc2, c, ws2);
    }
    __syncthreads();
    if (blockIdx.x == 0) gicp_state_store(states_out + b, st2);
    if (st2.phase != GICP_TRY) return;
  }
  const int ntiles = (n + kBlock - 1) / kBlock;
  if ((int)blockIdx.x >= ntiles) return;
  const int tile = (int)blockIdx.x;
  const int i = tile * kBlock + threadIdx.x;
  __shared__ double lds[4];
  double err = 0.0;
  if (i < n) {
    const int j = d.corr[i];
    if (j >= 0) {
      const Pose T0 = ROUND2 ? st2.x0 : states[b].x0;
      const Pose Ti = ROUND2 ? st2.xi : states[b].xi;
      const double R[9] = {T0.m[0], T0.m[1], T0.m[2], T0.m[4], T0.m[5], T0.m[6], T0.m[8], T0.m[9], T0.m[10]};
      const Sym3 M = gicp_mahalanobis(R, load_cov(d.cov, i), load_cov(tgt.cov, j));
      const float4 a = d.pts[i], bp = tgt.pts[j];
      err = gicp_point_terms<false>(Ti, M, a.x, a.y, a.z, bp.x, bp.y, bp.z, nullptr);
    }
  }
  block_reduce_store<1>(&err, partials_err + (size_t)b * max_blocks + tile, lds);
}
void launch_gicp_error(hipStream_t s, const CloudDesc* descs, TargetView tgt, const GicpState* states, double* partials_err, int max_blocks, int B) {
  hipLaunchKernelGGL(k_gicp_error<false>, dim3(max_blocks, B), dim3(kBlock), 0, s, descs, tgt, states, partials_err, max_blocks, GicpConsts{}, (GicpState*)nullptr, (const double*)nullptr, 0);
}
void launch_gicp_error_round2(hipStream_t s, const CloudDesc* descs, TargetView tgt, const GicpState* states_in, GicpState* states_out, GicpConsts c, const double* partials,
                              double* partials_err, int max_blocks, int err_blocks, int B, int lin_tile_points) {
  hipLaunchKernelGGL(k_gicp_error<true>, dim3(err_blocks, B), dim3(kBlock), 0, s, descs, tgt, states_in, partials_err, max_blocks, c, states_out, partials, lin_tile_points);
}

// The LM accept / reject step behind compute_error, run by ONE wave (the first 64 threads of the calling block; the others only pass the barriers):
// Returns (to thread 0) whether the problem finished with this step.
__device__ __forceinline__ bool gicp_decide_wave(int ntiles, GicpState* state, const GicpConsts& c, const double* __restrict__ problem_partials_err) {
  __shared__ GicpState st;  // as in gicp_solve_block: the one-lane control step works on LDS
  __shared__ double ws[kGicpControlWorkspace];
  bool finished_now = false;
  double s = 0;
  if (threadIdx.x < 64) {
    for (int k = threadIdx.x; k < (int)(sizeof(GicpState) / sizeof(double)); k += 64) reinterpret_cast<double*>(&st)[k] = reinterpret_cast<const double*>(state)[k];
    for (int t = threadIdx.x; t < ntiles; t += 64) s += problem_partials_err[t];
    s = wave_sum(s);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    gicp_after_error(st, s, c, ws);
    finished_now = st.phase == GICP_DONE;
  }
  __syncthreads();
  if (threadIdx.x < 64)
    for (int k = threadIdx.x; k < (int)(sizeof(GicpState) / sizeof(double)); k += 64) reinterpret_cast<double*>(state)[k] = reinterpret_cast<const double*>(&st)[k];
  return finished_now;
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HGS_CONTROL_WAVES))) void k_gicp_decide(const CloudDesc* descs, GicpState* states, GicpConsts c, const double* __restrict__ partials_err,
                                                   int max_blocks, Progress prog) {
  const int b = blockIdx.x;
  bool finished_now = false;
  if (HGS_CONTROL_PRIO) __builtin_amdgcn_s_setprio(HGS_CONTROL_PRIO);
  if (states[b].phase == GICP_TRY) {  // (block-uniform)
    const int ntiles = (descs[b].meta->nvalid + kBlock - 1) / kBlock;
    finished_now = gicp_decide_wave(ntiles, states + b, c, partials_err + (size_t)b * max_blocks);
    // (the tick below only tells the host how far the batch is; states[] is read by later kernels of this stream, after this one has ended)
  }
  if (threadIdx.x == 0) progress_tick(prog, finished_now);
}
void launch_gicp_decide(hipStream_t s, const CloudDesc* descs, GicpState* states, GicpConsts c, const double* partials_err, int max_blocks, int B,
                        Progress prog) {
  hipLaunchKernelGGL(k_gicp_decide, dim3(B), dim3(64), 0, s, descs, states, c, partials_err, max_blocks, prog);
}

__global__ void k_gicp_results(const GicpState* states, DevResult* out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  out[b] = gicp_result_of(states[b]);
}
void launch_gicp_results(hipStream_t s, const GicpState* states, DevResult* out, int B) {
  hipLaunchKernelGGL(k_gicp_results, dim3((B + 63) / 64), dim3(64), 0, s, states, out, B);
}

// ------------------------------------------------------------------------------------------------ fitness / NN queries
// getFitnessScore: per source point exact (unbounded) 1-NN in the target; sum d2 over d2 <= max_range.
// Algorithmic bytes per source point: 16 + 16 = 32.  Only the distance is needed: the quad walk runs without leaf / tie tracking.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(HGS_FITNESS_WAVES))) void k_fitness(const CloudDesc* descs, TargetView tgt, const DevResult* poses, double max_range,
                                                    double* __restrict__ partials, int max_blocks, int use_seed, int qpw) {
  const int b = blockIdx.y;
  const CloudDesc d = descs[b];
  const int n = d.meta->nvalid;
  const int tile_pts = (kBlock / 64) * qpw * kNW;
  const int ntiles = (n + tile_pts - 1) / tile_pts;
  const int tile = xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  __shared__ double lds[4 * 2];
  __shared__ unsigned arrivals;
  __shared__ __attribute__((aligned(512))) float park[kBlock / 64][kParkFloats];
  if (threadIdx.x == 0) arrivals = 0u;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float Tf[12];
  const float* Tc = poses[b].T;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int cc = 0; cc < 4; cc++) Tf[r * 4 + cc] = Tc[cc * 4 + r];
  const int idx = tile * tile_pts + wave * qpw + (int)(threadIdx.x & 63);
  const bool active = (int)(threadIdx.x & 63) < qpw && idx < n;
  const float4 a = active ? load_stream(d.pts + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  const F3 q = transform_point_f(Tf, a.x, a.y, a.z);
  // use_seed: corr[] holds this cloud's last GICP correspondences against this target — a tight starting bound
  int seed = (active && use_seed) ? __builtin_nontemporal_load(d.corr + idx) : -1;
  const BvhView tv = view_of(tgt);
  if (tgt.seed_bits && __ballot(active && !(seed >= 0 && seed < tv.n)) != 0ull) {  // NDT / VGICP / calc_fitness_score: no correspondences -> the seed grid
    if (active && !(seed >= 0 && seed < tv.n)) seed = seed_grid_lookup(tgt, q);
  }
  const bool seeded = seed >= 0 && seed < tv.n;
  float d2;
  int leaf;
  bool tie, found;
  if (__ballot(active && !seeded) != 0ull) wave_nn1_quad<true, false>(tv, park[wave], q, active, FLT_MAX, seed, qpw >> 1, d2, leaf, tie, found);
  else wave_nn1_quad<false, false>(tv, park[wave], q, active, FLT_MAX, seed, qpw >> 1, d2, leaf, tie, found);
  double acc[2] = {0.0, 0.0};
  if (active && found && (double)d2 <= max_range) acc[0] = (double)d2, acc[1] = 1.0;
  int lane;
  HGS_LANE_ID(lane);
  {
    const int slot[2] = {0, 1};
    wave_sums_to<2>(acc, slot, lds + wave * 2, lane);
  }
  last_wave_stores<2>(lds, &arrivals, partials + ((size_t)b * max_blocks + tile) * 2, lane);
}
void launch_fitness(hipStream_t s, const CloudDesc* descs, TargetView tgt, const DevResult* poses, double max_range, double* partials, int max_blocks,
                    int B, int use_seed, int qpw) {
  hipLaunchKernelGGL(k_fitness, dim3(HGS_GRID_X(max_blocks), B), dim3(kBlock), 0, s, descs, tgt, poses, max_range, partials, max_blocks, use_seed, qpw);
}
__global__ __launch_bounds__(64) void k_fitness_final(const CloudDesc* descs, const double* __restrict__ partials, int max_blocks, DevResult* out, int tile_pts) {
  const int b = blockIdx.x;
  const int ntiles = (descs[b].meta->nvalid + tile_pts - 1) / tile_pts;  // tiles of k_fitness
  double s = 0, c = 0;
  for (int t = threadIdx.x; t < ntiles; t += 64) {
    s += partials[((size_t)b * max_blocks + t) * 2];
    c += partials[((size_t)b * max_blocks + t) * 2 + 1];
  }
  s = wave_sum(s), c = wave_sum(c);
  if (threadIdx.x == 0) out[b].fit_sum = s, out[b].fit_count = (unsigned)(c + 0.5);
}
void launch_fitness_final(hipStream_t s, const CloudDesc* descs, const double* partials, int max_blocks, DevResult* out, int B, int tile_pts) {
  hipLaunchKernelGGL(k_fitness_final, dim3(B), dim3(64), 0, s, descs, partials, max_blocks, out, tile_pts);
}

// DevResult -> the public per-candidate record (hgs_result, include/hgs_registration.h), written where the all-gather of a
// sharded batch sends from: slot s < n_slots, padded with candidate_id -1 records beyond this rank's n candidates.
__global__ void k_results_to_records(const DevResult* __restrict__ res, const int* __restrict__ candidate_ids, int n, int n_slots, hgs_result* __restrict__ out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  hgs_result r;
  if (s < n) {
    const DevResult d = res[s];
#pragma unroll
    for (int k = 0; k < 16; k++) r.final_transformation[k] = d.T[k];
    r.converged = d.converged, r.iterations = d.iterations;
    r.error = d.error;
    r.num_inliers = d.fit_count;
    r.fitness_score = d.fit_count > 0 ? d.fit_sum / (double)d.fit_count : DBL_MAX;
    r.candidate_id = candidate_ids[s];
    r.lm_tries = d.lm_tries, r.reserved = 0;
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) r.final_transformation[k] = 0.f;
    r.converged = 0, r.iterations = 0, r.error = 0.0, r.fitness_score = DBL_MAX, r.num_inliers = 0, r.candidate_id = -1, r.lm_tries = 0, r.reserved = 0;
  }
  out[s] = r;
}
void launch_results_to_records(hipStream_t s, const DevResult* res, const int* candidate_ids, int n, int n_slots, hgs_result* out) {
  if (n_slots <= 0) return;
  hipLaunchKernelGGL(k_results_to_records, dim3((n_slots + 63) / 64), dim3(64), 0, s, res, candidate_ids, n, n_slots, out);
}

__global__ __launch_bounds__(kBlock) void k_nn_query(TargetView tgt, const float4* __restrict__ q, int nq, int* __restrict__ idx, float* __restrict__ d2out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nq) return;
  const float4 p = q[i];
  float d2;
  int orig;
  const int j = bvh_nn1(view_of(tgt), F3{p.x, p.y, p.z}, FLT_MAX, &d2, &orig);
  idx[i] = j >= 0 ? orig : -1;
  d2out[i] = d2;
}
void launch_nn_query(hipStream_t s, TargetView tgt, const float4* q, int nq, int* idx, float* d2) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(k_nn_query, dim3((nq + kBlock - 1) / kBlock), dim3(kBlock), 0, s, tgt, q, nq, idx, d2);
}

__global__ __launch_bounds__(kBlock) void k_transform(const float4* __restrict__ raw, int n, const float* __restrict__ T16, float4* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float Tf[12];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int cc = 0; cc < 4; cc++) Tf[r * 4 + cc] = T16[cc * 4 + r];
  const float4 a = raw[i];
  const F3 q = transform_point_f(Tf, a.x, a.y, a.z);
  out[i] = make_float4(q.x, q.y, q.z, 1.0f);
}
void launch_transform(hipStream_t s, const float4* raw, int n, const float* T16_dev, float4* out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_transform, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, raw, n, T16_dev, out);
}

// ------------------------------------------------------------------------------------------------ NDT target voxelisation
__global__ void k_ndt_grid_params(CloudDesc d, float inv_leaf) {
  CloudMeta* m = d.meta;
  m->ndt_ncells = 0;
  m->ndt_error = 0;
  if (m->nvalid <= 0) {
    for (int k = 0; k < 3; k++) m->ndt_min_b[k] = 0, m->ndt_max_b[k] = -1, m->ndt_div_mul[k] = 0;
    return;
  }
  long long div[3];
  for (int k = 0; k < 3; k++) {
    m->ndt_min_b[k] = (int)floorf(ord2f(m->bbmin[k]) * inv_leaf);
    m->ndt_max_b[k] = (int)floorf(ord2f(m->bbmax[k]) * inv_leaf);
    div[k] = (long long)m->ndt_max_b[k] - m->ndt_min_b[k] + 1;
  }
  if (div[0] * div[1] * div[2] > 2147483647LL) m->ndt_error = 1;
  m->ndt_div_mul[0] = 1, m->ndt_div_mul[1] = (int)div[0], m->ndt_div_mul[2] = (int)(div[0] * div[1]);
}
void launch_ndt_grid_params(hipStream_t s, CloudDesc desc, float inv_leaf) { hipLaunchKernelGGL(k_ndt_grid_params, dim3(1), dim3(1), 0, s, desc, inv_leaf); }

__global__ __launch_bounds__(kBlock) void k_ndt_cell_keys(CloudDesc d, float inv_leaf, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= d.n_input) return;
  const float4 p = d.raw[i];
  unsigned long long key = 0xffffffffull;
  const CloudMeta* m = d.meta;
  if (finite3(p) && !m->ndt_error) {
    const int cx = (int)floorf(p.x * inv_leaf) - m->ndt_min_b[0];
    const int cy = (int)floorf(p.y * inv_leaf) - m->ndt_min_b[1];
    const int cz = (int)floorf(p.z * inv_leaf) - m->ndt_min_b[2];
    key = (unsigned long long)(unsigned)(cx * m->ndt_div_mul[0] + cy * m->ndt_div_mul[1] + cz * m->ndt_div_mul[2]);
  }
  keys[i] = key;
  vals[i] = (unsigned)i;
}
void launch_ndt_cell_keys(hipStream_t s, CloudDesc desc, float inv_leaf, unsigned long long* keys, unsigned* vals) {
  if (desc.n_input <= 0) return;
  hipLaunchKernelGGL(k_ndt_cell_keys, dim3((desc.n_input + kBlock - 1) / kBlock), dim3(kBlock), 0, s, desc, inv_leaf, keys, vals);
}

// One thread per segment head accumulates its cell (double), finalises it and inserts it into the hash table.
__global__ __launch_bounds__(kBlock) void k_ndt_build_cells(CloudDesc d, const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                            int min_points, int* hash_keys, int* hash_vals, int hash_mask, NdtCellRec* cells) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= d.n_input) return;
  const unsigned long long key = keys[i];
  if (key == 0xffffffffull) return;
  if (i > 0 && keys[i - 1] == key) return;
  double sum[3] = {0, 0, 0};
  Sym3 sq = {0, 0, 0, 0, 0, 0};
  int n = 0;
  for (int j = i; j < d.n_input && keys[j] == key; j++) {
    const float4 p = d.raw[vals[j]];
    const double x = p.x, y = p.y, z = p.z;
    sum[0] += x, sum[1] += y, sum[2] += z;
    sq.xx += x * x, sq.xy += x * y, sq.xz += x * z, sq.yy += y * y, sq.yz += y * z, sq.zz += z * z;
    n++;
  }
  double mean[3];
  Sym3 icov;
  if (!ndt_finalize_cell(n, sum, sq, min_points, mean, &icov)) return;
  const int slot_c = atomicAdd(&d.meta->ndt_ncells, 1);
  NdtCellRec rec;
  rec.v0 = make_float4((float)icov.xx, (float)icov.xy, (float)icov.xz, (float)icov.yy);
  rec.v1 = make_float4((float)icov.yz, (float)icov.zz, (float)n, __int_as_float((int)key));
  rec.mean[0] = mean[0], rec.mean[1] = mean[1], rec.mean[2] = mean[2], rec.pad = 0.0;
  cells[slot_c] = rec;
  unsigned slot = (ndt_hash((int)key) >> 7) & (unsigned)hash_mask;
  for (;;) {
    const int prev = atomicCAS(&hash_keys[slot], -1, (int)key);
    if (prev == -1) {
      hash_vals[slot] = slot_c;
      break;
    }
    slot = (slot + 1) & (unsigned)hash_mask;
  }
}
void launch_ndt_build_cells(hipStream_t s, CloudDesc desc, const unsigned long long* sorted_keys, const unsigned* sorted_vals, int min_points,
                            int* hash_keys, int* hash_vals, int hash_mask, NdtCellRec* cells) {
  if (desc.n_input <= 0) return;
  hipLaunchKernelGGL(k_ndt_build_cells, dim3((desc.n_input + kBlock - 1) / kBlock), dim3(kBlock), 0, s, desc, sorted_keys, sorted_vals, min_points,
                     hash_keys, hash_vals, hash_mask, cells);
}

// ------------------------------------------------------------------------------------------------ NDT iteration
// one 64-thread block per problem: thread 0 sets the state up, all of them zero the problem's digit totals (a separate memset in front of this kernel
// was one more dispatch on the single-registration path)
__global__ __launch_bounds__(64) void k_ndt_init(NdtState* states, NdtAngles* angles, const float* guesses, NdtConsts c, int B, Progress prog, NdtAccum* accum) {
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0) prog.dev[0] = 0, prog.dev[1] = 0;
  if (accum) {
    static_assert(sizeof(NdtAccum) % sizeof(unsigned long long) == 0, "NdtAccum is zeroed word by word");
    unsigned long long* w = reinterpret_cast<unsigned long long*>(accum + b);
    for (int k = threadIdx.x; k < (int)(sizeof(NdtAccum) / sizeof(unsigned long long)); k += 64) w[k] = 0ull;
  }
  if (threadIdx.x == 0) {
    ndt_state_init(states[b], guesses + 16 * b);
    ndt_angle_tables(states[b].p, c.upstream_hd1_sign, angles[b]);
  }
}
void launch_ndt_init(hipStream_t s, NdtState* states, NdtAngles* angles, const float* guesses, NdtConsts c, int B, Progress prog, NdtAccum* accum_to_zero) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_ndt_init, dim3(B), dim3(64), 0, s, states, angles, guesses, c, B, prog, accum_to_zero);
}

// hash_kv[slot] = (key, cell index): one 8-byte load per probe in the derivative kernel
__global__ __launch_bounds__(kBlock) void k_ndt_pack_hash(const int* __restrict__ keys, const int* __restrict__ vals, int2* __restrict__ kv, int cap) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  const int k = keys[i];
  kv[i] = make_int2(k, k == -1 ? -1 : vals[i]);
}
void launch_ndt_pack_hash(hipStream_t s, const int* keys, const int* vals, int2* kv, int cap) {
  hipLaunchKernelGGL(k_ndt_pack_hash, dim3((cap + kBlock - 1) / kBlock), dim3(kBlock), 0, s, keys, vals, kv, cap);
}

// SVD-solve(A, b) by the calling wave: the three rotations of a Jacobi round on lanes 0..2, U and V (36 doubles each) in
// LDS.  Same arithmetic and the same pair order as solve_svd6 (hgs_math.h) => same bits; x is valid on lane 0.
// LDS accesses of one wave are performed in program order, and the volatile qualifier keeps the compiler from caching a
// column across rounds, so a lane reads the columns the other lanes rotated in the round before; the wave barrier between
// rounds emits no instruction — it states the lock-step dependence (and is where the host emulation of tests/emul lets the
// three lanes meet).
__device__ __forceinline__ void solve_svd6_wave(const double* A, const double* b, double* U, double* V, double* x) {
  const int lane = (int)(threadIdx.x & 63);
  if (lane < 36) U[lane] = A[lane], V[lane] = (lane % 7 == 0) ? 1.0 : 0.0;
  HGS_COMPILER_MEMORY_BARRIER();
  __builtin_amdgcn_wave_barrier();
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int round = 0; round < 5; round++) {
      // the three pairs of a round are disjoint columns: each lane loads its four columns (24 LDS reads in flight together), rotates
      // in registers, stores.  Plain pointers + a compiler-level memory barrier per round instead of volatile accesses, which were
      // issued and waited for one at a time (~48 LDS round trips per rotation: 40 us of a 92 us odometry iteration in round 2).
      if (lane < 3) {
        int p, q;
        svd6_pair(round, lane, &p, &q);
        if (svd6_rotate<double*>(U, V, p, q)) rotated = true;
      }
      HGS_COMPILER_MEMORY_BARRIER();
      __builtin_amdgcn_wave_barrier();
    }
    if (__ballot(rotated) == 0ull) break;
  }
  if (lane == 0) svd6_backsolve<const double*>(U, V, b, x);
}

// One NDT iteration of a batch in ONE launch (computeDerivatives + the Newton step of computeTransformation /
// computeStepLengthMT).  Work items are (problem, tile of 256 points); the resident blocks pull chunks of consecutive items
// from a queue in HBM, so the load balances itself whatever the sizes of the clouds and however many problems of the batch
// are still iterating (a finished problem's items cost one phase read each).
//   1. per source point: transform, look up the DIRECT1 / DIRECT7 / KDTREE cells (all hash probes of a point are issued
//      together, the cell records are fetched one visit ahead of the arithmetic), accumulate the point's score /
//      gradient / Hessian over its cells in neighbourhood order (double, like ndt_omp's per-point sums);
//   2. the per-point doubles are split into fixed-grid integer chunks (hgs_ndt.h "exact accumulation") and summed as
//      integers: lane (every lane adds its chunks to its own 64-bit slot of the block's table in LDS, ndt_reduce_sums: no cross-lane
//      step per tile) -> block (the 64 lanes of a slot added up once per flush, over all the tiles of one problem the
//      block works on in a row) -> problem (64-bit atomics in HBM, one per digit total and flush).  Integer addition is
//      associative: the totals do not depend on tiling, on which block took which tile or on the order the points are
//      stored in, so the points are read in whatever order is resident (Hilbert order when the cloud has a search index —
//      neighbouring lanes then share cells —, input order otherwise);
//   3. the block whose flush completes a problem's tile count turns the totals into doubles, runs the Newton step
//      (Jacobi-SVD solve on three lanes, step clamp, convergence test, next angle tables) and ticks the batch progress.
// Algorithmic bytes per source point: 16 + 7*40 = 296 (DIRECT7), 56 (DIRECT1); the cell table of a LiDAR scan is L2-resident.
__device__ __forceinline__ int ndt_cell_coord(float v) { return (int)floorf(fminf(fmaxf(v, -1.0e9f), 1.0e9f)); }  // NaN -> -1e9

// The cell indices of a lane's neighbourhood live in LDS ([offset][thread]): a lane takes them in the order of ITS valid cells, i.e. with a per-lane
// dynamic index, which the compiler served from a scratch array (one scratch load in front of every cell record load: +4 % on the NDT batch).
template <int NOFF>
__device__ __forceinline__ int ndt_pop_cell_lds(unsigned& mask, const int* col /* &ci_lds[0][thread] */) {
  if (!mask) return -1;
  const int o = __builtin_ctz(mask);
  mask &= mask - 1u;
  return col[o * kBlock];
}
// The angle tables of a problem as k_ndt_pass keeps them in LDS: component-major, and the rows in the order in which the per-cell arithmetic wants
// them as PAIRS (j rows 0,1 | 2,5 | 3,6 | 4,7: the point-gradient columns 4 and 5 side by side; then the 15 h rows), so that one ds_read_b64 delivers the
// two factors of a v_pk_mul_f32.  In the [row][component] order of NdtAngles the compiler built the same pairs with ~100 register moves per tile.
struct NdtAnglesLds {
  float tab[3][24];  // [component][slot]; slot 23 is padding (zero)
  float T[12];
};
__device__ __forceinline__ int ndt_angle_slot(int row /* 0..7: j, 8..22: h */) {
  return row >= 8 ? row : (row < 2 ? row : (row == 2 ? 2 : (row == 5 ? 3 : (row == 3 ? 4 : (row == 6 ? 5 : (row == 4 ? 6 : 7))))));
}
// ndt_point_derivatives on the LDS layout: per row (a0 x + a1 y) + a2 z, unfused, the scalar form's operations on two rows at a time
__device__ __forceinline__ void ndt_point_derivatives_lds(const NdtAnglesLds& a, float x, float y, float z, NdtPointDeriv& d) {
#pragma clang fp contract(off)
  const ndt_f2 X = {x, x}, Y = {y, y}, Z = {z, z};
  const ndt_f2* tx = reinterpret_cast<const ndt_f2*>(a.tab[0]);
  const ndt_f2* ty = reinterpret_cast<const ndt_f2*>(a.tab[1]);
  const ndt_f2* tz = reinterpret_cast<const ndt_f2*>(a.tab[2]);
  ndt_f2 r[12];
#pragma unroll
  for (int s = 0; s < 12; s++) r[s] = tx[s] * X + ty[s] * Y + tz[s] * Z;
  d.xj[0] = r[0].x, d.xj[1] = r[0].y, d.xj[2] = r[1].x, d.xj[5] = r[1].y, d.xj[3] = r[2].x, d.xj[6] = r[2].y, d.xj[4] = r[3].x, d.xj[7] = r[3].y;
#pragma unroll
  for (int k = 0; k < 15; k++) d.xh[k] = (k & 1) ? r[4 + (k >> 1)].y : r[4 + (k >> 1)].x;
}

// ---- the three memory stages in front of a tile's arithmetic; k_ndt_pass runs them for tile t + 1 between the pieces of tile t's
// digit reduction (point -> hash probes -> first cell record are dependent loads: with two waves per SIMD their latencies were exposed)
struct NdtGridBox {
  int mnx, mny, mnz, mxx, mxy, mxz, mul1, mul2;
};

// hash key of neighbourhood cell o of a transformed point; -1 outside the grid's box and for a lane without a point.
// A transformed source point can be anything (non-finite, 1e30): clamp before the conversion — such a coordinate is
// outside every grid either way, and float -> int of a NaN or an out-of-range value is not defined in C++
__device__ __forceinline__ int ndt_front_key(const F3& xt, int have, const NdtTargetView& tgt, const NdtGridBox& g, int search, int o) {
  const int cx = ndt_cell_coord(xt.x * tgt.inv_leaf), cy = ndt_cell_coord(xt.y * tgt.inv_leaf), cz = ndt_cell_coord(xt.z * tgt.inv_leaf);
  int ox, oy, oz;
  ndt_offset(search, o, &ox, &oy, &oz);
  const int px = cx + ox, py = cy + oy, pz = cz + oz;
  const bool in = have && px >= g.mnx && px <= g.mxx && py >= g.mny && py <= g.mxy && pz >= g.mnz && pz <= g.mxz;
  return in ? (px - g.mnx) + (py - g.mny) * g.mul1 + (pz - g.mnz) * g.mul2 : -1;
}

// x: the lane's source point; have: the lane has one
__device__ __forceinline__ void ndt_front_load(float4& x, int& have, const CloudDesc& d, int sorted, int i, int n) {
  have = i < n ? 1 : 0;
  x = make_float4(0.f, 0.f, 0.f, 0.f);
  if (have) x = HGS_LOAD_GLOBAL_XYZ((sorted ? d.pts : d.raw) + i);
}

// xt: the point under the problem's current pose; kv: the first probe of every neighbourhood cell, in flight on return
template <int NOFF>
__device__ __forceinline__ void ndt_front_issue_probes(const float4& x, int& have, F3& xt, unsigned long long (&kv)[NOFF], const NdtAnglesLds* ap, const NdtTargetView& tgt, const NdtGridBox& g,
                                                       int search, int sorted) {
  if (!sorted && !finite3(x)) have = 0;
  xt = transform_point_f(ap->T, x.x, x.y, x.z);
#pragma unroll
  for (int o = 0; o < NOFF; o++) {
    const int key = ndt_front_key(xt, have, tgt, g, search, o);
    const unsigned slot = (ndt_hash(key) >> 7) & (unsigned)tgt.hash_mask;
    kv[o] = key >= 0 ? reinterpret_cast<const unsigned long long*>(tgt.hash_kv)[slot] : ~0ull;  // {key, cell index}, raw
  }
}

// ci: cell record index or -1 per neighbourhood cell; vmask: the valid ones
template <int NOFF>
__device__ __forceinline__ void ndt_front_resolve(const F3& xt, int have, unsigned long long (&kv)[NOFF], int (&ci)[NOFF], unsigned& vmask, const NdtTargetView& tgt, const NdtGridBox& g, int search) {
  vmask = 0;
#pragma unroll
  for (int o = 0; o < NOFF; o++) {
    const int key = ndt_front_key(xt, have, tgt, g, search, o);  // recomputed (a dozen integer operations) rather than carried through the reduction
    unsigned slot = (ndt_hash(key) >> 7) & (unsigned)tgt.hash_mask;
    unsigned long long e = kv[o];
    while ((int)(unsigned)e != key && (int)(unsigned)e != -1) {  // linear probing (load factor <= 1/4: almost never taken)
      slot = (slot + 1u) & (unsigned)tgt.hash_mask;
      e = reinterpret_cast<const unsigned long long*>(tgt.hash_kv)[slot];
    }
    ci[o] = (key >= 0 && (int)(unsigned)e == key) ? (int)(unsigned)(e >> 32) : -1;
    if (ci[o] >= 0) vmask |= 1u << o;
  }
}

// Accumulators [K0, K1) of a tile's exact accumulation (round 4): every lane adds the two integer chunks of its per-point double to ITS OWN
// 64-bit slot of the block's accumulator table in LDS — slot [2k + chunk][lane], consecutive lanes on consecutive 8-byte words: ds_add_u64 without
// return, no bank conflict, nothing to wait for — and the 64 lanes of a slot are added up once per flush (ndt_flush) instead of once per tile.
// The chunk q (|q| < 2^49) arrives as the double 1.5 * 2^52 + q, whose bit pattern is 0x4338000000000000 + q: the constant's low word is zero, so
// ONE 32-bit subtraction on the high word leaves q in two's complement.  Integer addition is associative: the totals are the ones the per-tile
// cross-lane reduction of rounds 2-3 produced (25/27-bit digits through v_permlane32_swap / v_permlane16_swap / DPP: 31 vector instructions per sum
// and tile, 38 % of the kernel's; now 8 + two LDS atomics).  A slot takes at most kNdtFlushTiles tiles x 4 waves between flushes: |sum| < 2^56.
constexpr int kNdtFlushTiles = 32;  // a block's totals of one flush stay below 2^(49 + 5 + 8) = 2^62
constexpr unsigned long long kNdtMagicBits = 0x4338000000000000ull;
#ifndef HGS_OPAQUE_VGPR64  // (the host emulation supplies its own spelling)
#define HGS_OPAQUE_VGPR64(x) asm volatile("" : "+v"(x))
#endif
// The slots receive the RAW bit patterns (magic constant included): every slot of the table gets exactly one addition per wave and tile,
// so the flush takes 64 * (number of wave-tile additions) * 0x4338000000000000 off a slot's lane sum, modulo 2^64, instead of every lane
// subtracting it from every chunk.  The magic constant stays in a vector register pair the compiler cannot see through and the three powers
// of two in scalar pairs: a split is v_fma_f64, v_add_f64, v_fma_f64, v_fma_f64 and a compare (the literal-operand v_fmac_f64 the compiler
// prefers needs its 64-bit addend copied into the destination first: two v_mov per fma).
template <int K0, int K1>
__device__ __forceinline__ bool ndt_reduce_class(const double (&acc)[kAccNdt], unsigned long long (*slots)[64], int lane, double magic) {
  HGS_FP_STRICT
  bool bad = false;
  if (K0 < K1) {
    // the accumulators [K0, K1) share one exponent: its three powers of two and the range bound, once, as scalars
    const int E = ndt_sum_exponent(K0);
    const double s0 = HGS_NDT_EXP_COEFF(ldexp(1.0, -(E + kNdtChunkBits))), S0 = HGS_NDT_EXP_COEFF(ldexp(1.0, E + kNdtChunkBits)), s1 = HGS_NDT_EXP_COEFF(ldexp(1.0, -E));
    const double bound = HGS_NDT_EXP_COEFF(ldexp(1.0, E + 2 * kNdtChunkBits - 1));
#pragma unroll
    for (int k = K0; k < K1; k++) {
      const double t = acc[k];
      const double a = fma(t, s0, magic);  // ndt_exact_split, operation for operation
      const double q0 = a - magic;
      const double r = fma(-q0, S0, t);
      const double m1 = fma(r, s1, magic);
      if (!(fabs(t) < bound)) bad = true;
      atomicAdd(&slots[2 * k][lane], (unsigned long long)__double_as_longlong(a));
      atomicAdd(&slots[2 * k + 1][lane], (unsigned long long)__double_as_longlong(m1));
    }
  }
  return bad;
}
template <int K0, int K1>
__device__ __forceinline__ bool ndt_reduce_sums(const double (&acc)[kAccNdt], unsigned long long (*slots)[64], int lane) {
  double magic = kNdtMagic;
  HGS_OPAQUE_VGPR64(magic);
  constexpr int H1 = K1 < 36 ? K1 : 36, G0 = K0 > 36 ? K0 : 36, G1 = K1 < 42 ? K1 : 42, S0 = K0 > 42 ? K0 : 42;  // Hessian | gradient | score
  bool bad = ndt_reduce_class<K0, H1>(acc, slots, lane, magic);
  if (ndt_reduce_class<G0, G1>(acc, slots, lane, magic)) bad = true;
  if (ndt_reduce_class<S0, K1>(acc, slots, lane, magic)) bad = true;
  return bad;
}

struct NdtPassShared {
  NdtAnglesLds ang;
  unsigned long long slots[kAccNdt * 2][64];         // [accumulator * 2 + chunk][lane]: integer chunk sums of the tiles since the last flush (43 KB)
  unsigned long long part[kAccNdt * 2][4];           // flush: a slot's four 16-lane partial sums
  unsigned adds;                                     // wave-tile additions since the last flush (each put one magic constant into every lane of every slot)
  unsigned long long w[kAccNdt * 4 + 4];             // finish: the problem's totals (+ ticket, overflow)
  double acc[kAccNdt];
  double svd[72];
  int next, next_chunk, last, out_of_range;
  // copies of the kernel arguments the out-of-line helpers need: passing them by reference put them into scratch memory
  NdtConsts consts;
  Progress prog;
  int debug;
};

// The block that completed problem b's tile count: totals -> doubles, Newton step.  All threads of the block call it.
__device__ __noinline__ void ndt_finish_problem(NdtPassShared& S, NdtAccum& A, NdtState& st, NdtAngles& angles_b) {
  const int t = (int)threadIdx.x;
  const NdtConsts& c = S.consts;
  const int debug = S.debug;
  // read-and-clear (the next pass starts from zero), one exchange per thread so that they are all in flight together
  if (t < kAccNdt * 4) S.w[t] = atomicExch(&A.w[t], 0ull);
  else if (t == kAccNdt * 4 + 1) S.w[t] = atomicExch(&A.tiles_done, 0u);
  else if (t == kAccNdt * 4 + 2) S.w[t] = atomicExch(&A.overflow, 0u);
  __syncthreads();
  if (t < kAccNdt) {
    // chunk = (signed) high digits * 2^25 + low digits ; V = chunk0 * 2^50 + chunk1
    const __int128 q0 = (__int128)(long long)S.w[t * 4 + 2] * ((__int128)1 << 25) + (__int128)S.w[t * 4];
    const __int128 q1 = (__int128)(long long)S.w[t * 4 + 3] * ((__int128)1 << 25) + (__int128)S.w[t * 4 + 1];
    S.acc[t] = S.w[kAccNdt * 4 + 2] ? __longlong_as_double(0x7ff8000000000000ll) : ndt_i128_to_double(q0 * ((__int128)1 << kNdtChunkBits) + q1, ndt_sum_exponent(t));
  }
  __syncthreads();
  if (debug) {
    if (t < kAccNdt) A.out[t] = S.acc[t];
  } else if (t < 64) {
    double dp_newton[6] = {0, 0, 0, 0, 0, 0};
    const bool no_direction_needed = ndt_pass_is_last(st, c, S.acc);  // wave-uniform
    __builtin_amdgcn_wave_barrier();  // every lane has judged the state before lane 0 rewrites it below
    if (!no_direction_needed) {
      double ng[6];
      for (int i = 0; i < 6; i++) ng[i] = -S.acc[36 + i];
      solve_svd6_wave(S.acc, ng, S.svd, S.svd + 36, dp_newton);
    }
    if (t == 0) {
      ndt_after_derivatives(st, S.acc, c, dp_newton);
      if (st.phase != NDT_DONE) ndt_angle_tables(st.p, c.upstream_hd1_sign, angles_b);
      progress_tick(S.prog, st.phase == NDT_DONE);
    }
  }
  __syncthreads();
}

// Adds the block's digit totals of problem b to HBM and accounts for `tiles` tiles of it; true (block-uniform) if that
// completed the problem's pass.
__device__ __noinline__ bool ndt_flush(NdtPassShared& S, NdtAccum& A, int tiles, int tiles_total) {
  const int t = (int)threadIdx.x;
  __syncthreads();
  // 86 slots x 64 lanes -> 86 signed totals: thread (slot, quarter) adds the 16 lanes of its quarter, clears them, and the four quarters meet in S.part.
  // The order in which a thread takes its 16 words is rotated so that no wave instruction meets a busy bank (MI355X_MICROARCH.md, LDS): a slot row is
  // 512 bytes, so the bank of word (s, l) depends on l alone — ds_read_b64 is served in two 32-thread groups with banks (2 l) mod 64, ds_write_b64 in four
  // 16-thread groups with banks (2 l) mod 32.  With r = (j + 4 qd + (s & 3) + 4 ((s >> 2) & 1)) mod 16 the 16 threads of a write group (4 slots x 4
  // quarters) take 16 different r, and the 32 threads of a read group 32 different l mod 32.  Round 4 rotated by the slot only: quarters 0 / 2 and 1 / 3
  // of a slot met on one bank in every read (2-way) and all four quarters in every clearing write (4-way) — 4.1e6 conflict quad-cycles per launch, all of
  // them here (profiles/r05_ndt_lds_conflicts.md).  Integer sums: the order of the additions changes nothing.
#pragma unroll
  for (int s0 = 0; s0 < kAccNdt * 2; s0 += kBlock / 4) {
    const int s = s0 + (t >> 2), qd = t & 3;
    if (s < kAccNdt * 2) {
      unsigned long long v = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) {
#if HGS_NDT_FLUSH_ROT
        const int l = qd * 16 + ((j + 4 * qd + (s & 3) + 4 * ((s >> 2) & 1)) & 15);
#else
        const int l = qd * 16 + ((j + (t >> 2)) & 15);
#endif
        v += S.slots[s][l], S.slots[s][l] = 0;
      }
      S.part[s][qd] = v;
    }
  }
  __syncthreads();
  if (t < kAccNdt * 4) {
    // the HBM totals keep the digit format of rounds 2-3 (a problem's chunk sum can exceed 64 bits): total = high * 2^25 + low, high signed
    const int k = t >> 2, r = t & 3;  // r: chunk0 low, chunk1 low, chunk0 high, chunk1 high
    const long long tot = (long long)((S.part[2 * k + (r & 1)][0] + S.part[2 * k + (r & 1)][1]) + (S.part[2 * k + (r & 1)][2] + S.part[2 * k + (r & 1)][3]) -
                                      (unsigned long long)S.adds * (kNdtMagicBits << 6));
    const unsigned long long v = r < 2 ? (unsigned long long)(tot & 0x1ffffffll) : (unsigned long long)(tot >> 25);
    if (v) atomicAdd(&A.w[t], v);
  } else if (t == kAccNdt * 4 + 1) {
    if (S.out_of_range) atomicOr(&A.overflow, 1u), S.out_of_range = 0;
  }
  __syncthreads();  // S.adds has been read
  if (t == 0) S.adds = 0;
  // The ticket below must not overtake the additions above.  Everything involved is a device-scope atomic read-modify-write
  // (performed at the coherence point, never cached), so it is enough that this thread's have been acknowledged before the
  // barrier: no __threadfence() here — on gfx950 its L2 write-back + invalidate would throw the cell table out of the
  // XCD's L2 for every block of the kernel, several times per tile.
  HGS_WAIT_VMEM();
  __syncthreads();
  if (t == 0) S.last = atomicAdd(&A.tiles_done, (unsigned)tiles) + (unsigned)tiles == (unsigned)tiles_total ? 1 : 0;
  __syncthreads();
  return S.last != 0;
}

template <int NOFF>
__global__ __launch_bounds__(kBlock, 2) void k_ndt_pass(const CloudDesc* __restrict__ descs, NdtTargetView tgt, NdtState* states, NdtAngles* angles, NdtConsts c,
                                                         NdtAccum* accum, const int* __restrict__ tile_base /* [B + 1] */, unsigned long long* queues /* [2] */, int B,
                                                         int parity, int chunk, int sorted, int debug, Progress prog) {
  __shared__ NdtPassShared S;
  const int lane = (int)(threadIdx.x & 63);
  for (int k = threadIdx.x; k < kAccNdt * 2 * 64; k += kBlock) (&S.slots[0][0])[k] = 0;
  // two queue heads used alternately: this pass counts on queues[parity] from 0 and zeroes the other one for the next pass
  // (which starts after this kernel has ended, and the pass before, which used it, has)
  unsigned long long* queue = queues + parity;
  if (blockIdx.x == 0 && threadIdx.x == 0) queues[parity ^ 1] = 0ull;
  // the pass's items (tiles of all problems): a scalar, and the item arithmetic below in 32 bits — 64-bit compares have no scalar form and kept a
  // VGPR pair alive (and spilled) through the whole pass.  The host bounds the count (run_batch: HGS_ERR_INVALID_ARGUMENT beyond it) so that head + blocks * chunk fits.
  const int n_items = __builtin_amdgcn_readfirstlane(tile_base[B]);
  // Guided self-scheduling: a grab takes (items left) / (2 * blocks) items, between 1 and `chunk` — long runs of one problem's
  // tiles while there is plenty of work (every change of problem costs a flush: ~170 atomics and a ticket round trip), single
  // tiles at the end of the pass (when the blocks must finish together).  "Items left" is judged from this block's last grab.
  // The first run of every block is static (block x starts at item x * first_chunk): no atomic round trip in front of the
  // first tile; the queue head counts from gridDim.x * first_chunk on.
  const int first_chunk = max(1, min(chunk, n_items / (2 * (int)gridDim.x)));
  const int static_items = (int)gridDim.x * first_chunk;
  if (threadIdx.x == 0) S.out_of_range = 0, S.adds = 0, S.next = (int)blockIdx.x * first_chunk, S.next_chunk = first_chunk, S.consts = c, S.prog = prog, S.debug = debug;
  __syncthreads();
  const CloudMeta* m = tgt.meta;
  const NdtGridBox box = {m->ndt_min_b[0], m->ndt_min_b[1], m->ndt_min_b[2], m->ndt_max_b[0], m->ndt_max_b[1], m->ndt_max_b[2], m->ndt_div_mul[1], m->ndt_div_mul[2]};
  const float d1 = (float)c.gauss_d1, d2 = (float)c.gauss_d2;
  // The tile at hand: point, transformed point, first hash probes, cell indices; `staged`: filled during the previous tile's reduction.
  // Not for the 27-cell neighbourhood, whose 27 probes in flight do not fit the register file next to the 43 sums.
  constexpr bool kStaged = NOFF <= 7;
  float4 fx = make_float4(0.f, 0.f, 0.f, 0.f);
  F3 fxt = {0.f, 0.f, 0.f};
  int fhave = 0, fci[NOFF];
  __shared__ int ci_lds[NOFF][kBlock];
  // two blocks per CU (__launch_bounds__(kBlock, 2)) only while the static LDS of a block stays under half of the CU's 160 KB
  static_assert(sizeof(NdtPassShared) + sizeof(int) * NOFF * kBlock <= 80 * 1024, "k_ndt_pass: static LDS above 80 KB drops residency to one block per CU");
  int* const ci_col = &ci_lds[0][threadIdx.x];
  unsigned long long fkv[NOFF];
  unsigned fvmask = 0;
  bool staged = false;
  int cur_b = -1, cur_active = 0, cur_tiles = 0, cur_n = 0, cur_first = 0, cur_end = 0;  // block-uniform: the problem being worked on
  CloudDesc d{};
  int w = S.next;
  while (w < n_items) {
    __syncthreads();  // everybody has read S.next
    int nxt = 0;
    const int my_chunk = S.next_chunk;
    const int left = n_items - w - my_chunk;
    const int nxt_chunk = max(1, min(chunk, left / (2 * (int)gridDim.x)));
    if (threadIdx.x == 0) nxt = (int)min((unsigned long long)n_items, (unsigned long long)static_items + atomicAdd(queue, (unsigned long long)nxt_chunk));  // the next grab is in flight during this chunk
    const int hi = min(w + my_chunk, n_items);
    for (int it = w; it < hi; it++) {
      const int item = it;
      if (item < cur_first || item >= cur_end) {
        // another problem: hand in what was gathered for the previous one, then look the new one up
        if (cur_b >= 0 && cur_active && ndt_flush(S, accum[cur_b], cur_tiles, tile_base[cur_b + 1] - tile_base[cur_b]))
          ndt_finish_problem(S, accum[cur_b], states[cur_b], angles[cur_b]);
        int b0 = 0, b1 = B;  // tile_base[b0] <= item < tile_base[b1]
        while (b1 - b0 > 1) {
          const int mid = (b0 + b1) >> 1;
          if (tile_base[mid] <= item) b0 = mid;
          else b1 = mid;
        }
        cur_b = b0, cur_first = tile_base[b0], cur_end = tile_base[b0 + 1], cur_tiles = 0;
        cur_active = states[cur_b].phase == NDT_DERIV ? 1 : 0;
        if (cur_active) {
          d = descs[cur_b];
          cur_n = sorted ? d.meta->nvalid : d.n_input;
          __syncthreads();
          for (int k = threadIdx.x; k < (int)(sizeof(NdtAngles) / 4); k += kBlock) {  // j[8][3] | h[15][3] | T[12] -> NdtAnglesLds
            const float v = reinterpret_cast<const float*>(&angles[cur_b])[k];
            if (k < 69) S.ang.tab[k % 3][ndt_angle_slot(k / 3)] = v;
            else S.ang.T[k - 69] = v;
          }
          if (threadIdx.x < 3) S.ang.tab[threadIdx.x][23] = 0.f;
          __syncthreads();
        }
      }
      if (!cur_active) {  // finished in an earlier round: its first item carries the round's progress tick
        if (item == cur_first && threadIdx.x == 0 && !debug) progress_tick(prog, false);
        continue;
      }
      if (cur_tiles >= kNdtFlushTiles) {  // keeps the LDS slots' integer sums in range (a long run of one problem's tiles: one cloud on few blocks)
        ndt_flush(S, accum[cur_b], cur_tiles, tile_base[cur_b + 1] - tile_base[cur_b]);  // cannot complete the problem: this tile is still to come
        cur_tiles = 0;
      }
      cur_tiles++;
      const int tile = item - cur_first;
      if (tile * kBlock >= cur_n && !(tile == 0)) continue;  // the host's tile count is an upper bound (non-finite points); tile 0 always runs
      unsigned ang_off = 0;
      HGS_OPAQUE_OFFSET(ang_off);  // re-read the tables from LDS every tile: hoisted out of the loop they would pin 81 VGPRs
      const NdtAnglesLds* ap = reinterpret_cast<const NdtAnglesLds*>(reinterpret_cast<const char*>(&S.ang) + ang_off);
      if (!staged) {  // first tile of a run: the three dependent loads one after the other
        ndt_front_load(fx, fhave, d, sorted, tile * kBlock + (int)threadIdx.x, cur_n);
        ndt_front_issue_probes<NOFF>(fx, fhave, fxt, fkv, ap, tgt, box, c.search, sorted);
        ndt_front_resolve<NOFF>(fxt, fhave, fkv, fci, fvmask, tgt, box, c.search);
#pragma unroll
        for (int o = 0; o < NOFF; o++) ci_col[o * kBlock] = fci[o];
      }
      // the next item is the next tile of the same problem: its point, probes and cell indices are fetched under this tile's reduction
      const bool stage_next = kStaged && it + 1 < hi && item + 1 < cur_end && (tile + 1) * kBlock < cur_n;
      double acc[kAccNdt];
#pragma unroll
      for (int k = 0; k < kAccNdt; k++) acc[k] = 0.0;
      const F3 xt = fxt;
      unsigned vmask = fvmask;
      const bool any_cell = __ballot(vmask != 0u) != 0ull;
      if (any_cell) {
        // Visit the point's valid cells in neighbourhood order, the record of visit k+1 in flight during the arithmetic of
        // visit k.  A counted loop with a wave-uniform early exit: the `while (any lane has a cell)` form of the same loop
        // made the register allocator keep two copies of the 43 double sums (368 VGPRs instead of 216).
        int cur = ndt_pop_cell_lds<NOFF>(vmask, ci_col);
        NdtCellRec rc = tgt.cells[cur >= 0 ? cur : 0];
        NdtPointDeriv pd;
        ndt_point_derivatives_lds(*ap, fx.x, fx.y, fx.z, pd);
#pragma unroll 1
        for (int v = 0; v < NOFF; v++) {
          if (__ballot(cur >= 0) == 0ull) break;
          const int nx = ndt_pop_cell_lds<NOFF>(vmask, ci_col);
          NdtCellRec rn = rc;
          if (nx >= 0) rn = tgt.cells[nx];
          if (cur >= 0 && ndt_cell_in_reach(c, xt, rc.mean)) {
            const float icov[6] = {rc.v0.x, rc.v0.y, rc.v0.z, rc.v0.w, rc.v1.x, rc.v1.y};
            ndt_cell_terms_pk(d1, d2, pd, (float)((double)xt.x - rc.mean[0]), (float)((double)xt.y - rc.mean[1]), (float)((double)xt.z - rc.mean[2]), icov, acc);
          }
          cur = nx;
          rc = rn;
        }
      }
      // ---- exact accumulation of the tile's 43 sums (ndt_reduce_sums), in two pieces with the next tile's loads between them.
      // A wave none of whose points met a cell has nothing but zeros to add and skips the arithmetic.
      bool bad = false;
      // (the cell loop's last record prefetch may still be in flight into registers the reduction reuses: waited for here, in front of
      // the staged load, instead of where the compiler would put the wait — behind it, which made the load synchronous)
      HGS_WAIT_VMEM_TRACKED();
      if (stage_next) ndt_front_load(fx, fhave, d, sorted, (tile + 1) * kBlock + (int)threadIdx.x, cur_n);
      if (any_cell) {
        if (lane == 0) atomicAdd(&S.adds, 1u);
        bad = ndt_reduce_sums<0, 20>(acc, S.slots, lane);
      }
      if (stage_next) ndt_front_issue_probes<NOFF>(fx, fhave, fxt, fkv, ap, tgt, box, c.search, sorted);
      if (any_cell) {
        if (ndt_reduce_sums<20, kAccNdt>(acc, S.slots, lane)) bad = true;
        if (__ballot(bad) != 0ull && lane == 0) S.out_of_range = 1;
      }
      if (stage_next) {
        ndt_front_resolve<NOFF>(fxt, fhave, fkv, fci, fvmask, tgt, box, c.search);
#pragma unroll
        for (int o = 0; o < NOFF; o++) ci_col[o * kBlock] = fci[o];  // (tile t's cell loop has ended: one buffer is enough)
      }
      staged = stage_next;
    }
    __syncthreads();  // everybody has read S.next_chunk
    if (threadIdx.x == 0) S.next = nxt, S.next_chunk = nxt_chunk;
    __syncthreads();
    w = S.next;
  }
  if (cur_b >= 0 && cur_active && ndt_flush(S, accum[cur_b], cur_tiles, tile_base[cur_b + 1] - tile_base[cur_b]))
    ndt_finish_problem(S, accum[cur_b], states[cur_b], angles[cur_b]);
}
void launch_ndt_pass(hipStream_t s, const CloudDesc* descs, NdtTargetView tgt, NdtState* states, NdtAngles* angles, NdtConsts c, NdtAccum* accum, const int* tile_base,
                     unsigned long long* queues, int B, int parity, int blocks, int chunk, int sorted, int debug, Progress prog) {
  const dim3 grid(blocks < 1 ? 1 : blocks), block(kBlock);
  if (chunk < 1) chunk = 1;
  parity &= 1;
  if (c.search == 1) hipLaunchKernelGGL(k_ndt_pass<1>, grid, block, 0, s, descs, tgt, states, angles, c, accum, tile_base, queues, B, parity, chunk, sorted, debug, prog);
  else if (c.search == 2) hipLaunchKernelGGL(k_ndt_pass<7>, grid, block, 0, s, descs, tgt, states, angles, c, accum, tile_base, queues, B, parity, chunk, sorted, debug, prog);
  else hipLaunchKernelGGL(k_ndt_pass<27>, grid, block, 0, s, descs, tgt, states, angles, c, accum, tile_base, queues, B, parity, chunk, sorted, debug, prog);
}

__global__ void k_ndt_results(const CloudDesc* descs, const NdtState* states, DevResult* out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const NdtState& st = states[b];
  DevResult r;
  pose_to_colmajor_f(st.final_T, r.T);
  r.converged = st.converged, r.iterations = st.iterations, r.lm_tries = st.passes, r.pad = 0;
  const int n = descs[b].n_input;  // trans_probability_ = score / input_->points.size(): non-finite points included
  r.error = n > 0 ? st.score / (double)n : 0.0;
  r.fit_sum = 0, r.fit_count = 0, r.pad2 = 0;
  out[b] = r;
}
void launch_ndt_results(hipStream_t s, const CloudDesc* descs, const NdtState* states, DevResult* out, int B) {
  hipLaunchKernelGGL(k_ndt_results, dim3((B + 63) / 64), dim3(64), 0, s, descs, states, out, B);
}

// ------------------------------------------------------------------------------------------------ VGICP (fast_gicp::FastVGICP)
// Target voxelisation (GaussianVoxelMap, ADDITIVE): voxel = {n, mean of the points, mean of their GICP covariances}.
// Same sort-by-cell + segment-head machinery as the NDT target; the points are taken in Hilbert order because the
// covariances live there (the per-voxel sums are re-associated relative to upstream's input order, nothing else).
__global__ void k_vgicp_grid_params(CloudDesc d, double resolution) {
  CloudMeta* m = d.meta;
  m->vg_ncells = 0;
  m->vg_error = 0;
  if (m->nvalid <= 0) {
    for (int k = 0; k < 3; k++) m->vg_min_b[k] = 0, m->vg_max_b[k] = -1, m->vg_div_mul[k] = 0;
    return;
  }
  long long div[3];
  for (int k = 0; k < 3; k++) {
    m->vg_min_b[k] = vgicp_coord((double)ord2f(m->bbmin[k]), resolution);
    m->vg_max_b[k] = vgicp_coord((double)ord2f(m->bbmax[k]), resolution);
    div[k] = (long long)m->vg_max_b[k] - m->vg_min_b[k] + 1;
  }
  if (div[0] * div[1] * div[2] > 2147483647LL) m->vg_error = 1;
  m->vg_div_mul[0] = 1, m->vg_div_mul[1] = (int)div[0], m->vg_div_mul[2] = (int)(div[0] * div[1]);
}
void launch_vgicp_grid_params(hipStream_t s, CloudDesc desc, double resolution) {
  hipLaunchKernelGGL(k_vgicp_grid_params, dim3(1), dim3(1), 0, s, desc, resolution);
}

__global__ __launch_bounds__(kBlock) void k_vgicp_cell_keys(CloudDesc d, double resolution, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= d.n_input) return;
  unsigned long long key = 0xffffffffull;
  const CloudMeta* m = d.meta;
  if (i < m->nvalid && !m->vg_error) {
    const float4 p = d.pts[i];
    const int cx = vgicp_coord((double)p.x, resolution) - m->vg_min_b[0];
    const int cy = vgicp_coord((double)p.y, resolution) - m->vg_min_b[1];
    const int cz = vgicp_coord((double)p.z, resolution) - m->vg_min_b[2];
    key = (unsigned long long)(unsigned)(cx * m->vg_div_mul[0] + cy * m->vg_div_mul[1] + cz * m->vg_div_mul[2]);
  }
  keys[i] = key;
  vals[i] = (unsigned)i;
}
void launch_vgicp_cell_keys(hipStream_t s, CloudDesc desc, double resolution, unsigned long long* keys, unsigned* vals) {
  if (desc.n_input <= 0) return;
  hipLaunchKernelGGL(k_vgicp_cell_keys, dim3((desc.n_input + kBlock - 1) / kBlock), dim3(kBlock), 0, s, desc, resolution, keys, vals);
}

__global__ __launch_bounds__(kBlock) void k_vgicp_build_cells(CloudDesc d, const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                              int* hash_keys, int* hash_vals, int hash_mask, NdtCellRec* cells) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= d.n_input) return;
  const unsigned long long key = keys[i];
  if (key == 0xffffffffull) return;
  if (i > 0 && keys[i - 1] == key) return;
  double sum[3] = {0, 0, 0};
  Sym3 sc = {0, 0, 0, 0, 0, 0};
  int n = 0;
  for (int j = i; j < d.n_input && keys[j] == key; j++) {
    const int pi = (int)vals[j];
    const float4 p = d.pts[pi];
    const float4 c0 = d.cov[2 * pi], c1 = d.cov[2 * pi + 1];
    sum[0] += (double)p.x, sum[1] += (double)p.y, sum[2] += (double)p.z;
    sc.xx += (double)c0.x, sc.xy += (double)c0.y, sc.xz += (double)c0.z, sc.yy += (double)c0.w, sc.yz += (double)c1.x, sc.zz += (double)c1.y;
    n++;
  }
  NdtCellRec rec;
  vgicp_finalize_voxel(n, sum, sc, (int)key, &rec);
  const int slot_c = atomicAdd(&d.meta->vg_ncells, 1);
  cells[slot_c] = rec;
  unsigned slot = (ndt_hash((int)key) >> 7) & (unsigned)hash_mask;
  for (;;) {
    const int prev = atomicCAS(&hash_keys[slot], -1, (int)key);
    if (prev == -1) {
      hash_vals[slot] = slot_c;
      break;
    }
    slot = (slot + 1) & (unsigned)hash_mask;
  }
}
void launch_vgicp_build_cells(hipStream_t s, CloudDesc desc, const unsigned long long* sorted_keys, const unsigned* sorted_vals, int* hash_keys,
                              int* hash_vals, int hash_mask, NdtCellRec* cells) {
  if (desc.n_input <= 0) return;
  hipLaunchKernelGGL(k_vgicp_build_cells, dim3((desc.n_input + kBlock - 1) / kBlock), dim3(kBlock), 0, s, desc, sorted_keys, sorted_vals, hash_keys,
                     hash_vals, hash_mask, cells);
}

__device__ __forceinline__ NdtGrid vgicp_grid_of(const NdtTargetView& tgt) {
  NdtGrid g;
  g.hash_keys = tgt.hash_keys, g.hash_vals = tgt.hash_vals, g.cells = tgt.cells, g.hash_mask = tgt.hash_mask, g.inv_leaf = 0.f;
  for (int k = 0; k < 3; k++) g.min_b[k] = tgt.meta->vg_min_b[k], g.max_b[k] = tgt.meta->vg_max_b[k], g.div_mul[k] = tgt.meta->vg_div_mul[k];
  return g;
}

// update_correspondences + linearize of FastVGICP fused.  Algorithmic bytes per source point (DIRECT1):
// 16 (a_i) + 24 (C_A) + 40 (voxel: mean 12, cov 24, n 4) = 80; the voxel table of a LiDAR scan is L2-resident.
__global__ __launch_bounds__(kBlock) void k_vgicp_linearize(const CloudDesc* descs, NdtTargetView tgt, const GicpState* states, VgicpConsts c,
                                                            double* __restrict__ partials, int max_blocks) {
  const int b = blockIdx.y;
  if (states[b].phase != GICP_LINEARIZE) return;
  const CloudDesc d = descs[b];
  const int n = d.meta->nvalid;
  const int ntiles = (n + kBlock - 1) / kBlock;
  if ((int)blockIdx.x >= ntiles) return;
  const int tile = (int)blockIdx.x;
  const int i = tile * kBlock + threadIdx.x;
  __shared__ double lds[4 * kAcc];
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; k++) acc[k] = 0.0;
  if (i < n) {
    const Pose T = states[b].x0;
    const NdtGrid g = vgicp_grid_of(tgt);
    const float4 a = d.pts[i];
    int hits;
    acc[27] = vgicp_point_terms<true>(g, c, T, T, load_cov(d.cov, i), a.x, a.y, a.z, acc, &hits);
    d.corr[i] = hits;
  }
  block_reduce_store<kAcc>(acc, partials + ((size_t)b * max_blocks + tile) * kAcc, lds);
}
void launch_vgicp_linearize(hipStream_t s, const CloudDesc* descs, NdtTargetView tgt, const GicpState* states, VgicpConsts c, double* partials,
                            int max_blocks, int B) {
  hipLaunchKernelGGL(k_vgicp_linearize, dim3(max_blocks, B), dim3(kBlock), 0, s, descs, tgt, states, c, partials, max_blocks);
}

// compute_error(xi): voxels and Mahalanobis matrices of the linearisation pose x0, residuals at the LM trial pose xi.
__global__ __launch_bounds__(kBlock) void k_vgicp_error(const CloudDesc* descs, NdtTargetView tgt, const GicpState* states, VgicpConsts c,
                                                        double* __restrict__ partials_err, int max_blocks) {
  const int b = blockIdx.y;
  if (states[b].phase != GICP_TRY) return;
  const CloudDesc d = descs[b];
  const int n = d.meta->nvalid;
  const int ntiles = (n + kBlock - 1) / kBlock;
  if ((int)blockIdx.x >= ntiles) return;
  const int tile = (int)blockIdx.x;
  const int i = tile * kBlock + threadIdx.x;
  __shared__ double lds[4];
  double err = 0.0;
  if (i < n && d.corr[i] > 0) {
    const NdtGrid g = vgicp_grid_of(tgt);
    const float4 a = d.pts[i];
    err = vgicp_point_terms<false>(g, c, states[b].x0, states[b].xi, load_cov(d.cov, i), a.x, a.y, a.z, nullptr, nullptr);
  }
  block_reduce_store<1>(&err, partials_err + (size_t)b * max_blocks + tile, lds);
}
void launch_vgicp_error(hipStream_t s, const CloudDesc* descs, NdtTargetView tgt, const GicpState* states, VgicpConsts c, double* partials_err,
                        int max_blocks, int B) {
  hipLaunchKernelGGL(k_vgicp_error, dim3(max_blocks, B), dim3(kBlock), 0, s, descs, tgt, states, c, partials_err, max_blocks);
}

// ------------------------------------------------------------------------------------------------ prefilter (next row f2)
// PrefilteringNodelet::cloud_callback, apps/prefiltering_nodelet.cpp:131-133: distance_filter -> downsample
// (pcl::VoxelGrid) -> outlier_removal (pcl::RadiusOutlierRemoval / pcl::StatisticalOutlierRemoval), on the device.
// Working format: float4 {x, y, z, intensity}.  Every stage writes keep-flags; a rocPRIM exclusive scan turns them into
// output slots (order preserving, hence deterministic); the voxel grid reuses the sort-by-cell + segment-head scheme
// of the NDT / VGICP targets with FLOAT centroids accumulated in input order (CentroidPoint semantics).
// deskew != 0: the deskewing step of cloud_callback (:112, :182-243) on the way in — point i of the sweep rotated back by the
// first-order rotation of delta_t = scan_period * i / n at the gyro rate w (pf_deskew_point, hgs_math.h).
// (round 6) thread 0 also sets the pipeline's point count and the initial voxel-grid record {bbox min = max uint, bbox max = 0, ...} — two copy
// kernels less per sweep
__global__ __launch_bounds__(kBlock) void k_pf_load(const float4* __restrict__ staged, int n, float4* __restrict__ out, int deskew, float wx, float wy, float wz, double scan_period,
                                                    int* __restrict__ count_out, unsigned* __restrict__ meta_out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i == 0) {
    *count_out = n;
#pragma unroll
    for (int k = 0; k < 16; k++) meta_out[k] = k < 3 ? 0xffffffffu : 0u;
  }
  if (i >= n) return;
  const float4 p = staged[i];  // {x, y, z, intensity}, packed by the host (upload_points_packed)
  float x = p.x, y = p.y, z = p.z;
  if (deskew) pf_deskew_point(wx, wy, wz, scan_period, i, n, &x, &y, &z);
  out[i] = make_float4(x, y, z, p.w);
}
void launch_pf_load(hipStream_t s, const float4* staged, int n, float4* out, const float* deskew_w, double scan_period, int* count_out, unsigned* meta_out) {
  hipLaunchKernelGGL(k_pf_load, dim3(std::max(1, (n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, staged, n, out, deskew_w ? 1 : 0, deskew_w ? deskew_w[0] : 0.f,
                     deskew_w ? deskew_w[1] : 0.f, deskew_w ? deskew_w[2] : 0.f, scan_period, count_out, meta_out);
}

// keep[i] = near < |p| < far  (float norm against double thresholds, :170-173); use_filter == 0 keeps everything
__device__ __forceinline__ bool pf_distance_keeps(const float4& p, double near_thresh, double far_thresh) {
  HGS_FP_STRICT
  const float n2 = p.x * p.x + p.y * p.y + p.z * p.z;
  const double d = (double)sqrtf(n2);
  return d > near_thresh && d < far_thresh;
}
__global__ __launch_bounds__(kBlock) void k_pf_distance_flags(const float4* __restrict__ pts, int n, int use_filter, double near_thresh, double far_thresh,
                                                              unsigned* __restrict__ keep) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  keep[i] = (!use_filter || pf_distance_keeps(p, near_thresh, far_thresh)) ? 1u : 0u;
}
void launch_pf_distance_flags(hipStream_t s, const float4* pts, int n, int use_filter, double near_thresh, double far_thresh, unsigned* keep) {
  if (n > 0) hipLaunchKernelGGL(k_pf_distance_flags, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, n, use_filter, near_thresh, far_thresh, keep);
}

// out[slot[i]] = in[i] for kept points; *count = number kept (slot = exclusive scan of keep)
__global__ __launch_bounds__(kBlock) void k_pf_compact(const float4* __restrict__ in, int n, const unsigned* __restrict__ keep, const unsigned* __restrict__ slot,
                                                       float4* __restrict__ out, int* __restrict__ count) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (keep[i]) out[slot[i]] = in[i];
  if (i == n - 1) *count = (int)(slot[i] + keep[i]);
}
void launch_pf_compact(hipStream_t s, const float4* in, int n, const unsigned* keep, const unsigned* slot, float4* out, int* count) {
  if (n > 0) hipLaunchKernelGGL(k_pf_compact, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, in, n, keep, slot, out, count);
}

// pcl::VoxelGrid pass 1: bounding box of the finite points (getMinMax3D) into meta[0..5] (ordered-uint encoding)
// (round 6) dist_filter: the distance filter of :138-156 applied HERE and in k_pf_voxel_keys instead of by flags + scan + compaction in front (four launches
// less): a point the filter drops takes part in neither the bounding box nor a voxel — the voxel grid sees exactly the cloud the compaction would have handed it
__global__ __launch_bounds__(kBlock) void k_pf_bbox(const float4* __restrict__ pts, const int* __restrict__ count, unsigned* __restrict__ meta, int dist_filter, double near_thresh,
                                                    double far_thresh) {
  const int n = *count;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  // as in k_bbox_count: a thread's points loaded together, ONE set of atomics per block (their serialisation on the six words was this kernel's time)
  for (int i0 = blockIdx.x * kBlock * kBboxPerThread + threadIdx.x; i0 < n; i0 += gridDim.x * kBlock * kBboxPerThread) {
    float4 p[kBboxPerThread];
#pragma unroll
    for (int k = 0; k < kBboxPerThread; k++) {
      const int i = i0 + k * kBlock;
      p[k] = i < n ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
    }
#pragma unroll
    for (int k = 0; k < kBboxPerThread; k++)
      if (finite3(p[k]) && (!dist_filter || pf_distance_keeps(p[k], near_thresh, far_thresh))) {
        mn[0] = fminf(mn[0], p[k].x), mn[1] = fminf(mn[1], p[k].y), mn[2] = fminf(mn[2], p[k].z);
        mx[0] = fmaxf(mx[0], p[k].x), mx[1] = fmaxf(mx[1], p[k].y), mx[2] = fmaxf(mx[2], p[k].z);
      }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      mn[k] = fminf(mn[k], __shfl_down(mn[k], off, 64));
      mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off, 64));
    }
  __shared__ float s_mn[kBlock / 64][3], s_mx[kBlock / 64][3];
  const int wave = (int)(threadIdx.x >> 6);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; k++) s_mn[wave][k] = mn[k], s_mx[wave][k] = mx[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; w++)
      for (int k = 0; k < 3; k++) mn[k] = fminf(mn[k], s_mn[w][k]), mx[k] = fmaxf(mx[k], s_mx[w][k]);
    if (mn[0] <= mx[0])
      for (int k = 0; k < 3; k++) {
        atomicMin(&meta[k], f2ord(mn[k]));
        atomicMax(&meta[3 + k], f2ord(mx[k]));
      }
  }
}
// meta: [0..2] bbmin, [3..5] bbmax (ordered uints) -> [6..8] min_b, [9..11] div_mul, [12] error (index overflow)
__global__ void k_pf_grid(unsigned* meta, float inv_leaf) {
  int* im = reinterpret_cast<int*>(meta);
  if (meta[0] == 0xffffffffu) {  // no finite point
    for (int k = 6; k < 13; k++) im[k] = 0;
    return;
  }
  long long div[3];
  for (int k = 0; k < 3; k++) {
    im[6 + k] = (int)floorf(ord2f(meta[k]) * inv_leaf);
    div[k] = (long long)(int)floorf(ord2f(meta[3 + k]) * inv_leaf) - im[6 + k] + 1;
  }
  im[12] = div[0] * div[1] * div[2] > 2147483647LL ? 1 : 0;
  im[9] = 1, im[10] = (int)div[0], im[11] = (int)(div[0] * div[1]);
}
__global__ __launch_bounds__(kBlock) void k_pf_voxel_keys(const float4* __restrict__ pts, const int* __restrict__ count, const unsigned* __restrict__ meta, float inv_leaf,
                                                          int cap, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals, int dist_filter, double near_thresh,
                                                          double far_thresh) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  const int* im = reinterpret_cast<const int*>(meta);
  unsigned long long key = 0xffffffffull;
  if (i < *count && !im[12]) {
    const float4 p = pts[i];
    if (finite3(p) && (!dist_filter || pf_distance_keeps(p, near_thresh, far_thresh))) {
      const int ix = (int)floorf(p.x * inv_leaf) - im[6], iy = (int)floorf(p.y * inv_leaf) - im[7], iz = (int)floorf(p.z * inv_leaf) - im[8];
      key = (unsigned long long)(unsigned)(ix * im[9] + iy * im[10] + iz * im[11]);
    }
  }
  keys[i] = key;
  vals[i] = (unsigned)i;
}
// head flags of the sorted key runs (keep) — the scan of these gives each voxel its output slot
__global__ __launch_bounds__(kBlock) void k_pf_voxel_heads(const unsigned long long* __restrict__ keys, int cap, unsigned* __restrict__ head,
                                                           unsigned long long invalid_key) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  const unsigned long long k = keys[i];
  head[i] = (k != invalid_key && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}
// one thread per voxel: float centroid of x, y, z, intensity over the run in (stable) input order
__global__ __launch_bounds__(kBlock) void k_pf_voxel_centroids(const float4* __restrict__ pts, const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                               const unsigned* __restrict__ head, const unsigned* __restrict__ slot, int cap,
                                                               float4* __restrict__ out, int* __restrict__ count_out, unsigned* __restrict__ ukeys /* may be null */) {
  HGS_FP_STRICT
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  if (i == cap - 1) *count_out = (int)(slot[i] + head[i]);
  if (!head[i]) return;
  const unsigned long long key = keys[i];
  if (ukeys) ukeys[slot[i]] = (unsigned)key;  // ascending: the lookup table of k_pf_grid_radius_flags
  float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
  int n = 0;
  for (int j = i; j < cap && keys[j] == key; j++) {
    const float4 p = pts[vals[j]];
    sx += p.x, sy += p.y, sz += p.z, si += p.w;
    n++;
  }
  const float fn = (float)n;
  out[slot[i]] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
}
void launch_pf_bbox(hipStream_t s, const float4* pts, const int* count, int cap, unsigned* meta, int dist_filter, double near_thresh, double far_thresh) {
  int gx = (cap + kBlock * kBboxPerThread - 1) / (kBlock * kBboxPerThread);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_pf_bbox, dim3(gx), dim3(kBlock), 0, s, pts, count, meta, dist_filter, near_thresh, far_thresh);
}
void launch_pf_grid(hipStream_t s, unsigned* meta, float inv_leaf) { hipLaunchKernelGGL(k_pf_grid, dim3(1), dim3(1), 0, s, meta, inv_leaf); }
void launch_pf_voxel_keys(hipStream_t s, const float4* pts, const int* count, const unsigned* meta, float inv_leaf, int cap, unsigned long long* keys,
                          unsigned* vals, int dist_filter, double near_thresh, double far_thresh) {
  if (cap > 0)
    hipLaunchKernelGGL(k_pf_voxel_keys, dim3((cap + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, count, meta, inv_leaf, cap, keys, vals, dist_filter, near_thresh, far_thresh);
}
void launch_pf_voxel_heads(hipStream_t s, const unsigned long long* keys, int cap, unsigned* head, unsigned long long invalid_key) {
  if (cap > 0) hipLaunchKernelGGL(k_pf_voxel_heads, dim3((cap + kBlock - 1) / kBlock), dim3(kBlock), 0, s, keys, cap, head, invalid_key);
}
void launch_pf_voxel_centroids(hipStream_t s, const float4* pts, const unsigned long long* keys, const unsigned* vals, const unsigned* head, const unsigned* slot,
                               int cap, float4* out, int* count_out, unsigned* ukeys) {
  if (cap > 0)
    hipLaunchKernelGGL(k_pf_voxel_centroids, dim3((cap + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, keys, vals, head, slot, cap, out, count_out, ukeys);
}

// pcl::RadiusOutlierRemoval behind pcl::VoxelGrid WITHOUT a search tree (round 6): the cloud is one centroid per occupied voxel, in ascending voxel-key order
// (`cen[j]`, `ukeys[j]`), so the points within r of a centroid sit in the voxels its r-box overlaps — a few rows of consecutive keys, each found by one
// binary search in ukeys.  Same predicate as k_pf_radius_flags (keep iff more than min_neighbors points, the point itself included, lie strictly within
// the radius; the same float distance), same output order; what it saves is building a resident cloud + search index just to throw both away, and the host
// read-back of the voxel count that sizing them needs (hgs_engine.hip, prefilter_impl).  A centroid may round a hair outside its voxel: the box is taken
// 1e-3 voxel widths larger.  G lanes per centroid (below); the groups beyond *count clear their flag.
// (Round 6 also measured k_pf_voxel_centroids reading its run four entries at a time: 28.8 vs 29.1 us — its time is the longest run's ordered float sum, a
// near-sensor voxel with a few hundred returns, not the loads; and an L2 warm-up of small launches' trees — every block touching the whole tree once before its
// walks — for k_knn_cov / k_gicp_linearize: k_knn_cov unchanged, config 2's linearize 0.35 -> 0.50 ms; both removed.  profiles/r06_ab6_prefilter_warm.log)
template <int G>
__global__ __launch_bounds__(kBlock) void k_pf_grid_radius_flags(const float4* __restrict__ cen, const int* __restrict__ count, const unsigned* __restrict__ ukeys,
                                                                 const unsigned* __restrict__ meta, float inv_leaf, float radius, float r2, int min_neighbors, int cap,
                                                                 unsigned* __restrict__ keep) {
  // G adjacent lanes per centroid, lane g on the z-layer lo[2] + g of the r-box (G >= the box's layers: launch_pf_grid_radius_flags): a layer starts with ONE
  // bisection of the key table, its rows follow each other closely in it.  One thread per centroid walked all (2 ceil(r / leaf) + 2)^2 rows itself — a chain of
  // ~120 dependent loads for a centroid without neighbours, and nearly every wave holds one: 52 us of a 0.36 ms prefilter.  The G partial counts are summed
  // across the group (an integer count against a threshold: nothing about the predicate changes).
  static_assert(G == 4 || G == 8 || G == 16, "a power of two that divides the wave");
  const int t = blockIdx.x * kBlock + threadIdx.x;
  const int j = t / G, g = t & (G - 1);
  const int m = *count;
  int cnt = 0;
  if (j < m) {
    const int* im = reinterpret_cast<const int*>(meta);
    const float4 p = cen[j];
    const F3 q = {p.x, p.y, p.z};
    const int nx = im[10], ny = im[10] > 0 ? im[11] / im[10] : 0;
    int lo[3], hi[3];
    const float c[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = (int)floorf((c[a] - radius) * inv_leaf - 1.0e-3f) - im[6 + a];
      hi[a] = (int)floorf((c[a] + radius) * inv_leaf + 1.0e-3f) - im[6 + a];
    }
    lo[0] = max(lo[0], 0), lo[1] = max(lo[1], 0), lo[2] = max(lo[2], 0);
    hi[0] = min(hi[0], nx - 1), hi[1] = min(hi[1], ny - 1);  // (z: rows beyond the grid simply hold no key)
    int a = 0;  // the rows of a layer are visited in ascending key order: every lower bound is searched from the previous one (gallop, then bisect)
    // (a box of more than G layers — the host never launches one — would be walked by the group's last lane)
    const int iz0 = lo[2] + g, iz1 = g == G - 1 ? hi[2] : min(iz0, hi[2]);
    for (int iz = iz0; iz <= iz1 && cnt <= min_neighbors; iz++)
      for (int iy = lo[1]; iy <= hi[1] && cnt <= min_neighbors; iy++) {
        const long long row = (long long)iy * im[10] + (long long)iz * im[11];
        const long long k0 = row + lo[0], k1 = row + hi[0];
        if (k1 < 0 || k0 > 0xfffffffell) continue;
        const unsigned key0 = (unsigned)max(k0, 0ll), key1 = (unsigned)min(k1, 0xfffffffell);
        int b = a, step = a == 0 ? m : 1;  // (the first row: a plain bisection of the whole table)
        while (b < m && ukeys[b] < key0) a = b + 1, b = min(b + step, m), step <<= 1;
        while (a < b) {  // lower bound of key0 in [a, b)
          const int mid = (a + b) >> 1;
          if (ukeys[mid] < key0) a = mid + 1;
          else b = mid;
        }
        for (; a < m && ukeys[a] <= key1 && cnt <= min_neighbors; a++) {
          const float4 o = cen[a];
          cnt += dist2f(q, o.x, o.y, o.z) < r2 ? 1 : 0;
        }
      }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);  // (lane g == 0 ends with its group's total)
  if (g == 0 && j < cap) keep[j] = (j < m && cnt > min_neighbors) ? 1u : 0u;
}
void launch_pf_grid_radius_flags(hipStream_t s, const float4* cen, const int* count, const unsigned* ukeys, const unsigned* meta, float inv_leaf, float radius, float r2,
                                 int min_neighbors, int cap, unsigned* keep) {
  if (cap <= 0) return;
  // layers of the r-box: floor((c + r) / leaf + 1e-3) - floor((c - r) / leaf - 1e-3) + 1 <= 2 r / leaf + 2 + 1 (the engine asks for r / leaf <= 4)
  const int layers = (int)ceilf(2.f * radius * inv_leaf + 2.0e-3f) + 2;
  const long long threads4 = (long long)cap * 4, threads8 = (long long)cap * 8, threads16 = (long long)cap * 16;
  if (layers <= 4)
    hipLaunchKernelGGL(k_pf_grid_radius_flags<4>, dim3((unsigned)((threads4 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, cen, count, ukeys, meta, inv_leaf, radius, r2, min_neighbors, cap, keep);
  else if (layers <= 8)
    hipLaunchKernelGGL(k_pf_grid_radius_flags<8>, dim3((unsigned)((threads8 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, cen, count, ukeys, meta, inv_leaf, radius, r2, min_neighbors, cap, keep);
  else
    hipLaunchKernelGGL(k_pf_grid_radius_flags<16>, dim3((unsigned)((threads16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, cen, count, ukeys, meta, inv_leaf, radius, r2, min_neighbors, cap, keep);
}

// ---- pcl::ApproximateVoxelGrid (apps/prefiltering_nodelet.cpp:59-63, scan_matching_odometry_nodelet.cpp:91-96) ------------------
// PCL walks the points in input order through a 512-entry history table (hash of the voxel coordinates); a point that finds
// another voxel in its bucket flushes that voxel's running centroid to the output.  Per bucket the subsequence of its points
// is therefore cut into RUNS of equal voxel, each run is one output, a run that is followed by another run of its bucket is
// emitted when that next run's first point arrives (so evictions come out in the order of those points' input indices), and
// the last run of every bucket is emitted at the end in bucket order.  Parallel form: stable sort by bucket, run heads, one
// scan over the evicting points' input indices for the output slots, one thread per run for the float centroid in sequence
// order — the same output, in the same order, as the sequential filter.
__device__ __forceinline__ void approx_voxel_of(const float4& p, float inv_leaf, int* ix, int* iy, int* iz, unsigned* hash) {
  *ix = (int)floorf(p.x * inv_leaf), *iy = (int)floorf(p.y * inv_leaf), *iz = (int)floorf(p.z * inv_leaf);
  *hash = ((unsigned)*ix * 7171u + (unsigned)*iy * 3079u + (unsigned)*iz * 4231u) & 511u;
}
constexpr unsigned long long kApproxInvalidBucket = 1023ull;
__global__ __launch_bounds__(kBlock) void k_pf_approx_keys(const float4* __restrict__ pts, const int* __restrict__ count, float inv_leaf, int cap,
                                                           unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  unsigned long long key = kApproxInvalidBucket;
  if (i < *count) {
    const float4 p = pts[i];
    if (finite3(p)) {
      int ix, iy, iz;
      unsigned h;
      approx_voxel_of(p, inv_leaf, &ix, &iy, &iz, &h);
      key = h;
    }
  }
  keys[i] = key;
  vals[i] = (unsigned)i;
}
// sorted position j: head[j] = 1 if a run starts there; evict[input index of that point] = 1 if the run is not the first of its
// bucket (its first point evicts the previous run); bucket_used[h] = 1 for every bucket that holds points
__global__ __launch_bounds__(kBlock) void k_pf_approx_heads(const float4* __restrict__ pts, const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                            float inv_leaf, int cap, unsigned* __restrict__ head, unsigned* __restrict__ evict,
                                                            unsigned* __restrict__ bucket_used) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= cap) return;
  const unsigned long long h = keys[j];
  unsigned is_head = 0;
  if (h != kApproxInvalidBucket) {
    const bool first_of_bucket = j == 0 || keys[j - 1] != h;
    int ix, iy, iz, jx, jy, jz;
    unsigned hh;
    approx_voxel_of(pts[vals[j]], inv_leaf, &ix, &iy, &iz, &hh);
    bool other_voxel = false;
    if (!first_of_bucket) {
      approx_voxel_of(pts[vals[j - 1]], inv_leaf, &jx, &jy, &jz, &hh);
      other_voxel = ix != jx || iy != jy || iz != jz;
    }
    is_head = (first_of_bucket || other_voxel) ? 1u : 0u;
    if (other_voxel) evict[vals[j]] = 1u;
    if (first_of_bucket) bucket_used[h] = 1u;
  }
  head[j] = is_head;
}
// bucket_rank[h] = number of used buckets below h (512 entries: one block), total in bucket_rank[512]
__global__ __launch_bounds__(512) void k_pf_approx_bucket_ranks(const unsigned* __restrict__ bucket_used, unsigned* __restrict__ bucket_rank) {
  __shared__ unsigned s[512];
  const int t = threadIdx.x;
  s[t] = bucket_used[t];
  __syncthreads();
  unsigned r = 0;
  for (int k = 0; k < t; k++) r += s[k];
  bucket_rank[t] = r;
  if (t == 511) bucket_rank[512] = r + s[511];
}
// one thread per run: float sums in sequence order, then the slot: the eviction rank of the point that ends the run, or behind
// all evictions in bucket order
__global__ __launch_bounds__(kBlock) void k_pf_approx_centroids(const float4* __restrict__ pts, const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                                const unsigned* __restrict__ head, const unsigned* __restrict__ evict, const unsigned* __restrict__ evict_rank,
                                                                const unsigned* __restrict__ bucket_rank, int cap, int n_points, float4* __restrict__ out,
                                                                int* __restrict__ count_out) {
  HGS_FP_STRICT
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= cap) return;
  const unsigned n_evictions = n_points > 0 ? evict_rank[n_points - 1] + evict[n_points - 1] : 0u;
  if (j == 0) *count_out = (int)(n_evictions + bucket_rank[512]);
  if (!head[j]) return;
  const unsigned long long h = keys[j];
  float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
  int n = 0, e = j;
  for (; e < cap && keys[e] == h && (e == j || !head[e]); e++) {
    const float4 p = pts[vals[e]];
    sx += p.x, sy += p.y, sz += p.z, si += p.w;
    n++;
  }
  const bool evicted = e < cap && keys[e] == h;  // another run of this bucket follows: its first point flushed this one
  const unsigned slot = evicted ? evict_rank[vals[e]] : n_evictions + bucket_rank[h];
  const float fn = (float)n;
  out[slot] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
}
void launch_pf_approx_keys(hipStream_t s, const float4* pts, const int* count, float inv_leaf, int cap, unsigned long long* keys, unsigned* vals) {
  if (cap > 0) hipLaunchKernelGGL(k_pf_approx_keys, dim3((cap + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, count, inv_leaf, cap, keys, vals);
}
void launch_pf_approx_heads(hipStream_t s, const float4* pts, const unsigned long long* keys, const unsigned* vals, float inv_leaf, int cap, unsigned* head, unsigned* evict,
                            unsigned* bucket_used, unsigned* bucket_rank) {
  if (cap <= 0) return;
  hipLaunchKernelGGL(k_pf_approx_heads, dim3((cap + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, keys, vals, inv_leaf, cap, head, evict, bucket_used);
  hipLaunchKernelGGL(k_pf_approx_bucket_ranks, dim3(1), dim3(512), 0, s, bucket_used, bucket_rank);
}
void launch_pf_approx_centroids(hipStream_t s, const float4* pts, const unsigned long long* keys, const unsigned* vals, const unsigned* head, const unsigned* evict,
                                const unsigned* evict_rank, const unsigned* bucket_rank, int cap, float4* out, int* count_out) {
  if (cap > 0)
    hipLaunchKernelGGL(k_pf_approx_centroids, dim3((cap + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, keys, vals, head, evict, evict_rank, bucket_rank, cap, cap, out,
                       count_out);
}

// pcl::RadiusOutlierRemoval (:85-93): keep p iff more than min_neighbors points (p included) lie strictly within the
// radius.  One thread per Hilbert-sorted point, packet walk on the cloud's own tree; the flag lands at the point's
// ORIGINAL index so that the compaction keeps the input order.
__global__ __launch_bounds__(kBlock) void k_pf_radius_flags(CloudDesc d, float r2, int min_neighbors, unsigned* __restrict__ keep) {
  const int n = d.meta->nvalid;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (blockIdx.x * kBlock >= n) return;
  const bool active = i < n;
  BvhView tv;
  tv.nodes = d.nodes, tv.pts = d.pts, tv.lpts = d.lpts, tv.P = d.P, tv.n = n;
  __shared__ __attribute__((aligned(128))) float walk_slots[kBlock / 64][32];
  const float4 qp = active ? d.pts[i] : make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
  PacketWalk<RadiusCountLane> w[1];
  w[0].lane.r2 = active ? r2 : -1.f, w[0].lane.cnt = 0, w[0].lane.need = min_neighbors + 1;
  w[0].start(tv, F3{qp.x, qp.y, qp.z}, 31 - __clz(tv.P));
  wave_walk_multi<RadiusCountLane, 1>(tv, w, walk_slots[threadIdx.x >> 6]);
  if (active) keep[__float_as_int(qp.w)] = w[0].lane.cnt > min_neighbors ? 1u : 0u;
}
void launch_pf_radius_flags(hipStream_t s, CloudDesc d, float r2, int min_neighbors, unsigned* keep) {
  if (d.n_input > 0) hipLaunchKernelGGL(k_pf_radius_flags, dim3((d.n_input + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d, r2, min_neighbors, keep);
}

// pcl::StatisticalOutlierRemoval (:73-84), pass 1: mean distance of every point to its mean_k nearest OTHER points
// (k+1 search, the nearest — the point itself — dropped), at the point's ORIGINAL index.
template <int KMAX>
__global__ __launch_bounds__(kBlock) void k_pf_mean_knn_dist(CloudDesc d, int mean_k, double* __restrict__ dist) {
  HGS_FP_STRICT
  const int n = d.meta->nvalid;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (blockIdx.x * kBlock >= n) return;
  const bool active = i < n;
  BvhView tv;
  tv.nodes = d.nodes, tv.pts = d.pts, tv.lpts = d.lpts, tv.P = d.P, tv.n = n;
  __shared__ __attribute__((aligned(128))) float walk_slots[kBlock / 64][32];
  const float4 qp = active ? d.pts[i] : make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
  const int live = mean_k + 1 < KMAX ? mean_k + 1 : KMAX;
  PacketWalk<KnnRadiusLane<KMAX>> w[1];
  w[0].lane.init(live, active);
  w[0].start(tv, F3{qp.x, qp.y, qp.z}, 31 - __clz(tv.P));
  wave_walk_multi<KnnRadiusLane<KMAX>, 1>(tv, w, walk_slots[threadIdx.x >> 6]);
  if (!active) return;
  double sum = 0.0;
  int found = 0;
#pragma unroll
  for (int j = KMAX - live; j < KMAX; j++) {
    const float dd = w[0].lane.d[j];
    if (dd < FLT_MAX) {
      if (found > 0) sum += sqrt((double)dd);  // ascending; the first (the point itself) is dropped
      found++;
    }
  }
  dist[__float_as_int(qp.w)] = found > 1 ? sum / (double)mean_k : 0.0;
}
void launch_pf_mean_knn_dist(hipStream_t s, CloudDesc d, int mean_k, double* dist) {
  if (d.n_input <= 0) return;
  const dim3 grid((d.n_input + kBlock - 1) / kBlock), block(kBlock);
  if (mean_k + 1 <= 16) hipLaunchKernelGGL(k_pf_mean_knn_dist<16>, grid, block, 0, s, d, mean_k, dist);
  else if (mean_k + 1 <= 32) hipLaunchKernelGGL(k_pf_mean_knn_dist<32>, grid, block, 0, s, d, mean_k, dist);
  else hipLaunchKernelGGL(k_pf_mean_knn_dist<64>, grid, block, 0, s, d, mean_k < 63 ? mean_k : 63, dist);
}
// pass 2: sum and sum of squares of dist[0..n) in a fixed order (one block: per-thread strided sums, tree over the block)
__global__ __launch_bounds__(kBlock) void k_pf_dist_stats(const double* __restrict__ dist, int n, double* __restrict__ out2) {
  __shared__ double s1[kBlock], s2[kBlock];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const double v = dist[i];
    a += v, b += v * v;
  }
  s1[threadIdx.x] = a, s2[threadIdx.x] = b;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s1[threadIdx.x] += s1[threadIdx.x + off], s2[threadIdx.x] += s2[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out2[0] = s1[0], out2[1] = s2[0];
}
// pass 3: keep[i] = dist[i] <= mean + mul * stddev (sample standard deviation)
__global__ __launch_bounds__(kBlock) void k_pf_statistical_flags(const double* __restrict__ dist, int n, const double* __restrict__ stats, double stddev_mul,
                                                                 unsigned* __restrict__ keep) {
  HGS_FP_STRICT
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double dn = (double)n;
  const double mean = stats[0] / dn;
  const double var = n > 1 ? (stats[1] - stats[0] * stats[0] / dn) / (dn - 1.0) : 0.0;
  const double thr = mean + stddev_mul * sqrt(var);
  keep[i] = dist[i] <= thr ? 1u : 0u;
}
void launch_pf_statistical(hipStream_t s, const double* dist, int n, double* stats, double stddev_mul, unsigned* keep) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_pf_dist_stats, dim3(1), dim3(kBlock), 0, s, dist, n, stats);
  hipLaunchKernelGGL(k_pf_statistical_flags, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, dist, n, stats, stddev_mul, keep);
}

// ------------------------------------------------------------------------------------------------ map cloud (next row f3)
// MapCloudGenerator::generate, src/hdl_graph_slam/map_cloud_generator.cpp:13-51: every keyframe cloud transformed by
// its (float) pose and concatenated; with a positive resolution, the centres of the occupied voxels of the
// pcl::octree::OctreePointCloud built from the points in order, in the tree's depth-first order, instead of the points themselves.
__global__ __launch_bounds__(kBlock) void k_map_transform(const MapSource* __restrict__ srcs, float4* __restrict__ out) {
  HGS_FP_STRICT
  const MapSource m = srcs[blockIdx.y];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m.n) return;
  const float4 p = m.raw[i];
  float4 o;
  o.x = ((m.T[0] * p.x + m.T[4] * p.y) + m.T[8] * p.z) + m.T[12];
  o.y = ((m.T[1] * p.x + m.T[5] * p.y) + m.T[9] * p.z) + m.T[13];
  o.z = ((m.T[2] * p.x + m.T[6] * p.y) + m.T[10] * p.z) + m.T[14];
  o.w = m.intensity[i];
  out[m.offset + i] = o;
}
void launch_map_transform(hipStream_t s, const MapSource* srcs, int nsrc, int max_n, float4* out) {
  if (nsrc > 0 && max_n > 0) hipLaunchKernelGGL(k_map_transform, dim3((max_n + kBlock - 1) / kBlock, nsrc), dim3(kBlock), 0, s, srcs, out);
}

__global__ __launch_bounds__(kBlock) void k_map_first_finite(const float4* __restrict__ pts, int n, MapOctree* __restrict__ oct) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool ok = i < n && finite3(pts[i]);
  const unsigned long long m = __ballot(ok);
  if (m != 0ull && (threadIdx.x & 63) == 0) atomicMin(&oct->first, (int)(blockIdx.x * kBlock + (threadIdx.x & ~63u)) + (__ffsll((long long)m) - 1));
}
// The octree's bounding box is replayed event by event (see MapOctree, hgs_device.h): the box after the first finite point, then
// repeatedly { the first later point outside the current box (a grid-wide minimum), the doublings it forces (one thread) } until
// no point is outside — at most one round per tree level.
__global__ void k_map_octree_init(const float4* __restrict__ pts, int n, double res, MapOctree* __restrict__ oct) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  oct->next_viol = 0x7fffffff, oct->overflow = 0, oct->n_events = 0, oct->done = 1;
  if (oct->first >= n) return;
  const float4 p = pts[oct->first];
  const float q[3] = {p.x, p.y, p.z};
  MapOctreeEvent& e = oct->events[0];
  e.index = oct->first, e.gained[0] = e.gained[1] = e.gained[2] = 0ull;
  octree_box_first(q, res, e.mn, oct->mx, &e.depth);
  oct->n_events = 1, oct->done = 0;
}
__global__ __launch_bounds__(kBlock) void k_map_octree_scan(const float4* __restrict__ pts, int n, MapOctree* __restrict__ oct) {
  if (oct->done) return;
  const MapOctreeEvent& e = oct->events[oct->n_events - 1];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  bool out = false;
  if (i < n && i > e.index) {
    const float4 p = pts[i];
    const float q[3] = {p.x, p.y, p.z};
    out = finite3(p) && octree_box_violated(q, e.mn, oct->mx);
  }
  const unsigned long long m = __ballot(out);
  if (m != 0ull && (threadIdx.x & 63) == 0) atomicMin(&oct->next_viol, (int)(blockIdx.x * kBlock + (threadIdx.x & ~63u)) + (__ffsll((long long)m) - 1));
}
__global__ void k_map_octree_apply(const float4* __restrict__ pts, double res, MapOctree* __restrict__ oct) {
  if (blockIdx.x != 0 || threadIdx.x != 0 || oct->done) return;
  if (oct->next_viol == 0x7fffffff) {
    oct->done = 1;
    return;
  }
  const float4 p = pts[oct->next_viol];
  const float q[3] = {p.x, p.y, p.z};
  MapOctreeEvent e = oct->events[oct->n_events - 1];
  e.index = oct->next_viol;
  while (octree_box_violated(q, e.mn, oct->mx)) {
    if (e.depth >= kMapMaxDepth) {
      oct->overflow = 1, oct->done = 1;
      return;
    }
    octree_box_double(q, res, e.mn, oct->mx, &e.depth, e.gained);
  }
  oct->events[oct->n_events++] = e;  // at least one doubling per event: n_events <= kMapMaxDepth + 1
  oct->next_viol = 0x7fffffff;
}
__global__ __launch_bounds__(kBlock) void k_map_keys(const float4* __restrict__ pts, int n, double res, const MapOctree* __restrict__ oct,
                                                     unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
  HGS_FP_STRICT
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  unsigned long long code = kMapInvalidKey;
  const int ne = oct->n_events;
  const float4 p = pts[i];
  if (ne > 0 && finite3(p)) {
    int ei = ne - 1;
    while (ei > 0 && oct->events[ei].index > i) ei--;  // the box this point was inserted into
    const MapOctreeEvent& e = oct->events[ei];
    const MapOctreeEvent& last = oct->events[ne - 1];
    const float q[3] = {p.x, p.y, p.z};
    unsigned long long key[3];
    for (int a = 0; a < 3; a++) key[a] = (unsigned long long)(unsigned)(((double)q[a] - e.mn[a]) / res) + (last.gained[a] - e.gained[a]);
    code = octree_interleave(key, last.depth);
  }
  keys[i] = code;
  vals[i] = (unsigned)i;
}
// one thread per occupied voxel (head of a sorted key run): its centre, genLeafNodeCenterFromOctreeKey with the final box
__global__ __launch_bounds__(kBlock) void k_map_centers(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ head, const unsigned* __restrict__ slot, int n,
                                                        double res, const MapOctree* __restrict__ oct, float4* __restrict__ out, int* __restrict__ count_out) {
  HGS_FP_STRICT
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (i == n - 1) *count_out = (int)(slot[i] + head[i]);
  if (!head[i]) return;
  const MapOctreeEvent& last = oct->events[oct->n_events - 1];
  unsigned long long key[3];
  octree_deinterleave(keys[i], last.depth, key);
  out[slot[i]] = make_float4((float)(((double)key[0] + 0.5) * res + last.mn[0]), (float)(((double)key[1] + 0.5) * res + last.mn[1]),
                             (float)(((double)key[2] + 0.5) * res + last.mn[2]), 0.f);
}
void launch_map_first_finite(hipStream_t s, const float4* pts, int n, MapOctree* oct) {
  if (n > 0) hipLaunchKernelGGL(k_map_first_finite, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, n, oct);
}
void launch_map_octree_init(hipStream_t s, const float4* pts, int n, double res, MapOctree* oct) {
  hipLaunchKernelGGL(k_map_octree_init, dim3(1), dim3(64), 0, s, pts, n, res, oct);
}
void launch_map_octree_step(hipStream_t s, const float4* pts, int n, double res, MapOctree* oct) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_map_octree_scan, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, n, oct);
  hipLaunchKernelGGL(k_map_octree_apply, dim3(1), dim3(64), 0, s, pts, res, oct);
}
void launch_map_keys(hipStream_t s, const float4* pts, int n, double res, const MapOctree* oct, unsigned long long* keys, unsigned* vals) {
  if (n > 0) hipLaunchKernelGGL(k_map_keys, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, n, res, oct, keys, vals);
}
void launch_map_centers(hipStream_t s, const unsigned long long* keys, const unsigned* head, const unsigned* slot, int n, double res, const MapOctree* oct, float4* out,
                        int* count_out) {
  if (n > 0) hipLaunchKernelGGL(k_map_centers, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, keys, head, slot, n, res, oct, out, count_out);
}

// pack a resident float4 {x,y,z,intensity} array into a cloud: raw = {x,y,z,index}, intensity kept beside it
__global__ __launch_bounds__(kBlock) void k_pf_to_cloud(const float4* __restrict__ in, int n, float4* __restrict__ raw, float* __restrict__ intensity, CloudMeta* meta,
                                                        CloudDesc desc, CloudDesc* desc_out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i == 0) meta_reset(meta);
  if (desc_out && i == 0) *desc_out = desc;
  if (i >= n) return;
  const float4 p = in[i];
  raw[i] = make_float4(p.x, p.y, p.z, __int_as_float(i));
  intensity[i] = p.w;
}
void launch_pf_to_cloud(hipStream_t s, const float4* in, int n, float4* raw, float* intensity, CloudMeta* meta, const CloudDesc* desc, CloudDesc* desc_out) {
  hipLaunchKernelGGL(k_pf_to_cloud, dim3(std::max(1, (n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, in, n, raw, intensity, meta, desc ? *desc : CloudDesc{},
                     desc ? desc_out : nullptr);
}

}  // namespace hgs

// ------------------------------------------------------------------------------------------------ stage-level test hooks
namespace hgs {

__global__ void k_gicp_debug_state(GicpState* st, const double* T12) {
  float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  gicp_state_init(st[0], I);
  for (int i = 0; i < 12; i++) st[0].x0.m[i] = T12[i], st[0].xi.m[i] = T12[i];
}
void launch_gicp_debug_state(hipStream_t s, GicpState* st, const double* T12_dev) { hipLaunchKernelGGL(k_gicp_debug_state, dim3(1), dim3(1), 0, s, st, T12_dev); }

__global__ void k_ndt_debug_state(NdtState* st, NdtAngles* ang, const double* p6, NdtConsts c) {
  float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  ndt_state_init(st[0], I);
  for (int i = 0; i < 6; i++) st[0].p[i] = p6[i], st[0].p_acc[i] = p6[i];
  ndt_angle_tables(st[0].p, c.upstream_hd1_sign, ang[0]);
}
void launch_ndt_debug_state(hipStream_t s, NdtState* st, NdtAngles* ang, const double* p6_dev, NdtConsts c) {
  hipLaunchKernelGGL(k_ndt_debug_state, dim3(1), dim3(1), 0, s, st, ang, p6_dev, c);
}

// out[k] = sum over tiles of partials[t*N + k], in the order k_gicp_solve (N = kAcc) / k_ndt_solve (N = kAccNdt) use
template <int N, int THREADS>
__global__ __launch_bounds__(THREADS) void k_reduce_partials(const double* __restrict__ partials, int ntiles, double* out) {
  __shared__ double acc[N];
  __shared__ double scratch[THREADS];
  reduce_tiles<N, THREADS>(partials, ntiles, acc, scratch);
  if (threadIdx.x < N) out[threadIdx.x] = acc[threadIdx.x];
}
void launch_reduce_partials(hipStream_t s, const double* partials, int ntiles, int width, double* out) {
  if (width == kAccNdt) hipLaunchKernelGGL((k_reduce_partials<kAccNdt, kNdtSolveBlock>), dim3(1), dim3(kNdtSolveBlock), 0, s, partials, ntiles, out);
  else hipLaunchKernelGGL((k_reduce_partials<kAcc, kSolveBlock>), dim3(1), dim3(kSolveBlock), 0, s, partials, ntiles, out);
}

}  // namespace hgs
