// hgs_engine.hip — host side of the MI355X scan-matching backend: handle/cloud life cycle, batched
// orchestration of the kernels in hgs_kernels.hip, and the extern "C" boundary declared in
// include/hgs_registration.h (the replacement for the engines that
// hdl_graph_slam::select_registration_method constructs, src/hdl_graph_slam/registrations.cpp:22-124).
//
// Control flow: every registration (odometry: ScanMatchingOdometryNodelet::matching,
// apps/scan_matching_odometry_nodelet.cpp:165-262) is a batch of one; LoopDetector::matching
// (include/hdl_graph_slam/loop_detector.hpp:117-171) is a batch of B candidates.  All B optimisers advance in
// lock-step "rounds" of a fixed kernel sequence; each problem carries its own LM / Newton state machine in
// HBM, so there is no per-iteration host decision — the host only polls a done-counter.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hgs_registration.h"
#include "hgs_comm.h"
#include "hgs_device.h"
#include "hgs_sort.h"

using namespace hgs;

namespace {

thread_local std::string g_create_error;

struct DeviceBuffer {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct PinnedBuffer {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// Small host -> device uploads (cloud descriptors, guesses, work plans) go through a ring of pinned slots, each guarded by an event recorded
// behind its copy: a slot is rewritten only after the copy that read it has completed, so no caller has to synchronise the stream just to make a
// staging buffer reusable (rounds 1-4 did, once per index build, once per covariance pass, once per NDT plan: 15-30 us of host latency each on
// the single-registration path).
struct PinnedRing {
  static constexpr int kSlots = 8;
  PinnedBuffer buf[kSlots];
  hipEvent_t ev[kSlots] = {};
  bool pending[kSlots] = {};
  int next = 0;
  // a slot of at least `bytes` whose previous upload has completed; *slot identifies it for commit()
  hipError_t stage(size_t bytes, void** host, int* slot) {
    const int k = next;
    next = (next + 1) % kSlots;
    if (pending[k]) {
      const hipError_t e = hipEventSynchronize(ev[k]);
      if (e != hipSuccess) return e;
      pending[k] = false;
    }
    const hipError_t e = buf[k].reserve(bytes);
    if (e != hipSuccess) return e;
    *host = buf[k].p, *slot = k;
    return hipSuccess;
  }
  // the copy out of the slot has been enqueued on `stream`
  hipError_t commit(int slot, hipStream_t stream) {
    if (!ev[slot]) {
      const hipError_t e = hipEventCreateWithFlags(&ev[slot], hipEventDisableTiming);
      if (e != hipSuccess) return e;
    }
    const hipError_t e = hipEventRecord(ev[slot], stream);
    if (e == hipSuccess) pending[slot] = true;
    return e;
  }
  // H2D of `bytes` from `src` (any host memory) to `dst` through a slot
  hipError_t upload(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    void* host = nullptr;
    int slot = 0;
    hipError_t e = stage(bytes, &host, &slot);
    if (e != hipSuccess) return e;
    std::memcpy(host, src, bytes);
    e = hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    return commit(slot, stream);
  }
  void release() {
    for (int k = 0; k < kSlots; k++) {
      if (ev[k]) (void)hipEventDestroy(ev[k]);
      ev[k] = nullptr, pending[k] = false;
      buf[k].release();
    }
  }
};

// A few helper threads that pack point records next to the calling thread (upload_points_packed).  One core packs a cold 119 k-point sweep (3.8 MB
// read, 1.9 MB written) in 150 us; kWorkers + 1 threads in 59 us (same box, scripts/probes/upload_probe.py: hgs_set_source 0.164 -> 0.075 ms of a
// 0.39 ms odometry step).  The workers only touch host memory (no HIP call); they sleep on a condition variable between uploads and are joined by
// hgs_destroy.
struct PackPool {
  static constexpr int kWorkers = 3;
  struct Job {
    const char* src = nullptr;
    float* dst = nullptr;
    size_t n = 0, stride = 0, chunk = 0, nchunks = 0;
    bool has_intensity = false;
    bool scatter = false;  // false: pack strided records at src into float4 at dst (upload); true: scatter float4 at src into strided records at dst (hgs_transform_source)
    std::atomic<size_t> next{0};
    std::atomic<unsigned char>* ready = nullptr;  // [nchunks]
  };
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv;
  Job* job = nullptr;   // guarded by m
  unsigned long generation = 0;
  int active = 0;       // workers inside the current job
  bool stop = false;

  static void pack_chunk(const Job& j, size_t c) {
    const size_t i0 = c * j.chunk, m = std::min(j.chunk, j.n - i0);
    if (j.scatter) {  // x, y, z (and data[3] = 1 when the record has room for it) of the caller's records; everything else in them is left alone
      const float* s4 = reinterpret_cast<const float*>(j.src) + 4 * i0;
      char* o = reinterpret_cast<char*>(j.dst) + i0 * j.stride;
      for (size_t i = 0; i < m; i++) {
        float* f = reinterpret_cast<float*>(o + i * j.stride);
        f[0] = s4[4 * i], f[1] = s4[4 * i + 1], f[2] = s4[4 * i + 2];
        if (j.stride >= 16) f[3] = 1.0f;
      }
      j.ready[c].store(1, std::memory_order_release);
      return;
    }
    const char* s0 = j.src + i0 * j.stride;
    float* dst = j.dst + 4 * i0;
    if (j.has_intensity) {
      for (size_t i = 0; i < m; i++) {
        const float* f = reinterpret_cast<const float*>(s0 + i * j.stride);
        dst[4 * i] = f[0], dst[4 * i + 1] = f[1], dst[4 * i + 2] = f[2], dst[4 * i + 3] = f[4];
      }
    } else {
      for (size_t i = 0; i < m; i++) {
        const float* f = reinterpret_cast<const float*>(s0 + i * j.stride);
        dst[4 * i] = f[0], dst[4 * i + 1] = f[1], dst[4 * i + 2] = f[2], dst[4 * i + 3] = 0.f;
      }
    }
    j.ready[c].store(1, std::memory_order_release);
  }
  // takes chunks until none is left; false if there was none
  static bool help(Job& j) {
    const size_t c = j.next.fetch_add(1, std::memory_order_relaxed);
    if (c >= j.nchunks) return false;
    pack_chunk(j, c);
    return true;
  }
  void worker() {
    unsigned long seen = 0;
    for (;;) {
      Job* j = nullptr;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || (job && generation != seen); });
        if (stop) return;
        seen = generation, j = job, active++;
      }
      while (help(*j)) {
      }
      {
        std::lock_guard<std::mutex> lk(m);
        active--;
      }
      cv.notify_all();
    }
  }
  void post(Job* j) {
    {
      std::lock_guard<std::mutex> lk(m);
      if (threads.empty())
        for (int i = 0; i < kWorkers; i++) threads.emplace_back([this] { worker(); });
      job = j, generation++;
    }
    cv.notify_all();
  }
  // the job's memory may go away after this: no worker is inside it, none will enter it
  void retire() {
    std::unique_lock<std::mutex> lk(m);
    job = nullptr;
    cv.wait(lk, [&] { return active == 0; });
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    for (std::thread& t : threads) t.join();
    threads.clear();
  }
};

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct hgs_cloud {
  hgs_handle* owner = nullptr;
  size_t n_input = 0;
  int P = 1;
  void* block = nullptr;
  size_t block_bytes = 0;
  CloudDesc desc{};
  CloudDesc* dev_desc = nullptr;  // the descriptor inside the cloud's block (pad = 1, sort_off = 0), written by the kernel that fills raw[]: upload_descs of ONE cloud returns it
  float* intensity = nullptr;  // [n_input] PointXYZI intensity (what hgs_cloud_download / the prefilter hand back)
  bool has_index = false;
  bool corr_stale = false;  // hgs_cloud_invalidate: the correspondence seeds are forgotten when the index is rebuilt (k_gather_sorted)
  bool has_cov = false;
  int cov_k = 0;
  // NDT target tables
  bool has_ndt = false;
  double ndt_resolution = 0;
  int ndt_min_points = 0;
  void* ndt_block = nullptr;
  size_t ndt_block_bytes = 0;
  int* ndt_hash_keys = nullptr;
  int* ndt_hash_vals = nullptr;
  NdtCellRec* ndt_cells = nullptr;
  int2* ndt_hash_kv = nullptr;
  int ndt_hash_cap = 0;
  // VGICP Gaussian voxel map (target role)
  bool has_vg = false;
  double vg_resolution = 0;
  int vg_cov_k = 0;
  void* vg_block = nullptr;
  size_t vg_block_bytes = 0;
  // seed grid (target role of the 1-NN kernels: seed_grid_lookup, hgs_kernels.hip)
  bool has_seed = false;
  void* seed_block = nullptr;
  size_t seed_block_bytes = 0;
  int seed_bits = 0;
  int* vg_hash_keys = nullptr;
  int* vg_hash_vals = nullptr;
  NdtCellRec* vg_cells = nullptr;
  int vg_hash_cap = 0;
};

struct ProfEvent {
  int stage;
  hipEvent_t a, b;
};

struct hgs_handle {
  hgs_params prm{};
  int device = 0;
  hipStream_t stream = nullptr;
  // extra streams of a batch (run_batch): the problems are split into lanes whose launch chains are independent, so the
  // block-per-problem solve / decide kernels of one lane run under the point kernels of the others
  hipStream_t lane_stream[7] = {};  // kMaxLanes - 1
  hipEvent_t lane_event[8] = {};
  int lane_start = 1;                // 1: the host synchronises before it releases the lanes of a batch (open_lanes); HGS_LANE_START=0: not (A/B runs)
  int nn_qpw = 0;  // queries per packet of k_gicp_linearize in the two-launch LM rounds: 0 = by launch size (run_batch), 16 / 32 / 64
  int nn_qpw16_below = 96, nn_qpw32_below = 0;  // ... 16 up to this many 256-point tiles in the launch, 32 up to that many
  // levels of the Hilbert curve the index sort compares (HGS_HILBERT_LEVELS, A/B runs; 16 = all 48 bits, rounds 1-3).  64 x 119 k batch, index stage / step:
  // 16 -> 0.915 / 11.99 ms, 13 -> 0.83 / 11.93, 11 -> 0.75 / 11.91, 10 -> 0.75 / 12.1, 9 -> 0.75 / 12.7 (the walks slow down once a cell of the finest compared
  // level holds many points): 13 = 1/8192 of the cloud's extent (2.4 cm on a 200 m scan), one radix pass less for a batch and two for a pair
  int hilbert_levels = 13;
  hipEvent_t comm_event = nullptr;  // hgs_loop_match_batch_sharded: the gathered headers have reached the host
  // measured on the 16 x 120 k-point loop batch: 1 -> 2 -> 4 lanes = 2850 -> 2935 -> 2975 GICP reg/s, 735 -> 772 -> 797 NDT;
  // 8 lanes: 2440 / 665 (the HIP runtime multiplexes streams onto 4 hardware queues by default).  Round 2, 64 x 119 k-point batch:
  // 1 / 2 / 3 / 4 lanes = 3680 / 3999 / 3937 / 3913 GICP reg/s and 1064 / 1403 / 1384 / 1326 NDT — with 32 problems per lane a lane's
  // launches fill the device on their own and two chains are enough to cover each other's solves and tails.
  int prefilter_fast = 1;  // hgs_prefilter: distance filter inside the voxel grid's kernels, RadiusOutlierRemoval on the voxel grid (0: the separate passes + search tree; A/B, tests)
  int upload_trace = 0;
  int early_run_ahead = 2;  // rounds the host keeps queued in front of a single registration that hands its result over early (the surplus drains behind the caller's back; 3 / 4 measured: no gain for the odometry source, config 2 0.837 -> 0.844 / 0.852 ms back to back: profiles/r06_ab16_early_run_ahead.log)
  int early_result = 1;    // a single registration in two-launch rounds hands its result over in host-mapped memory (run_batch; 0: result kernel + copy + synchronisation)
  PinnedBuffer h_early;                // the host-mapped record
  hipEvent_t early_event = nullptr;    // behind the rounds that were still queued when hgs_align returned early
  bool early_pending = false, early_valid = false;
  int fused_rounds = 1;    // launches of a few GICP problems (below) of at most fused_rounds_below points each: the LM round in two launches (control steps replicated per block; 0: four launches per round)
  int fused_rounds_below = 262144;
  // ... or batches of more problems of at most that many 256-point tiles in total: the loop-closure batch of a KITTI run (candidate keyframes are prefiltered
  // sweeps of 11-14 k points).  Under four lanes: 6 / 12 / 24 candidates x 11 k points 0.62 -> 0.57 / 0.73 -> 0.68 / 0.88 -> 0.83 ms per detection, 48
  // candidates (2160 tiles) 1.07 -> 1.12.  Such batches run on ONE lane now (open_lanes), where the two-launch round is worth less: 6 / 12 candidates
  // 0.521 -> 0.514 / 0.633 -> 0.628 ms, 24 candidates (1080 tiles) 0.794 -> 0.819 — hence 640 (scripts/probes/small_batch_probe.py, profiles/r06_small_batch.log)
  int fused_rounds_max_problems = kFusedRoundMaxProblems, fused_rounds_max_blocks = 640;
  int cov_split = 1;       // non-FROBENIUS regularisations: search kernel + k_cov_regularize (0: one kernel with the eigen-decomposition inline; HGS_COV_SPLIT, A/B runs)
  int resident_descs = 1;  // HGS_RESIDENT_DESCS=0: every stage uploads its descriptor array (A/B runs)
  int knn_qpw_tiny = -1;  // queries per packet of small k_knn_cov launches: -1 = by launch size (queries_per_wave), 0 = 32 as before round 6, 8 / 16 / 24 = below knn_tiny_below queries
  int knn_tiny_below = 32768;
  int seed_grid = 1;    // targets of 1-NN searches get a seed grid (ensure_seed_grid); HGS_SEED_GRID=0 (A/B runs)
  int knn_replay = -1;  // k_knn_cov gather: -1 default (2), 0 tree walk, 1 leaf-log replay, 2 per-lane leaf lists; HGS_KNN_REPLAY (A/B runs, tests)
  int batch_lanes = 0;  // 0: open_lanes chooses (4; NDT_OMP above 32 problems 3; never more than the process's hardware-queue budget has room for); HGS_BATCH_LANES fixes it (A/B runs)
  std::string err;
  hgs_cloud* target = nullptr;
  hgs_cloud* source = nullptr;
  bool own_target = false, own_source = false;
  float final_T[16];

  DeviceBuffer staging, sort_keys[2], sort_vals[2], sort_tmp, descs, states, angles, partials, partials_err, results, guesses, done, misc;
  DeviceBuffer lane_partials[7], lane_partials_err[7];
  DeviceBuffer ndt_accum;  // NdtAccum per problem of the running NDT batch
  hgs::Comm* comm = nullptr;            // hgs_comm_init: the ranks of a sharded loop-closure batch
  DeviceBuffer comm_send, comm_recv, comm_ids;
  DeviceBuffer pf_ukeys;         // prefilter: the voxels' keys in output order (k_pf_grid_radius_flags)
  DeviceBuffer cov_raw;          // staged fp64 neighbourhood covariances between k_knn_cov<.., 2, ..> and k_cov_regularize
  DeviceBuffer ndt_plan;         // per lane: work queue head + tile prefix sums of the running NDT batch
  // blocks per k_ndt_pass launch (0: by the lane count, below); HGS_NDT_RESIDENT (A/B runs).  Two blocks per CU are resident (the round-4 kernel
  // holds 57 KB of LDS per block).  One lane: 768 — the third block per CU fills the tail (512 / 768 / 1024: 497 / 473 / 452 us per whole-device pass
  // but 1024 loses in the batch).  Several concurrent lanes: 512 per launch — the lanes fill each other's tails, and blocks beyond the resident
  // ones only queue in front of the other lanes' (64-candidate batch, 3 lanes, round 4: 384 / 448 / 512 / 576 / 768 = 2272 / 2284 / 2268 / 2213 /
  // 2162 registrations/s; profiles/r04_ndt_knobs.log)
  int ndt_resident_blocks = 0;
  int ndt_chunk = 0;             // largest queue grab in items (0: the default, 8; 1 = one tile per grab); HGS_NDT_CHUNK (A/B runs)
  int ndt_sort = -1;       // NDT source order: -1 Hilbert order if the source has an index, 1 build the index first, 0 input order (HGS_NDT_SORT, A/B runs)
  DeviceBuffer pf_a, pf_b, pf_keep, pf_slot, pf_small, pf_dist;  // prefilter work space
  PinnedBuffer h_results, h_small, h_flags, h_comm;  // h_flags: host-mapped progress mirror (Progress)
  PinnedRing up;                   // small uploads (descriptors, guesses, plans)
  PinnedRing up_points;            // small point uploads: pinned chunks the host packs into (upload_points_packed)
  PinnedBuffer up_big;             // large point uploads: one pinned image of the packed cloud, filled by the pack pool
  hipEvent_t up_big_event = nullptr;  // the last DMA out of up_big
  bool up_big_pending = false;
  PackPool pack_pool;
  PinnedBuffer h_xform;            // hgs_transform_source: the aligned cloud on its way down

  // freed cloud blocks kept for reuse: the odometry path creates and destroys one cloud per sweep, and hipMalloc /
  // hipFree (which synchronises the device) cost more than the upload itself
  std::vector<std::pair<void*, size_t>> block_pool;
  // every cloud this engine has created and not yet destroyed: hgs_destroy orphans them (frees their device memory, clears
  // `owner`) so that a later hgs_cloud_destroy / hgs_cloud_download on a cached pointer is safe instead of a use-after-free
  std::vector<hgs_cloud*> live_clouds;
  // One call at a time per engine: a handle is meant to be driven by one thread (like the pcl::Registration object it replaces), but a
  // garbage-collected wrapper may destroy a cloud from another thread while a long call is running on this engine's stream and
  // buffers — entry points that touch the engine take this lock (different engines never contend).
  std::recursive_mutex api_mutex;

  bool profiling = false;
  std::vector<ProfEvent> prof_events;
  std::vector<ProfEvent> prof_free;
  double prof_ms[HGS_STAGE_COUNT];
  uint64_t prof_launches[HGS_STAGE_COUNT];
};

namespace {

#define HGS_HIP(h, call)                                                                                           \
  do {                                                                                                             \
    hipError_t e__ = (call);                                                                                       \
    if (e__ != hipSuccess) {                                                                                       \
      char buf__[512];                                                                                             \
      snprintf(buf__, sizeof(buf__), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);  \
      (h)->err = buf__;                                                                                            \
      return HGS_ERR_HIP;                                                                                          \
    }                                                                                                              \
  } while (0)

#define HGS_TRY(expr)             \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != HGS_OK) return rc__; \
  } while (0)

struct StageTimer {
  hgs_handle* h;
  ProfEvent ev{};
  bool active = false;
  StageTimer(hgs_handle* h_, int stage) : h(h_) {
    if (!h->profiling) return;
    if (!h->prof_free.empty()) {
      ev = h->prof_free.back();
      h->prof_free.pop_back();
    } else {
      if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return;
    }
    ev.stage = stage;
    active = hipEventRecord(ev.a, h->stream) == hipSuccess;
  }
  ~StageTimer() {
    if (!active) return;
    (void)hipEventRecord(ev.b, h->stream);
    h->prof_events.push_back(ev);
  }
};

int set_device(hgs_handle* h) {
  HGS_HIP(h, hipSetDevice(h->device));
  return HGS_OK;
}

// ---- point upload --------------------------------------------------------------------------------------------
// Host -> device of n strided point records (pcl::PointXYZI: 32 bytes; anything with x, y, z at floats 0..2 and, from 20 bytes on, the intensity
// at float 4) as packed 16-byte records {x, y, z, intensity} in h->staging.  The HOST packs chunk after chunk into pinned buffers (half the bytes
// of a PointXYZI go over PCIe, none of them through the runtime's pageable staging path) and every chunk's DMA runs while the next chunk is packed.
// Round 4 handed the caller's pageable buffer to hipMemcpyAsync and synchronised behind the kernels that followed (profiles/r05_upload.md).
// The caller's buffer has been read completely when this returns; the device side is ordered on h->stream.
constexpr size_t kUploadChunkPoints = 16384;     // 256 KB per pinned chunk
#ifndef HGS_UPLOAD_PARALLEL_POINTS
#define HGS_UPLOAD_PARALLEL_POINTS 49152  // (A/B knob: a huge value keeps every upload on the calling thread)
#endif
constexpr size_t kUploadParallelPoints = HGS_UPLOAD_PARALLEL_POINTS;  // from three chunks on the pack pool helps (below: one thread, the ring of pinned chunks)
int upload_points_packed(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes, const float4** dev) {
  HGS_HIP(h, h->staging.reserve(std::max<size_t>(n, 1) * sizeof(float4)));
  *dev = h->staging.as<float4>();
  const bool has_intensity = stride_bytes >= 20;
  const char* src = static_cast<const char*>(pts);
  if (n >= kUploadParallelPoints) {
    // large cloud: kWorkers + this thread pack chunks into ONE pinned image; this thread sends chunk c down as soon as it is ready (in order)
    if (h->up_big_pending) {
      HGS_HIP(h, hipEventSynchronize(h->up_big_event));  // the previous upload's DMA has left the pinned image (it finished long ago)
      h->up_big_pending = false;
    }
    HGS_HIP(h, h->up_big.reserve(n * sizeof(float4)));
    const size_t nchunks = (n + kUploadChunkPoints - 1) / kUploadChunkPoints;
    std::vector<std::atomic<unsigned char>> ready(nchunks);
    for (auto& r : ready) r.store(0, std::memory_order_relaxed);
    PackPool::Job job;
    job.src = src, job.dst = h->up_big.as<float>(), job.n = n, job.stride = stride_bytes, job.chunk = kUploadChunkPoints, job.nchunks = nchunks;
    job.has_intensity = has_intensity, job.ready = ready.data();
    h->pack_pool.post(&job);
    hipError_t err = hipSuccess;
    for (size_t c = 0; c < nchunks; c++) {
      while (!ready[c].load(std::memory_order_acquire))
        if (!PackPool::help(job)) std::this_thread::yield();
      const size_t i0 = c * kUploadChunkPoints, m = std::min(kUploadChunkPoints, n - i0);
      if (err == hipSuccess) err = hipMemcpyAsync(h->staging.as<float4>() + i0, h->up_big.as<float4>() + i0, m * sizeof(float4), hipMemcpyHostToDevice, h->stream);
    }
    h->pack_pool.retire();  // (`job` and `ready` live on this stack frame)
    HGS_HIP(h, err);
    if (!h->up_big_event) HGS_HIP(h, hipEventCreateWithFlags(&h->up_big_event, hipEventDisableTiming));
    HGS_HIP(h, hipEventRecord(h->up_big_event, h->stream));
    h->up_big_pending = true;
    return HGS_OK;
  }
  for (size_t i0 = 0; i0 < n; i0 += kUploadChunkPoints) {
    const size_t m = std::min(kUploadChunkPoints, n - i0);
    void* staged = nullptr;
    int slot = 0;
    HGS_HIP(h, h->up_points.stage(kUploadChunkPoints * sizeof(float4), &staged, &slot));
    float* dst = static_cast<float*>(staged);
    const char* s0 = src + i0 * stride_bytes;
    if (has_intensity) {
      for (size_t i = 0; i < m; i++) {
        const float* f = reinterpret_cast<const float*>(s0 + i * stride_bytes);
        dst[4 * i] = f[0], dst[4 * i + 1] = f[1], dst[4 * i + 2] = f[2], dst[4 * i + 3] = f[4];
      }
    } else {
      for (size_t i = 0; i < m; i++) {
        const float* f = reinterpret_cast<const float*>(s0 + i * stride_bytes);
        dst[4 * i] = f[0], dst[4 * i + 1] = f[1], dst[4 * i + 2] = f[2], dst[4 * i + 3] = 0.f;
      }
    }
    HGS_HIP(h, hipMemcpyAsync(h->staging.as<float4>() + i0, staged, m * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    HGS_HIP(h, h->up_points.commit(slot, h->stream));
  }
  return HGS_OK;
}

// ---- cloud allocation --------------------------------------------------------------------------------------
int cloud_alloc(hgs_handle* h, size_t n, hgs_cloud** out) {
  hgs_cloud* c = new hgs_cloud();
  c->owner = h;
  c->n_input = n;
  c->P = next_pow2((int)((n + kLeaf - 1) / kLeaf));
  if (c->P < 1) c->P = 1;
  const size_t slots = (size_t)c->P * kLeaf;
  // the per-input-point arrays are sized for n rounded up to 8192 points: consecutive sweeps of one sensor differ by a few hundred returns, and a
  // block that is a few bytes too small for the next sweep cannot be reused from the pool — the odometry path then paid a hipMalloc and (pool full)
  // a hipFree, which synchronises the device, on most sweeps: 0.3-0.4 ms of a 0.7 ms raw-sweep step (profiles/r05_upload.md)
  const size_t n_cap = align_up(std::max<size_t>(n, 1), 8192);
  size_t off = 0;
  const size_t o_meta = off;
  off = align_up(off + sizeof(CloudMeta), 256);
  const size_t o_desc = off;
  off = align_up(off + sizeof(CloudDesc), 256);
  const size_t o_raw = off;
  off = align_up(off + n_cap * sizeof(float4), 256);
  const size_t o_pts = off;
  off = align_up(off + slots * sizeof(float4), 256);
  const size_t o_lpts = off;
  off = align_up(off + slots * sizeof(float4), 256);
  const size_t o_nodes = off;
  off = align_up(off + ((size_t)4 * c->P + 32) * sizeof(float4), 256);  // + four 128-byte groups: the 4-ary walks read whole groups, the quad walk four at a time
  const size_t o_cov = off;
  off = align_up(off + 2 * slots * sizeof(float4), 256);
  const size_t o_corr = off;
  off = align_up(off + slots * sizeof(int), 256);
  const size_t o_int = off;
  off = align_up(off + n_cap * sizeof(float), 256);
  c->block_bytes = off;
  hipError_t e = hipSuccess;
  for (size_t k = 0; k < h->block_pool.size(); k++) {
    if (h->block_pool[k].second >= off && h->block_pool[k].second <= 2 * off + (1u << 20)) {
      c->block = h->block_pool[k].first;
      c->block_bytes = h->block_pool[k].second;
      h->block_pool.erase(h->block_pool.begin() + k);
      break;
    }
  }
  if (!c->block) e = hipMalloc(&c->block, off);
  if (e != hipSuccess) {
    h->err = std::string("hipMalloc(cloud) failed: ") + hipGetErrorString(e);
    delete c;
    return HGS_ERR_HIP;
  }
  char* base = (char*)c->block;
  c->desc.meta = (CloudMeta*)(base + o_meta);
  c->dev_desc = (CloudDesc*)(base + o_desc);
  c->desc.raw = (const float4*)(base + o_raw);
  c->desc.pts = (float4*)(base + o_pts);
  c->desc.lpts = (float4*)(base + o_lpts);
  c->desc.nodes = (float4*)(base + o_nodes);
  c->desc.cov = (float4*)(base + o_cov);
  c->desc.corr = (int*)(base + o_corr);
  c->intensity = (float*)(base + o_int);
  c->desc.n_input = (int)n;
  c->desc.P = c->P;
  c->desc.sort_off = 0;
  c->desc.pad = 0;
  c->corr_stale = true;  // no correspondences yet: k_gather_sorted writes -1 when the index is built (every reader of corr has built it first)
  h->live_clouds.push_back(c);
  *out = c;
  return HGS_OK;
}

// Releases the device memory of a cloud (the struct itself stays): what hgs_destroy does to clouds that outlive their engine.
void cloud_release_device(hgs_cloud* c, bool pool) {
  if (c->block) {
    // stream order makes reuse safe: every kernel that touches the block was enqueued on the owner's stream
    hgs_handle* h = c->owner;
    if (pool && h && h->block_pool.size() < 6) h->block_pool.emplace_back(c->block, c->block_bytes);
    else (void)hipFree(c->block);
  }
  if (c->ndt_block) (void)hipFree(c->ndt_block);
  if (c->vg_block) (void)hipFree(c->vg_block);
  if (c->seed_block) (void)hipFree(c->seed_block);
  c->block = nullptr, c->ndt_block = nullptr, c->vg_block = nullptr, c->seed_block = nullptr;
  c->has_index = c->has_cov = c->has_ndt = c->has_vg = c->has_seed = false;
}

void cloud_free(hgs_cloud* c) {
  if (!c) return;
  cloud_release_device(c, true);
  if (hgs_handle* h = c->owner) {
    auto it = std::find(h->live_clouds.begin(), h->live_clouds.end(), c);
    if (it != h->live_clouds.end()) h->live_clouds.erase(it);
  }
  delete c;
}

// Upload descriptors of a set of clouds (with their sort offsets) into h->descs; returns device pointer.
// what a cloud's resident descriptor holds: pad = 1 (an index build always follows a creation or an invalidation, both of which ask for corr[] to be
// cleared; nothing else reads pad), sort_off = 0 (a cloud alone in a batch sort)
CloudDesc resident_desc(const hgs_cloud* c) {
  CloudDesc d = c->desc;
  d.sort_off = 0, d.pad = 1;
  return d;
}

int upload_descs(hgs_handle* h, const std::vector<hgs_cloud*>& clouds, bool with_sort_offsets, const CloudDesc** dev, size_t* total_n) {
  const size_t B = clouds.size();
  if (B == 1 && clouds[0]->dev_desc && h->resident_descs) {  // one cloud: its resident descriptor, no copy
    if (total_n) *total_n = clouds[0]->n_input;
    *dev = clouds[0]->dev_desc;
    return HGS_OK;
  }
  HGS_HIP(h, h->descs.reserve(B * sizeof(CloudDesc)));
  void* staged = nullptr;
  int slot = 0;
  HGS_HIP(h, h->up.stage(B * sizeof(CloudDesc), &staged, &slot));
  CloudDesc* hd = static_cast<CloudDesc*>(staged);
  size_t off = 0;
  for (size_t i = 0; i < B; i++) {
    hd[i] = clouds[i]->desc;
    hd[i].sort_off = with_sort_offsets ? (int)off : 0;
    off += clouds[i]->n_input;
  }
  if (total_n) *total_n = off;
  HGS_HIP(h, hipMemcpyAsync(h->descs.p, hd, B * sizeof(CloudDesc), hipMemcpyHostToDevice, h->stream));
  HGS_HIP(h, h->up.commit(slot, h->stream));
  *dev = h->descs.as<CloudDesc>();
  return HGS_OK;
}

// Queries per wave of the kNN covariance kernel.  A launch with few queries cannot fill 1024 SIMDs with 64-query packets
// (one 120 k-point cloud is 1.8 waves per SIMD) and is bound by the dependent-load chain of a single walk; 32-query
// packets walk fewer nodes and put more waves in flight (0.91 -> 0.66 ms for one 120 k-point cloud).  Large batches keep
// 64 (least total work); the 1-NN kernels always do (no measurable gain from shorter packets there).
// Round 6, second visit to the question: a launch of a few packets per SIMD lasts as long as its SLOWEST packet, and those are the packets over sparse far
// returns at ~4x the median's leaves (scripts/probes/knn_probe.py, profiles/r06_knn_probe_before.log).  Shorter packets split them — behind a full pre-fill
// window (k_knn_cov<.., WINDOW>: the first attempt, 16 own points in front of unfilled lists, measured equal).  Same-box (profiles/r06_ab8_knn_short_packets.log):
// the odometry source behind the KITTI prefilter (13.5 k points) hgs_align 0.390 -> 0.355 ms with 8-query packets (16: 0.361); one raw HDL-32E source (65 k)
// 0.843 -> 0.787 ms with 16 (8: 0.808, 24: 0.795); the HDL-32E PAIR in one launch (131 k queries) loses with every short packet (covariance stage 0.234 -> 0.27-0.39 ms).
// `tiny`: -1 = these tiers, 0 = none (32 as before), 8 / 16 / 24 = that packet below `tiny_below` queries (A/B runs, tests).
int queries_per_wave(size_t total_queries, int small, int tiny = 0, size_t tiny_below = 0) {
  if (tiny < 0) {
    if (total_queries < (size_t)32768) return 8;
    if (total_queries < (size_t)100000) return 16;
  } else if (tiny > 0 && total_queries < tiny_below) {
    return tiny;
  }
  return total_queries >= (size_t)600000 ? 64 : small;
}

// Build the search index (Hilbert sort + implicit tree) of every cloud in the list that lacks one — one
// batched kernel sequence and ONE radix sort for the whole list.
int ensure_index(hgs_handle* h, const std::vector<hgs_cloud*>& all) {
  std::vector<hgs_cloud*> todo;
  for (hgs_cloud* c : all)
    if (!c->has_index && std::find(todo.begin(), todo.end(), c) == todo.end()) todo.push_back(c);
  if (todo.empty()) return HGS_OK;
  // the key carries the cloud ordinal in bits 48.. ; process in chunks of at most 65536 clouds
  for (size_t start = 0; start < todo.size(); start += 65536) {
    std::vector<hgs_cloud*> chunk(todo.begin() + start, todo.begin() + std::min(todo.size(), start + 65536));
    StageTimer tm(h, HGS_STAGE_INDEX);
    const CloudDesc* d_descs = nullptr;
    size_t total = 0;
    for (hgs_cloud* c : chunk) c->desc.pad = c->corr_stale ? 1 : 0;  // read by k_gather_sorted
    HGS_TRY(upload_descs(h, chunk, true, &d_descs, &total));
    for (hgs_cloud* c : chunk) c->desc.pad = 0;  // (corr_stale is cleared where has_index is set: a failure below must not lose the invalidate request)
    int max_n = 0, max_P = 1;
    for (hgs_cloud* c : chunk) max_n = std::max(max_n, (int)c->n_input), max_P = std::max(max_P, c->P);
    const int nc = (int)chunk.size();
    // meta (nvalid, bbox) was filled at upload time
    if (total > 0) {
      for (int i = 0; i < 2; i++) {
        HGS_HIP(h, h->sort_keys[i].reserve(total * sizeof(uint64_t)));
        HGS_HIP(h, h->sort_vals[i].reserve(total * sizeof(uint32_t)));
      }
      // The sort compares the top `hilbert_levels` levels of the curve (3 bits each) + the cloud ordinal: points that share a cell of the finest
      // compared level stay in input order (stable sort), i.e. in firing order — still neighbours.  Fewer compared bits = fewer radix passes.
      const int drop_bits = 3 * (16 - h->hilbert_levels);
      launch_hilbert_keys(h->stream, d_descs, nc, max_n, h->sort_keys[0].as<unsigned long long>(), h->sort_vals[0].as<unsigned>(), drop_bits);
      int bits = 48;
      for (int v = nc - 1; v > 0; v >>= 1) bits++;
      size_t tmp_bytes = 0;
      int rc = hgs_sort_pairs_u64_u32(nullptr, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                      h->sort_vals[1].as<uint32_t>(), total, drop_bits, bits, h->stream);
      if (rc != 0) {
        h->err = "rocprim radix_sort_pairs (size query) failed";
        return HGS_ERR_HIP;
      }
      HGS_HIP(h, h->sort_tmp.reserve(tmp_bytes));
      rc = hgs_sort_pairs_u64_u32(h->sort_tmp.p, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                  h->sort_vals[1].as<uint32_t>(), total, drop_bits, bits, h->stream);
      if (rc != 0) {
        h->err = "rocprim radix_sort_pairs failed";
        return HGS_ERR_HIP;
      }
    }
    launch_gather_sorted(h->stream, d_descs, nc, max_P * kLeaf, h->sort_vals[1].as<unsigned>());
    launch_build_tree(h->stream, d_descs, nc, max_P);
    HGS_HIP(h, hipGetLastError());
    // (the device-side descs buffer is reused by later stages in stream order; the pinned staging slot is guarded by its event, PinnedRing)
    for (hgs_cloud* c : chunk) c->has_index = true, c->corr_stale = false;
  }
  return HGS_OK;
}

// what a cloud's cached covariances depend on besides its points
int cov_cache_key(const hgs_handle* h, int k) { return k | (h->prm.regularization_method << 16); }

int ensure_cov(hgs_handle* h, const std::vector<hgs_cloud*>& all, int k) {
  HGS_TRY(ensure_index(h, all));
  const int key = cov_cache_key(h, k);
  std::vector<hgs_cloud*> todo;
  for (hgs_cloud* c : all)
    if ((!c->has_cov || c->cov_k != key) && std::find(todo.begin(), todo.end(), c) == todo.end()) todo.push_back(c);
  if (todo.empty()) return HGS_OK;
  StageTimer tm(h, HGS_STAGE_COVARIANCE);
  const CloudDesc* d_descs = nullptr;
  HGS_TRY(upload_descs(h, todo, false, &d_descs, nullptr));
  int max_n = 0;
  for (hgs_cloud* c : todo) max_n = std::max(max_n, (int)c->n_input);
  size_t total_q = 0;
  for (hgs_cloud* c : todo) total_q += c->n_input;
  int qpw = queries_per_wave(total_q, 32, h->knn_qpw_tiny, (size_t)h->knn_tiny_below);  // (32: >= k points in the pre-fill window; shorter packets borrow a window)
  // the leaf-log gather pays for batches of LiDAR keyframes, not for one or two (dense) clouds: see launch_knn_cov
  // pass 2 of k_knn_cov: per-lane leaf lists (mode 2; measured against the leaf-log replay and the second tree walk: 64 LiDAR clouds
  // 4.25 -> 4.0 ms, dense 1 M-point pair 1.46 -> 1.27 ms, one HDL-32E pair unchanged); HGS_KNN_REPLAY=0|1|2 forces a mode (tests, A/B)
  const int gather = h->knn_replay >= 0 ? h->knn_replay : 2;
  if (qpw < 32 && (gather != 2 || k > 20)) qpw = 32;  // (the short-packet instantiation exists for the per-lane lists and k <= 20: launch_knn_cov)
  double* raw_stage = nullptr;
  if (h->prm.regularization_method != 0 && h->cov_split) {  // PLANE / MIN_EIG / ...: the eigen-decompositions in a kernel of their own (k_cov_regularize)
    HGS_HIP(h, h->cov_raw.reserve(todo.size() * (size_t)std::max(max_n, 1) * 6 * sizeof(double)));
    raw_stage = h->cov_raw.as<double>();
  }
  launch_knn_cov(h->stream, d_descs, (int)todo.size(), max_n, k, qpw, h->prm.regularization_method, gather, raw_stage);
  HGS_HIP(h, hipGetLastError());
  for (hgs_cloud* c : todo) c->has_cov = true, c->cov_k = key;
  return HGS_OK;
}

int ensure_ndt_target(hgs_handle* h, hgs_cloud* c) {
  const double res = h->prm.resolution;
  const int min_pts = h->prm.ndt_min_points_per_voxel;
  if (c->has_ndt && c->ndt_resolution == res && c->ndt_min_points == min_pts) return HGS_OK;
  StageTimer tm(h, HGS_STAGE_VOXELIZE);
  const size_t n = c->n_input;
  const int max_cells = (int)(n / (size_t)std::max(1, min_pts)) + 1;
  const int cap = next_pow2(std::max(64, 4 * max_cells));
  if (!c->ndt_block || c->ndt_hash_cap != cap) {
    if (c->ndt_block) (void)hipFree(c->ndt_block);
    c->ndt_block = nullptr;
    const size_t o_keys = 0, o_vals = align_up((size_t)cap * 4, 256), o_kv = o_vals + align_up((size_t)cap * 4, 256);
    const size_t o_cells = o_kv + align_up((size_t)cap * 8, 256);
    const size_t bytes = o_cells + (size_t)max_cells * sizeof(NdtCellRec);
    HGS_HIP(h, hipMalloc(&c->ndt_block, bytes));
    c->ndt_block_bytes = bytes;
    c->ndt_hash_keys = (int*)((char*)c->ndt_block + o_keys);
    c->ndt_hash_vals = (int*)((char*)c->ndt_block + o_vals);
    c->ndt_hash_kv = (int2*)((char*)c->ndt_block + o_kv);
    c->ndt_cells = (NdtCellRec*)((char*)c->ndt_block + o_cells);
    c->ndt_hash_cap = cap;
  }
  HGS_HIP(h, hipMemsetAsync(c->ndt_hash_keys, 0xff, (size_t)cap * 4, h->stream));
  const float inv_leaf = 1.0f / (float)res;
  launch_ndt_grid_params(h->stream, c->desc, inv_leaf);
  if (n > 0) {
    for (int i = 0; i < 2; i++) {
      HGS_HIP(h, h->sort_keys[i].reserve(n * sizeof(uint64_t)));
      HGS_HIP(h, h->sort_vals[i].reserve(n * sizeof(uint32_t)));
    }
    launch_ndt_cell_keys(h->stream, c->desc, inv_leaf, h->sort_keys[0].as<unsigned long long>(), h->sort_vals[0].as<unsigned>());
    size_t tmp_bytes = 0;
    int rc = hgs_sort_pairs_u64_u32(nullptr, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                    h->sort_vals[1].as<uint32_t>(), n, 0, 32, h->stream);
    if (rc != 0) {
      h->err = "rocprim radix_sort_pairs (size query) failed";
      return HGS_ERR_HIP;
    }
    HGS_HIP(h, h->sort_tmp.reserve(tmp_bytes));
    rc = hgs_sort_pairs_u64_u32(h->sort_tmp.p, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                h->sort_vals[1].as<uint32_t>(), n, 0, 32, h->stream);
    if (rc != 0) {
      h->err = "rocprim radix_sort_pairs failed";
      return HGS_ERR_HIP;
    }
    launch_ndt_build_cells(h->stream, c->desc, h->sort_keys[1].as<unsigned long long>(), h->sort_vals[1].as<unsigned>(), min_pts, c->ndt_hash_keys,
                           c->ndt_hash_vals, cap - 1, c->ndt_cells);
  }
  launch_ndt_pack_hash(h->stream, c->ndt_hash_keys, c->ndt_hash_vals, c->ndt_hash_kv, cap);
  HGS_HIP(h, hipGetLastError());
  c->has_ndt = true;
  c->ndt_resolution = res;
  c->ndt_min_points = min_pts;
  return HGS_OK;
}

// FastVGICP's GaussianVoxelMap of the target (needs the target's kNN covariances first).  Upstream rebuilds it at
// the start of every align(); the result only depends on (target, resolution, k), so it is cached on the cloud.
int ensure_vgicp_target(hgs_handle* h, hgs_cloud* c) {
  const double res = h->prm.resolution;
  const int k = h->prm.correspondence_randomness;
  if (c->has_vg && c->vg_resolution == res && c->vg_cov_k == cov_cache_key(h, k)) return HGS_OK;
  StageTimer tm(h, HGS_STAGE_VOXELIZE);
  const size_t n = c->n_input;
  const int max_cells = (int)n + 1;
  const int cap = next_pow2(std::max<int>(64, 2 * max_cells));
  if (!c->vg_block || c->vg_hash_cap != cap) {
    if (c->vg_block) (void)hipFree(c->vg_block);
    c->vg_block = nullptr;
    const size_t o_keys = 0, o_vals = align_up((size_t)cap * 4, 256), o_cells = o_vals + align_up((size_t)cap * 4, 256);
    const size_t bytes = o_cells + (size_t)max_cells * sizeof(NdtCellRec);
    HGS_HIP(h, hipMalloc(&c->vg_block, bytes));
    c->vg_block_bytes = bytes;
    c->vg_hash_keys = (int*)((char*)c->vg_block + o_keys);
    c->vg_hash_vals = (int*)((char*)c->vg_block + o_vals);
    c->vg_cells = (NdtCellRec*)((char*)c->vg_block + o_cells);
    c->vg_hash_cap = cap;
  }
  HGS_HIP(h, hipMemsetAsync(c->vg_hash_keys, 0xff, (size_t)cap * 4, h->stream));
  launch_vgicp_grid_params(h->stream, c->desc, res);
  if (n > 0) {
    for (int i = 0; i < 2; i++) {
      HGS_HIP(h, h->sort_keys[i].reserve(n * sizeof(uint64_t)));
      HGS_HIP(h, h->sort_vals[i].reserve(n * sizeof(uint32_t)));
    }
    launch_vgicp_cell_keys(h->stream, c->desc, res, h->sort_keys[0].as<unsigned long long>(), h->sort_vals[0].as<unsigned>());
    size_t tmp_bytes = 0;
    int rc = hgs_sort_pairs_u64_u32(nullptr, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                    h->sort_vals[1].as<uint32_t>(), n, 0, 32, h->stream);
    if (rc != 0) {
      h->err = "rocprim radix_sort_pairs (size query) failed";
      return HGS_ERR_HIP;
    }
    HGS_HIP(h, h->sort_tmp.reserve(tmp_bytes));
    rc = hgs_sort_pairs_u64_u32(h->sort_tmp.p, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                h->sort_vals[1].as<uint32_t>(), n, 0, 32, h->stream);
    if (rc != 0) {
      h->err = "rocprim radix_sort_pairs failed";
      return HGS_ERR_HIP;
    }
    launch_vgicp_build_cells(h->stream, c->desc, h->sort_keys[1].as<unsigned long long>(), h->sort_vals[1].as<unsigned>(), c->vg_hash_keys,
                             c->vg_hash_vals, cap - 1, c->vg_cells);
  }
  HGS_HIP(h, hipGetLastError());
  c->has_vg = true;
  c->vg_resolution = res;
  c->vg_cov_k = cov_cache_key(h, k);
  return HGS_OK;
}

NdtTargetView ndt_target_view(const hgs_handle* h, const hgs_cloud* t) {
  NdtTargetView tv;
  tv.hash_keys = t->ndt_hash_keys, tv.hash_vals = t->ndt_hash_vals, tv.cells = t->ndt_cells, tv.meta = t->desc.meta, tv.hash_kv = t->ndt_hash_kv;
  tv.hash_mask = t->ndt_hash_cap - 1, tv.inv_leaf = 1.0f / (float)h->prm.resolution;
  return tv;
}

NdtTargetView vgicp_target_view(const hgs_cloud* t) {
  NdtTargetView tv;
  tv.hash_keys = t->vg_hash_keys, tv.hash_vals = t->vg_hash_vals, tv.cells = t->vg_cells, tv.meta = t->desc.meta, tv.hash_kv = nullptr;
  tv.hash_mask = t->vg_hash_cap - 1, tv.inv_leaf = 0.f;
  return tv;
}

VgicpConsts vgicp_consts(const hgs_params& p) {
  VgicpConsts c;
  c.resolution = p.resolution;
  c.search = p.neighbor_search == HGS_DIRECT27 ? 3 : (p.neighbor_search == HGS_DIRECT7 ? 2 : 1);
  c.pad = 0;
  return c;
}

TargetView target_view(const hgs_cloud* c) {
  TargetView t;
  t.nodes = c->desc.nodes, t.pts = c->desc.pts, t.lpts = c->desc.lpts, t.cov = c->desc.cov, t.meta = c->desc.meta, t.P = c->P;
  t.seed_bits = c->has_seed ? c->seed_bits : 0, t.seed_tab = c->has_seed ? static_cast<const unsigned*>(c->seed_block) : nullptr;
  return t;
}

// The seed grid of a cloud that serves as the target of 1-NN searches (hgs_kernels.hip, seed_grid_lookup): built once per index, ~10 us.
int ensure_seed_grid(hgs_handle* h, hgs_cloud* c) {
  if (!h->seed_grid || c->has_seed || !c->has_index || c->n_input == 0) return HGS_OK;
  int bits = 12;
  while (bits < 24 && ((size_t)1 << bits) < 4 * c->n_input) bits++;
  const size_t bytes = seed_grid_entries(bits) * sizeof(unsigned);
  if (!c->seed_block || c->seed_bits != bits) {
    if (c->seed_block) (void)hipFree(c->seed_block);
    c->seed_block = nullptr;
    HGS_HIP(h, hipMalloc(&c->seed_block, bytes));
    c->seed_block_bytes = bytes, c->seed_bits = bits;
  }
  HGS_HIP(h, hipMemsetAsync(c->seed_block, 0xff, bytes, h->stream));
  launch_seed_grid_build(h->stream, c->desc.pts, c->desc.meta, (int)c->n_input, static_cast<unsigned*>(c->seed_block), bits);
  HGS_HIP(h, hipGetLastError());
  c->has_seed = true;
  return HGS_OK;
}

GicpConsts gicp_consts(const hgs_params& p) {
  GicpConsts c;
  const double thr = p.max_correspondence_distance;
  c.max_corr2 = thr * thr;
  c.search_bound2 = c.max_corr2 >= (double)FLT_MAX ? FLT_MAX : nextafterf((float)c.max_corr2, FLT_MAX);
  c.rotation_eps = p.rotation_epsilon;
  c.translation_eps = p.transformation_epsilon;
  c.lm_init_lambda_factor = p.lm_init_lambda_factor;
  c.lm_max_iterations = p.lm_max_iterations;
  c.max_iterations = p.max_iterations;
  c.k_correspondences = p.correspondence_randomness;
  return c;
}

NdtConsts ndt_consts(const hgs_params& p) {
  NdtConsts c;
  const double c1 = 10.0 * (1 - p.ndt_outlier_ratio);
  const double c2 = p.ndt_outlier_ratio / std::pow(p.resolution, 3);
  const double d3 = -std::log(c2);
  c.gauss_d1 = -std::log(c1 + c2) - d3;
  c.gauss_d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / c.gauss_d1);
  c.step_size = p.ndt_step_size;
  c.trans_eps = p.transformation_epsilon;
  c.max_iterations = p.max_iterations;
  c.search = p.neighbor_search == HGS_DIRECT1 ? 1 : (p.neighbor_search == HGS_KDTREE ? 0 : 2);
  c.kdtree_radius2 = (float)(p.resolution * p.resolution);
  c.line_search = p.ndt_line_search ? 1 : 0;
  c.upstream_hd1_sign = p.ndt_upstream_hd1_sign;
  c.pad = std::getenv("HGS_TRACE") ? 1 : 0;  // device-side per-iteration trace (parity debugging)
  return c;
}

// The lanes of a batch are HIP streams, and what makes them concurrent is that the runtime puts them on different hardware queues — of which it uses
// 4 by default (GPU_MAX_HW_QUEUES, read once when the HIP runtime initialises).  One engine's four lanes get one each; in a process that hosts several
// engines (the nodelet manager: the odometry engine AND the loop-closure engine) streams beyond the queue count share a queue and a batch runs 8 % slower
// (measured with three engines on 4 queues, profiles/r04_lanes_queues.log; with 8 queues the effect is gone).  The library does NOT touch the
// environment (rounds 1-4 called setenv from a static initialiser: a plugin must not change its host process).  Instead every stream an engine
// creates is counted against the process's queue budget — GPU_MAX_HW_QUEUES if the launcher set it (INTEGRATION.md recommends 8), HIP's default 4
// otherwise — and a batch opens only as many lanes as the budget still has room for (never fewer than one: the engine's own stream).
// (the budget is per DEVICE: in the one-process-several-GPUs shape — LoopMatcherHIP, bench.py --single-process — every GPU has its own queues)
constexpr int kMaxCountedDevices = 64;
std::atomic<int> g_streams_in_use[kMaxCountedDevices] = {};
std::atomic<int>& streams_in_use(int device) { return g_streams_in_use[(unsigned)device % kMaxCountedDevices]; }
int hw_queue_budget() {
  static const int budget = [] {
    const char* e = std::getenv("GPU_MAX_HW_QUEUES");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : 4;
  }();
  return budget;
}
constexpr int kMaxLanes = 8;  // HGS_BATCH_LANES up to 8 (A/B runs); the default choice stays at 3-4 and is further bounded by the queue budget above

// Progress mirror of one lane of a batch: device counters + two ints of host-mapped pinned memory the kernels write into.
int make_progress(hgs_handle* h, int lane, int B, Progress* out) {
  if (h->early_pending) {  // the rounds that were queued behind an early result (they tick the mirror) have to be through before the mirror is reset
    HGS_HIP(h, hipEventSynchronize(h->early_event));
    h->early_pending = false;
  }
  HGS_HIP(h, h->done.reserve(64));
  HGS_HIP(h, h->h_flags.reserve(64));
  volatile int* hf = h->h_flags.as<volatile int>() + 2 * lane;
  hf[0] = 0, hf[1] = 0;  // everything enqueued earlier on these streams has completed (results were fetched with a sync)
  out->dev = h->done.as<int>() + 2 * lane;
  out->host_done = hf, out->host_rounds = hf + 1;
  out->B = B, out->pad = 0;
  return HGS_OK;
}

// One lane of a batch: problems [b0, b0 + B) on their own stream with their own progress mirror and partial-sum buffers.
struct BatchLane {
  hipStream_t stream = nullptr;
  int b0 = 0, B = 0;
  Progress prog{};
  double* partials = nullptr;
  double* partials_err = nullptr;
  long round = 0;
  bool finished = false;
};

// Splits B problems into lanes (contiguous ranges) and makes the extra streams wait for what the main stream has enqueued
// so far (indices, covariances, descriptors, guesses).  Profiling keeps one lane: the stage timers bracket launches on the
// main stream and are meant to time kernels that have the device to themselves.
int open_lanes(hgs_handle* h, int B, size_t partial_bytes_per_problem, size_t partial_err_bytes_per_problem, std::vector<BatchLane>& lanes, int tiles_per_problem) {
  // round 3, 64 x 119 k FAST_GICP batch: 1 / 2 / 3 / 4 lanes = 5615 / 5755 / 5787 / 5811 registrations/s (round 2's kernels preferred 2 above 32 problems)
  // NDT (one launch per iteration, work queue inside): 2 / 3 / 4 lanes = 1679 / 1724 / 1652 on the 64-candidate batch (round 2: 1403 / 1384 / 1326)
  // round 6, batches of SMALL problems (candidate keyframes of a KITTI run: prefiltered sweeps of 11-14 k points; scripts/probes/small_batch_lanes_probe.py,
  // profiles/r06_small_batch_lanes.log): GICP's short kernels gain nothing from concurrent chains and pay for the extra streams — one lane: 6 / 12 / 24 / 48
  // candidates x 11 k points 0.558 -> 0.505 / 0.683 -> 0.624 / 0.810 -> 0.782 / 1.092 -> 1.019 ms per detection (96 candidates: equal; 65 k-point problems:
  // four lanes from 8 candidates on) — while NDT_OMP's passes want all four lanes also above 32 problems (48 x 11 k: 7.9 -> 6.8 ms).
  const bool small_problems = tiles_per_problem > 0 && tiles_per_problem <= 96;
  const int by_size = h->prm.method == HGS_NDT_OMP ? (B > 32 && !small_problems ? 3 : 4) : (small_problems && (long)B * tiles_per_problem <= 3072 ? 1 : 4);
  const int wanted = h->batch_lanes > 0 ? h->batch_lanes : by_size;
  int n = h->profiling ? 1 : std::max(1, std::min(std::min(wanted, kMaxLanes), B));  // (h_flags / done hold 2 ints per lane: 64 bytes = 8 lanes)
  // the process's hardware-queue budget (above): lane streams this engine already owns are free, new ones only while there is room.
  // HGS_BATCH_LANES (A/B runs) overrides the budget.
  if (h->batch_lanes <= 0) {
    int owned = 0;
    while (owned < kMaxLanes - 1 && h->lane_stream[owned]) owned++;
    const int room = std::max(0, hw_queue_budget() - streams_in_use(h->device).load(std::memory_order_relaxed));
    n = std::min(n, 1 + owned + room);
  }
  lanes.assign(n, BatchLane{});
  for (int i = 0, b0 = 0; i < n; i++) {
    BatchLane& L = lanes[i];
    L.b0 = b0, L.B = (B - b0) / (n - i);
    b0 += L.B;
    HGS_TRY(make_progress(h, i, L.B, &L.prog));
    if (i == 0) {
      L.stream = h->stream;
      L.partials = h->partials.as<double>(), L.partials_err = h->partials_err.as<double>();
      continue;
    }
    if (!h->lane_stream[i - 1]) {
      HGS_HIP(h, hipStreamCreateWithFlags(&h->lane_stream[i - 1], hipStreamNonBlocking));
      streams_in_use(h->device).fetch_add(1, std::memory_order_relaxed);
    }
    L.stream = h->lane_stream[i - 1];
    HGS_HIP(h, h->lane_partials[i - 1].reserve((size_t)L.B * partial_bytes_per_problem));
    HGS_HIP(h, h->lane_partials_err[i - 1].reserve((size_t)L.B * partial_err_bytes_per_problem));
    L.partials = h->lane_partials[i - 1].as<double>(), L.partials_err = h->lane_partials_err[i - 1].as<double>();
  }
  if (n > 1) {
    // Round 5 removed the host synchronisations behind the index build and the covariance pass (they only protected a pinned staging buffer: PinnedRing).
    // For a batch with several lanes one synchronisation stays HERE, because it measures faster: with it the host enqueues lane after lane onto an idle
    // device; without it every lane's first rounds are already queued behind one event when the covariance pass ends and all lanes are released in the
    // same instant — 5306 instead of 5381 registrations/s FAST_GICP, 7435 / 7574 PLANE, 2307 / 2317 NDT_OMP (same library, HGS_LANE_START=0 / 1,
    // profiles/r05_lane_start.log).  Releasing lane i + 1 behind lane i's first point kernel instead (a deliberate stagger) is worse than both
    // (5250 / 7269): the ramp costs three unseeded linearisations.  A single registration (one lane) never gets here.
    if (h->lane_start != 0) HGS_HIP(h, hipStreamSynchronize(h->stream));
    if (!h->lane_event[0]) HGS_HIP(h, hipEventCreateWithFlags(&h->lane_event[0], hipEventDisableTiming));
    HGS_HIP(h, hipEventRecord(h->lane_event[0], h->stream));
    for (int i = 1; i < n; i++) HGS_HIP(h, hipStreamWaitEvent(lanes[i].stream, h->lane_event[0], 0));
  }
  return HGS_OK;
}

// The main stream continues after every lane has finished.
int close_lanes(hgs_handle* h, std::vector<BatchLane>& lanes) {
  for (size_t i = 1; i < lanes.size(); i++) {
    if (!h->lane_event[i]) HGS_HIP(h, hipEventCreateWithFlags(&h->lane_event[i], hipEventDisableTiming));
    HGS_HIP(h, hipEventRecord(h->lane_event[i], lanes[i].stream));
    HGS_HIP(h, hipStreamWaitEvent(h->stream, h->lane_event[i], 0));
  }
  return HGS_OK;
}

// Keeps every lane's queue `kRunAhead` rounds ahead of the device without synchronising: enqueue_round(lane) is called
// whenever a lane that still has unfinished problems has fewer than kRunAhead rounds in flight, on_finished(lane) once
// when the lane's problems have all finished (or max_rounds is exhausted) — what it enqueues runs behind the lane's
// last rounds while the other lanes are still iterating; returns once every lane has finished.  If the mirrors stop advancing although the streams have drained (a launch
// failed) the loop keeps enqueueing up to max_rounds: the caller then sees the error from hipGetLastError.
constexpr long kRunAhead = 2;  // one round executing, one queued behind it (a round is >= 100 us, enqueueing one ~20 us)
template <typename F, typename G>
void drive_lanes(std::vector<BatchLane>& lanes, long max_rounds, F&& enqueue_round, G&& on_finished, long run_ahead = kRunAhead) {
  long spins = 0;
  for (;;) {
    bool all_finished = true, enqueued = false;
    for (BatchLane& L : lanes) {
      if (L.finished) continue;
      if (*L.prog.host_done || L.round >= max_rounds) {
        L.finished = true;
        on_finished(L);
        continue;
      }
      all_finished = false;
      if (L.round - (long)*L.prog.host_rounds < run_ahead) {
        enqueue_round(L);
        L.round++;
        enqueued = true;
      }
    }
    if (all_finished) return;
    if (enqueued) {
      spins = 0;
      continue;
    }
    if ((++spins & 0x3ff) == 0) {
      for (BatchLane& L : lanes)
        if (!L.finished && !*L.prog.host_done && hipStreamQuery(L.stream) == hipSuccess && !*L.prog.host_done &&
            L.round - (long)*L.prog.host_rounds >= run_ahead) {
          enqueue_round(L);  // drained without the mirror advancing: do not spin forever, max_rounds bounds the loop
          L.round++;
        }
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
}

// getFitnessScore of one lane's problems at the poses stored in h->results (exact 1-NN of every transformed source point
// in the target; FAST_GICP seeds the search with the final correspondences).
void lane_fitness(hgs_handle* h, BatchLane& L, const CloudDesc* d_descs, double max_range, int max_blocks, int qpw, int nn_tile) {
  StageTimer tm(h, HGS_STAGE_FITNESS);
  DevResult* res = h->results.as<DevResult>() + L.b0;
  launch_fitness(L.stream, d_descs + L.b0, target_view(h->target), res, max_range, L.partials_err, max_blocks, L.B, h->prm.method == HGS_FAST_GICP ? 1 : 0, qpw);
  launch_fitness_final(L.stream, d_descs + L.b0, L.partials_err, max_blocks, res, L.B, nn_tile);
}

// Run all B registrations (sources vs the handle's target) to completion; results land in h->results (device).
// With fit_max_range the getFitnessScore pass of every problem (fit_sum / fit_count of its result record) is enqueued on
// the problem's lane as soon as the lane has converged, i.e. under the remaining iterations of the other lanes.
int run_batch(hgs_handle* h, const std::vector<hgs_cloud*>& sources, const float* guesses_host, const double* fit_max_range = nullptr) {
  const int B = (int)sources.size();
  h->early_valid = false;
  hgs_cloud* tgt = h->target;
  const int method = h->prm.method;
  std::vector<hgs_cloud*> all(sources);
  if (fit_max_range) {
    std::vector<hgs_cloud*> searched(sources);
    searched.push_back(tgt);
    HGS_TRY(ensure_index(h, searched));
  }
  if (method == HGS_FAST_GICP || method == HGS_FAST_VGICP) {
    all.push_back(tgt);
    HGS_TRY(ensure_cov(h, all, h->prm.correspondence_randomness));
    if (method == HGS_FAST_VGICP) HGS_TRY(ensure_vgicp_target(h, tgt));
  } else {
    HGS_TRY(ensure_ndt_target(h, tgt));
  }
  if (fit_max_range && method != HGS_FAST_GICP) HGS_TRY(ensure_seed_grid(h, tgt));  // getFitnessScore without correspondences to start from (k_fitness)
  int max_n = 0;
  for (hgs_cloud* c : sources) max_n = std::max(max_n, (int)c->n_input);
  size_t total_q = 0;
  for (hgs_cloud* c : sources) total_q += c->n_input;
  (void)total_q;
  const int qpw = 64;  // queries per packet of the 1-NN kernels; shorter packets cost a batch ~13 % more work (round 2) ...
  const int nn_tile = (kBlock / 64) * qpw * kNW;                 // points per block of k_gicp_linearize / k_fitness (their own tiling formula)
  // ... but a launch of a FEW blocks lasts as long as one packet walk, and a shorter packet walks fewer nodes: the two-launch LM rounds of a small single
  // registration run k_gicp_linearize<true> with 16- / 32-query packets, whose last wave per block redoes the 64-point wave rows so that no bit of the
  // result depends on the packet size (hgs_kernels.hip).  Same-box (profiles/r06_ab12_nn_qpw.log, r06_ab13): the 13.5 k-point odometry source hgs_align
  // 0.344 -> 0.315 ms with 16; by size (profiles/r06_nn_qpw_sizes.log): 7 k points -6 %, 11 k -4 %, 16 k -3 %, 22 k -1 %, 64 k +2 % (32 never wins).
  // nn_qpw (option): 0 = by launch size, 16 / 32 / 64 = that packet.
  const bool gicp_round2 = method == HGS_FAST_GICP && h->fused_rounds && max_n <= h->fused_rounds_below &&
                           (B <= h->fused_rounds_max_problems || (long)B * ((max_n + kBlock - 1) / kBlock) <= (long)h->fused_rounds_max_blocks);
  const long tiles64 = (long)B * ((max_n + nn_tile - 1) / nn_tile);
  const int lin_qpw = !gicp_round2 ? 64 : h->nn_qpw > 0 ? h->nn_qpw : tiles64 <= (long)h->nn_qpw16_below ? 16 : tiles64 <= (long)h->nn_qpw32_below ? 32 : 64;
  const int lin_tile = (kBlock / 64) * lin_qpw * kNW;
  const int lin_blocks = std::max(1, (max_n + lin_tile - 1) / lin_tile);
  const int max_blocks = std::max(1, lin_qpw < 64 ? (max_n + 63) / 64 : (max_n + nn_tile - 1) / nn_tile);  // >= the tile (row) count of every kernel of the loop
  const int err_blocks = std::max(1, (max_n + kBlock - 1) / kBlock);
  const CloudDesc* d_descs = nullptr;
  HGS_TRY(upload_descs(h, sources, false, &d_descs, nullptr));
  const bool guess_in_args = B == 1 && method == HGS_FAST_GICP;  // (a single GICP registration: the guess rides in k_gicp_init1's arguments)
  HGS_HIP(h, h->guesses.reserve((size_t)B * 16 * sizeof(float)));
  if (!guess_in_args) HGS_HIP(h, h->up.upload(h->guesses.p, guesses_host, (size_t)B * 16 * sizeof(float), h->stream));  // (the caller's array is pageable: through a pinned slot)
  HGS_HIP(h, h->done.reserve(64));
  HGS_HIP(h, h->results.reserve((size_t)B * sizeof(DevResult)));
  HGS_HIP(h, h->partials.reserve((size_t)B * max_blocks * kAccNdt * sizeof(double)));
  HGS_HIP(h, h->partials_err.reserve((size_t)B * max_blocks * 2 * sizeof(double)));
  if (method == HGS_FAST_GICP || method == HGS_FAST_VGICP) {
    const bool voxel = method == HGS_FAST_VGICP;
    const GicpConsts c = gicp_consts(h->prm);
    const VgicpConsts vc = vgicp_consts(h->prm);
    // A launch of a few problems (a single registration: the odometry step, config 2) is a chain of ~4 us kernels in which the two per-problem control launches
    // of an LM round cost as much as its two point kernels: such launches run the round in TWO launches, the control steps replicated in every block of
    // k_gicp_linearize<true> / k_gicp_error<true> (hgs_kernels.hip).  The states then alternate between two buffers; a whole round leaves them in the first.
    const bool round2 = gicp_round2;
    // One registration without a fitness pass behind it (hgs_align: the odometry step): the kernel that finds it finished writes the result record into
    // host-mapped memory in front of the mirror's `done` flag; the poll below returns with it — no result kernel, no copy, no synchronisation.
    const bool early = round2 && B == 1 && !fit_max_range && h->early_result && !h->profiling;
    DevResult* early_out = nullptr;
    if (early) {
      HGS_HIP(h, h->h_early.reserve(sizeof(DevResult)));
      early_out = h->h_early.as<DevResult>();
    }
    HGS_HIP(h, h->states.reserve((size_t)B * sizeof(GicpState) * (round2 ? 2 : 1)));
    GicpState* st = h->states.as<GicpState>();
    GicpState* st_other = st + B;
    const TargetView tv = target_view(tgt);
    const NdtTargetView vtv = voxel ? vgicp_target_view(tgt) : NdtTargetView{};
    const long max_rounds = (long)std::max(1, c.max_iterations) * std::max(1, c.lm_max_iterations) + 2 + (round2 ? 1 : 0);  // (round2: a round's accept / reject runs in the next round's first kernel)
    std::vector<BatchLane> lanes;
    HGS_TRY(open_lanes(h, B, (size_t)max_blocks * kAccNdt * sizeof(double), (size_t)max_blocks * 2 * sizeof(double), lanes, err_blocks));
    auto finish_lane = [&](BatchLane& L) {
      if (early && *L.prog.host_done) {  // (not when the lane ran out of rounds: then the result kernel + copy below fetch whatever state it is in)
        h->early_valid = true;
        return;
      }
      launch_gicp_results(L.stream, st + L.b0, h->results.as<DevResult>() + L.b0, L.B);
      if (fit_max_range) lane_fitness(h, L, d_descs, *fit_max_range, max_blocks, qpw, nn_tile);
    };
    if (guess_in_args) launch_gicp_init1(lanes[0].stream, st, guesses_host, lanes[0].prog);
    else
      for (BatchLane& L : lanes) launch_gicp_init(L.stream, st + L.b0, h->guesses.as<float>() + (size_t)L.b0 * 16, L.B, L.prog);
    drive_lanes(lanes, max_rounds, [&](BatchLane& L) {
      const CloudDesc* dd = d_descs + L.b0;
      GicpState* ls = st + L.b0;
      if (round2) {
        {
          StageTimer tm(h, HGS_STAGE_LINEARIZE);
          launch_gicp_linearize_round2(L.stream, dd, tv, ls, st_other + L.b0, c, L.partials, L.partials_err, max_blocks, lin_blocks, L.B, lin_qpw, L.prog,
                                       h->results.as<DevResult>() + L.b0, early_out);
        }
        StageTimer tm(h, HGS_STAGE_ERROR);
        launch_gicp_error_round2(L.stream, dd, tv, st_other + L.b0, ls, c, L.partials, L.partials_err, max_blocks, err_blocks, L.B, lin_tile);
        return;
      }
      {
        StageTimer tm(h, HGS_STAGE_LINEARIZE);
        if (voxel) launch_vgicp_linearize(L.stream, dd, vtv, ls, vc, L.partials, max_blocks, L.B);
        else launch_gicp_linearize(L.stream, dd, tv, ls, c, L.partials, max_blocks, L.B, qpw);
      }
      {
        StageTimer tm(h, HGS_STAGE_SOLVE);
        launch_gicp_solve(L.stream, dd, ls, c, L.partials, max_blocks, L.B, voxel ? kBlock : nn_tile);
      }
      {
        StageTimer tm(h, HGS_STAGE_ERROR);
        if (voxel) launch_vgicp_error(L.stream, dd, vtv, ls, vc, L.partials_err, max_blocks, L.B);
        else launch_gicp_error(L.stream, dd, tv, ls, L.partials_err, max_blocks, L.B);
      }
      {
        StageTimer tm(h, HGS_STAGE_SOLVE);
        launch_gicp_decide(L.stream, dd, ls, c, L.partials_err, max_blocks, L.B, L.prog);
      }
    }, finish_lane, early ? (long)h->early_run_ahead : kRunAhead);
    HGS_TRY(close_lanes(h, lanes));
    if (h->early_valid) {  // rounds may still be queued (the host runs ahead): the next batch's mirror reset waits for them (make_progress)
      if (!h->early_event) HGS_HIP(h, hipEventCreateWithFlags(&h->early_event, hipEventDisableTiming));
      HGS_HIP(h, hipEventRecord(h->early_event, h->stream));
      h->early_pending = true;
    }
  } else {
    const NdtConsts c = ndt_consts(h->prm);
    HGS_HIP(h, h->states.reserve((size_t)B * sizeof(NdtState)));
    HGS_HIP(h, h->angles.reserve((size_t)B * sizeof(NdtAngles)));
    NdtState* st = h->states.as<NdtState>();
    NdtAngles* ang = h->angles.as<NdtAngles>();
    const NdtTargetView tv = ndt_target_view(h, tgt);
    // the sums are order-independent (hgs_ndt.h): read the sources in Hilbert order whenever they have a search index
    // (neighbouring lanes then share cells), in input order otherwise
    if (h->ndt_sort == 1) HGS_TRY(ensure_index(h, sources));
    bool sorted = h->ndt_sort != 0;
    for (hgs_cloud* sc : sources) sorted = sorted && sc->has_index;
    HGS_HIP(h, h->ndt_accum.reserve((size_t)B * sizeof(NdtAccum)));
    NdtAccum* accum = h->ndt_accum.as<NdtAccum>();  // zeroed by k_ndt_init (one dispatch less than a memset in front of it)
    // work plan of every lane: the (problem, tile) items of a pass are numbered by the prefix sums of the problems' tile counts
    // (from n_input: an upper bound of the finite points), and pulled from a queue head in HBM (k_ndt_pass)
    // one derivative pass per iteration as ndt_omp runs; up to 1 + 10 with the More-Thuente search
    const long max_rounds = ((long)c.max_iterations + 4) * (c.line_search ? 11 : 1);
    std::vector<BatchLane> lanes;
    HGS_TRY(open_lanes(h, B, (size_t)max_blocks * 2 * sizeof(double), (size_t)max_blocks * 2 * sizeof(double), lanes, err_blocks));
    auto finish_lane = [&](BatchLane& L) {
      launch_ndt_results(L.stream, d_descs + L.b0, st + L.b0, h->results.as<DevResult>() + L.b0, L.B);
      if (fit_max_range) lane_fitness(h, L, d_descs, *fit_max_range, max_blocks, qpw, nn_tile);
    };
    struct LanePlan {
      int* tile_base;
      unsigned long long* queues;  // two heads, used alternately (k_ndt_pass)
      int blocks, chunk;
    };
    std::vector<LanePlan> plans(lanes.size());
    {
      const size_t per_lane = align_up(16 + ((size_t)B + 1) * sizeof(int), 256);
      bool tile_overflow = false;
      constexpr long long kMaxNdtBlocks = 1 << 16, kMaxNdtChunk = 1 << 10;  // far above anything HGS_NDT_RESIDENT / HGS_NDT_CHUNK are used with
      HGS_HIP(h, h->ndt_plan.reserve(per_lane * lanes.size()));
      void* staged = nullptr;
      int plan_slot = 0;
      HGS_HIP(h, h->up.stage(per_lane * lanes.size(), &staged, &plan_slot));
      std::memset(staged, 0, per_lane * lanes.size());
      char* const host = static_cast<char*>(staged);
      for (size_t li = 0; li < lanes.size(); li++) {
        const BatchLane& L = lanes[li];
        int* tb = reinterpret_cast<int*>(host + li * per_lane + 16);
        tb[0] = 0;
        // k_ndt_pass does its item arithmetic in 32-bit ints (queue head + blocks * chunk must fit): bound the lane's tile count here
        long long tiles64 = 0;
        for (int k = 0; k < L.B; k++) {
          tiles64 += std::max<long long>(1, ((long long)sources[L.b0 + k]->n_input + kBlock - 1) / kBlock);
          tile_overflow = tile_overflow || tiles64 > (long long)INT_MAX - (long long)kMaxNdtBlocks * kMaxNdtChunk;
          tb[k + 1] = tile_overflow ? 0 : (int)tiles64;
        }
        LanePlan& P = plans[li];
        const int total = tb[L.B];
        if (tile_overflow) continue;
        P.blocks = std::max(1, std::min(total, h->ndt_resident_blocks > 0 ? std::min(h->ndt_resident_blocks, (int)kMaxNdtBlocks) : (lanes.size() > 1 ? 512 : 768)));
        // largest queue grab (the kernel sizes each grab by guided self-scheduling, at most this many items).  Fixed grabs
        // measured on the 16 x 119 k batch with 4 lanes: 1 -> 904, 2 -> 1062, 3 -> 980, 4 -> 936, 8 -> 845 registrations/s
        P.chunk = h->ndt_chunk > 0 ? std::min(h->ndt_chunk, (int)kMaxNdtChunk) : 8;
        P.queues = reinterpret_cast<unsigned long long*>((char*)h->ndt_plan.p + li * per_lane);
        P.tile_base = reinterpret_cast<int*>((char*)h->ndt_plan.p + li * per_lane + 16);
      }
      if (tile_overflow) {
        h->err = "NDT batch too large: the tiles of one lane do not fit 32-bit item arithmetic (more than ~5e11 source points in one call)";
        return HGS_ERR_INVALID_ARGUMENT;
      }
      HGS_HIP(h, hipMemcpyAsync(h->ndt_plan.p, host, per_lane * lanes.size(), hipMemcpyHostToDevice, h->stream));  // (round 4: a synchronous hipMemcpy of a pageable vector)
      HGS_HIP(h, h->up.commit(plan_slot, h->stream));
      // open_lanes() released lanes 1.. behind an event recorded BEFORE this copy: order them behind the plan as well, or a lane's first
      // k_ndt_init / k_ndt_pass could read the previous batch's tile_base / queue heads (round-5 advisor finding; tests/test_hip_parity.py
      // ::test_ndt_batches_of_different_shape_back_to_back poisons the plan between batches)
      if (lanes.size() > 1) {
        HGS_HIP(h, hipEventRecord(h->lane_event[0], h->stream));
        for (size_t i = 1; i < lanes.size(); i++) HGS_HIP(h, hipStreamWaitEvent(lanes[i].stream, h->lane_event[0], 0));
      }
    }
    for (BatchLane& L : lanes) launch_ndt_init(L.stream, st + L.b0, ang + L.b0, h->guesses.as<float>() + (size_t)L.b0 * 16, c, L.B, L.prog, accum + L.b0);
    drive_lanes(lanes, max_rounds, [&](BatchLane& L) {
      StageTimer tm(h, HGS_STAGE_LINEARIZE);
      const LanePlan& P = plans[&L - lanes.data()];
      launch_ndt_pass(L.stream, d_descs + L.b0, tv, st + L.b0, ang + L.b0, c, accum + L.b0, P.tile_base, P.queues, L.B, (int)(L.round & 1), P.blocks, P.chunk,
                      sorted ? 1 : 0, 0, L.prog);
    }, finish_lane);  // (NDT: one kernel per round, the Newton step inside it — nothing to stagger)
    HGS_TRY(close_lanes(h, lanes));
  }
  HGS_HIP(h, hipGetLastError());
  return HGS_OK;
}

// fitness of B sources against the target with the poses stored in h->results[b].T; fills fit_sum / fit_count
// use_corr_seeds: the sources' corr[] hold their correspondences against THIS target (hgs_fitness right behind a GICP align); otherwise the
// searches start from the target's seed grid
int run_fitness(hgs_handle* h, const std::vector<hgs_cloud*>& sources, double max_range, bool use_corr_seeds) {
  const int B = (int)sources.size();
  std::vector<hgs_cloud*> all(sources);
  all.push_back(h->target);
  HGS_TRY(ensure_index(h, all));
  if (!use_corr_seeds) HGS_TRY(ensure_seed_grid(h, h->target));
  int max_n = 0;
  for (hgs_cloud* c : sources) max_n = std::max(max_n, (int)c->n_input);
  const int qpw = 64;
  const int nn_tile = (kBlock / 64) * qpw * kNW;
  const int max_blocks = std::max(1, (max_n + nn_tile - 1) / nn_tile);
  const CloudDesc* d_descs = nullptr;
  HGS_TRY(upload_descs(h, sources, false, &d_descs, nullptr));
  HGS_HIP(h, h->partials_err.reserve((size_t)B * max_blocks * 2 * sizeof(double)));
  StageTimer tm(h, HGS_STAGE_FITNESS);
  launch_fitness(h->stream, d_descs, target_view(h->target), h->results.as<DevResult>(), max_range, h->partials_err.as<double>(), max_blocks, B,
                 use_corr_seeds ? 1 : 0, qpw);
  launch_fitness_final(h->stream, d_descs, h->partials_err.as<double>(), max_blocks, h->results.as<DevResult>(), B, nn_tile);
  HGS_HIP(h, hipGetLastError());
  return HGS_OK;
}

int fetch_results(hgs_handle* h, int B, std::vector<DevResult>& out) {
  HGS_HIP(h, h->h_results.reserve((size_t)B * sizeof(DevResult)));
  HGS_HIP(h, hipMemcpyAsync(h->h_results.p, h->results.p, (size_t)B * sizeof(DevResult), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  out.assign(h->h_results.as<DevResult>(), h->h_results.as<DevResult>() + B);
  return HGS_OK;
}

void to_public(const DevResult& d, int candidate, bool with_fitness, hgs_result* r) {
  std::memcpy(r->final_transformation, d.T, sizeof(float) * 16);
  r->converged = d.converged;
  r->iterations = d.iterations;
  r->error = d.error;
  r->lm_tries = d.lm_tries;
  r->candidate_id = candidate;
  r->reserved = 0;
  if (with_fitness) {
    r->num_inliers = d.fit_count;
    r->fitness_score = d.fit_count > 0 ? d.fit_sum / (double)d.fit_count : std::numeric_limits<double>::max();
  } else {
    r->num_inliers = 0;
    r->fitness_score = std::numeric_limits<double>::quiet_NaN();
  }
}

int upload_pose_as_result(hgs_handle* h, const float T[16]) {
  HGS_HIP(h, h->results.reserve(sizeof(DevResult)));
  HGS_HIP(h, h->h_results.reserve(sizeof(DevResult)));
  DevResult* r = h->h_results.as<DevResult>();
  std::memset(r, 0, sizeof(DevResult));
  std::memcpy(r->T, T, sizeof(float) * 16);
  HGS_HIP(h, hipMemcpyAsync(h->results.p, r, sizeof(DevResult), hipMemcpyHostToDevice, h->stream));
  return HGS_OK;
}

// The C-ABI never lets a C++ exception escape into PCL / ROS frames (include/hgs_registration.h): every entry point is a
// function-try-block whose handler ends here.  Called from inside a catch (...) handler.
int status_of_current_exception(hgs_handle* h) noexcept {
  int rc = HGS_ERR_INTERNAL;
  char what[256] = "unknown C++ exception inside the backend";
  try {
    throw;
  } catch (const std::bad_alloc&) {
    rc = HGS_ERR_OUT_OF_MEMORY;
    snprintf(what, sizeof(what), "out of host memory");
  } catch (const std::exception& e) {
    snprintf(what, sizeof(what), "C++ exception inside the backend: %s", e.what());
  } catch (...) {
  }
  try {
    if (h) h->err = what;
    else g_create_error = what;
  } catch (...) {
  }
  return rc;
}

}  // namespace

// =================================================================================================== C ABI
extern "C" {

int hgs_abi_version(void) { return HGS_ABI_VERSION; }

// Measurement / test knobs of one engine (A/B runs, tests that force a code path).  Rounds 1-5 read them from the environment in hgs_create; a plugin
// should not — the library now reads GPU_MAX_HW_QUEUES (HIP's own variable), HGS_COMM_TIMEOUT_MS and HGS_TRACE only.  The Python harness maps
// HGS_ENGINE_OPTIONS="key=value,..." onto this call (hdl_graph_slam_amd/registration.py); nothing in adapters/ uses it.
int hgs_debug_set_option(hgs_handle* h, const char* key, int value) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !key) return HGS_ERR_INVALID_ARGUMENT;
  const std::string k(key);
  if (k == "batch_lanes") h->batch_lanes = value <= 0 ? 0 : std::min(kMaxLanes, value);                  // 0: open_lanes chooses
  else if (k == "lane_start") h->lane_start = value != 0 ? 1 : 0;
  else if (k == "ndt_sort") h->ndt_sort = std::max(-1, std::min(1, value));
  else if (k == "cov_split") h->cov_split = value != 0 ? 1 : 0;
  else if (k == "fused_rounds") h->fused_rounds = value != 0 ? 1 : 0;
  else if (k == "early_result") h->early_result = value != 0 ? 1 : 0;
  else if (k == "early_run_ahead") h->early_run_ahead = std::max(1, std::min(8, value));
  else if (k == "fused_rounds_max_problems") h->fused_rounds_max_problems = std::max(0, value);
  else if (k == "fused_rounds_max_blocks") h->fused_rounds_max_blocks = std::max(0, value);
  else if (k == "fused_rounds_below") h->fused_rounds_below = std::max(0, value);
  else if (k == "resident_descs") h->resident_descs = value != 0 ? 1 : 0;
  else if (k == "knn_qpw_tiny") h->knn_qpw_tiny = value < 0 ? -1 : std::max(0, std::min(32, value & ~7));
  else if (k == "knn_tiny_below") h->knn_tiny_below = std::max(0, value);
  else if (k == "seed_grid") h->seed_grid = value != 0 ? 1 : 0;
  else if (k == "knn_replay") h->knn_replay = std::max(-1, std::min(2, value));                           // -1: the engine chooses
  else if (k == "ndt_resident") h->ndt_resident_blocks = std::max(0, value);
  else if (k == "ndt_chunk") h->ndt_chunk = std::max(0, value);
  else if (k == "hilbert_levels") h->hilbert_levels = std::max(4, std::min(16, value));
  else if (k == "nn_qpw") h->nn_qpw = value == 16 ? 16 : (value == 32 ? 32 : (value == 64 ? 64 : 0));
  else if (k == "nn_qpw16_below") h->nn_qpw16_below = std::max(0, value);
  else if (k == "nn_qpw32_below") h->nn_qpw32_below = std::max(0, value);
  else if (k == "upload_trace") h->upload_trace = value != 0 ? 1 : 0;
  else if (k == "prefilter_fast") h->prefilter_fast = value != 0 ? 1 : 0;
  else {
    h->err = "hgs_debug_set_option: unknown option '" + k + "'";
    return HGS_ERR_INVALID_ARGUMENT;
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_params_default(int32_t method, hgs_params* p) try {
  if (!p || method < HGS_FAST_GICP || method > HGS_NDT_OMP) return HGS_ERR_INVALID_ARGUMENT;
  std::memset(p, 0, sizeof(*p));
  p->method = method;
  p->max_iterations = 64;                 // reg_maximum_iterations            registrations.cpp:32,54,110
  p->transformation_epsilon = 0.01;       // reg_transformation_epsilon        registrations.cpp:31,53,109
  p->rotation_epsilon = 2e-3;             // fast_gicp LsqRegistration default
  p->max_correspondence_distance = method == HGS_FAST_GICP ? 2.5 : (double)FLT_MAX;  // registrations.cpp:33 (VGICP: not set)
  p->correspondence_randomness = 20;      // reg_correspondence_randomness     registrations.cpp:34,55
  p->neighbor_search = method == HGS_NDT_OMP ? HGS_DIRECT7 : HGS_DIRECT1;  // registrations.cpp:103
  p->resolution = method == HGS_NDT_OMP ? 0.5 : 1.0;                        // registrations.cpp:93 / :52
  p->ndt_step_size = 0.1;
  p->ndt_outlier_ratio = 0.55;
  p->ndt_min_points_per_voxel = 6;
  p->ndt_upstream_hd1_sign = 1;
  p->lm_max_iterations = 10;
  p->lm_init_lambda_factor = 1e-9;
  p->device_id = 0;
  p->regularization_method = HGS_REG_FROBENIUS;  // fast_gicp constructor default; hdl never calls the setter
  p->ndt_line_search = 0;                        // ndt_omp as it runs
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(nullptr);
}

int hgs_create(const hgs_params* p, hgs_handle** out) try {
  if (!p || !out) return HGS_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (p->method < HGS_FAST_GICP || p->method > HGS_NDT_OMP || p->max_iterations < 0 || p->correspondence_randomness < 1 || p->correspondence_randomness > 64 /* k_knn_cov's largest list */ || !(p->resolution > 0) ||
      p->regularization_method < HGS_REG_FROBENIUS || p->regularization_method > HGS_REG_NONE) {
    g_create_error = "invalid hgs_params";
    return HGS_ERR_INVALID_ARGUMENT;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || p->device_id < 0 || p->device_id >= ndev) {
    g_create_error = std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device ordinal out of range");
    return HGS_ERR_NO_DEVICE;
  }
  hgs_handle* h = new hgs_handle();
  h->prm = *p;
  h->device = p->device_id;
  for (int i = 0; i < 16; i++) h->final_T[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int i = 0; i < HGS_STAGE_COUNT; i++) h->prof_ms[i] = 0, h->prof_launches[i] = 0;
  // (measurement / test knobs: hgs_debug_set_option below — the library reads no tuning variable from the environment)
  if (hipSetDevice(h->device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    g_create_error = "hipSetDevice / hipStreamCreate failed";
    delete h;
    return HGS_ERR_HIP;
  }
  streams_in_use(h->device).fetch_add(1, std::memory_order_relaxed);
  *out = h;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(nullptr);
}

int hgs_destroy(hgs_handle* h) try {
  if (!h) return HGS_OK;
  // a call still running on another thread finishes first; a call that STARTS after this line races with the deletion below —
  // hgs_destroy must not be called concurrently with other calls on the engine or its clouds (include/hgs_registration.h)
  { std::lock_guard<std::recursive_mutex> wait_for_running_call(h->api_mutex); }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (hipStream_t ls : h->lane_stream)
    if (ls) (void)hipStreamSynchronize(ls);
  if (h->comm) hgs::comm_destroy(h->comm), h->comm = nullptr;
  if (h->own_target) cloud_free(h->target);
  if (h->own_source) cloud_free(h->source);
  // clouds the caller still holds (hgs_cloud_create / hgs_prefilter results, cached keyframes): their device memory goes with
  // the engine, the structs stay valid as orphans — hgs_cloud_destroy frees them, every other call rejects them
  for (hgs_cloud* c : h->live_clouds) {
    cloud_release_device(c, false);
    c->owner = nullptr;
    c->n_input = 0;
  }
  h->live_clouds.clear();
  DeviceBuffer* bufs[] = {&h->staging, &h->sort_keys[0], &h->sort_keys[1], &h->sort_vals[0], &h->sort_vals[1], &h->sort_tmp, &h->descs, &h->states,
                          &h->angles,  &h->partials,     &h->partials_err, &h->results,      &h->guesses,      &h->done,     &h->misc,
                          &h->pf_a,    &h->pf_b,         &h->pf_keep,      &h->pf_slot,      &h->pf_small,     &h->pf_dist,      &h->ndt_accum,    &h->ndt_plan,     &h->cov_raw,      &h->pf_ukeys,
                          &h->comm_send, &h->comm_recv,    &h->comm_ids};
  for (DeviceBuffer* b : bufs) b->release();
  for (int i = 0; i < kMaxLanes - 1; i++) h->lane_partials[i].release(), h->lane_partials_err[i].release();
  for (hipEvent_t ev : h->lane_event)
    if (ev) (void)hipEventDestroy(ev);
  if (h->comm_event) (void)hipEventDestroy(h->comm_event);
  if (h->early_event) (void)hipEventDestroy(h->early_event);
  for (hipStream_t ls : h->lane_stream)
    if (ls) (void)hipStreamDestroy(ls), streams_in_use(h->device).fetch_sub(1, std::memory_order_relaxed);
  for (auto& blk : h->block_pool) (void)hipFree(blk.first);
  h->block_pool.clear();
  h->up.release();
  h->up_points.release();
  h->pack_pool.shutdown();
  h->up_big.release();
  if (h->up_big_event) (void)hipEventDestroy(h->up_big_event);
  h->h_results.release();
  h->h_small.release();
  h->h_comm.release();
  h->h_flags.release();
  h->h_early.release();
  h->h_xform.release();
  for (auto& ev : h->prof_events) (void)hipEventDestroy(ev.a), (void)hipEventDestroy(ev.b);
  for (auto& ev : h->prof_free) (void)hipEventDestroy(ev.a), (void)hipEventDestroy(ev.b);
  if (h->stream) (void)hipStreamDestroy(h->stream), streams_in_use(h->device).fetch_sub(1, std::memory_order_relaxed);
  delete h;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(nullptr);
}

const char* hgs_last_error(const hgs_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int hgs_cloud_create(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes, hgs_cloud** out) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !out || (n > 0 && !pts) || stride_bytes < 12 || (stride_bytes % 4) != 0 || n > (size_t)1 << 30) return HGS_ERR_INVALID_ARGUMENT;
  HGS_TRY(set_device(h));
  const bool trace = h->upload_trace != 0;  // host-side phase times of an upload on stderr (diagnostics: hgs_debug_set_option "upload_trace")
  const auto t0 = std::chrono::steady_clock::now();
  hgs_cloud* c = nullptr;
  HGS_TRY(cloud_alloc(h, n, &c));
  const auto t1 = std::chrono::steady_clock::now();
  auto t2 = t1;
  {
    StageTimer tm(h, HGS_STAGE_UPLOAD);
    const float4* staged = nullptr;
    {
      const int rc = upload_points_packed(h, pts, n, stride_bytes, &staged);
      t2 = std::chrono::steady_clock::now();
      if (rc != HGS_OK) {
        cloud_free(c);
        return rc;
      }
    }
    const CloudDesc resident = resident_desc(c);
    launch_pack_aos(h->stream, staged, (int)n, const_cast<float4*>(c->desc.raw), c->intensity, c->desc.meta, &resident, c->dev_desc);  // (n == 0: the meta reset + descriptor alone)
    // nvalid + bounding box of the finite points
    launch_bbox_count(h->stream, c->dev_desc, 1, (int)n);  // (the meta record was reset and the descriptor written by the packing kernel)
    hipError_t e = hipGetLastError();
    // (no synchronisation: the caller's buffer was read by the host while packing; everything else is ordered on the stream)
    if (e != hipSuccess) {
      h->err = std::string("cloud upload failed: ") + hipGetErrorString(e);
      cloud_free(c);
      return HGS_ERR_HIP;
    }
  }
  if (trace) {
    const auto t3 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::fprintf(stderr, "[hgs upload] n %zu: alloc %.1f us, pack + DMA enqueue %.1f us, launches %.1f us\n", n, us(t0, t1), us(t1, t2), us(t2, t3));
  }
  *out = c;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_cloud_destroy(hgs_cloud* c) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (c && c->owner) api_lock__ = std::unique_lock<std::recursive_mutex>(c->owner->api_mutex);
  if (!c) return HGS_OK;
  hgs_handle* h = c->owner;
  if (h) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->target == c) h->target = nullptr, h->own_target = false;
    if (h->source == c) h->source = nullptr, h->own_source = false;
  }
  cloud_free(c);
  c = nullptr;  // gone: the handler below must not look at it
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(nullptr);
}

size_t hgs_cloud_size(const hgs_cloud* c) { return c ? c->n_input : 0; }

size_t hgs_cloud_device_bytes(const hgs_cloud* c) {
  if (!c || !c->owner) return 0;
  return (c->block ? c->block_bytes : 0) + (c->ndt_block ? c->ndt_block_bytes : 0) + (c->vg_block ? c->vg_block_bytes : 0) + (c->seed_block ? c->seed_block_bytes : 0);
}

int hgs_cloud_invalidate(hgs_cloud* c) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (c && c->owner) api_lock__ = std::unique_lock<std::recursive_mutex>(c->owner->api_mutex);
  if (!c) return HGS_ERR_INVALID_ARGUMENT;
  c->has_index = false, c->has_cov = false, c->has_ndt = false, c->has_vg = false, c->has_seed = false;
  // also forget the correspondences of earlier registrations (they seed the next search): a truly cold cloud.  No launch here — 65
  // memsets per loop-closure batch were 65 launches of ~3 us —: the kernel that rebuilds the cloud's sorted arrays clears them
  // (a cloud without an index cannot be searched before ensure_index has run).
  c->corr_stale = true;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception((c ? c->owner : nullptr));
}

int hgs_set_target_cloud(hgs_handle* h, hgs_cloud* c) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !c || c->owner != h) return HGS_ERR_INVALID_ARGUMENT;  // clouds belong to the engine (stream) that created them
  if (h->own_target && h->target != c) cloud_free(h->target);
  h->target = c;
  h->own_target = false;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}
int hgs_set_source_cloud(hgs_handle* h, hgs_cloud* c) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !c || c->owner != h) return HGS_ERR_INVALID_ARGUMENT;
  if (h->own_source && h->source != c) cloud_free(h->source);
  h->source = c;
  h->own_source = false;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}
int hgs_set_target(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h) return HGS_ERR_INVALID_ARGUMENT;
  hgs_cloud* c = nullptr;
  HGS_TRY(hgs_cloud_create(h, pts, n, stride_bytes, &c));
  if (h->own_target) cloud_free(h->target);
  h->target = c;
  h->own_target = true;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}
int hgs_set_source(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h) return HGS_ERR_INVALID_ARGUMENT;
  hgs_cloud* c = nullptr;
  HGS_TRY(hgs_cloud_create(h, pts, n, stride_bytes, &c));
  if (h->own_source) cloud_free(h->source);  // (its block is reused in stream order, or freed by hipFree, which waits by itself)
  h->source = c;
  h->own_source = true;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_align(hgs_handle* h, const float guess[16], hgs_result* out) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !guess || !out) return HGS_ERR_INVALID_ARGUMENT;
  if (!h->target) return HGS_ERR_NO_TARGET;
  if (!h->source) return HGS_ERR_NO_SOURCE;
  HGS_TRY(set_device(h));
  std::vector<hgs_cloud*> src{h->source};
  HGS_TRY(run_batch(h, src, guess));
  std::vector<DevResult> r;
  if (h->early_valid) r.assign(1, *h->h_early.as<DevResult>());  // (written in front of the `done` flag the poll has seen: k_gicp_linearize<true>)
  else HGS_TRY(fetch_results(h, 1, r));
  h->early_valid = false;
  to_public(r[0], 0, false, out);
  std::memcpy(h->final_T, out->final_transformation, sizeof(h->final_T));
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_transform_source(hgs_handle* h, const float T[16], void* out_pts, size_t stride_bytes) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !T || !out_pts || stride_bytes < 12) return HGS_ERR_INVALID_ARGUMENT;
  if (!h->source) return HGS_ERR_NO_SOURCE;
  HGS_TRY(set_device(h));
  const size_t n = h->source->n_input;
  if (n == 0) return HGS_OK;
  HGS_HIP(h, h->misc.reserve(n * sizeof(float4) + 256));
  HGS_HIP(h, h->h_small.reserve(64));
  float* dT = reinterpret_cast<float*>((char*)h->misc.p + n * sizeof(float4));
  std::memcpy(h->h_small.p, T, 64);
  HGS_HIP(h, hipMemcpyAsync(dT, h->h_small.p, 64, hipMemcpyHostToDevice, h->stream));
  launch_transform(h->stream, h->source->desc.raw, (int)n, dT, h->misc.as<float4>());
  // Down through a pinned staging buffer the engine keeps (round 4 copied into a fresh pageable std::vector: 1.9 MB of page faults and a staged
  // pageable D2H per align): one copy, then the scatter into the caller's strided records — for a large cloud a write-allocate pass over 3.8 MB, spread
  // over the pack pool's threads.  (Four overlapped pieces with an event each were tried first: at 13 k points their API calls cost more than they hid.)
  HGS_HIP(h, h->h_xform.reserve(n * sizeof(float4)));
  const float* host = h->h_xform.as<float>();
  char* o = (char*)out_pts;
  HGS_HIP(h, hipMemcpyAsync(h->h_xform.p, h->misc.p, n * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  const size_t nchunks = (n + kUploadChunkPoints - 1) / kUploadChunkPoints;
  std::vector<std::atomic<unsigned char>> ready(nchunks);
  for (auto& r : ready) r.store(0, std::memory_order_relaxed);
  PackPool::Job job;
  job.src = reinterpret_cast<const char*>(host), job.dst = reinterpret_cast<float*>(o), job.n = n, job.stride = stride_bytes, job.chunk = kUploadChunkPoints;
  job.nchunks = nchunks, job.scatter = true, job.ready = ready.data();
  if (n >= kUploadParallelPoints) {
    h->pack_pool.post(&job);
    while (PackPool::help(job)) {
    }
    h->pack_pool.retire();  // every chunk has been taken, and nobody is inside the job any more: all of them are done
  } else {
    while (PackPool::help(job)) {  // a small cloud: this thread alone
    }
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_fitness(hgs_handle* h, const float T[16], double max_range, double* score, uint32_t* num_inliers) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !T || !score) return HGS_ERR_INVALID_ARGUMENT;
  if (!h->target) return HGS_ERR_NO_TARGET;
  if (!h->source) return HGS_ERR_NO_SOURCE;
  HGS_TRY(set_device(h));
  HGS_TRY(upload_pose_as_result(h, T));
  std::vector<hgs_cloud*> src{h->source};
  HGS_TRY(run_fitness(h, src, max_range, h->prm.method == HGS_FAST_GICP));
  std::vector<DevResult> r;
  HGS_TRY(fetch_results(h, 1, r));
  *score = r[0].fit_count > 0 ? r[0].fit_sum / (double)r[0].fit_count : std::numeric_limits<double>::max();
  if (num_inliers) *num_inliers = r[0].fit_count;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_calc_fitness_score(hgs_handle* h, hgs_cloud* cloud1, hgs_cloud* cloud2, const float relpose[16], double max_range, double* score) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !cloud1 || !cloud2 || cloud1->owner != h || cloud2->owner != h || !relpose || !score) return HGS_ERR_INVALID_ARGUMENT;
  HGS_TRY(set_device(h));
  hgs_cloud* saved_t = h->target;
  h->target = cloud1;
  int rc = upload_pose_as_result(h, relpose);
  std::vector<hgs_cloud*> src{cloud2};
  if (rc == HGS_OK) rc = run_fitness(h, src, max_range, false);  // two keyframe clouds: cloud2 holds no correspondences against cloud1
  std::vector<DevResult> r;
  if (rc == HGS_OK) rc = fetch_results(h, 1, r);
  h->target = saved_t;
  if (rc != HGS_OK) return rc;
  *score = r[0].fit_count > 0 ? r[0].fit_sum / (double)r[0].fit_count : std::numeric_limits<double>::max();
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_nn_target(hgs_handle* h, const float* q_xyz, size_t nq, size_t stride_bytes, int32_t* idx, float* d2) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || (nq > 0 && (!q_xyz || !idx || !d2)) || stride_bytes < 12) return HGS_ERR_INVALID_ARGUMENT;
  if (!h->target) return HGS_ERR_NO_TARGET;
  if (nq == 0) return HGS_OK;
  HGS_TRY(set_device(h));
  std::vector<hgs_cloud*> all{h->target};
  HGS_TRY(ensure_index(h, all));
  HGS_HIP(h, h->misc.reserve(nq * (sizeof(float4) + 8) + 512));
  float4* dq = h->misc.as<float4>();
  int* didx = reinterpret_cast<int*>((char*)h->misc.p + align_up(nq * sizeof(float4), 256));
  float* dd2 = reinterpret_cast<float*>(didx + nq);
  const float4* staged = nullptr;
  HGS_TRY(upload_points_packed(h, q_xyz, nq, stride_bytes, &staged));
  launch_pack_aos(h->stream, staged, (int)nq, dq, nullptr, nullptr);
  launch_nn_query(h->stream, target_view(h->target), dq, (int)nq, didx, dd2);
  // the two result arrays come down in ONE copy into pinned memory (they are adjacent on the device: didx[nq] | dd2[nq]) and are handed to the
  // caller's (pageable) arrays from there
  HGS_HIP(h, h->h_xform.reserve(nq * 8));
  HGS_HIP(h, hipMemcpyAsync(h->h_xform.p, didx, nq * 8, hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  std::memcpy(idx, h->h_xform.p, nq * sizeof(int));
  std::memcpy(d2, static_cast<const char*>(h->h_xform.p) + nq * sizeof(int), nq * sizeof(float));
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_select_best(const hgs_result* records, size_t n, int32_t* best) try {
  if (!best || (n > 0 && !records)) return HGS_ERR_INVALID_ARGUMENT;
  // loop_detector.hpp:124,146-153: best_score starts at DBL_MAX; skip if !converged || score > best_score; else replace
  double best_score = std::numeric_limits<double>::max();
  int32_t b = -1;
  for (size_t i = 0; i < n; i++) {
    const double score = records[i].fitness_score;
    if (!records[i].converged || score > best_score || score != score) continue;
    best_score = score;
    b = (int32_t)i;
  }
  *best = b;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(nullptr);
}

int hgs_loop_match_batch(hgs_handle* h, hgs_cloud* const* candidates, size_t n_candidates, const float* guesses, double max_range, hgs_result* out,
                         int32_t* best) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || (n_candidates > 0 && (!candidates || !guesses || !out))) return HGS_ERR_INVALID_ARGUMENT;
  if (!h->target) return HGS_ERR_NO_TARGET;
  if (best) *best = -1;
  if (n_candidates == 0) return HGS_OK;
  HGS_TRY(set_device(h));
  std::vector<hgs_cloud*> src(candidates, candidates + n_candidates);
  for (hgs_cloud* c : src)
    if (!c || c->owner != h) return HGS_ERR_INVALID_ARGUMENT;
  {
    // every candidate carries its own correspondence scratch: the same cloud twice in one batch would race on it
    std::vector<hgs_cloud*> sorted(src);
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
      h->err = "hgs_loop_match_batch: candidate clouds must be distinct";
      return HGS_ERR_INVALID_ARGUMENT;
    }
  }
  HGS_TRY(run_batch(h, src, guesses, &max_range));
  std::vector<DevResult> r;
  HGS_TRY(fetch_results(h, (int)n_candidates, r));
  for (size_t i = 0; i < n_candidates; i++) to_public(r[i], (int)i, true, &out[i]);
  if (best) HGS_TRY(hgs_select_best(out, n_candidates, best));
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

// ---- multi-GPU: candidate-sharded loop-closure batch, one process per GPU ---------------------------------------
int hgs_comm_get_unique_id(void* id_out) try {
  if (!id_out) return HGS_ERR_INVALID_ARGUMENT;
  char err[256] = "";
  if (hgs::comm_unique_id(id_out, err, sizeof(err)) != 0) {
    g_create_error = err;
    return HGS_ERR_COMM;
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(nullptr);
}

int hgs_comm_init(hgs_handle* h, int32_t rank, int32_t world, const void* unique_id) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !unique_id || world < 1 || rank < 0 || rank >= world) return HGS_ERR_INVALID_ARGUMENT;
  HGS_TRY(set_device(h));
  if (h->comm) hgs::comm_destroy(h->comm), h->comm = nullptr;
  char err[256] = "";
  if (hgs::comm_create(&h->comm, rank, world, unique_id, h->device, err, sizeof(err)) != 0) {
    h->err = err;
    return HGS_ERR_COMM;
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_comm_finalize(hgs_handle* h) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h) return HGS_ERR_INVALID_ARGUMENT;
  if (h->comm) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    hgs::comm_destroy(h->comm);
    h->comm = nullptr;
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

// Pure host code of the sharded batch: the gathered slots -> the n_total records of the detection.  `gathered` holds `world`
// blocks of `per` slots; of rank r's block the first counts[r] slots carry records, the rest is padding.  A candidate nobody
// reported stays "not converged" (fitness DBL_MAX); a candidate reported twice (two ranks, or twice by one) is an error of the
// caller's partition: *duplicate_id receives the id (else -1) and the first report is kept.
int hgs_debug_merge_shard_records(const hgs_result* gathered, const int32_t* counts, int32_t world, size_t per, size_t n_total, hgs_result* all_out,
                                  int32_t* duplicate_id) try {
  if (!all_out || n_total == 0 || world < 1 || (per > 0 && !gathered) || !counts) return HGS_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < n_total; i++) {
    std::memset(&all_out[i], 0, sizeof(hgs_result));
    all_out[i].candidate_id = (int32_t)i;
    all_out[i].fitness_score = std::numeric_limits<double>::max();
  }
  std::vector<char> seen(n_total, 0);
  int32_t dup = -1;
  for (int32_t r = 0; r < world; r++) {
    const size_t n = counts[r] < 0 ? 0 : std::min<size_t>((size_t)counts[r], per);
    for (size_t k = 0; k < n; k++) {
      const hgs_result& rec = gathered[(size_t)r * per + k];
      if (rec.candidate_id < 0 || (size_t)rec.candidate_id >= n_total) continue;  // padding (a rank that failed after announcing its shard)
      if (seen[rec.candidate_id]) {
        if (dup < 0) dup = rec.candidate_id;
        continue;
      }
      seen[rec.candidate_id] = 1;
      all_out[rec.candidate_id] = rec;
    }
  }
  if (duplicate_id) *duplicate_id = dup;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(nullptr);
}

}  // extern "C"
namespace {
// Waits for something enqueued behind a collective (query = hipEventQuery / hipStreamQuery) WITHOUT an unbounded blocking call.  A rank that aborts
// its communicator only raises its own abort flag: on the intra-node transports (P2P / xGMI / shared memory) RCCL does not promise that the peers'
// all-gather kernel stops spinning, so hipEventSynchronize / hipStreamSynchronize behind it could wait forever.  The wait therefore polls, asks
// the communicator for an asynchronous error (ncclCommGetAsyncError) now and then, and gives up after the deadline (HGS_COMM_TIMEOUT_MS, default
// 60 s, 0 = none; it has to cover the skew with which the SLAM processes enter the detection, not the exchange itself, which is microseconds).
// On error or timeout this rank aborts ITS communicator (which ends its own stuck kernel) and the call returns HGS_ERR_COMM.
long comm_timeout_ms() {
  if (const char* e = std::getenv("HGS_COMM_TIMEOUT_MS")) return std::max(0l, std::atol(e));
  return 60000;
}
template <typename Q>
int comm_wait(hgs_handle* h, Q&& query, const char* what) {
  const long limit_ms = comm_timeout_ms();
  const auto t0 = std::chrono::steady_clock::now();
  auto next_check = t0 + std::chrono::milliseconds(1);
  long sleep_us = 0;  // spin for ~200 us, then sleep with exponential backoff up to 50 us: a rank waiting out peer skew must not burn the core that feeds the
                      // other lanes and engines, but this wait also sits behind the batch of EVERY sharded detection, where it overshoots by half its
                      // last sleep on average — hence 50 us and not the millisecond a pure skew wait could afford
  char err[256] = "";
  for (;;) {
    const hipError_t e = query();
    if (e == hipSuccess) return HGS_OK;
    if (e != hipErrorNotReady) {
      h->err = std::string("hgs_loop_match_batch_sharded: HIP error while waiting for ") + what + ": " + hipGetErrorString(e);
      hgs::comm_abort(h->comm);
      return HGS_ERR_HIP;
    }
    const auto now = std::chrono::steady_clock::now();
    if (now >= next_check) {  // time-based: once a millisecond, whatever the polling rate
      next_check = now + std::chrono::milliseconds(1);
      if (hgs::comm_async_error(h->comm, err, sizeof(err)) != 0) {
        h->err = std::string("hgs_loop_match_batch_sharded: the communicator reported an asynchronous error while waiting for ") + what + ": " + err;
        hgs::comm_abort(h->comm);
        return HGS_ERR_COMM;
      }
      const long waited = (long)std::chrono::duration_cast<std::chrono::milliseconds>(now - t0).count();
      if (limit_ms > 0 && waited > limit_ms) {
        h->err = std::string("hgs_loop_match_batch_sharded: gave up after ") + std::to_string(waited) + " ms waiting for " + what +
                 " (a peer has left the collective or never entered it); the communicator has been aborted";
        hgs::comm_abort(h->comm);
        return HGS_ERR_COMM;
      }
    }
    if (now - t0 > std::chrono::microseconds(200)) {
      sleep_us = sleep_us == 0 ? 5 : std::min(50l, sleep_us * 2);
      std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));
    }
  }
}
// Between the two collectives of a sharded batch the peers already count on this rank: whatever leaves that region without having enqueued the
// record all-gather — an exception nobody expected — must not leave them waiting for it.
struct CommAbortGuard {
  hgs::Comm* comm;
  bool armed = true;
  ~CommAbortGuard() {
    if (armed) hgs::comm_abort(comm);
  }
};
// test hook, compiled ONLY into the host-emulation build (tests/emul/simt.py passes -DHGS_TESTING; the shipped library has neither the code nor
// the string): HGS_FAULT_AFTER_HEADER = "<kind>:<rank>" makes that rank fail between the two collectives — kind "batch": std::bad_alloc where
// run_batch runs (handled: the rank sends padding and reports after the exchange); kind "guard": an exception outside every handler (the guard
// aborts the communicator, the peers get HGS_ERR_COMM)
#ifdef HGS_TESTING
bool fault_after_header(const char* kind, int rank) {
  const char* e = std::getenv("HGS_FAULT_AFTER_HEADER");
  if (!e) return false;
  const size_t n = std::strlen(kind);
  return std::strncmp(e, kind, n) == 0 && e[n] == ':' && std::atoi(e + n + 1) == rank;
}
#else
constexpr bool fault_after_header(const char*, int) { return false; }
#endif
}  // namespace
extern "C" {

// Collective.  Once the arguments that are the same on every rank have been checked, EVERY path of a rank reaches both
// collectives — a rank whose own share is unusable (no target, an invalid or duplicated candidate cloud, a failed launch)
// announces an empty shard / sends padding and reports its error AFTER the exchange, so that a bad keyframe on one rank costs
// that rank's candidates ("not converged" records everywhere) and never blocks the other SLAM processes.  The one thing a rank
// cannot do is take part without device memory for the exchange buffers: it then aborts the communicator (ncclCommAbort), which
// makes the peers' collectives fail with HGS_ERR_COMM instead of hanging.
//   1. all-gather of a 16-byte header per rank {shard size, status}, enqueued in front of the batch's kernels;
//   2. the batch (run_batch), while the headers travel;
//   3. all-gather of max(shard size) record slots per rank — not n_total: an even partition of 512 candidates over 8 ranks moves
//      8 x 64 x 112 B = 57 KB instead of 458 KB — built on the device behind the kernels that produced the results;
//   4. one D2H of world x max(shard) slots, merge (hgs_debug_merge_shard_records), sequential selection rule.
int hgs_loop_match_batch_sharded(hgs_handle* h, hgs_cloud* const* candidates, size_t n_mine, const int32_t* candidate_ids, const float* guesses,
                                 size_t n_total, double max_range, hgs_result* all_out, int32_t* best) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  // ---- errors that keep the rank out of the collective: only what a correct caller gets wrong on every rank alike
  if (!h || !all_out || n_total == 0 || n_total > (size_t)1 << 24) return HGS_ERR_INVALID_ARGUMENT;
  if (!h->comm) {
    h->err = "hgs_loop_match_batch_sharded: hgs_comm_init has not been called on this engine";
    return HGS_ERR_COMM;
  }
  if (best) *best = -1;
  const int world = hgs::comm_world(h->comm), rank = hgs::comm_rank(h->comm);
  // ---- this rank's own share: problems found here are reported after the exchange
  int local = HGS_OK;
  std::string local_err;
  auto fail_local = [&](int rc, const std::string& why) {
    if (local == HGS_OK) local = rc, local_err = why;
  };
  if (n_mine > n_total || (n_mine > 0 && (!candidates || !candidate_ids || !guesses))) fail_local(HGS_ERR_INVALID_ARGUMENT, "hgs_loop_match_batch_sharded: bad candidate arrays");
  if (!h->target) fail_local(HGS_ERR_NO_TARGET, "hgs_loop_match_batch_sharded: no target set on this rank");
  std::vector<hgs_cloud*> src;
  if (local == HGS_OK) {
    src.assign(candidates, candidates + n_mine);
    for (size_t i = 0; i < n_mine; i++)
      if (!src[i] || src[i]->owner != h || candidate_ids[i] < 0 || (size_t)candidate_ids[i] >= n_total)
        fail_local(HGS_ERR_INVALID_ARGUMENT, "hgs_loop_match_batch_sharded: candidate " + std::to_string(i) + " is not a cloud of this engine or its id is out of range");
    std::vector<hgs_cloud*> sorted(src);
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) fail_local(HGS_ERR_INVALID_ARGUMENT, "hgs_loop_match_batch_sharded: candidate clouds must be distinct");
  }
  const size_t n_send = local == HGS_OK ? n_mine : 0;
  // ---- exchange buffers (persistent, worst case: one rank holds every candidate)
  const size_t hdr_bytes = 16;
  const size_t ids_off = align_up((size_t)world * hdr_bytes + hdr_bytes, 16), rec_off = align_up(ids_off + n_total * sizeof(int32_t), 16);
  bool mem_ok = set_device(h) == HGS_OK;
  mem_ok = mem_ok && h->results.reserve(std::max<size_t>(n_total, 1) * sizeof(DevResult)) == hipSuccess;
  mem_ok = mem_ok && h->comm_send.reserve(n_total * sizeof(hgs_result) + hdr_bytes) == hipSuccess;
  mem_ok = mem_ok && h->comm_recv.reserve((size_t)world * (n_total * sizeof(hgs_result) + hdr_bytes)) == hipSuccess;
  mem_ok = mem_ok && h->comm_ids.reserve(n_total * sizeof(int32_t)) == hipSuccess;
  mem_ok = mem_ok && h->h_comm.reserve(rec_off + (size_t)world * n_total * sizeof(hgs_result)) == hipSuccess;
  if (!mem_ok) {
    hgs::comm_abort(h->comm);
    h->err = "hgs_loop_match_batch_sharded: no memory for the exchange buffers; the communicator has been aborted (the peers' collective fails instead of waiting)";
    return HGS_ERR_OUT_OF_MEMORY;
  }
  char err[256] = "";
  auto comm_failed = [&](const char* what) {
    h->err = std::string(what) + ": " + err;
    hgs::comm_abort(h->comm);  // whatever state the peers are in, they must not wait for this rank
    return HGS_ERR_COMM;
  };
  // everything the region between the two collectives needs is allocated HERE, in front of the first one: past this point the
  // peers count on this rank, and the only exits are the record all-gather or an abort of the communicator
  std::vector<int32_t> counts(world, 0), statuses(world, 0);
  // ---- 1. headers: device layout comm_send = [header | records], comm_recv = [world headers | world x per records]
  int32_t* h_hdr = static_cast<int32_t*>(h->h_comm.p);  // pinned: [my header][world gathered headers]
  h_hdr[0] = (int32_t)n_send, h_hdr[1] = local, h_hdr[2] = 0, h_hdr[3] = 0;
  char* d_send = static_cast<char*>(h->comm_send.p);
  char* d_recv = static_cast<char*>(h->comm_recv.p);
  bool hip_ok = hipMemcpyAsync(d_send, h_hdr, hdr_bytes, hipMemcpyHostToDevice, h->stream) == hipSuccess;
  if (hgs::comm_all_gather(h->comm, d_send, d_recv, hdr_bytes, h->stream, err, sizeof(err)) != 0) return comm_failed("header all-gather");
  CommAbortGuard guard{h->comm};  // from here to the enqueued record all-gather
  hip_ok = hip_ok && hipMemcpyAsync(h_hdr + 4, d_recv, (size_t)world * hdr_bytes, hipMemcpyDeviceToHost, h->stream) == hipSuccess;
  if (!h->comm_event) hip_ok = hip_ok && hipEventCreateWithFlags(&h->comm_event, hipEventDisableTiming) == hipSuccess;
  hip_ok = hip_ok && hipEventRecord(h->comm_event, h->stream) == hipSuccess;
  if (fault_after_header("guard", rank)) throw std::runtime_error("HGS_FAULT_AFTER_HEADER=guard");
  // ---- 2. the batch.  A C++ exception in here (std::bad_alloc of a host-side vector, ...) is this rank's own problem like any other
  // failure of its share: it sends padding and reports after the exchange
  size_t n_valid = 0;  // records this rank really has
  if (n_send > 0 && hip_ok) {
    int rc;
    try {
      if (fault_after_header("batch", rank)) throw std::bad_alloc();
      rc = run_batch(h, src, guesses, &max_range);
    } catch (...) {
      rc = status_of_current_exception(h);
    }
    if (rc == HGS_OK) {
      n_valid = n_send;
    } else if (local == HGS_OK) {
      local = rc;
      try {
        local_err = h->err;
      } catch (...) {  // no memory for the text: the status code stands
      }
    }
  }
  // ---- 3. records: everybody knows everybody's shard size now
  if (hip_ok) {
    const int rc = comm_wait(h, [&] { return hipEventQuery(h->comm_event); }, "the gathered shard headers");
    if (rc != HGS_OK) {
      guard.armed = false;  // comm_wait has aborted the communicator
      return rc;
    }
  }
  size_t per = 1;
  if (hip_ok)
    for (int r = 0; r < world; r++) {
      counts[r] = std::max(0, std::min<int32_t>(h_hdr[4 + 4 * r], (int32_t)n_total));
      statuses[r] = h_hdr[4 + 4 * r + 1];
      per = std::max(per, (size_t)counts[r]);
    }
  if (!hip_ok) {  // the header never arrived here: this rank cannot know `per`; the guard aborts the communicator
    h->err = std::string("hgs_loop_match_batch_sharded: HIP error around the header exchange: ") + hipGetErrorString(hipGetLastError());
    return HGS_ERR_HIP;
  }
  hgs_result* d_records = reinterpret_cast<hgs_result*>(d_send + hdr_bytes);
  if (n_valid > 0) {
    std::memcpy(static_cast<char*>(h->h_comm.p) + ids_off, candidate_ids, n_valid * sizeof(int32_t));
    if (hipMemcpyAsync(h->comm_ids.p, static_cast<char*>(h->h_comm.p) + ids_off, n_valid * sizeof(int32_t), hipMemcpyHostToDevice, h->stream) != hipSuccess) n_valid = 0;
  }
  launch_results_to_records(h->stream, h->results.as<DevResult>(), h->comm_ids.as<int>(), (int)n_valid, (int)per, d_records);  // padding beyond n_valid
  hgs_result* d_gathered = reinterpret_cast<hgs_result*>(d_recv + (size_t)world * hdr_bytes);
  if (hgs::comm_all_gather(h->comm, d_records, d_gathered, per * sizeof(hgs_result), h->stream, err, sizeof(err)) != 0) {
    guard.armed = false;  // comm_failed aborts
    return comm_failed("record all-gather");
  }
  guard.armed = false;  // both collectives are enqueued: the peers depend on nothing further from this rank
  // ---- 4. merge
  hgs_result* gathered = reinterpret_cast<hgs_result*>(static_cast<char*>(h->h_comm.p) + rec_off);
  HGS_HIP(h, hipMemcpyAsync(gathered, d_gathered, (size_t)world * per * sizeof(hgs_result), hipMemcpyDeviceToHost, h->stream));
  HGS_TRY(comm_wait(h, [&] { return hipStreamQuery(h->stream); }, "the gathered records"));
  int32_t dup = -1;
  HGS_TRY(hgs_debug_merge_shard_records(gathered, counts.data(), world, per, n_total, all_out, &dup));
  if (best) HGS_TRY(hgs_select_best(all_out, n_total, best));
  if (local != HGS_OK) {
    h->err = local_err;
    return local;
  }
  if (dup >= 0) {  // every rank sees the same gathered slots, so every rank returns this
    h->err = "hgs_loop_match_batch_sharded: candidate id " + std::to_string(dup) + " was reported more than once (the first report was kept)";
    return HGS_ERR_INVALID_ARGUMENT;
  }
  h->err.clear();
  for (int r = 0; r < world; r++)
    if (statuses[r] != HGS_OK && r != rank) h->err += "rank " + std::to_string(r) + " reported status " + std::to_string(statuses[r]) + " (its candidates are not converged); ";
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

// ---- prefilter (apps/prefiltering_nodelet.cpp:131-182) -------------------------------------------------------
namespace {

int scan_u32(hgs_handle* h, const uint32_t* in, uint32_t* out, size_t n) {
  size_t tmp = 0;
  if (hgs_exclusive_scan_u32(nullptr, &tmp, in, out, n, h->stream) != 0) {
    h->err = "rocprim exclusive_scan (size query) failed";
    return HGS_ERR_HIP;
  }
  HGS_HIP(h, h->sort_tmp.reserve(tmp));
  if (hgs_exclusive_scan_u32(h->sort_tmp.p, &tmp, in, out, n, h->stream) != 0) {
    h->err = "rocprim exclusive_scan failed";
    return HGS_ERR_HIP;
  }
  return HGS_OK;
}

// a resident cloud from a device array of {x, y, z, intensity}
int cloud_from_device(hgs_handle* h, const float4* src, size_t m, hgs_cloud** out) {
  hgs_cloud* c = nullptr;
  HGS_TRY(cloud_alloc(h, m, &c));
  const CloudDesc resident = resident_desc(c);
  launch_pf_to_cloud(h->stream, src, (int)m, const_cast<float4*>(c->desc.raw), c->intensity, c->desc.meta, &resident, c->dev_desc);
  launch_bbox_count(h->stream, c->dev_desc, 1, (int)m);  // (the meta record was reset and the descriptor written by k_pf_to_cloud)
  hipError_t e = hipGetLastError();
  // (no synchronisation: `src` is a device array of this engine, read in stream order)
  if (e != hipSuccess) {
    h->err = std::string("prefilter: building the resident cloud failed: ") + hipGetErrorString(e);
    cloud_free(c);
    return HGS_ERR_HIP;
  }
  *out = c;
  return HGS_OK;
}

}  // namespace

extern "C" int hgs_prefilter_params_default(hgs_prefilter_params* p) try {
  if (!p) return HGS_ERR_INVALID_ARGUMENT;
  std::memset(p, 0, sizeof(*p));
  p->use_distance_filter = 1;          // prefiltering_nodelet.cpp:94
  p->distance_near_thresh = 1.0;       // :95
  p->distance_far_thresh = 100.0;      // :96
  p->downsample_method = HGS_DOWNSAMPLE_VOXELGRID;   // :52
  p->downsample_resolution = 0.1;      // :53
  p->outlier_removal_method = HGS_OUTLIER_STATISTICAL;  // :73
  p->statistical_mean_k = 20;          // :75
  p->statistical_stddev = 1.0;         // :76
  p->radius_radius = 0.8;              // :85
  p->radius_min_neighbors = 2;         // :86
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(nullptr);
}

static int prefilter_impl(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes, const hgs_prefilter_params* p, const float* deskew_w, double scan_period,
                          hgs_cloud** out) {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !p || !out || (n > 0 && !pts) || stride_bytes < 12 || (stride_bytes % 4) != 0 || n > (size_t)1 << 30) return HGS_ERR_INVALID_ARGUMENT;
  if (p->downsample_method < HGS_DOWNSAMPLE_NONE || p->downsample_method > HGS_DOWNSAMPLE_APPROX_VOXELGRID || p->outlier_removal_method < HGS_OUTLIER_NONE ||
      p->outlier_removal_method > HGS_OUTLIER_RADIUS || (p->downsample_method != HGS_DOWNSAMPLE_NONE && !(p->downsample_resolution > 0)) ||
      (p->outlier_removal_method == HGS_OUTLIER_STATISTICAL && (p->statistical_mean_k < 1 || p->statistical_mean_k > 62)) ||
      (p->outlier_removal_method == HGS_OUTLIER_RADIUS && (!(p->radius_radius > 0) || p->radius_min_neighbors < 0)))
    return HGS_ERR_INVALID_ARGUMENT;
  HGS_TRY(set_device(h));
  *out = nullptr;
  StageTimer tm(h, HGS_STAGE_PREFILTER);
  const size_t cap = std::max<size_t>(n, 1);
  HGS_HIP(h, h->pf_a.reserve(cap * sizeof(float4)));
  HGS_HIP(h, h->pf_b.reserve(cap * sizeof(float4)));
  HGS_HIP(h, h->pf_keep.reserve(cap * sizeof(uint32_t)));
  HGS_HIP(h, h->pf_slot.reserve(cap * sizeof(uint32_t)));
  HGS_HIP(h, h->pf_small.reserve(256));
  HGS_HIP(h, h->h_small.reserve(256));
  float4* cur = h->pf_a.as<float4>();
  float4* other = h->pf_b.as<float4>();
  bool grid_radius = false;  // the outlier removal ran on the voxel grid (below): nothing left to do behind the read-back
  int* d_count = h->pf_small.as<int>();                    // [0] current point count
  unsigned* d_meta = h->pf_small.as<unsigned>() + 16;      // voxel-grid bbox / grid parameters
  double* d_stats = reinterpret_cast<double*>(h->pf_small.as<char>() + 192);
  {
    const float4* staged = nullptr;
    if (n > 0) HGS_TRY(upload_points_packed(h, pts, n, stride_bytes, &staged));
    launch_pf_load(h->stream, staged, (int)n, cur, deskew_w, scan_period, d_count, d_meta);  // (also: d_count = n, d_meta = the empty voxel-grid record)
  }
  // VoxelGrid behind the distance filter (every launch file's prefilter): the filter is applied inside the voxel grid's bounding-box and key kernels
  // instead of by flags + scan + compaction in front of them (four launches less; prefilter_fast = 0 keeps the separate pass — A/B, tests)
  const bool voxelgrid = n > 0 && p->downsample_method == HGS_DOWNSAMPLE_VOXELGRID;
  const int inline_dist = (voxelgrid && p->use_distance_filter && h->prefilter_fast) ? 1 : 0;
  if (n > 0 && p->use_distance_filter && !inline_dist) {
    launch_pf_distance_flags(h->stream, cur, (int)n, 1, p->distance_near_thresh, p->distance_far_thresh, h->pf_keep.as<unsigned>());
    HGS_TRY(scan_u32(h, h->pf_keep.as<uint32_t>(), h->pf_slot.as<uint32_t>(), n));
    launch_pf_compact(h->stream, cur, (int)n, h->pf_keep.as<unsigned>(), h->pf_slot.as<unsigned>(), other, d_count);
    std::swap(cur, other);
  }
  if (n > 0 && p->downsample_method == HGS_DOWNSAMPLE_VOXELGRID) {
    const float inv_leaf = 1.0f / (float)p->downsample_resolution;
    launch_pf_bbox(h->stream, cur, d_count, (int)n, d_meta, inline_dist, p->distance_near_thresh, p->distance_far_thresh);
    launch_pf_grid(h->stream, d_meta, inv_leaf);
    for (int i = 0; i < 2; i++) {
      HGS_HIP(h, h->sort_keys[i].reserve(n * sizeof(uint64_t)));
      HGS_HIP(h, h->sort_vals[i].reserve(n * sizeof(uint32_t)));
    }
    launch_pf_voxel_keys(h->stream, cur, d_count, d_meta, inv_leaf, (int)n, h->sort_keys[0].as<unsigned long long>(), h->sort_vals[0].as<unsigned>(), inline_dist,
                         p->distance_near_thresh, p->distance_far_thresh);
    size_t tmp_bytes = 0;
    int rc = hgs_sort_pairs_u64_u32(nullptr, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                    h->sort_vals[1].as<uint32_t>(), n, 0, 32, h->stream);
    if (rc == 0) {
      HGS_HIP(h, h->sort_tmp.reserve(tmp_bytes));
      rc = hgs_sort_pairs_u64_u32(h->sort_tmp.p, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                  h->sort_vals[1].as<uint32_t>(), n, 0, 32, h->stream);
    }
    if (rc != 0) {
      h->err = "rocprim radix_sort_pairs failed";
      return HGS_ERR_HIP;
    }
    launch_pf_voxel_heads(h->stream, h->sort_keys[1].as<unsigned long long>(), (int)n, h->pf_keep.as<unsigned>(), kVoxelInvalidKey);
    HGS_TRY(scan_u32(h, h->pf_keep.as<uint32_t>(), h->pf_slot.as<uint32_t>(), n));
    // RadiusOutlierRemoval right behind it (the KITTI launch file's prefilter) needs no search tree: the centroids come out in voxel-key order, a point's
    // neighbours are found by key (k_pf_grid_radius_flags) — no resident cloud + index built only to be thrown away, and no host read-back of the voxel
    // count in the middle of the pipeline to size them (the rows of the r-box: at most (2 ceil(r / leaf) + 2)^2 binary searches per point)
    grid_radius = h->prefilter_fast && p->outlier_removal_method == HGS_OUTLIER_RADIUS && p->radius_radius / p->downsample_resolution <= 4.0;
    unsigned* ukeys = nullptr;
    if (grid_radius) {
      HGS_HIP(h, h->pf_ukeys.reserve(n * sizeof(uint32_t)));
      ukeys = h->pf_ukeys.as<unsigned>();
    }
    launch_pf_voxel_centroids(h->stream, cur, h->sort_keys[1].as<unsigned long long>(), h->sort_vals[1].as<unsigned>(), h->pf_keep.as<unsigned>(),
                              h->pf_slot.as<unsigned>(), (int)n, other, d_count, ukeys);
    std::swap(cur, other);
    if (grid_radius) {
      launch_pf_grid_radius_flags(h->stream, cur, d_count, ukeys, d_meta, inv_leaf, (float)p->radius_radius, (float)(p->radius_radius * p->radius_radius),
                                  p->radius_min_neighbors, (int)n, h->pf_keep.as<unsigned>());
      HGS_TRY(scan_u32(h, h->pf_keep.as<uint32_t>(), h->pf_slot.as<uint32_t>(), n));
      launch_pf_compact(h->stream, cur, (int)n, h->pf_keep.as<unsigned>(), h->pf_slot.as<unsigned>(), other, d_count);
      std::swap(cur, other);
    }
  }
  if (n > 0 && p->downsample_method == HGS_DOWNSAMPLE_APPROX_VOXELGRID) {
    // pcl::ApproximateVoxelGrid: stable sort by history bucket, runs of equal voxel, eviction order (k_pf_approx_*)
    const float inv_leaf = 1.0f / (float)p->downsample_resolution;
    for (int i = 0; i < 2; i++) {
      HGS_HIP(h, h->sort_keys[i].reserve(n * sizeof(uint64_t)));
      HGS_HIP(h, h->sort_vals[i].reserve(n * sizeof(uint32_t)));
    }
    const size_t o_bucket = align_up(n * sizeof(uint32_t), 256);
    HGS_HIP(h, h->misc.reserve(o_bucket + 2048 * sizeof(uint32_t)));
    unsigned* d_head = h->misc.as<unsigned>();
    unsigned* d_bucket_used = reinterpret_cast<unsigned*>(h->misc.as<char>() + o_bucket);
    unsigned* d_bucket_rank = d_bucket_used + 512;
    launch_pf_approx_keys(h->stream, cur, d_count, inv_leaf, (int)n, h->sort_keys[0].as<unsigned long long>(), h->sort_vals[0].as<unsigned>());
    size_t tmp_bytes = 0;
    int rc = hgs_sort_pairs_u64_u32(nullptr, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                    h->sort_vals[1].as<uint32_t>(), n, 0, 10, h->stream);
    if (rc == 0) {
      HGS_HIP(h, h->sort_tmp.reserve(tmp_bytes));
      rc = hgs_sort_pairs_u64_u32(h->sort_tmp.p, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                  h->sort_vals[1].as<uint32_t>(), n, 0, 10, h->stream);
    }
    if (rc != 0) {
      h->err = "rocprim radix_sort_pairs failed";
      return HGS_ERR_HIP;
    }
    HGS_HIP(h, hipMemsetAsync(h->pf_keep.p, 0, n * sizeof(uint32_t), h->stream));
    HGS_HIP(h, hipMemsetAsync(d_bucket_used, 0, 1025 * sizeof(uint32_t), h->stream));
    launch_pf_approx_heads(h->stream, cur, h->sort_keys[1].as<unsigned long long>(), h->sort_vals[1].as<unsigned>(), inv_leaf, (int)n, d_head, h->pf_keep.as<unsigned>(),
                           d_bucket_used, d_bucket_rank);
    HGS_TRY(scan_u32(h, h->pf_keep.as<uint32_t>(), h->pf_slot.as<uint32_t>(), n));
    launch_pf_approx_centroids(h->stream, cur, h->sort_keys[1].as<unsigned long long>(), h->sort_vals[1].as<unsigned>(), d_head, h->pf_keep.as<unsigned>(),
                               h->pf_slot.as<unsigned>(), d_bucket_rank, (int)n, other, d_count);
    std::swap(cur, other);
  }
  HGS_HIP(h, hipGetLastError());
  int* hs = h->h_small.as<int>();
  // the point count (pf_small[0]) and the voxel grid's overflow flag (d_meta[12] = pf_small[28]) in ONE copy
  HGS_HIP(h, hipMemcpyAsync(hs, d_count, 32 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  size_t m = (size_t)std::max(0, hs[0]);
  if (n > 0 && p->downsample_method == HGS_DOWNSAMPLE_VOXELGRID && hs[16 + 12]) {
    h->err = "prefilter: voxel grid too fine for this cloud (index overflow; pcl::VoxelGrid refuses it too)";
    return HGS_ERR_INVALID_ARGUMENT;
  }
  hgs_cloud* c = nullptr;
  HGS_TRY(cloud_from_device(h, cur, m, &c));
  if (m > 0 && p->outlier_removal_method != HGS_OUTLIER_NONE && !grid_radius) {
    std::vector<hgs_cloud*> one{c};
    int rc = ensure_index(h, one);
    if (rc == HGS_OK) {
      hipError_t e = hipMemsetAsync(h->pf_keep.p, 0, m * sizeof(uint32_t), h->stream);  // non-finite points are dropped
      if (e != hipSuccess) rc = HGS_ERR_HIP;
    }
    if (rc == HGS_OK) {
      if (p->outlier_removal_method == HGS_OUTLIER_RADIUS) {
        launch_pf_radius_flags(h->stream, c->desc, (float)(p->radius_radius * p->radius_radius), p->radius_min_neighbors, h->pf_keep.as<unsigned>());
      } else {
        hipError_t e = h->pf_dist.reserve(m * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(h->pf_dist.p, 0, m * sizeof(double), h->stream);
        if (e != hipSuccess) rc = HGS_ERR_HIP;
        else {
          launch_pf_mean_knn_dist(h->stream, c->desc, p->statistical_mean_k, h->pf_dist.as<double>());
          launch_pf_statistical(h->stream, h->pf_dist.as<double>(), (int)m, d_stats, p->statistical_stddev, h->pf_keep.as<unsigned>());
        }
      }
    }
    if (rc == HGS_OK) rc = scan_u32(h, h->pf_keep.as<uint32_t>(), h->pf_slot.as<uint32_t>(), m);
    if (rc == HGS_OK) {
      launch_pf_compact(h->stream, cur, (int)m, h->pf_keep.as<unsigned>(), h->pf_slot.as<unsigned>(), other, d_count);
      hipError_t e = hipMemcpyAsync(hs, d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) rc = HGS_ERR_HIP;
    }
    cloud_free(c);
    c = nullptr;
    if (rc != HGS_OK) {
      if (h->err.empty()) h->err = "prefilter: outlier removal failed";
      return rc;
    }
    HGS_TRY(cloud_from_device(h, other, (size_t)std::max(0, hs[0]), &c));
  }
  *out = c;
  return HGS_OK;
}

extern "C" int hgs_prefilter(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes, const hgs_prefilter_params* p, hgs_cloud** out) try {
  return prefilter_impl(h, pts, n, stride_bytes, p, nullptr, 0.0, out);
} catch (...) {
  return status_of_current_exception(h);
}

extern "C" int hgs_prefilter_deskewed(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes, const hgs_prefilter_params* p, const double* imu_angular_velocity,
                                      double scan_period, hgs_cloud** out) try {
  if (!imu_angular_velocity) return prefilter_impl(h, pts, n, stride_bytes, p, nullptr, 0.0, out);  // empty imu_queue: the cloud passes as it is (:184-186)
  if (!std::isfinite(scan_period)) return HGS_ERR_INVALID_ARGUMENT;
  // ang_v(x, y, z) as floats, times -1 (:219-220)
  const float w[3] = {-(float)imu_angular_velocity[0], -(float)imu_angular_velocity[1], -(float)imu_angular_velocity[2]};
  return prefilter_impl(h, pts, n, stride_bytes, p, w, scan_period, out);
} catch (...) {
  return status_of_current_exception(h);
}

extern "C" int hgs_cloud_download(hgs_cloud* c, void* out_pts, size_t stride_bytes) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (c && c->owner) api_lock__ = std::unique_lock<std::recursive_mutex>(c->owner->api_mutex);
  if (!c || !c->owner || stride_bytes < 12 || (stride_bytes % 4) != 0 || (c->n_input > 0 && !out_pts)) return HGS_ERR_INVALID_ARGUMENT;
  hgs_handle* h = c->owner;
  HGS_TRY(set_device(h));
  const size_t n = c->n_input;
  if (n == 0) return HGS_OK;
  HGS_HIP(h, h->h_xform.reserve(n * (sizeof(float4) + sizeof(float))));  // pinned (round 4: two fresh pageable vectors per call)
  const float* xyzw = h->h_xform.as<float>();
  const float* inten = xyzw + 4 * n;
  HGS_HIP(h, hipMemcpyAsync(h->h_xform.p, c->desc.raw, n * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipMemcpyAsync(h->h_xform.as<float>() + 4 * n, c->intensity, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  char* o = (char*)out_pts;
  for (size_t i = 0; i < n; i++) {
    float* f = reinterpret_cast<float*>(o + i * stride_bytes);
    f[0] = xyzw[4 * i], f[1] = xyzw[4 * i + 1], f[2] = xyzw[4 * i + 2];
    if (stride_bytes >= 16) f[3] = 1.0f;
    if (stride_bytes >= 20) f[4] = inten[i];
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception((c ? c->owner : nullptr));
}

// ---- map cloud (src/hdl_graph_slam/map_cloud_generator.cpp:13-51) --------------------------------------------
extern "C" int hgs_map_cloud_generate(hgs_handle* h, hgs_cloud* const* keyframes, const float* poses /* 16 * n, column-major */, size_t n_keyframes,
                                      double resolution, hgs_cloud** out) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !out || (n_keyframes > 0 && (!keyframes || !poses))) return HGS_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  size_t total = 0;
  int max_n = 0;
  for (size_t k = 0; k < n_keyframes; k++) {
    if (!keyframes[k] || keyframes[k]->owner != h) return HGS_ERR_INVALID_ARGUMENT;
    total += keyframes[k]->n_input;
    max_n = std::max(max_n, (int)keyframes[k]->n_input);
  }
  if (total > (size_t)1 << 30) return HGS_ERR_INVALID_ARGUMENT;
  HGS_TRY(set_device(h));
  StageTimer tm(h, HGS_STAGE_PREFILTER);
  const size_t cap = std::max<size_t>(total, 1);
  HGS_HIP(h, h->pf_a.reserve(cap * sizeof(float4)));
  HGS_HIP(h, h->pf_b.reserve(cap * sizeof(float4)));
  HGS_HIP(h, h->pf_keep.reserve(cap * sizeof(uint32_t)));
  HGS_HIP(h, h->pf_slot.reserve(cap * sizeof(uint32_t)));
  HGS_HIP(h, h->pf_small.reserve(256 + sizeof(MapOctree)));
  HGS_HIP(h, h->h_small.reserve(std::max<size_t>(64, sizeof(MapOctree))));
  float4* all = h->pf_a.as<float4>();
  if (n_keyframes > 0 && total > 0) {
    std::vector<MapSource> srcs(n_keyframes);
    size_t off = 0;
    for (size_t k = 0; k < n_keyframes; k++) {
      srcs[k].raw = keyframes[k]->desc.raw, srcs[k].intensity = keyframes[k]->intensity;
      srcs[k].n = (int)keyframes[k]->n_input, srcs[k].offset = (int)off;
      std::memcpy(srcs[k].T, poses + 16 * k, sizeof(float) * 16);
      off += keyframes[k]->n_input;
    }
    HGS_HIP(h, h->misc.reserve(n_keyframes * sizeof(MapSource)));
    HGS_HIP(h, hipMemcpyAsync(h->misc.p, srcs.data(), n_keyframes * sizeof(MapSource), hipMemcpyHostToDevice, h->stream));
    launch_map_transform(h->stream, h->misc.as<MapSource>(), (int)n_keyframes, max_n, all);
    HGS_HIP(h, hipStreamSynchronize(h->stream));  // `srcs` is pageable host memory
  }
  size_t m = total;
  float4* result = all;
  if (resolution > 0.0 && total > 0) {
    int* d_count = h->pf_small.as<int>();
    MapOctree* d_oct = reinterpret_cast<MapOctree*>(h->pf_small.as<char>() + 256);
    MapOctree* h_oct = reinterpret_cast<MapOctree*>(h->h_small.p);
    const int n = (int)total;
    // replay the octree's bounding box: one round per point that forced the box to double (at most one per tree level), each a
    // grid-wide "first point outside" search and a one-thread update; the host only watches the done flag (map clouds are
    // generated every few seconds)
    std::memset(h_oct, 0, sizeof(MapOctree));
    h_oct->first = 0x7fffffff;
    HGS_HIP(h, hipMemcpyAsync(d_oct, h_oct, sizeof(MapOctree), hipMemcpyHostToDevice, h->stream));
    launch_map_first_finite(h->stream, all, n, d_oct);
    launch_map_octree_init(h->stream, all, n, resolution, d_oct);
    for (int round = 0; round <= kMapMaxEvents; round++) {
      launch_map_octree_step(h->stream, all, n, resolution, d_oct);
      HGS_HIP(h, hipMemcpyAsync(h_oct, d_oct, sizeof(MapOctree), hipMemcpyDeviceToHost, h->stream));
      HGS_HIP(h, hipStreamSynchronize(h->stream));
      if (h_oct->done) break;
    }
    if (h_oct->overflow || !h_oct->done) {
      h->err = "map cloud: resolution too fine for the extent of the map (the octree would be deeper than 21 levels)";
      return HGS_ERR_INVALID_ARGUMENT;
    }
    for (int i = 0; i < 2; i++) {
      HGS_HIP(h, h->sort_keys[i].reserve(total * sizeof(uint64_t)));
      HGS_HIP(h, h->sort_vals[i].reserve(total * sizeof(uint32_t)));
    }
    // interleaved keys use 3 * depth bits; one more so that the all-ones key of a non-finite point sorts behind every voxel
    const int key_bits = h_oct->n_events > 0 ? 3 * h_oct->events[h_oct->n_events - 1].depth + 1 : 1;
    launch_map_keys(h->stream, all, n, resolution, d_oct, h->sort_keys[0].as<unsigned long long>(), h->sort_vals[0].as<unsigned>());
    size_t tmp_bytes = 0;
    int rc = hgs_sort_pairs_u64_u32(nullptr, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                    h->sort_vals[1].as<uint32_t>(), total, 0, key_bits, h->stream);
    if (rc == 0) {
      HGS_HIP(h, h->sort_tmp.reserve(tmp_bytes));
      rc = hgs_sort_pairs_u64_u32(h->sort_tmp.p, &tmp_bytes, h->sort_keys[0].as<uint64_t>(), h->sort_keys[1].as<uint64_t>(), h->sort_vals[0].as<uint32_t>(),
                                  h->sort_vals[1].as<uint32_t>(), total, 0, key_bits, h->stream);
    }
    if (rc != 0) {
      h->err = "rocprim radix_sort_pairs failed";
      return HGS_ERR_HIP;
    }
    launch_pf_voxel_heads(h->stream, h->sort_keys[1].as<unsigned long long>(), n, h->pf_keep.as<unsigned>(), kMapInvalidKey);
    HGS_TRY(scan_u32(h, h->pf_keep.as<uint32_t>(), h->pf_slot.as<uint32_t>(), total));
    launch_map_centers(h->stream, h->sort_keys[1].as<unsigned long long>(), h->pf_keep.as<unsigned>(), h->pf_slot.as<unsigned>(), n, resolution, d_oct,
                       h->pf_b.as<float4>(), d_count);
    HGS_HIP(h, hipGetLastError());
    int* hs = h->h_small.as<int>();
    HGS_HIP(h, hipMemcpyAsync(hs, d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HGS_HIP(h, hipStreamSynchronize(h->stream));
    m = (size_t)std::max(0, hs[0]);
    result = h->pf_b.as<float4>();
  }
  HGS_HIP(h, hipGetLastError());
  return cloud_from_device(h, result, m, out);
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_profile_enable(hgs_handle* h, int enabled) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h) return HGS_ERR_INVALID_ARGUMENT;
  h->profiling = enabled != 0;
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_profile_read(hgs_handle* h, double* ms, uint64_t* launches, int reset) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h) return HGS_ERR_INVALID_ARGUMENT;
  HGS_TRY(set_device(h));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  for (auto& ev : h->prof_events) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, ev.a, ev.b) == hipSuccess) {
      h->prof_ms[ev.stage] += (double)t;
      h->prof_launches[ev.stage]++;
    }
    h->prof_free.push_back(ev);
  }
  h->prof_events.clear();
  for (int i = 0; i < HGS_STAGE_COUNT; i++) {
    if (ms) ms[i] = h->prof_ms[i];
    if (launches) launches[i] = h->prof_launches[i];
    if (reset) h->prof_ms[i] = 0, h->prof_launches[i] = 0;
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_synchronize(hgs_handle* h) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h) return HGS_ERR_INVALID_ARGUMENT;
  HGS_TRY(set_device(h));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

// ---- stage-level hooks for the parity tests ---------------------------------------------------------------
int hgs_debug_target_covariances(hgs_handle* h, float* out6) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !out6) return HGS_ERR_INVALID_ARGUMENT;
  if (!h->target) return HGS_ERR_NO_TARGET;
  HGS_TRY(set_device(h));
  hgs_cloud* t = h->target;
  std::vector<hgs_cloud*> all{t};
  HGS_TRY(ensure_cov(h, all, h->prm.correspondence_randomness));
  const size_t slots = (size_t)t->P * kLeaf;
  std::vector<float4> pts(slots), cov(2 * slots);
  int nvalid = 0;
  HGS_HIP(h, hipMemcpyAsync(pts.data(), t->desc.pts, slots * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipMemcpyAsync(cov.data(), t->desc.cov, 2 * slots * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipMemcpyAsync(&nvalid, &t->desc.meta->nvalid, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  std::memset(out6, 0, t->n_input * 6 * sizeof(float));
  for (int i = 0; i < nvalid; i++) {
    int o;
    std::memcpy(&o, &pts[i].w, 4);
    if (o < 0 || (size_t)o >= t->n_input) continue;
    float* p = out6 + (size_t)o * 6;
    p[0] = cov[2 * i].x, p[1] = cov[2 * i].y, p[2] = cov[2 * i].z, p[3] = cov[2 * i].w, p[4] = cov[2 * i + 1].x, p[5] = cov[2 * i + 1].y;
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_debug_gicp_linearize(hgs_handle* h, const double T12[12], double* H36, double* b6, double* err, int32_t* corr) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !T12 || !H36 || !b6 || !err) return HGS_ERR_INVALID_ARGUMENT;
  if (h->prm.method != HGS_FAST_GICP && h->prm.method != HGS_FAST_VGICP) return HGS_ERR_UNSUPPORTED;
  if (!h->target) return HGS_ERR_NO_TARGET;
  if (!h->source) return HGS_ERR_NO_SOURCE;
  HGS_TRY(set_device(h));
  hgs_cloud *s = h->source, *t = h->target;
  const bool voxel = h->prm.method == HGS_FAST_VGICP;
  std::vector<hgs_cloud*> all{s, t};
  HGS_TRY(ensure_cov(h, all, h->prm.correspondence_randomness));
  if (voxel) HGS_TRY(ensure_vgicp_target(h, t));
  const int max_blocks = std::max(1, ((int)s->n_input + kBlock - 1) / kBlock);
  std::vector<hgs_cloud*> src{s};
  const CloudDesc* d_descs = nullptr;
  HGS_TRY(upload_descs(h, src, false, &d_descs, nullptr));
  HGS_HIP(h, h->states.reserve(sizeof(GicpState)));
  HGS_HIP(h, h->partials.reserve((size_t)max_blocks * kAcc * sizeof(double)));
  HGS_HIP(h, h->misc.reserve(64 * sizeof(double)));
  HGS_HIP(h, hipMemsetAsync(h->partials.p, 0, (size_t)max_blocks * kAcc * sizeof(double), h->stream));
  HGS_HIP(h, hipMemcpyAsync(h->misc.p, T12, 12 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  launch_gicp_debug_state(h->stream, h->states.as<GicpState>(), h->misc.as<double>());
  if (voxel) launch_vgicp_linearize(h->stream, d_descs, vgicp_target_view(t), h->states.as<GicpState>(), vgicp_consts(h->prm), h->partials.as<double>(), max_blocks, 1);
  else launch_gicp_linearize(h->stream, d_descs, target_view(t), h->states.as<GicpState>(), gicp_consts(h->prm), h->partials.as<double>(), max_blocks, 1, 64);
  double* d_out = h->misc.as<double>() + 16;
  launch_reduce_partials(h->stream, h->partials.as<double>(), max_blocks, kAcc, d_out);
  double acc[kAcc];
  HGS_HIP(h, hipMemcpyAsync(acc, d_out, sizeof(acc), hipMemcpyDeviceToHost, h->stream));
  const size_t s_slots = (size_t)s->P * kLeaf, t_slots = (size_t)t->P * kLeaf;
  std::vector<float4> spts, tpts;
  std::vector<int> dcorr;
  int s_nvalid = 0;
  if (corr) {
    spts.resize(s_slots), tpts.resize(t_slots), dcorr.resize(s_slots);
    HGS_HIP(h, hipMemcpyAsync(spts.data(), s->desc.pts, s_slots * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    HGS_HIP(h, hipMemcpyAsync(tpts.data(), t->desc.pts, t_slots * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    HGS_HIP(h, hipMemcpyAsync(dcorr.data(), s->desc.corr, s_slots * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HGS_HIP(h, hipMemcpyAsync(&s_nvalid, &s->desc.meta->nvalid, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  }
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  int k = 0;
  for (int r = 0; r < 6; r++)
    for (int c = r; c < 6; c++) H36[r * 6 + c] = H36[c * 6 + r] = acc[k++];
  for (int i = 0; i < 6; i++) b6[i] = acc[21 + i];
  *err = acc[27];
  if (corr) {
    for (size_t i = 0; i < s->n_input; i++) corr[i] = voxel ? 0 : -1;
    for (int i = 0; i < s_nvalid; i++) {
      int so, to = -1;
      std::memcpy(&so, &spts[i].w, 4);
      const int j = dcorr[i];
      if (voxel) to = j;  // VGICP: number of voxel correspondences of that source point
      else if (j >= 0 && (size_t)j < t_slots) std::memcpy(&to, &tpts[j].w, 4);
      if (so >= 0 && (size_t)so < s->n_input) corr[so] = to;
    }
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_debug_ndt_cells(hgs_handle* h, int32_t cap, int32_t* ijk3, double* mean3, float* icov6, int32_t* npts, int32_t* n_cells) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !n_cells) return HGS_ERR_INVALID_ARGUMENT;
  if (h->prm.method != HGS_NDT_OMP) return HGS_ERR_UNSUPPORTED;
  if (!h->target) return HGS_ERR_NO_TARGET;
  HGS_TRY(set_device(h));
  hgs_cloud* t = h->target;
  HGS_TRY(ensure_ndt_target(h, t));
  CloudMeta meta;
  HGS_HIP(h, hipMemcpyAsync(&meta, t->desc.meta, sizeof(CloudMeta), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  *n_cells = meta.ndt_ncells;
  const int n = std::min<int>(meta.ndt_ncells, cap);
  if (n <= 0) return HGS_OK;
  std::vector<NdtCellRec> cells(n);
  HGS_HIP(h, hipMemcpyAsync(cells.data(), t->ndt_cells, (size_t)n * sizeof(NdtCellRec), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; i++) {
    int key;
    std::memcpy(&key, &cells[i].v1.w, 4);
    if (ijk3) {
      const int m1 = meta.ndt_div_mul[1], m2 = meta.ndt_div_mul[2];
      ijk3[3 * i] = key % m1 + meta.ndt_min_b[0];
      ijk3[3 * i + 1] = (key % m2) / m1 + meta.ndt_min_b[1];
      ijk3[3 * i + 2] = key / m2 + meta.ndt_min_b[2];
    }
    if (mean3) mean3[3 * i] = cells[i].mean[0], mean3[3 * i + 1] = cells[i].mean[1], mean3[3 * i + 2] = cells[i].mean[2];
    if (icov6) {
      float* o = icov6 + 6 * i;
      o[0] = cells[i].v0.x, o[1] = cells[i].v0.y, o[2] = cells[i].v0.z, o[3] = cells[i].v0.w, o[4] = cells[i].v1.x, o[5] = cells[i].v1.y;
    }
    if (npts) npts[i] = (int)cells[i].v1.z;
  }
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

int hgs_debug_ndt_derivatives(hgs_handle* h, const double p6[6], double* score, double* g6, double* H36) try {
  std::unique_lock<std::recursive_mutex> api_lock__;
  if (h) api_lock__ = std::unique_lock<std::recursive_mutex>((h)->api_mutex);
  if (!h || !p6 || !score || !g6 || !H36) return HGS_ERR_INVALID_ARGUMENT;
  if (h->prm.method != HGS_NDT_OMP) return HGS_ERR_UNSUPPORTED;
  if (!h->target) return HGS_ERR_NO_TARGET;
  if (!h->source) return HGS_ERR_NO_SOURCE;
  HGS_TRY(set_device(h));
  hgs_cloud *s = h->source, *t = h->target;
  HGS_TRY(ensure_ndt_target(h, t));
  const int max_blocks = std::max(1, ((int)s->n_input + kBlock - 1) / kBlock);
  std::vector<hgs_cloud*> src{s};
  const CloudDesc* d_descs = nullptr;
  HGS_TRY(upload_descs(h, src, false, &d_descs, nullptr));
  const NdtConsts c = ndt_consts(h->prm);
  HGS_HIP(h, h->states.reserve(sizeof(NdtState)));
  HGS_HIP(h, h->angles.reserve(sizeof(NdtAngles)));
  HGS_HIP(h, h->ndt_accum.reserve(sizeof(NdtAccum)));
  HGS_HIP(h, h->misc.reserve(128 * sizeof(double)));
  HGS_HIP(h, hipMemsetAsync(h->ndt_accum.p, 0, sizeof(NdtAccum), h->stream));
  HGS_HIP(h, hipMemcpyAsync(h->misc.p, p6, 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  launch_ndt_debug_state(h->stream, h->states.as<NdtState>(), h->angles.as<NdtAngles>(), h->misc.as<double>(), c);
  Progress none{};
  HGS_HIP(h, h->ndt_plan.reserve(256));
  struct {
    unsigned long long queue, pad;
    int tile_base[2];
  } plan{0, 0, {0, max_blocks}};
  HGS_HIP(h, hipMemcpy(h->ndt_plan.p, &plan, sizeof(plan), hipMemcpyHostToDevice));
  launch_ndt_pass(h->stream, d_descs, ndt_target_view(h, t), h->states.as<NdtState>(), h->angles.as<NdtAngles>(), c, h->ndt_accum.as<NdtAccum>(),
                  reinterpret_cast<const int*>((char*)h->ndt_plan.p + 16), reinterpret_cast<unsigned long long*>(h->ndt_plan.p), 1, 0, std::min(max_blocks, 96),
                  2 /* several tiles per block, several grabs per block */, (h->ndt_sort != 0 && s->has_index) ? 1 : 0, 1, none);
  double acc[kAccNdt];
  HGS_HIP(h, hipMemcpyAsync(acc, h->ndt_accum.as<NdtAccum>()->out, sizeof(acc), hipMemcpyDeviceToHost, h->stream));
  HGS_HIP(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < 36; i++) H36[i] = acc[i];
  for (int i = 0; i < 6; i++) g6[i] = acc[36 + i];
  *score = acc[42];
  return HGS_OK;
} catch (...) {
  return status_of_current_exception(h);
}

}  // extern "C"
