// hgs_device.h — structures shared between the host engine (hgs_engine.hip) and the kernels (hgs_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "hgs_gicp.h"
#include "hgs_ndt.h"
#include "hgs_vgicp.h"

namespace hgs {

// Per-cloud mutable header living in HBM next to the cloud's arrays.
struct CloudMeta {
  int nvalid;            // number of finite points (written by k_bbox_count)
  int ndt_ncells;        // number of valid Gaussian cells (NDT target)
  unsigned bbmin[3];     // order-preserving uint encoding of the float bounding box
  unsigned bbmax[3];
  int ndt_min_b[3], ndt_max_b[3], ndt_div_mul[3];
  int ndt_error;         // 1: voxel grid larger than INT_MAX cells (upstream aborts)
  // VGICP Gaussian voxel map of this cloud as a target (hgs_vgicp.h): same roles as the ndt_* fields
  int vg_ncells;
  int vg_min_b[3], vg_max_b[3], vg_div_mul[3];
  int vg_error;
  int pad[1];
};

// Read-only description of one cloud resident in HBM.
struct CloudDesc {
  const float4* raw;  // [n_input]  original order, .w = original index (int bits)
  float4* pts;        // [P*kLeaf]  Hilbert order, .w = original index, padding = +inf
  float4* lpts;       // [P*kLeaf]  the same points as SoA leaves {x[8],y[8],z[8],w[8]} (hgs_bvh.h)
  float4* nodes;      // [4*P+8]    implicit tree AABBs, grouped 4 nodes per 128-byte record (hgs_bvh.h)
  float4* cov;        // [2*P*kLeaf] GICP covariance of sorted point i: {xx,xy,xz,yy},{yz,zz,0,0}
  int* corr;          // [P*kLeaf]  scratch: correspondence (sorted target position or -1) when used as a source
  CloudMeta* meta;
  int n_input;
  int P;
  int sort_off;       // offset of this cloud's segment in the batch sort arrays
  int pad;            // index build only: 1 = also clear corr[] (the cloud was invalidated); 0 everywhere else
};

struct TargetView {
  const float4* nodes;
  const float4* pts;
  const float4* lpts;
  const float4* cov;
  const CloudMeta* meta;
  int P;
  int seed_bits;             // log2 of the finest seed table's size (0: no seed grid)
  const unsigned* seed_tab;  // the cloud's seed grid (seed_grid_lookup, hgs_kernels.hip): three direct-mapped tables back to back
};

// NDT target tables
struct NdtTargetView {
  const int* hash_keys;
  const int* hash_vals;
  const NdtCellRec* cells;
  const CloudMeta* meta;
  const int2* hash_kv;  // NDT only: (key, cell index) per slot — the derivative kernel's one-load probe
  int hash_mask;
  float inv_leaf;
};

// Integer totals of one NDT derivative pass of one problem (hgs_ndt.h "exact accumulation"), in HBM: the blocks of
// k_ndt_pass add into it with 64-bit atomics, the block whose contribution completes the problem's tile count reads and clears it.
struct NdtAccum {
  unsigned long long w[kAccNdt * 4];  // [accumulator * 4 + {chunk0 low digits, chunk1 low, chunk0 high (signed), chunk1 high (signed)}]
  unsigned tiles_done;                // tiles of the current pass accounted for
  unsigned overflow;                  // a per-point term left the fixed range (or was NaN): the pass yields NaN
  double out[kAccNdt];                // hgs_debug_ndt_derivatives: the totals as doubles
};

struct DevResult {
  float T[16];  // column-major
  int converged, iterations, lm_tries, pad;
  double error;
  double fit_sum;  // sum of d2 over inliers
  unsigned fit_count;
  unsigned pad2;
};

// Progress of a batch of optimisers: counters in HBM, mirrored into host-mapped pinned memory (see progress_tick).
struct Progress {
  int* dev;                   // [0] problems finished, [1] per-problem end-of-round ticks
  volatile int* host_done;    // 1 once every problem has finished
  volatile int* host_rounds;  // number of completed rounds
  int B;
  int pad;
};

constexpr int kFusedRoundMaxProblems = 4;  // launches of at most this many problems may run the LM round in two launches (control steps replicated per block)
constexpr int kBlock = 256;
constexpr int kKnnLaneList = 16;          // leaves a lane of k_knn_cov remembers as its own candidates' (more: the wave falls back to the replay / walk)
constexpr int kKnnLeafLog = 128;           // leaves pass 1 of k_knn_cov remembers per wave for pass 2 (more: pass 2 walks the tree)
constexpr int kNW = 1;                    // packets of 64 queries a wave walks in lock-step in the 1-NN kernels (hgs_wave_bvh.h)
constexpr int kTileNN = kBlock * kNW;     // source points per block of k_gicp_linearize / k_fitness

// ---- launchers (hgs_kernels.hip) --------------------------------------------------------------------------
void launch_pack_aos(hipStream_t s, const float4* staged /* {x, y, z, intensity} */, int n, float4* raw, float* intensity /* may be null */, CloudMeta* meta_to_reset /* may be null */,
                     const CloudDesc* desc = nullptr /* + the cloud's resident descriptor: *desc -> *desc_out */, CloudDesc* desc_out = nullptr);
void launch_meta_init(hipStream_t s, const CloudDesc* descs, int ncloud);
void launch_bbox_count(hipStream_t s, const CloudDesc* descs, int ncloud, int max_n);
void launch_hilbert_keys(hipStream_t s, const CloudDesc* descs, int ncloud, int max_n, unsigned long long* keys, unsigned* vals, int drop_bits);
void launch_gather_sorted(hipStream_t s, const CloudDesc* descs, int ncloud, int max_slots, const unsigned* sorted_vals);
void launch_build_tree(hipStream_t s, const CloudDesc* descs, int ncloud, int max_P);
void launch_knn_cov(hipStream_t s, const CloudDesc* descs, int ncloud, int max_n, int k, int qpw, int reg_method, int gather /* 0 walk, 1 leaf-log replay, 2 per-lane lists */,
                    double* raw_stage = nullptr /* ncloud * max_n * 6 doubles: non-FROBENIUS regularisations run as search + k_cov_regularize */);

void launch_gicp_init(hipStream_t s, GicpState* states, const float* guesses, int B, Progress prog);
struct Guess16 {
  float m[16];  // column-major
};
void launch_gicp_init1(hipStream_t s, GicpState* states, const float* guess_host /* [16] */, Progress prog);
void launch_gicp_linearize(hipStream_t s, const CloudDesc* descs, TargetView tgt, const GicpState* states, GicpConsts c, double* partials, int max_blocks, int B,
                           int qpw);
void launch_gicp_solve(hipStream_t s, const CloudDesc* descs, GicpState* states, GicpConsts c, const double* partials, int max_blocks, int B,
                       int tile_points /* points per block of the linearize kernel that filled `partials` */);
void launch_gicp_error(hipStream_t s, const CloudDesc* descs, TargetView tgt, const GicpState* states, double* partials_err, int max_blocks, int B);
// two launches per LM round (hgs_kernels.hip, "two launches per LM round"): the control steps replicated in every block, states ping-pong between two buffers
void launch_gicp_linearize_round2(hipStream_t s, const CloudDesc* descs, TargetView tgt, const GicpState* states_in, GicpState* states_out, GicpConsts c, double* partials,
                                  const double* partials_err, int max_blocks /* row stride of the partials */, int lin_blocks /* grid */, int B, int qpw, Progress prog,
                                  DevResult* results, DevResult* early_out /* host-mapped, or null */);
void launch_gicp_error_round2(hipStream_t s, const CloudDesc* descs, TargetView tgt, const GicpState* states_in, GicpState* states_out, GicpConsts c, const double* partials,
                              double* partials_err, int max_blocks, int err_blocks /* grid */, int B, int lin_tile_points);
void launch_gicp_decide(hipStream_t s, const CloudDesc* descs, GicpState* states, GicpConsts c, const double* partials_err, int max_blocks, int B, Progress prog);
void launch_gicp_results(hipStream_t s, const GicpState* states, DevResult* out, int B);

void launch_fitness(hipStream_t s, const CloudDesc* descs, TargetView tgt, const DevResult* poses, double max_range, double* partials, int max_blocks, int B,
                    int use_seed, int qpw);
void launch_fitness_final(hipStream_t s, const CloudDesc* descs, const double* partials, int max_blocks, DevResult* out, int B, int tile_pts);
}  // namespace hgs
struct hgs_result;  // include/hgs_registration.h
namespace hgs {
void launch_results_to_records(hipStream_t s, const DevResult* res, const int* candidate_ids, int n, int n_slots, ::hgs_result* out);
void launch_nn_query(hipStream_t s, TargetView tgt, const float4* q, int nq, int* idx, float* d2);
// seed grid of a target cloud: every sorted point enters three direct-mapped tables (cells of 0.25 / 1 / 4 m, sizes 2^bits, 2^(bits-2), 2^(bits-4))
void launch_seed_grid_build(hipStream_t s, const float4* pts, const CloudMeta* meta, int n_max, unsigned* tab, int bits);
inline size_t seed_grid_entries(int bits) { return ((size_t)1 << bits) + ((size_t)1 << (bits - 2)) + ((size_t)1 << (bits - 4)); }
void launch_transform(hipStream_t s, const float4* raw, int n, const float* T16_colmajor_dev, float4* out);

void launch_ndt_grid_params(hipStream_t s, CloudDesc desc, float inv_leaf);
void launch_ndt_cell_keys(hipStream_t s, CloudDesc desc, float inv_leaf, unsigned long long* keys, unsigned* vals);
void launch_ndt_build_cells(hipStream_t s, CloudDesc desc, const unsigned long long* sorted_keys, const unsigned* sorted_vals, int min_points,
                            int* hash_keys, int* hash_vals, int hash_mask, NdtCellRec* cells);
void launch_ndt_init(hipStream_t s, NdtState* states, NdtAngles* angles, const float* guesses, NdtConsts c, int B, Progress prog, NdtAccum* accum_to_zero /* [B], may be null */);
void launch_ndt_pack_hash(hipStream_t s, const int* keys, const int* vals, int2* kv, int cap);
// one Newton iteration of B problems in one launch: `blocks` resident blocks pull runs of consecutive (problem, tile) items —
// numbered by tile_base[B + 1], the prefix sums of the problems' tile counts — from queues[parity & 1] (zero at the start of the
// pass; the kernel zeroes the other head for the next pass, so the caller alternates parity), at most `chunk` items per grab;
// sorted: read the sources in Hilbert order (they have a search index); debug: only leave the totals in accum[].out
void launch_ndt_pass(hipStream_t s, const CloudDesc* descs, NdtTargetView tgt, NdtState* states, NdtAngles* angles, NdtConsts c, NdtAccum* accum, const int* tile_base,
                     unsigned long long* queues, int B, int parity, int blocks, int chunk, int sorted, int debug, Progress prog);
void launch_ndt_results(hipStream_t s, const CloudDesc* descs, const NdtState* states, DevResult* out, int B);

void launch_vgicp_grid_params(hipStream_t s, CloudDesc desc, double resolution);
void launch_vgicp_cell_keys(hipStream_t s, CloudDesc desc, double resolution, unsigned long long* keys, unsigned* vals);
void launch_vgicp_build_cells(hipStream_t s, CloudDesc desc, const unsigned long long* sorted_keys, const unsigned* sorted_vals, int* hash_keys,
                              int* hash_vals, int hash_mask, NdtCellRec* cells);
void launch_vgicp_linearize(hipStream_t s, const CloudDesc* descs, NdtTargetView tgt, const GicpState* states, VgicpConsts c, double* partials,
                            int max_blocks, int B);
void launch_vgicp_error(hipStream_t s, const CloudDesc* descs, NdtTargetView tgt, const GicpState* states, VgicpConsts c, double* partials_err,
                        int max_blocks, int B);

// map cloud (src/hdl_graph_slam/map_cloud_generator.cpp)
struct MapSource {
  const float4* raw;       // keyframe cloud, original order
  const float* intensity;
  int n, offset;           // size, position of its first point in the concatenated cloud
  float T[16];             // keyframe pose, column-major float
};
void launch_map_transform(hipStream_t s, const MapSource* srcs, int nsrc, int max_n, float4* out);
// The growth of pcl::octree::OctreePointCloud's bounding box while the map's points are added in order (hgs_math.h "octree box"):
// events[0] is the first finite point, every later event a point that lay outside the box of its moment and the state after the
// doublings it forced.  A point's key is taken against the box of the last event at or before it and then gains what the later
// doublings added — exactly what re-rooting the tree does to the leaves inserted earlier.
constexpr int kMapMaxDepth = 21;                  // 3 * 21 interleaved key bits < 64
constexpr int kMapMaxEvents = kMapMaxDepth + 3;
struct MapOctreeEvent {
  int index, depth;
  double mn[3];
  unsigned long long gained[3];
};
struct MapOctree {
  int first;      // index of the first finite point (INT_MAX: none)
  int next_viol;  // lowest index after the last event that lies outside the current box (INT_MAX: none found)
  int n_events, overflow, done, pad;
  double mx[3];   // the current box's max; min / depth / gained: events[n_events - 1]
  MapOctreeEvent events[kMapMaxEvents];
};
void launch_map_first_finite(hipStream_t s, const float4* pts, int n, MapOctree* oct);
void launch_map_octree_init(hipStream_t s, const float4* pts, int n, double res, MapOctree* oct);
void launch_map_octree_step(hipStream_t s, const float4* pts, int n, double res, MapOctree* oct);
void launch_map_keys(hipStream_t s, const float4* pts, int n, double res, const MapOctree* oct, unsigned long long* keys, unsigned* vals);
void launch_map_centers(hipStream_t s, const unsigned long long* keys, const unsigned* head, const unsigned* slot, int n, double res, const MapOctree* oct, float4* out,
                        int* count_out);

// prefilter (apps/prefiltering_nodelet.cpp)
void launch_pf_load(hipStream_t s, const float4* staged, int n, float4* out, const float* deskew_w /* null: no deskewing */, double scan_period, int* count_out,
                    unsigned* meta_out /* [16]: the voxel-grid record, initialised here */);
void launch_pf_distance_flags(hipStream_t s, const float4* pts, int n, int use_filter, double near_thresh, double far_thresh, unsigned* keep);
void launch_pf_compact(hipStream_t s, const float4* in, int n, const unsigned* keep, const unsigned* slot, float4* out, int* count);
void launch_pf_bbox(hipStream_t s, const float4* pts, const int* count, int cap, unsigned* meta, int dist_filter = 0, double near_thresh = 0, double far_thresh = 0);
void launch_pf_grid(hipStream_t s, unsigned* meta, float inv_leaf);
void launch_pf_voxel_keys(hipStream_t s, const float4* pts, const int* count, const unsigned* meta, float inv_leaf, int cap, unsigned long long* keys, unsigned* vals,
                          int dist_filter = 0, double near_thresh = 0, double far_thresh = 0);
void launch_pf_voxel_heads(hipStream_t s, const unsigned long long* keys, int cap, unsigned* head, unsigned long long invalid_key);
constexpr unsigned long long kVoxelInvalidKey = 0xffffffffull;  // prefilter voxel grid: 31-bit linear indices like pcl::VoxelGrid
constexpr unsigned long long kMapInvalidKey = ~0ull;            // map cloud: interleaved keys use at most 63 bits
void launch_pf_voxel_centroids(hipStream_t s, const float4* pts, const unsigned long long* keys, const unsigned* vals, const unsigned* head, const unsigned* slot, int cap,
                               float4* out, int* count_out, unsigned* ukeys = nullptr /* [cap]: the voxels' keys in output order */);
void launch_pf_grid_radius_flags(hipStream_t s, const float4* cen, const int* count, const unsigned* ukeys, const unsigned* meta, float inv_leaf, float radius, float r2,
                                 int min_neighbors, int cap, unsigned* keep);
void launch_pf_approx_keys(hipStream_t s, const float4* pts, const int* count, float inv_leaf, int cap, unsigned long long* keys, unsigned* vals);
void launch_pf_approx_heads(hipStream_t s, const float4* pts, const unsigned long long* keys, const unsigned* vals, float inv_leaf, int cap, unsigned* head, unsigned* evict,
                            unsigned* bucket_used /* [512], zeroed */, unsigned* bucket_rank /* [513] */);
void launch_pf_approx_centroids(hipStream_t s, const float4* pts, const unsigned long long* keys, const unsigned* vals, const unsigned* head, const unsigned* evict,
                                const unsigned* evict_rank, const unsigned* bucket_rank, int cap, float4* out, int* count_out);
void launch_pf_radius_flags(hipStream_t s, CloudDesc d, float r2, int min_neighbors, unsigned* keep);
void launch_pf_mean_knn_dist(hipStream_t s, CloudDesc d, int mean_k, double* dist);
void launch_pf_statistical(hipStream_t s, const double* dist, int n, double* stats, double stddev_mul, unsigned* keep);
void launch_pf_to_cloud(hipStream_t s, const float4* in, int n, float4* raw, float* intensity, CloudMeta* meta_to_reset, const CloudDesc* desc = nullptr, CloudDesc* desc_out = nullptr);

// stage-level test hooks
void launch_gicp_debug_state(hipStream_t s, GicpState* st, const double* T12_dev);
void launch_ndt_debug_state(hipStream_t s, NdtState* st, NdtAngles* ang, const double* p6_dev, NdtConsts c);
void launch_reduce_partials(hipStream_t s, const double* partials, int ntiles, int width /* kAcc | kAccNdt */, double* out);

}  // namespace hgs
