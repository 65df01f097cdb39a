/* hgs_registration.h — C-ABI of the MI355X-native scan-matching backend for hdl_graph_slam.
 *
 * This is the drop-in boundary behind
 *     pcl::Registration<PointXYZI,PointXYZI>::Ptr hdl_graph_slam::select_registration_method(ros::NodeHandle&)
 *     (reference: include/hdl_graph_slam/registrations.hpp:17, src/hdl_graph_slam/registrations.cpp:22-124).
 * The pcl::Registration adapter that forwards to these entry points is adapters/registration_hip.hpp; the
 * reference-side patch is shown in INTEGRATION.md.  Every function is extern "C", takes plain pointers and
 * sizes, returns an int status (HGS_OK == 0) and never throws.  There is no global mutable state: each
 * hgs_handle owns one HIP stream and its device buffers, so the two engine instances that live in one
 * nodelet manager (odometry: apps/scan_matching_odometry_nodelet.cpp:106, loop closure:
 * include/hdl_graph_slam/loop_detector.hpp:47) can run concurrently from different threads.
 * Thread safety: calls on ONE engine (and on its clouds) are serialised by a per-engine mutex, different engines are
 * independent.  hgs_destroy is the exception: it must not race with ANY other call on that engine or on one of its clouds
 * (including hgs_cloud_destroy / hgs_cloud_download from a garbage-collector thread) — the mutex it would have to wait on is
 * part of the object it deletes.  Destroy the engine after the threads that use it have finished with it; clouds that outlive
 * it are orphaned safely (hgs_cloud_destroy still frees them).
 *
 * Conventions
 *  - points: array of records with three consecutive floats x,y,z at the start of each record and a byte stride
 *    (32 for pcl::PointXYZI, 16 for float4, 12 for packed xyz).  Non-finite points are ignored.
 *  - 4x4 transforms: float[16], COLUMN-major (Eigen::Matrix4f::data()), mapping source -> target frame.
 *  - indices returned to the caller always refer to the caller's original point order.
 */
#ifndef HGS_REGISTRATION_H
#define HGS_REGISTRATION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGS_ABI_VERSION 5

enum hgs_status {
  HGS_OK = 0,
  HGS_ERR_INVALID_ARGUMENT = 1,
  HGS_ERR_NO_TARGET = 2,
  HGS_ERR_NO_SOURCE = 3,
  HGS_ERR_HIP = 4,      /* a HIP runtime call failed; hgs_last_error() has the text */
  HGS_ERR_NO_DEVICE = 5,
  HGS_ERR_UNSUPPORTED = 6,
  HGS_ERR_OUT_OF_MEMORY = 7, /* host allocation failed inside the backend (std::bad_alloc caught at the boundary) */
  HGS_ERR_INTERNAL = 8,      /* any other C++ exception caught at the boundary; hgs_last_error() has the text */
  HGS_ERR_COMM = 9           /* RCCL call failed / hgs_comm_init missing; hgs_last_error() has the text */
};

/* registration_method strings of registrations.cpp:26-121 that this backend implements. */
enum hgs_method {
  HGS_FAST_GICP = 0,  /* fast_gicp::FastGICP      — registrations.cpp:27-36   */
  HGS_FAST_VGICP = 1, /* fast_gicp::FastVGICP     — registrations.cpp:48-56   */
  HGS_NDT_OMP = 2     /* pclomp::NormalDistributionsTransform — registrations.cpp:101-120 */
};

/* reg_nn_search_method (registrations.cpp:103,112-118) / FastVGICP neighbour search. */
enum hgs_neighbor_search {
  HGS_KDTREE = 0,  /* NDT_OMP only: radius search (resolution) over the centroids of the valid cells */
  HGS_DIRECT1 = 1,
  HGS_DIRECT7 = 2,
  HGS_DIRECT27 = 3
};

/* fast_gicp::RegularizationMethod of the k-NN covariances (FastGICP / FastVGICP::setRegularizationMethod).  hdl_graph_slam
 * never calls the setter, so the constructor default of the fast_gicp checkout applies: FROBENIUS per SURVEY.md A.2; a
 * checkout whose FastGICP constructor selects PLANE is matched by setting HGS_REG_PLANE. */
enum hgs_regularization {
  HGS_REG_FROBENIUS = 0,          /* ((C + 1e-3 I)^-1 / ||.||_F)^-1                         */
  HGS_REG_PLANE = 1,              /* U diag(1, 1, 1e-3) V^T                                 */
  HGS_REG_MIN_EIG = 2,            /* U diag(max(sigma_i, 1e-3)) V^T                         */
  HGS_REG_NORMALIZED_MIN_EIG = 3, /* U diag(max(sigma_i / sigma_max, 1e-3)) V^T             */
  HGS_REG_NONE = 4                /* C itself                                               */
};

typedef struct hgs_params {
  int32_t method;                     /* hgs_method                                                        */
  int32_t max_iterations;             /* reg_maximum_iterations           (64)                             */
  double transformation_epsilon;      /* reg_transformation_epsilon       (0.01)                           */
  double rotation_epsilon;            /* fast_gicp LsqRegistration        (2e-3), not exposed by hdl       */
  double max_correspondence_distance; /* reg_max_correspondence_distance  (2.5), FAST_GICP only            */
  int32_t correspondence_randomness;  /* reg_correspondence_randomness    (20) = k of the covariance kNN, 1..64 */
  int32_t neighbor_search;            /* hgs_neighbor_search: NDT (DIRECT7), VGICP (DIRECT1)               */
  double resolution;                  /* reg_resolution                   (NDT 0.5 / VGICP 1.0)            */
  double ndt_step_size;               /* pclomp NDT step_size_            (0.1)                            */
  double ndt_outlier_ratio;           /* pclomp NDT outlier_ratio_        (0.55)                           */
  int32_t ndt_min_points_per_voxel;   /* VoxelGridCovariance              (6)                              */
  int32_t ndt_upstream_hd1_sign;      /* 1: reproduce upstream h_ang_d1 = (.., .., +sy); 0: exact 2nd derivative */
  int32_t lm_max_iterations;          /* fast_gicp LM inner tries         (10)                             */
  double lm_init_lambda_factor;       /* fast_gicp                        (1e-9)                           */
  int32_t device_id;                  /* HIP device ordinal                                                */
  int32_t regularization_method;      /* hgs_regularization               (FROBENIUS), FAST_GICP / FAST_VGICP  */
  int32_t ndt_line_search;            /* 0: ndt_omp as it runs (its More-Thuente loop never executes: every step is the
                                         Newton direction with length clamp(|dp|, eps/2, step_size)); 1: a working
                                         More-Thuente search (<= 10 trials, mu 1e-4, nu 0.9) — NOT the reference's result  */
  int32_t reserved;
} hgs_params;

typedef struct hgs_result {
  float final_transformation[16]; /* getFinalTransformation(), column-major                                    */
  int32_t converged;              /* hasConverged()                                                            */
  int32_t iterations;             /* outer iterations executed                                                 */
  double error;                   /* GICP/VGICP: last accepted sum e^T M e; NDT: trans_probability (score/N)  */
  double fitness_score;           /* getFitnessScore(max_range) — filled by the batch entry point, else NaN    */
  uint32_t num_inliers;           /* #source points with d2 <= max_range in that score                         */
  int32_t candidate_id;           /* index into the caller's candidate list (batch), else 0                    */
  int32_t lm_tries;               /* GICP: total LM tries; NDT: derivative passes                              */
  int32_t reserved;
} hgs_result;

typedef struct hgs_handle hgs_handle; /* one registration engine instance (one pcl::Registration object)   */
typedef struct hgs_cloud hgs_cloud;   /* a point cloud resident in HBM (a KeyFrame::cloud, keyframe.hpp:42) */

/* ---- engine life cycle -------------------------------------------------------------------------------- */
/* Fill *p with the factory defaults of registrations.cpp for `method`. */
int hgs_params_default(int32_t method, hgs_params* p);
/* Replaces `new fast_gicp::FastGICP / pclomp::NormalDistributionsTransform` + setters, registrations.cpp:27-36,101-120. */
int hgs_create(const hgs_params* p, hgs_handle** out);
/* Destroys the engine.  Clouds created on it that the caller has not destroyed yet are ORPHANED: their device memory is
 * released with the engine, the hgs_cloud objects stay valid for exactly one more call — hgs_cloud_destroy — and every other
 * entry point rejects them with HGS_ERR_INVALID_ARGUMENT (hgs_cloud_size reports 0).  Either order of destruction is safe. */
int hgs_destroy(hgs_handle* h);
const char* hgs_last_error(const hgs_handle* h); /* h may be NULL: error of the last failed hgs_create on this thread */
int hgs_abi_version(void);

/* ---- clouds resident on the device -------------------------------------------------------------------- */
/* Upload a host cloud: records of `stride_bytes` with x, y, z at floats 0..2 and, from 20 bytes on, the intensity at float 4
 * (pcl::PointXYZI: 32).  `pts` may be pageable memory and may be freed or overwritten as soon as the call returns: the host packs the
 * records into pinned chunks while it reads them and the DMA + device-side packing run behind it on the handle's stream (no
 * synchronisation; an error of those shows up at the next call that waits).  The cloud caches its search structure and
 * covariances once an engine has computed them, like fast_gicp keeps them per input pointer.  A cloud belongs to the
 * handle that created it (its memory is recycled in that handle's stream order): using it with another handle is
 * rejected with HGS_ERR_INVALID_ARGUMENT.  The library does not change the process environment; the number of hardware
 * queues the HIP runtime spreads streams over (GPU_MAX_HW_QUEUES, read once at HIP's initialisation) is the launcher's to set:
 * INTEGRATION.md. */
int hgs_cloud_create(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes, hgs_cloud** out);
int hgs_cloud_destroy(hgs_cloud* c);
size_t hgs_cloud_size(const hgs_cloud* c);
/* Device memory the cloud currently holds, in bytes: points, search index, covariances, correspondence seeds and — once it has served as an NDT /
 * VGICP target — its voxel tables.  What a cache of resident keyframes (adapters/loop_match_hip.hpp; the reference keeps every KeyFrame::cloud for the
 * life of the graph, include/hdl_graph_slam/keyframe.hpp:44) budgets against.  0 for NULL or an orphaned cloud. */
size_t hgs_cloud_device_bytes(const hgs_cloud* c);
/* Drop cached search structure / covariances (forces the cold path again; used by benchmarks). */
int hgs_cloud_invalidate(hgs_cloud* c);

/* ---- pcl::Registration surface (methods the callers use: SURVEY §8b) ---------------------------------- */
/* setInputTarget — scan_matching_odometry_nodelet.cpp:172,246 ; loop_detector.hpp:122 */
int hgs_set_target(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes);
int hgs_set_target_cloud(hgs_handle* h, hgs_cloud* c); /* borrowed, must outlive its use */
/* setInputSource — scan_matching_odometry_nodelet.cpp:177 ; loop_detector.hpp:136 */
int hgs_set_source(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes);
int hgs_set_source_cloud(hgs_handle* h, hgs_cloud* c);
/* align(output, guess) + hasConverged() + getFinalTransformation() — odometry_nodelet.cpp:210-220 ; loop_detector.hpp:143-153 */
int hgs_align(hgs_handle* h, const float guess[16], hgs_result* out);
/* The `output` cloud of align(): out_pts[i].xyz = T * source[i].xyz (records of stride_bytes; other bytes untouched). */
int hgs_transform_source(hgs_handle* h, const float T[16], void* out_pts, size_t stride_bytes);
/* getFitnessScore(max_range) with PCL's semantics (information_matrix_calculator.cpp:49-80): mean of the
 * squared 1-NN distances d2 over source points with d2 <= max_range (sic), DBL_MAX if none. */
int hgs_fitness(hgs_handle* h, const float T[16], double max_range, double* score, uint32_t* num_inliers);
/* getSearchMethodTarget()->nearestKSearch(pt, 1, ...) for a batch of query points (odometry_nodelet.cpp:314-321). */
int hgs_nn_target(hgs_handle* h, const float* q_xyz, size_t nq, size_t stride_bytes, int32_t* idx, float* d2);

/* ---- loop-closure batch: LoopDetector::matching, loop_detector.hpp:117-171 ----------------------------- */
/* Registers every candidate (source) against the handle's current target with its own guess, then evaluates
 * getFitnessScore(max_range) for each.  out[i] is filled for all candidates.  *best receives the index the
 * sequential rule of loop_detector.hpp:147-153 selects (skip non-converged, skip score > best, ties replace ->
 * the LAST minimal candidate wins), or -1 if none. */
int hgs_loop_match_batch(hgs_handle* h, hgs_cloud* const* candidates, size_t n_candidates, const float* guesses /* 16*n */,
                         double max_range, hgs_result* out, int32_t* best);
/* Pure host helper: the sequential selection rule above applied to an arbitrary record list (used after the
 * multi-GPU all-gather of per-candidate records). */
int hgs_select_best(const hgs_result* records, size_t n, int32_t* best);

/* ---- the same batch sharded over the GPUs of a node, one process (one engine) per GPU — SURVEY §8e --------------------
 * The caller partitions the n_total candidates of a detection over the ranks any way it likes — typically by a stable keyframe
 * identity (keyframe id mod world), so that a keyframe's resident cloud, index and covariances stay on one GPU across
 * detections; a rank may hold none.  Every rank holds the query keyframe as its target
 * (hgs_set_target*: replicated, rebuilding its 1 MB index per rank is cheaper than broadcasting it), registers ITS candidates
 * and contributes their records to ONE all-gather of fixed-size hgs_result records (RCCL ncclAllGather over xGMI, issued on the
 * engine's stream directly behind the kernels that produced the records: no torch, no host staging); afterwards every rank
 * holds all n_total records in candidate order and applies the sequential rule of loop_detector.hpp:146-153 to them.
 * This replaces the loop of loop_detector.hpp:135-154 on a multi-GPU node.
 *
 * hgs_comm_get_unique_id: rank 0 creates the 128-byte RCCL id and hands it to the other ranks out of band (MPI, a file,
 * torch.distributed.broadcast_object_list — whatever launched the processes).  hgs_comm_init is collective over the ranks. */
#define HGS_COMM_UNIQUE_ID_BYTES 128
int hgs_comm_get_unique_id(void* id_out /* HGS_COMM_UNIQUE_ID_BYTES */);
int hgs_comm_init(hgs_handle* h, int32_t rank, int32_t world, const void* unique_id /* HGS_COMM_UNIQUE_ID_BYTES */);
int hgs_comm_finalize(hgs_handle* h); /* also done by hgs_destroy */
/* candidates / candidate_ids / guesses describe THIS rank's n_mine candidates (candidate_ids[i] in [0, n_total) is the position
 * of candidates[i] in the detection's candidate list); all_out receives n_total records, all_out[c].candidate_id == c;
 * *best as hgs_loop_match_batch.  Collective: every rank of the communicator must call it, with the same n_total.
 * Failure behaviour (a collective must not hang the other SLAM processes): only argument errors a correct caller makes on
 * every rank alike (null all_out, n_total == 0, no hgs_comm_init) return before the exchange.  A rank whose OWN share is
 * unusable (no target, a candidate that is not a cloud of this engine, duplicated clouds, a failed launch) still takes part:
 * it contributes no records, the peers see its candidates as "not converged" (fitness DBL_MAX), and it returns its error
 * AFTER the exchange.  A candidate id reported more than once makes every rank return HGS_ERR_INVALID_ARGUMENT (first report
 * kept).  A C++ exception on a rank where its batch runs is treated like any other failure of its share (padding, error
 * after the exchange); a rank that cannot allocate the exchange buffers, whose collective call fails, or that leaves the
 * region between the two collectives any other way aborts ITS communicator (ncclCommAbort).  RCCL does not promise that a
 * peer's abort ends this rank's own all-gather kernel on the intra-node transports, so no rank ever waits behind a
 * collective with a blocking synchronise: it polls the stream, asks ncclCommGetAsyncError, and after a deadline
 * (environment HGS_COMM_TIMEOUT_MS, default 60000, 0 = none; it has to cover the skew with which the processes enter the
 * detection) aborts its own communicator and returns HGS_ERR_COMM; hgs_comm_init creates a new one.  What has run: the
 * two-rank paths on the emulated communicator of tests/emul; on real RCCL only world size 1.
 * Traffic: one 16-byte header per rank (shard size, status) and max(shard size) record slots per rank. */
int hgs_loop_match_batch_sharded(hgs_handle* h, hgs_cloud* const* candidates, size_t n_mine, const int32_t* candidate_ids,
                                 const float* guesses /* 16*n_mine */, size_t n_total, double max_range, hgs_result* all_out,
                                 int32_t* best);

/* ---- "next" row f1: InformationMatrixCalculator::calc_fitness_score (information_matrix_calculator.cpp:49-80) */
int hgs_calc_fitness_score(hgs_handle* h, hgs_cloud* cloud1, hgs_cloud* cloud2, const float relpose[16], double max_range, double* score);

/* ---- "next" row f2: the prefilter in front of the path (apps/prefiltering_nodelet.cpp:131-182) -------------------- */
enum hgs_downsample_method { HGS_DOWNSAMPLE_NONE = 0, HGS_DOWNSAMPLE_VOXELGRID = 1, HGS_DOWNSAMPLE_APPROX_VOXELGRID = 2 }; /* :51-72; pcl::VoxelGrid / pcl::ApproximateVoxelGrid */
/* Deviation from PCL, stated: HGS_DOWNSAMPLE_APPROX_VOXELGRID drops non-finite input points, pcl::ApproximateVoxelGrid has no
 * finiteness check (a NaN point hashes into some bucket and poisons that voxel's centroid).  With use_distance_filter = 1 (the
 * nodelet's default) the distance filter in front has removed them already and the outputs are identical; they differ only for
 * use_distance_filter = 0 on an is_dense = false cloud, where PCL's output contains NaN centroids. */
enum hgs_outlier_removal { HGS_OUTLIER_NONE = 0, HGS_OUTLIER_STATISTICAL = 1, HGS_OUTLIER_RADIUS = 2 }; /* :73-93 */
typedef struct hgs_prefilter_params {
  int32_t use_distance_filter;     /* use_distance_filter   (true)   :94                                  */
  int32_t downsample_method;       /* downsample_method     (VOXELGRID)                                    */
  double distance_near_thresh;     /* distance_near_thresh  (1.0)    :95                                  */
  double distance_far_thresh;      /* distance_far_thresh   (100.0)  :96                                  */
  double downsample_resolution;    /* downsample_resolution (0.1)    :53                                  */
  int32_t outlier_removal_method;  /* outlier_removal_method (STATISTICAL)                                 */
  int32_t statistical_mean_k;      /* statistical_mean_k    (20)                                           */
  double statistical_stddev;       /* statistical_stddev    (1.0)                                          */
  double radius_radius;            /* radius_radius         (0.8)                                          */
  int32_t radius_min_neighbors;    /* radius_min_neighbors  (2)                                            */
  int32_t reserved;
} hgs_prefilter_params;
int hgs_prefilter_params_default(hgs_prefilter_params* p);
/* distance_filter -> downsample -> outlier_removal of PrefilteringNodelet::cloud_callback (:131-133) on the device.
 * Input: pcl::PointXYZI records (intensity = float 4 of each record when stride >= 20).  The result stays resident as
 * an hgs_cloud that can be handed to hgs_set_source_cloud / hgs_set_target_cloud directly (no second upload) and
 * fetched with hgs_cloud_download. */
int hgs_prefilter(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes, const hgs_prefilter_params* p, hgs_cloud** out);
/* The same with the deskewing step of cloud_callback in front (apps/prefiltering_nodelet.cpp:112, :182-243, rosparam
 * "deskewing"): point i of the n input points is rotated back by the first-order rotation the sensor made during
 * scan_period * i / n at the gyro rate of ONE sensor_msgs/Imu sample — imu_angular_velocity[3] = that message's
 * angular_velocity (the nodelet takes the first queued sample stamped after the cloud, else the newest; choosing it stays
 * host logic).  NULL = the nodelet's empty imu_queue: no deskewing.  scan_period: rosparam "scan_period" (0.1). */
int hgs_prefilter_deskewed(hgs_handle* h, const void* pts, size_t n, size_t stride_bytes, const hgs_prefilter_params* p, const double* imu_angular_velocity,
                           double scan_period, hgs_cloud** out);
/* Copy a resident cloud back: out_pts[i] = {x, y, z, (1.0), intensity, ...} with the PointXYZI layout for stride >= 20,
 * packed xyz(+w) otherwise.  Needs room for hgs_cloud_size(c) records. */
int hgs_cloud_download(hgs_cloud* c, void* out_pts, size_t stride_bytes);

/* ---- "next" row f3: MapCloudGenerator::generate (src/hdl_graph_slam/map_cloud_generator.cpp:13-51) ------------------ */
/* Every resident keyframe cloud transformed by its pose (float 4x4, column-major, 16 floats each) and concatenated; with
 * resolution > 0 what pcl::octree::OctreePointCloud(resolution).addPointsFromInputCloud() + getOccupiedVoxelCenters() return
 * (:39-44): the octree's bounding box is replayed in input order (the first finite point on the corner of a 2x2x2 root, every
 * point outside doubling the box towards it), the voxel centres (intensity 0) come in the tree's depth-first order; else the
 * transformed points with their intensity.  The map stays resident (*out); fetch it with hgs_cloud_download.  The octree
 * may be up to 21 levels deep (2^21 voxels per axis: 0.01 m over 20 km); HGS_ERR_INVALID_ARGUMENT beyond that. */
int hgs_map_cloud_generate(hgs_handle* h, hgs_cloud* const* keyframes, const float* poses, size_t n_keyframes, double resolution, hgs_cloud** out);

/* ---- measurement ---------------------------------------------------------------------------------------- */
enum hgs_stage {
  HGS_STAGE_UPLOAD = 0,     /* H2D + pack                                            */
  HGS_STAGE_INDEX = 1,      /* spatial sort + bounding-interval tree build           */
  HGS_STAGE_COVARIANCE = 2, /* kNN covariance pre-pass (GICP/VGICP)                  */
  HGS_STAGE_VOXELIZE = 3,   /* Gaussian voxel table build (NDT / VGICP)              */
  HGS_STAGE_LINEARIZE = 4,  /* correspondence search + J^T M J accumulation (GICP) / NDT derivatives */
  HGS_STAGE_ERROR = 5,      /* LM trial error evaluation                             */
  HGS_STAGE_SOLVE = 6,      /* 6x6 reductions + solve + LM/Newton update             */
  HGS_STAGE_FITNESS = 7,    /* fitness score NN pass                                 */
  HGS_STAGE_PREFILTER = 8,  /* distance filter + voxel grid + outlier removal        */
  HGS_STAGE_COUNT = 9
};
/* Enable/disable hipEvent bracketing of every kernel stage on the handle's stream (adds sync cost when read). */
int hgs_profile_enable(hgs_handle* h, int enabled);
/* Accumulated GPU milliseconds and launch counts per stage since the last reset; arrays of HGS_STAGE_COUNT. */
int hgs_profile_read(hgs_handle* h, double* ms, uint64_t* launches, int reset);
/* Block until everything enqueued on the handle's stream has finished. */
int hgs_synchronize(hgs_handle* h);

/* ---- stage-level hooks (parity tests: one kernel stage at a caller-chosen pose; not used by the adapters) ------ */
/* GICP covariances of the current target, original point order: out6[n][6] = xx,xy,xz,yy,yz,zz (float as stored). */
int hgs_debug_target_covariances(hgs_handle* h, float* out6);
/* One fused correspondence + linearisation pass at pose T (double, row-major 3x4): H[36] row-major, b[6], sum of
 * errors, and per source point (original order) the original target index of its correspondence or -1. */
int hgs_debug_gicp_linearize(hgs_handle* h, const double T12[12], double* H36, double* b6, double* err, int32_t* corr);
/* Valid Gaussian cells of the NDT target (any order): linear key, grid coordinates, mean, inverse covariance, count. */
int hgs_debug_ndt_cells(hgs_handle* h, int32_t cap, int32_t* ijk3, double* mean3, float* icov6, int32_t* npts, int32_t* n_cells);
/* One NDT derivative pass at p = (tx,ty,tz,rx,ry,rz): score, gradient[6], Hessian[36]. */
int hgs_debug_ndt_derivatives(hgs_handle* h, const double p6[6], double* score, double* g6, double* H36);
/* the pure-host merge step of hgs_loop_match_batch_sharded: `gathered` = world blocks of `per` slots, of rank r's block the first
 * counts[r] carry records; fills all_out[0 .. n_total) (a candidate nobody reported: not converged, fitness DBL_MAX);
 * *duplicate_id = an id reported more than once (first report kept) or -1.  Needs no device: callable on a CPU-only box. */
int hgs_debug_merge_shard_records(const hgs_result* gathered, const int32_t* counts, int32_t world, size_t per, size_t n_total,
                                  hgs_result* all_out, int32_t* duplicate_id);
/* A measurement / test knob of one engine: "batch_lanes", "lane_start", "ndt_sort", "cov_split", "resident_descs", "knn_qpw_tiny",
 * "seed_grid", "knn_replay", "ndt_resident", "ndt_chunk", "hilbert_levels", "nn_qpw", "nn_qpw16_below", "nn_qpw32_below", "fused_rounds", "fused_rounds_below", "fused_rounds_max_problems", "fused_rounds_max_blocks", "early_result", "early_run_ahead",
 * "knn_tiny_below", "upload_trace", "prefilter_fast".  Results never depend on them (the tests
 * that force a code path check exactly that); A/B runs and those tests are the only callers — the library reads no tuning variable from the environment. */
int hgs_debug_set_option(hgs_handle* h, const char* key, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* HGS_REGISTRATION_H */
