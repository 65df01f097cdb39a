// resident_clouds_hip.hpp — header-only bindings of the three "next" rows of SURVEY §8(f) that sit beside the registration path:
//
//   f1  InformationMatrixCalculator::calc_fitness_score   src/hdl_graph_slam/information_matrix_calculator.cpp:49-80   -> hgs_calc_fitness_score
//   f2  PrefilteringNodelet::cloud_callback :131-133      apps/prefiltering_nodelet.cpp (distance filter, downsample, outlier removal) -> hgs_prefilter
//   f3  MapCloudGenerator::generate                       src/hdl_graph_slam/map_cloud_generator.cpp:13-51              -> hgs_map_cloud_generate
//
// integration/hdl_graph_slam_hip.patch calls them from the reference's own functions behind #ifdef USE_HGS_HIP; every call returns false when the
// device path could not run (no engine, out of memory, an argument the device refuses) and the reference's CPU code below the hunk runs instead —
// the same "fall through to what was there" rule as LoopDetector::matching's batch hunk.
//
// f1 and f3 work on KEYFRAME clouds: `pcl::PointCloud::ConstPtr`s the pose graph keeps for its whole life (include/hdl_graph_slam/keyframe.hpp:44).
// calc_fitness_score is a static function that sees two cloud pointers and nothing else, so the device copies live in a process-wide cache keyed by
// the cloud OBJECT: the cache holds the shared_ptr (the address cannot be reused for another cloud while its device copy exists), uploads a cloud the
// first time it is seen and releases the least recently used ones beyond a byte budget (setCapacity; default 8 GiB of the 288 GB).  The search index a
// cloud gets as `cloud1` of one edge is reused by every later edge and map update that names it.
//
// Thread safety: one mutex around the engine (hdl_graph_slam_nodelet calls these from its optimisation timer, its map timer and its services).
// Like registration_hip.hpp this header cannot be compiled against the real PCL in this repository's image; tests/cpp/integration_main.cpp compiles
// it against tests/mock_pcl and runs it inside the patched reference functions (host emulation of the kernels on the CPU, libhgs_hip.so with -m gpu).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include "hgs_registration.h"

namespace hgs_hip {

template <typename PointT>
class ResidentCloudsHIP {
public:
  using Cloud = pcl::PointCloud<PointT>;
  using CloudConstPtr = typename Cloud::ConstPtr;

  // the process-wide instance the static calc_fitness_score reaches
  static ResidentCloudsHIP& instance() {
    static ResidentCloudsHIP* self = new ResidentCloudsHIP();  // never destroyed: at static-destruction time the HIP runtime may be gone already
    return *self;
  }
  ResidentCloudsHIP() = default;
  ~ResidentCloudsHIP() { release(); }
  ResidentCloudsHIP(const ResidentCloudsHIP&) = delete;
  ResidentCloudsHIP& operator=(const ResidentCloudsHIP&) = delete;

  // rosparams `reg_hip_device` (0) and `reg_hip_next_rows` (true) of the nodelet that constructs InformationMatrixCalculator / MapCloudGenerator
  void configure(int device_id, bool enabled) {
    std::lock_guard<std::mutex> lock(mutex_);
    if (device_id != device_id_ && handle_) release_locked();
    device_id_ = device_id, enabled_ = enabled;
  }
  void setCapacity(size_t max_bytes) {
    std::lock_guard<std::mutex> lock(mutex_);
    max_bytes_ = max_bytes;
  }
  bool enabled() const { return enabled_; }
  size_t resident() const { return clouds_.size(); }
  size_t resident_bytes() const { return bytes_; }
  size_t device_calls() const { return device_calls_; }  // successful device-side calls (tests: "the hunk really took the device path")
  const std::string& last_error() const { return last_error_; }

  // f1 — information_matrix_calculator.cpp:49-80.  `relpose_colmajor`: relpose.cast<float>().matrix().data()
  bool fitness_score(const CloudConstPtr& cloud1, const CloudConstPtr& cloud2, const float relpose_colmajor[16], double max_range, double* score) {
    std::lock_guard<std::mutex> lock(mutex_);
    if (!enabled_ || !cloud1 || !cloud2 || cloud1->points.empty()) return false;  // (an empty target: PCL's kd-tree path decides what that means)
    if (cloud2->points.empty()) {  // nr == 0 -> std::numeric_limits<double>::max() (:76-79); nothing to launch
      *score = 1.7976931348623157e308;
      return true;
    }
    if (!engine()) return false;
    const uint64_t tick = ++tick_;
    hgs_cloud* c1 = resident_cloud(cloud1, tick);
    hgs_cloud* c2 = c1 ? resident_cloud(cloud2, tick) : nullptr;
    if (!c1 || !c2) return false;
    const bool ok = check(hgs_calc_fitness_score(handle_, c1, c2, relpose_colmajor, max_range, score), "hgs_calc_fitness_score");
    enforce_capacity(tick);
    device_calls_ += ok ? 1 : 0;
    return ok;
  }

  // f3 — map_cloud_generator.cpp:13-51 for resolution > 0: the voxel centres of pcl::octree in the octree's own order, intensity 0.
  // `poses_colmajor`: keyframe->pose.matrix().cast<float>() of every snapshot, 16 floats each.
  bool map_cloud(const std::vector<CloudConstPtr>& keyframe_clouds, const std::vector<float>& poses_colmajor, double resolution, Cloud& out) {
    std::lock_guard<std::mutex> lock(mutex_);
    if (!enabled_ || keyframe_clouds.empty() || poses_colmajor.size() != keyframe_clouds.size() * 16 || !(resolution > 0.0)) return false;
    if (!engine()) return false;
    const uint64_t tick = ++tick_;
    std::vector<hgs_cloud*> dev;
    for (const CloudConstPtr& c : keyframe_clouds) {
      hgs_cloud* d = c ? resident_cloud(c, tick) : nullptr;
      if (!d) return false;
      dev.push_back(d);
    }
    hgs_cloud* map = nullptr;
    bool ok = check(hgs_map_cloud_generate(handle_, dev.data(), poses_colmajor.data(), dev.size(), resolution, &map), "hgs_map_cloud_generate");
    if (ok) {
      out.points.resize(hgs_cloud_size(map));
      ok = out.points.empty() || check(hgs_cloud_download(map, out.points.data(), sizeof(PointT)), "hgs_cloud_download");
    }
    if (map) hgs_cloud_destroy(map);
    enforce_capacity(tick);
    device_calls_ += ok ? 1 : 0;
    return ok;
  }

  // f2 — apps/prefiltering_nodelet.cpp:131-133 (+ the deskewing step of :112,182-243 when `imu_angular_velocity` is given): the filtered cloud comes back
  // as PointXYZI records (x, y, z, 1, intensity).  The input is a sensor sweep, not a keyframe: it is not cached.
  bool prefilter(const Cloud& src, const hgs_prefilter_params& params, const double* imu_angular_velocity, double scan_period, Cloud& out) {
    std::lock_guard<std::mutex> lock(mutex_);
    if (!enabled_ || src.points.empty()) return false;
    if (!engine()) return false;
    hgs_cloud* filtered = nullptr;
    bool ok = check(imu_angular_velocity ? hgs_prefilter_deskewed(handle_, src.points.data(), src.points.size(), sizeof(PointT), &params, imu_angular_velocity, scan_period, &filtered)
                                         : hgs_prefilter(handle_, src.points.data(), src.points.size(), sizeof(PointT), &params, &filtered),
                    "hgs_prefilter");
    if (ok) {
      out.points.resize(hgs_cloud_size(filtered));
      ok = out.points.empty() || check(hgs_cloud_download(filtered, out.points.data(), sizeof(PointT)), "hgs_cloud_download");
    }
    if (filtered) hgs_cloud_destroy(filtered);
    device_calls_ += ok ? 1 : 0;
    return ok;
  }

  // the rosparams of PrefilteringNodelet::initialize_params (apps/prefiltering_nodelet.cpp:50-99) as an hgs_prefilter_params
  static hgs_prefilter_params prefilter_params(const std::string& downsample_method, double downsample_resolution, const std::string& outlier_removal_method, int statistical_mean_k,
                                               double statistical_stddev, double radius_radius, int radius_min_neighbors, bool use_distance_filter, double distance_near_thresh,
                                               double distance_far_thresh) {
    hgs_prefilter_params p;
    hgs_prefilter_params_default(&p);
    p.downsample_method = downsample_method == "VOXELGRID" ? HGS_DOWNSAMPLE_VOXELGRID : (downsample_method == "APPROX_VOXELGRID" ? HGS_DOWNSAMPLE_APPROX_VOXELGRID : HGS_DOWNSAMPLE_NONE);
    p.downsample_resolution = downsample_resolution;
    p.outlier_removal_method = outlier_removal_method == "STATISTICAL" ? HGS_OUTLIER_STATISTICAL : (outlier_removal_method == "RADIUS" ? HGS_OUTLIER_RADIUS : HGS_OUTLIER_NONE);
    p.statistical_mean_k = statistical_mean_k, p.statistical_stddev = statistical_stddev;
    p.radius_radius = radius_radius, p.radius_min_neighbors = radius_min_neighbors;
    p.use_distance_filter = use_distance_filter ? 1 : 0;
    p.distance_near_thresh = distance_near_thresh, p.distance_far_thresh = distance_far_thresh;
    return p;
  }

  void forget_all() {
    std::lock_guard<std::mutex> lock(mutex_);
    while (!clouds_.empty()) drop(clouds_.begin());
  }

private:
  struct Entry {
    CloudConstPtr keep;   // holds the host cloud: its address stays unique while the device copy exists
    hgs_cloud* dev;
    uint64_t last_used;
    size_t bytes;
  };
  std::mutex mutex_;
  hgs_handle* handle_ = nullptr;
  int device_id_ = 0;
  bool enabled_ = true;
  bool create_failed_ = false;
  std::unordered_map<const void*, Entry> clouds_;
  size_t bytes_ = 0, max_bytes_ = (size_t)8 << 30, device_calls_ = 0;
  uint64_t tick_ = 0;
  std::string last_error_;

  bool engine() {
    if (handle_) return true;
    if (create_failed_) return false;  // do not retry (and log) on every edge
    hgs_params p;
    hgs_params_default(HGS_FAST_GICP, &p);  // the engine only serves searches and the map cloud: no registration parameter matters
    p.device_id = device_id_;
    if (hgs_create(&p, &handle_) != HGS_OK) {
      last_error_ = std::string("hgs_create: ") + hgs_last_error(nullptr);
      handle_ = nullptr, create_failed_ = true;
      return false;
    }
    return true;
  }
  hgs_cloud* resident_cloud(const CloudConstPtr& c, uint64_t tick) {
    auto it = clouds_.find(c.get());
    if (it == clouds_.end()) {
      hgs_cloud* d = nullptr;
      if (!check(hgs_cloud_create(handle_, c->points.data(), c->points.size(), sizeof(PointT), &d), "hgs_cloud_create")) return nullptr;
      it = clouds_.emplace(c.get(), Entry{c, d, tick, 0}).first;
    }
    it->second.last_used = tick;
    return it->second.dev;
  }
  void drop(typename std::unordered_map<const void*, Entry>::iterator it) {
    bytes_ -= std::min(bytes_, it->second.bytes);
    hgs_cloud_destroy(it->second.dev);
    clouds_.erase(it);
  }
  void enforce_capacity(uint64_t tick) {
    for (auto& kv : clouds_) {
      if (kv.second.last_used != tick) continue;
      const size_t b = hgs_cloud_device_bytes(kv.second.dev);
      bytes_ = bytes_ - std::min(bytes_, kv.second.bytes) + b;
      kv.second.bytes = b;
    }
    if (!max_bytes_ || bytes_ <= max_bytes_) return;
    std::vector<std::pair<uint64_t, const void*>> order;
    for (const auto& kv : clouds_) order.emplace_back(kv.second.last_used, kv.first);
    std::sort(order.begin(), order.end());
    for (const auto& o : order) {
      if (bytes_ <= max_bytes_) break;
      drop(clouds_.find(o.second));
    }
  }
  bool check(int rc, const char* what) {
    if (rc != HGS_OK) last_error_ = std::string(what) + ": " + hgs_last_error(handle_);
    return rc == HGS_OK;
  }
  void release_locked() {
    while (!clouds_.empty()) drop(clouds_.begin());
    if (handle_) hgs_destroy(handle_);
    handle_ = nullptr, create_failed_ = false;
  }
  void release() {
    std::lock_guard<std::mutex> lock(mutex_);
    release_locked();
  }
};

}  // namespace hgs_hip
