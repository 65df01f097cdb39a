// registration_hip.hpp — header-only pcl::Registration adapter over the C-ABI of include/hgs_registration.h.
//
// This is the object hdl_graph_slam::select_registration_method (src/hdl_graph_slam/registrations.cpp:22-124) returns for
// the new registration_method values "FAST_GICP_HIP", "FAST_VGICP_HIP" and "NDT_HIP" (see INTEGRATION.md for the patch).
// The nodelets keep calling the pcl::Registration surface they already use:
//     setInputTarget / setInputSource / align / hasConverged / getFinalTransformation / getFitnessScore /
//     getSearchMethodTarget()->nearestKSearch
// and every numerically heavy step runs in the HIP library.  The class is logic-free: it forwards and copies results.
//
// PCL facts relied on (PCL 1.8-1.12): setInputSource / setInputTarget are virtual; computeTransformation(output, guess) is the
// pure virtual called by the non-virtual align(), which has already run initCompute() and copied *input_ into `output`; converged_,
// final_transformation_, nr_iterations_ are protected members read by hasConverged() / getFinalTransformation().
//
// What an integrated system pays per align() besides the device work, and what this adapter does about each item:
//   * initCompute() calls tree_->setInputCloud(target_) on every NEW target — with PCL's own tree a KdTreeFLANN build on the CPU
//     (milliseconds for 13 k points, tens of ms for 119 k) that the device engines never use.  The adapter installs a LazyKdTree
//     (below) with setSearchMethodTarget(): its setInputCloud() only remembers the cloud, and the FLANN index is built at the first
//     nearestKSearch / radiusSearch — i.e. only if somebody really calls the NON-virtual getFitnessScore() /
//     getSearchMethodTarget() through the base pointer.  Semantics unchanged, cost moved to the callers that want it.  (Callers
//     patched by integration/hdl_graph_slam_hip.patch use fitnessScoreHIP() / nearestTargetHIP() instead and never build it.)
//   * uploads are deferred to the call that needs them: setInputTarget / setInputSource only keep the shared pointer (as PCL does);
//     a target that is replaced before it is ever aligned against, or an adapter whose batch path is served by LoopMatcherHIP,
//     never uploads (nor even creates its engine).
//   * the aligned cloud: pcl::Registration's contract is that align() fills `output` with T * input.  That is a D2H of the whole
//     cloud (3.8 MB at 119 k points).  LoopDetector::matching discards it (loop_detector.hpp:134,143): setAlignedCloudMode(ALIGNED_CLOUD_NONE)
//     skips it (output then stays the copy of the input that align() made); ALIGNED_CLOUD_HOST transforms that host copy in place instead of
//     downloading (pcl::transformPointCloud's arithmetic, as the reference's engines do).  Default: ALIGNED_CLOUD_AUTO (host below 49152 points).
//
// Not compilable against the real PCL in this repository's image (no PCL / ROS); tests/test_adapter_cpp.py compiles it against
// a stand-in of the pcl::Registration / pcl::search::KdTree interfaces (tests/mock_pcl) and runs it on the GPU through the real library.
//
// Life cycle: a setter that changes an engine parameter after the engine exists drops the engine (recreate()); the next call that needs
// one creates it and uploads the clouds pcl::Registration holds.  hgs_cloud handles a caller obtained through nativeHandle() belong to
// the destroyed engine — hgs_destroy orphans them (safe to hgs_cloud_destroy, rejected by every other call), so configure the
// object fully (the factory does) before caching device clouds.
#pragma once

#include <atomic>
#include <cstring>
#include <limits>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/search/kdtree.h>
#include <pcl/registration/registration.h>

#include "hgs_registration.h"

namespace hgs_hip {

// pcl::search::KdTree whose index is built at the first query instead of at setInputCloud() (see the header comment).  Thread-safe
// the way PCL's tree is used: queries are const and may come from several threads; the one-time build is guarded.
template <typename PointT>
class LazyKdTree : public pcl::search::KdTree<PointT> {
  using Base = pcl::search::KdTree<PointT>;

public:
  using PointCloudConstPtr = typename Base::PointCloudConstPtr;
  using IndicesConstPtr = typename Base::IndicesConstPtr;
#if PCL_VERSION_COMPARE(>=, 1, 12, 0)
  bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override {
    remember(cloud, indices);
    return true;
  }
#else
  void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override { remember(cloud, indices); }
#endif
  int nearestKSearch(const PointT& p, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const override {
    ensure_built();
    return Base::nearestKSearch(p, k, k_indices, k_sqr_distances);
  }
  int radiusSearch(const PointT& p, double radius, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const override {
    ensure_built();
    return Base::radiusSearch(p, radius, k_indices, k_sqr_distances, max_nn);
  }
  bool built() const { return built_.load(std::memory_order_acquire); }

private:
  void remember(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices) {
    std::lock_guard<std::mutex> lock(mutex_);
    this->input_ = cloud;      // what getInputCloud() returns
    this->indices_ = indices;
    built_.store(false, std::memory_order_release);
  }
  void ensure_built() const {
    if (built_.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lock(mutex_);
    if (built_.load(std::memory_order_relaxed)) return;
    const_cast<LazyKdTree*>(this)->Base::setInputCloud(this->input_, this->indices_);  // the real (FLANN) build, once per target that is queried
    built_.store(true, std::memory_order_release);
  }
  mutable std::mutex mutex_;
  mutable std::atomic<bool> built_{false};
};

template <typename PointSource, typename PointTarget>
class RegistrationHIP : public pcl::Registration<PointSource, PointTarget, float> {
public:
  using Base = pcl::Registration<PointSource, PointTarget, float>;
  using Matrix4 = typename Base::Matrix4;
  using PointCloudSource = typename Base::PointCloudSource;
  using PointCloudSourceConstPtr = typename Base::PointCloudSourceConstPtr;
  using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;
#if PCL_VERSION >= PCL_VERSION_CALC(1, 10, 0)
  using Ptr = pcl::shared_ptr<RegistrationHIP<PointSource, PointTarget>>;
#else
  using Ptr = boost::shared_ptr<RegistrationHIP<PointSource, PointTarget>>;
#endif

  // `method`: HGS_FAST_GICP / HGS_FAST_VGICP / HGS_NDT_OMP.  Parameters start at the factory defaults of registrations.cpp.
  explicit RegistrationHIP(int method, int device_id = 0) {
    this->reg_name_ = method == HGS_NDT_OMP ? "hgs_hip::NDT" : (method == HGS_FAST_VGICP ? "hgs_hip::FastVGICP" : "hgs_hip::FastGICP");
    if (hgs_params_default(method, &params_) != HGS_OK) throw std::invalid_argument("RegistrationHIP: unknown method");
    params_.device_id = device_id;
    lazy_tree_.reset(new LazyKdTree<PointTarget>());
    this->setSearchMethodTarget(lazy_tree_);  // initCompute() now "builds" this one: a pointer copy
  }
  ~RegistrationHIP() override { hgs_destroy(handle_); }
  RegistrationHIP(const RegistrationHIP&) = delete;
  RegistrationHIP& operator=(const RegistrationHIP&) = delete;

  // ---- setters the factory calls (registrations.cpp:30-34,41-44,51-55,107-119). They must precede the first cloud.
  void setNumThreads(int) {}  // reg_num_threads has no meaning on the device
  // extension: the GPU this engine runs on (the constructor's device_id); the clouds already set follow it to the new device
  void setDevice(int device_id) {
    params_.device_id = device_id;
    recreate();
  }
  void setTransformationEpsilon(double eps) { params_.transformation_epsilon = eps; Base::setTransformationEpsilon(eps); recreate(); }
  void setMaximumIterations(int n) { params_.max_iterations = n; Base::setMaximumIterations(n); recreate(); }
  void setMaxCorrespondenceDistance(double d) { params_.max_correspondence_distance = d; Base::setMaxCorrespondenceDistance(d); recreate(); }
  void setCorrespondenceRandomness(int k) { params_.correspondence_randomness = k; recreate(); }
  void setResolution(double r) { params_.resolution = r; recreate(); }
  void setNeighborhoodSearchMethod(int hgs_neighbor_search_value) { params_.neighbor_search = hgs_neighbor_search_value; recreate(); }
  void setRotationEpsilon(double eps) { params_.rotation_epsilon = eps; recreate(); }
  // extension: 1 = a working More-Thuente line search in NDT (ndt_omp itself never runs its loop; 0 reproduces it)
  void setNdtLineSearch(bool on) { params_.ndt_line_search = on ? 1 : 0; recreate(); }
  // fast_gicp::FastGICP::setRegularizationMethod (hgs_regularization value); never called by hdl_graph_slam
  void setRegularizationMethod(int hgs_regularization_value) { params_.regularization_method = hgs_regularization_value; recreate(); }

  // extension: how align() fills `output` (PCL's contract: T * input).
  //   ALIGNED_CLOUD_DEVICE            the device transforms the resident source and the result comes down (hgs_transform_source)
  //   ALIGNED_CLOUD_HOST              the adapter transforms align()'s host copy of the input in place, with the float arithmetic of
  //                                   pcl::transformPointCloud — what fast_gicp / ndt_omp themselves do at the end of computeTransformation; no
  //                                   device round trip (cheaper for the 10-40 k-point clouds the odometry nodelet handles: DESIGN.md 1.1)
  //   ALIGNED_CLOUD_NONE              `output` stays the copy of the input that align() made (callers that discard it: LoopDetector::matching)
  //   ALIGNED_CLOUD_AUTO (default)    HOST below 49152 points, DEVICE from there on (where the download, spread over the library's pack threads, wins)
  enum AlignedCloudMode { ALIGNED_CLOUD_DEVICE = 0, ALIGNED_CLOUD_HOST = 1, ALIGNED_CLOUD_NONE = 2, ALIGNED_CLOUD_AUTO = 3 };
  void setAlignedCloudMode(AlignedCloudMode m) { aligned_mode_ = m; }
  // the rosparam form (reg_hip_aligned_cloud): "auto" | "device" | "host" | "none"; "true" = device, "false" = none, anything else = auto
  void setAlignedCloudMode(const std::string& m) {
    aligned_mode_ = (m == "host") ? ALIGNED_CLOUD_HOST : (m == "device" || m == "true") ? ALIGNED_CLOUD_DEVICE : ((m == "none" || m == "false") ? ALIGNED_CLOUD_NONE : ALIGNED_CLOUD_AUTO);
  }
  void setAlignedCloudOutput(bool on) { aligned_mode_ = on ? ALIGNED_CLOUD_DEVICE : ALIGNED_CLOUD_NONE; }
  const hgs_params& params() const { return params_; }
  // true once somebody has queried PCL's CPU tree of the current target through the base pointer (tests, diagnostics)
  bool cpuTreeBuilt() const { return lazy_tree_->built(); }

  // ---- pcl::Registration virtuals: keep the pointer (fast_gicp: same pointer -> keep the cached structures); the upload is deferred
  void setInputSource(const PointCloudSourceConstPtr& cloud) override {
    if (cloud == this->input_) return;
    Base::setInputSource(cloud);
    uploaded_source_ = nullptr;  // (the identity below is only ever compared with the cloud the base class keeps alive)
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {
    if (cloud == this->target_) return;
    Base::setInputTarget(cloud);
    uploaded_target_ = nullptr;
  }

  // ---- device versions of the two non-virtual queries the callers use
  // getFitnessScore(max_range): apps/scan_matching_odometry_nodelet.cpp:307, include/hdl_graph_slam/loop_detector.hpp:146
  double fitnessScoreHIP(double max_range = std::numeric_limits<double>::max()) {
    double score = std::numeric_limits<double>::max();
    hgs_handle* h = handle();
    if (h && !uploadsCurrent()) return score;  // (logged by handle(): the engine holds older clouds than pcl::Registration does)
    check(hgs_fitness(h, this->final_transformation_.data(), max_range, &score, nullptr), "hgs_fitness");
    return score;
  }
  // getSearchMethodTarget()->nearestKSearch(pt, 1, ...) over a whole cloud: apps/scan_matching_odometry_nodelet.cpp:314-321
  void nearestTargetHIP(const pcl::PointCloud<PointSource>& queries, std::vector<int>& indices, std::vector<float>& sq_dists) {
    indices.assign(queries.size(), -1);
    sq_dists.assign(queries.size(), std::numeric_limits<float>::max());
    hgs_handle* h = handle();
    if (h && !uploadsCurrent()) return;  // no neighbour found: index -1 (the patched status path counts no inlier)
    check(hgs_nn_target(h, reinterpret_cast<const float*>(queries.points.data()), queries.size(), sizeof(PointSource), indices.data(),
                        sq_dists.data()),
          "hgs_nn_target");
  }
  hgs_handle* nativeHandle() { return handle(); }  // (engine created and the current clouds uploaded if they were not yet)
  const hgs_result& lastResult() const { return last_; }

protected:
  void computeTransformation(PointCloudSource& output, const Matrix4& guess) override {
    this->converged_ = false;
    hgs_handle* h = handle();
    // a cloud that could not be uploaded must not be replaced silently by the one the engine still holds from the previous sweep / keyframe
    const int rc = (h && !uploadsCurrent()) ? (int)HGS_ERR_HIP : hgs_align(h, guess.data(), &last_);
    if (rc != HGS_OK) {  // failure is signalled the way the callers expect: hasConverged() == false, pose unchanged
      PCL_ERROR("[%s] %s\n", this->reg_name_.c_str(), (h && !uploadsCurrent()) ? "the current clouds are not on the device (upload failed)" : hgs_last_error(handle_));
      this->final_transformation_ = guess;
      return;
    }
    std::memcpy(this->final_transformation_.data(), last_.final_transformation, sizeof(float) * 16);
    this->transformation_ = this->final_transformation_;
    this->converged_ = last_.converged != 0;
    this->nr_iterations_ = last_.iterations;
    // align() copied *input_ into output; overwrite xyz with T * input (other fields are kept)
    const AlignedCloudMode mode = aligned_mode_ != ALIGNED_CLOUD_AUTO ? aligned_mode_ : (output.points.size() < 49152 ? ALIGNED_CLOUD_HOST : ALIGNED_CLOUD_DEVICE);
    if (mode == ALIGNED_CLOUD_DEVICE) {
      check(hgs_transform_source(handle_, last_.final_transformation, output.points.data(), sizeof(PointSource)), "hgs_transform_source");
    } else if (mode == ALIGNED_CLOUD_HOST) {
      const float* T = last_.final_transformation;  // column-major
      for (auto& p : output.points) {
        const float x = p.x, y = p.y, z = p.z;
        p.x = T[0] * x + T[4] * y + T[8] * z + T[12];
        p.y = T[1] * x + T[5] * y + T[9] * z + T[13];
        p.z = T[2] * x + T[6] * y + T[10] * z + T[14];
      }
    }
  }

private:
  // Never throws into the nodelet: when the engine cannot be created (no usable HIP device, out of memory) this logs and returns
  // nullptr; every C-ABI entry point rejects a null handle, so setInputSource / setInputTarget log a second line and align()
  // ends with hasConverged() == false and the guess as the final transformation — the failure signal the callers test
  // (apps/scan_matching_odometry_nodelet.cpp:214, include/hdl_graph_slam/loop_detector.hpp:147).  Creation is retried on the
  // next call.
  // The clouds pcl::Registration holds are uploaded when they differ from what the engine has (deferred from setInputTarget /
  // setInputSource; also what makes a cloud set during a transient creation failure, or before a parameter change, reach the new engine:
  // scan_matching_odometry sets its keyframe ONCE and only replaces it after a successful match).
  hgs_handle* handle() {
    if (!handle_) {
      if (hgs_create(&params_, &handle_) != HGS_OK) {
        PCL_ERROR("[%s] hgs_create failed: %s\n", this->reg_name_.c_str(), hgs_last_error(nullptr));
        handle_ = nullptr;
        return nullptr;
      }
      uploaded_target_ = nullptr, uploaded_source_ = nullptr;
    }
    if (this->target_ && this->target_.get() != uploaded_target_) {
      if (check(hgs_set_target(handle_, this->target_->points.data(), this->target_->points.size(), sizeof(PointTarget)), "hgs_set_target"))
        uploaded_target_ = this->target_.get();
    }
    if (this->input_ && this->input_.get() != uploaded_source_) {
      if (check(hgs_set_source(handle_, this->input_->points.data(), this->input_->points.size(), sizeof(PointSource)), "hgs_set_source"))
        uploaded_source_ = this->input_.get();
    }
    return handle_;
  }
  // true when the engine holds exactly the clouds pcl::Registration holds (handle() uploads what differs; a failed upload leaves them different)
  bool uploadsCurrent() const {
    return (!this->target_ || this->target_.get() == uploaded_target_) && (!this->input_ || this->input_.get() == uploaded_source_);
  }
  void recreate() {  // parameters are fixed at creation: drop the engine; the next handle() creates one and uploads the current clouds
    if (!handle_) return;
    hgs_destroy(handle_);
    handle_ = nullptr;
    uploaded_target_ = nullptr, uploaded_source_ = nullptr;
  }
  bool check(int rc, const char* what) {
    if (rc != HGS_OK) PCL_ERROR("[%s] %s failed (%d): %s\n", this->reg_name_.c_str(), what, rc, hgs_last_error(handle_));
    return rc == HGS_OK;
  }

  hgs_params params_{};
  hgs_handle* handle_ = nullptr;
  hgs_result last_{};
  AlignedCloudMode aligned_mode_ = ALIGNED_CLOUD_AUTO;
  const void* uploaded_target_ = nullptr;  // the clouds the engine currently holds (identity only, never dereferenced)
  const void* uploaded_source_ = nullptr;
#if PCL_VERSION_COMPARE(>=, 1, 10, 0)
  pcl::shared_ptr<LazyKdTree<PointTarget>> lazy_tree_;
#else
  boost::shared_ptr<LazyKdTree<PointTarget>> lazy_tree_;
#endif
};

}  // namespace hgs_hip
