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
// pure virtual called by the non-virtual align(), which has already copied *input_ into `output`; converged_,
// final_transformation_, nr_iterations_ are protected members read by hasConverged() / getFinalTransformation().
// getFitnessScore() and getSearchMethodTarget() are NON-virtual and keep working on PCL's own CPU kd-tree (tree_), which
// align() builds in initCompute(); callers that want them on the device use fitnessScoreHIP() / nearestTargetHIP() below
// (LoopDetector's batched path in INTEGRATION.md does).
//
// Not compilable against the real PCL in this repository's image (no PCL / ROS); tests/test_adapter_cpp.py compiles it against
// a minimal stand-in of the pcl::Registration interface (tests/mock_pcl) and runs it on the GPU through the real library.
//
// Life cycle: a setter that changes an engine parameter after the engine exists re-creates the engine (recreate()); the
// adapter's own target / source are uploaded again, but hgs_cloud handles a caller obtained through nativeHandle() belong to
// the destroyed engine — hgs_destroy orphans them (safe to hgs_cloud_destroy, rejected by every other call), so configure the
// object fully (the factory does) before caching device clouds.
#pragma once

#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>

#include "hgs_registration.h"

namespace hgs_hip {

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

  // ---- pcl::Registration virtuals
  void setInputSource(const PointCloudSourceConstPtr& cloud) override {
    if (cloud == this->input_ && handle_) return;  // fast_gicp: same pointer -> keep the cached structures
    Base::setInputSource(cloud);
    const bool fresh = !handle_;  // a handle created by this call has uploaded the clouds pcl::Registration holds, this one included
    if (hgs_handle* hh = handle()) {
      if (!fresh) check(hgs_set_source(hh, cloud->points.data(), cloud->points.size(), sizeof(PointSource)), "hgs_set_source");
    } else {
      check(HGS_ERR_NO_DEVICE, "hgs_set_source");
    }
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {
    if (cloud == this->target_ && handle_) return;
    Base::setInputTarget(cloud);
    const bool fresh = !handle_;
    if (hgs_handle* hh = handle()) {
      if (!fresh) check(hgs_set_target(hh, cloud->points.data(), cloud->points.size(), sizeof(PointTarget)), "hgs_set_target");
    } else {
      check(HGS_ERR_NO_DEVICE, "hgs_set_target");
    }
  }

  // ---- device versions of the two non-virtual queries the callers use
  // getFitnessScore(max_range): apps/scan_matching_odometry_nodelet.cpp:307, include/hdl_graph_slam/loop_detector.hpp:146
  double fitnessScoreHIP(double max_range = std::numeric_limits<double>::max()) {
    double score = std::numeric_limits<double>::max();
    check(hgs_fitness(handle(), this->final_transformation_.data(), max_range, &score, nullptr), "hgs_fitness");
    return score;
  }
  // getSearchMethodTarget()->nearestKSearch(pt, 1, ...) over a whole cloud: apps/scan_matching_odometry_nodelet.cpp:314-321
  void nearestTargetHIP(const pcl::PointCloud<PointSource>& queries, std::vector<int>& indices, std::vector<float>& sq_dists) {
    indices.resize(queries.size());
    sq_dists.resize(queries.size());
    check(hgs_nn_target(handle(), reinterpret_cast<const float*>(queries.points.data()), queries.size(), sizeof(PointSource), indices.data(),
                        sq_dists.data()),
          "hgs_nn_target");
  }
  hgs_handle* nativeHandle() { return handle(); }
  const hgs_result& lastResult() const { return last_; }

protected:
  void computeTransformation(PointCloudSource& output, const Matrix4& guess) override {
    this->converged_ = false;
    const int rc = hgs_align(handle(), guess.data(), &last_);
    if (rc != HGS_OK) {  // failure is signalled the way the callers expect: hasConverged() == false, pose unchanged
      PCL_ERROR("[%s] %s\n", this->reg_name_.c_str(), hgs_last_error(handle_));
      this->final_transformation_ = guess;
      return;
    }
    std::memcpy(this->final_transformation_.data(), last_.final_transformation, sizeof(float) * 16);
    this->transformation_ = this->final_transformation_;
    this->converged_ = last_.converged != 0;
    this->nr_iterations_ = last_.iterations;
    // align() copied *input_ into output; overwrite xyz with T * input (other fields are kept)
    hgs_transform_source(handle(), last_.final_transformation, output.points.data(), sizeof(PointSource));
  }

private:
  // Never throws into the nodelet: when the engine cannot be created (no usable HIP device, out of memory) this logs and returns
  // nullptr; every C-ABI entry point rejects a null handle, so setInputSource / setInputTarget log a second line and align()
  // ends with hasConverged() == false and the guess as the final transformation — the failure signal the callers test
  // (apps/scan_matching_odometry_nodelet.cpp:214, include/hdl_graph_slam/loop_detector.hpp:147).  Creation is retried on the
  // next call.
  // A (re)created engine holds no clouds: the ones pcl::Registration already has (set while creation was failing, or before a parameter
  // change) are uploaded right away — otherwise a keyframe set during one transient failure would be lost for good, because a repeated
  // setInputTarget with the same pointer returns early and scan_matching_odometry only replaces the keyframe after a successful match.
  hgs_handle* handle() {
    if (handle_) return handle_;
    if (hgs_create(&params_, &handle_) != HGS_OK) {
      PCL_ERROR("[%s] hgs_create failed: %s\n", this->reg_name_.c_str(), hgs_last_error(nullptr));
      handle_ = nullptr;
      return nullptr;
    }
    if (this->target_) check(hgs_set_target(handle_, this->target_->points.data(), this->target_->points.size(), sizeof(PointTarget)), "hgs_set_target");
    if (this->input_) check(hgs_set_source(handle_, this->input_->points.data(), this->input_->points.size(), sizeof(PointSource)), "hgs_set_source");
    return handle_;
  }
  void recreate() {  // parameters are fixed at creation: drop the engine; handle() uploads the current clouds into the new one
    if (!handle_) return;
    hgs_destroy(handle_);
    handle_ = nullptr;
    (void)handle();
  }
  void check(int rc, const char* what) {
    if (rc != HGS_OK) PCL_ERROR("[%s] %s failed (%d): %s\n", this->reg_name_.c_str(), what, rc, hgs_last_error(handle_));
  }

  hgs_params params_{};
  hgs_handle* handle_ = nullptr;
  hgs_result last_{};
};

}  // namespace hgs_hip
