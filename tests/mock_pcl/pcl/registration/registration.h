// Minimal stand-in for <pcl/registration/registration.h> (TEST ONLY): the members and the align() contract that
// adapters/registration_hip.hpp relies on.  Matrix4 mimics Eigen::Matrix4f (column-major).
#pragma once
#include <cstdio>
#include <string>
#include "../point_cloud.h"
#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
namespace pcl {
struct MockMatrix4f {
  float d[16];
  float* data() { return d; }
  const float* data() const { return d; }
  static MockMatrix4f Identity() {
    MockMatrix4f m{};
    for (int i = 0; i < 4; i++) m.d[i * 5] = 1.f;
    return m;
  }
  float& operator()(int r, int c) { return d[c * 4 + r]; }
  float operator()(int r, int c) const { return d[c * 4 + r]; }
};
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
public:
  using Matrix4 = MockMatrix4f;
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using Ptr = std::shared_ptr<Registration>;
  virtual ~Registration() = default;
  virtual void setInputSource(const PointCloudSourceConstPtr& c) { input_ = c; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& c) { target_ = c; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  bool hasConverged() const { return converged_; }
  Matrix4 getFinalTransformation() const { return final_transformation_; }
  void align(PointCloudSource& output, const Matrix4& guess) {  // non-virtual, like PCL
    output.points = input_->points;
    for (auto& p : output.points) p.data3 = 1.0f;
    converged_ = false;
    final_transformation_ = transformation_ = Matrix4::Identity();
    computeTransformation(output, guess);
  }
protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
  std::string reg_name_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  Matrix4 final_transformation_ = Matrix4::Identity(), transformation_ = Matrix4::Identity();
  bool converged_ = false;
  int nr_iterations_ = 0, max_iterations_ = 10;
  double transformation_epsilon_ = 0, corr_dist_threshold_ = 0;
};
}  // namespace pcl
