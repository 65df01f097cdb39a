// Minimal stand-in for <pcl/registration/registration.h> (TEST ONLY): the members and the align() / initCompute() /
// getFitnessScore() contract of pcl::Registration (PCL 1.10 registration.h / impl/registration.hpp) that
// adapters/registration_hip.hpp and the reference's callers rely on:
//   * setInputSource / setInputTarget are virtual; setInputTarget marks the target "updated";
//   * align(output, guess) is NON-virtual: initCompute() (which rebuilds tree_ on an updated target unless force_no_recompute_),
//     output = *input_, data[3] = 1, converged_ = false, final_transformation_ = I, then computeTransformation(output, guess);
//   * getFitnessScore(max_range) and getSearchMethodTarget() are NON-virtual and work on tree_ (virtual calls INTO the tree);
//   * setSearchMethodTarget(tree, force_no_recompute) replaces tree_ and marks the target "updated".
// Matrix4 is Eigen::Matrix4f from tests/mock_eigen (column-major, like Eigen).
#pragma once
#include <cstdio>
#include <limits>
#include <string>
#include <Eigen/Dense>
#include "../point_cloud.h"
#include "../point_types.h"
#include "../search/kdtree.h"
#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
namespace pcl {
using MockMatrix4f = Eigen::Matrix4f;
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
public:
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using KdTree = pcl::search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using Ptr = std::shared_ptr<Registration>;
  Registration() : tree_(new KdTree) {}
  virtual ~Registration() = default;
  virtual void setInputSource(const PointCloudSourceConstPtr& c) { input_ = c; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& c) {
    target_ = c;
    target_cloud_updated_ = true;
  }
  PointCloudSourceConstPtr getInputSource() const { return input_; }
  PointCloudTargetConstPtr getInputTarget() const { return target_; }
  void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false) {
    tree_ = tree;
    force_no_recompute_ = force_no_recompute;
    target_cloud_updated_ = true;
  }
  KdTreePtr getSearchMethodTarget() const { return tree_; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  bool hasConverged() const { return converged_; }
  Matrix4 getFinalTransformation() const { return final_transformation_; }
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double fitness_score = 0.0;
    std::vector<int> nn_indices(1);
    std::vector<float> nn_dists(1);
    int nr = 0;
    const Matrix4& T = final_transformation_;
    for (const PointSource& p : input_->points) {  // pcl::transformPointCloud (the float arithmetic of its scalar path), then the 1-NN in the target
      PointSource q = p;
      q.x = T(0, 0) * p.x + T(0, 1) * p.y + T(0, 2) * p.z + T(0, 3);
      q.y = T(1, 0) * p.x + T(1, 1) * p.y + T(1, 2) * p.z + T(1, 3);
      q.z = T(2, 0) * p.x + T(2, 1) * p.y + T(2, 2) * p.z + T(2, 3);
      if (tree_->nearestKSearch(q, 1, nn_indices, nn_dists) < 1) continue;
      if (nn_dists[0] <= max_range) {  // (squared distance against the un-squared range: PCL's own quirk)
        fitness_score += nn_dists[0];
        nr++;
      }
    }
    return nr > 0 ? fitness_score / nr : std::numeric_limits<double>::max();
  }
  void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  void align(PointCloudSource& output, const Matrix4& guess) {  // non-virtual, like PCL
    if (!initCompute()) return;
    output.points = input_->points;
    converged_ = false;
    final_transformation_ = transformation_ = Matrix4::Identity();
    for (auto& p : output.points) p.data3 = 1.0f;
    computeTransformation(output, guess);
  }

protected:
  bool initCompute() {
    if (!target_) {
      PCL_ERROR("[pcl::registration::%s::compute] No input target dataset was given!\n", reg_name_.c_str());
      return false;
    }
    if (target_cloud_updated_ && !force_no_recompute_) {  // only update the target kd-tree if a new target cloud was set
      tree_->setInputCloud(target_);
      target_cloud_updated_ = false;
    }
    return (bool)input_;
  }
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
  std::string reg_name_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  KdTreePtr tree_;
  Matrix4 final_transformation_ = Matrix4::Identity(), transformation_ = Matrix4::Identity();
  bool converged_ = false, target_cloud_updated_ = true, force_no_recompute_ = false;
  int nr_iterations_ = 0, max_iterations_ = 10;
  double transformation_epsilon_ = 0, corr_dist_threshold_ = 0;
};
}  // namespace pcl
