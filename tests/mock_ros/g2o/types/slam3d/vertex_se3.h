// Minimal stand-in for <g2o/types/slam3d/vertex_se3.h> (TEST ONLY): the pose-graph node a KeyFrame points at (keyframe.hpp:51);
// LoopDetector reads estimate() and, through KeyFrame::id(), id().
#pragma once
#include <Eigen/Dense>
namespace g2o {
class VertexSE3 {
public:
  const Eigen::Isometry3d& estimate() const { return estimate_; }
  void setEstimate(const Eigen::Isometry3d& e) { estimate_ = e; }
  int id() const { return id_; }
  void setId(int i) { id_ = i; }

private:
  Eigen::Isometry3d estimate_;
  int id_ = 0;
};
}  // namespace g2o
