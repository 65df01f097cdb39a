// Minimal stand-in for <g2o/types/slam3d/vertex_se3.h> (TEST ONLY): the pose-graph node a KeyFrame points at (keyframe.hpp:51);
// LoopDetector reads estimate() and, through KeyFrame::id(), id(); KeyFrame::load reaches it by dynamic_cast from HyperGraph::Vertex.
#pragma once
#include <Eigen/Dense>
#include <g2o/core/hyper_graph.h>
namespace g2o {
class VertexSE3 : public HyperGraph::Vertex {
public:
  const Eigen::Isometry3d& estimate() const { return estimate_; }
  void setEstimate(const Eigen::Isometry3d& e) { estimate_ = e; }

private:
  Eigen::Isometry3d estimate_;
};
}  // namespace g2o
