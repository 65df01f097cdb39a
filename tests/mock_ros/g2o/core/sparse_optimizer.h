// Minimal stand-in for <g2o/core/sparse_optimizer.h> (TEST ONLY): src/hdl_graph_slam/keyframe.cpp includes it and uses nothing from it.
#pragma once
#include "hyper_graph.h"
namespace g2o {
class SparseOptimizer : public HyperGraph {};
}  // namespace g2o
