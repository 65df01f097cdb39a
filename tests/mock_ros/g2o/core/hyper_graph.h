// Minimal stand-in for <g2o/core/hyper_graph.h> (TEST ONLY): hdl_graph_slam/graph_slam.hpp only names the types.  (The real header drags in
// the standard containers; loop_detector.hpp uses std::deque without including it.)
#pragma once
#include <deque>
#include <map>
#include <set>
#include <vector>
namespace g2o {
class HyperGraph {
public:
  class Edge {};
  class Vertex {};
};
class SparseOptimizer;
}  // namespace g2o
