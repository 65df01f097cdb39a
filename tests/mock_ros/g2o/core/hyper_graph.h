// Minimal stand-in for <g2o/core/hyper_graph.h> (TEST ONLY): hdl_graph_slam/graph_slam.hpp only names the types; KeyFrame::load
// (src/hdl_graph_slam/keyframe.cpp:127-136) looks its vertex up in `vertices()` (g2o: an id -> Vertex* hash map) and dynamic_casts it.
// (The real header drags in the standard containers; loop_detector.hpp uses std::deque without including it.)
#pragma once
#include <deque>
#include <map>
#include <set>
#include <unordered_map>
#include <vector>
namespace g2o {
class HyperGraph {
public:
  class Edge {};
  class Vertex {
  public:
    virtual ~Vertex() = default;
    int id() const { return id_; }
    void setId(int i) { id_ = i; }

  protected:
    int id_ = 0;
  };
  using VertexIDMap = std::unordered_map<int, Vertex*>;
  VertexIDMap& vertices() { return vertices_; }
  const VertexIDMap& vertices() const { return vertices_; }

private:
  VertexIDMap vertices_;
};
}  // namespace g2o
