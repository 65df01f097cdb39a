// Minimal stand-in for <pcl/point_types.h> (TEST ONLY): the memory layout of pcl::PointXYZI (32 bytes).
#pragma once
#include <Eigen/Dense>  // (the real header pulls Eigen in; hdl_graph_slam/keyframe.hpp relies on that)
namespace pcl {
struct alignas(16) PointXYZI {
  float x, y, z, data3;
  float intensity, pad[3];
};
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is 32 bytes");
}  // namespace pcl
