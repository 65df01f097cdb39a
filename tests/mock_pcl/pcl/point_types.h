// Minimal stand-in for <pcl/point_types.h> (TEST ONLY): the memory layout of pcl::PointXYZI (32 bytes).
#pragma once
#include <Eigen/Dense>  // (the real header pulls Eigen in; hdl_graph_slam/keyframe.hpp relies on that)
namespace pcl {
// Eigen::Map<Vector4f> over a point's data[4], as far as MapCloudGenerator::generate needs it (map_cloud_generator.cpp:27:
// `dst_pt.getVector4fMap() = pose * src_pt.getVector4fMap()`); the product is Eigen's coefficient-wise row times vector sum, left to right
struct Vector4fMapConst {
  const float* p;
};
struct Vector4fMap {
  float* p;
  Vector4fMap& operator=(const Eigen::Vector4f& v) {
    for (int i = 0; i < 4; i++) p[i] = v(i);
    return *this;
  }
};
inline Eigen::Vector4f operator*(const Eigen::Matrix4f& m, const Vector4fMapConst& v) {
  Eigen::Vector4f o;
  for (int r = 0; r < 4; r++) o(r) = ((m(r, 0) * v.p[0] + m(r, 1) * v.p[1]) + m(r, 2) * v.p[2]) + m(r, 3) * v.p[3];
  return o;
}
struct alignas(16) PointXYZI {
  float x, y, z, data3;
  float intensity, pad[3];
  Vector4fMap getVector4fMap() { return Vector4fMap{&x}; }
  Vector4fMapConst getVector4fMap() const { return Vector4fMapConst{&x}; }
};
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is 32 bytes");
}  // namespace pcl
