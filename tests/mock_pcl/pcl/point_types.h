// Minimal stand-in for <pcl/point_types.h> (TEST ONLY): the memory layout of pcl::PointXYZI (32 bytes).
#pragma once
#include <cmath>
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
// Eigen::Map<Vector3f> over x, y, z: norm() (PrefilteringNodelet::distance_filter, apps/prefiltering_nodelet.cpp:170: Eigen's sqrt of the float sum of
// squares), conversion to a Vector3f and assignment from one (the deskewing step, :236-240)
struct Vector3fMapConst {
  const float* p;
  float norm() const { return std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); }
  operator Eigen::Vector3f() const { return Eigen::Vector3f(p[0], p[1], p[2]); }
};
struct Vector3fMap {
  float* p;
  Vector3fMap& operator=(const Eigen::Vector3f& v) {
    p[0] = v(0), p[1] = v(1), p[2] = v(2);
    return *this;
  }
};
inline Eigen::Vector3f operator*(const Eigen::Quaternionf& q, const Vector3fMapConst& v) { return q * Eigen::Vector3f(v); }
struct alignas(16) PointXYZI {
  float x, y, z, data3;
  float intensity, pad[3];
  Vector4fMap getVector4fMap() { return Vector4fMap{&x}; }
  Vector4fMapConst getVector4fMap() const { return Vector4fMapConst{&x}; }
  Vector3fMap getVector3fMap() { return Vector3fMap{&x}; }
  Vector3fMapConst getVector3fMap() const { return Vector3fMapConst{&x}; }
};
struct alignas(16) PointXYZRGB {
  float x, y, z, data3;
  unsigned char b, g, r, a;
  float pad[3];
  struct Map4 {
    float* p;
    Map4& operator=(const Vector4fMapConst& v) {
      for (int i = 0; i < 4; i++) p[i] = v.p[i];
      return *this;
    }
  };
  Map4 getVector4fMap() { return Map4{&x}; }
};
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is 32 bytes");
}  // namespace pcl
