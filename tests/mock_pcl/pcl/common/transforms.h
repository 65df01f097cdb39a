// Minimal stand-in for <pcl/common/transforms.h> (TEST ONLY): pcl::transformPointCloud(cloud_in, cloud_out, Eigen::Affine3f) as
// InformationMatrixCalculator::calc_fitness_score calls it (src/hdl_graph_slam/information_matrix_calculator.cpp:57).
//
// The ARITHMETIC ORDER is PCL's, not this repository's (the oracle and the kernels use an fma chain):
//   * PCL >= 1.10 (noetic), pcl/common/impl/transforms.hpp, detail::Transformer<float>::se3 on SSE2 targets:
//         p0 = x * c0;  p1 = y * c1;  p2 = z * c2;  out = p0 + (p1 + (p2 + c3))          (four-wide, unfused)
//   * PCL 1.8 (melodic) and the non-SSE Transformer, with -DHGS_MOCK_PCL_1_8:
//         out = ((m00 * x + m01 * y) + m02 * z) + m03                                     (left to right, unfused)
// Every product and sum is rounded to float (volatile-free: the file that includes this header is compiled with -ffp-contract=off).
// Non-finite input points are copied through untransformed when the cloud is not dense, as PCL does; all fields are copied first
// (copy_all_fields = true), so intensity survives.
#pragma once
#include <cmath>
#include <Eigen/Dense>
#include "../point_cloud.h"
namespace pcl {
namespace detail {
inline void se3(const Eigen::Matrix4f& m, const float* src, float* tgt) {
  for (int r = 0; r < 3; r++) {
#ifdef HGS_MOCK_PCL_1_8
    tgt[r] = ((m(r, 0) * src[0] + m(r, 1) * src[1]) + m(r, 2) * src[2]) + m(r, 3);
#else
    const float p0 = src[0] * m(r, 0), p1 = src[1] * m(r, 1), p2 = src[2] * m(r, 2);
    tgt[r] = p0 + (p1 + (p2 + m(r, 3)));
#endif
  }
#ifndef HGS_MOCK_PCL_1_8
  // the SSE lane 3: x*m30 + (y*m31 + (z*m32 + m33)) = 1 for a rigid transform (data[3] stays 1)
  tgt[3] = src[0] * m(3, 0) + (src[1] * m(3, 1) + (src[2] * m(3, 2) + m(3, 3)));
#endif
}
}  // namespace detail

template <typename PointT>
void transformPointCloud(const PointCloud<PointT>& cloud_in, PointCloud<PointT>& cloud_out, const Eigen::Isometry3f& transform, bool copy_all_fields = true) {
  if (&cloud_in != &cloud_out) {
    cloud_out.points.resize(cloud_in.points.size());
    if (copy_all_fields) cloud_out.points = cloud_in.points;
  }
  const Eigen::Matrix4f& m = transform.matrix();
  for (size_t i = 0; i < cloud_in.points.size(); i++) {
    const PointT& p = cloud_in.points[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    const float src[4] = {p.x, p.y, p.z, 1.0f};
    float tgt[4] = {0, 0, 0, 1.0f};
    detail::se3(m, src, tgt);
    cloud_out.points[i].x = tgt[0], cloud_out.points[i].y = tgt[1], cloud_out.points[i].z = tgt[2], cloud_out.points[i].data3 = tgt[3];
  }
}
// the Matrix4f overload (ScanMatchingOdometryNodelet::matching publishes the aligned cloud with it)
template <typename PointT>
void transformPointCloud(const PointCloud<PointT>& cloud_in, PointCloud<PointT>& cloud_out, const Eigen::Matrix4f& transform, bool copy_all_fields = true) {
  transformPointCloud(cloud_in, cloud_out, Eigen::Isometry3f(transform), copy_all_fields);
}
}  // namespace pcl
