// Minimal stand-in for <pcl/point_cloud.h> (TEST ONLY).
#pragma once
#include <memory>
#include <vector>
#define PCL_VERSION_CALC(a, b, c) ((a) * 100000 + (b) * 100 + (c))
#define PCL_VERSION PCL_VERSION_CALC(1, 10, 0)
namespace pcl {
template <typename T>
using shared_ptr = std::shared_ptr<T>;
template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  size_t size() const { return points.size(); }
};
}  // namespace pcl
