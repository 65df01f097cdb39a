// Minimal stand-in for <pcl/point_cloud.h> (TEST ONLY).
#pragma once
#include <cstddef>
#include <memory>
#include <string>
#include <vector>
#define PCL_VERSION_CALC(a, b, c) ((a) * 100000 + (b) * 100 + (c))
#ifdef HGS_MOCK_PCL_1_12   // the other side of the adapter's PCL_VERSION_COMPARE branches: search::KdTree::setInputCloud returns bool from 1.12 on
#define PCL_VERSION PCL_VERSION_CALC(1, 12, 1)
#else
#define PCL_VERSION PCL_VERSION_CALC(1, 10, 0)
#endif
#define PCL_VERSION_COMPARE(OP, MAJ, MIN, PATCH) (PCL_VERSION OP PCL_VERSION_CALC(MAJ, MIN, PATCH))
namespace pcl {
template <typename T>
using shared_ptr = std::shared_ptr<T>;
using IndicesConstPtr = std::shared_ptr<const std::vector<int>>;
struct PCLHeader {  // pcl::PCLHeader: shared by the clouds of every point type
  unsigned seq = 0;
  unsigned long long stamp = 0;
  std::string frame_id;
};
template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  PCLHeader header;
  unsigned width = 0, height = 0;
  bool is_dense = true;
  Ptr makeShared() const { return std::make_shared<PointCloud<PointT>>(*this); }
  void reserve(size_t n) { points.reserve(n); }
  void push_back(const PointT& p) { points.push_back(p); }
  typename std::vector<PointT>::const_iterator begin() const { return points.begin(); }
  typename std::vector<PointT>::const_iterator end() const { return points.end(); }
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void resize(size_t n) { points.resize(n); }
  const PointT& at(size_t i) const { return points.at(i); }
  PointT& at(size_t i) { return points.at(i); }
  const PointT& operator[](size_t i) const { return points[i]; }
  PointT& operator[](size_t i) { return points[i]; }
};
}  // namespace pcl
