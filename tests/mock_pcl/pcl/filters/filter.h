// Minimal stand-ins for PCL's filters (TEST ONLY) as apps/prefiltering_nodelet.cpp:50-99,159-182 uses them: pcl::Filter<PointT> (setInputCloud, filter) and
// VoxelGrid / ApproximateVoxelGrid / RadiusOutlierRemoval / StatisticalOutlierRemoval / PassThrough.  The filtering itself is the oracle's restatement of PCL
// (oracle/prefilter.hpp) behind PCL's interface, so that the UNPATCHED filter chain of the nodelet can run next to the device path in the tests.
#pragma once
#include <memory>
#include <vector>
#include "../point_cloud.h"
#include "../point_types.h"
#include "../../../../oracle/prefilter.hpp"
namespace pcl {
template <typename PointT>
class Filter {
public:
  using Ptr = std::shared_ptr<Filter<PointT>>;
  virtual ~Filter() = default;
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& cloud) { input_ = cloud; }
  void filter(PointCloud<PointT>& out) {
    std::vector<hgso::PfPoint> in;
    for (const PointT& p : input_->points) in.push_back({p.x, p.y, p.z, p.intensity});
    const std::vector<hgso::PfPoint> res = apply(in);
    out.points.clear();
    for (const hgso::PfPoint& q : res) {
      PointT p{};
      p.x = q.x, p.y = q.y, p.z = q.z, p.data3 = 1.0f, p.intensity = q.intensity;
      out.points.push_back(p);
    }
    out.width = (unsigned)out.points.size(), out.height = 1, out.is_dense = true;
  }

protected:
  virtual std::vector<hgso::PfPoint> apply(const std::vector<hgso::PfPoint>& in) = 0;
  typename PointCloud<PointT>::ConstPtr input_;
};
template <typename PointT>
class PassThrough : public Filter<PointT> {
protected:
  std::vector<hgso::PfPoint> apply(const std::vector<hgso::PfPoint>& in) override { return in; }
};
template <typename PointT>
class VoxelGrid : public Filter<PointT> {
public:
  using Ptr = std::shared_ptr<VoxelGrid<PointT>>;
  void setLeafSize(float x, float, float) { leaf_ = x; }

protected:
  std::vector<hgso::PfPoint> apply(const std::vector<hgso::PfPoint>& in) override {
    std::vector<hgso::PfPoint> out;
    if (!hgso::pf_voxelgrid(in, leaf_, out)) out = in;  // (pcl::VoxelGrid on index overflow: warns and returns the input)
    return out;
  }
  double leaf_ = 0.1;
};
template <typename PointT>
class ApproximateVoxelGrid : public Filter<PointT> {
public:
  using Ptr = std::shared_ptr<ApproximateVoxelGrid<PointT>>;
  void setLeafSize(float x, float, float) { leaf_ = x; }

protected:
  std::vector<hgso::PfPoint> apply(const std::vector<hgso::PfPoint>& in) override {
    std::vector<hgso::PfPoint> out;
    hgso::pf_approx_voxelgrid(in, leaf_, out);
    return out;
  }
  double leaf_ = 0.1;
};
template <typename PointT>
class RadiusOutlierRemoval : public Filter<PointT> {
public:
  using Ptr = std::shared_ptr<RadiusOutlierRemoval<PointT>>;
  void setRadiusSearch(double r) { radius_ = r; }
  void setMinNeighborsInRadius(int n) { min_ = n; }

protected:
  std::vector<hgso::PfPoint> apply(const std::vector<hgso::PfPoint>& in) override { return hgso::pf_radius_outlier_removal(in, radius_, min_); }
  double radius_ = 0.8;
  int min_ = 2;
};
template <typename PointT>
class StatisticalOutlierRemoval : public Filter<PointT> {
public:
  using Ptr = std::shared_ptr<StatisticalOutlierRemoval<PointT>>;
  void setMeanK(int k) { k_ = k; }
  void setStddevMulThresh(double s) { s_ = s; }

protected:
  std::vector<hgso::PfPoint> apply(const std::vector<hgso::PfPoint>& in) override { return hgso::pf_statistical_outlier_removal(in, k_, s_); }
  int k_ = 20;
  double s_ = 1.0;
};
}  // namespace pcl
