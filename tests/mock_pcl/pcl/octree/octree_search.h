// Minimal stand-in for <pcl/octree/octree_search.h> (TEST ONLY): pcl::octree::OctreePointCloud as MapCloudGenerator::generate uses it
// (src/hdl_graph_slam/map_cloud_generator.cpp:39-44: constructor(resolution), setInputCloud, addPointsFromInputCloud, getOccupiedVoxelCenters).
// The voxel centres come from the oracle's restatement of PCL's octree (oracle/mapcloud.hpp: bounding box replayed in input order, depth-first
// leaf order) — this header only gives it PCL's interface, so that the UNPATCHED reference function can run next to the patched one in the tests.
#pragma once
#include <array>
#include <vector>
#include "../point_cloud.h"
#include "../point_types.h"
#include "../../../../oracle/mapcloud.hpp"
namespace pcl {
namespace octree {
template <typename PointT>
class OctreePointCloud {
public:
  explicit OctreePointCloud(double resolution) : resolution_(resolution) {}
  void setInputCloud(const typename pcl::PointCloud<PointT>::ConstPtr& cloud) { input_ = cloud; }
  void addPointsFromInputCloud() {
    std::vector<hgso::PfPoint> pts;
    for (const PointT& p : input_->points) pts.push_back({p.x, p.y, p.z, p.intensity});
    const std::array<float, 16> identity = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    centres_.clear();
    hgso::map_cloud_generate({pts}, {identity}, resolution_, centres_);
  }
  template <typename Vec>
  int getOccupiedVoxelCenters(Vec& out) const {
    out.clear();
    for (const hgso::PfPoint& c : centres_) {
      PointT p{};
      p.x = c.x, p.y = c.y, p.z = c.z, p.data3 = 1.0f;
      out.push_back(p);
    }
    return (int)out.size();
  }

private:
  double resolution_;
  typename pcl::PointCloud<PointT>::ConstPtr input_;
  std::vector<hgso::PfPoint> centres_;
};
}  // namespace octree
}  // namespace pcl
