// Minimal stand-in for <pcl_ros/transforms.h> (TEST ONLY): the base_link_frame branch of PrefilteringNodelet::cloud_callback is not exercised; identity.
#pragma once
#include <pcl/point_cloud.h>
#include <tf/transform_listener.h>
namespace pcl_ros {
template <typename PointT>
void transformPointCloud(const pcl::PointCloud<PointT>& in, pcl::PointCloud<PointT>& out, const tf::StampedTransform&) {
  out = in;
}
}  // namespace pcl_ros
