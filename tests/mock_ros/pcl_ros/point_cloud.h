// Minimal stand-in for <pcl_ros/point_cloud.h> (TEST ONLY): lets pcl::PointCloud be a message type; pulls in pcl_conversions::fromPCL (header stamps are microseconds).
#pragma once
#include <cstdint>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/common/transforms.h>  // (the real pcl_ros headers pull PCL's common headers in: the odometry nodelet calls pcl::transformPointCloud without including it)
#include <ros/time.h>
namespace pcl_conversions {
inline ros::Time fromPCL(uint64_t stamp_us) { return ros::Time((uint32_t)(stamp_us / 1000000ull), (uint32_t)(stamp_us % 1000000ull) * 1000u); }
}  // namespace pcl_conversions
