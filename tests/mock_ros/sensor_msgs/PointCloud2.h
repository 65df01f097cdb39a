// Minimal stand-in for <sensor_msgs/PointCloud2.h> (TEST ONLY): the header the odometry nodelet reads + the cloud itself as the payload
// (the serialised form is ROS plumbing; pcl::fromROSMsg below hands the payload over).
#pragma once
#include <memory>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct PointCloud2 {
  std_msgs::Header header;
  pcl::PointCloud<pcl::PointXYZI> mock_payload;
};
using PointCloud2ConstPtr = std::shared_ptr<const PointCloud2>;
}  // namespace sensor_msgs
namespace pcl {
template <typename PointT>
void fromROSMsg(const sensor_msgs::PointCloud2& msg, PointCloud<PointT>& cloud) {
  cloud = msg.mock_payload;
  cloud.header.frame_id = msg.header.frame_id;
  cloud.header.stamp = (unsigned long long)msg.header.stamp.sec * 1000000ull + msg.header.stamp.nsec / 1000u;
}
}  // namespace pcl
