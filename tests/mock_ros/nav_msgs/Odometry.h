// Minimal stand-in for <nav_msgs/Odometry.h> (TEST ONLY).
#pragma once
#include <geometry_msgs/Pose.h>
namespace nav_msgs {
struct Odometry {
  std_msgs::Header header;
  std::string child_frame_id;
  geometry_msgs::PoseWithCovariance pose;
  geometry_msgs::TwistWithCovariance twist;
};
using OdometryConstPtr = std::shared_ptr<const Odometry>;
}  // namespace nav_msgs
