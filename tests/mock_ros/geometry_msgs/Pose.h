// Minimal stand-ins for the geometry_msgs types hdl_graph_slam's odometry nodelet and ros_utils.hpp use (TEST ONLY).
#pragma once
#include <memory>
#include <string>
#include <std_msgs/Header.h>
namespace geometry_msgs {
struct Quaternion {
  double x = 0, y = 0, z = 0, w = 1;
};
struct Vector3 {
  double x = 0, y = 0, z = 0;
};
struct Point {
  double x = 0, y = 0, z = 0;
};
struct Pose {
  Point position;
  Quaternion orientation;
};
struct Transform {
  Vector3 translation;
  Quaternion rotation;
};
struct TransformStamped {
  std_msgs::Header header;
  std::string child_frame_id;
  Transform transform;
};
struct PoseWithCovariance {
  Pose pose;
};
struct PoseWithCovarianceStamped {
  std_msgs::Header header;
  PoseWithCovariance pose;
};
using PoseWithCovarianceStampedConstPtr = std::shared_ptr<const PoseWithCovarianceStamped>;
struct Twist {
  Vector3 linear, angular;
};
struct TwistWithCovariance {
  Twist twist;
};
}  // namespace geometry_msgs
