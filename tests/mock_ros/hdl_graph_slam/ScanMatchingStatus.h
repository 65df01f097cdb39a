// Stand-in for the message header ROS generates from msg/ScanMatchingStatus.msg (TEST ONLY): Header header, bool has_converged, float32 matching_error,
// float32 inlier_fraction, geometry_msgs/Pose relative_pose, std_msgs/String[] prediction_labels, geometry_msgs/Pose[] prediction_errors.
#pragma once
#include <vector>
#include <geometry_msgs/Pose.h>
namespace hdl_graph_slam {
struct ScanMatchingStatus {
  std_msgs::Header header;
  bool has_converged = false;
  float matching_error = 0, inlier_fraction = 0;
  geometry_msgs::Pose relative_pose;
  std::vector<std_msgs::String> prediction_labels;
  std::vector<geometry_msgs::Pose> prediction_errors;
};
}  // namespace hdl_graph_slam
