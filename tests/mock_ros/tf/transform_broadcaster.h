// Minimal stand-in for <tf/transform_broadcaster.h> (TEST ONLY): remembers what was sent (the tests read the last transform per child frame).
#pragma once
#include <map>
#include <string>
#include <geometry_msgs/TransformStamped.h>
namespace tf {
class TransformBroadcaster {
public:
  static std::map<std::string, geometry_msgs::TransformStamped>& sent() {
    static std::map<std::string, geometry_msgs::TransformStamped> m;
    return m;
  }
  void sendTransform(const geometry_msgs::TransformStamped& t) { sent()[t.child_frame_id] = t; }
};
}  // namespace tf
