// Minimal stand-in for <tf/transform_listener.h> (TEST ONLY): PrefilteringNodelet only uses it when base_link_frame is set, which the tests leave empty.
#pragma once
#include <string>
#include <ros/time.h>
namespace tf {
struct StampedTransform {};
class TransformListener {
public:
  bool canTransform(const std::string&, const std::string&, const ros::Time&) const { return false; }
  bool waitForTransform(const std::string&, const std::string&, const ros::Time&, const ros::Duration&) const { return false; }
  void lookupTransform(const std::string&, const std::string&, const ros::Time&, StampedTransform&) const {}
};
}  // namespace tf
