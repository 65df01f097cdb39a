// Minimal stand-in for <tf/transform_listener.h> (TEST ONLY): the nodelets only consult tf when base_link_frame / enable_robot_odometry_init_guess are set,
// which the tests leave off; every lookup fails the way a missing transform does.
#pragma once
#include <string>
#include <ros/time.h>
namespace tf {
struct Vector3 {
  double x_ = 0, y_ = 0, z_ = 0;
  double x() const { return x_; }
  double y() const { return y_; }
  double z() const { return z_; }
};
struct Quaternion {
  double x_ = 0, y_ = 0, z_ = 0, w_ = 1;
  double x() const { return x_; }
  double y() const { return y_; }
  double z() const { return z_; }
  double w() const { return w_; }
};
struct StampedTransform {
  ros::Time stamp_;
  Vector3 getOrigin() const { return Vector3(); }
  Quaternion getRotation() const { return Quaternion(); }
};
class TransformListener {
public:
  bool canTransform(const std::string&, const std::string&, const ros::Time&) const { return false; }
  bool waitForTransform(const std::string&, const std::string&, const ros::Time&, const ros::Duration&) const { return false; }
  void lookupTransform(const std::string&, const std::string&, const ros::Time&, StampedTransform&) const {}
  bool waitForTransform(const std::string&, const ros::Time&, const std::string&, const ros::Time&, const std::string&, const ros::Duration&) const { return false; }
  void lookupTransform(const std::string&, const ros::Time&, const std::string&, const ros::Time&, const std::string&, StampedTransform&) const {}
};
}  // namespace tf
