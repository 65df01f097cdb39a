// Minimal stand-in for <sensor_msgs/Imu.h> (TEST ONLY): the fields PrefilteringNodelet::deskewing reads (apps/prefiltering_nodelet.cpp:206-220).
#pragma once
#include <memory>
#include <ros/time.h>
namespace sensor_msgs {
struct Imu {
  struct {
    ros::Time stamp;
  } header;
  struct {
    double x = 0, y = 0, z = 0;
  } angular_velocity;
};
using ImuConstPtr = std::shared_ptr<const Imu>;
}  // namespace sensor_msgs
