// Minimal stand-ins for std_msgs/Header, String, Time (TEST ONLY).
#pragma once
#include <memory>
#include <string>
#include <ros/time.h>
namespace std_msgs {
struct Header {
  unsigned seq = 0;
  ros::Time stamp;
  std::string frame_id;
};
using HeaderPtr = std::shared_ptr<Header>;
struct String {
  std::string data;
};
}  // namespace std_msgs
