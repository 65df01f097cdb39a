// Minimal stand-in for <ros/ros.h> (TEST ONLY): ros::NodeHandle::param<T>(name, default) over a string map — the only part of roscpp that
// select_registration_method (src/hdl_graph_slam/registrations.cpp:22-124) and LoopDetector's constructor (loop_detector.hpp:39-49) use.
#pragma once
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include "time.h"
namespace ros {
class NodeHandle {
public:
  std::map<std::string, std::string> params;  // test code fills this
  template <typename T>
  T param(const std::string& name, const T& default_value) const {
    auto it = params.find(name);
    if (it == params.end()) return default_value;
    return parse<T>(it->second);
  }

private:
  template <typename T>
  static T parse(const std::string& v) {
    std::istringstream is(v);
    T out{};
    is >> std::boolalpha >> out;
    return out;
  }
};
template <>
inline std::string NodeHandle::parse<std::string>(const std::string& v) {
  return v;
}
}  // namespace ros
// rosconsole's stream macros, to stderr (KeyFrame::load, src/hdl_graph_slam/keyframe.cpp:121-135, reports its failures with them)
#ifndef ROS_ERROR_STREAM
#define ROS_ERROR_STREAM(args) (std::cerr << "[ERROR] " << args << std::endl)
#define ROS_WARN_STREAM(args) (std::cerr << "[ WARN] " << args << std::endl)
#define ROS_INFO_STREAM(args) (std::cerr << "[ INFO] " << args << std::endl)
#endif
