// Minimal stand-in for <ros/ros.h> (TEST ONLY): ros::NodeHandle::param<T>(name, default) over a string map — the only part of roscpp that
// select_registration_method (src/hdl_graph_slam/registrations.cpp:22-124) and LoopDetector's constructor (loop_detector.hpp:39-49) use.
#pragma once
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <type_traits>
#include <vector>
#include <sstream>
#include <string>
#include <boost/bind.hpp>
#include "time.h"
namespace ros {
// topics of the stand-in graph: what a nodelet subscribed to (type-erased callbacks the test invokes) and what it published (copies the test reads back)
namespace mock {
inline std::map<std::string, std::function<void(const void*)>>& callbacks() {
  static std::map<std::string, std::function<void(const void*)>> m;
  return m;
}
inline std::map<std::string, std::vector<std::shared_ptr<void>>>& published() {
  static std::map<std::string, std::vector<std::shared_ptr<void>>> m;
  return m;
}
inline std::map<std::string, unsigned>& subscribers() {  // what Publisher::getNumSubscribers() reports per topic (default 0)
  static std::map<std::string, unsigned> m;
  return m;
}
}  // namespace mock
class Subscriber {};
class Publisher {
public:
  std::string topic;
  template <typename M>
  void publish(const M& msg) const {
    mock::published()[topic].push_back(std::make_shared<M>(msg));
  }
  unsigned getNumSubscribers() const {
    auto it = mock::subscribers().find(topic);
    return it == mock::subscribers().end() ? 0u : it->second;
  }
};
class NodeHandle {
public:
  std::map<std::string, std::string> params;  // test code fills this
  // nh.subscribe(topic, queue, &Class::callback, this): the callback lands in mock::callbacks()[topic]
  template <typename M, typename T>
  Subscriber subscribe(const std::string& topic, unsigned, void (T::*fp)(M), T* obj) {
    using Arg = typename std::remove_cv<typename std::remove_reference<M>::type>::type;
    mock::callbacks()[topic] = [obj, fp](const void* msg) { (obj->*fp)(*static_cast<const Arg*>(msg)); };
    return Subscriber();
  }
  // nh.subscribe<M>(topic, queue, boost::bind(&Class::callback, this, _1, ...)): the callback takes a shared_ptr<const M>
  template <typename M, typename F>
  Subscriber subscribe(const std::string& topic, unsigned, F f) {
    std::function<void(const std::shared_ptr<const M>&)> fn = f;
    mock::callbacks()[topic] = [fn](const void* msg) { fn(*static_cast<const std::shared_ptr<const M>*>(msg)); };
    return Subscriber();
  }
  template <typename M>
  Publisher advertise(const std::string& topic, unsigned) {
    Publisher p;
    p.topic = topic;
    return p;
  }
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
inline bool ok() { return true; }
}  // namespace ros
// rosconsole's stream macros, to stderr (KeyFrame::load, src/hdl_graph_slam/keyframe.cpp:121-135, reports its failures with them)
#ifndef ROS_ERROR_STREAM
#define ROS_ERROR_STREAM(args) (std::cerr << "[ERROR] " << args << std::endl)
#define ROS_WARN_STREAM(args) (std::cerr << "[ WARN] " << args << std::endl)
#define ROS_INFO_STREAM(args) (std::cerr << "[ INFO] " << args << std::endl)
#endif
