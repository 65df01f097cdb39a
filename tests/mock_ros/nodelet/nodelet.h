// Minimal stand-in for <nodelet/nodelet.h> (TEST ONLY): a nodelet gets its two node handles from the base class and is started with onInit().
#pragma once
#include <iostream>
#include <ros/ros.h>
#define NODELET_DEBUG(...) ((void)0)
#define NODELET_INFO_STREAM(args) (std::cerr << "[ INFO] " << args << std::endl)
#define NODELET_WARN_STREAM(args) (std::cerr << "[ WARN] " << args << std::endl)
namespace nodelet {
class Nodelet {
public:
  virtual ~Nodelet() = default;
  virtual void onInit() = 0;
  ros::NodeHandle mock_nh, mock_private_nh;  // the test fills mock_private_nh.params before onInit()

protected:
  ros::NodeHandle& getNodeHandle() { return mock_nh; }
  ros::NodeHandle& getPrivateNodeHandle() { return mock_private_nh; }
};
}  // namespace nodelet
