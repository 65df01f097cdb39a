// Minimal stand-in for <sensor_msgs/PointCloud2.h> (TEST ONLY): only named as the advertised message type.
#pragma once
namespace sensor_msgs {
struct PointCloud2 {};
}  // namespace sensor_msgs
