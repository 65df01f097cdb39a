// Minimal stand-in for <pcl/point_types.h> (TEST ONLY): the memory layout of pcl::PointXYZI (32 bytes).
#pragma once
namespace pcl {
struct alignas(16) PointXYZI {
  float x, y, z, data3;
  float intensity, pad[3];
};
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is 32 bytes");
}  // namespace pcl
