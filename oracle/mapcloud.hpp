// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp header). Parity unpinned by the reference.
//
// CPU restatement of MapCloudGenerator::generate (src/hdl_graph_slam/map_cloud_generator.cpp:13-51), "next" row f3:
//   1. every keyframe cloud is transformed by its pose cast to float (:23-30): dst = pose * (x, y, z, 1) as a float
//      4x4 times vector product, intensity copied; the clouds are concatenated in keyframe order;
//   2. resolution <= 0 returns that cloud (:36-37);
//   3. otherwise the cloud goes through pcl::octree::OctreePointCloud(resolution).addPointsFromInputCloud() and the
//      result is getOccupiedVoxelCenters() (:39-44).  PCL is not under /root/reference; its published behaviour
//      [UPSTREAM-KNOWLEDGE, PCL 1.10 octree_pointcloud.hpp]: the first finite point p0 centres the initial voxel
//      (min = p0 - resolution / 2, doubles), the bounding box only ever grows by whole multiples of the voxel size, so
//      the voxel lattice is  cell(p) = floor((p - min0) / resolution)  per axis with min0 = p0 - resolution / 2, and a
//      voxel centre is  (float)((cell + 0.5) * resolution + min0);  non-finite points are skipped; the returned points
//      carry x, y, z only (intensity 0).
//      PCL returns the centres in octree traversal order; here they come in ascending (z, y, x) cell order — the map
//      cloud is an unordered set for its consumers (visualisation, save_map_service).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <vector>
#include "prefilter.hpp"

namespace hgso {

// pose16: column-major float 4x4
inline PfPoint map_transform_point(const float* pose16, const PfPoint& p) {
  PfPoint o;
  o.x = ((pose16[0] * p.x + pose16[4] * p.y) + pose16[8] * p.z) + pose16[12];
  o.y = ((pose16[1] * p.x + pose16[5] * p.y) + pose16[9] * p.z) + pose16[13];
  o.z = ((pose16[2] * p.x + pose16[6] * p.y) + pose16[10] * p.z) + pose16[14];
  o.intensity = p.intensity;
  return o;
}

inline bool map_cloud_generate(const std::vector<std::vector<PfPoint>>& keyframes, const std::vector<std::array<float, 16>>& poses, double resolution,
                               std::vector<PfPoint>& out) {
  std::vector<PfPoint> cloud;
  for (size_t k = 0; k < keyframes.size(); k++)
    for (const PfPoint& p : keyframes[k]) cloud.push_back(map_transform_point(poses[k].data(), p));
  if (resolution <= 0.0) {
    out.swap(cloud);
    return true;
  }
  out.clear();
  size_t first = 0;
  while (first < cloud.size() && !pf_finite(cloud[first])) first++;
  if (first == cloud.size()) return true;
  const double min0[3] = {(double)cloud[first].x - resolution / 2, (double)cloud[first].y - resolution / 2, (double)cloud[first].z - resolution / 2};
  std::vector<std::array<long long, 3>> cells;
  cells.reserve(cloud.size());
  for (const PfPoint& p : cloud) {
    if (!pf_finite(p)) continue;
    cells.push_back({(long long)std::floor(((double)p.x - min0[0]) / resolution), (long long)std::floor(((double)p.y - min0[1]) / resolution),
                     (long long)std::floor(((double)p.z - min0[2]) / resolution)});
  }
  long long mn[3], mx[3];
  for (int a = 0; a < 3; a++) mn[a] = mx[a] = cells[0][a];
  for (auto& c : cells)
    for (int a = 0; a < 3; a++) mn[a] = std::min(mn[a], c[a]), mx[a] = std::max(mx[a], c[a]);
  const long long dx = mx[0] - mn[0] + 1, dy = mx[1] - mn[1] + 1, dz = mx[2] - mn[2] + 1;
  if (!((double)dx * (double)dy * (double)dz < 4.6e18)) return false;  // linear cell indices must fit 62 bits
  std::vector<long long> keys;
  keys.reserve(cells.size());
  for (auto& c : cells) keys.push_back((c[0] - mn[0]) + (c[1] - mn[1]) * dx + (c[2] - mn[2]) * dx * dy);
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  for (long long k : keys) {
    const long long cx = k % dx + mn[0], cy = (k / dx) % dy + mn[1], cz = k / (dx * dy) + mn[2];
    out.push_back({(float)(((double)cx + 0.5) * resolution + min0[0]), (float)(((double)cy + 0.5) * resolution + min0[1]),
                   (float)(((double)cz + 0.5) * resolution + min0[2]), 0.f});
  }
  return true;
}

}  // namespace hgso
