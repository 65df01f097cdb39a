// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp header). Parity unpinned by the reference.
//
// CPU restatement of MapCloudGenerator::generate (src/hdl_graph_slam/map_cloud_generator.cpp:13-51), "next" row f3:
//   1. every keyframe cloud is transformed by its pose cast to float (:23-30): dst = pose * (x, y, z, 1) as a float
//      4x4 times vector product, intensity copied; the clouds are concatenated in keyframe order;
//   2. resolution <= 0 returns that cloud (:36-37);
//   3. otherwise the cloud goes through pcl::octree::OctreePointCloud(resolution).addPointsFromInputCloud() and the
//      result is getOccupiedVoxelCenters() (:39-44).  PCL is not under /root/reference; its published behaviour
//      [UPSTREAM-KNOWLEDGE, PCL 1.10 octree_pointcloud.hpp / octree_base.hpp], restated step by step:
//      * finite points are inserted in input order (addPointIdx: adoptBoundingBoxToPoint, genOctreeKeyforPoint, createLeaf);
//      * the FIRST point defines the box: min = p - resolution / 2, max = p + resolution / 2 (doubles), then getKeyBitSize() on
//        the still empty tree: at least 2 voxels per axis => depth 1, and the box is widened symmetrically to the side length
//        2 * resolution — the first point ends up ON the voxel boundary in the middle of a 2 x 2 x 2 root;
//      * a later point outside [min, max) doubles the box until it fits: per axis the box grows towards LOWER values unless
//        the point violates the UPPER bound there (min -= side), depth += 1, max = min + 2^depth * resolution - FLT_EPSILON;
//        the old root becomes child ((!upper_x) << 2 | (!upper_y) << 1 | !upper_z) of the new one, i.e. the keys of
//        everything inserted before gain 2^old_depth on the axes that grew downwards;
//      * key(p) = (unsigned)((p - min) / resolution) per axis with the min of the moment of insertion;
//      * getOccupiedVoxelCenters walks the tree depth-first, children 0..7, child = (x_bit << 2 | y_bit << 1 | z_bit) from the
//        most significant key bit down: the leaves come out in ascending order of the bit-interleaved key (x most
//        significant); a centre is (float)((key + 0.5) * resolution + min) with the FINAL min; x, y, z only (intensity 0).
//      The box depends on the insertion order (which point forced which doubling), so the restatement replays it.
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

// The bounding box of pcl::octree::OctreePointCloud while points are added (adoptBoundingBoxToPoint / getKeyBitSize).
struct OctreeBox {
  double mn[3], mx[3];
  int depth;
};
constexpr double kOctreeMinValue = 1.1920928955078125e-07;  // (double)std::numeric_limits<float>::epsilon()

inline OctreeBox octree_box_first(const PfPoint& p, double res) {
  OctreeBox b;
  const float q[3] = {p.x, p.y, p.z};
  for (int a = 0; a < 3; a++) b.mn[a] = (double)q[a] - res / 2, b.mx[a] = (double)q[a] + res / 2;
  // getKeyBitSize() with leaf_count_ == 0
  unsigned max_key[3];
  for (int a = 0; a < 3; a++) max_key[a] = (unsigned)std::ceil((b.mx[a] - b.mn[a] - kOctreeMinValue) / res);
  const unsigned max_voxels = std::max(std::max(std::max(max_key[0], max_key[1]), max_key[2]), 2u);
  b.depth = (int)std::max(std::min(32u, (unsigned)std::ceil(std::log2((double)max_voxels) - kOctreeMinValue)), 0u);
  const double side = (double)(1 << b.depth) * res;
  for (int a = 0; a < 3; a++) {
    const double oversize = (side - (b.mx[a] - b.mn[a])) / 2.0;
    if (oversize > kOctreeMinValue) b.mn[a] -= oversize, b.mx[a] += oversize;
  }
  return b;
}
inline bool octree_box_violated(const OctreeBox& b, const PfPoint& p) {
  const float q[3] = {p.x, p.y, p.z};
  for (int a = 0; a < 3; a++)
    if ((double)q[a] < b.mn[a] || (double)q[a] >= b.mx[a]) return true;
  return false;
}
// one doubling towards p; grew[a] = 1 where min moved (the keys inserted so far gain 2^old_depth on that axis)
inline void octree_box_double(OctreeBox& b, const PfPoint& p, double res, int* grew) {
  const float q[3] = {p.x, p.y, p.z};
  double side = (double)(1 << b.depth) * res;
  for (int a = 0; a < 3; a++) {
    const bool upper = (double)q[a] >= b.mx[a];
    grew[a] = upper ? 0 : 1;
    if (!upper) b.mn[a] -= side;
  }
  b.depth++;
  side = (double)(1 << b.depth) * res - kOctreeMinValue;
  for (int a = 0; a < 3; a++) b.mx[a] = b.mn[a] + side;
}
constexpr int kOctreeMaxDepth = 21;  // 3 * 21 interleaved key bits fit 63: 2^21 voxels per axis (0.01 m over 20 km)

inline unsigned long long octree_interleave(const unsigned long long* key, int depth) {
  unsigned long long m = 0;
  for (int bit = depth - 1; bit >= 0; bit--) m = (m << 3) | (((key[0] >> bit) & 1ull) << 2) | (((key[1] >> bit) & 1ull) << 1) | ((key[2] >> bit) & 1ull);
  return m;
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
  OctreeBox box{};
  bool defined = false;
  // keys as inserted + the doublings that had happened by then; offsets[e] = what the keys have gained after e doublings
  std::vector<std::array<unsigned long long, 3>> keys;
  std::vector<int> doublings_at_insert;
  std::vector<std::array<unsigned long long, 3>> gained(1, {0ull, 0ull, 0ull});
  for (const PfPoint& p : cloud) {
    if (!pf_finite(p)) continue;
    if (!defined) box = octree_box_first(p, resolution), defined = true;
    while (octree_box_violated(box, p)) {
      if (box.depth >= kOctreeMaxDepth) return false;
      int grew[3];
      const unsigned long long step = 1ull << box.depth;
      octree_box_double(box, p, resolution, grew);
      auto g = gained.back();
      for (int a = 0; a < 3; a++) g[a] += grew[a] ? step : 0ull;
      gained.push_back(g);
    }
    keys.push_back({(unsigned long long)(unsigned)(((double)p.x - box.mn[0]) / resolution), (unsigned long long)(unsigned)(((double)p.y - box.mn[1]) / resolution),
                    (unsigned long long)(unsigned)(((double)p.z - box.mn[2]) / resolution)});
    doublings_at_insert.push_back((int)gained.size() - 1);
  }
  if (!defined) return true;
  std::vector<unsigned long long> codes;
  codes.reserve(keys.size());
  for (size_t i = 0; i < keys.size(); i++) {
    unsigned long long k[3];
    for (int a = 0; a < 3; a++) k[a] = keys[i][a] + (gained.back()[a] - gained[doublings_at_insert[i]][a]);
    codes.push_back(octree_interleave(k, box.depth));
  }
  std::sort(codes.begin(), codes.end());
  codes.erase(std::unique(codes.begin(), codes.end()), codes.end());
  for (unsigned long long m : codes) {
    unsigned long long k[3] = {0, 0, 0};
    for (int bit = 0; bit < box.depth; bit++) {
      const unsigned long long t = (m >> (3 * bit)) & 7ull;
      k[0] |= ((t >> 2) & 1ull) << bit, k[1] |= ((t >> 1) & 1ull) << bit, k[2] |= (t & 1ull) << bit;
    }
    out.push_back({(float)(((double)k[0] + 0.5) * resolution + box.mn[0]), (float)(((double)k[1] + 0.5) * resolution + box.mn[1]),
                   (float)(((double)k[2] + 0.5) * resolution + box.mn[2]), 0.f});
  }
  return true;
}

}  // namespace hgso
