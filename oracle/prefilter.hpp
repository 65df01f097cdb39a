// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp header). Parity unpinned by the reference.
//
// CPU restatement of the prefilter that feeds the scan-matching path, apps/prefiltering_nodelet.cpp:
//   deskewing        (:182-243)  optional, before everything else: see pf_deskew below
//   distance_filter  (:165-182)  keep p iff near < |p| < far, |p| the FLOAT norm compared against double thresholds
//   downsample       (:138-149)  pcl::VoxelGrid with leaf = downsample_resolution (:52-58)
//   outlier_removal  (:151-163)  pcl::RadiusOutlierRemoval (:85-93) or pcl::StatisticalOutlierRemoval (:73-84)
// PCL is not under /root/reference; the three filters follow SURVEY.md "next row f2" and PCL 1.10's published
// behaviour [UPSTREAM-KNOWLEDGE]:
//   VoxelGrid: inverse leaf = 1.0f / leaf (float); min/max over the finite points; min_b = floor(min * inv),
//     idx = (floor(x*inv) - min_b.x) + (floor(y*inv) - min_b.y) * div.x + (floor(z*inv) - min_b.z) * div.x*div.y;
//     points sorted by idx (PCL: std::sort, order inside a voxel unspecified — here: input order, a stable sort);
//     output = per-voxel centroid of x, y, z and intensity accumulated in FLOAT (CentroidPoint) and divided by the
//     count, voxels in ascending idx order.
//   RadiusOutlierRemoval: keep p iff the radius search around p (which finds p itself) returns MORE than
//     min_neighbors points, i.e. at least min_neighbors other points with d2 < r2 (float d2, FLANN's strict compare).
//   StatisticalOutlierRemoval: d_i = mean distance (sqrt of float d2, accumulated in double) to the mean_k nearest
//     OTHER points (k+1 search, self dropped); mean and sample stddev over all d_i in double;
//     keep p iff d_i <= mean + stddev_mul * stddev.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>
#include "gicp.hpp"  // OCloud (finite points + kd-tree)

namespace hgso {

struct PfPoint {
  float x, y, z, intensity;
};

struct PrefilterParams {  // mirrors hgs_prefilter_params of include/hgs_registration.h
  int32_t use_distance_filter;
  int32_t downsample_method;       // 0 NONE, 1 VOXELGRID
  double distance_near_thresh, distance_far_thresh;
  double downsample_resolution;
  int32_t outlier_removal_method;  // 0 NONE, 1 STATISTICAL, 2 RADIUS
  int32_t statistical_mean_k;
  double statistical_stddev;
  double radius_radius;
  int32_t radius_min_neighbors;
  int32_t reserved;
};

// deskewing (:182-243): every point i of the sweep is rotated back by the rotation the sensor made during
// delta_t = scan_period * i / size, from ONE gyro sample: delta_q = (1, delta_t/2 * w) with w = -(imu angular velocity)
// as floats — a first-order, NOT normalised quaternion —, p' = delta_q.inverse() * p.  Eigen is not under /root/reference;
// [UPSTREAM-KNOWLEDGE] Eigen 3.3: inverse() = conjugate().coeffs() / squaredNorm() (float, the SSE2 reduction order
// (x^2 + z^2) + (y^2 + w^2)), and q * v = v + w * uv + q.vec().cross(uv) with uv = 2 * q.vec().cross(v), all in float,
// no fused multiply-add.  Non-finite points stay non-finite; intensity is kept.
inline void pf_deskew(std::vector<PfPoint>& pts, const double imu_angular_velocity[3], double scan_period) {
  const float w[3] = {-(float)imu_angular_velocity[0], -(float)imu_angular_velocity[1], -(float)imu_angular_velocity[2]};
  const size_t n = pts.size();
  for (size_t i = 0; i < n; i++) {
    const double delta_t = scan_period * (double)i / (double)n;
    const float qw = 1.f, qx = (float)(delta_t / 2.0 * (double)w[0]), qy = (float)(delta_t / 2.0 * (double)w[1]), qz = (float)(delta_t / 2.0 * (double)w[2]);
    const float n2 = (qx * qx + qz * qz) + (qy * qy + qw * qw);
    float ix = 0.f, iy = 0.f, iz = 0.f, iw = 0.f;
    if (n2 > 0.f) ix = -qx / n2, iy = -qy / n2, iz = -qz / n2, iw = qw / n2;
    const float vx = pts[i].x, vy = pts[i].y, vz = pts[i].z;
    float ux = iy * vz - iz * vy, uy = iz * vx - ix * vz, uz = ix * vy - iy * vx;
    ux += ux, uy += uy, uz += uz;
    pts[i].x = (vx + iw * ux) + (iy * uz - iz * uy);
    pts[i].y = (vy + iw * uy) + (iz * ux - ix * uz);
    pts[i].z = (vz + iw * uz) + (ix * uy - iy * ux);
  }
}

inline std::vector<PfPoint> pf_distance_filter(const std::vector<PfPoint>& in, double near, double far) {
  std::vector<PfPoint> out;
  out.reserve(in.size());
  for (const PfPoint& p : in) {
    const float n2 = p.x * p.x + p.y * p.y + p.z * p.z;  // Eigen's squaredNorm in float, then sqrt
    const double d = (double)std::sqrt(n2);
    if (d > near && d < far) out.push_back(p);
  }
  return out;
}

inline bool pf_finite(const PfPoint& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); }

// returns false if the voxel grid would overflow int indices (PCL warns and leaves the cloud untouched)
inline bool pf_voxelgrid(const std::vector<PfPoint>& in, double leaf, std::vector<PfPoint>& out) {
  out.clear();
  const float inv = 1.0f / (float)leaf;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  bool any = false;
  for (const PfPoint& p : in)
    if (pf_finite(p)) {
      any = true;
      mn[0] = std::min(mn[0], p.x), mn[1] = std::min(mn[1], p.y), mn[2] = std::min(mn[2], p.z);
      mx[0] = std::max(mx[0], p.x), mx[1] = std::max(mx[1], p.y), mx[2] = std::max(mx[2], p.z);
    }
  if (!any) return true;
  int min_b[3];
  long long div[3];
  for (int k = 0; k < 3; k++) {
    min_b[k] = (int)std::floor(mn[k] * inv);
    div[k] = (long long)(int)std::floor(mx[k] * inv) - min_b[k] + 1;
  }
  if (div[0] * div[1] * div[2] > 2147483647LL) return false;
  struct KV {
    uint32_t key;
    uint32_t idx;
  };
  std::vector<KV> kv;
  kv.reserve(in.size());
  for (size_t i = 0; i < in.size(); i++) {
    const PfPoint& p = in[i];
    if (!pf_finite(p)) continue;
    const int ix = (int)std::floor(p.x * inv) - min_b[0], iy = (int)std::floor(p.y * inv) - min_b[1], iz = (int)std::floor(p.z * inv) - min_b[2];
    kv.push_back({(uint32_t)(ix + iy * (int)div[0] + iz * (int)(div[0] * div[1])), (uint32_t)i});
  }
  std::stable_sort(kv.begin(), kv.end(), [](const KV& a, const KV& b) { return a.key < b.key; });
  for (size_t i = 0; i < kv.size();) {
    size_t j = i;
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    for (; j < kv.size() && kv[j].key == kv[i].key; j++) {
      const PfPoint& p = in[kv[j].idx];
      sx += p.x, sy += p.y, sz += p.z, si += p.intensity;
    }
    const float n = (float)(j - i);
    out.push_back({sx / n, sy / n, sz / n, si / n});
    i = j;
  }
  return true;
}

// pcl::ApproximateVoxelGrid<PointT>::applyFilter (apps/prefiltering_nodelet.cpp:59-63, apps/scan_matching_odometry_nodelet.cpp:91-96)
// [UPSTREAM-KNOWLEDGE, PCL 1.8-1.12 approximate_voxel_grid.hpp]: a 512-entry history table indexed by
//   hash = (ix * 7171 + iy * 3079 + iz * 4231) & 511,  (ix, iy, iz) = floor(p * inverse_leaf)  (float)
// is walked over the points IN INPUT ORDER: a point whose bucket holds another voxel first flushes that bucket's centroid to the
// output (sum of the fields in float / count) and takes the bucket over; at the end the occupied buckets are flushed in bucket
// order.  The same voxel can therefore appear several times in the output, and the output order is the order of the evictions.
// downsample_all_data_ defaults to true: the centroid covers x, y, z and intensity.  Non-finite points are skipped here (PCL
// converts floor(NaN) to int, which is undefined).
inline void pf_approx_voxelgrid(const std::vector<PfPoint>& in, double leaf, std::vector<PfPoint>& out) {
  out.clear();
  const float inv = 1.0f / (float)leaf;
  struct Entry {
    int ix = 0, iy = 0, iz = 0, count = 0;
    float sx = 0, sy = 0, sz = 0, si = 0;
  };
  std::vector<Entry> hist(512);
  auto flush = [&](Entry& e) {
    const float n = (float)e.count;
    out.push_back({e.sx / n, e.sy / n, e.sz / n, e.si / n});
    e.count = 0, e.sx = e.sy = e.sz = e.si = 0.f;
  };
  for (const PfPoint& p : in) {
    if (!pf_finite(p)) continue;
    const int ix = (int)std::floor(p.x * inv), iy = (int)std::floor(p.y * inv), iz = (int)std::floor(p.z * inv);
    const unsigned hash = ((unsigned)ix * 7171u + (unsigned)iy * 3079u + (unsigned)iz * 4231u) & 511u;  // two's-complement wrap of PCL's int arithmetic
    Entry& e = hist[hash];
    if (e.count && (ix != e.ix || iy != e.iy || iz != e.iz)) flush(e);
    e.ix = ix, e.iy = iy, e.iz = iz;
    e.count++;
    e.sx += p.x, e.sy += p.y, e.sz += p.z, e.si += p.intensity;
  }
  for (Entry& e : hist)
    if (e.count) flush(e);
}

inline OCloud pf_tree_of(const std::vector<PfPoint>& in) {
  OCloud c;
  c.assign(in.data(), in.size(), sizeof(PfPoint));
  return c;
}

inline std::vector<PfPoint> pf_radius_outlier_removal(const std::vector<PfPoint>& in, double radius, int min_neighbors) {
  OCloud c = pf_tree_of(in);
  const float r2 = (float)(radius * radius);
  std::vector<char> keep(in.size(), 0);
#pragma omp parallel for schedule(guided, 8)
  for (long i = 0; i < (long)c.pts.size(); i++) {
    // points with d2 < r2, the point itself included; more than min_neighbors of them keeps the point
    if (c.tree.count_within(c.pts[i], r2, min_neighbors + 1) > min_neighbors) keep[c.orig[i]] = 1;
  }
  std::vector<PfPoint> out;
  for (size_t i = 0; i < in.size(); i++)
    if (keep[i]) out.push_back(in[i]);
  return out;
}

inline std::vector<PfPoint> pf_statistical_outlier_removal(const std::vector<PfPoint>& in, int mean_k, double stddev_mul) {
  OCloud c = pf_tree_of(in);
  const size_t n = c.pts.size();
  std::vector<double> dist(n, 0.0);
#pragma omp parallel for schedule(guided, 8)
  for (long i = 0; i < (long)n; i++) {
    std::vector<Neighbor> nb(mean_k + 1);
    const int found = c.tree.knn(c.pts[i], mean_k + 1, nb.data());
    double s = 0;
    for (int j = 1; j < found; j++) s += std::sqrt((double)nb[j].d2);  // nb[0] is the point itself
    dist[i] = found > 1 ? s / (double)mean_k : 0.0;
  }
  double sum = 0, sq = 0;
  for (double d : dist) sum += d, sq += d * d;
  const double mean = n ? sum / (double)n : 0.0;
  const double var = n > 1 ? (sq - sum * sum / (double)n) / ((double)n - 1.0) : 0.0;
  const double thr = mean + stddev_mul * std::sqrt(var);
  std::vector<char> keep(in.size(), 0);
  for (size_t i = 0; i < n; i++)
    if (dist[i] <= thr) keep[c.orig[i]] = 1;
  std::vector<PfPoint> out;
  for (size_t i = 0; i < in.size(); i++)
    if (keep[i]) out.push_back(in[i]);
  return out;
}

inline bool prefilter(const std::vector<PfPoint>& in, const PrefilterParams& p, std::vector<PfPoint>& out) {
  std::vector<PfPoint> cur = p.use_distance_filter ? pf_distance_filter(in, p.distance_near_thresh, p.distance_far_thresh) : in;
  if (p.downsample_method == 1) {
    std::vector<PfPoint> ds;
    if (!pf_voxelgrid(cur, p.downsample_resolution, ds)) return false;
    cur.swap(ds);
  } else if (p.downsample_method == 2) {
    std::vector<PfPoint> ds;
    pf_approx_voxelgrid(cur, p.downsample_resolution, ds);
    cur.swap(ds);
  }
  if (p.outlier_removal_method == 1) cur = pf_statistical_outlier_removal(cur, p.statistical_mean_k, p.statistical_stddev);
  else if (p.outlier_removal_method == 2) cur = pf_radius_outlier_removal(cur, p.radius_radius, p.radius_min_neighbors);
  out.swap(cur);
  return true;
}

}  // namespace hgso
