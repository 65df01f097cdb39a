// Minimal stand-in for <pcl/search/kdtree.h> (TEST ONLY): pcl::search::KdTree with the virtual surface of pcl::search::Search
// that pcl::Registration uses (setInputCloud, nearestKSearch, radiusSearch).  It is a REAL exact kd-tree (median splits, like
// FLANN's single-tree index in spirit) so that (a) getFitnessScore / nearestKSearch through the base pointer give exact answers
// and (b) a build costs what a build costs — the adapter test measures that the lazy tree skips it.  `builds()` counts them.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>
#include "../point_cloud.h"
namespace pcl {
namespace search {
template <typename PointT>
class KdTree {
public:
  using PointCloud = pcl::PointCloud<PointT>;
  using PointCloudConstPtr = typename PointCloud::ConstPtr;
  using IndicesConstPtr = pcl::IndicesConstPtr;
  using Ptr = std::shared_ptr<KdTree<PointT>>;
  using ConstPtr = std::shared_ptr<const KdTree<PointT>>;
  virtual ~KdTree() = default;
#ifdef HGS_MOCK_PCL_1_12
  virtual bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) {
    build_index(cloud, indices);
    return true;
  }
#else
  virtual void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) { build_index(cloud, indices); }
#endif
  void build_index(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices) {
    input_ = cloud;
    indices_ = indices;
    order_.resize(cloud ? cloud->size() : 0);
    std::iota(order_.begin(), order_.end(), 0);
    // non-finite points never enter the index (FLANN is given finite points only)
    order_.erase(std::remove_if(order_.begin(), order_.end(), [&](int i) { const PointT& p = cloud->points[i]; return !(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z)); }), order_.end());
    split_.assign(order_.size(), 0);
    if (!order_.empty()) build(0, (int)order_.size());
    builds_counter()++;
  }
  PointCloudConstPtr getInputCloud() const { return input_; }
  virtual int nearestKSearch(const PointT& q, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const {
    std::vector<std::pair<float, int>> heap;  // max-heap of the k best (d2, index)
    if (!order_.empty() && k > 0) knn(0, (int)order_.size(), q, k, heap);
    std::sort_heap(heap.begin(), heap.end());
    k_indices.resize(heap.size());
    k_sqr_distances.resize(heap.size());
    for (size_t i = 0; i < heap.size(); i++) k_indices[i] = heap[i].second, k_sqr_distances[i] = heap[i].first;
    return (int)heap.size();
  }
  virtual int radiusSearch(const PointT& q, double radius, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const {
    std::vector<std::pair<float, int>> found;
    if (!order_.empty()) radius_rec(0, (int)order_.size(), q, (float)(radius * radius), found);
    std::sort(found.begin(), found.end());
    if (max_nn > 0 && found.size() > max_nn) found.resize(max_nn);
    k_indices.resize(found.size());
    k_sqr_distances.resize(found.size());
    for (size_t i = 0; i < found.size(); i++) k_indices[i] = found[i].second, k_sqr_distances[i] = found[i].first;
    return (int)found.size();
  }
  // number of index builds in this process (all instances): what the lazy-tree test counts
  static std::atomic<long>& builds_counter() {
    static std::atomic<long> n{0};
    return n;
  }

protected:
  PointCloudConstPtr input_;
  IndicesConstPtr indices_;

private:
  static float coord(const PointT& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }
  static float dist2(const PointT& a, const PointT& b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
  }
  void build(int lo, int hi) {  // node = the median element of [lo, hi) along the widest axis
    if (hi - lo <= 1) return;
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = lo; i < hi; i++)
      for (int a = 0; a < 3; a++) {
        const float v = coord(input_->points[order_[i]], a);
        mn[a] = std::min(mn[a], v), mx[a] = std::max(mx[a], v);
      }
    int axis = 0;
    for (int a = 1; a < 3; a++)
      if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
    const int mid = (lo + hi) / 2;
    std::nth_element(order_.begin() + lo, order_.begin() + mid, order_.begin() + hi,
                     [&](int i, int j) { return coord(input_->points[i], axis) < coord(input_->points[j], axis); });
    split_[mid] = (unsigned char)axis;
    build(lo, mid);
    build(mid + 1, hi);
  }
  void knn(int lo, int hi, const PointT& q, int k, std::vector<std::pair<float, int>>& heap) const {
    if (hi <= lo) return;
    const int mid = (lo + hi) / 2;
    const PointT& p = input_->points[order_[mid]];
    const float d2 = dist2(p, q);
    if ((int)heap.size() < k) {
      heap.emplace_back(d2, order_[mid]);
      std::push_heap(heap.begin(), heap.end());
    } else if (std::make_pair(d2, order_[mid]) < heap.front()) {
      std::pop_heap(heap.begin(), heap.end());
      heap.back() = std::make_pair(d2, order_[mid]);
      std::push_heap(heap.begin(), heap.end());
    }
    if (hi - lo == 1) return;
    const int axis = split_[mid];
    const float diff = coord(q, axis) - coord(p, axis);
    const bool left_first = diff < 0;
    if (left_first) knn(lo, mid, q, k, heap);
    else knn(mid + 1, hi, q, k, heap);
    if ((int)heap.size() < k || diff * diff <= heap.front().first) {
      if (left_first) knn(mid + 1, hi, q, k, heap);
      else knn(lo, mid, q, k, heap);
    }
  }
  void radius_rec(int lo, int hi, const PointT& q, float r2, std::vector<std::pair<float, int>>& out) const {
    if (hi <= lo) return;
    const int mid = (lo + hi) / 2;
    const PointT& p = input_->points[order_[mid]];
    const float d2 = dist2(p, q);
    if (d2 <= r2) out.emplace_back(d2, order_[mid]);
    if (hi - lo == 1) return;
    const int axis = split_[mid];
    const float diff = coord(q, axis) - coord(p, axis);
    if (diff < 0 || diff * diff <= r2) radius_rec(lo, mid, q, r2, out);
    if (diff >= 0 || diff * diff <= r2) radius_rec(mid + 1, hi, q, r2, out);
  }
  std::vector<int> order_;
  std::vector<unsigned char> split_;
};
}  // namespace search
}  // namespace pcl
