// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp header). Parity unpinned by the reference.
//
// Exact k-nearest-neighbour search on a float point set: the role pcl::search::KdTree /
// pcl::KdTreeFLANN plays behind pcl::Registration::tree_ (getFitnessScore, getSearchMethodTarget;
// restated in-tree by src/hdl_graph_slam/information_matrix_calculator.cpp:49-80) and behind
// fast_gicp's source/target kd-trees (call site src/hdl_graph_slam/registrations.cpp:27-36).
//
// Contract shared with the HIP path so that index parity can be bit-exact:
//   d2(q,p) = fmaf(dz,dz, fmaf(dy,dy, dx*dx))   with dx = q.x - p.x … in float,
//   ties on d2 are resolved towards the LOWER point index.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <numeric>
#include <cfloat>

namespace hgso {

struct P3f {
  float x, y, z;
};

inline float dist2f(const P3f& q, const P3f& p) {
  const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
  return std::fmaf(dz, dz, std::fmaf(dy, dy, dx * dx));
}

struct Neighbor {
  float d2;
  int idx;
  bool operator<(const Neighbor& o) const { return d2 < o.d2 || (d2 == o.d2 && idx < o.idx); }
};

class KdTree {
public:
  static constexpr int kLeaf = 12;

  void build(const std::vector<P3f>& pts) {
    pts_ = &pts;
    const int n = (int)pts.size();
    order_.resize(n);
    std::iota(order_.begin(), order_.end(), 0);
    nodes_.clear();
    nodes_.reserve(n / 4 + 16);
    if (n > 0) build_rec(0, n);
  }

  int size() const { return pts_ ? (int)pts_->size() : 0; }

  // k nearest (ascending by (d2, idx)); returns how many were found (min(k, n)). Only points with
  // d2 <= bound2 are reported (pass FLT_MAX for unbounded).
  int knn(const P3f& q, int k, Neighbor* out, float bound2 = FLT_MAX) const {
    if (nodes_.empty() || k <= 0) return 0;
    Search s{q, k, 0, out, bound2};
    search_rec(0, s);
    return s.count;
  }

  // exact 1-NN; idx = -1 if the set is empty (or nothing within bound2).
  Neighbor nn(const P3f& q, float bound2 = FLT_MAX) const {
    Neighbor r{FLT_MAX, -1};
    knn(q, 1, &r, bound2);
    return r;
  }

  // number of points with d2 < r2 (strict), counting stops at `cap`
  int count_within(const P3f& q, float r2, int cap) const {
    if (nodes_.empty()) return 0;
    int cnt = 0;
    count_rec(0, q, r2, cap, cnt);
    return cnt;
  }

private:
  void count_rec(int id, const P3f& q, float r2, int cap, int& cnt) const {
    if (cnt >= cap) return;
    const Node& nd = nodes_[id];
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi && cnt < cap; i++)
        if (dist2f(q, (*pts_)[order_[i]]) < r2) cnt++;
      return;
    }
    const float qc = nd.dim == 0 ? q.x : (nd.dim == 1 ? q.y : q.z);
    const float diff = qc - nd.split;
    count_rec(diff < 0 ? nd.left : nd.right, q, r2, cap, cnt);
    if (diff * diff < r2) count_rec(diff < 0 ? nd.right : nd.left, q, r2, cap, cnt);
  }

  struct Node {
    int lo, hi;      // range in order_
    int left, right; // children (-1 for leaf)
    int dim;
    float split;
  };
  struct Search {
    P3f q;
    int k, count;
    Neighbor* best;  // sorted ascending, size count
    float bound2;
    float worst() const { return count < k ? bound2 : best[k - 1].d2; }
  };

  int build_rec(int lo, int hi) {
    const int id = (int)nodes_.size();
    nodes_.push_back({lo, hi, -1, -1, 0, 0.f});
    if (hi - lo <= kLeaf) return id;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = lo; i < hi; i++) {
      const P3f& p = (*pts_)[order_[i]];
      const float c[3] = {p.x, p.y, p.z};
      for (int d = 0; d < 3; d++) mn[d] = std::min(mn[d], c[d]), mx[d] = std::max(mx[d], c[d]);
    }
    int dim = 0;
    for (int d = 1; d < 3; d++)
      if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
    if (!(mx[dim] > mn[dim])) return id;  // all points coincide: keep as a (large) leaf
    const int mid = (lo + hi) / 2;
    auto coord = [&](int i) {
      const P3f& p = (*pts_)[i];
      return dim == 0 ? p.x : (dim == 1 ? p.y : p.z);
    };
    std::nth_element(order_.begin() + lo, order_.begin() + mid, order_.begin() + hi, [&](int a, int b) { return coord(a) < coord(b); });
    const float split = coord(order_[mid]);
    const int l = build_rec(lo, mid);
    const int r = build_rec(mid, hi);
    nodes_[id].left = l;
    nodes_[id].right = r;
    nodes_[id].dim = dim;
    nodes_[id].split = split;
    return id;
  }

  static void insert(Search& s, Neighbor c) {
    if (s.count < s.k) {
      int j = s.count++;
      while (j > 0 && c < s.best[j - 1]) s.best[j] = s.best[j - 1], j--;
      s.best[j] = c;
    } else if (c < s.best[s.k - 1]) {
      int j = s.k - 1;
      while (j > 0 && c < s.best[j - 1]) s.best[j] = s.best[j - 1], j--;
      s.best[j] = c;
    }
  }

  void search_rec(int id, Search& s) const {
    const Node& nd = nodes_[id];
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; i++) {
        const int pi = order_[i];
        const float d2 = dist2f(s.q, (*pts_)[pi]);
        if (d2 <= s.worst()) insert(s, {d2, pi});
      }
      return;
    }
    const float qc = nd.dim == 0 ? s.q.x : (nd.dim == 1 ? s.q.y : s.q.z);
    const float diff = qc - nd.split;
    const int near = diff < 0 ? nd.left : nd.right;
    const int far = diff < 0 ? nd.right : nd.left;
    search_rec(near, s);
    // float rounding is monotone, so diff*diff is a lower bound of every float d2 on the far side.
    if (diff * diff <= s.worst()) search_rec(far, s);
  }

  const std::vector<P3f>* pts_ = nullptr;
  std::vector<int> order_;
  std::vector<Node> nodes_;
};

}  // namespace hgso
