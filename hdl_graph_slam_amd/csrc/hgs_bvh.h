// hgs_bvh.h — exact nearest-neighbour search on a Hilbert-sorted point array through an implicit, perfectly
// balanced bounding-interval tree (heap layout, no pointers, stack-free traversal).
//
// This replaces the kd-trees the reference path relies on: pcl::Registration::tree_ (getFitnessScore /
// getSearchMethodTarget, restated in src/hdl_graph_slam/information_matrix_calculator.cpp:49-80) and fast_gicp's
// source/target trees (call site src/hdl_graph_slam/registrations.cpp:27-36).  Semantics are those of an exact
// kd-tree query: true nearest under d2 = fma(dz,dz,fma(dy,dy,dx*dx)) in float, ties towards the lower ORIGINAL index.
//
// Layout in HBM (per cloud):
//   pts   : float4[P*LEAF]   sorted along a 48-bit Hilbert curve; .w = original index (int bits); padding = +inf
//   nodes : float4[4*P]      node n (1 <= n < 2P) has AABB {nodes[2n] = min.xyz, nodes[2n+1] = max.xyz};
//                            children of n are 2n, 2n+1, so both child boxes are one aligned 64-byte read at 4n.
//                            Leaves are nodes P..2P-1; leaf l owns pts[l*LEAF .. l*LEAF+LEAF).  Empty nodes have
//                            min = +inf, max = -inf (box distance = +inf, never visited).
#pragma once
#include "hgs_math.h"

namespace hgs {

constexpr int kLeaf = 8;

#if defined(__HIPCC__)
typedef float4 Float4;
#else
struct alignas(16) Float4 {
  float x, y, z, w;
};
#endif

struct BvhView {
  const Float4* nodes;
  const Float4* pts;
  int P;       // number of leaves (power of two)
  int n;       // number of valid (finite) points
};

HGS_HD int float_as_int_hd(float f) {
  union {
    float f;
    int i;
  } u;
  u.f = f;
  return u.i;
}
HGS_HD float int_as_float_hd(int i) {
  union {
    float f;
    int i;
  } u;
  u.i = i;
  return u.f;
}

HGS_HD int highest_bit(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return 31 - __clz((int)v);
#else
  return 31 - __builtin_clz(v);
#endif
}

// Exact 1-NN. Returns the position in the sorted array (or -1) and its squared distance; only points with
// d2 <= bound2 qualify.  orig_idx receives the original index of the hit.
HGS_HD int bvh_nn1(const BvhView& t, const F3& q, float bound2, float* out_d2, int* orig_idx) {
  float best = bound2;
  int best_pos = -1, best_orig = 0x7fffffff;
  if (t.n <= 0) {
    *out_d2 = best;
    *orig_idx = -1;
    return -1;
  }
  uint32_t node = 1, pending = 0;
  int depth = 0;
  for (;;) {
    bool pruned = false;
    while ((int)node < t.P) {
      const Float4 a0 = t.nodes[4 * node + 0], a1 = t.nodes[4 * node + 1];
      const Float4 b0 = t.nodes[4 * node + 2], b1 = t.nodes[4 * node + 3];
      const float d0 = box_dist2f(q, a0.x, a0.y, a0.z, a1.x, a1.y, a1.z);
      const float d1 = box_dist2f(q, b0.x, b0.y, b0.z, b1.x, b1.y, b1.z);
      const int near = d1 < d0 ? 1 : 0;
      const float dn = near ? d1 : d0, df = near ? d0 : d1;
      if (!(dn <= best)) {
        pruned = true;
        break;
      }
      depth++;
      if (df <= best) pending |= 1u << depth;
      node = 2 * node + near;
    }
    if (!pruned) {
      const int base = ((int)node - t.P) * kLeaf;
#pragma unroll
      for (int l = 0; l < kLeaf; l++) {
        const Float4 p = t.pts[base + l];
        const float d = dist2f(q, p.x, p.y, p.z);
        const int oi = float_as_int_hd(p.w);
        if (d < best || (d == best && oi < best_orig && d <= bound2)) {
          best = d;
          best_pos = base + l;
          best_orig = oi;
        }
      }
    }
    // backtrack to the deepest pending sibling that can still matter
    for (;;) {
      if (!pending) {
        *out_d2 = best;
        *orig_idx = best_pos >= 0 ? best_orig : -1;
        return best_pos;
      }
      const int lvl = highest_bit(pending);
      pending ^= 1u << lvl;
      node = (node >> (depth - lvl)) ^ 1u;
      depth = lvl;
      const Float4 m0 = t.nodes[2 * node], m1 = t.nodes[2 * node + 1];
      if (box_dist2f(q, m0.x, m0.y, m0.z, m1.x, m1.y, m1.z) <= best) break;
    }
  }
}

// Exact k-NN with a register-resident sorted list of KMAX slots; k <= KMAX real slots are used, the others are
// pre-filled with -1 so that every index is a compile-time constant (no scratch memory).
template <int KMAX>
struct KnnList {
  float d[KMAX];
  int pos[KMAX];
  HGS_HD void init(int k) {
#pragma unroll
    for (int i = 0; i < KMAX; i++) {
      d[i] = (i < KMAX - k) ? -1.f : FLT_MAX;
      pos[i] = -1;
    }
  }
  HGS_HD float worst() const { return d[KMAX - 1]; }
  HGS_HD void insert(float dist, int p) {
    // caller guarantees dist < worst()
    d[KMAX - 1] = dist;
    pos[KMAX - 1] = p;
#pragma unroll
    for (int i = KMAX - 1; i > 0; i--) {
      const bool sw = d[i] < d[i - 1];
      const float dl = sw ? d[i] : d[i - 1], dh = sw ? d[i - 1] : d[i];
      const int pl = sw ? pos[i] : pos[i - 1], ph = sw ? pos[i - 1] : pos[i];
      d[i - 1] = dl, d[i] = dh;
      pos[i - 1] = pl, pos[i] = ph;
    }
  }
};

template <int KMAX>
HGS_HD void bvh_knn(const BvhView& t, const F3& q, int k, KnnList<KMAX>& list) {
  list.init(k);
  if (t.n <= 0) return;
  uint32_t node = 1, pending = 0;
  int depth = 0;
  for (;;) {
    bool pruned = false;
    while ((int)node < t.P) {
      const Float4 a0 = t.nodes[4 * node + 0], a1 = t.nodes[4 * node + 1];
      const Float4 b0 = t.nodes[4 * node + 2], b1 = t.nodes[4 * node + 3];
      const float d0 = box_dist2f(q, a0.x, a0.y, a0.z, a1.x, a1.y, a1.z);
      const float d1 = box_dist2f(q, b0.x, b0.y, b0.z, b1.x, b1.y, b1.z);
      const int near = d1 < d0 ? 1 : 0;
      const float dn = near ? d1 : d0, df = near ? d0 : d1;
      if (!(dn < list.worst())) {
        pruned = true;
        break;
      }
      depth++;
      if (df < list.worst()) pending |= 1u << depth;
      node = 2 * node + near;
    }
    if (!pruned) {
      const int base = ((int)node - t.P) * kLeaf;
#pragma unroll
      for (int l = 0; l < kLeaf; l++) {
        const Float4 p = t.pts[base + l];
        const float d = dist2f(q, p.x, p.y, p.z);
        if (d < list.worst()) list.insert(d, base + l);
      }
    }
    for (;;) {
      if (!pending) return;
      const int lvl = highest_bit(pending);
      pending ^= 1u << lvl;
      node = (node >> (depth - lvl)) ^ 1u;
      depth = lvl;
      const Float4 m0 = t.nodes[2 * node], m1 = t.nodes[2 * node + 1];
      if (box_dist2f(q, m0.x, m0.y, m0.z, m1.x, m1.y, m1.z) < list.worst()) break;
    }
  }
}

}  // namespace hgs
