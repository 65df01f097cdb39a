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
//   lpts  : float[P*32]      the same points leaf by leaf in SoA form {x[8], y[8], z[8], w[8]} (128 bytes per leaf):
//                            what the wave-cooperative walk reads (one coalesced record fetch per wave, packed-fp32 math)
//   nodes : float[(P/2)*32]  heap-ordered tree, node n (1 <= n < 2P), children 2n, 2n+1, leaves P..2P-1 (leaf l owns
//                            pts[l*LEAF .. l*LEAF+LEAF)).  Boxes are stored in GROUPS of four siblings-of-siblings:
//                            group G holds nodes 4G..4G+3 — exactly the four grandchildren of node G — as one aligned
//                            128-byte record {mnx[4], mny[4], mnz[4], mxx[4], mxy[4], mxz[4], pad[8]}, so a 4-ary
//                            step of the walk is one load and its box arithmetic pairs up for v_pk_*_f32.
//                            Empty nodes have min = +inf, max = -inf (box distance = +inf, never visited).
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
  const Float4* nodes;  // grouped boxes, 8 Float4 per group
  const Float4* pts;
  const Float4* lpts;   // SoA leaves, 8 Float4 per leaf (may be null where only the per-lane search is used)
  int P;       // number of leaves (power of two)
  int n;       // number of valid (finite) points
};

// box of node n inside the grouped layout
HGS_HD void bvh_load_box(const Float4* nodes, uint32_t n, float* mn, float* mx) {
  const float* g = reinterpret_cast<const float*>(nodes) + 32 * (size_t)(n >> 2) + (n & 3u);
  mn[0] = g[0], mn[1] = g[4], mn[2] = g[8];
  mx[0] = g[12], mx[1] = g[16], mx[2] = g[20];
}
HGS_HD void bvh_store_box(Float4* nodes, uint32_t n, const float* mn, const float* mx) {
  float* g = reinterpret_cast<float*>(nodes) + 32 * (size_t)(n >> 2) + (n & 3u);
  g[0] = mn[0], g[4] = mn[1], g[8] = mn[2];
  g[12] = mx[0], g[16] = mx[1], g[20] = mx[2];
}
HGS_HD float bvh_box_dist2(const Float4* nodes, uint32_t n, const F3& q) {
  float mn[3], mx[3];
  bvh_load_box(nodes, n, mn, mx);
  return box_dist2f(q, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]);
}

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
      const float d0 = bvh_box_dist2(t.nodes, 2 * node, q);
      const float d1 = bvh_box_dist2(t.nodes, 2 * node + 1, q);
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
      if (bvh_box_dist2(t.nodes, node, q) <= best) break;
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
  // Sorted insertion without a carried element: with d ascending, the new d[i] is the median of (d[i-1], d[i], dist)
  // — one v_med3_f32 per slot — and pos[i] follows the two comparison masks.  Equal distances keep arrival order.
  HGS_HD void insert(float dist, int p) {
    // caller guarantees dist < worst()
    bool c_hi = dist < d[KMAX - 1];
#pragma unroll
    for (int i = KMAX - 1; i > 0; i--) {
      const bool c_lo = dist < d[i - 1];
      pos[i] = c_lo ? pos[i - 1] : (c_hi ? p : pos[i]);
#if defined(__HIP_DEVICE_COMPILE__)
      d[i] = __builtin_amdgcn_fmed3f(d[i - 1], d[i], dist);
#else
      d[i] = fmaxf(d[i - 1], fminf(d[i], dist));
#endif
      c_hi = c_lo;
    }
    pos[0] = c_hi ? p : pos[0];
    d[0] = fminf(d[0], dist);
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
      const float d0 = bvh_box_dist2(t.nodes, 2 * node, q);
      const float d1 = bvh_box_dist2(t.nodes, 2 * node + 1, q);
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
      if (bvh_box_dist2(t.nodes, node, q) < list.worst()) break;
    }
  }
}

}  // namespace hgs
