// hgs_wave_bvh.h — wave-cooperative ("packet") exact nearest-neighbour search on the implicit tree of hgs_bvh.h.
//
// gfx950 executes 64 lanes in lock-step; the queries of one wave are consecutive points of a Hilbert-sorted cloud
// (after a near-rigid transform), i.e. spatially compact.  Instead of 64 private, divergent descents — every node
// read a 64-way gather, every lane waiting for the slowest — the wave walks ONE path set: the union of the nodes any
// lane still needs.  The traversal state (node, depth, pending-sibling mask) is wave-uniform and lives in SGPRs, a
// node's two child boxes are a single aligned 64-byte read shared by all lanes, a leaf is one 128-byte read, and
// the per-lane work is the box / point distance arithmetic and the private best-so-far.  Results are identical to
// bvh_nn1 / bvh_knn (exact search; ties towards the lower original index), whatever the grouping of queries.
//
// All 64 lanes must call these functions together from wave-uniform control flow; lanes without a query pass
// active = false.
#pragma once
#include <hip/hip_runtime.h>
#include "hgs_bvh.h"

namespace hgs {

__device__ __forceinline__ void wave_nn1(const BvhView& t, const F3& q, bool active, float bound2, float& best, int& best_pos, int& best_orig) {
  best = active ? bound2 : -1.0f;  // an inactive lane wants nothing: no box distance is <= -1
  best_pos = -1;
  best_orig = 0x7fffffff;
  if (t.n <= 0) return;
  unsigned node = 1, pending = 0;
  int depth = 0;
  for (;;) {
    bool pruned = false;
    while ((int)node < t.P) {
      const float4 a0 = t.nodes[4 * node + 0], a1 = t.nodes[4 * node + 1];
      const float4 b0 = t.nodes[4 * node + 2], b1 = t.nodes[4 * node + 3];
      const float d0 = box_dist2f(q, a0.x, a0.y, a0.z, a1.x, a1.y, a1.z);
      const float d1 = box_dist2f(q, b0.x, b0.y, b0.z, b1.x, b1.y, b1.z);
      const bool w0 = d0 <= best, w1 = d1 <= best;
      const unsigned long long m0 = __ballot(w0), m1 = __ballot(w1);
      if ((m0 | m1) == 0ull) {
        pruned = true;
        break;
      }
      // the lanes that want a child vote for the closer one; the wave descends into the majority's choice first
      const int v1 = __popcll(__ballot(w1 && (!w0 || d1 < d0)));
      const int v0 = __popcll(__ballot(w0 && (!w1 || d0 <= d1)));
      unsigned first = v1 > v0 ? 1u : 0u;
      if ((first ? m1 : m0) == 0ull) first ^= 1u;
      const unsigned long long mother = first ? m0 : m1;
      depth++;
      if (mother != 0ull) pending |= 1u << depth;
      node = 2 * node + first;
    }
    if (!pruned) {
      const int base = ((int)node - t.P) * kLeaf;
#pragma unroll
      for (int l = 0; l < kLeaf; l++) {
        const float4 p = t.pts[base + l];
        const float d = dist2f(q, p.x, p.y, p.z);
        const int oi = __float_as_int(p.w);
        if (d < best || (d == best && oi < best_orig)) {
          best = d;
          best_pos = base + l;
          best_orig = oi;
        }
      }
    }
    for (;;) {
      if (!pending) return;
      const int lvl = 31 - __clz((int)pending);
      pending ^= 1u << lvl;
      node = (node >> (depth - lvl)) ^ 1u;
      depth = lvl;
      const float4 m0 = t.nodes[2 * node], m1 = t.nodes[2 * node + 1];
      if (__ballot(box_dist2f(q, m0.x, m0.y, m0.z, m1.x, m1.y, m1.z) <= best) != 0ull) break;
    }
  }
}

template <int KMAX>
__device__ __forceinline__ void wave_knn(const BvhView& t, const F3& q, bool active, int k, KnnList<KMAX>& list) {
  list.init(k);
  if (!active) {
#pragma unroll
    for (int i = 0; i < KMAX; i++) list.d[i] = -1.f;  // worst() = -1: wants nothing, never inserts
  }
  if (t.n <= 0) return;
  unsigned node = 1, pending = 0;
  int depth = 0;
  for (;;) {
    bool pruned = false;
    while ((int)node < t.P) {
      const float4 a0 = t.nodes[4 * node + 0], a1 = t.nodes[4 * node + 1];
      const float4 b0 = t.nodes[4 * node + 2], b1 = t.nodes[4 * node + 3];
      const float d0 = box_dist2f(q, a0.x, a0.y, a0.z, a1.x, a1.y, a1.z);
      const float d1 = box_dist2f(q, b0.x, b0.y, b0.z, b1.x, b1.y, b1.z);
      const float w = list.worst();
      const bool w0 = d0 < w, w1 = d1 < w;
      const unsigned long long m0 = __ballot(w0), m1 = __ballot(w1);
      if ((m0 | m1) == 0ull) {
        pruned = true;
        break;
      }
      const int v1 = __popcll(__ballot(w1 && (!w0 || d1 < d0)));
      const int v0 = __popcll(__ballot(w0 && (!w1 || d0 <= d1)));
      unsigned first = v1 > v0 ? 1u : 0u;
      if ((first ? m1 : m0) == 0ull) first ^= 1u;
      const unsigned long long mother = first ? m0 : m1;
      depth++;
      if (mother != 0ull) pending |= 1u << depth;
      node = 2 * node + first;
    }
    if (!pruned) {
      const int base = ((int)node - t.P) * kLeaf;
#pragma unroll
      for (int l = 0; l < kLeaf; l++) {
        const float4 p = t.pts[base + l];
        const float d = dist2f(q, p.x, p.y, p.z);
        if (d < list.worst()) list.insert(d, base + l);
      }
    }
    for (;;) {
      if (!pending) return;
      const int lvl = 31 - __clz((int)pending);
      pending ^= 1u << lvl;
      node = (node >> (depth - lvl)) ^ 1u;
      depth = lvl;
      const float4 m0 = t.nodes[2 * node], m1 = t.nodes[2 * node + 1];
      if (__ballot(box_dist2f(q, m0.x, m0.y, m0.z, m1.x, m1.y, m1.z) < list.worst()) != 0ull) break;
    }
  }
}

}  // namespace hgs
