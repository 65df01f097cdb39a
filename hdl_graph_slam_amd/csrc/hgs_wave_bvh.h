// hgs_wave_bvh.h — wave-cooperative ("packet") exact nearest-neighbour search on the implicit tree of hgs_bvh.h.
//
// gfx950 executes 64 lanes in lock-step; the queries of one wave are consecutive points of a Hilbert-sorted cloud
// (after a near-rigid transform), i.e. spatially compact.  Instead of 64 private, divergent descents — every node
// read a 64-way gather, every lane waiting for the slowest — the wave walks ONE path set: the union of the nodes any
// lane still needs.  The traversal state (node, depth, pending-children masks) is wave-uniform and lives in SGPRs;
// a node group or a leaf is one 128-byte record fetched once per wave (fetch_record below) and the per-lane work is
// the box / point distance arithmetic against it plus the private best-so-far.
// The boxes of a group and the points of a leaf are stored SoA so that adjacent slots form the two halves of
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 operands.
//
// The walk is 4-ary: the four grandchildren 4n..4n+3 of node n are one 128-byte group record, and a grandchild box
// that passes implies its parent box passes, so two binary levels are decided per dependent load.  When the tree
// height is odd the walk starts at the virtual node 0 whose group holds {empty, empty, node 2, node 3}.  The children
// of a leaf-parent group are visited in place, re-tested against the tightened bounds with their box distances still
// in registers; a pending child of an inner group is entered directly when popped (its own group evaluation is the
// re-test).  Children are tried nearest-first as seen by the middle lane of the wave.
//
// Results are identical to bvh_nn1 / bvh_knn (exact search; ties towards the lower original index), whatever the
// grouping of queries.  All 64 lanes must call these functions together from wave-uniform control flow; lanes
// without a query pass active = false.
#pragma once
#include <hip/hip_runtime.h>
#include "hgs_bvh.h"

namespace hgs {

typedef float hgs_f16v __attribute__((ext_vector_type(16)));
typedef float hgs_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ hgs_f2 pk_axis_gap(hgs_f2 mn, hgs_f2 mx, hgs_f2 q) {
  const hgs_f2 a = mn - q, b = q - mx;
  return hgs_f2{fmaxf(fmaxf(a.x, b.x), 0.f), fmaxf(fmaxf(a.y, b.y), 0.f)};
}
// squared distances from q to the two boxes whose bounds sit in the two halves of the operands (== box_dist2f)
__device__ __forceinline__ hgs_f2 pk_box_dist2(hgs_f2 qx, hgs_f2 qy, hgs_f2 qz, hgs_f2 mnx, hgs_f2 mny, hgs_f2 mnz, hgs_f2 mxx, hgs_f2 mxy, hgs_f2 mxz) {
  const hgs_f2 dx = pk_axis_gap(mnx, mxx, qx), dy = pk_axis_gap(mny, mxy, qy), dz = pk_axis_gap(mnz, mxz, qz);
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}
// squared distances from q to two points (bit-identical to dist2f per half: exact products, fused multiply-adds)
__device__ __forceinline__ hgs_f2 pk_dist2(hgs_f2 qx, hgs_f2 qy, hgs_f2 qz, hgs_f2 px, hgs_f2 py, hgs_f2 pz) {
  const hgs_f2 dx = qx - px, dy = qy - py, dz = qz - pz;
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

// One packet walk as an explicit state machine: every step consumes exactly one 128-byte record (a group of boxes or
// a leaf of points) and decides which record comes next.
// Lane concept:  bool wants(float box_d2) const;
//                void visit_leaf(const hgs_f16v& xy, const hgs_f16v& zw, hgs_f2 qx, hgs_f2 qy, hgs_f2 qz, int base);
//                    xy = {x[8], y[8]}, zw = {z[8], w[8]} of the leaf whose first point is sorted position `base`
enum { WALK_GROUP = 0, WALK_LEAF = 1, WALK_DONE = 2 };

template <class Lane>
struct PacketWalk {
  // wave-uniform
  unsigned node;            // WALK_GROUP: the node whose group (its four grandchildren) is evaluated next
  int bd;                   // binary depth of `node`
  unsigned long long pend;  // 4 bits per group level: children of inner groups some lane wanted, not yet entered
  unsigned todo, lbase;     // leaves of the current leaf-parent group still to visit, node id of its slot 0
  int nextc;
  int kind;
  unsigned ldnode;          // WALK_LEAF: the leaf node to visit next
  unsigned skip_lo, skip_n; // leaf nodes [skip_lo, skip_lo + skip_n) are never visited (the caller has handled them)
  int order_lane;           // the lane whose preference orders the children (middle of the packet's active lanes)
  // per lane
  float d[4];
  hgs_f2 qx, qy, qz;
  Lane lane;

  __device__ __forceinline__ void start(const BvhView& t, const F3& q, int k) {
    qx = hgs_f2{q.x, q.x}, qy = hgs_f2{q.y, q.y}, qz = hgs_f2{q.z, q.z};
    pend = 0, todo = 0, lbase = 0, nextc = 0, ldnode = 1, skip_lo = 0, skip_n = 0, order_lane = 32;
    d[0] = d[1] = d[2] = d[3] = 0.f;
    if (t.n <= 0) {
      kind = WALK_DONE, node = 1, bd = 0;
    } else if (k == 0) {
      kind = WALK_LEAF, node = 1, bd = 0;  // the root is the only leaf
    } else {
      kind = WALK_GROUP;
      node = (k & 1) ? 0u : 1u;
      bd = (k & 1) ? -1 : 0;
    }
  }
  __device__ __forceinline__ const hgs_f16v* record(const BvhView& t) const {
    const float4* p = kind == WALK_LEAF ? t.lpts + 8 * (size_t)(ldnode - (unsigned)t.P) : (kind == WALK_GROUP ? t.nodes + 8 * (size_t)node : t.nodes);
    return reinterpret_cast<const hgs_f16v*>(p);
  }
  __device__ __forceinline__ void step(const BvhView& t, int k, const hgs_f16v& lo, const hgs_f16v& hi) {
    bool descended = false;
    if (kind == WALK_LEAF) {
      lane.visit_leaf(lo, hi, qx, qy, qz, ((int)ldnode - t.P) * kLeaf);
    } else {
      // lo = {mnx[4], mny[4], mnz[4], mxx[4]}  hi = {mxy[4], mxz[4], pad[8]}
      const hgs_f2 d01 = pk_box_dist2(qx, qy, qz, hgs_f2{lo[0], lo[1]}, hgs_f2{lo[4], lo[5]}, hgs_f2{lo[8], lo[9]}, hgs_f2{lo[12], lo[13]},
                                      hgs_f2{hi[0], hi[1]}, hgs_f2{hi[4], hi[5]});
      const hgs_f2 d23 = pk_box_dist2(qx, qy, qz, hgs_f2{lo[2], lo[3]}, hgs_f2{lo[6], lo[7]}, hgs_f2{lo[10], lo[11]}, hgs_f2{lo[14], lo[15]},
                                      hgs_f2{hi[2], hi[3]}, hgs_f2{hi[6], hi[7]});
      d[0] = d01.x, d[1] = d01.y, d[2] = d23.x, d[3] = d23.y;
      unsigned any = 0;
      float dw[4];  // distance of the children this lane wants, +inf for the others
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const bool w = lane.wants(d[c]);
        if (__ballot(w) != 0ull) any |= 1u << c;
        dw[c] = w ? d[c] : INFINITY;
      }
      const float dmin = fminf(fminf(dw[0], dw[1]), fminf(dw[2], dw[3]));
      const int pref = dmin == INFINITY ? -1 : (dw[0] == dmin ? 0 : (dw[1] == dmin ? 1 : (dw[2] == dmin ? 2 : 3)));
      if (any) {
        // order: the nearest child the middle lane of the packet wants (else the lowest wanted slot)
        int cstar = __builtin_amdgcn_readlane(pref, order_lane);
        if (cstar < 0) cstar = __builtin_ctz(any);
        const int cd = bd + 2;
        if (cd == k) {
          todo = any, lbase = node << 2, nextc = cstar;
        } else {
          pend |= (unsigned long long)(any & ~(1u << cstar)) << (4 * (cd >> 1));
          node = (node << 2) + (unsigned)cstar;
          bd = cd;
          descended = true;
        }
      }
    }
    if (descended) return;  // kind stays WALK_GROUP
    for (;;) {
      if (todo) {
        // leaves of a leaf-parent group: re-test against the bounds the previous leaves tightened
        const int c = nextc;
        todo &= ~(1u << c);
        nextc = todo ? __builtin_ctz(todo) : 0;
        if (lbase + (unsigned)c - skip_lo < skip_n) continue;
        const float dc = c == 0 ? d[0] : (c == 1 ? d[1] : (c == 2 ? d[2] : d[3]));
        if (__ballot(lane.wants(dc)) != 0ull) {
          kind = WALK_LEAF;
          ldnode = lbase + (unsigned)c;
          return;
        }
        continue;
      }
      if (!pend) {
        kind = WALK_DONE;
        return;
      }
      // backtrack: enter the lowest pending child of the deepest pending group
      const int idx = (63 - __clzll((long long)pend)) >> 2;
      const unsigned nib = (unsigned)(pend >> (4 * idx)) & 0xfu;
      const int c = __builtin_ctz(nib);
      pend &= ~(1ull << (4 * idx + c));
      const int pcd = 2 * idx + (k & 1);
      node = ((node >> (bd - pcd)) & ~3u) + (unsigned)c;
      bd = pcd;
      kind = WALK_GROUP;
      return;
    }
  }
};

// Record delivery.  A record (128 bytes) is the same for all 64 lanes.  Scalar loads would put it straight into
// SGPRs, but the scalar-memory path sustains only a few outstanding misses per CU: with every wave of the chip chasing
// its own chain of cold records, kernels built on s_load sat ~50 % of their wave-cycles in s_waitcnt and ran at the
// same speed whatever their occupancy or VALU count (profiles/r01_*pmc*).  The vector path has the memory-level
// parallelism: lane l fetches word l of the record (one coalesced 128-byte global_load_dword), the wave parks it in
// its private 128-byte LDS slot and reads it back as broadcast ds_read_b128 — all 64 lanes end up with the whole
// record in VGPRs, L2-hit latency ~200 cycles instead of ~1000.
__device__ __forceinline__ void fetch_record(const hgs_f16v* rec, float* slot /* wave-private, 128-byte aligned, 32 floats */, hgs_f16v& lo, hgs_f16v& hi) {
  const int l = (int)(__lane_id() & 31u);
  const float v = reinterpret_cast<const float*>(rec)[l];
  __builtin_amdgcn_wave_barrier();  // the previous record's reads are issued before the slot is overwritten
  slot[l] = v;
  __builtin_amdgcn_wave_barrier();
  const hgs_f16v* r = reinterpret_cast<const hgs_f16v*>(slot);
  lo = r[0], hi = r[1];
}

// The leaves a walk visited, in order (wave-uniform; ids in wave-private LDS).  count may run past cap: the list is then
// incomplete and the caller must not use it.
struct LeafLog {
  unsigned* ids;
  int cap, count;
};

// NW independent packet walks in lock-step (NW = 1 everywhere today: with VGPR-resident records occupancy provides the
// overlap; the multi-walk form is kept because it costs nothing at NW = 1).
template <class Lane, int NW>
__device__ __forceinline__ void wave_walk_multi(const BvhView& t, PacketWalk<Lane> (&w)[NW], float* slots /* NW * 32 floats, wave-private LDS */,
                                                LeafLog* log = nullptr /* NW == 1 only */) {
  const int k = 31 - __clz(t.P);  // P = 2^k leaves
  for (;;) {
    bool alive = false;
#pragma unroll
    for (int i = 0; i < NW; i++) alive = alive || w[i].kind != WALK_DONE;
    if (!alive) return;
    if (NW == 1 && log && w[0].kind == WALK_LEAF) {
      if (log->count < log->cap && (__lane_id() & 63u) == 0u) log->ids[log->count] = w[0].ldnode;
      log->count++;
    }
    hgs_f16v lo[NW], hi[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) fetch_record(w[i].record(t), slots + 32 * i, lo[i], hi[i]);
#pragma unroll
    for (int i = 0; i < NW; i++)
      if (w[i].kind != WALK_DONE) w[i].step(t, k, lo[i], hi[i]);
  }
}

// Best-so-far of one lane as a single 64-bit key {d2 bits : original index}: squared distances are non-negative
// floats, whose bit patterns order like unsigned integers, so "closer, or equally close with the lower original index"
// is ONE unsigned 64-bit compare (v_cmp_lt_u64) instead of three compares and mask algebra on the scalar unit.
struct Nn1Lane {
  unsigned long long key;
  int pos;
  __device__ __forceinline__ float best() const { return __uint_as_float((unsigned)(key >> 32)); }
  __device__ __forceinline__ int orig() const { return (int)(unsigned)key; }
  __device__ __forceinline__ bool wants(float d) const { return d <= best(); }
  __device__ __forceinline__ void consider(float d, int oi, int p) {
    const unsigned long long nk = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)oi;
    const bool better = nk < key;
    key = better ? nk : key;
    pos = better ? p : pos;
  }
  __device__ __forceinline__ void visit_leaf(const hgs_f16v& xy, const hgs_f16v& zw, hgs_f2 qx, hgs_f2 qy, hgs_f2 qz, int base) {
#pragma unroll
    for (int l = 0; l < 8; l += 2) {
      const hgs_f2 d = pk_dist2(qx, qy, qz, hgs_f2{xy[l], xy[l + 1]}, hgs_f2{xy[8 + l], xy[9 + l]}, hgs_f2{zw[l], zw[l + 1]});
      consider(d.x, __float_as_int(zw[8 + l]), base + l);
      consider(d.y, __float_as_int(zw[9 + l]), base + l + 1);
    }
  }
};

// Exact 1-NN of NW queries per lane among the points with d2 <= bound2.  `seed` (a position in the sorted target, or
// anything out of range) is an optional starting candidate — typically the correspondence of the previous
// linearisation: any real target point is a valid upper bound, so the result is unchanged, only the ball the wave has
// to cover shrinks from max_correspondence_distance to about the true NN distance.
template <int NW>
__device__ __forceinline__ void wave_nn1(const BvhView& t, float* slots, const F3 (&q)[NW], const bool (&active)[NW], float bound2, const int (&seed)[NW],
                                         float (&best)[NW], int (&best_pos)[NW], int (&best_orig)[NW], int qpw = 64) {
  PacketWalk<Nn1Lane> w[NW];
  const int k = 31 - __clz(t.P);
#pragma unroll
  for (int i = 0; i < NW; i++) {
    Nn1Lane& lane = w[i].lane;
    // a lane without a query sits infinitely far away with an unbeatable key: it wants no box and accepts no point
    const F3 qq = active[i] ? q[i] : F3{FLT_MAX, FLT_MAX, FLT_MAX};
    lane.key = active[i] ? (((unsigned long long)__float_as_uint(bound2) << 32) | 0x7fffffffull) : 0ull;
    lane.pos = -1;
    if (active[i] && seed[i] >= 0 && seed[i] < t.n) {
      const float4 p = t.pts[seed[i]];
      const float d = dist2f(qq, p.x, p.y, p.z);
      if (d <= bound2) lane.key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w), lane.pos = seed[i];
    }
    w[i].start(t, qq, k);
    w[i].order_lane = qpw >> 1;
  }
  wave_walk_multi<Nn1Lane, NW>(t, w, slots);
#pragma unroll
  for (int i = 0; i < NW; i++) best[i] = w[i].lane.best(), best_pos[i] = w[i].lane.pos, best_orig[i] = w[i].lane.orig();
}

// ---- 1-NN, "quad" walk (round 3): what k_gicp_linearize and k_fitness run ------------------------------------------
// The generic walk above fetches one 128-byte record per step, waits for it, and keeps a 64-bit {d2 : original index} key
// per lane up to date at every tested point.  Measured (profiles/r02_fast_gicp_pmc_summary.md): the 1-NN kernels are bound by
// VALU issue (a leaf visit cost ~80 vector instructions, 32 of them the per-point key compare/select chain, a group step
// ~50) plus one dependent L2 round trip per step.  This walk removes both:
//   * QUAD FETCH.  The records of the four children 4n..4n+3 of a node are adjacent in memory (group records are indexed by
//     node id, leaf records by leaf id), so when a group has a wanted child the wave fetches all four child records with ONE
//     512-byte load (lane l: bytes 8l..8l+7) and parks them in a wave-private LDS slot of the level.  The child descended
//     into AND every sibling popped later are then read from LDS: a packet waits for memory once per group that has wanted
//     children (~25 times per 64 queries) instead of once per step (~53), and the leaves of a leaf-parent group are all
//     there after one wait.  A level's slot stays valid until the walk evaluates another node of the parent level, which
//     cannot happen before every pending child of the current one has been popped (pops are deepest-first).  Slots exist
//     for the leaf quad and the three group levels above it; all higher levels share one slot and re-fetch on a pop (rare).
//   * MIN-ONLY LEAVES.  A leaf visit keeps, per lane, the minimum squared distance (v_min3 tree), the leaf that produced it
//     and how many visited leaves reached exactly that minimum: 24 + 4 + 6 vector instructions.  Position and original index
//     of the winner are resolved once per packet (nn1_resolve_leaf: the lane re-reads its best leaf and takes the lowest
//     original index among the points at exactly the minimum — bit-identical distances, same fma chain).  A lane whose minimum
//     was reached in two different leaves (equidistant points in different leaves: duplicated clouds, regular grids) cannot
//     be resolved from one leaf: the caller then runs the exact keyed walk (wave_nn1) for that wave.  getFitnessScore needs
//     the distance only — no leaf, no count, no resolve.
//   * CHILD ORDER only where it pays: an unseeded walk (first linearisation, NDT's fitness pass) descends nearest-first as seen
//     by the middle lane (93 vs 150 steps per packet); a seeded walk takes the lowest wanted child (the seed's bound is
//     already tight: 53.4 vs 53.5 steps, and 14 vector instructions less per group).
// Results are those of wave_nn1 (exact; ties -> lowest original index) except that a point at a squared distance of exactly
// FLT_MAX is not found by an unbounded search (the exclusive bound is clamped to FLT_MAX so that empty boxes, whose distance
// is +inf, are never wanted).
typedef float hgs_f8v __attribute__((ext_vector_type(8)));
// (measurement builds only, -DHGS_KNN_PROBE: scripts/probes/knn_probe.py — per-wave phase clocks and step counts of k_knn_cov; the product build compiles none of it)
#ifdef HGS_KNN_PROBE
__device__ unsigned long long g_knn_probe[(1 << 16) * 8];
__device__ __forceinline__ unsigned knn_probe_wave() { return (((unsigned)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 0xffffu; }
#define HGS_PROBE_COUNT(kind) do { if ((__lane_id() & 63u) == 0u) g_knn_probe[knn_probe_wave() * 8 + 5 + (kind)] += 1ull; } while (0)
#define HGS_PROBE_TIME(slot) do { if ((__lane_id() & 63u) == 0u) g_knn_probe[knn_probe_wave() * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define HGS_PROBE_COUNT(kind) ((void)0)
#define HGS_PROBE_TIME(slot) ((void)0)
#endif
constexpr int kParkLevels = 4;                     // leaf quad + three group levels with a slot of their own
constexpr int kParkSlots = kParkLevels + 1;        // + the slot all higher levels share
constexpr int kParkFloats = kParkSlots * 128;      // wave-private LDS floats (2560 bytes)

// the four records (512 contiguous bytes) starting at `quad` -> LDS slot
__device__ __forceinline__ void fetch_quad(const Float4* quad, float* slot) {
  const int l = (int)(__lane_id() & 63u);
  const hgs_f2 v = reinterpret_cast<const hgs_f2*>(quad)[l];
  __builtin_amdgcn_wave_barrier();  // earlier reads of this slot are issued before it is overwritten
  reinterpret_cast<hgs_f2*>(slot)[l] = v;
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float nn1_exclusive_bound(float d2_inclusive) {
  const unsigned u = __float_as_uint(d2_inclusive) + 1u;  // next float up: "m < bound" == "m <= d2_inclusive"
  return __uint_as_float(u < 0x7f7fffffu ? u : 0x7f7fffffu);
}

// TRACK: also keep the best leaf and the number of leaves that reached the minimum (what nn1_resolve_leaf needs)
template <bool ORDERED, bool TRACK>
__device__ __forceinline__ void wave_nn1_quad(const BvhView& t, float* park /* kParkFloats, wave-private, 512-byte aligned */, const F3& q0, bool active,
                                              float bound2, int seed, int order_lane, float& best_out, int& leaf_out, bool& tie_out, bool& found_out) {
  const int k = 31 - __clz(t.P);
  const F3 q = active ? q0 : F3{FLT_MAX, FLT_MAX, FLT_MAX};  // a lane without a query is infinitely far from everything
  const hgs_f2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
  float best = active ? nn1_exclusive_bound(bound2) : 0.f;
  if (active && seed >= 0 && seed < t.n) {
    const Float4 p = t.pts[seed];
    const float d = dist2f(q, p.x, p.y, p.z);
    if (d <= bound2) best = nn1_exclusive_bound(d);  // the seed's own leaf will be visited (box_d2 <= d) and set the leaf
  }
  const float best_init = best;
  int leaf = -1, cnt = 0;
  float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
  auto wanted = [&]() -> unsigned {
    return (__ballot(d0 <= best) != 0ull ? 1u : 0u) | (__ballot(d1 <= best) != 0ull ? 2u : 0u) | (__ballot(d2 <= best) != 0ull ? 4u : 0u) |
           (__ballot(d3 <= best) != 0ull ? 8u : 0u);
  };
  auto visit_leaf = [&](const float* rec, int ldnode) {
    const hgs_f16v xy = *reinterpret_cast<const hgs_f16v*>(rec);
    const hgs_f8v z = *reinterpret_cast<const hgs_f8v*>(rec + 16);
    const hgs_f2 a = pk_dist2(qx, qy, qz, hgs_f2{xy[0], xy[1]}, hgs_f2{xy[8], xy[9]}, hgs_f2{z[0], z[1]});
    const hgs_f2 b = pk_dist2(qx, qy, qz, hgs_f2{xy[2], xy[3]}, hgs_f2{xy[10], xy[11]}, hgs_f2{z[2], z[3]});
    const hgs_f2 c = pk_dist2(qx, qy, qz, hgs_f2{xy[4], xy[5]}, hgs_f2{xy[12], xy[13]}, hgs_f2{z[4], z[5]});
    const hgs_f2 e = pk_dist2(qx, qy, qz, hgs_f2{xy[6], xy[7]}, hgs_f2{xy[14], xy[15]}, hgs_f2{z[6], z[7]});
    const float m = fminf(fminf(fminf(a.x, a.y), fminf(b.x, b.y)), fminf(fminf(c.x, c.y), fminf(e.x, e.y)));
    if (TRACK) {
      const bool better = m < best;
      cnt = better ? 1 : cnt + (m == best ? 1 : 0);
      leaf = better ? ldnode : leaf;
    }
    best = fminf(best, m);
  };
  if (t.n > 0 && k <= 1) {  // one or two leaves (at most 16 points): no boxes worth testing
    const int l = (int)(__lane_id() & 31u);
    for (int lf = 0; lf < t.P; lf++) {
      const float v = reinterpret_cast<const float*>(t.lpts + 8 * lf)[l];
      __builtin_amdgcn_wave_barrier();
      park[l] = v;
      __builtin_amdgcn_wave_barrier();
      visit_leaf(park, t.P + lf);
    }
  } else if (t.n > 0) {
    unsigned node = (k & 1) ? 0u : 1u;
    int bd = (k & 1) ? -1 : 0;
    unsigned long long pend = 0;
    const auto slot_of = [&](int depth) { const int s = (k - depth) >> 1; return park + 128 * (s < kParkLevels ? s : kParkLevels); };
    fetch_quad(t.nodes, slot_of(bd));
    for (;;) {
      {  // the group of `node`: boxes of its four grandchildren  {mnx[4], mny[4], mnz[4], mxx[4]} {mxy[4], mxz[4]}
        const float* rec = slot_of(bd) + 32 * (node & 3u);
        const hgs_f16v lo = *reinterpret_cast<const hgs_f16v*>(rec);
        const hgs_f8v hi = *reinterpret_cast<const hgs_f8v*>(rec + 16);
        const hgs_f2 d01 = pk_box_dist2(qx, qy, qz, hgs_f2{lo[0], lo[1]}, hgs_f2{lo[4], lo[5]}, hgs_f2{lo[8], lo[9]}, hgs_f2{lo[12], lo[13]}, hgs_f2{hi[0], hi[1]},
                                        hgs_f2{hi[4], hi[5]});
        const hgs_f2 d23 = pk_box_dist2(qx, qy, qz, hgs_f2{lo[2], lo[3]}, hgs_f2{lo[6], lo[7]}, hgs_f2{lo[10], lo[11]}, hgs_f2{lo[14], lo[15]}, hgs_f2{hi[2], hi[3]},
                                        hgs_f2{hi[6], hi[7]});
        d0 = d01.x, d1 = d01.y, d2 = d23.x, d3 = d23.y;
      }
      const unsigned any = wanted();
      if (any) {
        int cstar = __builtin_ctz(any);
        if (ORDERED) {  // the nearest child the middle lane of the packet wants (else the lowest wanted slot)
          const float w0 = d0 <= best ? d0 : INFINITY, w1 = d1 <= best ? d1 : INFINITY, w2 = d2 <= best ? d2 : INFINITY, w3 = d3 <= best ? d3 : INFINITY;
          const float dmin = fminf(fminf(w0, w1), fminf(w2, w3));
          const int pref = dmin == INFINITY ? -1 : (w0 == dmin ? 0 : (w1 == dmin ? 1 : (w2 == dmin ? 2 : 3)));
          const int p = __builtin_amdgcn_readlane(pref, order_lane);
          if (p >= 0) cstar = p;
        }
        const int cd = bd + 2;
        const unsigned base = node << 2;
        if (cd == k) {  // the children are leaves: all four records with one fetch, visited in place
          fetch_quad(t.lpts + 8 * (size_t)(base - (unsigned)t.P), park);
          unsigned todo = any;
          int c = cstar;
          while (todo) {
            todo &= ~(1u << c);
            visit_leaf(park + 32 * c, (int)(base + (unsigned)c));
            todo &= wanted();  // the visit tightened the bounds: drop the leaves nobody wants any more
            c = todo ? __builtin_ctz(todo) : 0;
          }
        } else {
          pend |= (unsigned long long)(any & ~(1u << cstar)) << (4 * (cd >> 1));
          fetch_quad(t.nodes + 8 * (size_t)base, slot_of(cd));
          node = base + (unsigned)cstar;
          bd = cd;
          continue;
        }
      }
      if (!pend) break;
      // backtrack: the lowest pending child of the deepest pending group; its record is parked unless its level shares the top slot
      const int idx = (63 - __clzll((long long)pend)) >> 2;
      const unsigned nib = (unsigned)(pend >> (4 * idx)) & 0xfu;
      const int c = __builtin_ctz(nib);
      pend &= ~(1ull << (4 * idx + c));
      const int pcd = 2 * idx + (k & 1);
      node = ((node >> (bd - pcd)) & ~3u) + (unsigned)c;
      bd = pcd;
      if (((k - bd) >> 1) >= kParkLevels) fetch_quad(t.nodes + 8 * (size_t)(node & ~3u), slot_of(bd));
    }
  }
  best_out = best;
  leaf_out = leaf;
  tie_out = TRACK && leaf >= 0 && cnt > 1;
  found_out = best < best_init;
}

// Position (sorted) and original index of the point of leaf node `leaf` at squared distance exactly d2 from q with the lowest
// original index.  Per-lane gather of the lane's own best leaf (128 contiguous bytes of the AoS point array).
__device__ __forceinline__ void nn1_resolve_leaf(const BvhView& t, const F3& q, int leaf, float d2, int& pos_out, int& orig_out) {
  int pos = -1, orig = 0x7fffffff;
  if (leaf >= 0) {
    const int base = (leaf - t.P) * kLeaf;
#pragma unroll
    for (int l = 0; l < kLeaf; l++) {
      const Float4 p = t.pts[base + l];
      const float d = dist2f(q, p.x, p.y, p.z);
      const int oi = __float_as_int(p.w);
      const bool take = d == d2 && oi < orig;
      orig = take ? oi : orig;
      pos = take ? base + l : pos;
    }
  }
  pos_out = pos, orig_out = pos >= 0 ? orig : -1;
}

// The quad walk for any Lane (concept of PacketWalk above): what k_knn_cov and the prefilter's outlier walks run.  Same
// traversal as wave_walk_multi<Lane, 1> — nearest-first by the order lane, leaves of a leaf-parent group in place with the
// bounds re-tested after every visit, pending children entered directly, the skip window, the leaf log — with the records
// fetched four at a time and parked per level (fetch_quad): one memory wait per group that has wanted children.
template <class Lane>
__device__ __forceinline__ void wave_walk_quad(const BvhView& t, Lane& lane, const F3& q, float* park /* kParkFloats */, int order_lane, unsigned skip_lo = 0,
                                               unsigned skip_n = 0, LeafLog* log = nullptr) {
  const int k = 31 - __clz(t.P);
  if (t.n <= 0) return;
  const hgs_f2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
  auto visit = [&](const float* rec, unsigned ldnode) {
    if (log) {
      if (log->count < log->cap && (__lane_id() & 63u) == 0u) log->ids[log->count] = ldnode;
      log->count++;
    }
    HGS_PROBE_COUNT(1);
    const hgs_f16v xy = *reinterpret_cast<const hgs_f16v*>(rec);
    const hgs_f16v zw = *reinterpret_cast<const hgs_f16v*>(rec + 16);
    lane.visit_leaf(xy, zw, qx, qy, qz, ((int)ldnode - t.P) * kLeaf);
  };
  if (k <= 1) {
    const int l = (int)(__lane_id() & 31u);
    for (int lf = 0; lf < t.P; lf++) {
      if ((unsigned)(t.P + lf) - skip_lo < skip_n) continue;
      const float v = reinterpret_cast<const float*>(t.lpts + 8 * lf)[l];
      __builtin_amdgcn_wave_barrier();
      park[l] = v;
      __builtin_amdgcn_wave_barrier();
      visit(park, (unsigned)(t.P + lf));
    }
    return;
  }
  float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
  auto wanted = [&]() -> unsigned {
    return (__ballot(lane.wants(d0)) != 0ull ? 1u : 0u) | (__ballot(lane.wants(d1)) != 0ull ? 2u : 0u) | (__ballot(lane.wants(d2)) != 0ull ? 4u : 0u) |
           (__ballot(lane.wants(d3)) != 0ull ? 8u : 0u);
  };
  unsigned node = (k & 1) ? 0u : 1u;
  int bd = (k & 1) ? -1 : 0;
  unsigned long long pend = 0;
  const auto slot_of = [&](int depth) { const int s = (k - depth) >> 1; return park + 128 * (s < kParkLevels ? s : kParkLevels); };
  fetch_quad(t.nodes, slot_of(bd));
  for (;;) {
    HGS_PROBE_COUNT(0);
    {
      const float* rec = slot_of(bd) + 32 * (node & 3u);
      const hgs_f16v lo = *reinterpret_cast<const hgs_f16v*>(rec);
      const hgs_f8v hi = *reinterpret_cast<const hgs_f8v*>(rec + 16);
      const hgs_f2 d01 = pk_box_dist2(qx, qy, qz, hgs_f2{lo[0], lo[1]}, hgs_f2{lo[4], lo[5]}, hgs_f2{lo[8], lo[9]}, hgs_f2{lo[12], lo[13]}, hgs_f2{hi[0], hi[1]},
                                      hgs_f2{hi[4], hi[5]});
      const hgs_f2 d23 = pk_box_dist2(qx, qy, qz, hgs_f2{lo[2], lo[3]}, hgs_f2{lo[6], lo[7]}, hgs_f2{lo[10], lo[11]}, hgs_f2{lo[14], lo[15]}, hgs_f2{hi[2], hi[3]},
                                      hgs_f2{hi[6], hi[7]});
      d0 = d01.x, d1 = d01.y, d2 = d23.x, d3 = d23.y;
    }
    const unsigned any = wanted();
    if (any) {
      int cstar = __builtin_ctz(any);
      {  // the nearest child the middle lane of the packet wants (else the lowest wanted slot)
        const float w0 = lane.wants(d0) ? d0 : INFINITY, w1 = lane.wants(d1) ? d1 : INFINITY, w2 = lane.wants(d2) ? d2 : INFINITY, w3 = lane.wants(d3) ? d3 : INFINITY;
        const float dmin = fminf(fminf(w0, w1), fminf(w2, w3));
        const int pref = dmin == INFINITY ? -1 : (w0 == dmin ? 0 : (w1 == dmin ? 1 : (w2 == dmin ? 2 : 3)));
        const int p = __builtin_amdgcn_readlane(pref, order_lane);
        if (p >= 0) cstar = p;
      }
      const int cd = bd + 2;
      const unsigned base = node << 2;
      if (cd == k) {
        unsigned todo = any;
        for (unsigned c = 0; c < 4; c++)
          if (base + c - skip_lo < skip_n) todo &= ~(1u << c);  // leaves the caller has handled
        if (todo) {
          fetch_quad(t.lpts + 8 * (size_t)(base - (unsigned)t.P), park);
          int c = (todo >> cstar) & 1u ? cstar : __builtin_ctz(todo);
          while (todo) {
            todo &= ~(1u << c);
            visit(park + 32 * c, base + (unsigned)c);
            todo &= wanted();
            c = todo ? __builtin_ctz(todo) : 0;
          }
        }
      } else {
        pend |= (unsigned long long)(any & ~(1u << cstar)) << (4 * (cd >> 1));
        fetch_quad(t.nodes + 8 * (size_t)base, slot_of(cd));
        node = base + (unsigned)cstar;
        bd = cd;
        continue;
      }
    }
    if (!pend) return;
    const int idx = (63 - __clzll((long long)pend)) >> 2;
    const unsigned nib = (unsigned)(pend >> (4 * idx)) & 0xfu;
    const int c = __builtin_ctz(nib);
    pend &= ~(1ull << (4 * idx + c));
    const int pcd = 2 * idx + (k & 1);
    node = ((node >> (bd - pcd)) & ~3u) + (unsigned)c;
    bd = pcd;
    if (((k - bd) >> 1) >= kParkLevels) fetch_quad(t.nodes + 8 * (size_t)(node & ~3u), slot_of(bd));
  }
}

// ---- k-NN radius: sorted list of the k smallest squared distances only (no positions) -----------------------------
// With d ascending, inserting x makes the new d[i] the median of (d[i-1], d[i], x): one v_med3_f32 per slot.
// INCLUSIVE: prune with box_d2 <= bound instead of <, see wants().
// MARK: the lane also remembers WHICH leaves gave it a candidate (a point at or within its bound of the moment — a superset of the
// leaves that hold its final k nearest, equidistant candidates at the k-th distance included): list[e * stride], e < cnt.  cnt runs
// past cap when the list overflows (the caller then must not use it).  k_knn_cov's gather pass reads it: every lane sums over ITS
// leaves instead of all lanes over the union of everybody's.
template <int KMAX, bool INCLUSIVE = false, bool MARK = false>
struct KnnRadiusLane {
  unsigned* list;
  int stride, cnt, cap;
  float d[KMAX];  // slots [KMAX-k, KMAX) are live, the others hold -1 and never move
  __device__ __forceinline__ void init(int k, bool active) {
#pragma unroll
    for (int i = 0; i < KMAX; i++) d[i] = (!active || i < KMAX - k) ? -1.f : FLT_MAX;
  }
  __device__ __forceinline__ float worst() const { return d[KMAX - 1]; }
  // INCLUSIVE (<= instead of <): a leaf whose box is exactly as far as the current k-th distance cannot shorten the list, but it may
  // hold a point AT the final k-th distance — one of the equidistant candidates the gather pass chooses from by original index.
  // With <= the leaves this walk visits are a superset of the leaves the gather pass needs (box_d2 <= r2 <= worst at any time),
  // which is what lets k_knn_cov<.., REPLAY> replay the visited leaves instead of walking the tree a second time.
  __device__ __forceinline__ bool wants(float box_d2) const { return INCLUSIVE ? box_d2 <= worst() : box_d2 < worst(); }
  __device__ __forceinline__ void insert(float x) {  // x < worst()
#pragma unroll
    for (int i = KMAX - 1; i > 0; i--) d[i] = __builtin_amdgcn_fmed3f(d[i - 1], d[i], x);
    d[0] = fminf(d[0], x);
  }
  // The same from wave-uniform control flow (a lane with nothing to insert passes FLT_MAX, which leaves its list as it is): no
  // divergent branch around the chain.  Stopping the chain early at the first five-slot segment no lane's value reaches (13 instead
  // of 20 v_med3 on average) was measured and is slower (covariance stage 3.94 vs 3.56 ms: the ballots serialise the chain).
  __device__ __forceinline__ void insert_wave(float x) { insert(x); }
  __device__ __forceinline__ void visit_leaf(const hgs_f16v& xy, const hgs_f16v& zw, hgs_f2 qx, hgs_f2 qy, hgs_f2 qz, int base) {
    hgs_f2 dd[4];
#pragma unroll
    for (int l = 0; l < 8; l += 2) dd[l >> 1] = pk_dist2(qx, qy, qz, hgs_f2{xy[l], xy[l + 1]}, hgs_f2{xy[8 + l], xy[9 + l]}, hgs_f2{zw[l], zw[l + 1]});
    if (MARK) {
      const float m = fminf(fminf(fminf(dd[0].x, dd[0].y), fminf(dd[1].x, dd[1].y)), fminf(fminf(dd[2].x, dd[2].y), fminf(dd[3].x, dd[3].y)));
      if (m <= worst()) {  // <=: a point exactly at the bound cannot enter the list but may be one of the equidistant k-th neighbours
        if (cnt < cap) list[cnt * stride] = (unsigned)(base >> 3);
        cnt++;
      }
    }
#pragma unroll
    for (int l = 0; l < 4; l++) {
      if (__ballot(dd[l].x < worst()) != 0ull) { HGS_PROBE_COUNT(2); insert_wave(dd[l].x < worst() ? dd[l].x : FLT_MAX); }
      if (__ballot(dd[l].y < worst()) != 0ull) { HGS_PROBE_COUNT(2); insert_wave(dd[l].y < worst() ? dd[l].y : FLT_MAX); }
    }
  }
};

// ---- k-NN gather: all points with d2 < r2 summed for the covariance; of those at exactly r2 the `ties_left` with the
// lowest original indices (the order a kd-tree k-NN returns equal distances in) are remembered and added afterwards.
// Every lane has at least one such point — its k-th neighbour — and almost never a second one, so the common
// instantiation (TIES = 1) tracks a single minimum branch-free; TIES = 4 keeps a sorted list and is only run by waves in
// which some lane needs more than one point at the k-th distance (duplicates / exactly equidistant points); a lane that
// needs more than four (regular grids) gets them four at a time from further walks that only look at the points at
// exactly r2 with an original index above the last one taken (rearm_ties).
template <int TIES>
struct KnnGatherLane {
  float r2;       // squared distance of the k-th neighbour (-1: no query)
  int ties_left;  // how many points at exactly r2 belong to the k nearest (more than TIES: the first TIES by index)
  int found;
  double s1[3];   // sum (p - q)
  double s2[6];   // sum (p - q)(p - q)^T  xx,xy,xz,yy,yz,zz
  float qx0, qy0, qz0;
  int tie_orig[TIES], tie_pos[TIES];  // ascending by original index
  int min_orig;    // TIES > 1: only points at r2 with an original index >= min_orig are candidates
  bool ties_only;  // TIES > 1: a further walk — the points closer than r2 have been summed already
  __device__ __forceinline__ void init(float r2_, int ties, float qx, float qy, float qz) {
    r2 = r2_, ties_left = ties, found = 0, min_orig = 0, ties_only = false;
    s1[0] = s1[1] = s1[2] = 0.0;
#pragma unroll
    for (int j = 0; j < 6; j++) s2[j] = 0.0;
    qx0 = qx, qy0 = qy, qz0 = qz;
#pragma unroll
    for (int t = 0; t < TIES; t++) tie_orig[t] = 0x7fffffff, tie_pos[t] = -1;
  }
  // after finish(): arm another walk for the next TIES equidistant points (r2_ < 0: this lane is complete)
  __device__ __forceinline__ void rearm_ties(float r2_, int ties) {
    min_orig = (int)((unsigned)tie_orig[TIES - 1] + 1u);  // unsigned: a lane that is complete still holds the INT_MAX sentinel
    r2 = r2_, ties_left = ties, ties_only = true;
#pragma unroll
    for (int t = 0; t < TIES; t++) tie_orig[t] = 0x7fffffff, tie_pos[t] = -1;
  }
  __device__ __forceinline__ bool wants(float box_d2) const { return box_d2 <= r2; }
  __device__ __forceinline__ void add(float px, float py, float pz) {
    const double dx = (double)px - (double)qx0, dy = (double)py - (double)qy0, dz = (double)pz - (double)qz0;
    s1[0] += dx, s1[1] += dy, s1[2] += dz;
    s2[0] += dx * dx, s2[1] += dx * dy, s2[2] += dx * dz, s2[3] += dy * dy, s2[4] += dy * dz, s2[5] += dz * dz;
    found++;
  }
  __device__ __forceinline__ void take(float dd, float px, float py, float pz, int orig, int pos) {
    if (TIES == 1) {
      if (dd < r2) add(px, py, pz);
      const bool sw = (dd == r2) & (orig < tie_orig[0]);
      tie_orig[0] = sw ? orig : tie_orig[0], tie_pos[0] = sw ? pos : tie_pos[0];
      return;
    }
    if (!ties_only && dd < r2) add(px, py, pz);
    if (__ballot(dd == r2) != 0ull) {
      if (dd == r2 && orig >= min_orig) {
        int o = orig, p = pos;
#pragma unroll
        for (int t = 0; t < TIES; t++) {
          const bool sw = o < tie_orig[t];
          const int to = tie_orig[t], tp = tie_pos[t];
          tie_orig[t] = sw ? o : to, tie_pos[t] = sw ? p : tp;
          o = sw ? to : o, p = sw ? tp : p;
        }
      }
    }
  }
  __device__ __forceinline__ void visit_leaf(const hgs_f16v& xy, const hgs_f16v& zw, hgs_f2 qx, hgs_f2 qy, hgs_f2 qz, int base) {
#pragma unroll
    for (int l = 0; l < 8; l += 2) {
      const hgs_f2 dd = pk_dist2(qx, qy, qz, hgs_f2{xy[l], xy[l + 1]}, hgs_f2{xy[8 + l], xy[9 + l]}, hgs_f2{zw[l], zw[l + 1]});
      take(dd.x, xy[l], xy[8 + l], zw[l], __float_as_int(zw[8 + l]), base + l);
      take(dd.y, xy[l + 1], xy[9 + l], zw[l + 1], __float_as_int(zw[9 + l]), base + l + 1);
    }
  }
  // after the walk: the remembered equidistant points, lowest original index first
  __device__ __forceinline__ void finish(const float4* pts) {
#pragma unroll
    for (int t = 0; t < TIES; t++)
      if (t < ties_left && tie_pos[t] >= 0) {
        const float4 p = pts[tie_pos[t]];
        add(p.x, p.y, p.z);
      }
  }
};

// ---- radius count: how many points lie strictly within r2 of the query, counting stops once `need` are found ------
struct RadiusCountLane {
  float r2;   // -1: no query
  int cnt, need;
  __device__ __forceinline__ bool wants(float box_d2) const { return box_d2 < r2 && cnt < need; }
  __device__ __forceinline__ void visit_leaf(const hgs_f16v& xy, const hgs_f16v& zw, hgs_f2 qx, hgs_f2 qy, hgs_f2 qz, int base) {
#pragma unroll
    for (int l = 0; l < 8; l += 2) {
      const hgs_f2 dd = pk_dist2(qx, qy, qz, hgs_f2{xy[l], xy[l + 1]}, hgs_f2{xy[8 + l], xy[9 + l]}, hgs_f2{zw[l], zw[l + 1]});
      cnt += (dd.x < r2 ? 1 : 0) + (dd.y < r2 ? 1 : 0);
    }
  }
};

}  // namespace hgs
