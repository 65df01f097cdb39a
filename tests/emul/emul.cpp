// tests/emul/emul.cpp — TEST-ONLY host harness.  Compiles the HGS_HD per-item functions of
// hdl_graph_slam_amd/csrc/{hgs_math,hgs_bvh,hgs_gicp,hgs_ndt}.h with g++ and drives them serially in the same
// order the HIP kernels of hgs_kernels.hip do (one "thread" per point, 256-point tiles, tile-ordered sums).
// Purpose: validate tree traversal, per-point arithmetic and the LM / Newton state machines against the oracle on
// the CPU-only build box before the kernels run on an MI355X.  It is never part of the product library and is
// not a fallback: hdl_graph_slam_amd/ does not reference it.
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <unordered_map>
#include <vector>

#include "../../hdl_graph_slam_amd/csrc/hgs_gicp.h"
#include "../../hdl_graph_slam_amd/csrc/hgs_ndt.h"
#include "../../hdl_graph_slam_amd/csrc/hgs_vgicp.h"
#include "../../include/hgs_registration.h"

using namespace hgs;

struct ECloud {
  std::vector<Float4> raw, pts, nodes, cov;
  std::vector<int> corr;
  int n_input = 0, nvalid = 0, P = 1;
  float bbmin[3], bbmax[3];
  BvhView view() const { return BvhView{nodes.data(), pts.data(), nullptr, P, nvalid}; }
};

static bool finite3(const Float4& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); }

static void build_cloud(ECloud& c, const float* xyz, int n, size_t stride_floats) {
  c.n_input = n;
  c.raw.resize(n);
  for (int i = 0; i < n; i++) c.raw[i] = Float4{xyz[i * stride_floats], xyz[i * stride_floats + 1], xyz[i * stride_floats + 2], int_as_float_hd(i)};
  for (int k = 0; k < 3; k++) c.bbmin[k] = INFINITY, c.bbmax[k] = -INFINITY;
  c.nvalid = 0;
  for (auto& p : c.raw)
    if (finite3(p)) {
      c.nvalid++;
      c.bbmin[0] = fminf(c.bbmin[0], p.x), c.bbmin[1] = fminf(c.bbmin[1], p.y), c.bbmin[2] = fminf(c.bbmin[2], p.z);
      c.bbmax[0] = fmaxf(c.bbmax[0], p.x), c.bbmax[1] = fmaxf(c.bbmax[1], p.y), c.bbmax[2] = fmaxf(c.bbmax[2], p.z);
    }
  c.P = 1;
  while (c.P * kLeaf < n) c.P <<= 1;
  // k_hilbert_keys
  std::vector<uint64_t> keys(n);
  std::vector<uint32_t> vals(n);
  const float ext = fmaxf(c.bbmax[0] - c.bbmin[0], fmaxf(c.bbmax[1] - c.bbmin[1], c.bbmax[2] - c.bbmin[2]));
  const float sc = ext > 0.f ? 65535.f / ext : 0.f;
  for (int i = 0; i < n; i++) {
    uint64_t code = 0xffffffffffffull;
    const Float4 p = c.raw[i];
    if (finite3(p)) {
      const uint32_t qx = std::min(65535u, (uint32_t)((p.x - c.bbmin[0]) * sc));
      const uint32_t qy = std::min(65535u, (uint32_t)((p.y - c.bbmin[1]) * sc));
      const uint32_t qz = std::min(65535u, (uint32_t)((p.z - c.bbmin[2]) * sc));
      code = hilbert48(qx, qy, qz);
      if (code == 0xffffffffffffull) code--;
    }
    keys[i] = code;
    vals[i] = i;
  }
  std::stable_sort(vals.begin(), vals.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
  // k_gather_sorted
  c.pts.assign((size_t)c.P * kLeaf, Float4{INFINITY, INFINITY, INFINITY, int_as_float_hd(-1)});
  for (int i = 0; i < c.nvalid; i++) c.pts[i] = c.raw[vals[i]];
  // k_build_bottom / k_build_top
  c.nodes.assign((size_t)4 * c.P + 8, Float4{0, 0, 0, 0});
  for (int leaf = 0; leaf < c.P; leaf++) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int l = 0; l < kLeaf; l++) {
      const int idx = leaf * kLeaf + l;
      if (idx < c.nvalid) {
        const Float4 p = c.pts[idx];
        mn[0] = fminf(mn[0], p.x), mn[1] = fminf(mn[1], p.y), mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x), mx[1] = fmaxf(mx[1], p.y), mx[2] = fmaxf(mx[2], p.z);
      }
    }
    bvh_store_box(c.nodes.data(), (uint32_t)(c.P + leaf), mn, mx);
  }
  for (int id = c.P - 1; id >= 1; id--) {
    float amn[3], amx[3], bmn[3], bmx[3];
    bvh_load_box(c.nodes.data(), 2 * (uint32_t)id, amn, amx);
    bvh_load_box(c.nodes.data(), 2 * (uint32_t)id + 1, bmn, bmx);
    const float mn[3] = {fminf(amn[0], bmn[0]), fminf(amn[1], bmn[1]), fminf(amn[2], bmn[2])};
    const float mx[3] = {fmaxf(amx[0], bmx[0]), fmaxf(amx[1], bmx[1]), fmaxf(amx[2], bmx[2])};
    bvh_store_box(c.nodes.data(), (uint32_t)id, mn, mx);
  }
  c.cov.clear();
  c.corr.assign((size_t)c.P * kLeaf, -1);
}

// Models k_knn_cov's two passes per query: (1) r2 = distance of the k-th neighbour and how many points at exactly r2
// belong to the k nearest; (2) every point closer than r2 plus that many points at r2, lowest original index first.
template <int KMAX>
static void knn_cov_t(ECloud& c, int k, int reg_method) {
  c.cov.assign((size_t)2 * c.P * kLeaf, Float4{0, 0, 0, 0});
  const BvhView tv = c.view();
  std::vector<std::pair<int, int>> tied;  // (original index, position)
  for (int i = 0; i < c.nvalid; i++) {
    const Float4 qp = c.pts[i];
    const F3 q = {qp.x, qp.y, qp.z};
    KnnList<KMAX> list;
    bvh_knn<KMAX>(tv, q, k, list);
    const float r2 = list.worst();
    int n_lt = 0, live = 0;
    for (int j = 0; j < KMAX; j++) live += list.d[j] >= 0.f ? 1 : 0, n_lt += (list.d[j] >= 0.f && list.d[j] < r2) ? 1 : 0;
    const int ties = live - n_lt;
    double s1[3] = {0, 0, 0};
    Sym3 s2 = {0, 0, 0, 0, 0, 0};
    int found = 0;
    auto add = [&](const Float4& p) {
      const double dx = (double)p.x - (double)q.x, dy = (double)p.y - (double)q.y, dz = (double)p.z - (double)q.z;
      s1[0] += dx, s1[1] += dy, s1[2] += dz;
      s2.xx += dx * dx, s2.xy += dx * dy, s2.xz += dx * dz, s2.yy += dy * dy, s2.yz += dy * dz, s2.zz += dz * dz;
      found++;
    };
    tied.clear();
    std::vector<uint32_t> stack{1u};
    while (!stack.empty()) {
      const uint32_t node = stack.back();
      stack.pop_back();
      if (!(bvh_box_dist2(tv.nodes, node, q) <= r2)) continue;
      if ((int)node < tv.P) {
        stack.push_back(2 * node + 1), stack.push_back(2 * node);
        continue;
      }
      const int base = ((int)node - tv.P) * kLeaf;
      for (int l = 0; l < kLeaf; l++) {
        const Float4 p = tv.pts[base + l];
        const float d = dist2f(q, p.x, p.y, p.z);
        if (d < r2) add(p);
        else if (d == r2) tied.emplace_back(float_as_int_hd(p.w), base + l);
      }
    }
    std::sort(tied.begin(), tied.end());
    for (int t = 0; t < ties && t < (int)tied.size(); t++) add(tv.pts[tied[t].second]);
    const Sym3 cv = gicp_regularized_cov(s1, s2, found, k, reg_method);
    c.cov[2 * i] = Float4{(float)cv.xx, (float)cv.xy, (float)cv.xz, (float)cv.yy};
    c.cov[2 * i + 1] = Float4{(float)cv.yz, (float)cv.zz, 0, 0};
  }
}
static void knn_cov(ECloud& c, int k, int reg_method) {
  if (k <= 8) knn_cov_t<8>(c, k, reg_method);
  else if (k <= 16) knn_cov_t<16>(c, k, reg_method);
  else if (k <= 20) knn_cov_t<20>(c, k, reg_method);
  else knn_cov_t<32>(c, k, reg_method);
}
static Sym3 load_cov(const std::vector<Float4>& cov, int i) {
  const Float4 a = cov[2 * i], b = cov[2 * i + 1];
  return sym3_from_floats(a.x, a.y, a.z, a.w, b.x, b.y);
}

static GicpConsts gicp_consts(const hgs_params& p) {
  GicpConsts c;
  c.max_corr2 = p.max_correspondence_distance * p.max_correspondence_distance;
  c.search_bound2 = c.max_corr2 >= (double)FLT_MAX ? FLT_MAX : nextafterf((float)c.max_corr2, FLT_MAX);
  c.rotation_eps = p.rotation_epsilon, c.translation_eps = p.transformation_epsilon;
  c.lm_init_lambda_factor = p.lm_init_lambda_factor, c.lm_max_iterations = p.lm_max_iterations, c.max_iterations = p.max_iterations;
  c.k_correspondences = p.correspondence_randomness;
  return c;
}

// k_gicp_linearize over all tiles, partials summed in tile order (k_gicp_solve)
static void gicp_linearize(ECloud& src, const ECloud& tgt, const Pose& T, const GicpConsts& c, double* acc_out) {
  const int ntiles = (src.nvalid + 255) / 256;
  std::vector<double> total(kAcc, 0.0);
  const BvhView tv = tgt.view();
  float Tf[12];
  pose_to_float(T, Tf);
  for (int tile = 0; tile < ntiles; tile++) {
    double part[kAcc] = {0};
    for (int t = 0; t < 256; t++) {
      const int i = tile * 256 + t;
      if (i >= src.nvalid) break;
      double acc[kAcc] = {0};
      const Float4 a = src.pts[i];
      const F3 q = transform_point_f(Tf, a.x, a.y, a.z);
      float d2;
      int orig;
      int j = bvh_nn1(tv, q, c.search_bound2, &d2, &orig);
      if (j >= 0 && !((double)d2 < c.max_corr2)) j = -1;
      src.corr[i] = j;
      if (j >= 0) {
        const double R[9] = {T.m[0], T.m[1], T.m[2], T.m[4], T.m[5], T.m[6], T.m[8], T.m[9], T.m[10]};
        const Sym3 M = gicp_mahalanobis(R, load_cov(src.cov, i), load_cov(tgt.cov, j));
        const Float4 bp = tgt.pts[j];
        acc[27] = gicp_point_terms<true>(T, M, a.x, a.y, a.z, bp.x, bp.y, bp.z, acc);
      }
      for (int k = 0; k < kAcc; k++) part[k] += acc[k];
    }
    for (int k = 0; k < kAcc; k++) total[k] += part[k];
  }
  for (int k = 0; k < kAcc; k++) acc_out[k] = total[k];
}
static double gicp_error(const ECloud& src, const ECloud& tgt, const Pose& T0, const Pose& Ti) {
  double total = 0;
  for (int i = 0; i < src.nvalid; i++) {
    const int j = src.corr[i];
    if (j < 0) continue;
    const double R[9] = {T0.m[0], T0.m[1], T0.m[2], T0.m[4], T0.m[5], T0.m[6], T0.m[8], T0.m[9], T0.m[10]};
    const Sym3 M = gicp_mahalanobis(R, load_cov(src.cov, i), load_cov(tgt.cov, j));
    const Float4 a = src.pts[i], bp = tgt.pts[j];
    total += gicp_point_terms<false>(Ti, M, a.x, a.y, a.z, bp.x, bp.y, bp.z, nullptr);
  }
  return total;
}

// ---- NDT target
struct ENdt {
  std::vector<int> hash_keys, hash_vals;
  std::vector<NdtCellRec> cells;
  NdtGrid grid;
};
static void ndt_build(ENdt& e, const ECloud& c, double resolution, int min_points) {
  const float inv_leaf = 1.0f / (float)resolution;
  NdtGrid& g = e.grid;
  g.inv_leaf = inv_leaf;
  long long div[3];
  for (int k = 0; k < 3; k++) {
    g.min_b[k] = (int)floorf(c.bbmin[k] * inv_leaf);
    g.max_b[k] = (int)floorf(c.bbmax[k] * inv_leaf);
    div[k] = (long long)g.max_b[k] - g.min_b[k] + 1;
  }
  g.div_mul[0] = 1, g.div_mul[1] = (int)div[0], g.div_mul[2] = (int)(div[0] * div[1]);
  std::vector<std::pair<uint32_t, int>> kv;
  for (int i = 0; i < c.n_input; i++) {
    const Float4 p = c.raw[i];
    if (!finite3(p)) continue;
    const int cx = (int)floorf(p.x * inv_leaf) - g.min_b[0], cy = (int)floorf(p.y * inv_leaf) - g.min_b[1], cz = (int)floorf(p.z * inv_leaf) - g.min_b[2];
    kv.push_back({(uint32_t)(cx * g.div_mul[0] + cy * g.div_mul[1] + cz * g.div_mul[2]), i});
  }
  std::stable_sort(kv.begin(), kv.end(), [](auto& a, auto& b) { return a.first < b.first; });
  int cap = 64;
  while (cap < 4 * (int)(c.n_input / std::max(1, min_points) + 1)) cap <<= 1;
  e.hash_keys.assign(cap, -1);
  e.hash_vals.assign(cap, -1);
  e.cells.clear();
  g.hash_mask = cap - 1;
  for (size_t i = 0; i < kv.size();) {
    size_t j = i;
    double sum[3] = {0, 0, 0};
    Sym3 sq = {0, 0, 0, 0, 0, 0};
    int n = 0;
    for (; j < kv.size() && kv[j].first == kv[i].first; j++) {
      const Float4 p = c.raw[kv[j].second];
      const double x = p.x, y = p.y, z = p.z;
      sum[0] += x, sum[1] += y, sum[2] += z;
      sq.xx += x * x, sq.xy += x * y, sq.xz += x * z, sq.yy += y * y, sq.yz += y * z, sq.zz += z * z;
      n++;
    }
    double mean[3];
    Sym3 icov;
    if (ndt_finalize_cell(n, sum, sq, min_points, mean, &icov)) {
      NdtCellRec rec;
      rec.v0 = Float4{(float)icov.xx, (float)icov.xy, (float)icov.xz, (float)icov.yy};
      rec.v1 = Float4{(float)icov.yz, (float)icov.zz, (float)n, int_as_float_hd((int)kv[i].first)};
      rec.mean[0] = mean[0], rec.mean[1] = mean[1], rec.mean[2] = mean[2], rec.pad = 0.0;
      const int slot_c = (int)e.cells.size();
      e.cells.push_back(rec);
      uint32_t slot = (ndt_hash((int)kv[i].first) >> 7) & (uint32_t)g.hash_mask;
      while (e.hash_keys[slot] != -1) slot = (slot + 1) & (uint32_t)g.hash_mask;
      e.hash_keys[slot] = (int)kv[i].first;
      e.hash_vals[slot] = slot_c;
    }
    i = j;
  }
  g.hash_keys = e.hash_keys.data(), g.hash_vals = e.hash_vals.data(), g.cells = e.cells.data();
}
static void ndt_derivatives(const ECloud& src, const ENdt& e, const NdtAngles& ang, const NdtConsts& c, double* acc_out) {
  const int ntiles = (src.n_input + 255) / 256;
  std::vector<double> total(kAccNdt, 0.0);
  const NdtGrid& g = e.grid;
  for (int tile = 0; tile < ntiles; tile++) {
    double part[kAccNdt] = {0};
    for (int t = 0; t < 256; t++) {
      const int i = tile * 256 + t;
      if (i >= src.n_input) break;
      const Float4 x = src.raw[i];
      if (!finite3(x)) continue;
      double acc[kAccNdt] = {0};
      const F3 xt = transform_point_f(ang.T, x.x, x.y, x.z);
      const int cx = (int)floorf(xt.x * g.inv_leaf), cy = (int)floorf(xt.y * g.inv_leaf), cz = (int)floorf(xt.z * g.inv_leaf);
      NdtPointDeriv pd;
      ndt_point_derivatives(ang, x.x, x.y, x.z, pd);
      const int nn = ndt_num_offsets(c.search);
      for (int o = 0; o < nn; o++) {
        int ox, oy, oz;
        ndt_offset(c.search, o, &ox, &oy, &oz);
        const int ci = ndt_lookup(g, cx + ox, cy + oy, cz + oz);
        if (ci < 0) continue;
        const NdtCellRec rec = g.cells[ci];
        if (!ndt_cell_in_reach(c, xt, rec.mean)) continue;
        const float icov[6] = {rec.v0.x, rec.v0.y, rec.v0.z, rec.v0.w, rec.v1.x, rec.v1.y};
        ndt_cell_terms(c, pd, (float)((double)xt.x - rec.mean[0]), (float)((double)xt.y - rec.mean[1]), (float)((double)xt.z - rec.mean[2]), icov, acc);
      }
      for (int k = 0; k < kAccNdt; k++) part[k] += acc[k];
    }
    for (int k = 0; k < kAccNdt; k++) total[k] += part[k];
  }
  for (int k = 0; k < kAccNdt; k++) acc_out[k] = total[k];
}
static NdtConsts ndt_consts(const hgs_params& p) {
  NdtConsts c;
  const double c1 = 10.0 * (1 - p.ndt_outlier_ratio), c2 = p.ndt_outlier_ratio / std::pow(p.resolution, 3), d3 = -std::log(c2);
  c.gauss_d1 = -std::log(c1 + c2) - d3;
  c.gauss_d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / c.gauss_d1);
  c.step_size = p.ndt_step_size, c.trans_eps = p.transformation_epsilon, c.max_iterations = p.max_iterations;
  c.search = p.neighbor_search == HGS_DIRECT1 ? 1 : (p.neighbor_search == HGS_KDTREE ? 0 : 2);
  c.upstream_hd1_sign = p.ndt_upstream_hd1_sign, c.pad = 0;
  c.kdtree_radius2 = (float)(p.resolution * p.resolution), c.line_search = p.ndt_line_search ? 1 : 0;
  return c;
}

// ---- VGICP target (k_vgicp_grid_params / k_vgicp_cell_keys / sort / k_vgicp_build_cells)
static void vgicp_build(ENdt& e, const ECloud& c, double resolution) {
  NdtGrid& g = e.grid;
  g.inv_leaf = 0.f;
  long long div[3];
  for (int k = 0; k < 3; k++) {
    g.min_b[k] = vgicp_coord((double)c.bbmin[k], resolution);
    g.max_b[k] = vgicp_coord((double)c.bbmax[k], resolution);
    div[k] = (long long)g.max_b[k] - g.min_b[k] + 1;
  }
  g.div_mul[0] = 1, g.div_mul[1] = (int)div[0], g.div_mul[2] = (int)(div[0] * div[1]);
  std::vector<std::pair<uint32_t, int>> kv;
  for (int i = 0; i < c.nvalid; i++) {
    const Float4 p = c.pts[i];
    const int cx = vgicp_coord((double)p.x, resolution) - g.min_b[0], cy = vgicp_coord((double)p.y, resolution) - g.min_b[1],
              cz = vgicp_coord((double)p.z, resolution) - g.min_b[2];
    kv.push_back({(uint32_t)(cx * g.div_mul[0] + cy * g.div_mul[1] + cz * g.div_mul[2]), i});
  }
  std::stable_sort(kv.begin(), kv.end(), [](auto& a, auto& b) { return a.first < b.first; });
  int cap = 64;
  while (cap < 2 * (c.n_input + 1)) cap <<= 1;
  e.hash_keys.assign(cap, -1);
  e.hash_vals.assign(cap, -1);
  e.cells.clear();
  g.hash_mask = cap - 1;
  for (size_t i = 0; i < kv.size();) {
    size_t j = i;
    double sum[3] = {0, 0, 0};
    Sym3 sc = {0, 0, 0, 0, 0, 0};
    int n = 0;
    for (; j < kv.size() && kv[j].first == kv[i].first; j++) {
      const int pi = kv[j].second;
      const Float4 p = c.pts[pi], c0 = c.cov[2 * pi], c1 = c.cov[2 * pi + 1];
      sum[0] += (double)p.x, sum[1] += (double)p.y, sum[2] += (double)p.z;
      sc.xx += (double)c0.x, sc.xy += (double)c0.y, sc.xz += (double)c0.z, sc.yy += (double)c0.w, sc.yz += (double)c1.x, sc.zz += (double)c1.y;
      n++;
    }
    NdtCellRec rec;
    vgicp_finalize_voxel(n, sum, sc, (int)kv[i].first, &rec);
    const int slot_c = (int)e.cells.size();
    e.cells.push_back(rec);
    uint32_t slot = (ndt_hash((int)kv[i].first) >> 7) & (uint32_t)g.hash_mask;
    while (e.hash_keys[slot] != -1) slot = (slot + 1) & (uint32_t)g.hash_mask;
    e.hash_keys[slot] = (int)kv[i].first;
    e.hash_vals[slot] = slot_c;
    i = j;
  }
  g.hash_keys = e.hash_keys.data(), g.hash_vals = e.hash_vals.data(), g.cells = e.cells.data();
}
static VgicpConsts vgicp_consts(const hgs_params& p) {
  VgicpConsts c;
  c.resolution = p.resolution;
  c.search = p.neighbor_search == HGS_DIRECT27 ? 3 : (p.neighbor_search == HGS_DIRECT7 ? 2 : 1);
  c.pad = 0;
  return c;
}
// k_vgicp_linearize over all tiles, partials summed in tile order
static void vgicp_linearize(ECloud& src, const ENdt& e, const Pose& T, const VgicpConsts& c, double* acc_out) {
  const int ntiles = (src.nvalid + 255) / 256;
  std::vector<double> total(kAcc, 0.0);
  for (int tile = 0; tile < ntiles; tile++) {
    double part[kAcc] = {0};
    for (int t = 0; t < 256; t++) {
      const int i = tile * 256 + t;
      if (i >= src.nvalid) break;
      double acc[kAcc] = {0};
      const Float4 a = src.pts[i];
      int hits;
      acc[27] = vgicp_point_terms<true>(e.grid, c, T, T, load_cov(src.cov, i), a.x, a.y, a.z, acc, &hits);
      src.corr[i] = hits;
      for (int k = 0; k < kAcc; k++) part[k] += acc[k];
    }
    for (int k = 0; k < kAcc; k++) total[k] += part[k];
  }
  for (int k = 0; k < kAcc; k++) acc_out[k] = total[k];
}
static double vgicp_error(const ECloud& src, const ENdt& e, const VgicpConsts& c, const Pose& T0, const Pose& Ti) {
  double total = 0;
  for (int i = 0; i < src.nvalid; i++) {
    if (src.corr[i] <= 0) continue;
    const Float4 a = src.pts[i];
    total += vgicp_point_terms<false>(e.grid, c, T0, Ti, load_cov(src.cov, i), a.x, a.y, a.z, nullptr, nullptr);
  }
  return total;
}

struct EmulHandle {
  hgs_params prm;
  ECloud src, tgt;
  ENdt ndt;
  ENdt vg;
  bool have_src = false, have_tgt = false;
};

extern "C" {

EmulHandle* emul_create(const hgs_params* p) {
  auto* h = new EmulHandle();
  h->prm = *p;
  return h;
}
void emul_destroy(EmulHandle* h) { delete h; }

static void prep(EmulHandle* h, ECloud& c, const void* pts, size_t n, size_t stride, bool is_target) {
  build_cloud(c, (const float*)pts, (int)n, stride / 4);
  if (h->prm.method == HGS_FAST_GICP || h->prm.method == HGS_FAST_VGICP) knn_cov(c, h->prm.correspondence_randomness, h->prm.regularization_method);
  if (h->prm.method == HGS_FAST_VGICP && is_target) vgicp_build(h->vg, c, h->prm.resolution);
  if (h->prm.method == HGS_NDT_OMP && is_target) ndt_build(h->ndt, c, h->prm.resolution, h->prm.ndt_min_points_per_voxel);
}
int emul_set_target(EmulHandle* h, const void* pts, size_t n, size_t stride) {
  prep(h, h->tgt, pts, n, stride, true);
  h->have_tgt = true;
  return 0;
}
int emul_set_source(EmulHandle* h, const void* pts, size_t n, size_t stride) {
  prep(h, h->src, pts, n, stride, false);
  h->have_src = true;
  return 0;
}

int emul_align(EmulHandle* h, const float* guess, hgs_result* out) {
  if (h->prm.method == HGS_FAST_GICP || h->prm.method == HGS_FAST_VGICP) {
    const bool voxel = h->prm.method == HGS_FAST_VGICP;
    const GicpConsts c = gicp_consts(h->prm);
    const VgicpConsts vc = vgicp_consts(h->prm);
    GicpState st;
    gicp_state_init(st, guess);
    long rounds = 0;
    while (st.phase != GICP_DONE && rounds < 100000) {
      if (st.phase == GICP_LINEARIZE) {
        double acc[kAcc];
        if (voxel) vgicp_linearize(h->src, h->vg, st.x0, vc, acc);
        else gicp_linearize(h->src, h->tgt, st.x0, c, acc);
        gicp_after_linearize(st, acc, c);
      }
      if (st.phase == GICP_TRY) {
        const double yi = voxel ? vgicp_error(h->src, h->vg, vc, st.x0, st.xi) : gicp_error(h->src, h->tgt, st.x0, st.xi);
        gicp_after_error(st, yi, c);
      }
      rounds++;
    }
    pose_to_colmajor_f(st.x0, out->final_transformation);
    out->converged = st.converged, out->iterations = st.iterations, out->lm_tries = st.lm_tries_total, out->error = st.y0;
  } else {
    const NdtConsts c = ndt_consts(h->prm);
    NdtState st;
    NdtAngles ang;
    ndt_state_init(st, guess);
    ndt_angle_tables(st.p, c.upstream_hd1_sign, ang);
    long rounds = 0;
    while (st.phase != NDT_DONE && rounds < 100000) {
      double acc[kAccNdt];
      ndt_derivatives(h->src, h->ndt, ang, c, acc);
      ndt_after_derivatives(st, acc, c);
      if (getenv("HGS_EMUL_TRACE")) printf("emul it=%d p=%.6f %.6f %.6f %.6f %.6f %.6f score=%.6f a_t=%.6f\n", st.iterations, st.p[0], st.p[1], st.p[2], st.p[3], st.p[4], st.p[5], st.score, st.a_t);
      if (st.phase != NDT_DONE) ndt_angle_tables(st.p, c.upstream_hd1_sign, ang);
      rounds++;
    }
    pose_to_colmajor_f(st.final_T, out->final_transformation);
    out->converged = st.converged, out->iterations = st.iterations, out->lm_tries = st.passes;
    out->error = h->src.nvalid > 0 ? st.score / h->src.nvalid : 0.0;
  }
  out->fitness_score = std::numeric_limits<double>::quiet_NaN();
  out->num_inliers = 0, out->candidate_id = 0, out->reserved = 0;
  return 0;
}

int emul_fitness(EmulHandle* h, const float* T16, double max_range, double* score, uint32_t* ninl) {
  float Tf[12];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) Tf[r * 4 + c] = T16[c * 4 + r];
  const int ntiles = (h->src.nvalid + 255) / 256;
  double s = 0, cnt = 0;
  const BvhView tv = h->tgt.view();
  for (int tile = 0; tile < ntiles; tile++) {
    double ps = 0, pc = 0;
    for (int t = 0; t < 256 && tile * 256 + t < h->src.nvalid; t++) {
      const Float4 a = h->src.pts[tile * 256 + t];
      const F3 q = transform_point_f(Tf, a.x, a.y, a.z);
      float d2;
      int orig;
      const int j = bvh_nn1(tv, q, FLT_MAX, &d2, &orig);
      if (j >= 0 && (double)d2 <= max_range) ps += (double)d2, pc += 1.0;
    }
    s += ps, cnt += pc;
  }
  *ninl = (uint32_t)(cnt + 0.5);
  *score = cnt > 0 ? s / cnt : std::numeric_limits<double>::max();
  return 0;
}

int emul_nn_target(EmulHandle* h, const float* q, size_t nq, size_t stride, int32_t* idx, float* d2) {
  const BvhView tv = h->tgt.view();
  for (size_t i = 0; i < nq; i++) {
    const float* f = (const float*)((const char*)q + i * stride);
    int orig;
    const int j = bvh_nn1(tv, F3{f[0], f[1], f[2]}, FLT_MAX, &d2[i], &orig);
    idx[i] = j >= 0 ? orig : -1;
  }
  return 0;
}

// covariances of the target in ORIGINAL order: out[n][6]
int emul_target_covariances(EmulHandle* h, float* out6) {
  for (int i = 0; i < h->tgt.n_input * 6; i++) out6[i] = 0.f;
  for (int i = 0; i < h->tgt.nvalid; i++) {
    const int o = float_as_int_hd(h->tgt.pts[i].w);
    const Float4 a = h->tgt.cov[2 * i], b = h->tgt.cov[2 * i + 1];
    float* p = out6 + (size_t)o * 6;
    p[0] = a.x, p[1] = a.y, p[2] = a.z, p[3] = a.w, p[4] = b.x, p[5] = b.y;
  }
  return 0;
}

int emul_gicp_linearize(EmulHandle* h, const double* T12, double* H36, double* b6, double* err, int32_t* corr_orig) {
  Pose T;
  for (int i = 0; i < 12; i++) T.m[i] = T12[i];
  double acc[kAcc];
  const bool voxel = h->prm.method == HGS_FAST_VGICP;
  if (voxel) vgicp_linearize(h->src, h->vg, T, vgicp_consts(h->prm), acc);
  else gicp_linearize(h->src, h->tgt, T, gicp_consts(h->prm), acc);
  int k = 0;
  for (int r = 0; r < 6; r++)
    for (int c = r; c < 6; c++) H36[r * 6 + c] = H36[c * 6 + r] = acc[k++];
  for (int i = 0; i < 6; i++) b6[i] = acc[21 + i];
  *err = acc[27];
  if (corr_orig) {
    for (int i = 0; i < h->src.n_input; i++) corr_orig[i] = voxel ? 0 : -1;
    for (int i = 0; i < h->src.nvalid; i++) {
      const int j = h->src.corr[i];
      corr_orig[float_as_int_hd(h->src.pts[i].w)] = voxel ? j : (j >= 0 ? float_as_int_hd(h->tgt.pts[j].w) : -1);
    }
  }
  return 0;
}

int emul_ndt_cells(EmulHandle* h, int cap, int32_t* key, double* mean3, float* icov6, int32_t* npts) {
  const int n = (int)h->ndt.cells.size();
  for (int i = 0; i < n && i < cap; i++) {
    const NdtCellRec& r = h->ndt.cells[i];
    key[i] = float_as_int_hd(r.v1.w);
    mean3[3 * i] = r.mean[0], mean3[3 * i + 1] = r.mean[1], mean3[3 * i + 2] = r.mean[2];
    float* o = icov6 + 6 * i;
    o[0] = r.v0.x, o[1] = r.v0.y, o[2] = r.v0.z, o[3] = r.v0.w, o[4] = r.v1.x, o[5] = r.v1.y;
    npts[i] = (int)r.v1.z;
  }
  return n;
}
int emul_ndt_grid(EmulHandle* h, int32_t* min_b, int32_t* div_mul) {
  for (int k = 0; k < 3; k++) min_b[k] = h->ndt.grid.min_b[k], div_mul[k] = h->ndt.grid.div_mul[k];
  return 0;
}
int emul_ndt_derivatives(EmulHandle* h, const double* p6, double* score, double* g6, double* H36) {
  const NdtConsts c = ndt_consts(h->prm);
  NdtAngles ang;
  ndt_angle_tables(p6, c.upstream_hd1_sign, ang);
  double acc[kAccNdt];
  ndt_derivatives(h->src, h->ndt, ang, c, acc);
  for (int i = 0; i < 36; i++) H36[i] = acc[i];
  for (int i = 0; i < 6; i++) g6[i] = acc[36 + i];
  *score = acc[42];
  return 0;
}

// tree statistics for tuning: average nodes + leaves visited by a 1-NN query is not measurable through the
// product path; this returns the tree's sorted point order so tests can check it is a permutation.
int emul_sorted_order(EmulHandle* h, int target, int32_t* out) {
  const ECloud& c = target ? h->tgt : h->src;
  for (int i = 0; i < c.nvalid; i++) out[i] = float_as_int_hd(c.pts[i].w);
  return c.nvalid;
}

}  // extern "C"

// ---- host simulation of the wave-cooperative 4-ary walk of hgs_wave_bvh.h (64 lanes in lock-step) -----------------
// Mirrors wave_walk step by step: used to (a) check on the CPU that the packet search returns exactly what the
// per-lane search returns and (b) count group evaluations / leaf visits per wave when tuning the traversal.
struct SimStats {
  long long groups = 0, leaves = 0, waves = 0, mismatches = 0, inserts = 0, retests = 0;
};
static int g_sim_variant = 0;
static int g_sim_order = 0;  // 0: majority vote for the nearest wanted child; 1: child wanted by most lanes; 2: nearest child of the middle lane
template <class Wants, class Leaf>
static void sim_wave_walk_v1(const BvhView& t, const F3* q, int nl, Wants wants, Leaf visit_leaf, SimStats& st);
template <class Wants, class Leaf>
static void sim_wave_walk_wide(const BvhView& t, const F3* q, int nl, Wants wants, Leaf visit_leaf, SimStats& st, int S);
template <class Wants, class Leaf>
static void sim_wave_walk(const BvhView& t, const F3* q, int nl, Wants wants, Leaf visit_leaf, SimStats& st) {
  if (g_sim_variant == 1) return sim_wave_walk_v1(t, q, nl, wants, visit_leaf, st);
  if (g_sim_variant >= 2 && g_sim_variant <= 4) return sim_wave_walk_wide(t, q, nl, wants, visit_leaf, st, g_sim_variant);
  if (t.n <= 0) return;
  int k = 0;
  while ((1 << k) < t.P) k++;
  const int kodd = k & 1;
  unsigned node = 1;
  int bd = 0;
  unsigned long long pend = 0;
  for (;;) {
    unsigned base = 0, allowed = 0;
    int cd = 0, s = 2;
    bool have = false;
    if (bd < k) {
      s = (bd == 0 && kodd) ? 1 : 2;
      base = node << s, cd = bd + s, allowed = (1u << (1u << s)) - 1u, have = true;
    } else {
      st.leaves++;
      visit_leaf(((int)node - t.P) * kLeaf);
    }
    for (;;) {
      if (!have) {
        if (!pend) return;
        const int idx = (63 - __builtin_clzll(pend)) >> 2;
        allowed = (unsigned)(pend >> (4 * idx)) & 0xfu;
        pend &= ~(0xfull << (4 * idx));
        cd = 2 * idx + kodd;
        s = cd == 1 ? 1 : 2;
        base = ((node >> (bd - cd)) >> s) << s;
      }
      have = false;
      st.groups++;
      unsigned any = 0;
      int votes[4] = {0, 0, 0, 0};
      for (int l = 0; l < nl; l++) {
        float pd = INFINITY;
        int pref = -1;
        for (int c = 0; c < (1 << s); c++) {
          if (!((allowed >> c) & 1u)) continue;
          const float d = bvh_box_dist2(t.nodes, base + c, q[l]);
          if (!wants(l, d)) continue;
          any |= 1u << c;
          if (pref < 0 || d < pd) pd = d, pref = c;
        }
        if (pref >= 0) votes[pref]++;
      }
      if (!any) continue;
      int cstar = 0, vbest = -1;
      for (int c = 0; c < 4; c++)
        if (votes[c] > vbest) vbest = votes[c], cstar = c;
      pend |= (unsigned long long)(any & ~(1u << cstar)) << (4 * (cd >> 1));
      node = base + (unsigned)cstar;
      bd = cd;
      break;
    }
  }
}

// variant 1: a popped child is entered directly (its own group evaluation prunes it), and the children of a
// leaf-parent group are visited in place, re-tested against the updated bounds with the distances still in registers
template <class Wants, class Leaf>
static void sim_wave_walk_v1(const BvhView& t, const F3* q, int nl, Wants wants, Leaf visit_leaf, SimStats& st) {
  if (t.n <= 0) return;
  int k = 0;
  while ((1 << k) < t.P) k++;
  const int kodd = k & 1;
  if (k == 0) {
    st.leaves++;
    visit_leaf(0);
    return;
  }
  unsigned node = 1;
  int bd = 0;
  unsigned long long pend = 0;
  for (;;) {
    // evaluate the group below `node`
    const int s = (bd == 0 && kodd) ? 1 : 2;
    const unsigned base = node << s;
    const int cd = bd + s;
    st.groups++;
    float d[64][4];
    unsigned any = 0;
    int votes[4] = {0, 0, 0, 0};
    for (int l = 0; l < nl; l++) {
      float pd = INFINITY;
      int pref = -1;
      for (int c = 0; c < (1 << s); c++) {
        d[l][c] = bvh_box_dist2(t.nodes, base + c, q[l]);
        if (!wants(l, d[l][c])) continue;
        any |= 1u << c;
        if (g_sim_order == 1) votes[c]++;
        if (pref < 0 || d[l][c] < pd) pd = d[l][c], pref = c;
      }
      if (pref >= 0 && g_sim_order == 0) votes[pref]++;
    }
    if (g_sim_order == 2 && any) {
      const int rep = nl / 2;
      float bestd = INFINITY;
      int bc = -1;
      for (int c = 0; c < 4; c++)
        if (((any >> c) & 1u) && (bc < 0 || d[rep][c] < bestd)) bestd = d[rep][c], bc = c;
      votes[bc] = 1;
    }
    bool descended = false;
    if (any) {
      if (cd == k) {  // children are leaves: visit them in place, most-voted first
        unsigned todo = any;
        while (todo) {
          int c = -1, vb = -1;
          for (int cc = 0; cc < 4; cc++)
            if (((todo >> cc) & 1u) && votes[cc] > vb) vb = votes[cc], c = cc;
          todo &= ~(1u << c);
          bool still = false;
          for (int l = 0; l < nl; l++) still = still || wants(l, d[l][c]);
          st.retests++;
          if (!still) continue;
          st.leaves++;
          visit_leaf(((int)(base + c) - t.P) * kLeaf);
        }
      } else {
        int cstar = 0, vbest = -1;
        for (int c = 0; c < 4; c++)
          if (votes[c] > vbest) vbest = votes[c], cstar = c;
        pend |= (unsigned long long)(any & ~(1u << cstar)) << (4 * (cd >> 1));
        node = base + (unsigned)cstar;
        bd = cd;
        descended = true;
      }
    }
    if (descended) continue;
    if (!pend) return;
    const int idx = (63 - __builtin_clzll(pend)) >> 2;
    const unsigned nib = (unsigned)(pend >> (4 * idx)) & 0xfu;
    const int c = __builtin_ctz(nib);
    pend &= ~(1ull << (4 * idx + c));
    const int pcd = 2 * idx + kodd;
    const int ps = pcd == 1 ? 1 : 2;
    node = (((node >> (bd - pcd)) >> ps) << ps) + (unsigned)c;
    bd = pcd;
  }
}

// variants 2 / 3 / 4: a WIDE step — the 2^S descendants S binary levels below the node are one record, evaluated together (S = 2 is the 4-ary
// step again, without the quad fetch; S = 3 an 8-ary, S = 4 a 16-ary node).  One dependent fetch per entered node; the unchosen wanted children
// wait on a stack and are re-tested when popped (their box distance against the lanes' current bounds: one box test, no fetch — the parent's
// record is still parked).  Counts steps (= dependent fetches), box tests and leaf visits: round 5's "wider node" experiment.
static long long g_sim_box_tests = 0;
template <class Wants, class Leaf>
static void sim_wave_walk_wide(const BvhView& t, const F3* q, int nl, Wants wants, Leaf visit_leaf, SimStats& st, int S) {
  if (t.n <= 0) return;
  int k = 0;
  while ((1 << k) < t.P) k++;
  if (k == 0) {
    st.leaves++;
    visit_leaf(0);
    return;
  }
  struct Pending {
    unsigned node;
    int depth;
  };
  std::vector<Pending> stack;
  unsigned node = 1;
  int bd = 0;
  for (;;) {
    const int s = std::min(S, k - bd);
    const int nc = 1 << s;
    const unsigned base = node << s;
    const int cd = bd + s;
    st.groups++;
    g_sim_box_tests += nc;
    std::vector<std::array<float, 64>> d(nc);
    std::vector<int> votes(nc, 0);
    std::vector<char> any(nc, 0);
    for (int l = 0; l < nl; l++) {
      float pd = INFINITY;
      int pref = -1;
      for (int c = 0; c < nc; c++) {
        d[c][l] = bvh_box_dist2(t.nodes, base + c, q[l]);
        if (!wants(l, d[c][l])) continue;
        any[c] = 1;
        if (pref < 0 || d[c][l] < pd) pd = d[c][l], pref = c;
      }
      if (pref >= 0) votes[pref]++;
    }
    std::vector<int> order;
    for (int c = 0; c < nc; c++)
      if (any[c]) order.push_back(c);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return votes[a] > votes[b]; });
    bool descended = false;
    if (!order.empty()) {
      if (cd == k) {
        for (int c : order) {
          bool still = false;
          for (int l = 0; l < nl; l++) still = still || wants(l, d[c][l]);
          st.retests++;
          if (!still) continue;
          st.leaves++;
          visit_leaf(((int)(base + c) - t.P) * kLeaf);
        }
      } else {
        for (size_t i = order.size(); i-- > 1;) stack.push_back(Pending{base + (unsigned)order[i], cd});  // best of the rest on top
        node = base + (unsigned)order[0];
        bd = cd;
        descended = true;
      }
    }
    if (descended) continue;
    for (;;) {
      if (stack.empty()) return;
      const Pending p = stack.back();
      stack.pop_back();
      bool still = false;
      g_sim_box_tests++;
      st.retests++;
      for (int l = 0; l < nl && !still; l++) still = wants(l, bvh_box_dist2(t.nodes, p.node, q[l]));
      if (!still) continue;
      node = p.node, bd = p.depth;
      break;
    }
  }
}

extern "C" void emul_set_sim_variant(int v) { g_sim_variant = v & 0xff, g_sim_order = v >> 8; }
extern "C" long long emul_sim_box_tests(int reset) {
  const long long v = g_sim_box_tests;
  if (reset) g_sim_box_tests = 0;
  return v;
}

extern "C" int emul_walk_stats(EmulHandle* h, const float* T16, float bound2, int use_seed, long long* out5) {
  float Tf[12];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) Tf[r * 4 + c] = T16[c * 4 + r];
  const BvhView tv = h->tgt.view();
  SimStats st;
  for (int w0 = 0; w0 < h->src.nvalid; w0 += 64) {
    const int nl = std::min(64, h->src.nvalid - w0);
    F3 q[64];
    float best[64];
    int pos[64], orig[64];
    for (int l = 0; l < nl; l++) {
      const Float4 a = h->src.pts[w0 + l];
      q[l] = transform_point_f(Tf, a.x, a.y, a.z);
      best[l] = bound2, pos[l] = -1, orig[l] = 0x7fffffff;
      const int seed = use_seed ? h->src.corr[w0 + l] : -1;
      if (seed >= 0 && seed < tv.n) {
        const Float4 p = tv.pts[seed];
        const float d = dist2f(q[l], p.x, p.y, p.z);
        if (d <= bound2) best[l] = d, pos[l] = seed, orig[l] = float_as_int_hd(p.w);
      }
    }
    st.waves++;
    sim_wave_walk(
        tv, q, nl, [&](int l, float d) { return d <= best[l]; },
        [&](int pbase) {
          for (int l = 0; l < nl; l++)
            for (int j = 0; j < kLeaf; j++) {
              const Float4 p = tv.pts[pbase + j];
              const float d = dist2f(q[l], p.x, p.y, p.z);
              const int oi = float_as_int_hd(p.w);
              if (d < best[l] || (d == best[l] && oi < orig[l])) best[l] = d, pos[l] = pbase + j, orig[l] = oi;
            }
        },
        st);
    for (int l = 0; l < nl; l++) {
      float d2;
      int o;
      const int j = bvh_nn1(tv, q[l], bound2, &d2, &o);
      if (j != pos[l] || (j >= 0 && d2 != best[l])) st.mismatches++;
    }
  }
  out5[0] = st.groups, out5[1] = st.leaves, out5[2] = st.waves, out5[3] = st.mismatches, out5[4] = 0;
  return 0;
}

// kNN variant on the target cloud against itself (k_knn_cov): also counts leaf points for which at least one lane inserts
extern "C" int emul_walk_stats_knn(EmulHandle* h, int k, long long* out5) {
  const BvhView tv = h->tgt.view();
  SimStats st;
  for (int w0 = 0; w0 < h->tgt.nvalid; w0 += 64) {
    const int nl = std::min(64, h->tgt.nvalid - w0);
    F3 q[64];
    std::vector<std::vector<float>> lists(nl);
    for (int l = 0; l < nl; l++) {
      const Float4 a = h->tgt.pts[w0 + l];
      q[l] = F3{a.x, a.y, a.z};
    }
    auto worst = [&](int l) { return (int)lists[l].size() < k ? FLT_MAX : lists[l].back(); };
    st.waves++;
    sim_wave_walk(
        tv, q, nl, [&](int l, float d) { return d < worst(l); },
        [&](int pbase) {
          for (int j = 0; j < kLeaf; j++) {
            const Float4 p = tv.pts[pbase + j];
            bool anyins = false;
            for (int l = 0; l < nl; l++) {
              const float d = dist2f(q[l], p.x, p.y, p.z);
              if (d < worst(l)) {
                auto& L = lists[l];
                L.insert(std::upper_bound(L.begin(), L.end(), d), d);
                if ((int)L.size() > k) L.pop_back();
                anyins = true;
              }
            }
            if (anyins) st.inserts++;
          }
        },
        st);
  }
  out5[0] = st.groups, out5[1] = st.leaves, out5[2] = st.waves, out5[3] = 0, out5[4] = st.inserts;
  return 0;
}
