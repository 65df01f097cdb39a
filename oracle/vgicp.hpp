// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp header). Parity unpinned by the reference.
//
// CPU restatement of fast_gicp::FastVGICP + GaussianVoxelMap (ADDITIVE mode), the engine hdl_graph_slam
// constructs for registration_method == "FAST_VGICP" (src/hdl_graph_slam/registrations.cpp:48-56).
// Not under /root/reference; follows SURVEY.md Appendix A.3 [UPSTREAM-KNOWLEDGE]:
//   voxel key  = floor(p / resolution - 0.5)      (voxel centres sit on integer multiples of the resolution)
//   voxel      = { n, mean = sum(b_j)/n, cov = sum(C_B,j)/n } over the target points falling in it
//   per iteration: for each source point, look up the voxel(s) of T a_i (DIRECT1/7/27); every hit is a
//   correspondence with weight w = sqrt(n); M = (C_voxel + R C_A R^T)^-1; same J as FastGICP.
//   The voxel map is rebuilt at the start of every align (computeTransformation resets it).
#pragma once
#include <unordered_map>
#include "gicp.hpp"

namespace hgso {

struct GaussVoxel {
  int n = 0;
  V3 mean{0, 0, 0};
  M3 cov = M3::zero();
};
struct VoxelKey {
  int x, y, z;
  bool operator==(const VoxelKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelKeyHash {
  size_t operator()(const VoxelKey& k) const { return ((size_t)(uint32_t)k.x * 73856093u) ^ ((size_t)(uint32_t)k.y * 19349669u) ^ ((size_t)(uint32_t)k.z * 83492791u); }
};

class FastVGICP {
public:
  explicit FastVGICP(const hgs_params& p) : prm(p), base(p) {}
  hgs_params prm;
  FastGICP base;  // reuse covariance + LM machinery state containers
  std::shared_ptr<OCloud> source, target;
  std::unordered_map<VoxelKey, GaussVoxel, VoxelKeyHash> voxels;
  bool voxels_valid = false;
  std::vector<std::pair<int, const GaussVoxel*>> voxel_correspondences;
  std::vector<M3> voxel_mahalanobis;
  std::vector<GicpTraceEntry> trace;
  double lm_lambda = -1, last_error = 0;
  int lm_tries_total = 0;

  void set_target(std::shared_ptr<OCloud> t) {
    target = t;
    voxels_valid = false;
  }
  void ensure_covs() {
    const int k = prm.correspondence_randomness;
    if (source && source->covs.size() != source->pts.size()) calculate_covariances(*source, k, prm.regularization_method);
    if (target && target->covs.size() != target->pts.size()) calculate_covariances(*target, k, prm.regularization_method);
  }
  VoxelKey coord_of(V3 p) const {
    return {(int)std::floor(p.x / prm.resolution - 0.5), (int)std::floor(p.y / prm.resolution - 0.5), (int)std::floor(p.z / prm.resolution - 0.5)};
  }
  void create_voxelmap() {
    voxels.clear();
    for (size_t i = 0; i < target->pts.size(); i++) {
      const P3f& p = target->pts[i];
      GaussVoxel& v = voxels[coord_of({p.x, p.y, p.z})];
      v.n++;
      v.mean = v.mean + V3{p.x, p.y, p.z};
      v.cov = v.cov + target->covs[i];
    }
    for (auto& kv : voxels) {
      kv.second.mean = (1.0 / kv.second.n) * kv.second.mean;
      kv.second.cov = (1.0 / kv.second.n) * kv.second.cov;
    }
    voxels_valid = true;
  }

  void update_correspondences(const Iso& T) {
    voxel_correspondences.clear();
    static const int off7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    std::vector<std::array<int, 3>> offsets;
    if (prm.neighbor_search == HGS_DIRECT27) {
      for (int i = -1; i <= 1; i++)
        for (int j = -1; j <= 1; j++)
          for (int k = -1; k <= 1; k++) offsets.push_back({i, j, k});
    } else if (prm.neighbor_search == HGS_DIRECT7) {
      for (auto& o : off7) offsets.push_back({o[0], o[1], o[2]});
    } else {
      offsets.push_back({0, 0, 0});
    }
    for (size_t i = 0; i < source->pts.size(); i++) {
      const P3f& a = source->pts[i];
      const V3 ta = apply(T, V3{a.x, a.y, a.z});
      const VoxelKey c = coord_of(ta);
      for (auto& o : offsets) {
        auto it = voxels.find({c.x + o[0], c.y + o[1], c.z + o[2]});
        if (it != voxels.end()) voxel_correspondences.push_back({(int)i, &it->second});
      }
    }
    voxel_mahalanobis.resize(voxel_correspondences.size());
    const M3 Rt = transpose(T.R);
#pragma omp parallel for schedule(guided, 8)
    for (long i = 0; i < (long)voxel_correspondences.size(); i++) {
      const auto& vc = voxel_correspondences[i];
      voxel_mahalanobis[i] = inverse(vc.second->cov + T.R * source->covs[vc.first] * Rt);
    }
  }

  double accumulate(const Iso& T, M6* H, V6* b) const {
    const long n = (long)voxel_correspondences.size();
    const int nt = omp_get_max_threads();
    std::vector<M6> Hs(nt, M6::zero());
    std::vector<V6> bs(nt, V6::zero());
    double sum = 0;
#pragma omp parallel for reduction(+ : sum) schedule(guided, 8)
    for (long i = 0; i < n; i++) {
      const auto& vc = voxel_correspondences[i];
      const P3f& a = source->pts[vc.first];
      const V3 ta = apply(T, V3{a.x, a.y, a.z});
      const V3 e = vc.second->mean - ta;
      const double w = std::sqrt((double)vc.second->n);
      const M3& M = voxel_mahalanobis[i];
      const V3 Me = M * e;
      sum += w * dot(e, Me);
      if (!H) continue;
      double J[3][6] = {{0}};
      const M3 S = skew(ta);
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) J[r][c] = S.m[r][c];
        J[r][3 + r] = -1.0;
      }
      double MJ[3][6];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 6; c++) MJ[r][c] = M.m[r][0] * J[0][c] + M.m[r][1] * J[1][c] + M.m[r][2] * J[2][c];
      M6& Ht = Hs[omp_get_thread_num()];
      V6& btv = bs[omp_get_thread_num()];
      for (int r = 0; r < 6; r++) {
        for (int c = 0; c < 6; c++) Ht.m[r][c] += w * (J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c]);
        btv.v[r] += w * (J[0][r] * Me.x + J[1][r] * Me.y + J[2][r] * Me.z);
      }
    }
    if (H) {
      *H = M6::zero();
      *b = V6::zero();
      for (int t = 0; t < nt; t++)
        for (int r = 0; r < 6; r++) {
          for (int c = 0; c < 6; c++) H->m[r][c] += Hs[t].m[r][c];
          b->v[r] += bs[t].v[r];
        }
    }
    return sum;
  }

  double linearize(const Iso& T, M6* H, V6* b) {
    if (!voxels_valid) create_voxelmap();
    update_correspondences(T);
    return accumulate(T, H, b);
  }
  double compute_error(const Iso& T) const { return accumulate(T, nullptr, nullptr); }

  bool step_lm(Iso& x0, Iso& delta) {
    M6 H;
    V6 b;
    const double y0 = linearize(x0, &H, &b);
    last_error = y0;
    if (lm_lambda < 0.0) {
      double mx = 0;
      for (int i = 0; i < 6; i++) mx = std::max(mx, std::fabs(H.m[i][i]));
      lm_lambda = prm.lm_init_lambda_factor * mx;
    }
    double nu = 2.0;
    for (int i = 0; i < prm.lm_max_iterations; i++) {
      lm_tries_total++;
      M6 A = H;
      for (int d = 0; d < 6; d++) A.m[d][d] += lm_lambda;
      V6 nb;
      for (int d = 0; d < 6; d++) nb.v[d] = -b.v[d];
      const V6 d = solve_ldlt6(A, nb);
      delta = se3_exp(d);
      const Iso xi = delta * x0;
      const double yi = compute_error(xi);
      V6 ld;
      for (int k = 0; k < 6; k++) ld.v[k] = lm_lambda * d.v[k] - b.v[k];
      const double rho = (y0 - yi) / dot(d, ld);
      if (rho < 0) {
        if (base.is_converged(delta)) return true;
        lm_lambda = nu * lm_lambda;
        nu = 2 * nu;
        continue;
      }
      x0 = xi;
      lm_lambda = lm_lambda * std::max(1.0 / 3.0, 1.0 - std::pow(2 * rho - 1, 3));
      return true;
    }
    return false;
  }

  void align(const float guess[16], hgs_result* out) {
    trace.clear();
    ensure_covs();
    voxels_valid = false;  // FastVGICP::computeTransformation resets the voxel map
    Iso x0 = iso_from_colmajor_f(guess);
    lm_lambda = -1.0;
    lm_tries_total = 0;
    bool converged = false;
    int it = 0;
    for (; it < prm.max_iterations && !converged; it++) {
      Iso delta = Iso::identity();
      const bool ok = step_lm(x0, delta);
      GicpTraceEntry te{};
      te.error = last_error, te.lambda = lm_lambda;
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) te.T[r * 4 + c] = x0.R.m[r][c];
      }
      te.T[3] = x0.t.x, te.T[7] = x0.t.y, te.T[11] = x0.t.z;
      trace.push_back(te);
      if (!ok) {
        it++;
        break;
      }
      converged = base.is_converged(delta);
    }
    iso_to_colmajor_f(x0, out->final_transformation);
    out->converged = converged ? 1 : 0;
    out->iterations = it;
    out->error = last_error;
    out->fitness_score = std::numeric_limits<double>::quiet_NaN();
    out->num_inliers = 0;
    out->candidate_id = 0;
    out->lm_tries = lm_tries_total;
    out->reserved = 0;
  }
};

}  // namespace hgso
