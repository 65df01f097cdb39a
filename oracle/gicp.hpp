// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp header). Parity unpinned by the reference.
//
// CPU restatement (double precision, OpenMP over points) of fast_gicp::FastGICP + LsqRegistration, the
// engine hdl_graph_slam constructs for registration_method == "FAST_GICP"
// (src/hdl_graph_slam/registrations.cpp:27-36; launch default launch/hdl_graph_slam.launch:73,127).
// fast_gicp is NOT under /root/reference (cloned at unpinned HEAD by docker/noetic/Dockerfile:15); the
// algorithm below follows its published sources as restated in SURVEY.md Appendix A.2:
//   calculate_covariances  : exact k-NN (k = reg_correspondence_randomness) -> 4xk neighbour matrix, centred,
//                            C = N N^T / k ; FROBENIUS regularisation  C' = ((C+1e-3 I)^-1 / ||.||_F)^-1
//   update_correspondences : q = float(T) a_i ; exact 1-NN in the target ; valid iff d2 < max_corr^2 ;
//                            M_i = (C_B + R C_A R^T)^-1
//   linearize              : e = b_j - T a_i ; J = [ skew(T a_i) | -I ] ; H += J^T M J ; b += J^T M e ; err += e^T M e
//   step_lm / is_converged : Levenberg-Marquardt on SE(3) with se3_exp(d) left-multiplied.
#pragma once
#include <vector>
#include <memory>
#include <omp.h>
#include "linalg.hpp"
#include "kdtree.hpp"
#include "../include/hgs_registration.h"

namespace hgso {

// Float point transform shared (bit-for-bit) with the HIP path: q = fma chain starting from the translation.
inline P3f transform_point_f(const float Tf[12], const P3f& a) {  // Tf row-major 3x4
  P3f q;
  q.x = std::fmaf(Tf[2], a.z, std::fmaf(Tf[1], a.y, std::fmaf(Tf[0], a.x, Tf[3])));
  q.y = std::fmaf(Tf[6], a.z, std::fmaf(Tf[5], a.y, std::fmaf(Tf[4], a.x, Tf[7])));
  q.z = std::fmaf(Tf[10], a.z, std::fmaf(Tf[9], a.y, std::fmaf(Tf[8], a.x, Tf[11])));
  return q;
}
inline void iso_to_rowmajor_f(const Iso& T, float Tf[12]) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Tf[r * 4 + c] = (float)T.R.m[r][c];
  }
  Tf[3] = (float)T.t.x;
  Tf[7] = (float)T.t.y;
  Tf[11] = (float)T.t.z;
}

struct OCloud {
  std::vector<P3f> pts;       // finite points only
  std::vector<int> orig;      // index in the caller's array
  KdTree tree;
  std::vector<M3> covs;       // GICP covariances (empty until computed)
  size_t n_input = 0;

  void assign(const void* data, size_t n, size_t stride) {
    pts.clear();
    orig.clear();
    covs.clear();
    n_input = n;
    const char* p = (const char*)data;
    for (size_t i = 0; i < n; i++) {
      const float* f = (const float*)(p + i * stride);
      if (std::isfinite(f[0]) && std::isfinite(f[1]) && std::isfinite(f[2])) {
        pts.push_back({f[0], f[1], f[2]});
        orig.push_back((int)i);
      }
    }
    tree.build(pts);
  }
};

// fast_gicp::FastGICP::calculate_covariances.  `method` = hgs_regularization (fast_gicp::RegularizationMethod):
// FROBENIUS ((C + 1e-3 I)^-1 / ||.||_F)^-1 ; NONE ; PLANE / MIN_EIG / NORMALIZED_MIN_EIG rebuild U diag(values) V^T from
// the JacobiSVD of C — C is symmetric positive semi-definite, so its SVD is its eigen-decomposition (U = V).
inline void calculate_covariances(OCloud& c, int k, int method = HGS_REG_FROBENIUS) {
  const int n = (int)c.pts.size();
  c.covs.assign(n, M3::zero());
#pragma omp parallel for schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    std::vector<Neighbor> nb(k);
    const int found = c.tree.knn(c.pts[i], k, nb.data());
    double mean[3] = {0, 0, 0};
    for (int j = 0; j < found; j++) {
      const P3f& p = c.pts[nb[j].idx];
      mean[0] += p.x, mean[1] += p.y, mean[2] += p.z;
    }
    for (double& m : mean) m /= found;
    M3 C = M3::zero();
    for (int j = 0; j < found; j++) {
      const P3f& p = c.pts[nb[j].idx];
      const double d[3] = {p.x - mean[0], p.y - mean[1], p.z - mean[2]};
      for (int r = 0; r < 3; r++)
        for (int s = 0; s < 3; s++) C.m[r][s] += d[r] * d[s];
    }
    // upstream divides by k_correspondences_ (== found whenever the cloud has >= k points)
    C = (1.0 / k) * C;
    if (method == HGS_REG_NONE) {
      c.covs[i] = C;
    } else if (method == HGS_REG_FROBENIUS) {
      for (int r = 0; r < 3; r++) C.m[r][r] += 1e-3;
      const M3 Cinv = inverse(C);
      c.covs[i] = inverse((1.0 / frobenius(Cinv)) * Cinv);
    } else {
      double ev[3];  // ascending
      M3 evec;
      eig_sym3(C, ev, evec);
      double values[3];
      if (method == HGS_REG_PLANE) {
        values[0] = 1e-3, values[1] = 1.0, values[2] = 1.0;  // (1, 1, 1e-3) against singular values in descending order
      } else {
        const double scale = method == HGS_REG_NORMALIZED_MIN_EIG ? ev[2] : 1.0;
        for (int a = 0; a < 3; a++) values[a] = std::max(std::fabs(ev[a]) / scale, 1e-3);
      }
      M3 out = M3::zero();
      for (int a = 0; a < 3; a++)
        for (int r = 0; r < 3; r++)
          for (int s = 0; s < 3; s++) out.m[r][s] += values[a] * evec.m[r][a] * evec.m[s][a];
      c.covs[i] = out;
    }
  }
}

struct GicpTraceEntry {
  double T[12];  // row-major 3x4 of x0 after the iteration
  double error;  // y0 of the linearisation
  double lambda;
  int lm_tries;
};

class FastGICP {
public:
  explicit FastGICP(const hgs_params& p) : prm(p) {}

  hgs_params prm;
  std::shared_ptr<OCloud> source, target;
  std::vector<int> correspondences;  // target index (compacted) or -1
  std::vector<M3> mahalanobis;
  std::vector<GicpTraceEntry> trace;
  M6 final_hessian = M6::zero();

  void ensure_covs() {
    if (source && source->covs.size() != source->pts.size()) calculate_covariances(*source, prm.correspondence_randomness, prm.regularization_method);
    if (target && target->covs.size() != target->pts.size()) calculate_covariances(*target, prm.correspondence_randomness, prm.regularization_method);
  }

  void update_correspondences(const Iso& T) {
    const int n = (int)source->pts.size();
    correspondences.assign(n, -1);
    mahalanobis.assign(n, M3::zero());
    float Tf[12];
    iso_to_rowmajor_f(T, Tf);
    const double thr2 = prm.max_correspondence_distance * prm.max_correspondence_distance;
    const float bound2 = thr2 >= (double)FLT_MAX ? FLT_MAX : std::nextafter((float)thr2, FLT_MAX);
    const M3 Rt = transpose(T.R);
#pragma omp parallel for schedule(guided, 8)
    for (int i = 0; i < n; i++) {
      const P3f q = transform_point_f(Tf, source->pts[i]);
      const Neighbor nb = target->tree.nn(q, bound2);
      if (nb.idx < 0 || !((double)nb.d2 < thr2)) continue;
      correspondences[i] = nb.idx;
      const M3 RCR = target->covs[nb.idx] + T.R * source->covs[i] * Rt;
      mahalanobis[i] = inverse(RCR);
    }
  }

  // returns sum of errors; H,b optional
  double accumulate(const Iso& T, M6* H, V6* b) const {
    const int n = (int)source->pts.size();
    const int nt = omp_get_max_threads();
    std::vector<M6> Hs(nt, M6::zero());
    std::vector<V6> bs(nt, V6::zero());
    double sum = 0;
#pragma omp parallel for reduction(+ : sum) schedule(guided, 8)
    for (int i = 0; i < n; i++) {
      const int j = correspondences[i];
      if (j < 0) continue;
      const P3f& a = source->pts[i];
      const P3f& bt = target->pts[j];
      const V3 ta = apply(T, V3{a.x, a.y, a.z});
      const V3 e = V3{bt.x, bt.y, bt.z} - ta;
      const M3& M = mahalanobis[i];
      const V3 Me = M * e;
      sum += dot(e, Me);
      if (!H) continue;
      // J = [skew(ta) | -I]  (3x6)
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
        for (int c = 0; c < 6; c++) Ht.m[r][c] += J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c];
        btv.v[r] += J[0][r] * Me.x + J[1][r] * Me.y + J[2][r] * Me.z;
      }
    }
    if (H) {
      *H = M6::zero();
      *b = V6::zero();
      for (int t = 0; t < nt; t++) {
        for (int r = 0; r < 6; r++) {
          for (int c = 0; c < 6; c++) H->m[r][c] += Hs[t].m[r][c];
          b->v[r] += bs[t].v[r];
        }
      }
    }
    return sum;
  }

  double linearize(const Iso& T, M6* H, V6* b) {
    update_correspondences(T);
    return accumulate(T, H, b);
  }
  double compute_error(const Iso& T) const { return accumulate(T, nullptr, nullptr); }

  bool is_converged(const Iso& delta) const {
    double rmax = 0, tmax = 0;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) rmax = std::max(rmax, std::fabs(delta.R.m[r][c] - (r == c ? 1.0 : 0.0)));
    tmax = std::max({std::fabs(delta.t.x), std::fabs(delta.t.y), std::fabs(delta.t.z)});
    return std::max(rmax / prm.rotation_epsilon, tmax / prm.transformation_epsilon) < 1.0;
  }

  double lm_lambda = -1.0;
  double last_error = 0.0;
  int lm_tries_total = 0;

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
        if (is_converged(delta)) return true;
        lm_lambda = nu * lm_lambda;
        nu = 2 * nu;
        continue;
      }
      x0 = xi;
      lm_lambda = lm_lambda * std::max(1.0 / 3.0, 1.0 - std::pow(2 * rho - 1, 3));
      final_hessian = H;
      return true;
    }
    return false;
  }

  // LsqRegistration::computeTransformation
  void align(const float guess[16], hgs_result* out) {
    trace.clear();
    ensure_covs();
    Iso x0 = iso_from_colmajor_f(guess);
    lm_lambda = -1.0;
    lm_tries_total = 0;
    bool converged = false;
    int it = 0;
    for (; it < prm.max_iterations && !converged; it++) {
      Iso delta = Iso::identity();
      const int tries_before = lm_tries_total;
      if (!step_lm(x0, delta)) {
        GicpTraceEntry te{};
        te.error = last_error, te.lambda = lm_lambda, te.lm_tries = lm_tries_total - tries_before;
        store_T(x0, te.T);
        trace.push_back(te);
        it++;
        break;  // "lm not converged"
      }
      converged = is_converged(delta);
      GicpTraceEntry te{};
      te.error = last_error, te.lambda = lm_lambda, te.lm_tries = lm_tries_total - tries_before;
      store_T(x0, te.T);
      trace.push_back(te);
    }
    final_T = x0;
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

  Iso final_T = Iso::identity();

private:
  static void store_T(const Iso& x, double T[12]) {
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) T[r * 4 + c] = x.R.m[r][c];
    }
    T[3] = x.t.x, T[7] = x.t.y, T[11] = x.t.z;
  }
};

// pcl::Registration::getFitnessScore(max_range) — verbatim semantics of
// src/hdl_graph_slam/information_matrix_calculator.cpp:49-80 (note: SQUARED distance compared with max_range).
inline double fitness_score(const OCloud& target, const OCloud& source, const float T_colmajor[16], double max_range, uint32_t* n_inliers) {
  float Tf[12];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) Tf[r * 4 + c] = T_colmajor[c * 4 + r];
  const int n = (int)source.pts.size();
  double sum = 0;
  long nr = 0;
#pragma omp parallel for reduction(+ : sum, nr) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    const P3f q = transform_point_f(Tf, source.pts[i]);
    const Neighbor nb = target.tree.nn(q);
    if (nb.idx >= 0 && (double)nb.d2 <= max_range) {
      sum += (double)nb.d2;
      nr++;
    }
  }
  if (n_inliers) *n_inliers = (uint32_t)nr;
  return nr > 0 ? sum / (double)nr : std::numeric_limits<double>::max();
}

}  // namespace hgso
