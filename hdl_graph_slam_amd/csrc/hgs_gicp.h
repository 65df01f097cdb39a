// hgs_gicp.h — per-point arithmetic and the per-problem Levenberg-Marquardt state machine of the GICP engine
// (fast_gicp::FastGICP + LsqRegistration; reference call site src/hdl_graph_slam/registrations.cpp:27-36).
// HGS_HD: the HIP kernels call these; the test-only host harness (tests/emul) calls the very same functions.
#pragma once
#include "hgs_bvh.h"

namespace hgs {

// Accumulator layout shared by kernels and the 6x6 stage: [0..20] upper triangle of H (row-major, r<=c),
// [21..26] b, [27] sum of errors.
constexpr int kAcc = 28;

struct GicpConsts {
  double max_corr2;         // max_correspondence_distance^2 (double, compared against float d2 like upstream)
  float search_bound2;      // float upper bound handed to the tree search
  double rotation_eps, translation_eps;
  double lm_init_lambda_factor;
  int lm_max_iterations;
  int max_iterations;
  int k_correspondences;
};

// Regularised covariance of one point from its k nearest neighbours (fast_gicp calculate_covariances, FROBENIUS):
//   C = sum (p-mean)(p-mean)^T / k ; C' = ||(C + 1e-3 I)^-1||_F * (C + 1e-3 I)   (== ((C+1e-3I)^-1 / ||.||_F)^-1)
// s1 = sum (p - q), s2 = sum (p - q)(p - q)^T over the `found` neighbours (shifted by the query for conditioning).
HGS_HD Sym3 gicp_neighbour_cov(const double* s1, const Sym3& s2, int found, int k, double lambda) {
  const double inv_f = 1.0 / (double)found, inv_k = 1.0 / (double)k;
  const double mx = s1[0] * inv_f, my = s1[1] * inv_f, mz = s1[2] * inv_f;
  // sum (d-m)(d-m)^T = s2 - found * m m^T
  Sym3 c;
  c.xx = (s2.xx - found * mx * mx) * inv_k + lambda;
  c.xy = (s2.xy - found * mx * my) * inv_k;
  c.xz = (s2.xz - found * mx * mz) * inv_k;
  c.yy = (s2.yy - found * my * my) * inv_k + lambda;
  c.yz = (s2.yz - found * my * mz) * inv_k;
  c.zz = (s2.zz - found * mz * mz) * inv_k + lambda;
  return c;
}
HGS_HD Sym3 gicp_regularized_cov(const double* s1, const Sym3& s2, int found, int k) {
  const Sym3 c = gicp_neighbour_cov(s1, s2, found, k, 1e-3);
  const Sym3 ci = sym3_inverse(c);
  const double f = sqrt(ci.xx * ci.xx + ci.yy * ci.yy + ci.zz * ci.zz + 2.0 * (ci.xy * ci.xy + ci.xz * ci.xz + ci.yz * ci.yz));
  Sym3 o;
  o.xx = f * c.xx, o.xy = f * c.xy, o.xz = f * c.xz, o.yy = f * c.yy, o.yz = f * c.yz, o.zz = f * c.zz;
  return o;
}
// The other fast_gicp::RegularizationMethod values (method = hgs_regularization): NONE keeps C; PLANE / MIN_EIG /
// NORMALIZED_MIN_EIG replace the singular values of C (JacobiSVD upstream; C is symmetric positive semi-definite, so U = V
// = its eigenvectors and the singular values are its eigenvalues) by (1, 1, 1e-3) in descending order / max(sigma, 1e-3) /
// max(sigma / sigma_max, 1e-3) and rebuild U diag(values) V^T.
// the part behind the neighbourhood covariance c (what k_cov_regularize runs on the covariances k_knn_cov<.., 2, ..> staged in fp64)
HGS_HD Sym3 gicp_regularize_cov(const Sym3& c, int method) {
  if (method == 4) return c;
  const double A[9] = {c.xx, c.xy, c.xz, c.xy, c.yy, c.yz, c.xz, c.yz, c.zz};
  double w[3], V[9];
  eig_sym3(A, w, V);  // ascending
  double v[3];
  if (method == 1) {
    v[0] = 1e-3, v[1] = 1.0, v[2] = 1.0;
  } else {
    const double scale = method == 3 ? w[2] : 1.0;
    for (int i = 0; i < 3; i++) {
      const double sv = fabs(w[i]) / scale;  // a singular value
      v[i] = sv > 1e-3 ? sv : 1e-3;
    }
  }
  Sym3 o = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 3; i++) {
    const double x = V[0 * 3 + i], y = V[1 * 3 + i], z = V[2 * 3 + i];
    o.xx += v[i] * x * x, o.xy += v[i] * x * y, o.xz += v[i] * x * z, o.yy += v[i] * y * y, o.yz += v[i] * y * z, o.zz += v[i] * z * z;
  }
  return o;
}
HGS_HD Sym3 gicp_regularized_cov(const double* s1, const Sym3& s2, int found, int k, int method) {
  if (method == 0) return gicp_regularized_cov(s1, s2, found, k);
  return gicp_regularize_cov(gicp_neighbour_cov(s1, s2, found, k, 0.0), method);
}

// Mahalanobis matrix of one correspondence at the linearisation pose: M = (C_B + R C_A R^T)^-1
HGS_HD Sym3 gicp_mahalanobis(const double* R /*3x3 row-major*/, const Sym3& ca, const Sym3& cb) {
  Sym3 rcr = sym3_rotate(R, ca);
  rcr.xx += cb.xx, rcr.xy += cb.xy, rcr.xz += cb.xz, rcr.yy += cb.yy, rcr.yz += cb.yz, rcr.zz += cb.zz;
  return sym3_inverse(rcr);
}

// One correspondence's contribution to the normal equations, in pieces (J = [skew(Ta) | -I], e = b - T a):
//   gicp_point_residual : T a, M e, e^T M e
//   gicp_terms_b        : J^T M e and the error        -> acc[21..27]
//   gicp_terms_tt       : the translation block M      -> acc[15..20]
//   gicp_terms_rt       : the rotation-translation block -> acc[3,4,5, 8,9,10, 12,13,14]
//   gicp_terms_rr       : the rotation block (upper triangle) -> acc[0,1,2, 6,7, 11]
// k_gicp_linearize reduces them over the wave piece by piece (only one piece's values are live at a time); gicp_point_terms
// below adds all of them to acc[28] with the same arithmetic.
struct GicpPointResidual {
  double x, y, z;        // T a
  double mex, mey, mez;  // M e
  double err;            // e^T M e
};
HGS_HD GicpPointResidual gicp_point_residual(const Pose& T, const Sym3& M, float ax, float ay, float az, double bx, double by, double bz) {
  GicpPointResidual r;
  r.x = T.m[0] * ax + T.m[1] * ay + T.m[2] * az + T.m[3];
  r.y = T.m[4] * ax + T.m[5] * ay + T.m[6] * az + T.m[7];
  r.z = T.m[8] * ax + T.m[9] * ay + T.m[10] * az + T.m[11];
  const double ex = bx - r.x, ey = by - r.y, ez = bz - r.z;
  r.mex = M.xx * ex + M.xy * ey + M.xz * ez;
  r.mey = M.xy * ex + M.yy * ey + M.yz * ez;
  r.mez = M.xz * ex + M.yz * ey + M.zz * ez;
  r.err = ex * r.mex + ey * r.mey + ez * r.mez;
  return r;
}
// b = J^T M e = [ S^T Me ; -Me ], then the error
HGS_HD void gicp_terms_b(const GicpPointResidual& r, double* o /*[7]*/) {
  o[0] = r.z * r.mey - r.y * r.mez;
  o[1] = -r.z * r.mex + r.x * r.mez;
  o[2] = r.y * r.mex - r.x * r.mey;
  o[3] = -r.mex, o[4] = -r.mey, o[5] = -r.mez;
  o[6] = r.err;
}
HGS_HD void gicp_terms_tt(const Sym3& M, double* o /*[6]*/) { o[0] = M.xx, o[1] = M.xy, o[2] = M.xz, o[3] = M.yy, o[4] = M.yz, o[5] = M.zz; }
// H_rt = -S^T M, S = skew(Ta): (S^T X)[0][j] = z X[1][j] - y X[2][j]; [1][j] = -z X[0][j] + x X[2][j]; [2][j] = y X[0][j] - x X[1][j]
HGS_HD void gicp_terms_rt(const GicpPointResidual& r, const Sym3& M, double* o /*[9] row-major*/) {
  const double x = r.x, y = r.y, z = r.z;
  const double Mr[3][3] = {{M.xx, M.xy, M.xz}, {M.xy, M.yy, M.yz}, {M.xz, M.yz, M.zz}};
  for (int j = 0; j < 3; j++) {
    o[0 + j] = -(z * Mr[1][j] - y * Mr[2][j]);
    o[3 + j] = -(-z * Mr[0][j] + x * Mr[2][j]);
    o[6 + j] = -(y * Mr[0][j] - x * Mr[1][j]);
  }
}
// H_rr = S^T (M S), upper triangle (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
HGS_HD void gicp_terms_rr(const GicpPointResidual& r, const Sym3& M, double* o /*[6]*/) {
  const double x = r.x, y = r.y, z = r.z;
  const double Mr[3][3] = {{M.xx, M.xy, M.xz}, {M.xy, M.yy, M.yz}, {M.xz, M.yz, M.zz}};
  double A[3][3];  // A = M S: S[:,0]=(0,z,-y) S[:,1]=(-z,0,x) S[:,2]=(y,-x,0)
  for (int q = 0; q < 3; q++) {
    A[q][0] = Mr[q][1] * z - Mr[q][2] * y;
    A[q][1] = -Mr[q][0] * z + Mr[q][2] * x;
    A[q][2] = Mr[q][0] * y - Mr[q][1] * x;
  }
  o[0] = z * A[1][0] - y * A[2][0], o[1] = z * A[1][1] - y * A[2][1], o[2] = z * A[1][2] - y * A[2][2];
  o[3] = -z * A[0][1] + x * A[2][1], o[4] = -z * A[0][2] + x * A[2][2];
  o[5] = y * A[0][2] - x * A[1][2];
}

// all 28 terms of one point in accumulator order (what gicp_point_terms adds to acc[28])
HGS_HD void gicp_point_terms_by_slot(const GicpPointResidual& r, const Sym3& M, double* t /*[28]*/) {
  double rr[6], rt[9], tt[6], bb[7];
  gicp_terms_rr(r, M, rr), gicp_terms_rt(r, M, rt), gicp_terms_tt(M, tt), gicp_terms_b(r, bb);
  t[0] = rr[0], t[1] = rr[1], t[2] = rr[2], t[3] = rt[0], t[4] = rt[1], t[5] = rt[2];
  t[6] = rr[3], t[7] = rr[4], t[8] = rt[3], t[9] = rt[4], t[10] = rt[5];
  t[11] = rr[5], t[12] = rt[6], t[13] = rt[7], t[14] = rt[8];
  for (int k = 0; k < 6; k++) t[15 + k] = tt[k];
  for (int k = 0; k < 7; k++) t[21 + k] = bb[k];
}
// residual e = b - T a, returns e^T M e; optionally adds J^T M J, J^T M e to acc[28] (upper triangle row-major, b, -)
template <bool WITH_JACOBIAN>
HGS_HD double gicp_point_terms(const Pose& T, const Sym3& M, float ax, float ay, float az, double bx, double by, double bz, double* acc) {
  const GicpPointResidual r = gicp_point_residual(T, M, ax, ay, az, bx, by, bz);
  if (WITH_JACOBIAN) {
    double rr[6], rt[9], tt[6], bb[7];
    gicp_terms_rr(r, M, rr), gicp_terms_rt(r, M, rt), gicp_terms_tt(M, tt), gicp_terms_b(r, bb);
    acc[0] += rr[0], acc[1] += rr[1], acc[2] += rr[2], acc[3] += rt[0], acc[4] += rt[1], acc[5] += rt[2];
    acc[6] += rr[3], acc[7] += rr[4], acc[8] += rt[3], acc[9] += rt[4], acc[10] += rt[5];
    acc[11] += rr[5], acc[12] += rt[6], acc[13] += rt[7], acc[14] += rt[8];
    for (int k = 0; k < 6; k++) acc[15 + k] += tt[k];
    for (int k = 0; k < 6; k++) acc[21 + k] += bb[k];
  }
  return r.err;
}

HGS_HD Sym3 sym3_from_floats(float xx, float xy, float xz, float yy, float yz, float zz) {
  Sym3 s;
  s.xx = xx, s.xy = xy, s.xz = xz, s.yy = yy, s.yz = yz, s.zz = zz;
  return s;
}

// ---- per-problem LM state machine (LsqRegistration::computeTransformation / step_lm / is_converged) ----------
enum GicpPhase { GICP_LINEARIZE = 0, GICP_TRY = 1, GICP_DONE = 2 };

struct GicpState {
  Pose x0;        // current estimate
  Pose xi;        // trial estimate of the running LM try
  double H[36];   // J^T M J at x0
  double b[6];
  double d[6];    // last LM step
  double y0, yi;
  double lambda, nu;
  int phase;
  int iterations;  // completed outer iterations
  int lm_try;      // tries within the current step_lm
  int lm_tries_total;
  int converged;
  int pad;
};

HGS_HD void gicp_state_init(GicpState& s, const float* guess_colmajor) {
  s.x0 = pose_from_colmajor_f(guess_colmajor);
  s.xi = s.x0;
  for (int i = 0; i < 36; i++) s.H[i] = 0;
  for (int i = 0; i < 6; i++) s.b[i] = 0, s.d[i] = 0;
  s.y0 = s.yi = 0;
  s.lambda = -1.0;
  s.nu = 2.0;
  s.phase = GICP_LINEARIZE;
  s.iterations = 0, s.lm_try = 0, s.lm_tries_total = 0, s.converged = 0, s.pad = 0;
}

HGS_HD bool gicp_is_converged(const Pose& delta, const GicpConsts& c) {
  double rmax = 0, tmax = 0;
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) {
      const double v = fabs(delta.m[r * 4 + cc] - (r == cc ? 1.0 : 0.0));
      if (v > rmax) rmax = v;
    }
  for (int r = 0; r < 3; r++) {
    const double v = fabs(delta.m[r * 4 + 3]);
    if (v > tmax) tmax = v;
  }
  const double a = rmax / c.rotation_eps, b = tmax / c.translation_eps;
  return (a > b ? a : b) < 1.0;
}

// d = LDLT(H + lambda I).solve(-b); xi = se3_exp(d) * x0
// ws: nullptr (host, oracle-side callers: local arrays) or kGicpControlWorkspace doubles the caller owns (LDS in k_gicp_solve / k_gicp_decide)
constexpr int kGicpControlWorkspace = 36 + 6 + kLdlt6Workspace;
HGS_HD void gicp_solve_try(GicpState& s, Pose* delta_out, double* ws) {
  double *A = ws, *nb = ws + 36;
  for (int i = 0; i < 36; i++) A[i] = s.H[i];
  for (int i = 0; i < 6; i++) A[i * 7] += s.lambda, nb[i] = -s.b[i];
  solve_ldlt6_ws(A, nb, s.d, ws + 42);
  const Pose delta = se3_exp(s.d);
  s.xi = pose_mul(delta, s.x0);
  if (delta_out) *delta_out = delta;
}

// After a linearisation: acc holds the reduced upper-triangle H, b, error at x0.
HGS_HD void gicp_after_linearize(GicpState& s, const double* acc, const GicpConsts& c, double* ws) {
  int k = 0;
  for (int r = 0; r < 6; r++)
    for (int cc = r; cc < 6; cc++) {
      s.H[r * 6 + cc] = acc[k];
      s.H[cc * 6 + r] = acc[k];
      k++;
    }
  for (int i = 0; i < 6; i++) s.b[i] = acc[21 + i];
  s.y0 = acc[27];
  if (s.lambda < 0.0) {
    double mx = 0;
    for (int i = 0; i < 6; i++) {
      const double v = fabs(s.H[i * 7]);
      if (v > mx) mx = v;
    }
    s.lambda = c.lm_init_lambda_factor * mx;
  }
  s.nu = 2.0;
  s.lm_try = 0;
  gicp_solve_try(s, nullptr, ws);
  s.phase = GICP_TRY;
}

// After the trial error yi at xi is known: accept / reject exactly as step_lm + the outer loop do.
HGS_HD void gicp_after_error(GicpState& s, double yi, const GicpConsts& c, double* ws) {
  s.yi = yi;
  s.lm_try++;
  s.lm_tries_total++;
  double denom = 0;
  for (int i = 0; i < 6; i++) denom += s.d[i] * (s.lambda * s.d[i] - s.b[i]);
  const double rho = (s.y0 - yi) / denom;
  const Pose delta = se3_exp(s.d);
  bool step_ok, finished_step;
  if (rho < 0) {
    if (gicp_is_converged(delta, c)) {
      step_ok = true, finished_step = true;  // step_lm returns true without moving x0
    } else {
      s.lambda = s.nu * s.lambda;
      s.nu = 2 * s.nu;
      if (s.lm_try >= c.lm_max_iterations) {
        step_ok = false, finished_step = true;  // "lm not converged": outer loop breaks
      } else {
        gicp_solve_try(s, nullptr, ws);  // next try against the same linearisation
        return;                      // stay in GICP_TRY
      }
    }
  } else {
    s.x0 = s.xi;
    const double t = 2 * rho - 1;
    const double f = 1.0 - t * t * t;
    s.lambda = s.lambda * (f > 1.0 / 3.0 ? f : 1.0 / 3.0);
    step_ok = true, finished_step = true;
  }
  (void)finished_step;
  s.iterations++;
  if (!step_ok) {
    s.converged = 0;
    s.phase = GICP_DONE;
    return;
  }
  s.converged = gicp_is_converged(delta, c) ? 1 : 0;
  if (s.converged || s.iterations >= c.max_iterations) {
    s.phase = GICP_DONE;
  } else {
    s.phase = GICP_LINEARIZE;
  }
}
// the same steps with the workspace as a local array (host-side callers: tests/emul, the host mirror of the kernels)
HGS_HD void gicp_after_linearize(GicpState& s, const double* acc, const GicpConsts& c) {
  double ws[kGicpControlWorkspace];
  gicp_after_linearize(s, acc, c, ws);
}
HGS_HD void gicp_after_error(GicpState& s, double yi, const GicpConsts& c) {
  double ws[kGicpControlWorkspace];
  gicp_after_error(s, yi, c, ws);
}

}  // namespace hgs
