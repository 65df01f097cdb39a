// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into / imported by the product path.
// Parity status: UNPINNED by the reference (hdl_graph_slam ships no tests or golden vectors and its
// registration arithmetic lives in un-vendored ndt_omp / fast_gicp / PCL); see DESIGN.md.
//
// Dependency-free small dense linear algebra in double precision used by the CPU restatement of the
// registration engines that hdl_graph_slam selects in src/hdl_graph_slam/registrations.cpp:22-124.
// (Upstream uses Eigen: Matrix3d/Matrix4d inverse, SelfAdjointEigenSolver<Matrix3d>, LDLT<6x6>, JacobiSVD<6x6>.)
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <limits>

namespace hgso {

struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Row-major 3x3.
struct M3 {
  double m[3][3];
  static M3 zero() {
    M3 r;
    std::memset(r.m, 0, sizeof(r.m));
    return r;
  }
  static M3 identity() {
    M3 r = zero();
    r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0;
    return r;
  }
};

inline M3 operator*(const M3& a, const M3& b) {
  M3 r = M3::zero();
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 3; j++) r.m[i][j] += a.m[i][k] * b.m[k][j];
  return r;
}
inline M3 operator+(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
inline M3 operator*(double s, const M3& a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j];
  return r;
}
inline V3 operator*(const M3& a, V3 v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline M3 transpose(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
inline double det(const M3& a) {
  return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) - a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0]) +
         a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}
// Cofactor inverse (what Eigen does for fixed 3x3).
inline M3 inverse(const M3& a) {
  const double d = det(a);
  const double id = 1.0 / d;
  M3 r;
  r.m[0][0] = (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) * id;
  r.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * id;
  r.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * id;
  r.m[1][0] = (a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2]) * id;
  r.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * id;
  r.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * id;
  r.m[2][0] = (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]) * id;
  r.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * id;
  r.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * id;
  return r;
}
inline double frobenius(const M3& a) {
  double s = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) s += a.m[i][j] * a.m[i][j];
  return std::sqrt(s);
}
inline M3 skew(V3 v) {
  M3 r = M3::zero();
  r.m[0][1] = -v.z;
  r.m[0][2] = v.y;
  r.m[1][0] = v.z;
  r.m[1][2] = -v.x;
  r.m[2][0] = -v.y;
  r.m[2][1] = v.x;
  return r;
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations. Eigenvalues ascending (as
// Eigen::SelfAdjointEigenSolver), eigenvectors in the COLUMNS of evec.
inline void eig_sym3(const M3& a_in, double eval[3], M3& evec) {
  M3 a = a_in;
  evec = M3::identity();
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = a.m[0][1] * a.m[0][1] + a.m[0][2] * a.m[0][2] + a.m[1][2] * a.m[1][2];
    double diag = a.m[0][0] * a.m[0][0] + a.m[1][1] * a.m[1][1] + a.m[2][2] * a.m[2][2];
    if (off <= 1e-60 || off <= 1e-34 * diag) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a.m[p][q] == 0.0) continue;
        const double theta = (a.m[q][q] - a.m[p][p]) / (2.0 * a.m[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {  // A <- A J
          const double akp = a.m[k][p], akq = a.m[k][q];
          a.m[k][p] = c * akp - s * akq;
          a.m[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- J^T A
          const double apk = a.m[p][k], aqk = a.m[q][k];
          a.m[p][k] = c * apk - s * aqk;
          a.m[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = evec.m[k][p], vkq = evec.m[k][q];
          evec.m[k][p] = c * vkp - s * vkq;
          evec.m[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int order[3] = {0, 1, 2};
  double d[3] = {a.m[0][0], a.m[1][1], a.m[2][2]};
  std::sort(order, order + 3, [&](int i, int j) { return d[i] < d[j]; });
  M3 v;
  for (int c = 0; c < 3; c++) {
    eval[c] = d[order[c]];
    for (int r = 0; r < 3; r++) v.m[r][c] = evec.m[r][order[c]];
  }
  evec = v;
}

// ---------------------------------------------------------------- 6x6
struct V6 {
  double v[6];
  static V6 zero() {
    V6 r;
    for (double& x : r.v) x = 0;
    return r;
  }
};
struct M6 {
  double m[6][6];
  static M6 zero() {
    M6 r;
    std::memset(r.m, 0, sizeof(r.m));
    return r;
  }
};
inline double dot(const V6& a, const V6& b) {
  double s = 0;
  for (int i = 0; i < 6; i++) s += a.v[i] * b.v[i];
  return s;
}
inline double norm(const V6& a) { return std::sqrt(dot(a, a)); }

// Solve A x = b for symmetric A through an LDL^T factorisation with symmetric (diagonal) pivoting —
// the role Eigen::LDLT<Matrix<double,6,6>> plays in fast_gicp's LsqRegistration::step_lm.
inline V6 solve_ldlt6(const M6& A_in, const V6& b_in) {
  double A[6][6];
  std::memcpy(A, A_in.m, sizeof(A));
  int perm[6] = {0, 1, 2, 3, 4, 5};
  double L[6][6] = {{0}};
  double D[6];
  for (int k = 0; k < 6; k++) {
    int piv = k;
    double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < 6; i++)
      if (std::fabs(A[i][i]) > best) best = std::fabs(A[i][i]), piv = i;
    if (piv != k) {
      for (int j = 0; j < 6; j++) std::swap(A[k][j], A[piv][j]);
      for (int i = 0; i < 6; i++) std::swap(A[i][k], A[i][piv]);
      for (int j = 0; j < k; j++) std::swap(L[k][j], L[piv][j]);
      std::swap(perm[k], perm[piv]);
    }
    D[k] = A[k][k];
    L[k][k] = 1.0;
    for (int i = k + 1; i < 6; i++) L[i][k] = (D[k] != 0.0) ? A[i][k] / D[k] : 0.0;
    for (int i = k + 1; i < 6; i++)
      for (int j = k + 1; j < 6; j++) A[i][j] -= L[i][k] * D[k] * L[j][k];
  }
  double y[6], z[6];
  for (int i = 0; i < 6; i++) {
    double s = b_in.v[perm[i]];
    for (int j = 0; j < i; j++) s -= L[i][j] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < 6; i++) y[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
    for (int j = i + 1; j < 6; j++) s -= L[j][i] * z[j];
    z[i] = s;
  }
  V6 x;
  for (int i = 0; i < 6; i++) x.v[perm[i]] = z[i];
  return x;
}

// x = pinv(A) b through a one-sided (Hestenes) Jacobi SVD, singular values below
// 6*eps*sigma_max dropped — the role of Eigen::JacobiSVD<Matrix<double,6,6>>(H, FullU|FullV).solve(-g)
// in ndt_omp's computeTransformation.
inline V6 solve_svd6(const M6& A, const V6& b) {
  double U[6][6], V[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) U[i][j] = A.m[i][j], V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    // pair order: a round-robin tournament (5 rounds of 3 disjoint pairs) instead of the row-cyclic order — any order that
    // visits every pair once per sweep is a valid Jacobi sweep; this one lets the device run a round's three rotations in
    // lock-step (they commute exactly), see hgs_math.h
    static const int kPairs[15][2] = {{0, 5}, {1, 4}, {2, 3}, {0, 4}, {3, 5}, {1, 2}, {0, 3}, {2, 4}, {1, 5}, {0, 2}, {1, 3}, {4, 5}, {0, 1}, {2, 5}, {3, 4}};
    for (int pi = 0; pi < 15; pi++) {
        const int p = kPairs[pi][0], q = kPairs[pi][1];
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 6; k++) alpha += U[k][p] * U[k][p], beta += U[k][q] * U[k][q], gamma += U[k][p] * U[k][q];
        if (gamma == 0.0 || std::fabs(gamma) <= DBL_EPSILON * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 6; k++) {
          const double up = U[k][p], uq = U[k][q];
          U[k][p] = c * up - s * uq;
          U[k][q] = s * up + c * uq;
          const double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq;
          V[k][q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sigma[6], smax = 0;
  for (int j = 0; j < 6; j++) {
    double s = 0;
    for (int k = 0; k < 6; k++) s += U[k][j] * U[k][j];
    sigma[j] = std::sqrt(s);
    smax = std::max(smax, sigma[j]);
  }
  const double thr = 6.0 * std::numeric_limits<double>::epsilon() * smax;
  V6 x = V6::zero();
  for (int j = 0; j < 6; j++) {
    if (!(sigma[j] > thr) || sigma[j] == 0.0) continue;
    double ub = 0;  // (u_j . b) / sigma_j, with u_j = U[:,j]/sigma_j
    for (int k = 0; k < 6; k++) ub += U[k][j] * b.v[k];
    const double coef = ub / (sigma[j] * sigma[j]);
    for (int k = 0; k < 6; k++) x.v[k] += coef * V[k][j];
  }
  return x;
}

// ---------------------------------------------------------------- rigid transforms (double, row-major 4x4)
struct Iso {
  M3 R;
  V3 t;
  static Iso identity() { return {M3::identity(), {0, 0, 0}}; }
};
inline Iso operator*(const Iso& a, const Iso& b) { return {a.R * b.R, a.R * b.t + a.t}; }
inline V3 apply(const Iso& T, V3 p) { return T.R * p + T.t; }

// fast_gicp so3.hpp: so3_exp / se3_exp (rotation part first, translation last; d = [omega, v]).
inline M3 quat_to_rot(double w, double x, double y, double z) {
  // Eigen::Quaterniond::toRotationMatrix() (no normalisation)
  M3 r;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  r.m[0][0] = 1 - (tyy + tzz);
  r.m[0][1] = txy - twz;
  r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz;
  r.m[1][1] = 1 - (txx + tzz);
  r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy;
  r.m[2][1] = tyz + twx;
  r.m[2][2] = 1 - (txx + tyy);
  return r;
}
inline M3 so3_exp(V3 omega) {
  const double theta_sq = dot(omega, omega);
  double imag, real;
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double theta = std::sqrt(theta_sq);
    const double half = 0.5 * theta;
    imag = std::sin(half) / theta;
    real = std::cos(half);
  }
  return quat_to_rot(real, imag * omega.x, imag * omega.y, imag * omega.z);
}
inline Iso se3_exp(const V6& a) {
  const V3 omega{a.v[0], a.v[1], a.v[2]};
  const V3 v{a.v[3], a.v[4], a.v[5]};
  const double theta = std::sqrt(dot(omega, omega));
  const M3 R = so3_exp(omega);
  const M3 Om = skew(omega);
  const M3 Om2 = Om * Om;
  M3 Vm;
  if (theta < 1e-10) {
    Vm = R;
  } else {
    const double th2 = theta * theta;
    Vm = M3::identity() + ((1.0 - std::cos(theta)) / th2) * Om + ((theta - std::sin(theta)) / (th2 * theta)) * Om2;
  }
  return {R, Vm * v};
}

// Column-major float[16] <-> Iso (the Eigen::Matrix4f the nodelets pass around).
inline Iso iso_from_colmajor_f(const float* m) {
  Iso T;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T.R.m[r][c] = (double)m[c * 4 + r];
  }
  T.t = {(double)m[12], (double)m[13], (double)m[14]};
  return T;
}
inline void iso_to_colmajor_f(const Iso& T, float* m) {
  for (int i = 0; i < 16; i++) m[i] = 0.f;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) m[c * 4 + r] = (float)T.R.m[r][c];
  m[12] = (float)T.t.x;
  m[13] = (float)T.t.y;
  m[14] = (float)T.t.z;
  m[15] = 1.f;
}

}  // namespace hgso
