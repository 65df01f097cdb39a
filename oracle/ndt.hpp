// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp header). Parity unpinned by the reference.
//
// CPU restatement of pclomp::NormalDistributionsTransform + pclomp::VoxelGridCovariance (koide3/ndt_omp,
// cloned at unpinned HEAD by docker/noetic/Dockerfile:14, NOT under /root/reference), the engine
// hdl_graph_slam constructs for registration_method == "NDT_OMP" (the factory default,
// src/hdl_graph_slam/registrations.cpp:26,101-120).  Follows SURVEY.md Appendix A.1:
//   * target voxelisation into Gaussian cells (>= 6 points, eigenvalue floor 0.01*lambda_max, inverse cov),
//   * DIRECT1 / DIRECT7 / KDTREE neighbourhoods,
//   * Magnusson (2009) eq. 6.9-6.13 score / gradient / Hessian with ndt_omp's float per-point temporaries and
//     double accumulators,
//   * Newton step via SVD solve, step length = clamp(|dp|, eps/2, step_size) (the More-Thuente loop of the
//     PCL-1.8-derived code never runs: `interval_converged = (step_max - step_min) > 0`),
//   * convergence: iter > max_iterations || (iter > 0 && step < transformation_epsilon).
#pragma once
#include <cstring>
#include <limits>
#include <unordered_map>
#include <vector>
#include <memory>
#include <omp.h>
#include "linalg.hpp"
#include "kdtree.hpp"
#include "gicp.hpp"  // OCloud, transform_point_f
#include "../include/hgs_registration.h"

namespace hgso {

struct NdtCell {
  int n = 0;
  double sum[3] = {0, 0, 0};
  double sq[3][3] = {{0}};
  V3 mean{0, 0, 0};
  M3 cov = M3::zero(), icov = M3::zero();
  int ijk[3] = {0, 0, 0};
};

class NdtVoxelGrid {
public:
  float inv_leaf = 1.f;
  double leaf = 1.0;
  int min_b[3] = {0, 0, 0}, max_b[3] = {0, 0, 0}, div_b[3] = {1, 1, 1};
  long divb_mul[3] = {1, 1, 1};
  int min_points = 6;
  std::unordered_map<long, NdtCell> leaves;
  std::vector<long> valid_keys;     // cells with n >= min_points that passed the eigen check (sorted by key)
  std::vector<P3f> centroids;       // of valid cells (KDTREE mode)
  KdTree centroid_tree;

  // pclomp::VoxelGridCovariance::applyFilter
  void build(const std::vector<P3f>& pts, double resolution, int min_points_per_voxel) {
    leaves.clear();
    valid_keys.clear();
    centroids.clear();
    leaf = resolution;
    inv_leaf = 1.0f / (float)resolution;
    min_points = min_points_per_voxel;
    if (pts.empty()) return;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (const P3f& p : pts) {
      mn[0] = std::min(mn[0], p.x), mn[1] = std::min(mn[1], p.y), mn[2] = std::min(mn[2], p.z);
      mx[0] = std::max(mx[0], p.x), mx[1] = std::max(mx[1], p.y), mx[2] = std::max(mx[2], p.z);
    }
    for (int d = 0; d < 3; d++) {
      min_b[d] = (int)std::floor(mn[d] * inv_leaf);
      max_b[d] = (int)std::floor(mx[d] * inv_leaf);
      div_b[d] = max_b[d] - min_b[d] + 1;
    }
    divb_mul[0] = 1, divb_mul[1] = div_b[0], divb_mul[2] = (long)div_b[0] * div_b[1];
    for (const P3f& p : pts) {
      const int ijk[3] = {cell_coord(p.x), cell_coord(p.y), cell_coord(p.z)};
      NdtCell& c = leaves[key_of(ijk)];
      if (c.n == 0) c.ijk[0] = ijk[0], c.ijk[1] = ijk[1], c.ijk[2] = ijk[2];
      const double v[3] = {p.x, p.y, p.z};
      for (int r = 0; r < 3; r++) {
        c.sum[r] += v[r];
        for (int s = 0; s < 3; s++) c.sq[r][s] += v[r] * v[s];
      }
      c.n++;
    }
    for (auto& kv : leaves) {
      NdtCell& c = kv.second;
      const double n = (double)c.n;
      const V3 pt_sum{c.sum[0], c.sum[1], c.sum[2]};
      // leaf.mean_ /= leaf.nr_points: a true division per component in the Eigen 3.3 of the ROS distributions
      // hdl_graph_slam builds on (Eigen 3.2 multiplied by the reciprocal, an ulp apart)
      c.mean = V3{pt_sum.x / n, pt_sum.y / n, pt_sum.z / n};
      if (c.n < min_points) continue;
      const double ps[3] = {pt_sum.x, pt_sum.y, pt_sum.z}, mu[3] = {c.mean.x, c.mean.y, c.mean.z};
      for (int r = 0; r < 3; r++)
        for (int s = 0; s < 3; s++) c.cov.m[r][s] = (c.sq[r][s] - 2.0 * (ps[r] * mu[s])) / n + mu[r] * mu[s];
      c.cov = ((n - 1.0) / n) * c.cov;
      double ev[3];
      M3 evec;
      // SelfAdjointEigenSolver reads the lower triangle only: symmetrise from it.
      M3 sym = c.cov;
      for (int r = 0; r < 3; r++)
        for (int s = r + 1; s < 3; s++) sym.m[r][s] = sym.m[s][r];
      eig_sym3(sym, ev, evec);
      if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) {
        c.n = -1;
        continue;
      }
      const double min_ev = 0.01 * ev[2];
      if (ev[0] < min_ev) {
        ev[0] = min_ev;
        if (ev[1] < min_ev) ev[1] = min_ev;
        M3 L = M3::zero();
        L.m[0][0] = ev[0], L.m[1][1] = ev[1], L.m[2][2] = ev[2];
        c.cov = evec * L * inverse(evec);
      }
      c.icov = inverse(c.cov);
      bool bad = false;
      for (int r = 0; r < 3; r++)
        for (int s = 0; s < 3; s++)
          if (std::isinf(c.icov.m[r][s])) bad = true;
      if (bad) c.n = -1;
    }
    for (auto& kv : leaves)
      if (kv.second.n >= min_points) valid_keys.push_back(kv.first);
    std::sort(valid_keys.begin(), valid_keys.end());
    for (long k : valid_keys) {
      const NdtCell& c = leaves[k];
      centroids.push_back({(float)c.mean.x, (float)c.mean.y, (float)c.mean.z});
    }
    centroid_tree.build(centroids);
  }

  int cell_coord(float v) const { return (int)std::floor(v * inv_leaf); }
  long key_of(const int ijk[3]) const { return (long)(ijk[0] - min_b[0]) * divb_mul[0] + (long)(ijk[1] - min_b[1]) * divb_mul[1] + (long)(ijk[2] - min_b[2]) * divb_mul[2]; }

  // getNeighborhoodAtPoint1/7 ; KDTREE = radiusSearch(resolution) on valid-cell centroids
  int neighborhood(const P3f& q, int method, const NdtCell** out /* >= 7, KDTREE: cap */, int cap) const {
    int cnt = 0;
    if (method == HGS_KDTREE) {
      std::vector<Neighbor> nb(cap);
      const float r2 = (float)(leaf * leaf);
      const int found = centroid_tree.knn(q, cap, nb.data(), r2);
      for (int i = 0; i < found; i++) out[cnt++] = &leaves.at(valid_keys[nb[i].idx]);
      return cnt;
    }
    static const int off7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    const int nn = method == HGS_DIRECT1 ? 1 : 7;
    const int ijk[3] = {cell_coord(q.x), cell_coord(q.y), cell_coord(q.z)};
    for (int i = 0; i < nn; i++) {
      const int c[3] = {ijk[0] + off7[i][0], ijk[1] + off7[i][1], ijk[2] + off7[i][2]};
      bool inside = true;
      for (int d = 0; d < 3; d++) inside = inside && c[d] >= min_b[d] && c[d] <= max_b[d];
      if (!inside) continue;
      auto it = leaves.find(key_of(c));
      if (it != leaves.end() && it->second.n >= min_points) out[cnt++] = &it->second;
    }
    return cnt;
  }
};

// More-Thuente line search state (Sun & Yuan, as in pcl::NormalDistributionsTransform::computeStepLengthMT /
// trialValueSelectionMT / updateIntervalMT), driven one trial at a time: wants_another_trial() consumes the value and
// directional derivative of phi at the trial step a_t and, if the search goes on, leaves the next trial step in a_next.
struct MoreThuente {
  double phi_0, d_phi_0, a_l, f_l, g_l, a_u, f_u, g_u, a_next;
  bool open_interval, interval_converged;
  int trials;
  static constexpr double mu = 1e-4, nu = 0.9;
  static constexpr int max_trials = 10;
  void start(double phi0, double dphi0) {
    phi_0 = phi0, d_phi_0 = dphi0;
    a_l = a_u = 0;
    f_l = f_u = 0;                          // psi(0) = 0
    g_l = g_u = dphi0 - mu * dphi0;         // dpsi(0)
    open_interval = true, interval_converged = false, trials = 0, a_next = 0;
  }
  static double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
    if (f_t > f_l) {  // case 1
      const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
      const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
      const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
      return std::fabs(a_c - a_l) < std::fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
    }
    if (g_t * g_l < 0) {  // case 2
      const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
      const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
      const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
      return std::fabs(a_c - a_t) >= std::fabs(a_s - a_t) ? a_c : a_s;
    }
    if (std::fabs(g_t) <= std::fabs(g_l)) {  // case 3
      const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
      const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
      const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
      const double a_n = std::fabs(a_c - a_t) < std::fabs(a_s - a_t) ? a_c : a_s;
      return a_t > a_l ? std::min(a_t + 0.66 * (a_u - a_t), a_n) : std::max(a_t + 0.66 * (a_u - a_t), a_n);
    }
    const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = std::sqrt(z * z - g_t * g_u);  // case 4
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
  }
  bool update_interval(double a_t, double f_t, double g_t) {
    if (f_t > f_l) {
      a_u = a_t, f_u = f_t, g_u = g_t;
      return false;
    }
    if (g_t * (a_l - a_t) > 0) {
      a_l = a_t, f_l = f_t, g_l = g_t;
      return false;
    }
    if (g_t * (a_l - a_t) < 0) {
      a_u = a_l, f_u = f_l, g_u = g_l;
      a_l = a_t, f_l = f_t, g_l = g_t;
      return false;
    }
    return true;
  }
  bool wants_another_trial(double a_t, double phi_t, double d_phi_t, double step_max, double step_min) {
    const double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
    if (trials == 0) {
      interval_converged = (step_max - step_min) < 0;
    } else {
      // bookkeeping of the trial just evaluated (the tail of upstream's loop body)
      if (open_interval && psi_t <= 0 && d_psi_t >= 0) {
        open_interval = false;
        f_l += phi_0 - mu * d_phi_0 * a_l, g_l += mu * d_phi_0;
        f_u += phi_0 - mu * d_phi_0 * a_u, g_u += mu * d_phi_0;
      }
      interval_converged = open_interval ? update_interval(a_t, psi_t, d_psi_t) : update_interval(a_t, phi_t, d_phi_t);
    }
    if (interval_converged || trials >= max_trials || (psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) return false;
    a_next = open_interval ? trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t) : trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_next = std::min(a_next, step_max);
    a_next = std::max(a_next, step_min);
    // (not in PCL) the clamped trial is the point just evaluated, or the interpolation broke down (a_t == a_l gives 0/0):
    // the search cannot make progress — PCL would re-evaluate the same point until max_trials
    if (a_next == a_t || a_next != a_next) return false;
    trials++;
    return true;
  }
};

// Eigen's MatrixBase::eulerAngles(0,1,2) (Graphics Gems IV variant) on a float rotation, as ndt_omp uses to
// initialise p from the guess.
// Float transcendental = the double function rounded once (a correctly rounded float result, which is what a
// conforming atan2f/sinf/cosf returns up to its <= 1 ulp slack); written this way so that glibc here and the device
// libm in the HIP path agree bit-for-bit — near pi one float ulp is 2.4e-7 rad, enough to change the NDT trajectory.
inline float atan2_f(float y, float x) { return (float)std::atan2((double)y, (double)x); }
inline float sin_f(float a) { return (float)std::sin((double)a); }
inline float cos_f(float a) { return (float)std::cos((double)a); }

inline void euler_angles_xyz_f(const float R[3][3], float out[3]) {
  // i=0, j=1, k=2, "odd" = 0
  float r0 = atan2_f(R[1][2], R[2][2]);
  const float c2 = std::sqrt(R[0][0] * R[0][0] + R[0][1] * R[0][1]);
  float r1;
  if (r0 > 0.f) {
    r0 -= (float)M_PI;
    r1 = atan2_f(-R[0][2], -c2);
  } else {
    r1 = atan2_f(-R[0][2], c2);
  }
  const float s1 = sin_f(r0), c1 = cos_f(r0);
  const float r2 = atan2_f(s1 * R[2][0] - c1 * R[1][0], c1 * R[1][1] - s1 * R[2][1]);
  out[0] = -r0, out[1] = -r1, out[2] = -r2;
}

// T(p) = Trans(t) * Rx * Ry * Rz  (double, then handed over as float like upstream's Matrix4f)
inline Iso ndt_pose_from_p(const double p[6]) {
  const double cx = std::cos(p[3]), sx = std::sin(p[3]), cy = std::cos(p[4]), sy = std::sin(p[4]), cz = std::cos(p[5]), sz = std::sin(p[5]);
  Iso T;
  T.R.m[0][0] = cy * cz, T.R.m[0][1] = -cy * sz, T.R.m[0][2] = sy;
  T.R.m[1][0] = cx * sz + sx * sy * cz, T.R.m[1][1] = cx * cz - sx * sy * sz, T.R.m[1][2] = -sx * cy;
  T.R.m[2][0] = sx * sz - cx * sy * cz, T.R.m[2][1] = cx * sy * sz + sx * cz, T.R.m[2][2] = cx * cy;
  T.t = {p[0], p[1], p[2]};
  return T;
}

// exp() as a FIXED sequence of IEEE-754 operations (nothing the math library is free to round differently): Cody-Waite
// range reduction by ln 2 (split hi/lo), degree-13 Taylor polynomial by Horner with fma, exact scaling by 2^k.  ndt_omp
// calls float exp(); the restatement evaluates this double sequence and rounds once to float — correctly rounded except for
// ~1e-8 of the arguments, and the same bits on every machine, which libm's exp is not.
inline double exp_fixed(double x) {
  if (x != x) return x;
  if (x < -110.0) return 0.0;                                       // float result: below the smallest subnormal
  if (x > 90.0) return std::numeric_limits<double>::infinity();     // float result: above FLT_MAX
  const double k = std::nearbyint(x * 0x1.71547652b82fep+0);        // round half to even (default rounding mode)
  const double r = std::fma(-k, 0x1.a39ef35793c76p-33, std::fma(-k, 0x1.62e42fee00000p-1, x));
  static const double inv_fact[14] = {1.0, 1.0, 0.5, 0x1.5555555555555p-3, 0x1.5555555555555p-5, 0x1.1111111111111p-7, 0x1.6c16c16c16c17p-10,
                                      0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-16, 0x1.71de3a556c734p-19, 0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26,
                                      0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33};
  double p = inv_fact[13];
  for (int n = 12; n >= 0; n--) p = std::fma(p, r, inv_fact[n]);
  return std::ldexp(p, (int)k);
}

// Order-independent total of the per-point score / gradient / Hessian contributions ("exact" sum mode).  ndt_omp adds the N
// per-point doubles serially (sum mode 0 below restates that); any parallel implementation re-associates the sum, and the
// iteration amplifies the resulting last-bit differences on weakly constrained scans.  Definition used by sum mode 1 — and by
// the device backend under test, which has no other mode: every per-point double t is cut on a fixed binary grid into
//   q0 = RN(t / 2^(E+50))  and  q1 = RN((t - q0 2^(E+50)) / 2^E)            (two 50-bit integers; bits below 2^E rounded off)
// the q's are added as integers and V = (sum q0) 2^50 + (sum q1) is rounded to double once: total = RN(V) 2^E.
// E = -47 for Hessian entries (|t| < 2^52), -51 for the gradient (|t| < 2^48), -59 for the score (|t| < 2^40); a term outside
// its range, or NaN, makes every total of the pass NaN.
struct ExactSum {
  __int128 q0 = 0, q1 = 0;
  int E = 0;
  bool bad = false;
  static constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
  static long long chunk(double magic_added) {
    unsigned long long b;
    std::memcpy(&b, &magic_added, 8);
    return (long long)(b & 0xfffffffffffffull) - (1ll << 51);
  }
  void add(double t) {
    if (!(std::fabs(t) < std::ldexp(1.0, E + 99))) {
      bad = true;
      return;
    }
    const double a = std::fma(t, std::ldexp(1.0, -(E + 50)), kMagic);
    const double r = std::fma(-(a - kMagic), std::ldexp(1.0, E + 50), t);
    const double b = std::fma(r, std::ldexp(1.0, -E), kMagic);
    q0 += chunk(a), q1 += chunk(b);
  }
  double total() const {
    const __int128 v = q0 * ((__int128)1 << 50) + q1;
    if (v == 0) return 0.0;
    const bool neg = v < 0;
    unsigned __int128 u = neg ? (unsigned __int128)0 - (unsigned __int128)v : (unsigned __int128)v;
    int hb = 127;
    while (!((u >> hb) & 1)) hb--;
    double d;
    if (hb <= 52) {
      d = (double)(unsigned long long)u;
    } else {
      const int sh = hb - 52;
      unsigned long long m = (unsigned long long)(u >> sh);
      const unsigned __int128 rem = u & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
      if (rem > half || (rem == half && (m & 1ull))) m++;  // round half to even
      d = std::ldexp((double)m, sh);
    }
    d = std::ldexp(d, E);
    return neg ? -d : d;
  }
};

struct NdtTraceEntry {
  double p[6];
  double score;
  double step;
};

class NdtOmp {
public:
  explicit NdtOmp(const hgs_params& p) : prm(p) {}
  hgs_params prm;
  std::shared_ptr<OCloud> source, target;
  NdtVoxelGrid grid;
  bool grid_valid = false;
  std::vector<NdtTraceEntry> trace;
  double gauss_d1 = 0, gauss_d2 = 0;
  int derivative_passes = 0;
  int sum_mode = 0;  // 0: serial double sum over the points (ndt_omp); 1: ExactSum (order-independent; what the device computes); 2: per-thread accumulators ("optimised-CPU" baseline variant)

  void set_target(std::shared_ptr<OCloud> t) {
    target = t;
    grid.build(t->pts, prm.resolution, prm.ndt_min_points_per_voxel);
    grid_valid = true;
  }

  void init_gauss() {
    const double c1 = 10.0 * (1 - prm.ndt_outlier_ratio);
    const double c2 = prm.ndt_outlier_ratio / std::pow(prm.resolution, 3);
    const double d3 = -std::log(c2);
    gauss_d1 = -std::log(c1 + c2) - d3;
    gauss_d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / gauss_d1);
  }

  // computeAngleDerivatives: rows of j_ang (8x3) and h_ang (15x3 -> stored 16), float like ndt_omp
  void angle_derivatives(const double p[6], float j_ang[8][3], float h_ang[16][3]) const {
    double cx, cy, cz, sx, sy, sz;
    if (std::fabs(p[3]) < 10e-5) cx = 1.0, sx = 0.0; else cx = std::cos(p[3]), sx = std::sin(p[3]);
    if (std::fabs(p[4]) < 10e-5) cy = 1.0, sy = 0.0; else cy = std::cos(p[4]), sy = std::sin(p[4]);
    if (std::fabs(p[5]) < 10e-5) cz = 1.0, sz = 0.0; else cz = std::cos(p[5]), sz = std::sin(p[5]);
    const double j[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy}, {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},
                            {-sy * cz, sy * sz, cy},                                     {sx * cy * cz, -sx * cy * sz, sx * sy},
                            {-cx * cy * cz, cx * cy * sz, -cx * sy},                      {-cy * sz, -cy * cz, 0},
                            {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0},         {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};
    const double hd1z = prm.ndt_upstream_hd1_sign ? sy : -sy;
    const double h[16][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy},   // a2
                             {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},  // a3
                             {cx * cy * cz, -cx * cy * sz, cx * sy},                        // b2
                             {sx * cy * cz, -sx * cy * sz, sx * sy},                        // b3
                             {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0},          // c2
                             {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0},          // c3
                             {-cy * cz, cy * sz, hd1z},                                     // d1
                             {-sx * sy * cz, sx * sy * sz, sx * cy},                        // d2
                             {cx * sy * cz, -cx * sy * sz, -cx * cy},                       // d3
                             {sy * sz, sy * cz, 0},                                         // e1
                             {-sx * cy * sz, -sx * cy * cz, 0},                             // e2
                             {cx * cy * sz, cx * cy * cz, 0},                               // e3
                             {-cy * cz, cy * sz, 0},                                        // f1
                             {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0},         // f2
                             {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0},         // f3
                             {0, 0, 0}};
    for (int r = 0; r < 8; r++)
      for (int c = 0; c < 3; c++) j_ang[r][c] = (float)j[r][c];
    for (int r = 0; r < 16; r++)
      for (int c = 0; c < 3; c++) h_ang[r][c] = (float)h[r][c];
  }

  // computeDerivatives: source transformed (float) by T(p); returns score, fills g,H
  double derivatives(const double p[6], V6& g, M6& H, bool compute_hessian = true) {
    derivative_passes++;
    float j_ang[8][3], h_ang[16][3];
    angle_derivatives(p, j_ang, h_ang);
    const Iso T = ndt_pose_from_p(p);
    float Tf[12];
    iso_to_rowmajor_f(T, Tf);
    const int n = (int)source->pts.size();
    // sum modes 0 / 1 keep ndt_omp's N-long per-point result arrays (344 bytes per point written and read back per pass); mode 2 — the
    // "optimised-CPU" baseline variant of SURVEY 8d, never the parity reference — adds into one accumulator set per thread instead
    const bool per_thread = sum_mode == 2;
    const int nt = per_thread ? omp_get_max_threads() : 0;
    std::vector<double> scores(per_thread ? nt : n, 0.0);
    std::vector<V6> gs(per_thread ? nt : n, V6::zero());
    std::vector<M6> Hs(per_thread ? nt : n, M6::zero());
    const float d1 = (float)gauss_d1, d2 = (float)gauss_d2;
    const int method = prm.neighbor_search;
#pragma omp parallel
    {
    double t_score = 0;  // (sum mode 2) this thread's totals: private, written once to the shared arrays at the end
    V6 t_g = V6::zero();
    M6 t_H = M6::zero();
#pragma omp for schedule(guided, 8) nowait
    for (int idx = 0; idx < n; idx++) {
      V6 gp = V6::zero();
      M6 Hp = M6::zero();
      double sp = 0;
      const P3f x = source->pts[idx];
      const P3f xt = transform_point_f(Tf, x);
      const NdtCell* cells[64];
      const int nc = grid.neighborhood(xt, method, cells, 64);
      // computePointDerivatives (float): point_gradient (3x6), point_hessian (18x6 blocks)
      float xj[8], xh[16];
      for (int r = 0; r < 8; r++) xj[r] = j_ang[r][0] * x.x + j_ang[r][1] * x.y + j_ang[r][2] * x.z;
      for (int r = 0; r < 16; r++) xh[r] = h_ang[r][0] * x.x + h_ang[r][1] * x.y + h_ang[r][2] * x.z;
      float pg[3][6] = {{0}};
      pg[0][0] = pg[1][1] = pg[2][2] = 1.f;
      pg[1][3] = xj[0], pg[2][3] = xj[1];
      pg[0][4] = xj[2], pg[1][4] = xj[3], pg[2][4] = xj[4];
      pg[0][5] = xj[5], pg[1][5] = xj[6], pg[2][5] = xj[7];
      // ph[i][j] = d2 x'/dp_i dp_j (3-vector), non-zero for i,j in 3..5
      float ph[6][6][3] = {{{0}}};
      const float a[3] = {0.f, xh[0], xh[1]}, b[3] = {0.f, xh[2], xh[3]}, c[3] = {0.f, xh[4], xh[5]};
      const float d[3] = {xh[6], xh[7], xh[8]}, e[3] = {xh[9], xh[10], xh[11]}, f[3] = {xh[12], xh[13], xh[14]};
      for (int k = 0; k < 3; k++) {
        ph[3][3][k] = a[k];
        ph[3][4][k] = ph[4][3][k] = b[k];
        ph[3][5][k] = ph[5][3][k] = c[k];
        ph[4][4][k] = d[k];
        ph[4][5][k] = ph[5][4][k] = e[k];
        ph[5][5][k] = f[k];
      }
      for (int ci = 0; ci < nc; ci++) {
        const NdtCell& cell = *cells[ci];
        // x_trans (double) = transformed point - mean, then cast to float (updateDerivatives)
        const double qd[3] = {(double)xt.x - cell.mean.x, (double)xt.y - cell.mean.y, (double)xt.z - cell.mean.z};
        const float q[3] = {(float)qd[0], (float)qd[1], (float)qd[2]};
        float ci3[3][3];
        for (int r = 0; r < 3; r++)
          for (int s = 0; s < 3; s++) ci3[r][s] = (float)cell.icov.m[r][s];
        float qC[3];  // q^T C^-1 (row)
        for (int s = 0; s < 3; s++) qC[s] = q[0] * ci3[0][s] + q[1] * ci3[1][s] + q[2] * ci3[2][s];
        const float qCq = qC[0] * q[0] + qC[1] * q[1] + qC[2] * q[2];
        // upstream: float exp(); evaluated as exp_fixed in double and rounded once (same bits on every machine)
        float e_x_cov_x = (float)exp_fixed((double)(-d2 * qCq * 0.5f));
        const float score_inc = -d1 * e_x_cov_x;
        e_x_cov_x = d2 * e_x_cov_x;
        if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
        e_x_cov_x *= d1;
        sp += score_inc;
        float Cpg[3][6];  // C^-1 * point_gradient
        for (int r = 0; r < 3; r++)
          for (int k = 0; k < 6; k++) Cpg[r][k] = ci3[r][0] * pg[0][k] + ci3[r][1] * pg[1][k] + ci3[r][2] * pg[2][k];
        float qCpg[6];
        for (int k = 0; k < 6; k++) qCpg[k] = q[0] * Cpg[0][k] + q[1] * Cpg[1][k] + q[2] * Cpg[2][k];
        for (int k = 0; k < 6; k++) gp.v[k] += (double)(e_x_cov_x * qCpg[k]);
        if (compute_hessian) {
          for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
              const float qCh = qC[0] * ph[i][j][0] + qC[1] * ph[i][j][1] + qC[2] * ph[i][j][2];
              const float pgCpg = pg[0][j] * Cpg[0][i] + pg[1][j] * Cpg[1][i] + pg[2][j] * Cpg[2][i];
              Hp.m[i][j] += (double)(e_x_cov_x * (-d2 * qCpg[i] * qCpg[j] + qCh + pgCpg));
            }
        }
      }
      if (per_thread) {
        t_score += sp;
        for (int r = 0; r < 6; r++) {
          t_g.v[r] += gp.v[r];
          for (int c = 0; c < 6; c++) t_H.m[r][c] += Hp.m[r][c];
        }
      } else {
        scores[idx] = sp;
        gs[idx] = gp;
        Hs[idx] = Hp;
      }
    }
    if (per_thread) {
      const int t = omp_get_thread_num();
      scores[t] = t_score, gs[t] = t_g, Hs[t] = t_H;
    }
    }  // omp parallel
    double score = 0;
    g = V6::zero();
    H = M6::zero();
    if (per_thread) {  // the threads' totals in thread order (the result depends on the schedule in the last bits: a baseline variant, not a reference)
      for (int t = 0; t < nt; t++) {
        score += scores[t];
        for (int r = 0; r < 6; r++) {
          g.v[r] += gs[t].v[r];
          for (int c = 0; c < 6; c++) H.m[r][c] += Hs[t].m[r][c];
        }
      }
      return score;
    }
    if (sum_mode == 1) {
      ExactSum es, eg[6], eH[6][6];
      es.E = -59;
      for (int r = 0; r < 6; r++) {
        eg[r].E = -51;
        for (int c = 0; c < 6; c++) eH[r][c].E = -47;
      }
      for (int i = 0; i < n; i++) {
        es.add(scores[i]);
        for (int r = 0; r < 6; r++) {
          eg[r].add(gs[i].v[r]);
          for (int c = 0; c < 6; c++) eH[r][c].add(Hs[i].m[r][c]);
        }
      }
      bool bad = es.bad;
      for (int r = 0; r < 6; r++) {
        bad = bad || eg[r].bad;
        for (int c = 0; c < 6; c++) bad = bad || eH[r][c].bad;
      }
      const double nan = std::numeric_limits<double>::quiet_NaN();
      score = bad ? nan : es.total();
      for (int r = 0; r < 6; r++) {
        g.v[r] = bad ? nan : eg[r].total();
        for (int c = 0; c < 6; c++) H.m[r][c] = bad ? nan : eH[r][c].total();
      }
      return score;
    }
    for (int i = 0; i < n; i++) {  // serial, order-invariant sum like upstream
      score += scores[i];
      for (int r = 0; r < 6; r++) {
        g.v[r] += gs[i].v[r];
        for (int c = 0; c < 6; c++) H.m[r][c] += Hs[i].m[r][c];
      }
    }
    return score;
  }

  void align(const float guess[16], hgs_result* out) {
    trace.clear();
    derivative_passes = 0;
    init_gauss();
    // p from the guess: translation + eulerAngles(0,1,2) of the float rotation
    float R[3][3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) R[r][c] = guess[c * 4 + r];
    float eul[3];
    euler_angles_xyz_f(R, eul);
    double p[6] = {guess[12], guess[13], guess[14], eul[0], eul[1], eul[2]};
    V6 g;
    M6 H;
    double score = derivatives(p, g, H);
    bool converged = false;
    int nr_iterations = 0;
    const int n = (int)source->pts.size();
    Iso finalT = iso_from_colmajor_f(guess);
    while (!converged) {
      V6 ng;
      for (int k = 0; k < 6; k++) ng.v[k] = -g.v[k];
      V6 dp = solve_svd6(H, ng);
      double dpn = norm(dp);
      if (dpn == 0 || dpn != dpn) {
        converged = dpn == dpn;
        break;
      }
      for (double& v : dp.v) v /= dpn;
      // computeStepLengthMT
      double d_phi_0 = -dot(g, dp);
      double a_t;
      if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) {
          a_t = 0;
          goto step_done;
        }
        d_phi_0 *= -1;
        for (double& v : dp.v) v *= -1;
      }
      {
        const double step_max = prm.ndt_step_size, step_min = prm.transformation_epsilon / 2;
        a_t = std::min(dpn, step_max);
        a_t = std::max(a_t, step_min);
        double xt[6];
        for (int k = 0; k < 6; k++) xt[k] = p[k] + dp.v[k] * a_t;
        finalT = ndt_pose_from_p(xt);
        const double phi_0 = -score;
        score = derivatives(xt, g, H);
        if (prm.ndt_line_search) {
          // the More-Thuente loop as PCL >= 1.9 runs it (interval_converged = (step_max - step_min) < 0); every trial is a full
          // derivative pass, so the Hessian of the accepted trial needs no separate computeHessian
          MoreThuente mt;
          mt.start(phi_0, d_phi_0);
          double phi_t = -score, d_phi_t = -dot(g, dp);
          while (mt.wants_another_trial(a_t, phi_t, d_phi_t, step_max, step_min)) {
            a_t = mt.a_next;
            for (int k = 0; k < 6; k++) xt[k] = p[k] + dp.v[k] * a_t;
            finalT = ndt_pose_from_p(xt);
            score = derivatives(xt, g, H);
            phi_t = -score, d_phi_t = -dot(g, dp);
          }
        }
      }
    step_done:
      for (int k = 0; k < 6; k++) p[k] += dp.v[k] * a_t;
      if (nr_iterations > prm.max_iterations || (nr_iterations && std::fabs(a_t) < prm.transformation_epsilon)) converged = true;
      nr_iterations++;
      NdtTraceEntry te;
      for (int k = 0; k < 6; k++) te.p[k] = p[k];
      te.score = score, te.step = a_t;
      trace.push_back(te);
    }
    final_T = finalT;
    iso_to_colmajor_f(finalT, out->final_transformation);
    out->converged = converged ? 1 : 0;
    out->iterations = nr_iterations;
    // trans_probability_ = score / input_->points.size(): the size of the cloud as handed in, non-finite points included
    out->error = source->n_input > 0 ? score / (double)source->n_input : 0.0;
    out->fitness_score = std::numeric_limits<double>::quiet_NaN();
    out->num_inliers = 0;
    out->candidate_id = 0;
    out->lm_tries = derivative_passes;
    out->reserved = 0;
  }
  Iso final_T = Iso::identity();
};

}  // namespace hgso
