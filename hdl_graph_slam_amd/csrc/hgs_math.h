// hgs_math.h — small fixed-size linear algebra for the registration kernels (gfx950).
// Everything is HGS_HD so that the same per-item code that runs inside the HIP kernels can be exercised by the
// test-only host harness (tests/emul); the product library never runs these on the CPU.
#pragma once
#include <math.h>
#include <stdint.h>
#include <float.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HGS_HD __host__ __device__ __forceinline__
#else
#define HGS_HD inline
#endif

namespace hgs {

// First statement of a function body: no fused multiply-add contraction inside it, so that the device evaluates the
// expression tree exactly like the (contraction-free) CPU oracle.  Used on the one-thread-per-problem NDT control path,
// whose Newton iteration amplifies last-bit differences (DESIGN.md "NDT conditioning"); never in per-point fp64 code.
#if defined(__clang__)
#define HGS_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define HGS_FP_STRICT
#endif

// ---- float point helpers (bit-exact contract with the oracle: explicit fma chains) -------------------------
struct F3 {
  float x, y, z;
};

HGS_HD float dist2f(const F3& q, float px, float py, float pz) {
  const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// q = T * a with T row-major 3x4 float
HGS_HD F3 transform_point_f(const float* T, float ax, float ay, float az) {
  F3 q;
  q.x = fmaf(T[2], az, fmaf(T[1], ay, fmaf(T[0], ax, T[3])));
  q.y = fmaf(T[6], az, fmaf(T[5], ay, fmaf(T[4], ax, T[7])));
  q.z = fmaf(T[10], az, fmaf(T[9], ay, fmaf(T[8], ax, T[11])));
  return q;
}

// squared distance from q to an axis-aligned box; monotone lower bound of dist2f to any point inside
HGS_HD float box_dist2f(const F3& q, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
  const float dx = fmaxf(fmaxf(mnx - q.x, q.x - mxx), 0.f);
  const float dy = fmaxf(fmaxf(mny - q.y, q.y - mxy), 0.f);
  const float dz = fmaxf(fmaxf(mnz - q.z, q.z - mxz), 0.f);
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// ---- symmetric 3x3 (xx,xy,xz,yy,yz,zz) in double ------------------------------------------------------------
struct Sym3 {
  double xx, xy, xz, yy, yz, zz;
};

HGS_HD Sym3 sym3_inverse(const Sym3& a) {
  const double c00 = a.yy * a.zz - a.yz * a.yz;
  const double c01 = a.xz * a.yz - a.xy * a.zz;
  const double c02 = a.xy * a.yz - a.xz * a.yy;
  const double det = a.xx * c00 + a.xy * c01 + a.xz * c02;
  const double id = 1.0 / det;
  Sym3 r;
  r.xx = c00 * id;
  r.xy = c01 * id;
  r.xz = c02 * id;
  r.yy = (a.xx * a.zz - a.xz * a.xz) * id;
  r.yz = (a.xy * a.xz - a.xx * a.yz) * id;
  r.zz = (a.xx * a.yy - a.xy * a.xy) * id;
  return r;
}

// R C R^T for symmetric C, R row-major 3x3 (double)
HGS_HD Sym3 sym3_rotate(const double* R, const Sym3& c) {
  double W[9];  // W = R * C
  for (int r = 0; r < 3; r++) {
    const double a = R[r * 3 + 0], b = R[r * 3 + 1], d = R[r * 3 + 2];
    W[r * 3 + 0] = a * c.xx + b * c.xy + d * c.xz;
    W[r * 3 + 1] = a * c.xy + b * c.yy + d * c.yz;
    W[r * 3 + 2] = a * c.xz + b * c.yz + d * c.zz;
  }
  Sym3 o;
  o.xx = W[0] * R[0] + W[1] * R[1] + W[2] * R[2];
  o.xy = W[0] * R[3] + W[1] * R[4] + W[2] * R[5];
  o.xz = W[0] * R[6] + W[1] * R[7] + W[2] * R[8];
  o.yy = W[3] * R[3] + W[4] * R[4] + W[5] * R[5];
  o.yz = W[3] * R[6] + W[4] * R[7] + W[5] * R[8];
  o.zz = W[6] * R[6] + W[7] * R[7] + W[8] * R[8];
  return o;
}

// ---- rigid transform: double row-major 3x4 ------------------------------------------------------------------
struct Pose {
  double m[12];
};

HGS_HD Pose pose_identity() {
  Pose p;
  for (int i = 0; i < 12; i++) p.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
  return p;
}
HGS_HD Pose pose_mul(const Pose& a, const Pose& b) {
  Pose r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 4; j++) {
      double s = a.m[i * 4 + 0] * b.m[0 * 4 + j] + a.m[i * 4 + 1] * b.m[1 * 4 + j] + a.m[i * 4 + 2] * b.m[2 * 4 + j];
      if (j == 3) s += a.m[i * 4 + 3];
      r.m[i * 4 + j] = s;
    }
  }
  return r;
}
HGS_HD void pose_to_float(const Pose& p, float* T12) {
  for (int i = 0; i < 12; i++) T12[i] = (float)p.m[i];
}
HGS_HD Pose pose_from_colmajor_f(const float* m16) {
  Pose p;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) p.m[r * 4 + c] = (double)m16[c * 4 + r];
  return p;
}
HGS_HD void pose_to_colmajor_f(const Pose& p, float* m16) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) m16[c * 4 + r] = (float)p.m[r * 4 + c];
  m16[3] = 0.f, m16[7] = 0.f, m16[11] = 0.f, m16[15] = 1.f;
}

// se3_exp(d), d = [omega(3), v(3)] — fast_gicp so3.hpp semantics (quaternion based SO(3) exponential)
HGS_HD Pose se3_exp(const double* d) {
  const double wx = d[0], wy = d[1], wz = d[2];
  const double theta_sq = wx * wx + wy * wy + wz * wz;
  double imag, real;
  if (theta_sq < 1e-10) {
    const double tq = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * tq;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * tq;
  } else {
    const double th = sqrt(theta_sq), half = 0.5 * th;
    imag = sin(half) / th;
    real = cos(half);
  }
  const double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
  Pose p;
  {
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    p.m[0] = 1 - (tyy + tzz), p.m[1] = txy - twz, p.m[2] = txz + twy;
    p.m[4] = txy + twz, p.m[5] = 1 - (txx + tzz), p.m[6] = tyz - twx;
    p.m[8] = txz - twy, p.m[9] = tyz + twx, p.m[10] = 1 - (txx + tyy);
  }
  const double theta = sqrt(theta_sq);
  double V[9];
  if (theta < 1e-10) {
    V[0] = p.m[0], V[1] = p.m[1], V[2] = p.m[2], V[3] = p.m[4], V[4] = p.m[5], V[5] = p.m[6], V[6] = p.m[8], V[7] = p.m[9], V[8] = p.m[10];
  } else {
    const double a = (1.0 - cos(theta)) / theta_sq;
    const double b = (theta - sin(theta)) / (theta_sq * theta);
    // Omega = skew(w), Omega^2 = w w^T - |w|^2 I
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    const double O2[9] = {wx * wx - theta_sq, wx * wy, wx * wz, wx * wy, wy * wy - theta_sq, wy * wz, wx * wz, wy * wz, wz * wz - theta_sq};
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * O[i] + b * O2[i];
  }
  p.m[3] = V[0] * d[3] + V[1] * d[4] + V[2] * d[5];
  p.m[7] = V[3] * d[3] + V[4] * d[4] + V[5] * d[5];
  p.m[11] = V[6] * d[3] + V[7] * d[4] + V[8] * d[5];
  return p;
}

// ---- 6x6 solvers (row-major A[36]) ----------------------------------------------------------------------------
// Symmetric solve via LDL^T with diagonal pivoting (role of Eigen::LDLT in fast_gicp's step_lm).
// The factorisation's working arrays are indexed by the pivot, i.e. dynamically: as locals they live in scratch memory on the device (592 bytes and
// ~300 scratch instructions in each of k_gicp_solve / k_gicp_decide in round 3).  `ws` (kLdlt6Workspace doubles, LDS on the device) holds them instead;
// the arithmetic is the same either way.
constexpr int kLdlt6Workspace = 36 + 36 + 6 + 6 + 6 + 6;
HGS_HD void solve_ldlt6_ws(const double* A_in, const double* b_in, double* x, double* ws) {
  double *A = ws, *L = ws + 36, *D = ws + 72, *y = ws + 78, *z = ws + 84, *permd = ws + 90;  // (the permutation as small integers in doubles: one workspace type)
  for (int i = 0; i < 36; i++) A[i] = A_in[i], L[i] = 0.0;
  for (int i = 0; i < 6; i++) permd[i] = (double)i;
  for (int k = 0; k < 6; k++) {
    int piv = k;
    double best = fabs(A[k * 6 + k]);
    for (int i = k + 1; i < 6; i++) {
      const double v = fabs(A[i * 6 + i]);
      if (v > best) best = v, piv = i;
    }
    if (piv != k) {
      for (int j = 0; j < 6; j++) {
        const double t = A[k * 6 + j];
        A[k * 6 + j] = A[piv * 6 + j];
        A[piv * 6 + j] = t;
      }
      for (int i = 0; i < 6; i++) {
        const double t = A[i * 6 + k];
        A[i * 6 + k] = A[i * 6 + piv];
        A[i * 6 + piv] = t;
      }
      for (int j = 0; j < k; j++) {
        const double t = L[k * 6 + j];
        L[k * 6 + j] = L[piv * 6 + j];
        L[piv * 6 + j] = t;
      }
      const double t = permd[k];
      permd[k] = permd[piv];
      permd[piv] = t;
    }
    D[k] = A[k * 6 + k];
    L[k * 6 + k] = 1.0;
    for (int i = k + 1; i < 6; i++) L[i * 6 + k] = (D[k] != 0.0) ? A[i * 6 + k] / D[k] : 0.0;
    for (int i = k + 1; i < 6; i++)
      for (int j = k + 1; j < 6; j++) A[i * 6 + j] -= L[i * 6 + k] * D[k] * L[j * 6 + k];
  }
  for (int i = 0; i < 6; i++) {
    double s = b_in[(int)permd[i]];
    for (int j = 0; j < i; j++) s -= L[i * 6 + j] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < 6; i++) y[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
    for (int j = i + 1; j < 6; j++) s -= L[j * 6 + i] * z[j];
    z[i] = s;
  }
  for (int i = 0; i < 6; i++) x[(int)permd[i]] = z[i];
}
HGS_HD void solve_ldlt6(const double* A_in, const double* b_in, double* x) {
  double A[36], L[36], D[6], y[6], z[6];
  int perm[6];
  for (int i = 0; i < 36; i++) A[i] = A_in[i], L[i] = 0.0;
  for (int i = 0; i < 6; i++) perm[i] = i;
  for (int k = 0; k < 6; k++) {
    int piv = k;
    double best = fabs(A[k * 6 + k]);
    for (int i = k + 1; i < 6; i++) {
      const double v = fabs(A[i * 6 + i]);
      if (v > best) best = v, piv = i;
    }
    if (piv != k) {
      for (int j = 0; j < 6; j++) {
        const double t = A[k * 6 + j];
        A[k * 6 + j] = A[piv * 6 + j];
        A[piv * 6 + j] = t;
      }
      for (int i = 0; i < 6; i++) {
        const double t = A[i * 6 + k];
        A[i * 6 + k] = A[i * 6 + piv];
        A[i * 6 + piv] = t;
      }
      for (int j = 0; j < k; j++) {
        const double t = L[k * 6 + j];
        L[k * 6 + j] = L[piv * 6 + j];
        L[piv * 6 + j] = t;
      }
      const int t = perm[k];
      perm[k] = perm[piv];
      perm[piv] = t;
    }
    D[k] = A[k * 6 + k];
    L[k * 6 + k] = 1.0;
    for (int i = k + 1; i < 6; i++) L[i * 6 + k] = (D[k] != 0.0) ? A[i * 6 + k] / D[k] : 0.0;
    for (int i = k + 1; i < 6; i++)
      for (int j = k + 1; j < 6; j++) A[i * 6 + j] -= L[i * 6 + k] * D[k] * L[j * 6 + k];
  }
  for (int i = 0; i < 6; i++) {
    double s = b_in[perm[i]];
    for (int j = 0; j < i; j++) s -= L[i * 6 + j] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < 6; i++) y[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
    for (int j = i + 1; j < 6; j++) s -= L[j * 6 + i] * z[j];
    z[i] = s;
  }
  for (int i = 0; i < 6; i++) x[perm[i]] = z[i];
}

// x = pinv(A) b via one-sided Jacobi SVD, singular values <= 6 eps sigma_max dropped (role of Eigen::JacobiSVD
// .solve in ndt_omp's Newton step).
//
// The 15 column pairs of a sweep are visited as a round-robin tournament, 5 rounds of 3 disjoint pairs:
//   (0,5)(1,4)(2,3) | (0,4)(3,5)(1,2) | (0,3)(2,4)(1,5) | (0,2)(1,3)(4,5) | (0,1)(2,5)(3,4)
// Rotations of disjoint pairs touch disjoint columns, so the three of a round commute exactly: this serial code (host,
// tests/emul; the oracle uses the same order) and the device version that runs a round on three lanes (hgs_kernels.hip,
// solve_svd6_wave) produce the same bits.  The rotation chain (3 divisions + 2 square roots in fp64) is the latency of
// ndt_omp's per-iteration solve; in lock-step it is paid 5 instead of 15 times per sweep.
HGS_HD void svd6_pair(int round, int j, int* p, int* q) {
  const int t = round * 3 + j;
  // packed (p << 4 | q), 15 entries
  const unsigned char pq = t == 0 ? 0x05 : t == 1 ? 0x14 : t == 2 ? 0x23 : t == 3 ? 0x04 : t == 4 ? 0x35 : t == 5 ? 0x12 : t == 6 ? 0x03 : t == 7 ? 0x24
                         : t == 8 ? 0x15 : t == 9 ? 0x02 : t == 10 ? 0x13 : t == 11 ? 0x45 : t == 12 ? 0x01 : t == 13 ? 0x25 : 0x34;
  *p = pq >> 4, *q = pq & 15;
}

// One Hestenes rotation of columns p < q of U (and V); false if the pair is already orthogonal to working precision.
// P is `double*` (local arrays on the host) or `volatile double*` (LDS shared by the lanes of a wave on the device).
// The four columns are read up front (24 independent loads: on the device they are all in flight together — LDS round trips,
// not arithmetic, bounded the wave version of round 2), rotated in registers and written back; the arithmetic and its order are
// those of the in-place form.
template <typename P>
HGS_HD bool svd6_rotate(P U, P V, int p, int q) {
  HGS_FP_STRICT
  double up[6], uq[6], vp[6], vq[6];
  for (int k = 0; k < 6; k++) up[k] = U[k * 6 + p], uq[k] = U[k * 6 + q], vp[k] = V[k * 6 + p], vq[k] = V[k * 6 + q];
  double alpha = 0, beta = 0, gamma = 0;
  for (int k = 0; k < 6; k++) {
    alpha += up[k] * up[k];
    beta += uq[k] * uq[k];
    gamma += up[k] * uq[k];
  }
  if (gamma == 0.0 || fabs(gamma) <= DBL_EPSILON * sqrt(alpha * beta)) return false;
  const double zeta = (beta - alpha) / (2.0 * gamma);
  const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
  for (int k = 0; k < 6; k++) {
    U[k * 6 + p] = c * up[k] - s * uq[k];
    U[k * 6 + q] = s * up[k] + c * uq[k];
    V[k * 6 + p] = c * vp[k] - s * vq[k];
    V[k * 6 + q] = s * vp[k] + c * vq[k];
  }
  return true;
}

// x = V diag(1/sigma_j) U_j^T b over the singular values above the threshold (U holds A V = U_unit * sigma).
template <typename P>
HGS_HD void svd6_backsolve(P U, P V, const double* b, double* x) {
  HGS_FP_STRICT
  double sig2[6], smax2 = 0;
  for (int j = 0; j < 6; j++) {
    double s = 0;
    for (int k = 0; k < 6; k++) s += U[k * 6 + j] * U[k * 6 + j];
    sig2[j] = s;
    if (s > smax2) smax2 = s;
  }
  const double thr = 6.0 * DBL_EPSILON * sqrt(smax2);
  for (int k = 0; k < 6; k++) x[k] = 0.0;
  for (int j = 0; j < 6; j++) {
    const double sg = sqrt(sig2[j]);
    if (!(sg > thr) || sg == 0.0) continue;
    double ub = 0;
    for (int k = 0; k < 6; k++) ub += U[k * 6 + j] * b[k];
    const double coef = ub / sig2[j];
    for (int k = 0; k < 6; k++) x[k] += coef * V[k * 6 + j];
  }
}

HGS_HD void solve_svd6(const double* A, const double* b, double* x) {
  double U[36], V[36];
  for (int i = 0; i < 36; i++) U[i] = A[i], V[i] = (i % 7 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int round = 0; round < 5; round++)
      for (int j = 0; j < 3; j++) {
        int p, q;
        svd6_pair(round, j, &p, &q);
        if (svd6_rotate<double*>(U, V, p, q)) rotated = true;
      }
    if (!rotated) break;
  }
  svd6_backsolve<const double*>(U, V, b, x);
}

// ---- octree box: the bounding box of pcl::octree::OctreePointCloud while points are added ---------------------------------
// (MapCloudGenerator::generate, src/hdl_graph_slam/map_cloud_generator.cpp:39-44; PCL 1.10 octree_pointcloud.hpp
// adoptBoundingBoxToPoint / getKeyBitSize / genOctreeKeyforPoint / genLeafNodeCenterFromOctreeKey.)  All in double, unfused.
constexpr double kOctreeMinValue = 1.1920928955078125e-07;  // (double)std::numeric_limits<float>::epsilon()
// first point: min = p - res/2, max = p + res/2, then getKeyBitSize() on the empty tree: >= 2 voxels per axis (depth 1) and the
// box widened symmetrically to that side length — the point ends up on the corner shared by the root's eight voxels
HGS_HD void octree_box_first(const float* p, double res, double* mn, double* mx, int* depth) {
  HGS_FP_STRICT
  unsigned max_voxels = 2u;
  for (int a = 0; a < 3; a++) {
    mn[a] = (double)p[a] - res / 2, mx[a] = (double)p[a] + res / 2;
    const unsigned mk = (unsigned)ceil((mx[a] - mn[a] - kOctreeMinValue) / res);
    max_voxels = mk > max_voxels ? mk : max_voxels;
  }
  const unsigned d = (unsigned)ceil(log2((double)max_voxels) - kOctreeMinValue);
  *depth = (int)(d < 32u ? d : 32u);
  const double side = (double)(1 << *depth) * res;
  for (int a = 0; a < 3; a++) {
    const double oversize = (side - (mx[a] - mn[a])) / 2.0;
    if (oversize > kOctreeMinValue) mn[a] -= oversize, mx[a] += oversize;
  }
}
HGS_HD bool octree_box_violated(const float* p, const double* mn, const double* mx) {
  return (double)p[0] < mn[0] || (double)p[0] >= mx[0] || (double)p[1] < mn[1] || (double)p[1] >= mx[1] || (double)p[2] < mn[2] || (double)p[2] >= mx[2];
}
// one doubling towards p: per axis the box grows downwards unless p violates the upper bound there; gained[a] += 2^old_depth
// where it grew downwards (the old root becomes the child with that bit set)
HGS_HD void octree_box_double(const float* p, double res, double* mn, double* mx, int* depth, unsigned long long* gained) {
  HGS_FP_STRICT
  double side = (double)(1 << *depth) * res;
  for (int a = 0; a < 3; a++) {
    if (!((double)p[a] >= mx[a])) mn[a] -= side, gained[a] += 1ull << *depth;
  }
  *depth += 1;
  side = (double)(1 << *depth) * res - kOctreeMinValue;
  for (int a = 0; a < 3; a++) mx[a] = mn[a] + side;
}
// (x_bit << 2 | y_bit << 1 | z_bit) per level, most significant level first: ascending = the depth-first child order 0..7
HGS_HD unsigned long long octree_interleave(const unsigned long long* key, int depth) {
  unsigned long long m = 0;
  for (int bit = depth - 1; bit >= 0; bit--) m = (m << 3) | (((key[0] >> bit) & 1ull) << 2) | (((key[1] >> bit) & 1ull) << 1) | ((key[2] >> bit) & 1ull);
  return m;
}
HGS_HD void octree_deinterleave(unsigned long long m, int depth, unsigned long long* key) {
  key[0] = key[1] = key[2] = 0ull;
  for (int bit = 0; bit < depth; bit++) {
    const unsigned long long t = (m >> (3 * bit)) & 7ull;
    key[0] |= ((t >> 2) & 1ull) << bit, key[1] |= ((t >> 1) & 1ull) << bit, key[2] |= (t & 1ull) << bit;
  }
}

// ---- deskewing of one sweep point (apps/prefiltering_nodelet.cpp:229-239) ---------------------------------------------
// p' = delta_q.inverse() * p with delta_q = (1, delta_t/2 * w), delta_t = scan_period * i / n and w = -(gyro rate) as floats:
// Eigen's float arithmetic restated (inverse = conjugate / squaredNorm, the (x^2 + z^2) + (y^2 + w^2) reduction;
// q * v = v + w * uv + q.vec x uv with uv = 2 q.vec x v), unfused.
HGS_HD void pf_deskew_point(float wx, float wy, float wz, double scan_period, int i, int n, float* x, float* y, float* z) {
  HGS_FP_STRICT
  const double delta_t = scan_period * (double)i / (double)n;
  const float qw = 1.f, qx = (float)(delta_t / 2.0 * (double)wx), qy = (float)(delta_t / 2.0 * (double)wy), qz = (float)(delta_t / 2.0 * (double)wz);
  const float n2 = (qx * qx + qz * qz) + (qy * qy + qw * qw);
  float ix = 0.f, iy = 0.f, iz = 0.f, iw = 0.f;
  if (n2 > 0.f) ix = -qx / n2, iy = -qy / n2, iz = -qz / n2, iw = qw / n2;
  const float vx = *x, vy = *y, vz = *z;
  float ux = iy * vz - iz * vy, uy = iz * vx - ix * vz, uz = ix * vy - iy * vx;
  ux += ux, uy += uy, uz += uz;
  *x = (vx + iw * ux) + (iy * uz - iz * uy);
  *y = (vy + iw * uy) + (iz * ux - ix * uz);
  *z = (vz + iw * uz) + (ix * uy - iy * ux);
}

// ---- symmetric 3x3 eigen decomposition (cyclic Jacobi), eigenvalues ascending, eigenvectors in columns of V ----
HGS_HD void eig_sym3(const double* A_in, double* w, double* V) {
  HGS_FP_STRICT
  double a[9];
  for (int i = 0; i < 9; i++) a[i] = A_in[i], V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    const double dg = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
    if (off <= 1e-60 || off <= 1e-34 * dg) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a[p * 3 + q] == 0.0) continue;
        const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * a[p * 3 + q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {
          const double akp = a[k * 3 + p], akq = a[k * 3 + q];
          a[k * 3 + p] = c * akp - s * akq;
          a[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
          a[p * 3 + k] = c * apk - s * aqk;
          a[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  double d[3] = {a[0], a[4], a[8]};
  // sort ascending (3 elements), permuting columns of V
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2 - i; j++)
      if (d[j] > d[j + 1]) {
        const double t = d[j];
        d[j] = d[j + 1];
        d[j + 1] = t;
        for (int k = 0; k < 3; k++) {
          const double tv = V[k * 3 + j];
          V[k * 3 + j] = V[k * 3 + j + 1];
          V[k * 3 + j + 1] = tv;
        }
      }
  w[0] = d[0], w[1] = d[1], w[2] = d[2];
}

HGS_HD void mat3_inverse(const double* a, double* r) {
  HGS_FP_STRICT
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double id = 1.0 / (a[0] * c00 + a[1] * c01 + a[2] * c02);
  r[0] = c00 * id;
  r[1] = (a[2] * a[7] - a[1] * a[8]) * id;
  r[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  r[3] = c01 * id;
  r[4] = (a[0] * a[8] - a[2] * a[6]) * id;
  r[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  r[6] = c02 * id;
  r[7] = (a[1] * a[6] - a[0] * a[7]) * id;
  r[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// ---- Hilbert curve index, 16 bits per axis -> 48-bit key (Skilling's transpose algorithm) --------------------
HGS_HD uint64_t hilbert48(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t X[3] = {x, y, z};
  const uint32_t M = 1u << 15;
  for (uint32_t Q = M; Q > 1; Q >>= 1) {
    const uint32_t P = Q - 1;
    for (int i = 0; i < 3; i++) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const uint32_t t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  uint32_t t = 0;
  for (uint32_t Q = M; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  X[0] ^= t, X[1] ^= t, X[2] ^= t;
  uint64_t key = 0;
  for (int b = 15; b >= 0; b--) {
    key = (key << 3) | (uint64_t)((((X[0] >> b) & 1u) << 2) | (((X[1] >> b) & 1u) << 1) | ((X[2] >> b) & 1u));
  }
  return key;
}

}  // namespace hgs
