// hgs_ndt.h — per-cell / per-point arithmetic and the Newton state machine of the NDT engine
// (pclomp::NormalDistributionsTransform + VoxelGridCovariance; reference call site
// src/hdl_graph_slam/registrations.cpp:101-120).  HGS_HD like hgs_gicp.h.
#pragma once
#include <string.h>
#include "hgs_bvh.h"

namespace hgs {

// One Gaussian cell in HBM: 64 bytes (algorithmic 40 B: mean 12, icov 24, n 4).  The mean stays in double — the
// residual x' - mean is formed in double and rounded once, exactly as ndt_omp does, which keeps the Newton
// trajectory on the oracle's; the inverse covariance is consumed in float upstream, so it is stored in float.
//   v0 = (icov.xx, icov.xy, icov.xz, icov.yy)  v1 = (icov.yz, icov.zz, n, key)  mean[3], pad
struct alignas(16) NdtCellRec {
  Float4 v0, v1;
  double mean[3];
  double pad;
};

struct NdtGrid {
  const int* hash_keys;       // open addressing, -1 = empty; key = linear cell index inside the bounding grid
  const int* hash_vals;       // -> index into cells
  const NdtCellRec* cells;
  int hash_mask;              // capacity - 1 (power of two)
  int min_b[3], max_b[3];
  int div_mul[3];             // (1, div.x, div.x*div.y)
  float inv_leaf;
};

HGS_HD uint32_t ndt_hash(int key) { return (uint32_t)key * 2654435761u; }

HGS_HD int ndt_lookup(const NdtGrid& g, int cx, int cy, int cz) {
  if (cx < g.min_b[0] || cx > g.max_b[0] || cy < g.min_b[1] || cy > g.max_b[1] || cz < g.min_b[2] || cz > g.max_b[2]) return -1;
  const int key = (cx - g.min_b[0]) * g.div_mul[0] + (cy - g.min_b[1]) * g.div_mul[1] + (cz - g.min_b[2]) * g.div_mul[2];
  uint32_t slot = (ndt_hash(key) >> 7) & (uint32_t)g.hash_mask;
  for (;;) {
    const int k = g.hash_keys[slot];
    if (k == key) return g.hash_vals[slot];
    if (k == -1) return -1;
    slot = (slot + 1) & (uint32_t)g.hash_mask;
  }
}

// VoxelGridCovariance second pass for one cell: from n, sum p, sum p p^T (double) to mean / inverse covariance.
// Returns false if the cell is rejected (fewer than min_points, bad eigenvalues, non-finite inverse).
HGS_HD bool ndt_finalize_cell(int n, const double* sum, const Sym3& sq, int min_points, double* mean, Sym3* icov) {
  HGS_FP_STRICT
  const double dn = (double)n;
  mean[0] = sum[0] / dn, mean[1] = sum[1] / dn, mean[2] = sum[2] / dn;
  if (n < min_points) return false;
  // cov = (sum pp^T - 2 sum_p mean^T)/n + mean mean^T, then *(n-1)/n  (upstream's single-pass form).  sum_p mean^T is not
  // symmetric in floating point: the eigen-solver reads the LOWER triangle (as SelfAdjointEigenSolver does), the inverse
  // below takes the full matrix.  Only the sign of a ~0 eigenvalue of a degenerate cell (6 collinear returns of one
  // vertical structure) depends on this, but that sign decides whether the cell exists.
  const double f = (dn - 1.0) / dn;
  const double q[9] = {sq.xx, sq.xy, sq.xz, sq.xy, sq.yy, sq.yz, sq.xz, sq.yz, sq.zz};
  double C[9], S[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[r * 3 + c] = ((q[r * 3 + c] - 2.0 * (sum[r] * mean[c])) / dn + mean[r] * mean[c]) * f;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) S[r * 3 + c] = r >= c ? C[r * 3 + c] : C[c * 3 + r];
  double w[3], V[9];
  eig_sym3(S, w, V);
  if (w[0] < 0 || w[1] < 0 || w[2] <= 0) return false;
  const double mn = 0.01 * w[2];
  if (w[0] < mn) {
    w[0] = mn;
    if (w[1] < mn) w[1] = mn;
    // cov = V diag(w) V^-1
    double Vi[9], VL[9];
    mat3_inverse(V, Vi);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) VL[r * 3 + c] = V[r * 3 + c] * w[c];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) C[r * 3 + c] = VL[r * 3 + 0] * Vi[0 * 3 + c] + VL[r * 3 + 1] * Vi[1 * 3 + c] + VL[r * 3 + 2] * Vi[2 * 3 + c];
  }
  double Ci[9];
  mat3_inverse(C, Ci);
  for (int i = 0; i < 9; i++)
    if (isinf(Ci[i])) return false;
  icov->xx = Ci[0], icov->xy = Ci[1], icov->xz = Ci[2], icov->yy = Ci[4], icov->yz = Ci[5], icov->zz = Ci[8];
  return true;
}

struct NdtConsts {
  double gauss_d1, gauss_d2;
  double step_size, trans_eps;
  int max_iterations;
  int search;  // HGS_KDTREE = 0 / HGS_DIRECT1 = 1 / HGS_DIRECT7 = 2 (the hgs_neighbor_search values)
  int upstream_hd1_sign;
  int pad;
  float kdtree_radius2;  // KDTREE: (float)(resolution^2), the radius of VoxelGridCovariance::radiusSearch
  int line_search;       // hgs_params.ndt_line_search
};

// exp() of the Gaussian term as a fixed sequence of IEEE operations (round-to-even multiply, fma, rint, ldexp), so that
// every implementation — this header on gfx950, the CPU oracle — returns the same bits; libm exp() differs between
// the device library and glibc in the last place of a few results, and ndt_omp's iteration amplifies exactly that.
// ndt_omp evaluates exp in float; here: Cody-Waite reduction + degree-13 Taylor polynomial in double (error < 1 ulp of the
// double result), rounded once to float by the caller.  Only the range the float result can represent is resolved.
#if defined(__HIP_DEVICE_COMPILE__)
// A polynomial coefficient materialised where it is used (two s_mov_b32 into a scalar pair that v_fma_f64 reads directly): left to
// itself the compiler hoists the dozen 64-bit constants out of k_ndt_pass's cell loop into VGPR pairs — 20 registers of a kernel
// that sits at the two-waves-per-SIMD limit — and copies one into the accumulator in front of every v_fmac_f64.
__device__ __forceinline__ double ndt_exp_coeff(double c) {
  int lo = (int)(unsigned)((unsigned long long)__double_as_longlong(c) & 0xffffffffull), hi = (int)(unsigned)((unsigned long long)__double_as_longlong(c) >> 32);
  asm volatile("" : "+s"(lo), "+s"(hi));
  return __hiloint2double(hi, lo);
}
#define HGS_NDT_EXP_COEFF(c) ndt_exp_coeff(c)
// One Horner step p * r + c with the coefficient read from its scalar pair: v_fma_f64 (three-address).  Left to itself the compiler picks
// v_fmac_f64, whose addend is its destination: two v_mov_b32 per step to copy the coefficient there first (20 of ~330 vector instructions per
// visited cell of k_ndt_pass).  Same IEEE operation, same bits.
__device__ __forceinline__ double ndt_horner_step(double p, double r, double c) {
  double out;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(p), "v"(r), "s"(c));
  return out;
}
#define HGS_NDT_HORNER(p, r, c) ndt_horner_step(p, r, c)
#else
#define HGS_NDT_EXP_COEFF(c) (c)
#define HGS_NDT_HORNER(p, r, c) fma(p, r, c)
#endif
HGS_HD double ndt_exp(double x) {
  HGS_FP_STRICT
  if (x != x) return x;
  if (x < -110.0) return 0.0;        // below the smallest float subnormal
  if (x > 90.0) return INFINITY;     // above FLT_MAX
  const double kd = rint(x * 0x1.71547652b82fep+0);
  const double r = fma(-kd, 0x1.a39ef35793c76p-33, fma(-kd, 0x1.62e42fee00000p-1, x));
  // (first step with the leading coefficient as the scalar factor: as a literal it would sit in a vector register pair for the whole kernel)
  double p = HGS_NDT_HORNER(r, HGS_NDT_EXP_COEFF(0x1.6124613a86d09p-33), 0x1.1eed8eff8d898p-29);
  p = HGS_NDT_HORNER(p, r, 0x1.ae64567f544e4p-26);
  p = HGS_NDT_HORNER(p, r, 0x1.27e4fb7789f5cp-22);
  p = HGS_NDT_HORNER(p, r, 0x1.71de3a556c734p-19);
  p = HGS_NDT_HORNER(p, r, 0x1.a01a01a01a01ap-16);
  p = HGS_NDT_HORNER(p, r, 0x1.a01a01a01a01ap-13);
  p = HGS_NDT_HORNER(p, r, 0x1.6c16c16c16c17p-10);
  p = HGS_NDT_HORNER(p, r, 0x1.1111111111111p-7);
  p = HGS_NDT_HORNER(p, r, 0x1.5555555555555p-5);
  p = HGS_NDT_HORNER(p, r, 0x1.5555555555555p-3);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kd);
}

// ---- order-independent ("exact") accumulation of the per-point score / gradient / Hessian contributions ------------
// ndt_omp sums the N per-point doubles serially; a parallel sum re-associates them, and the Newton iteration of a weakly
// constrained scan amplifies that last-bit difference until two implementations part.  Here every per-point double t is
// split on a FIXED binary grid into two 50-bit integer chunks
//     q0 = RN(t / 2^(E+50)),   q1 = RN((t - q0 2^(E+50)) / 2^E)           (what lies below 2^E is rounded away)
// which are summed as integers — associative, so any tiling / atomics / thread order gives the same total — and the total
// V = (sum q0) 2^50 + (sum q1) is rounded ONCE to double: result = RN(V) 2^E.  E depends only on which accumulator it is:
// Hessian -47 (range |t| < 2^52, resolution 7e-15), gradient -51 (2^48, 4e-16), score -59 (2^40, 2e-18): finer than the
// rounding error of the serial double sum it replaces.  A term outside its range (or NaN) poisons the pass: all sums NaN,
// which ends the registration unconverged like a NaN Newton step does upstream.  The oracle implements the same definition
// (oracle/ndt.hpp, sum mode "exact") independently.
constexpr int kNdtChunkBits = 50;
constexpr double kNdtMagic = 6755399441055744.0;  // 1.5 * 2^52: adding it leaves RN(x) in the low mantissa bits
HGS_HD int ndt_sum_exponent(int k /* accumulator 0..42 */) { return k < 36 ? -47 : (k < 42 ? -51 : -59); }

// m0 / m1 = kNdtMagic + q0 / q1 (as doubles: mantissa field = 2^51 + q); false if t is out of range or NaN.
HGS_HD bool ndt_exact_split(double t, int E, double* m0, double* m1) {
  HGS_FP_STRICT
  const double s0 = ldexp(1.0, -(E + kNdtChunkBits)), S0 = ldexp(1.0, E + kNdtChunkBits), s1 = ldexp(1.0, -E);
  const double a = fma(t, s0, kNdtMagic);
  const double q0 = a - kNdtMagic;
  const double r = fma(-q0, S0, t);
  *m0 = a;
  *m1 = fma(r, s1, kNdtMagic);
  return fabs(t) < ldexp(1.0, E + 2 * kNdtChunkBits - 1);
}
HGS_HD long long ndt_chunk_of(double m) {  // the signed integer a magic-added double carries
  unsigned long long b;
  memcpy(&b, &m, 8);
  return (long long)(b & 0xfffffffffffffull) - (1ll << 51);
}
// RN-even of the 128-bit integer V (two's complement in hi:lo), times 2^E.
HGS_HD double ndt_i128_to_double(__int128 v, int E) {
  if (v == 0) return 0.0;
  const bool neg = v < 0;
  unsigned __int128 u = neg ? (unsigned __int128)0 - (unsigned __int128)v : (unsigned __int128)v;
  const unsigned long long hi = (unsigned long long)(u >> 64), lo = (unsigned long long)u;
  const int hb = hi ? 127 - __builtin_clzll(hi) : 63 - __builtin_clzll(lo);
  double d;
  if (hb <= 52) {
    d = (double)lo;
  } else {
    const int sh = hb - 52;
    unsigned long long m = (unsigned long long)(u >> sh);
    const unsigned __int128 rem = u & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
    if (rem > half || (rem == half && (m & 1ull))) m++;
    d = ldexp((double)m, sh);
  }
  d = ldexp(d, E);
  return neg ? -d : d;
}

// Neighbourhood of a transformed point (getNeighborhoodAtPoint1 / 7, or the KDTREE search of ndt_omp: a radius search of
// `resolution` around the point on the kd-tree of the valid cells' centroids).  A centroid lies inside its own cell, so
// every centroid within one resolution of the point belongs to one of the 27 cells around the point's cell: KDTREE is
// the 27-neighbourhood filtered by the float distance to the (float) centroid — no second search structure.
HGS_HD int ndt_num_offsets(int search) { return search == 1 ? 1 : (search == 2 ? 7 : 27); }
HGS_HD void ndt_offset(int search, int o, int* ox, int* oy, int* oz) {
  if (search == 0) {
    *ox = o / 9 - 1, *oy = (o / 3) % 3 - 1, *oz = o % 3 - 1;
  } else {  // centre, +x, -x, +y, -y, +z, -z
    *ox = (o == 1) - (o == 2), *oy = (o == 3) - (o == 4), *oz = (o == 5) - (o == 6);
  }
}
HGS_HD bool ndt_cell_in_reach(const NdtConsts& c, const F3& xt, const double* mean) {
  return c.search != 0 || dist2f(xt, (float)mean[0], (float)mean[1], (float)mean[2]) <= c.kdtree_radius2;
}

// Angular derivative tables (computeAngleDerivatives): 8 j_ang rows and 15 h_ang rows as float triples.
struct NdtAngles {
  float j[8][3];
  float h[15][3];
  float T[12];  // float pose row-major 3x4 used to transform the cloud
};

HGS_HD Pose ndt_pose_from_p(const double* p) {
  HGS_FP_STRICT
  const double cx = cos(p[3]), sx = sin(p[3]), cy = cos(p[4]), sy = sin(p[4]), cz = cos(p[5]), sz = sin(p[5]);
  Pose T;
  T.m[0] = cy * cz, T.m[1] = -cy * sz, T.m[2] = sy, T.m[3] = p[0];
  T.m[4] = cx * sz + sx * sy * cz, T.m[5] = cx * cz - sx * sy * sz, T.m[6] = -sx * cy, T.m[7] = p[1];
  T.m[8] = sx * sz - cx * sy * cz, T.m[9] = cx * sy * sz + sx * cz, T.m[10] = cx * cy, T.m[11] = p[2];
  return T;
}

HGS_HD void ndt_angle_tables(const double* p, int upstream_hd1_sign, NdtAngles& a) {
  HGS_FP_STRICT
  double cx, cy, cz, sx, sy, sz;
  if (fabs(p[3]) < 10e-5) cx = 1.0, sx = 0.0; else cx = cos(p[3]), sx = sin(p[3]);
  if (fabs(p[4]) < 10e-5) cy = 1.0, sy = 0.0; else cy = cos(p[4]), sy = sin(p[4]);
  if (fabs(p[5]) < 10e-5) cz = 1.0, sz = 0.0; else cz = cos(p[5]), sz = sin(p[5]);
  const double j[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy}, {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},
                          {-sy * cz, sy * sz, cy},                                     {sx * cy * cz, -sx * cy * sz, sx * sy},
                          {-cx * cy * cz, cx * cy * sz, -cx * sy},                      {-cy * sz, -cy * cz, 0},
                          {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0},         {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};
  const double hd1z = upstream_hd1_sign ? sy : -sy;
  const double h[15][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy},  {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},
                           {cx * cy * cz, -cx * cy * sz, cx * sy},                       {sx * cy * cz, -sx * cy * sz, sx * sy},
                           {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0},         {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0},
                           {-cy * cz, cy * sz, hd1z},                                    {-sx * sy * cz, sx * sy * sz, sx * cy},
                           {cx * sy * cz, -cx * sy * sz, -cx * cy},                      {sy * sz, sy * cz, 0},
                           {-sx * cy * sz, -sx * cy * cz, 0},                            {cx * cy * sz, cx * cy * cz, 0},
                           {-cy * cz, cy * sz, 0},                                       {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0},
                           {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0}};
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 3; c++) a.j[r][c] = (float)j[r][c];
  for (int r = 0; r < 15; r++)
    for (int c = 0; c < 3; c++) a.h[r][c] = (float)h[r][c];
  const Pose T = ndt_pose_from_p(p);
  pose_to_float(T, a.T);
}

// Per-point derivative temporaries (computePointDerivatives), float like ndt_omp.
struct NdtPointDeriv {
  float xj[8];   // j_ang rows . x
  float xh[15];  // h_ang rows . x
};

HGS_HD void ndt_point_derivatives(const NdtAngles& a, float x, float y, float z, NdtPointDeriv& d) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
#pragma unroll
  for (int r = 0; r < 8; r++) d.xj[r] = a.j[r][0] * x + a.j[r][1] * y + a.j[r][2] * z;
#pragma unroll
  for (int r = 0; r < 15; r++) d.xh[r] = a.h[r][0] * x + a.h[r][1] * y + a.h[r][2] * z;
}

// Accumulator layout of one NDT derivative pass: [0..35] the FULL 6x6 Hessian, row-major — upstream evaluates all 36
// entries in float and (i,j) / (j,i) round differently, so the matrix handed to the SVD solve is slightly asymmetric;
// the Newton direction of a weakly constrained scan is sensitive to exactly that, hence no mirrored upper triangle —
// [36..41] gradient, [42] score.
constexpr int kAccNdt = 43;

// updateDerivatives for one (point, cell): adds to acc[kAccNdt]. Float temporaries, double
// accumulation, exactly the operation order of the oracle (oracle/ndt.hpp) so the two agree to float rounding.
HGS_HD void ndt_cell_terms(const NdtConsts& c, const NdtPointDeriv& pd, float qx, float qy, float qz,  // q = x' - mean (float)
                           const float* ci /*icov xx,xy,xz,yy,yz,zz*/, double* acc) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float d1 = (float)c.gauss_d1, d2 = (float)c.gauss_d2;
  const float C[3][3] = {{ci[0], ci[1], ci[2]}, {ci[1], ci[3], ci[4]}, {ci[2], ci[4], ci[5]}};
  float qC[3];
#pragma unroll
  for (int s = 0; s < 3; s++) qC[s] = qx * C[0][s] + qy * C[1][s] + qz * C[2][s];
  const float qCq = qC[0] * qx + qC[1] * qy + qC[2] * qz;
  float e = (float)ndt_exp((double)(-d2 * qCq * 0.5f));  // fixed-sequence double exp rounded once: same bits everywhere
  const float score_inc = -d1 * e;
  e = d2 * e;
  if (e > 1.f || e < 0.f || e != e) return;
  e *= d1;
  acc[42] += (double)score_inc;
  // point gradient pg (3x6) = [ I | columns 3..5 from xj ], with pg[0][3] = 0.  Upstream multiplies the full matrices
  // in float; a product with the identity part contributes exact zeros (x*1 = x, 0*y = 0, z + 0 = z for finite
  // operands), so the structurally-zero terms are left out here — the remaining operations and their order are
  // upstream's, the results are bit-identical (up to the sign of an exact zero) and a fifth of the arithmetic is gone.
  const float pg13 = pd.xj[0], pg23 = pd.xj[1];
  const float pg04 = pd.xj[2], pg14 = pd.xj[3], pg24 = pd.xj[4];
  const float pg05 = pd.xj[5], pg15 = pd.xj[6], pg25 = pd.xj[7];
  float Cpg[3][6], qCpg[6];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    Cpg[r][0] = C[r][0], Cpg[r][1] = C[r][1], Cpg[r][2] = C[r][2];
    Cpg[r][3] = C[r][1] * pg13 + C[r][2] * pg23;
    Cpg[r][4] = C[r][0] * pg04 + C[r][1] * pg14 + C[r][2] * pg24;
    Cpg[r][5] = C[r][0] * pg05 + C[r][1] * pg15 + C[r][2] * pg25;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) qCpg[k] = qx * Cpg[0][k] + qy * Cpg[1][k] + qz * Cpg[2][k];
#pragma unroll
  for (int k = 0; k < 6; k++) acc[36 + k] += (double)(e * qCpg[k]);
  // second derivatives d2x'/dp_i dp_j (3-vectors), non-zero for i,j in {3,4,5}
  const float ph[6][3] = {{0.f, pd.xh[0], pd.xh[1]},          // (3,3) a
                          {0.f, pd.xh[2], pd.xh[3]},          // (3,4) b
                          {0.f, pd.xh[4], pd.xh[5]},          // (3,5) c
                          {pd.xh[6], pd.xh[7], pd.xh[8]},     // (4,4) d
                          {pd.xh[9], pd.xh[10], pd.xh[11]},   // (4,5) e
                          {pd.xh[12], pd.xh[13], pd.xh[14]}}; // (5,5) f
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float qCh = 0.f;
      if (i >= 3 && j >= 3) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int hidx = (lo == 3) ? (hi - 3) : (lo == 4 ? (hi - 4 + 3) : 5);
        // rows a, b, c have a zero first component: qC[0]*0 drops out exactly
        qCh = hidx < 3 ? qC[1] * ph[hidx][1] + qC[2] * ph[hidx][2] : qC[0] * ph[hidx][0] + qC[1] * ph[hidx][1] + qC[2] * ph[hidx][2];
      }
      // pg[.][j]^T Cpg[.][i]
      const float pgCpg = j < 3 ? Cpg[j][i]
                                : (j == 3 ? pg13 * Cpg[1][i] + pg23 * Cpg[2][i]
                                          : (j == 4 ? pg04 * Cpg[0][i] + pg14 * Cpg[1][i] + pg24 * Cpg[2][i] : pg05 * Cpg[0][i] + pg15 * Cpg[1][i] + pg25 * Cpg[2][i]));
      acc[i * 6 + j] += (double)(e * (-d2 * qCpg[i] * qCpg[j] + qCh + pgCpg));
    }
}

#if defined(__HIPCC__)
// updateDerivatives for one (point, cell) as the derivative kernel runs it: the SAME float operations in the SAME order as
// ndt_cell_terms above (so the same bits), written on pairs of entries so that gfx950 issues v_pk_mul_f32 / v_pk_add_f32 —
// two entries per VALU slot; the reference arithmetic is unfused (separate multiply and add), which is exactly what the
// packed instructions provide.  acc[kAccNdt] are the calling point's double sums over its cells (visited in
// neighbourhood order, like ndt_omp's per-point accumulation).
typedef float ndt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ndt_f2 ndt_s2(float a) { return ndt_f2{a, a}; }

__device__ __forceinline__ void ndt_cell_terms_pk(float d1, float d2, const NdtPointDeriv& pd, float qx, float qy, float qz, const float* ci, double* acc) {
#pragma clang fp contract(off)
  const float c00 = ci[0], c01 = ci[1], c02 = ci[2], c11 = ci[3], c12 = ci[4], c22 = ci[5];
  const ndt_f2 QX = ndt_s2(qx), QY = ndt_s2(qy), QZ = ndt_s2(qz);
  // qC = q^T C^-1 ; entries 0,1 as a pair.  (Cpg[r][k] = C[r][k] for k < 3, so qCpg[0..2] are these same numbers.)
  const ndt_f2 qC01 = QX * ndt_f2{c00, c01} + QY * ndt_f2{c01, c11} + QZ * ndt_f2{c02, c12};
  const float qC2 = qx * c02 + qy * c12 + qz * c22;
  const float qCq = qC01.x * qx + qC01.y * qy + qC2 * qz;
  float e = (float)ndt_exp((double)(-d2 * qCq * 0.5f));
  const float score_inc = -d1 * e;
  e = d2 * e;
  if (e > 1.f || e < 0.f || e != e) return;
  e *= d1;
  acc[42] += (double)score_inc;
  const float pg13 = pd.xj[0], pg23 = pd.xj[1];
  const ndt_f2 pg0_45 = {pd.xj[2], pd.xj[5]}, pg1_45 = {pd.xj[3], pd.xj[6]}, pg2_45 = {pd.xj[4], pd.xj[7]};
  // Cpg = C^-1 * point_gradient, columns 3 | (4,5); rows 0..2
  const ndt_f2 Cpg01_3 = ndt_f2{c01, c11} * ndt_s2(pg13) + ndt_f2{c02, c12} * ndt_s2(pg23);  // (Cpg[0][3], Cpg[1][3])
  const float Cpg2_3 = c12 * pg13 + c22 * pg23;
  const ndt_f2 Cpg0_45 = ndt_s2(c00) * pg0_45 + ndt_s2(c01) * pg1_45 + ndt_s2(c02) * pg2_45;
  const ndt_f2 Cpg1_45 = ndt_s2(c01) * pg0_45 + ndt_s2(c11) * pg1_45 + ndt_s2(c12) * pg2_45;
  const ndt_f2 Cpg2_45 = ndt_s2(c02) * pg0_45 + ndt_s2(c12) * pg1_45 + ndt_s2(c22) * pg2_45;
  // qCpg[k] = q . Cpg[.][k]
  const float qCpg3 = qx * Cpg01_3.x + qy * Cpg01_3.y + qz * Cpg2_3;
  const ndt_f2 qCpg45 = QX * Cpg0_45 + QY * Cpg1_45 + QZ * Cpg2_45;
  const ndt_f2 qCpg01 = qC01, qCpg23 = {qC2, qCpg3};
  const ndt_f2 E2 = ndt_s2(e);
  {
    const ndt_f2 g01 = E2 * qCpg01, g23 = E2 * qCpg23, g45 = E2 * qCpg45;
    acc[36] += (double)g01.x, acc[37] += (double)g01.y, acc[38] += (double)g23.x, acc[39] += (double)g23.y, acc[40] += (double)g45.x, acc[41] += (double)g45.y;
  }
  // q^T C^-1 (second derivatives): a, b, c have a zero first component
  const float qC0 = qC01.x, qC1 = qC01.y;
  const float hA = qC1 * pd.xh[0] + qC2 * pd.xh[1], hB = qC1 * pd.xh[2] + qC2 * pd.xh[3], hC = qC1 * pd.xh[4] + qC2 * pd.xh[5];
  const float hD = qC0 * pd.xh[6] + qC1 * pd.xh[7] + qC2 * pd.xh[8], hE = qC0 * pd.xh[9] + qC1 * pd.xh[10] + qC2 * pd.xh[11],
              hF = qC0 * pd.xh[12] + qC1 * pd.xh[13] + qC2 * pd.xh[14];
  // Cpg by column i (rows 0..2): column i < 3 is C's, 3..5 from above
  const float cg0[6] = {c00, c01, c02, Cpg01_3.x, Cpg0_45.x, Cpg0_45.y};
  const float cg1[6] = {c01, c11, c12, Cpg01_3.y, Cpg1_45.x, Cpg1_45.y};
  const float cg2[6] = {c02, c12, c22, Cpg2_3, Cpg2_45.x, Cpg2_45.y};
  const ndt_f2 ND2 = ndt_s2(-d2);
  const ndt_f2 u01 = ND2 * qCpg01, u23 = ND2 * qCpg23, u45 = ND2 * qCpg45;   // (-d2 * qCpg[i])
  const float u[6] = {u01.x, u01.y, u23.x, u23.y, u45.x, u45.y};
  const float h23[6] = {0.f, 0.f, 0.f, hA, hB, hC};                      // qCh(i, 3)
  const ndt_f2 h45[6] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {hB, hC}, {hD, hE}, {hE, hF}};  // qCh(i, 4), qCh(i, 5)
#pragma unroll
  for (int i = 0; i < 6; i++) {
    // pg[.][j]^T Cpg[.][i] for j = 3 and (4,5)
    const float P3 = pg13 * cg1[i] + pg23 * cg2[i];
    const ndt_f2 P45 = pg0_45 * ndt_s2(cg0[i]) + pg1_45 * ndt_s2(cg1[i]) + pg2_45 * ndt_s2(cg2[i]);
    const ndt_f2 U = ndt_s2(u[i]);
    ndt_f2 t01 = U * qCpg01, t23 = U * qCpg23, t45 = U * qCpg45;
    if (i >= 3) t23 = t23 + ndt_f2{0.f, h23[i]}, t45 = t45 + h45[i];
    t01 = t01 + ndt_f2{cg0[i], cg1[i]};
    t23 = t23 + ndt_f2{cg2[i], P3};
    t45 = t45 + P45;
    t01 = E2 * t01, t23 = E2 * t23, t45 = E2 * t45;
    acc[i * 6 + 0] += (double)t01.x, acc[i * 6 + 1] += (double)t01.y, acc[i * 6 + 2] += (double)t23.x;
    acc[i * 6 + 3] += (double)t23.y, acc[i * 6 + 4] += (double)t45.x, acc[i * 6 + 5] += (double)t45.y;
  }
}
#endif  // __HIPCC__

// Eigen's eulerAngles(0,1,2) on the float rotation of the guess (ndt_omp's initial p)
HGS_HD void ndt_euler_xyz_f(const float* g16_colmajor, float* out) {
  HGS_FP_STRICT
  const float R00 = g16_colmajor[0], R01 = g16_colmajor[4], R02 = g16_colmajor[8];
  const float R10 = g16_colmajor[1], R11 = g16_colmajor[5], R12 = g16_colmajor[9];
  const float R20 = g16_colmajor[2], R21 = g16_colmajor[6], R22 = g16_colmajor[10];
  // float transcendental = double function rounded once: identical on the device libm and on glibc (one float ulp
  // near pi is 2.4e-7 rad — enough to move the Newton trajectory of a weakly constrained scan); no FMA contraction
  float r0 = (float)atan2((double)R12, (double)R22);
  const float c2 = sqrtf(R00 * R00 + R01 * R01);
  float r1;
  if (r0 > 0.f) {
    r0 -= 3.14159265358979323846f;
    r1 = (float)atan2((double)-R02, (double)-c2);
  } else {
    r1 = (float)atan2((double)-R02, (double)c2);
  }
  const float s1 = (float)sin((double)r0), c1 = (float)cos((double)r0);
  const float r2 = (float)atan2((double)(s1 * R20 - c1 * R10), (double)(c1 * R11 - s1 * R21));
  out[0] = -r0, out[1] = -r1, out[2] = -r2;
}

// ---- per-problem Newton state machine (computeTransformation + computeStepLengthMT without the dead MT loop) --
enum NdtPhase { NDT_DERIV = 0, NDT_DONE = 1 };

// More-Thuente line search (Sun & Yuan; the structure of pcl::NormalDistributionsTransform::computeStepLengthMT /
// trialValueSelectionMT / updateIntervalMT), one trial per derivative pass: wants_another_trial() consumes phi and its
// directional derivative at the trial step and, if the search goes on, leaves the next trial step in a_next.  Every trial
// is a full derivative pass, so the accepted trial's Hessian needs no separate computeHessian.  Only used with
// hgs_params.ndt_line_search (ndt_omp itself never enters the loop).
struct MoreThuente {
  double phi_0, d_phi_0, a_l, f_l, g_l, a_u, f_u, g_u, a_next;
  int open_interval, trials;
};
HGS_HD void mt_start(MoreThuente& m, double phi_0, double d_phi_0) {
  HGS_FP_STRICT
  m.phi_0 = phi_0, m.d_phi_0 = d_phi_0;
  m.a_l = m.a_u = 0.0, m.f_l = m.f_u = 0.0;
  m.g_l = m.g_u = d_phi_0 - 1e-4 * d_phi_0;
  m.open_interval = 1, m.trials = 0, m.a_next = 0.0;
}
HGS_HD double mt_cubic(double a_l, double f_l, double g_l, double a_t, double f_t, double g_t) {
  HGS_FP_STRICT
  const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
  return a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
}
HGS_HD double mt_trial_value(const MoreThuente& m, double a_t, double f_t, double g_t) {
  HGS_FP_STRICT
  if (f_t > m.f_l) {  // higher value: the minimum is bracketed between a_l and a_t
    const double a_c = mt_cubic(m.a_l, m.f_l, m.g_l, a_t, f_t, g_t);
    const double a_q = m.a_l - 0.5 * (m.a_l - a_t) * m.g_l / (m.g_l - (m.f_l - f_t) / (m.a_l - a_t));
    return fabs(a_c - m.a_l) < fabs(a_q - m.a_l) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * m.g_l < 0) {  // lower value, derivative of opposite sign
    const double a_c = mt_cubic(m.a_l, m.f_l, m.g_l, a_t, f_t, g_t);
    const double a_s = m.a_l - (m.a_l - a_t) / (m.g_l - g_t) * m.g_l;
    return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
  }
  if (fabs(g_t) <= fabs(m.g_l)) {  // lower value, same sign, smaller derivative
    const double a_c = mt_cubic(m.a_l, m.f_l, m.g_l, a_t, f_t, g_t);
    const double a_s = m.a_l - (m.a_l - a_t) / (m.g_l - g_t) * m.g_l;
    const double a_n = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
    const double lim = a_t + 0.66 * (m.a_u - a_t);
    return a_t > m.a_l ? (lim < a_n ? lim : a_n) : (lim > a_n ? lim : a_n);
  }
  return mt_cubic(m.a_u, m.f_u, m.g_u, a_t, f_t, g_t);  // lower value, same sign, larger derivative
}
HGS_HD bool mt_update_interval(MoreThuente& m, double a_t, double f_t, double g_t) {
  HGS_FP_STRICT
  if (f_t > m.f_l) {
    m.a_u = a_t, m.f_u = f_t, m.g_u = g_t;
    return false;
  }
  if (g_t * (m.a_l - a_t) > 0) {
    m.a_l = a_t, m.f_l = f_t, m.g_l = g_t;
    return false;
  }
  if (g_t * (m.a_l - a_t) < 0) {
    m.a_u = m.a_l, m.f_u = m.f_l, m.g_u = m.g_l;
    m.a_l = a_t, m.f_l = f_t, m.g_l = g_t;
    return false;
  }
  return true;
}
HGS_HD bool mt_wants_another_trial(MoreThuente& m, double a_t, double phi_t, double d_phi_t, double step_max, double step_min) {
  HGS_FP_STRICT
  const double mu = 1e-4, nu = 0.9;
  const double psi_t = phi_t - m.phi_0 - mu * m.d_phi_0 * a_t, d_psi_t = d_phi_t - mu * m.d_phi_0;
  bool interval_converged;
  if (m.trials == 0) {
    interval_converged = (step_max - step_min) < 0;
  } else {  // bookkeeping of the trial just evaluated (the tail of upstream's loop body)
    if (m.open_interval && psi_t <= 0 && d_psi_t >= 0) {
      m.open_interval = 0;
      m.f_l += m.phi_0 - mu * m.d_phi_0 * m.a_l, m.g_l += mu * m.d_phi_0;
      m.f_u += m.phi_0 - mu * m.d_phi_0 * m.a_u, m.g_u += mu * m.d_phi_0;
    }
    interval_converged = m.open_interval ? mt_update_interval(m, a_t, psi_t, d_psi_t) : mt_update_interval(m, a_t, phi_t, d_phi_t);
  }
  if (interval_converged || m.trials >= 10 || (psi_t <= 0 && d_phi_t <= -nu * m.d_phi_0)) return false;
  double a = m.open_interval ? mt_trial_value(m, a_t, psi_t, d_psi_t) : mt_trial_value(m, a_t, phi_t, d_phi_t);
  a = step_max < a ? step_max : a;  // std::min / std::max as upstream applies them (a NaN passes through)
  a = a < step_min ? step_min : a;
  // (not in PCL) the clamped trial is the point just evaluated, or the interpolation broke down (a_t == a_l gives 0/0): the
  // search cannot make progress — PCL would re-evaluate the same point until the trial limit
  if (a == a_t || a != a) return false;
  m.a_next = a;
  m.trials++;
  return true;
}

struct NdtState {
  double p[6];      // parameters the NEXT derivative pass is evaluated at
  double p_acc[6];  // upstream's `p` (accumulated; differs from the evaluation point only by rounding)
  double dp[6];     // unit direction of the pending step
  double a_t;       // pending step length
  double score;
  Pose final_T;
  int phase;
  int iterations;
  int passes;
  int converged;
  int first;  // 1 until the initial derivative pass has been consumed
  int searching;  // ndt_line_search: the pending pass is a trial of the line search described by `mt`
  MoreThuente mt;
};

HGS_HD void ndt_state_init(NdtState& s, const float* guess_colmajor) {
  float e[3];
  ndt_euler_xyz_f(guess_colmajor, e);
  s.p[0] = guess_colmajor[12], s.p[1] = guess_colmajor[13], s.p[2] = guess_colmajor[14];
  s.p[3] = e[0], s.p[4] = e[1], s.p[5] = e[2];
  for (int i = 0; i < 6; i++) s.p_acc[i] = s.p[i], s.dp[i] = 0;
  s.a_t = 0, s.score = 0;
  s.final_T = pose_from_colmajor_f(guess_colmajor);
  s.phase = NDT_DERIV;
  s.iterations = 0, s.passes = 0, s.converged = 0, s.first = 1, s.searching = 0;
  mt_start(s.mt, 0.0, 0.0);
}

// True if the pass that has just been reduced needs no Newton direction: it completes the final iteration, or it is a trial of
// the line search that will be followed by another trial (judged here on a copy of the search state, exactly as
// ndt_after_derivatives will judge it).
HGS_HD bool ndt_pass_is_last(const NdtState& s, const NdtConsts& c, const double* acc) {
  HGS_FP_STRICT  // the same arithmetic as ndt_after_derivatives: both must reach the same verdict
  double a_t = s.a_t;
  if (s.searching) {
    MoreThuente m = s.mt;
    double d_phi_t = 0;
    for (int i = 0; i < 6; i++) d_phi_t -= acc[36 + i] * s.dp[i];
    if (mt_wants_another_trial(m, s.a_t, -acc[42], d_phi_t, c.step_size, c.trans_eps / 2)) return true;
  }
  return !s.first && ((s.iterations > c.max_iterations) || (s.iterations && fabs(a_t) < c.trans_eps));
}

// Consumes the reduced {H, g, score} of a derivative pass evaluated at s.p and prepares the next evaluation point.
// dp_newton = SVD-solve(H, -g) of this pass (unused if ndt_pass_is_last): computed by the caller because the device runs it
// on three lanes (solve_svd6_wave) while this control path is one thread's.
HGS_HD void ndt_after_derivatives(NdtState& s, const double* acc, const NdtConsts& c, const double* dp_newton) {
  HGS_FP_STRICT
  double g[6];
  for (int i = 0; i < 6; i++) g[i] = acc[36 + i];
  s.score = acc[42];
  s.passes++;
  if (s.searching) {
    // this pass evaluated the trial step s.a_t along s.dp from p_acc
    double d_phi_t = 0;
    for (int i = 0; i < 6; i++) d_phi_t -= g[i] * s.dp[i];
    const double step_min = c.trans_eps / 2;
    if (mt_wants_another_trial(s.mt, s.a_t, -s.score, d_phi_t, c.step_size, step_min)) {
      s.a_t = s.mt.a_next;
      for (int i = 0; i < 6; i++) s.p[i] = s.p_acc[i] + s.dp[i] * s.a_t;
      s.phase = NDT_DERIV;
      return;
    }
    s.searching = 0;
  }
  if (!s.first) {
    // finish the iteration whose step produced this pass: p += dp * a_t ; convergence test ; iter++
    for (int i = 0; i < 6; i++) s.p_acc[i] += s.dp[i] * s.a_t;
    s.final_T = ndt_pose_from_p(s.p);
    const bool conv = (s.iterations > c.max_iterations) || (s.iterations && fabs(s.a_t) < c.trans_eps);
    s.iterations++;
    if (conv) {
      s.converged = 1;
      s.phase = NDT_DONE;
      return;
    }
  }
  s.first = 0;
  for (;;) {
    // Newton direction: dp = SVD-solve(H, -g)
    double dp[6];
    for (int i = 0; i < 6; i++) dp[i] = dp_newton[i];
    double nrm = 0;
    for (int i = 0; i < 6; i++) nrm += dp[i] * dp[i];
    nrm = sqrt(nrm);
    if (nrm == 0 || nrm != nrm) {
      s.converged = (nrm == nrm) ? 1 : 0;
      s.phase = NDT_DONE;
      return;
    }
    for (int i = 0; i < 6; i++) dp[i] /= nrm;
    double d_phi_0 = 0;
    for (int i = 0; i < 6; i++) d_phi_0 -= g[i] * dp[i];
    if (d_phi_0 >= 0) {
      if (d_phi_0 == 0) {
        // step length 0: no new derivative pass; the iteration completes immediately with a_t = 0
        const bool conv = (s.iterations > c.max_iterations) || (s.iterations && 0.0 < c.trans_eps);
        s.iterations++;
        if (conv) {
          s.converged = 1;
          s.phase = NDT_DONE;
          return;
        }
        continue;  // same H, g -> same direction: upstream would spin until max_iterations; so do we
      }
      for (int i = 0; i < 6; i++) dp[i] = -dp[i];
    }
    double a_t = nrm < c.step_size ? nrm : c.step_size;
    const double step_min = c.trans_eps / 2;
    a_t = a_t > step_min ? a_t : step_min;
    for (int i = 0; i < 6; i++) {
      s.dp[i] = dp[i];
      s.p[i] = s.p_acc[i] + dp[i] * a_t;
    }
    s.a_t = a_t;
    s.phase = NDT_DERIV;
    if (c.line_search) {
      mt_start(s.mt, -s.score, d_phi_0 < 0 ? d_phi_0 : -d_phi_0);  // d_phi_0 of the (possibly reversed) direction
      s.searching = 1;
    }
    return;
  }
}

// Serial form (host execution in tests/emul).
HGS_HD void ndt_after_derivatives(NdtState& s, const double* acc, const NdtConsts& c) {
  double ng[6], dp[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; i++) ng[i] = -acc[36 + i];
  if (!ndt_pass_is_last(s, c, acc)) solve_svd6(acc, ng, dp);
  ndt_after_derivatives(s, acc, c, dp);
}

}  // namespace hgs
