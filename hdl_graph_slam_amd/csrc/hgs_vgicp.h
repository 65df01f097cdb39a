// hgs_vgicp.h — per-voxel / per-point arithmetic of the voxelised GICP engine (fast_gicp::FastVGICP +
// GaussianVoxelMap, ADDITIVE mode; reference call site src/hdl_graph_slam/registrations.cpp:48-56).
// The Levenberg-Marquardt control (LsqRegistration) is the GICP state machine of hgs_gicp.h unchanged; what differs
// is the correspondence model: no nearest-neighbour search, every source point is matched against the Gaussian
// voxel(s) of the target its transformed position falls into (DIRECT1 / 7 / 27), weighted by sqrt(#points).
// HGS_HD like hgs_gicp.h: the HIP kernels and the test-only host harness (tests/emul) call the same functions.
#pragma once
#include "hgs_gicp.h"
#include "hgs_ndt.h"

namespace hgs {

// Voxel coordinate of fast_gicp's GaussianVoxelMap: floor(p / resolution - 0.5), evaluated in double (with a true
// division, like upstream) so that host and device bin identically.
HGS_HD int vgicp_coord(double v, double resolution) {
  HGS_FP_STRICT
  return (int)floor(v / resolution - 0.5);
}

// The voxel table reuses the open-addressing hash + 64-byte record of the NDT target (hgs_ndt.h):
//   v0 = (cov.xx, cov.xy, cov.xz, cov.yy)  v1 = (cov.yz, cov.zz, n, key)  mean[3] (double)
// holding the MEAN of the per-point GICP covariances of the voxel's points (not an inverse).
struct VgicpConsts {
  double resolution;
  int search;  // hgs_neighbor_search: HGS_DIRECT1 / HGS_DIRECT7 / HGS_DIRECT27
  int pad;
};

HGS_HD void vgicp_finalize_voxel(int n, const double* sum_p, const Sym3& sum_c, int key, NdtCellRec* rec) {
  const double inv = 1.0 / (double)n;
  rec->v0.x = (float)(sum_c.xx * inv), rec->v0.y = (float)(sum_c.xy * inv), rec->v0.z = (float)(sum_c.xz * inv), rec->v0.w = (float)(sum_c.yy * inv);
  rec->v1.x = (float)(sum_c.yz * inv), rec->v1.y = (float)(sum_c.zz * inv), rec->v1.z = (float)n, rec->v1.w = int_as_float_hd(key);
  rec->mean[0] = sum_p[0] * inv, rec->mean[1] = sum_p[1] * inv, rec->mean[2] = sum_p[2] * inv, rec->pad = 0.0;
}

HGS_HD int vgicp_num_offsets(int search) { return search == 3 /*HGS_DIRECT27*/ ? 27 : (search == 2 /*HGS_DIRECT7*/ ? 7 : 1); }
HGS_HD void vgicp_offset(int search, int o, int* ox, int* oy, int* oz) {
  if (search == 3) {  // i, j, k in -1..1, k fastest (the order fast_gicp enumerates them in)
    *ox = o / 9 - 1, *oy = (o / 3) % 3 - 1, *oz = o % 3 - 1;
  } else {            // centre, +x, -x, +y, -y, +z, -z
    *ox = (o == 1) - (o == 2), *oy = (o == 3) - (o == 4), *oz = (o == 5) - (o == 6);
  }
}

// All voxel correspondences of one source point a (covariance ca).  The voxels and the Mahalanobis matrices are
// those of the LINEARISATION pose T0 (update_correspondences); the residual is evaluated at Te (== T0 in linearize,
// the LM trial pose in compute_error).  Adds the weighted J^T M J, J^T M e terms to acc[0..26] when WITH_JACOBIAN
// and returns the weighted error sum; *n_hits receives the number of voxels hit.
template <bool WITH_JACOBIAN>
HGS_HD double vgicp_point_terms(const NdtGrid& g, const VgicpConsts& c, const Pose& T0, const Pose& Te, const Sym3& ca, float ax, float ay, float az,
                                double* acc, int* n_hits) {
  const double x = T0.m[0] * ax + T0.m[1] * ay + T0.m[2] * az + T0.m[3];
  const double y = T0.m[4] * ax + T0.m[5] * ay + T0.m[6] * az + T0.m[7];
  const double z = T0.m[8] * ax + T0.m[9] * ay + T0.m[10] * az + T0.m[11];
  const int cx = vgicp_coord(x, c.resolution), cy = vgicp_coord(y, c.resolution), cz = vgicp_coord(z, c.resolution);
  const double R[9] = {T0.m[0], T0.m[1], T0.m[2], T0.m[4], T0.m[5], T0.m[6], T0.m[8], T0.m[9], T0.m[10]};
  const int nn = vgicp_num_offsets(c.search);
  double err = 0.0;
  int hits = 0;
  for (int o = 0; o < nn; o++) {
    int ox, oy, oz;
    vgicp_offset(c.search, o, &ox, &oy, &oz);
    const int ci = ndt_lookup(g, cx + ox, cy + oy, cz + oz);
    if (ci < 0) continue;
    hits++;
    const NdtCellRec rec = g.cells[ci];
    Sym3 M = gicp_mahalanobis(R, ca, sym3_from_floats(rec.v0.x, rec.v0.y, rec.v0.z, rec.v0.w, rec.v1.x, rec.v1.y));
    const double w = sqrt((double)rec.v1.z);
    M.xx *= w, M.xy *= w, M.xz *= w, M.yy *= w, M.yz *= w, M.zz *= w;
    err += gicp_point_terms<WITH_JACOBIAN>(Te, M, ax, ay, az, rec.mean[0], rec.mean[1], rec.mean[2], acc);
  }
  if (n_hits) *n_hits = hits;
  return err;
}

}  // namespace hgs
