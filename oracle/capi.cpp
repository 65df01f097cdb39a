// ORACLE — TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library; the product path (hdl_graph_slam_amd/) never does.  Parity unpinned by the reference.
//
// C entry points (ctypes) over the CPU restatement of the registration engines behind
// hdl_graph_slam::select_registration_method (src/hdl_graph_slam/registrations.cpp:22-124).
#include <cstdio>
#include <memory>
#include <vector>
#include "gicp.hpp"
#include "ndt.hpp"
#include "vgicp.hpp"
#include "prefilter.hpp"
#include "mapcloud.hpp"

using namespace hgso;

struct hgso_handle {
  hgs_params prm;
  std::shared_ptr<OCloud> source, target;
  std::unique_ptr<FastGICP> gicp;
  std::unique_ptr<FastVGICP> vgicp;
  std::unique_ptr<NdtOmp> ndt;
  float final_T[16];
};

extern "C" {

int hgso_set_num_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}

hgso_handle* hgso_create(const hgs_params* p) {
  auto* h = new hgso_handle();
  h->prm = *p;
  if (p->method == HGS_FAST_GICP) h->gicp.reset(new FastGICP(*p));
  else if (p->method == HGS_FAST_VGICP) h->vgicp.reset(new FastVGICP(*p));
  else h->ndt.reset(new NdtOmp(*p));
  for (int i = 0; i < 16; i++) h->final_T[i] = (i % 5 == 0) ? 1.f : 0.f;
  return h;
}
void hgso_destroy(hgso_handle* h) { delete h; }

int hgso_set_target(hgso_handle* h, const void* pts, size_t n, size_t stride) {
  h->target = std::make_shared<OCloud>();
  h->target->assign(pts, n, stride);
  if (h->gicp) h->gicp->target = h->target;
  if (h->vgicp) h->vgicp->set_target(h->target);
  if (h->ndt) h->ndt->set_target(h->target);
  return 0;
}
int hgso_set_source(hgso_handle* h, const void* pts, size_t n, size_t stride) {
  h->source = std::make_shared<OCloud>();
  h->source->assign(pts, n, stride);
  if (h->gicp) h->gicp->source = h->source;
  if (h->vgicp) h->vgicp->source = h->source;
  if (h->ndt) h->ndt->source = h->source;
  return 0;
}
int hgso_align(hgso_handle* h, const float guess[16], hgs_result* out) {
  if (!h->source || !h->target) return 1;
  if (h->gicp) h->gicp->align(guess, out);
  else if (h->vgicp) h->vgicp->align(guess, out);
  else h->ndt->align(guess, out);
  for (int i = 0; i < 16; i++) h->final_T[i] = out->final_transformation[i];
  return 0;
}
int hgso_fitness(hgso_handle* h, const float T[16], double max_range, double* score, uint32_t* n_inliers) {
  if (!h->source || !h->target) return 1;
  *score = fitness_score(*h->target, *h->source, T, max_range, n_inliers);
  return 0;
}
// exact 1-NN on the target for arbitrary queries; idx refers to the caller's original target order
int hgso_nn_target(hgso_handle* h, const float* q, size_t nq, size_t stride, int32_t* idx, float* d2) {
  if (!h->target) return 1;
  const char* base = (const char*)q;
#pragma omp parallel for schedule(guided, 8)
  for (long i = 0; i < (long)nq; i++) {
    const float* f = (const float*)(base + i * stride);
    const Neighbor nb = h->target->tree.nn({f[0], f[1], f[2]});
    idx[i] = nb.idx >= 0 ? h->target->orig[nb.idx] : -1;
    d2[i] = nb.d2;
  }
  return 0;
}

// ---- iteration traces ---------------------------------------------------------------------------------
int hgso_trace_len(hgso_handle* h) {
  if (h->gicp) return (int)h->gicp->trace.size();
  if (h->vgicp) return (int)h->vgicp->trace.size();
  return (int)h->ndt->trace.size();
}
// GICP/VGICP: out[14] = T(12 row-major 3x4), error, lambda ; NDT: out[8] = p(6), score, step
int hgso_trace_get(hgso_handle* h, int i, double* out) {
  if (h->gicp || h->vgicp) {
    const GicpTraceEntry& e = h->gicp ? h->gicp->trace[i] : h->vgicp->trace[i];
    for (int k = 0; k < 12; k++) out[k] = e.T[k];
    out[12] = e.error, out[13] = e.lambda;
  } else {
    const NdtTraceEntry& e = h->ndt->trace[i];
    for (int k = 0; k < 6; k++) out[k] = e.p[k];
    out[6] = e.score, out[7] = e.step;
  }
  return 0;
}

// ---- stage-level access for kernel parity tests --------------------------------------------------------
// brute-force and kd-tree kNN on an arbitrary float3 set (stride in bytes)
int hgso_knn(const void* pts, size_t n, size_t stride, const void* queries, size_t nq, size_t qstride, int k, int brute, int32_t* out_idx, float* out_d2) {
  OCloud c;
  c.assign(pts, n, stride);
  const char* qb = (const char*)queries;
#pragma omp parallel for schedule(guided, 8)
  for (long i = 0; i < (long)nq; i++) {
    const float* f = (const float*)(qb + i * qstride);
    const P3f q{f[0], f[1], f[2]};
    std::vector<Neighbor> nb(k, Neighbor{FLT_MAX, -1});
    int found = 0;
    if (brute) {
      std::vector<Neighbor> all(c.pts.size());
      for (size_t j = 0; j < c.pts.size(); j++) all[j] = {dist2f(q, c.pts[j]), (int)j};
      found = std::min<int>(k, (int)all.size());
      std::partial_sort(all.begin(), all.begin() + found, all.end());
      for (int j = 0; j < found; j++) nb[j] = all[j];
    } else {
      found = c.tree.knn(q, k, nb.data());
    }
    for (int j = 0; j < k; j++) {
      out_idx[i * k + j] = j < found ? c.orig[nb[j].idx] : -1;
      out_d2[i * k + j] = j < found ? nb[j].d2 : FLT_MAX;
    }
  }
  return 0;
}

// GICP covariances of a cloud: out_cov[n][6] = xx,xy,xz,yy,yz,zz (double), rows for non-finite points are 0
int hgso_covariances(const void* pts, size_t n, size_t stride, int k, int method, double* out_cov6) {
  OCloud c;
  c.assign(pts, n, stride);
  calculate_covariances(c, k, method);
  for (size_t i = 0; i < n * 6; i++) out_cov6[i] = 0;
  for (size_t i = 0; i < c.pts.size(); i++) {
    double* o = out_cov6 + (size_t)c.orig[i] * 6;
    const M3& C = c.covs[i];
    o[0] = C.m[0][0], o[1] = C.m[0][1], o[2] = C.m[0][2], o[3] = C.m[1][1], o[4] = C.m[1][2], o[5] = C.m[2][2];
  }
  return 0;
}

static Iso iso_from_rowmajor12(const double* T) {
  Iso x;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) x.R.m[r][c] = T[r * 4 + c];
  x.t = {T[3], T[7], T[11]};
  return x;
}

// One GICP/VGICP linearisation at pose T (double row-major 3x4): H[36] row-major, b[6], err; corr[n_source]
// (original target index or -1; VGICP: number of voxel correspondences of that point).
int hgso_gicp_linearize(hgso_handle* h, const double* T12, double* H, double* b, double* err, int32_t* corr) {
  if (!h->source || !h->target || (!h->gicp && !h->vgicp)) return 1;
  const Iso T = iso_from_rowmajor12(T12);
  M6 Hm;
  V6 bv;
  if (h->gicp) {
    h->gicp->ensure_covs();
    *err = h->gicp->linearize(T, &Hm, &bv);
    if (corr) {
      for (size_t i = 0; i < h->source->n_input; i++) corr[i] = -1;
      for (size_t i = 0; i < h->source->pts.size(); i++) {
        const int j = h->gicp->correspondences[i];
        corr[h->source->orig[i]] = j >= 0 ? h->target->orig[j] : -1;
      }
    }
  } else {
    h->vgicp->ensure_covs();
    *err = h->vgicp->linearize(T, &Hm, &bv);
    if (corr) {
      for (size_t i = 0; i < h->source->n_input; i++) corr[i] = 0;
      for (auto& vc : h->vgicp->voxel_correspondences) corr[h->source->orig[vc.first]]++;
    }
  }
  for (int r = 0; r < 6; r++) {
    for (int c = 0; c < 6; c++) H[r * 6 + c] = Hm.m[r][c];
    b[r] = bv.v[r];
  }
  return 0;
}
// error at pose Ti with the correspondences / Mahalanobis matrices of the last linearize
int hgso_gicp_error(hgso_handle* h, const double* T12, double* err) {
  if (!h->gicp && !h->vgicp) return 1;
  const Iso T = iso_from_rowmajor12(T12);
  *err = h->gicp ? h->gicp->compute_error(T) : h->vgicp->compute_error(T);
  return 0;
}

// NDT: voxel table export. Returns number of valid cells; fills up to cap cells:
// cell_ijk[3*i], mean[3*i], icov6[6*i] (xx,xy,xz,yy,yz,zz), npts[i]; sorted by linear key
// NDT: how the per-point contributions of a derivative pass are added up: 0 = serially in double (ndt_omp, the default),
// 1 = order-independent exact accumulation (ndt.hpp ExactSum; what a parallel backend can reproduce bit for bit),
// 2 = one accumulator set per OpenMP thread (no N-long arrays, no serial sum): the "optimised-CPU" baseline variant of SURVEY 8d — timing only
int hgso_ndt_set_sum_mode(hgso_handle* h, int mode) {
  if (!h->ndt || mode < 0 || mode > 2) return 1;
  h->ndt->sum_mode = mode;
  return 0;
}
int hgso_ndt_cells(hgso_handle* h, int cap, int32_t* ijk, double* mean, double* icov6, int32_t* npts) {
  if (!h->ndt) return -1;
  const NdtVoxelGrid& g = h->ndt->grid;
  int i = 0;
  for (long key : g.valid_keys) {
    if (i >= cap) break;
    const NdtCell& c = g.leaves.at(key);
    ijk[3 * i] = c.ijk[0], ijk[3 * i + 1] = c.ijk[1], ijk[3 * i + 2] = c.ijk[2];
    mean[3 * i] = c.mean.x, mean[3 * i + 1] = c.mean.y, mean[3 * i + 2] = c.mean.z;
    double* o = icov6 + 6 * i;
    o[0] = c.icov.m[0][0], o[1] = c.icov.m[0][1], o[2] = c.icov.m[0][2], o[3] = c.icov.m[1][1], o[4] = c.icov.m[1][2], o[5] = c.icov.m[2][2];
    npts[i] = c.n;
    i++;
  }
  return (int)g.valid_keys.size();
}
// NDT: score / gradient / Hessian at p (tx,ty,tz,rx,ry,rz)
int hgso_ndt_derivatives(hgso_handle* h, const double* p, double* score, double* g, double* H) {
  if (!h->ndt || !h->source || !h->target) return 1;
  h->ndt->init_gauss();
  V6 gv;
  M6 Hm;
  *score = h->ndt->derivatives(p, gv, Hm);
  for (int r = 0; r < 6; r++) {
    g[r] = gv.v[r];
    for (int c = 0; c < 6; c++) H[r * 6 + c] = Hm.m[r][c];
  }
  return 0;
}
// Prefilter (apps/prefiltering_nodelet.cpp): points are records with x,y,z at floats 0..2 and intensity at float 4
// (pcl::PointXYZI) when the stride allows, else intensity 0.  out receives {x,y,z,intensity} float4 records;
// returns the number of output points or -1 (voxel index overflow).
long hgso_prefilter(const void* pts, size_t n, size_t stride, const PrefilterParams* prm, float* out4, size_t cap) {
  std::vector<PfPoint> in(n), out;
  for (size_t i = 0; i < n; i++) {
    const float* f = (const float*)((const char*)pts + i * stride);
    in[i] = {f[0], f[1], f[2], stride >= 20 ? f[4] : 0.f};
  }
  if (!prefilter(in, *prm, out)) return -1;
  for (size_t i = 0; i < out.size() && i < cap; i++) out4[4 * i] = out[i].x, out4[4 * i + 1] = out[i].y, out4[4 * i + 2] = out[i].z, out4[4 * i + 3] = out[i].intensity;
  return (long)out.size();
}

// The same with the deskewing step of cloud_callback (:112) in front: imu_angular_velocity as in the sensor_msgs/Imu message.
long hgso_prefilter_deskewed(const void* pts, size_t n, size_t stride, const PrefilterParams* prm, const double* imu_angular_velocity, double scan_period,
                             float* out4, size_t cap) {
  std::vector<PfPoint> in(n), out;
  for (size_t i = 0; i < n; i++) {
    const float* f = (const float*)((const char*)pts + i * stride);
    in[i] = {f[0], f[1], f[2], stride >= 20 ? f[4] : 0.f};
  }
  pf_deskew(in, imu_angular_velocity, scan_period);
  if (!prefilter(in, *prm, out)) return -1;
  for (size_t i = 0; i < out.size() && i < cap; i++) out4[4 * i] = out[i].x, out4[4 * i + 1] = out[i].y, out4[4 * i + 2] = out[i].z, out4[4 * i + 3] = out[i].intensity;
  return (long)out.size();
}

// MapCloudGenerator::generate: n_kf clouds given as one concatenated record array + per-keyframe sizes; poses: 16
// column-major floats each.  Returns the number of output points ({x,y,z,intensity} float4 records) or -1.
long hgso_map_cloud(const void* pts, const size_t* sizes, size_t n_kf, size_t stride, const float* poses16, double resolution, float* out4, size_t cap) {
  std::vector<std::vector<PfPoint>> kfs(n_kf);
  std::vector<std::array<float, 16>> poses(n_kf);
  const char* base = (const char*)pts;
  size_t off = 0;
  for (size_t k = 0; k < n_kf; k++) {
    for (size_t i = 0; i < sizes[k]; i++) {
      const float* f = (const float*)(base + (off + i) * stride);
      kfs[k].push_back({f[0], f[1], f[2], stride >= 20 ? f[4] : 0.f});
    }
    off += sizes[k];
    for (int j = 0; j < 16; j++) poses[k][j] = poses16[16 * k + j];
  }
  std::vector<PfPoint> out;
  if (!map_cloud_generate(kfs, poses, resolution, out)) return -1;
  for (size_t i = 0; i < out.size() && i < cap; i++) out4[4 * i] = out[i].x, out4[4 * i + 1] = out[i].y, out4[4 * i + 2] = out[i].z, out4[4 * i + 3] = out[i].intensity;
  return (long)out.size();
}

// helpers exposed for unit tests of the math primitives
int hgso_se3_exp(const double* d6, double* T12) {
  V6 d;
  for (int i = 0; i < 6; i++) d.v[i] = d6[i];
  const Iso x = se3_exp(d);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T12[r * 4 + c] = x.R.m[r][c];
  T12[3] = x.t.x, T12[7] = x.t.y, T12[11] = x.t.z;
  return 0;
}
int hgso_solve6(const double* A36, const double* b6, int use_svd, double* x6) {
  M6 A;
  V6 b;
  for (int r = 0; r < 6; r++) {
    b.v[r] = b6[r];
    for (int c = 0; c < 6; c++) A.m[r][c] = A36[r * 6 + c];
  }
  const V6 x = use_svd ? solve_svd6(A, b) : solve_ldlt6(A, b);
  for (int r = 0; r < 6; r++) x6[r] = x.v[r];
  return 0;
}
int hgso_eig_sym3(const double* A9, double* eval3, double* evec9) {
  M3 A, V;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) A.m[r][c] = A9[r * 3 + c];
  eig_sym3(A, eval3, V);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) evec9[r * 3 + c] = V.m[r][c];
  return 0;
}
int hgso_euler_xyz(const float* R9_rowmajor, float* out3) {
  float R[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[r][c] = R9_rowmajor[r * 3 + c];
  euler_angles_xyz_f(R, out3);
  return 0;
}
int hgso_ndt_pose(const double* p6, double* T12) {
  const Iso x = ndt_pose_from_p(p6);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T12[r * 4 + c] = x.R.m[r][c];
  T12[3] = x.t.x, T12[7] = x.t.y, T12[11] = x.t.z;
  return 0;
}

}  // extern "C"
