// Measurement driver (bench.py --config 3, `adapter_path`): what a sweep costs through adapters/registration_hip.hpp — the object the patched
// factory returns — driven through the pcl::Registration BASE pointer exactly as ScanMatchingOdometryNodelet::matching does
// (apps/scan_matching_odometry_nodelet.cpp:165-262: setInputSource(filtered); a fresh `aligned` cloud; align(*aligned, guess); hasConverged;
// getFinalTransformation; keyframe switch -> setInputTarget), next to the same stream through the bare C-ABI (hgs_set_source + hgs_align).
// pcl::Registration here is the stand-in of tests/mock_pcl, which reproduces what PCL's align() does on the host (initCompute, output = *input_,
// data[3] = 1); its kd-tree is a real exact kd-tree, so the "eager tree" mode pays a real index build per new target, as KdTreeFLANN would.
//   adapter_bench <method 0|1|2> <resolution> <warmup> <keyframe_delta_trans> <scan0.bin> <scan1.bin> ...     (raw PointXYZI records; scan0 = first keyframe)
// Prints one JSON object.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include <pcl/point_types.h>
#include "../../adapters/registration_hip.hpp"

using PointT = pcl::PointXYZI;
using Cloud = pcl::PointCloud<PointT>;
using Clock = std::chrono::steady_clock;

static Cloud::Ptr load(const char* path) {
  auto c = std::make_shared<Cloud>();
  std::ifstream f(path, std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  c->points.resize(raw.size() / sizeof(PointT));
  std::memcpy(c->points.data(), raw.data(), c->points.size() * sizeof(PointT));
  return c;
}
static double pct(std::vector<double> v, double p) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  const double k = p * (v.size() - 1);
  const size_t lo = (size_t)k;
  const size_t hi = std::min(v.size() - 1, lo + 1);
  return v[lo] + (v[hi] - v[lo]) * (k - lo);
}
static float trans_norm(const Eigen::Matrix4f& T) { return std::sqrt(T(0, 3) * T(0, 3) + T(1, 3) * T(1, 3) + T(2, 3) * T(2, 3)); }

// An engine that does nothing: align() through it costs exactly what pcl::Registration::align itself costs on the host (every engine pays it,
// the reference's CPU engines included)
class NullEngine : public pcl::Registration<PointT, PointT, float> {
protected:
  void computeTransformation(PointCloudSource&, const Matrix4& guess) override {
    final_transformation_ = guess;
    converged_ = true;
  }
};

struct Trace {
  std::vector<double> ms;
  std::vector<Eigen::Matrix4f> poses;
  int keyframes = 1, converged = 0;
  long cpu_tree_builds = 0;
};

// the caller: frame-to-keyframe with the translation part of the keyframe rule (launch/hdl_graph_slam_kitti.launch:41-43: 5 m)
template <typename SetTarget, typename Step>
static Trace drive(const std::vector<Cloud::Ptr>& scans, int warmup, float delta_trans, SetTarget&& set_target, Step&& step) {
  Trace tr;
  const long builds0 = pcl::search::KdTree<PointT>::builds_counter().load();
  set_target(scans[0]);
  Eigen::Matrix4f prev = Eigen::Matrix4f::Identity();
  for (size_t i = 1; i < scans.size(); i++) {
    const auto t0 = Clock::now();
    Eigen::Matrix4f T;
    const bool ok = step(scans[i], prev, T);
    if (ok) prev = T;
    bool switched = false;
    if (ok && trans_norm(T) > delta_trans) {  // :241-252 (the target rebuild is part of the sweep that triggers it)
      set_target(scans[i]);
      prev = Eigen::Matrix4f::Identity();
      switched = true;
    }
    const double ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
    if ((int)i > warmup) {
      tr.ms.push_back(ms);
      tr.poses.push_back(T);
      tr.converged += ok ? 1 : 0;
      tr.keyframes += switched ? 1 : 0;
    }
  }
  tr.cpu_tree_builds = pcl::search::KdTree<PointT>::builds_counter().load() - builds0;
  return tr;
}
static void print_trace(const char* name, const Trace& t, bool last = false) {
  double mean = 0;
  for (double v : t.ms) mean += v;
  mean /= std::max<size_t>(1, t.ms.size());
  std::printf("\"%s\": {\"p50_ms\": %.4f, \"p10_ms\": %.4f, \"p90_ms\": %.4f, \"p99_ms\": %.4f, \"mean_ms\": %.4f, \"max_ms\": %.4f, \"sweeps\": %zu, \"converged\": %d, \"keyframe_switches\": %d, "
              "\"cpu_kdtree_builds\": %ld}%s",
              name, pct(t.ms, 0.5), pct(t.ms, 0.1), pct(t.ms, 0.9), pct(t.ms, 0.99), mean, pct(t.ms, 1.0), t.ms.size(), t.converged, t.keyframes - 1, t.cpu_tree_builds, last ? "" : ", ");
}

int main(int argc, char** argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: %s method resolution warmup keyframe_delta_trans scan0.bin scan1.bin scan2.bin ...\n", argv[0]);
    return 2;
  }
  const int method = std::atoi(argv[1]), warmup = std::atoi(argv[3]);
  const double resolution = std::atof(argv[2]);
  const float delta_trans = (float)std::atof(argv[4]);
  std::vector<Cloud::Ptr> scans;
  for (int i = 5; i < argc; i++) scans.push_back(load(argv[i]));
  size_t mean_pts = 0;
  for (auto& s : scans) mean_pts += s->size();
  mean_pts /= scans.size();

  auto configure = [&](hgs_hip::RegistrationHIP<PointT, PointT>& reg) {  // what the patched factory does with the reg_* rosparams
    reg.setTransformationEpsilon(0.01);
    reg.setMaximumIterations(64);
    if (method == HGS_FAST_GICP) {
      reg.setMaxCorrespondenceDistance(2.5);
      reg.setCorrespondenceRandomness(20);
    } else {
      reg.setResolution(resolution);
      if (method == HGS_NDT_OMP) reg.setNeighborhoodSearchMethod(HGS_DIRECT7);
    }
  };

  // ---- A: the bare C-ABI (with the two calls of a sweep timed separately)
  Trace t_abi;
  std::vector<double> abi_set_source_ms, abi_align_ms, abi_iterations;
  {
    hgs_params p;
    hgs_params_default(method, &p);
    p.transformation_epsilon = 0.01, p.max_iterations = 64;
    if (method != HGS_FAST_GICP) p.resolution = resolution;
    hgs_handle* h = nullptr;
    if (hgs_create(&p, &h) != HGS_OK) {
      std::fprintf(stderr, "hgs_create: %s\n", hgs_last_error(nullptr));
      return 1;
    }
    t_abi = drive(scans, warmup, delta_trans, [&](const Cloud::Ptr& c) { hgs_set_target(h, c->points.data(), c->size(), sizeof(PointT)); },
                  [&](const Cloud::Ptr& c, const Eigen::Matrix4f& guess, Eigen::Matrix4f& T) {
                    hgs_result r;
                    const auto t0 = Clock::now();
                    if (hgs_set_source(h, c->points.data(), c->size(), sizeof(PointT)) != HGS_OK) return false;
                    const auto t1 = Clock::now();
                    if (hgs_align(h, guess.data(), &r) != HGS_OK) return false;
                    const auto t2 = Clock::now();
                    abi_set_source_ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
                    abi_align_ms.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
                    abi_iterations.push_back((double)(method == HGS_NDT_OMP ? r.lm_tries : r.iterations));
                    std::memcpy(T.data(), r.final_transformation, sizeof(float) * 16);
                    return r.converged != 0;
                  });
    hgs_destroy(h);
  }
  // ---- B-D: through the adapter, held by the base pointer like the nodelet holds it
  using Hip = hgs_hip::RegistrationHIP<PointT, PointT>;
  auto adapter_run = [&](Hip::AlignedCloudMode mode, bool eager_tree) {
    auto hip = std::make_shared<Hip>(method, 0);
    configure(*hip);
    hip->setAlignedCloudMode(mode);
    pcl::Registration<PointT, PointT>::Ptr registration = hip;
    if (eager_tree) registration->setSearchMethodTarget(std::make_shared<pcl::search::KdTree<PointT>>());  // PCL's default tree: built by initCompute() on every new target
    return drive(scans, warmup, delta_trans, [&](const Cloud::Ptr& c) { registration->setInputTarget(c); },
                 [&](const Cloud::Ptr& c, const Eigen::Matrix4f& guess, Eigen::Matrix4f& T) {
                   registration->setInputSource(c);
                   Cloud::Ptr aligned(new Cloud());  // :209
                   registration->align(*aligned, guess);
                   T = registration->getFinalTransformation();
                   return registration->hasConverged();
                 });
  };
  const Trace t_adapter = adapter_run(Hip::ALIGNED_CLOUD_DEVICE, false), t_host = adapter_run(Hip::ALIGNED_CLOUD_HOST, false), t_noout = adapter_run(Hip::ALIGNED_CLOUD_NONE, false),
              t_eager = adapter_run(Hip::ALIGNED_CLOUD_DEVICE, true);
  // ---- E: pcl::Registration::align on its own
  Trace t_null;
  {
    pcl::Registration<PointT, PointT>::Ptr registration = std::make_shared<NullEngine>();
    registration->setSearchMethodTarget(std::make_shared<hgs_hip::LazyKdTree<PointT>>());
    t_null = drive(scans, warmup, 1e30f, [&](const Cloud::Ptr& c) { registration->setInputTarget(c); },
                   [&](const Cloud::Ptr& c, const Eigen::Matrix4f& guess, Eigen::Matrix4f& T) {
                     registration->setInputSource(c);
                     Cloud::Ptr aligned(new Cloud());
                     registration->align(*aligned, guess);
                     T = registration->getFinalTransformation();
                     return true;
                   });
  }
  // the adapter must return the C-ABI's poses bit for bit (same library, same inputs, same guesses)
  double max_diff = 0;
  for (size_t i = 0; i < std::min(t_abi.poses.size(), t_adapter.poses.size()); i++)
    for (int k = 0; k < 16; k++) max_diff = std::max(max_diff, (double)std::fabs(t_abi.poses[i].data()[k] - t_adapter.poses[i].data()[k]));

  std::printf("{\"points_per_sweep\": %zu, \"sweeps_in_stream\": %zu, ", mean_pts, scans.size() - 1);
  print_trace("c_abi", t_abi);
  print_trace("adapter", t_adapter);
  print_trace("adapter_aligned_cloud_on_host", t_host);
  print_trace("adapter_without_aligned_cloud", t_noout);
  print_trace("adapter_with_eager_cpu_kdtree", t_eager);
  print_trace("pcl_align_alone", t_null);
  std::printf("\"c_abi_calls\": {\"hgs_set_source_p50_ms\": %.4f, \"hgs_align_p50_ms\": %.4f, \"passes_or_iterations_p50\": %.1f}, ", pct(abi_set_source_ms, 0.5), pct(abi_align_ms, 0.5),
              pct(abi_iterations, 0.5));
  std::printf("\"max_abs_pose_diff_adapter_vs_c_abi\": %.3g}\n", max_diff);
  return 0;
}
