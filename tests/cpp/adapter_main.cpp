// TEST ONLY — drives adapters/registration_hip.hpp the way the reference's callers drive a pcl::Registration:
//   first frame:  setInputTarget(keyframe)                    apps/scan_matching_odometry_nodelet.cpp:166-174
//   every frame:  setInputSource(filtered); align(*aligned, guess); hasConverged(); getFinalTransformation()   :176-221
// Usage: adapter_main <method 0|2> <target.bin> <source.bin>   (raw PointXYZI records); prints the final transform.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <pcl/point_types.h>
#include "../../adapters/registration_hip.hpp"

using PointT = pcl::PointXYZI;

static pcl::PointCloud<PointT>::Ptr load(const char* path) {
  auto c = std::make_shared<pcl::PointCloud<PointT>>();
  FILE* f = std::fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  PointT p;
  while (std::fread(&p, sizeof(PointT), 1, f) == 1) c->points.push_back(p);
  std::fclose(f);
  return c;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s method target.bin source.bin\n", argv[0]);
    return 2;
  }
  try {
    // what the new factory branch does: construct + setters from the reg_* rosparams
    auto make = [&](int device) {
      auto reg = std::make_shared<hgs_hip::RegistrationHIP<PointT, PointT>>(std::atoi(argv[1]), device);
      reg->setTransformationEpsilon(0.01);
      reg->setMaximumIterations(64);
      if (std::atoi(argv[1]) == HGS_FAST_GICP) {
        reg->setMaxCorrespondenceDistance(2.5);
        reg->setCorrespondenceRandomness(20);
      } else {
        reg->setResolution(1.0);
        reg->setNeighborhoodSearchMethod(HGS_DIRECT7);
      }
      return reg;
    };
    pcl::Registration<PointT, PointT>::Ptr registration = make(0);
    auto keyframe = load(argv[2]);
    auto filtered = load(argv[3]);
    registration->setInputTarget(keyframe);
    registration->setInputSource(filtered);
    pcl::PointCloud<PointT> aligned;
    registration->align(aligned, pcl::MockMatrix4f::Identity());
    const auto T = registration->getFinalTransformation();
    std::printf("converged %d\n", (int)registration->hasConverged());
    for (int i = 0; i < 16; i++) std::printf("%.9g ", T.data()[i]);
    std::printf("\n");
    auto* hip = dynamic_cast<hgs_hip::RegistrationHIP<PointT, PointT>*>(registration.get());
    std::printf("fitness %.12g\n", hip->fitnessScoreHIP());
    std::printf("aligned0 %.6f %.6f %.6f n %zu\n", aligned.points[0].x, aligned.points[0].y, aligned.points[0].z, aligned.size());
    {
      // an engine that cannot be created (device 4096 does not exist) must not throw into the caller: the nodelets test
      // hasConverged() (apps/scan_matching_odometry_nodelet.cpp:214) and expect the guess back
      pcl::Registration<PointT, PointT>::Ptr broken = make(4096);
      broken->setInputTarget(keyframe);
      broken->setInputSource(filtered);
      pcl::PointCloud<PointT> out;
      auto guess = pcl::MockMatrix4f::Identity();
      guess.data()[12] = 0.25f;
      broken->align(out, guess);
      const auto Tb = broken->getFinalTransformation();
      std::printf("no_device converged %d guess_kept %d\n", (int)broken->hasConverged(), (int)(Tb.data()[12] == 0.25f && Tb.data()[0] == 1.0f));
      // ... and the failure must not be permanent: scan_matching_odometry sets the keyframe ONCE (a repeated setInputTarget with the same
      // pointer returns early), so the engine that can finally be created has to pick up the clouds that were set while it could not
      auto* late = dynamic_cast<hgs_hip::RegistrationHIP<PointT, PointT>*>(broken.get());
      late->setDevice(0);
      broken->setInputTarget(keyframe);  // same pointers as before: early returns in a naive adapter
      broken->setInputSource(filtered);
      pcl::PointCloud<PointT> out2;
      broken->align(out2, pcl::MockMatrix4f::Identity());
      const auto Tl = broken->getFinalTransformation();
      bool same = true;
      for (int i = 0; i < 16; i++) same = same && Tl.data()[i] == T.data()[i];
      std::printf("recovered converged %d same_pose %d\n", (int)broken->hasConverged(), (int)same);
    }
    {
      // The CPU kd-tree of pcl::Registration (tree_): align() -> initCompute() must NOT build it (LazyKdTree); the non-virtual
      // getFitnessScore() / getSearchMethodTarget()->nearestKSearch() through the BASE pointer must still work — they build it then, once
      // (apps/scan_matching_odometry_nodelet.cpp:307,316 unpatched), and agree with the device versions.
      auto& builds = pcl::search::KdTree<PointT>::builds_counter();
      std::printf("cpu_tree_builds_after_aligns %ld built %d\n", builds.load(), (int)hip->cpuTreeBuilt());
      const double fit_cpu = registration->getFitnessScore();
      std::printf("cpu_tree_builds_after_getFitnessScore %ld fitness_cpu %.12g\n", builds.load(), fit_cpu);
      std::vector<int> idx;
      std::vector<float> d2;
      registration->getSearchMethodTarget()->nearestKSearch(aligned.points[0], 1, idx, d2);
      std::vector<int> idx_dev;
      std::vector<float> d2_dev;
      pcl::PointCloud<PointT> one;
      one.points.push_back(aligned.points[0]);
      hip->nearestTargetHIP(one, idx_dev, d2_dev);
      std::printf("nn_cpu %d %.9g nn_hip %d %.9g builds %ld\n", idx[0], d2[0], idx_dev[0], d2_dev[0], builds.load());
      // a new target (keyframe switch, :246): align again -> still no second build; the aligned cloud is optional
      auto keyframe2 = std::make_shared<pcl::PointCloud<PointT>>(*filtered);
      registration->setInputTarget(keyframe2);
      // the host form of the aligned cloud (pcl::transformPointCloud's arithmetic on align()'s copy of the input) against the device's
      hip->setAlignedCloudMode("host");
      pcl::PointCloud<PointT> on_host, on_device;
      registration->align(on_host, pcl::MockMatrix4f::Identity());
      hip->setAlignedCloudMode("device");
      registration->align(on_device, pcl::MockMatrix4f::Identity());
      float worst = 0.f;
      for (size_t i = 0; i < on_host.size(); i++)
        worst = std::max(worst, std::max(std::fabs(on_host.points[i].x - on_device.points[i].x), std::max(std::fabs(on_host.points[i].y - on_device.points[i].y), std::fabs(on_host.points[i].z - on_device.points[i].z))));
      std::printf("aligned_cloud host_vs_device n %zu max_abs_diff %.3g\n", on_host.size(), (double)worst);
      hip->setAlignedCloudOutput(false);
      pcl::PointCloud<PointT> untouched;
      registration->align(untouched, pcl::MockMatrix4f::Identity());
      bool is_input = untouched.size() == filtered->size();
      for (size_t i = 0; is_input && i < untouched.size(); i++) is_input = untouched.points[i].x == filtered->points[i].x && untouched.points[i].z == filtered->points[i].z;
      std::printf("new_target converged %d builds %d output_is_input_copy %d\n", (int)registration->hasConverged(), (int)builds.load(), (int)is_input);
    }
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
  return 0;
}
