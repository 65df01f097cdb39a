// TEST ONLY — compiled against the PATCHED reference sources (integration/hdl_graph_slam_hip.patch applied to a scratch copy of
// src/hdl_graph_slam/registrations.cpp and include/hdl_graph_slam/loop_detector.hpp; tests/test_integration_patch.py) and the stand-in headers
// of tests/mock_ros, tests/mock_pcl, tests/mock_eigen.  It calls the reference's OWN entry points:
//   factory:  hdl_graph_slam::select_registration_method(pnh) with registration_method = FAST_GICP_HIP / FAST_VGICP_HIP / NDT_HIP (and NDT_OMP, the
//             untouched default branch), then setInputTarget / setInputSource / align like apps/scan_matching_odometry_nodelet.cpp:166-221;
//   loop:     hdl_graph_slam::LoopDetector(pnh).detect(keyframes, new_keyframes, graph) — the patched matching() runs all candidates as one device
//             batch; with reg_hip_num_devices beyond the box's GPU count the matcher cannot be created and the SAME call takes the reference's
//             sequential loop through the adapter (align + the non-virtual getFitnessScore on PCL's lazily built CPU tree).
// Usage: integration_main factory <target.bin> <source.bin>
//        integration_main loop <registration_method> <hip_num_devices> <target.bin> <guesses.bin> <cand0.bin> [cand1.bin ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include <hdl_graph_slam/loop_detector.hpp>

using PointT = pcl::PointXYZI;

// ---- the few members of the reference's classes that live in .cpp files this test does not build (keyframe.cpp / graph_slam.cpp need g2o + PCL io)
namespace hdl_graph_slam {
KeyFrame::KeyFrame(const ros::Time& stamp, const Eigen::Isometry3d& odom, double accum_distance, const pcl::PointCloud<PointT>::ConstPtr& cloud)
    : stamp(stamp), odom(odom), accum_distance(accum_distance), cloud(cloud), node(nullptr) {}
KeyFrame::~KeyFrame() {}
long KeyFrame::id() const { return node->id(); }
Eigen::Isometry3d KeyFrame::estimate() const { return node->estimate(); }
GraphSLAM::GraphSLAM(const std::string&) : robust_kernel_factory(nullptr) {}
GraphSLAM::~GraphSLAM() {}
}  // namespace hdl_graph_slam

static pcl::PointCloud<PointT>::Ptr load(const char* path) {
  auto c = std::make_shared<pcl::PointCloud<PointT>>();
  std::ifstream f(path, std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  c->points.resize(raw.size() / sizeof(PointT));
  std::memcpy(c->points.data(), raw.data(), c->points.size() * sizeof(PointT));
  return c;
}
static void print_pose(const char* tag, const Eigen::Matrix4f& T) {
  std::printf("%s", tag);
  for (int i = 0; i < 16; i++) std::printf(" %.9g", T.data()[i]);
  std::printf("\n");
}

static int run_factory(const char* target_path, const char* source_path) {
  auto keyframe = load(target_path), filtered = load(source_path);
  for (const char* name : {"FAST_GICP_HIP", "FAST_VGICP_HIP", "NDT_HIP", "NDT_OMP"}) {
    ros::NodeHandle pnh;
    pnh.params["registration_method"] = name;
    pnh.params["reg_resolution"] = "1.0";
    auto registration = hdl_graph_slam::select_registration_method(pnh);
    const bool is_hip = dynamic_cast<hgs_hip::RegistrationHIP<PointT, PointT>*>(registration.get()) != nullptr;
    registration->setInputTarget(keyframe);
    registration->setInputSource(filtered);
    pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
    registration->align(*aligned, Eigen::Matrix4f::Identity());
    std::printf("factory %s hip %d converged %d cpu_tree_builds %ld\n", name, (int)is_hip, (int)registration->hasConverged(), pcl::search::KdTree<PointT>::builds_counter().load());
    print_pose("pose", registration->getFinalTransformation());
  }
  return 0;
}

static int run_loop(int argc, char** argv) {
  const std::string method = argv[2];
  auto target = load(argv[4]);
  std::ifstream gf(argv[5], std::ios::binary);
  std::vector<char> graw((std::istreambuf_iterator<char>(gf)), std::istreambuf_iterator<char>());
  const size_t K = (size_t)argc - 6;
  if (graw.size() != K * 16 * sizeof(float)) return 4;
  const float* guesses = reinterpret_cast<const float*>(graw.data());

  ros::NodeHandle pnh;
  pnh.params["registration_method"] = method;
  pnh.params["reg_resolution"] = "1.0";
  pnh.params["reg_hip_num_devices"] = argv[3];
  pnh.params["distance_thresh"] = "1000.0";
  pnh.params["fitness_score_max_range"] = "4.0";
  pnh.params["fitness_score_thresh"] = "1000.0";
  hdl_graph_slam::LoopDetector detector(pnh);
  hdl_graph_slam::GraphSLAM graph;

  // pose-graph nodes: the new keyframe at the origin, candidate k where its initial guess puts it (loop_detector.hpp:137-142 turns them back into the guess)
  std::vector<std::unique_ptr<g2o::VertexSE3>> nodes;
  std::vector<hdl_graph_slam::KeyFrame::Ptr> keyframes;
  for (size_t k = 0; k < K; k++) {
    auto kf = std::make_shared<hdl_graph_slam::KeyFrame>(ros::Time(0.1 * k), Eigen::Isometry3d::Identity(), 1.0 * k, load(argv[6 + k]));
    nodes.emplace_back(new g2o::VertexSE3());
    Eigen::Matrix4f G;
    std::memcpy(G.data(), guesses + 16 * k, 16 * sizeof(float));
    nodes.back()->setEstimate(Eigen::Isometry3d(G.cast<double>()));
    nodes.back()->setId(100 + (int)k);
    kf->node = nodes.back().get();
    keyframes.push_back(kf);
  }
  auto new_keyframe = std::make_shared<hdl_graph_slam::KeyFrame>(ros::Time(100.0), Eigen::Isometry3d::Identity(), 1000.0, target);
  nodes.emplace_back(new g2o::VertexSE3());
  nodes.back()->setId(999);
  new_keyframe->node = nodes.back().get();

  for (int detection = 0; detection < 2; detection++) {  // the second detection finds the candidates resident on the device
    std::deque<hdl_graph_slam::KeyFrame::Ptr> new_keyframes = {new_keyframe};
    new_keyframe->accum_distance = 1000.0 + 100.0 * detection;  // (min_edge_interval: a new loop edge must be far from the last one)
    auto loops = detector.detect(keyframes, new_keyframes, graph);
    std::printf("\ndetection %d loops %zu cpu_tree_builds %ld\n", detection, loops.size(), pcl::search::KdTree<PointT>::builds_counter().load());
    for (const auto& loop : loops) {
      std::printf("matched %ld\n", loop->key2->id() - 100);
      print_pose("relpose", loop->relative_pose);
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  try {
    if (argc >= 4 && std::string(argv[1]) == "factory") return run_factory(argv[2], argv[3]);
    if (argc >= 7 && std::string(argv[1]) == "loop") return run_loop(argc, argv);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  std::fprintf(stderr, "usage: %s factory target.bin source.bin | loop method n_devices target.bin guesses.bin cand0.bin ...\n", argv[0]);
  return 2;
}
