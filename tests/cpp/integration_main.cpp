// TEST ONLY — compiled against the PATCHED reference sources (integration/hdl_graph_slam_hip.patch applied to a scratch copy of
// src/hdl_graph_slam/registrations.cpp and include/hdl_graph_slam/loop_detector.hpp; tests/test_integration_patch.py) and the stand-in headers
// of tests/mock_ros, tests/mock_pcl, tests/mock_eigen.  It calls the reference's OWN entry points:
//   factory:  hdl_graph_slam::select_registration_method(pnh) with registration_method = FAST_GICP_HIP / FAST_VGICP_HIP / NDT_HIP (and NDT_OMP, the
//             untouched default branch), then setInputTarget / setInputSource / align like apps/scan_matching_odometry_nodelet.cpp:166-221;
//   loop:     hdl_graph_slam::LoopDetector(pnh).detect(keyframes, new_keyframes, graph) — the patched matching() runs all candidates as one device
//             batch; with reg_hip_num_devices beyond the box's GPU count the matcher cannot be created and the SAME call takes the reference's
//             sequential loop through the adapter (align + the non-virtual getFitnessScore on PCL's lazily built CPU tree).
//   f1:       hdl_graph_slam::InformationMatrixCalculator(nh).calc_information_matrix / ::calc_fitness_score — the patched translation unit (device path
//             over resident keyframe clouds) next to the UNPATCHED one compiled under the name InformationMatrixCalculatorCPU;
//   f3:       hdl_graph_slam::MapCloudGenerator().generate(snapshots, resolution) — patched (device) next to the unpatched MapCloudGeneratorCPU;
//   f2:       hgs_hip::ResidentCloudsHIP::prefilter with the parameter mapping the apps/prefiltering_nodelet.cpp hunk uses.
// Usage: integration_main factory <target.bin> <source.bin> [reg_regularization_method]
//        integration_main loop <registration_method> <hip_num_devices> <target.bin> <guesses.bin> <cand0.bin> [cand1.bin ...]
//        integration_main loop_dup ...        the same with candidate 0 listed twice: the device batch is refused, the reference's sequential loop must run
//        integration_main loop_lru ...        the same with reg_hip_resident_keyframes = 2: keyframes are evicted and uploaded again between detections
//        integration_main infomat <cloud1.bin> <cloud2.bin> <16 doubles row-major relpose> <max_range|max>
//        integration_main mapcloud <resolution> <poses.bin: K x 16 floats column-major> <out_device.bin> <out_cpu.bin> <kf0.bin> [kf1.bin ...]
//        integration_main prefilter <cloud.bin> <out.bin> <downsample_method> <resolution> <outlier_removal_method>
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
#include <hdl_graph_slam/information_matrix_calculator.hpp>
#include <hdl_graph_slam/map_cloud_generator.hpp>
#include <resident_clouds_hip.hpp>
// the unpatched translation units are linked in under other class names (tests/integration_build.py): declare those classes from the same headers
#undef INFORMATION_MATRIX_CALCULATOR_HPP
#define InformationMatrixCalculator InformationMatrixCalculatorCPU
#include <hdl_graph_slam/information_matrix_calculator.hpp>
#undef InformationMatrixCalculator
#undef MAP_CLOUD_GENERATOR_HPP
#define MapCloudGenerator MapCloudGeneratorCPU
#include <hdl_graph_slam/map_cloud_generator.hpp>
#undef MapCloudGenerator

using PointT = pcl::PointXYZI;

// ---- the few members of the reference's classes that live in .cpp files this test does not build (keyframe.cpp / graph_slam.cpp need g2o + PCL io)
namespace hdl_graph_slam {
KeyFrame::KeyFrame(const ros::Time& stamp, const Eigen::Isometry3d& odom, double accum_distance, const pcl::PointCloud<PointT>::ConstPtr& cloud)
    : stamp(stamp), odom(odom), accum_distance(accum_distance), cloud(cloud), node(nullptr) {}
KeyFrame::~KeyFrame() {}
long KeyFrame::id() const { return node->id(); }
Eigen::Isometry3d KeyFrame::estimate() const { return node->estimate(); }
KeyFrameSnapshot::KeyFrameSnapshot(const Eigen::Isometry3d& pose, const pcl::PointCloud<PointT>::ConstPtr& cloud) : pose(pose), cloud(cloud) {}
KeyFrameSnapshot::~KeyFrameSnapshot() {}
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

static int run_factory(const char* target_path, const char* source_path, const char* regularization) {
  auto keyframe = load(target_path), filtered = load(source_path);
  for (const char* name : {"FAST_GICP_HIP", "FAST_VGICP_HIP", "NDT_HIP", "NDT_OMP"}) {
    ros::NodeHandle pnh;
    pnh.params["registration_method"] = name;
    pnh.params["reg_resolution"] = "1.0";
    if (regularization) pnh.params["reg_regularization_method"] = regularization;
    auto registration = hdl_graph_slam::select_registration_method(pnh);
    const bool is_hip = dynamic_cast<hgs_hip::RegistrationHIP<PointT, PointT>*>(registration.get()) != nullptr;
    registration->setInputTarget(keyframe);
    registration->setInputSource(filtered);
    pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
    registration->align(*aligned, Eigen::Matrix4f::Identity());
    std::printf("factory %s hip %d converged %d cpu_tree_builds %ld", name, (int)is_hip, (int)registration->hasConverged(), pcl::search::KdTree<PointT>::builds_counter().load());
    if (auto hip = dynamic_cast<hgs_hip::RegistrationHIP<PointT, PointT>*>(registration.get())) std::printf(" regularization %d", hip->params().regularization_method);
    std::printf("\n");
    print_pose("pose", registration->getFinalTransformation());
  }
  return 0;
}

static int run_loop(int argc, char** argv) {
  const std::string variant = argv[1];
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
  if (variant == "loop_lru") pnh.params["reg_hip_resident_keyframes"] = "2";
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
  if (variant == "loop_dup") keyframes.push_back(keyframes.front());  // the same keyframe twice: hgs_loop_match_batch refuses a duplicated cloud
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

// f1: the patched InformationMatrixCalculator (device) and the unpatched one (CPU: the reference's own loop over PCL's kd-tree) on the same pair
static int run_infomat(char** argv) {
  auto c1 = load(argv[2]), c2 = load(argv[3]);
  Eigen::Isometry3d relpose = Eigen::Isometry3d::Identity();
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) relpose.matrix()(r, c) = std::strtod(argv[4 + r * 4 + c], nullptr);
  const bool dflt = std::string(argv[20]) == "max";
  const double max_range = dflt ? 0.0 : std::strtod(argv[20], nullptr);
  ros::NodeHandle nh;
  hdl_graph_slam::InformationMatrixCalculator calc(nh);
  hdl_graph_slam::InformationMatrixCalculatorCPU calc_cpu(nh);
  auto& cache = hgs_hip::ResidentCloudsHIP<PointT>::instance();
  for (int pass = 0; pass < 2; pass++) {  // the second pass finds both clouds (and cloud1's index) resident
    const double s = dflt ? hdl_graph_slam::InformationMatrixCalculator::calc_fitness_score(c1, c2, relpose) : hdl_graph_slam::InformationMatrixCalculator::calc_fitness_score(c1, c2, relpose, max_range);
    std::printf("fitness_device %.17g device_calls %zu resident %zu bytes %zu\n", s, cache.device_calls(), cache.resident(), cache.resident_bytes());
  }
  const double s_cpu = dflt ? hdl_graph_slam::InformationMatrixCalculatorCPU::calc_fitness_score(c1, c2, relpose) : hdl_graph_slam::InformationMatrixCalculatorCPU::calc_fitness_score(c1, c2, relpose, max_range);
  std::printf("fitness_cpu %.17g cpu_tree_builds %ld\n", s_cpu, pcl::search::KdTree<PointT>::builds_counter().load());
  const Eigen::MatrixXd inf = calc.calc_information_matrix(c1, c2, relpose), inf_cpu = calc_cpu.calc_information_matrix(c1, c2, relpose);
  std::printf("infomat_device");
  for (int i = 0; i < 6; i++) std::printf(" %.17g", inf(i, i));
  std::printf("\ninfomat_cpu");
  for (int i = 0; i < 6; i++) std::printf(" %.17g", inf_cpu(i, i));
  std::printf("\n");
  if (!cache.last_error().empty()) std::printf("error %s\n", cache.last_error().c_str());
  return 0;
}

static void dump(const char* path, const pcl::PointCloud<PointT>& c) {
  std::ofstream os(path, std::ios::binary);
  os.write(reinterpret_cast<const char*>(c.points.data()), (std::streamsize)(c.points.size() * sizeof(PointT)));
}

// f3: the patched MapCloudGenerator::generate (device) and the unpatched one (CPU) on the same snapshots
static int run_mapcloud(int argc, char** argv) {
  const double resolution = std::strtod(argv[2], nullptr);
  std::ifstream pf(argv[3], std::ios::binary);
  std::vector<char> praw((std::istreambuf_iterator<char>(pf)), std::istreambuf_iterator<char>());
  const size_t K = (size_t)argc - 6;
  if (praw.size() != K * 16 * sizeof(float)) return 4;
  std::vector<hdl_graph_slam::KeyFrameSnapshot::Ptr> snapshots;
  for (size_t k = 0; k < K; k++) {
    Eigen::Matrix4f P;
    std::memcpy(P.data(), praw.data() + k * 16 * sizeof(float), 16 * sizeof(float));
    snapshots.push_back(std::make_shared<hdl_graph_slam::KeyFrameSnapshot>(Eigen::Isometry3d(P.cast<double>()), load(argv[6 + k])));
  }
  auto& cache = hgs_hip::ResidentCloudsHIP<PointT>::instance();
  hdl_graph_slam::MapCloudGenerator gen;
  hdl_graph_slam::MapCloudGeneratorCPU gen_cpu;
  auto dev = gen.generate(snapshots, resolution);
  auto cpu = gen_cpu.generate(snapshots, resolution);
  std::printf("mapcloud device %zu cpu %zu device_calls %zu resident %zu\n", dev ? dev->size() : 0, cpu ? cpu->size() : 0, cache.device_calls(), cache.resident());
  if (!cache.last_error().empty()) std::printf("error %s\n", cache.last_error().c_str());
  if (dev) dump(argv[4], *dev);
  if (cpu) dump(argv[5], *cpu);
  return 0;
}

// f2: what the apps/prefiltering_nodelet.cpp hunk does with its rosparams and one sweep
static int run_prefilter(char** argv) {
  auto src = load(argv[2]);
  const hgs_prefilter_params pp = hgs_hip::ResidentCloudsHIP<PointT>::prefilter_params(argv[4], std::strtod(argv[5], nullptr), argv[6], 20, 1.0, 0.8, 2, true, 1.0, 100.0);
  pcl::PointCloud<PointT> out;
  auto& cache = hgs_hip::ResidentCloudsHIP<PointT>::instance();
  const bool ok = cache.prefilter(*src, pp, nullptr, 0.0, out);
  std::printf("prefilter ok %d in %zu out %zu device_calls %zu\n", (int)ok, src->size(), out.size(), cache.device_calls());
  if (!cache.last_error().empty()) std::printf("error %s\n", cache.last_error().c_str());
  dump(argv[3], out);
  return ok ? 0 : 5;
}

int main(int argc, char** argv) {
  try {
    if (argc >= 4 && std::string(argv[1]) == "factory") return run_factory(argv[2], argv[3], argc >= 5 ? argv[4] : nullptr);
    if (argc >= 7 && std::string(argv[1]).rfind("loop", 0) == 0) return run_loop(argc, argv);
    if (argc == 21 && std::string(argv[1]) == "infomat") return run_infomat(argv);
    if (argc >= 7 && std::string(argv[1]) == "mapcloud") return run_mapcloud(argc, argv);
    if (argc == 7 && std::string(argv[1]) == "prefilter") return run_prefilter(argv);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  std::fprintf(stderr, "usage: %s factory target.bin source.bin | loop method n_devices target.bin guesses.bin cand0.bin ...\n", argv[0]);
  return 2;
}
