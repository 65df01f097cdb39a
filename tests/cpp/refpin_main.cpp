// TEST INFRASTRUCTURE — "reference code run here".  Driver around two UNMODIFIED reference translation units,
//   /root/reference/src/hdl_graph_slam/information_matrix_calculator.cpp   (calc_fitness_score :49-80, calc_information_matrix :25-47)
//   /root/reference/src/hdl_graph_slam/keyframe.cpp                        (KeyFrame::save :21-58, KeyFrame::load :60-145)
// compiled where they lie (tests/refpin_build.py) against the stand-in ROS / PCL / Eigen / g2o / boost headers of tests/mock_*.
// What this pins is the arithmetic and the text format those two files contain themselves: the loop, the `nn_dists[0] <= max_range`
// comparison of a SQUARED distance, the double accumulation, the DBL_MAX return, the weight() curve, the token stream of `data`.
// What stays ours: the kd-tree (exact, like FLANN's at eps = 0), pcl::transformPointCloud's order of operations (pcl/common/transforms.h in
// tests/mock_pcl: PCL's, restated from upstream knowledge), the PCD writer, Eigen's print_matrix.  Nothing in the product path includes this.
//
//   refpin_main fitness  <cloud1.bin> <cloud2.bin> <max_range|max> <16 doubles, row-major relpose>
//   refpin_main infomat  <cloud1.bin> <cloud2.bin> <16 doubles> [key=value ...]           (rosparams of the constructor, :10-21)
//   refpin_main kf_save  <directory> <spec>                                                (spec: tokens, see read_spec)
//   refpin_main kf_load  <directory> <node id> <points out .bin>
// Clouds are raw 32-byte pcl::PointXYZI records.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <string>

#include <hdl_graph_slam/information_matrix_calculator.hpp>
#include <hdl_graph_slam/keyframe.hpp>
#include <g2o/core/hyper_graph.h>
#include <g2o/types/slam3d/vertex_se3.h>

using PointT = pcl::PointXYZI;

static pcl::PointCloud<PointT>::Ptr read_cloud(const char* path) {
  pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
  std::ifstream is(path, std::ios::binary | std::ios::ate);
  if (!is) {
    std::fprintf(stderr, "cannot open %s\n", path);
    std::exit(2);
  }
  const size_t bytes = (size_t)is.tellg();
  is.seekg(0);
  c->points.resize(bytes / sizeof(PointT));
  is.read(reinterpret_cast<char*>(c->points.data()), (std::streamsize)(c->points.size() * sizeof(PointT)));
  return c;
}

static Eigen::Isometry3d read_pose(char** argv) {
  Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) T.matrix()(r, c) = std::strtod(argv[r * 4 + c], nullptr);
  return T;
}

static void print_iso(const char* name, const Eigen::Isometry3d& T) {
  std::printf("%s", name);
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) std::printf(" %.17g", T.matrix()(r, c));
  std::printf("\n");
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string cmd = argv[1];
  if (cmd == "fitness" && argc == 5 + 16) {
    auto c1 = read_cloud(argv[2]), c2 = read_cloud(argv[3]);
    const bool dflt = std::strcmp(argv[4], "max") == 0;
    const Eigen::Isometry3d T = read_pose(argv + 5);
    // the default argument of the declaration (information_matrix_calculator.hpp:34) when "max"
    const double s = dflt ? hdl_graph_slam::InformationMatrixCalculator::calc_fitness_score(c1, c2, T)
                          : hdl_graph_slam::InformationMatrixCalculator::calc_fitness_score(c1, c2, T, std::strtod(argv[4], nullptr));
    std::printf("fitness %.17g\n", s);
    return 0;
  }
  if (cmd == "infomat" && argc >= 4 + 16) {
    auto c1 = read_cloud(argv[2]), c2 = read_cloud(argv[3]);
    const Eigen::Isometry3d T = read_pose(argv + 4);
    ros::NodeHandle nh;
    for (int i = 4 + 16; i < argc; i++) {
      const std::string kv = argv[i];
      const size_t eq = kv.find('=');
      nh.params[kv.substr(0, eq)] = kv.substr(eq + 1);
    }
    hdl_graph_slam::InformationMatrixCalculator calc(nh);
    const Eigen::MatrixXd inf = calc.calc_information_matrix(c1, c2, T);
    std::printf("infomat %d %d", inf.rows(), inf.cols());
    for (int r = 0; r < inf.rows(); r++)
      for (int c = 0; c < inf.cols(); c++) std::printf(" %.17g", inf(r, c));
    std::printf("\n");
    return 0;
  }
  if (cmd == "kf_save" && argc == 4) {
    // spec tokens: stamp <sec> <nsec> | estimate <16> | odom <16> | accum_distance <d> | floor_coeffs <4> | utm_coord <3> | acceleration <3> |
    //              orientation <w x y z> | id <n> | cloud <path>      (hexfloat or decimal: strtod reads both)
    std::ifstream spec(argv[3]);
    ros::Time stamp;
    Eigen::Isometry3d odom = Eigen::Isometry3d::Identity(), est = Eigen::Isometry3d::Identity();
    double accum = 0;
    long id = 0;
    pcl::PointCloud<PointT>::Ptr cloud(new pcl::PointCloud<PointT>());
    boost::optional<Eigen::Vector4d> floor;
    boost::optional<Eigen::Vector3d> utm, acc;
    boost::optional<Eigen::Quaterniond> ori;
    std::string tok;
    auto num = [&]() {
      std::string t;
      spec >> t;
      return std::strtod(t.c_str(), nullptr);
    };
    while (spec >> tok) {
      if (tok == "stamp") {
        stamp.sec = (uint32_t)num(), stamp.nsec = (uint32_t)num();
      } else if (tok == "estimate" || tok == "odom") {
        Eigen::Isometry3d& T = tok == "odom" ? odom : est;
        for (int r = 0; r < 4; r++)
          for (int c = 0; c < 4; c++) T.matrix()(r, c) = num();
      } else if (tok == "accum_distance") {
        accum = num();
      } else if (tok == "floor_coeffs") {
        Eigen::Vector4d v;
        for (int i = 0; i < 4; i++) v[i] = num();
        floor = v;
      } else if (tok == "utm_coord" || tok == "acceleration") {
        Eigen::Vector3d v;
        for (int i = 0; i < 3; i++) v[i] = num();
        (tok == "utm_coord" ? utm : acc) = v;
      } else if (tok == "orientation") {
        const double w = num(), x = num(), y = num(), z = num();
        ori = Eigen::Quaterniond(w, x, y, z);
      } else if (tok == "id") {
        id = (long)num();
      } else if (tok == "cloud") {
        std::string p;
        spec >> p;
        cloud = read_cloud(p.c_str());
      }
    }
    g2o::VertexSE3 node;
    node.setId((int)id);
    node.setEstimate(est);
    hdl_graph_slam::KeyFrame kf(stamp, odom, accum, cloud);
    kf.node = &node;
    kf.floor_coeffs = floor, kf.utm_coord = utm, kf.acceleration = acc, kf.orientation = ori;
    kf.save(argv[2]);
    return 0;
  }
  if (cmd == "kf_load" && argc == 5) {
    g2o::HyperGraph graph;
    g2o::VertexSE3 node;
    const int id = std::atoi(argv[3]);
    node.setId(id);
    graph.vertices()[id] = &node;
    hdl_graph_slam::KeyFrame kf(ros::Time(), Eigen::Isometry3d::Identity(), -1, nullptr);
    const bool ok = kf.load(argv[2], &graph);
    std::printf("loaded %d\n", ok ? 1 : 0);
    if (!ok) return 0;
    std::printf("stamp %u %u\n", kf.stamp.sec, kf.stamp.nsec);
    print_iso("estimate", kf.estimate());
    print_iso("odom", kf.odom);
    std::printf("accum_distance %.17g\n", kf.accum_distance);
    if (kf.floor_coeffs) std::printf("floor_coeffs %.17g %.17g %.17g %.17g\n", (*kf.floor_coeffs)[0], (*kf.floor_coeffs)[1], (*kf.floor_coeffs)[2], (*kf.floor_coeffs)[3]);
    if (kf.utm_coord) std::printf("utm_coord %.17g %.17g %.17g\n", (*kf.utm_coord)[0], (*kf.utm_coord)[1], (*kf.utm_coord)[2]);
    if (kf.acceleration) std::printf("acceleration %.17g %.17g %.17g\n", (*kf.acceleration)[0], (*kf.acceleration)[1], (*kf.acceleration)[2]);
    if (kf.orientation) {
      const Eigen::Quaterniond& q = *kf.orientation;
      std::printf("orientation %.17g %.17g %.17g %.17g\n", q.w(), q.x(), q.y(), q.z());
    }
    std::printf("id %ld\n", kf.id());
    std::printf("points %zu\n", kf.cloud->points.size());
    std::ofstream os(argv[4], std::ios::binary);
    os.write(reinterpret_cast<const char*>(kf.cloud->points.data()), (std::streamsize)(kf.cloud->points.size() * sizeof(PointT)));
    return 0;
  }
  std::fprintf(stderr, "usage: refpin_main fitness|infomat|kf_save|kf_load ...\n");
  return 2;
}
