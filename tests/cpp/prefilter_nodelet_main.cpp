// TEST ONLY — the PATCHED apps/prefiltering_nodelet.cpp (integration/hdl_graph_slam_hip.patch applied to a scratch copy; tests/integration_build.py) is
// included as it is: the nodelet class lives in that .cpp.  The stand-in ROS graph (tests/mock_ros/ros/ros.h) hands the test the callback the nodelet
// subscribed with and the clouds it published, so the reference's own PrefilteringNodelet::onInit / initialize_params / cloud_callback RUN:
//   prefilter_nodelet_main <cloud.bin> <out_device.bin> <out_cpu.bin> [rosparam=value ...]
// runs one sweep twice — with the backend (the USE_HGS_HIP hunk: one hgs_prefilter call) and with use_hip_prefilter=false (the nodelet's own filter
// chain; PCL's filters are the oracle's restatement behind PCL's interface, tests/mock_pcl/pcl/filters/filter.h) — and writes both published clouds.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "apps/prefiltering_nodelet.cpp"

using PointT = pcl::PointXYZI;

static pcl::PointCloud<PointT> load(const char* path) {
  pcl::PointCloud<PointT> c;
  std::ifstream f(path, std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  c.points.resize(raw.size() / sizeof(PointT));
  std::memcpy(c.points.data(), raw.data(), c.points.size() * sizeof(PointT));
  c.header.frame_id = "velodyne", c.header.stamp = 1600000000123456ull;
  return c;
}

static size_t run(const pcl::PointCloud<PointT>& cloud, int argc, char** argv, bool hip, const char* out_path) {
  hdl_graph_slam::PrefilteringNodelet nodelet;
  for (int i = 4; i < argc; i++) {
    const std::string kv = argv[i];
    const size_t eq = kv.find('=');
    nodelet.mock_private_nh.params[kv.substr(0, eq)] = kv.substr(eq + 1);
  }
  nodelet.mock_private_nh.params["use_hip_prefilter"] = hip ? "true" : "false";
  ros::mock::published().clear();
  nodelet.onInit();
  ros::mock::callbacks().at("/velodyne_points")(&cloud);
  const auto& pub = ros::mock::published()["/filtered_points"];
  if (pub.size() != 1) return (size_t)-1;
  const auto& out = *static_cast<const pcl::PointCloud<PointT>*>(pub[0].get());
  std::ofstream os(out_path, std::ios::binary);
  os.write(reinterpret_cast<const char*>(out.points.data()), (std::streamsize)(out.points.size() * sizeof(PointT)));
  if (out.header.frame_id != cloud.header.frame_id || out.header.stamp != cloud.header.stamp) return (size_t)-2;
  return out.points.size();
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const pcl::PointCloud<PointT> cloud = load(argv[1]);
  auto& cache = hgs_hip::ResidentCloudsHIP<PointT>::instance();
  const size_t n_dev = run(cloud, argc, argv, true, argv[2]);
  const size_t calls = cache.device_calls();
  const size_t n_cpu = run(cloud, argc, argv, false, argv[3]);
  std::printf("prefilter_nodelet in %zu device %zd cpu %zd device_calls %zu device_calls_after_cpu_run %zu\n", cloud.size(), (ssize_t)n_dev, (ssize_t)n_cpu, calls, cache.device_calls());
  if (!cache.last_error().empty()) std::printf("error %s\n", cache.last_error().c_str());
  return 0;
}
