// TEST ONLY — the PATCHED apps/scan_matching_odometry_nodelet.cpp (integration/hdl_graph_slam_hip.patch applied to a scratch copy) included as it is, linked
// with the patched src/hdl_graph_slam/registrations.cpp: the reference's OWN ScanMatchingOdometryNodelet — onInit, initialize_params, cloud_callback, matching,
// publish_odometry and (with a status subscriber) publish_scan_matching_status incl. the USE_HGS_HIP hunk — runs on a stream of sweeps against the stand-in ROS
// graph of tests/mock_ros.
//   odometry_nodelet_main <registration_method> <status subscribers 0|1> [rosparam=value ...] -- <sweep0.bin> <sweep1.bin> ...
// Per sweep it prints the published odometry (position, orientation) and, when the status is subscribed, has_converged / matching_error / inlier_fraction.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include <pcl/common/transforms.h>
#include "apps/scan_matching_odometry_nodelet.cpp"

using PointT = pcl::PointXYZI;

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  hdl_graph_slam::ScanMatchingOdometryNodelet nodelet;
  nodelet.mock_private_nh.params["registration_method"] = argv[1];
  ros::mock::subscribers()["/scan_matching_odometry/status"] = (unsigned)std::atoi(argv[2]);
  int i = 3;
  for (; i < argc && std::string(argv[i]) != "--"; i++) {
    const std::string kv = argv[i];
    const size_t eq = kv.find('=');
    nodelet.mock_private_nh.params[kv.substr(0, eq)] = kv.substr(eq + 1);
  }
  nodelet.onInit();
  int k = 0;
  for (i++; i < argc; i++, k++) {
    auto msg = std::make_shared<sensor_msgs::PointCloud2>();
    std::ifstream f(argv[i], std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    msg->mock_payload.points.resize(raw.size() / sizeof(PointT));
    std::memcpy(msg->mock_payload.points.data(), raw.data(), msg->mock_payload.points.size() * sizeof(PointT));
    msg->header.frame_id = "velodyne";
    msg->header.stamp = ros::Time(100.0 + 0.1 * k);
    const sensor_msgs::PointCloud2ConstPtr cmsg = msg;
    ros::mock::published().clear();
    ros::mock::callbacks().at("/filtered_points")(&cmsg);
    const auto& odom = ros::mock::published()["/odom"];
    if (odom.size() != 1) return 3;
    const auto& o = *static_cast<const nav_msgs::Odometry*>(odom[0].get());
    std::printf("odom %d %.9g %.9g %.9g %.9g %.9g %.9g %.9g frame %s child %s cpu_tree_builds %ld\n", k, o.pose.pose.position.x, o.pose.pose.position.y, o.pose.pose.position.z,
                o.pose.pose.orientation.w, o.pose.pose.orientation.x, o.pose.pose.orientation.y, o.pose.pose.orientation.z, o.header.frame_id.c_str(), o.child_frame_id.c_str(),
                pcl::search::KdTree<PointT>::builds_counter().load());
    const auto& st = ros::mock::published()["/scan_matching_odometry/status"];
    for (const auto& s : st) {
      const auto& m = *static_cast<const hdl_graph_slam::ScanMatchingStatus*>(s.get());
      std::printf("status %d converged %d matching_error %.9g inlier_fraction %.9g\n", k, (int)m.has_converged, (double)m.matching_error, (double)m.inlier_fraction);
    }
  }
  return 0;
}
