// Minimal stand-in for <pcl/io/pcd_io.h> (TEST ONLY): pcl::io::savePCDFileBinary / loadPCDFile for pcl::PointXYZI as KeyFrame::save / load use
// them (src/hdl_graph_slam/keyframe.cpp:57,139).  Our restatement of PCL's writer: PCD v0.7 header (FIELDS x y z intensity, 4-byte floats,
// VIEWPOINT 0 0 0 1 0 0 0, DATA binary) and 16 bytes per point — PCL's PCDWriter::writeBinary copies the registered fields only, not the
// 32-byte in-memory record.  The reader takes binary files with the fields in any order.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include "../point_cloud.h"
#include "../point_types.h"
namespace pcl {
namespace io {
inline int savePCDFileBinary(const std::string& file, const PointCloud<PointXYZI>& cloud) {
  std::ofstream os(file, std::ios::binary);
  if (!os) return -1;
  const size_t n = cloud.points.size();
  os << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH " << n
     << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
  for (const PointXYZI& p : cloud.points) {
    const float rec[4] = {p.x, p.y, p.z, p.intensity};
    os.write(reinterpret_cast<const char*>(rec), sizeof(rec));
  }
  return os ? 0 : -1;
}
inline int loadPCDFile(const std::string& file, PointCloud<PointXYZI>& cloud) {
  std::ifstream is(file, std::ios::binary);
  if (!is) return -1;
  std::map<std::string, std::vector<std::string>> hdr;
  std::string line;
  while (std::getline(is, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ls(line);
    std::string key, tok;
    ls >> key;
    while (ls >> tok) hdr[key].push_back(tok);
    if (key == "DATA") break;
  }
  if (hdr["DATA"].empty() || hdr["DATA"][0] != "binary") return -1;
  const std::vector<std::string>& fields = hdr["FIELDS"];
  std::vector<int> offset(fields.size());
  int rec = 0;
  for (size_t f = 0; f < fields.size(); f++) {
    offset[f] = rec;
    rec += std::stoi(hdr["SIZE"][f]) * (hdr["COUNT"].empty() ? 1 : std::stoi(hdr["COUNT"][f]));
  }
  const size_t n = (size_t)std::stoll(hdr["POINTS"].empty() ? hdr["WIDTH"][0] : hdr["POINTS"][0]);
  std::vector<char> buf(n * (size_t)rec);
  is.read(buf.data(), (std::streamsize)buf.size());
  cloud.points.assign(n, PointXYZI{});
  for (size_t i = 0; i < n; i++) {
    PointXYZI& p = cloud.points[i];
    p.data3 = 1.0f;
    for (size_t f = 0; f < fields.size(); f++) {
      float v;
      std::memcpy(&v, buf.data() + i * (size_t)rec + offset[f], 4);
      if (fields[f] == "x") p.x = v;
      else if (fields[f] == "y") p.y = v;
      else if (fields[f] == "z") p.z = v;
      else if (fields[f] == "intensity") p.intensity = v;
    }
  }
  return 0;
}
}  // namespace io
}  // namespace pcl
