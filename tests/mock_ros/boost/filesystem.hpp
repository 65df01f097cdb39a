// Minimal stand-in for <boost/filesystem.hpp> (TEST ONLY): the two calls of KeyFrame::save (src/hdl_graph_slam/keyframe.cpp:22-24), over std::filesystem.
#pragma once
#include <filesystem>
#include <fstream>
#include <string>
namespace boost {
namespace filesystem {
inline bool is_directory(const std::string& p) { return std::filesystem::is_directory(p); }
inline bool create_directory(const std::string& p) { return std::filesystem::create_directory(p); }
}  // namespace filesystem
}  // namespace boost
