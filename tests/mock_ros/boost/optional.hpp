// Minimal stand-in for <boost/optional.hpp> (TEST ONLY): the members of hdl_graph_slam::KeyFrame that are boost::optional (keyframe.hpp:45-49).
#pragma once
#include <optional>
namespace boost {
template <typename T>
using optional = std::optional<T>;
}  // namespace boost
