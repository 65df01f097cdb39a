// Minimal stand-in for <boost/format.hpp> (TEST ONLY): boost::format("%.3f") % value, streamed (loop_detector.hpp:158).
#pragma once
#include <cstdio>
#include <iostream>
#include <string>
namespace boost {
class format {
public:
  explicit format(const char* f) : fmt_(f) {}
  format& operator%(double v) {
    char buf[128];
    std::snprintf(buf, sizeof(buf), fmt_.c_str(), v);
    out_ = buf;
    return *this;
  }
  friend std::ostream& operator<<(std::ostream& os, const format& f) { return os << f.out_; }

private:
  std::string fmt_, out_;
};
}  // namespace boost
