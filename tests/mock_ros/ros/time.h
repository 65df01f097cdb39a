// Minimal stand-in for <ros/time.h> (TEST ONLY): ros::Time as roscpp lays it out — two uint32 members `sec`, `nsec` (KeyFrame::save / load,
// src/hdl_graph_slam/keyframe.cpp:27,72, stream them directly) — plus now(), toSec(), comparisons and the difference LoopDetector::matching prints.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <ostream>
namespace ros {
struct Duration {
  double s = 0;
  Duration() = default;
  explicit Duration(double t) : s(t) {}
  Duration(int sec, int nsec) : s((double)sec + 1e-9 * (double)nsec) {}
  double toSec() const { return s; }
};
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() = default;
  Time(uint32_t s, uint32_t ns) : sec(s), nsec(ns) {}
  explicit Time(double t) {
    const double fl = std::floor(t);
    sec = (uint32_t)fl, nsec = (uint32_t)std::lround((t - fl) * 1e9);
    if (nsec >= 1000000000u) sec += 1, nsec -= 1000000000u;
  }
  static Time now() { return Time(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()); }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  Duration operator-(const Time& o) const { return Duration(((double)sec - (double)o.sec) + 1e-9 * ((double)nsec - (double)o.nsec)); }
  bool operator>(const Time& o) const { return sec > o.sec || (sec == o.sec && nsec > o.nsec); }
  bool operator<(const Time& o) const { return o > *this; }
  bool isZero() const { return sec == 0 && nsec == 0; }
  Time operator+(const Duration& d) const { return Time(toSec() + d.s); }
};
inline std::ostream& operator<<(std::ostream& os, const Time& t) { return os << t.sec << "." << t.nsec; }
}  // namespace ros
