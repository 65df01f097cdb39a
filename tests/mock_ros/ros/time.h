// Minimal stand-in for <ros/time.h> (TEST ONLY).
#pragma once
#include <chrono>
namespace ros {
struct Duration {
  double s = 0;
  double toSec() const { return s; }
};
struct Time {
  double s = 0;
  Time() = default;
  explicit Time(double t) : s(t) {}
  static Time now() { return Time(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()); }
  double toSec() const { return s; }
  Duration operator-(const Time& o) const { return Duration{s - o.s}; }
};
}  // namespace ros
