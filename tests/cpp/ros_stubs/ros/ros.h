#pragma once
// stand-in for <ros/ros.h>: see README.md in this directory (type-check only)
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "console.h"
namespace ros {
struct Duration {
  Duration() {}
  explicit Duration(double) {}
  double toSec() const { return 0; }
  bool sleep() const { return true; }
  bool operator>(const Duration &) const { return false; }
  bool operator<(const Duration &) const { return false; }
};
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() {}
  explicit Time(double) {}
  static Time now() { return Time(); }
  double toSec() const { return 0; }
  Duration operator-(const Time &) const { return Duration(); }
  Time operator+(const Duration &) const { return Time(); }
  bool operator>(const Time &) const { return false; }
  bool operator<(const Time &) const { return false; }
  bool operator>=(const Time &) const { return false; }
  bool operator<=(const Time &) const { return false; }
  bool operator==(const Time &) const { return true; }
};
struct Rate { explicit Rate(double) {} bool sleep() { return true; } };
struct TimerEvent {};
struct Timer { void stop() {} void start() {} void setPeriod(const Duration &) {} };
struct Publisher {
  template <class M> void publish(const M &) const {}
  uint32_t getNumSubscribers() const { return 0; }
};
struct Subscriber { void shutdown() {} };
struct ServiceServer {};
struct NodeHandle {
  NodeHandle() {}
  explicit NodeHandle(const std::string &) {}
  template <class M> Publisher advertise(const std::string &, uint32_t, bool = false) { return Publisher(); }
  template <class M, class T> Subscriber subscribe(const std::string &, uint32_t, void (T::*)(const M &), T *) { return Subscriber(); }
  template <class M, class T> Subscriber subscribe(const std::string &, uint32_t, void (T::*)(M), T *) { return Subscriber(); }
  template <class T, class Req, class Res> ServiceServer advertiseService(const std::string &, bool (T::*)(Req &, Res &), T *) { return ServiceServer(); }
  template <class T> Timer createTimer(Duration, void (T::*)(const TimerEvent &), T *, bool = false, bool = true) { return Timer(); }
  template <class V> bool getParam(const std::string &, V &) const { return false; }
  template <class V> bool param(const std::string &, V &, const V &) const { return false; }
};
namespace param {
template <class V> bool get(const std::string &, V &) { return false; }
}
namespace service {
inline bool waitForService(const std::string &, Duration = Duration()) { return false; }
template <class S> bool call(const std::string &, S &) { return false; }
}
inline void init(int &, char **, const std::string &) {}
inline bool ok() { return false; }
inline void spin() {}
inline void spinOnce() {}
inline void shutdown() {}
}  // namespace ros
