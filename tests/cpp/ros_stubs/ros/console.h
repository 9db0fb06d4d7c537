#pragma once
#include <sstream>
#define ROS_STUB_LOG(...) do { if (false) ::ros::stub_printf(__VA_ARGS__); } while (0)
#define ROS_STUB_STREAM(x) do { if (false) { std::ostringstream ros_stub_ss_; ros_stub_ss_ << x; } } while (0)
namespace ros { inline void stub_printf(const char *, ...) {} }
#define ROS_INFO(...) ROS_STUB_LOG(__VA_ARGS__)
#define ROS_WARN(...) ROS_STUB_LOG(__VA_ARGS__)
#define ROS_ERROR(...) ROS_STUB_LOG(__VA_ARGS__)
#define ROS_DEBUG(...) ROS_STUB_LOG(__VA_ARGS__)
#define ROS_INFO_STREAM(x) ROS_STUB_STREAM(x)
#define ROS_WARN_STREAM(x) ROS_STUB_STREAM(x)
#define ROS_ERROR_STREAM(x) ROS_STUB_STREAM(x)
#define ROS_WARN_THROTTLE(period, ...) ROS_STUB_LOG(__VA_ARGS__)
#define ROS_INFO_THROTTLE(period, ...) ROS_STUB_LOG(__VA_ARGS__)
