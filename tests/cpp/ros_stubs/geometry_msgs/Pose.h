#pragma once
#include <std_msgs/Header.h>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Point32 { float x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct PoseArray { std_msgs::Header header; std::vector<Pose> poses; };
}
