#pragma once
#include <geometry_msgs/Pose.h>
namespace sensor_msgs { struct PointCloud { std_msgs::Header header; std::vector<geometry_msgs::Point32> points; }; }
