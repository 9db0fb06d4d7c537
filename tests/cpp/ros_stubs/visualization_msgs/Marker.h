#pragma once
#include <geometry_msgs/Pose.h>
namespace visualization_msgs {
struct Marker {
  enum { ARROW = 0, CUBE = 1, SPHERE = 2, LINE_STRIP = 4, LINE_LIST = 5, POINTS = 8 };
  enum { ADD = 0, MODIFY = 0, DELETE = 2, DELETEALL = 3 };
  std_msgs::Header header; std::string ns; int32_t id = 0, type = 0, action = 0;
  geometry_msgs::Pose pose; geometry_msgs::Vector3 scale; std_msgs::ColorRGBA color; ros::Duration lifetime;
  std::vector<geometry_msgs::Point> points; std::vector<std_msgs::ColorRGBA> colors;
};
}
