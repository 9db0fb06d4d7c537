#pragma once
#include <geometry_msgs/Pose.h>
namespace pose_graph_tools_msgs {
struct PoseGraphEdge {
  enum { ODOM = 0, LOOPCLOSE = 1, LANDMARK = 2, REJECTED_LOOPCLOSE = 3, MESH = 4, POSE_MESH = 5, MESH_POSE = 6 };
  std_msgs::Header header; uint64_t key_from = 0, key_to = 0; int32_t robot_from = 0, robot_to = 0; int32_t type = 0;
  geometry_msgs::Pose pose; double covariance[36] = {0};
};
struct PoseGraphNode { std_msgs::Header header; int32_t robot_id = 0; uint64_t key = 0; geometry_msgs::Pose pose; };
struct PoseGraph { std_msgs::Header header; std::vector<PoseGraphNode> nodes; std::vector<PoseGraphEdge> edges; };
typedef std::shared_ptr<const PoseGraph> PoseGraphConstPtr;
}
