#pragma once
#include <pose_graph_tools_msgs/PoseGraph.h>
namespace pose_graph_tools_msgs {
struct PoseGraphQueryRequest { uint16_t robot_id = 0; };
struct PoseGraphQueryResponse { PoseGraph pose_graph; };
struct PoseGraphQuery { typedef PoseGraphQueryRequest Request; typedef PoseGraphQueryResponse Response; Request request; Response response; };
}
