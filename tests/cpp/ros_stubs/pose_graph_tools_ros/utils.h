#pragma once
#include <pose_graph_tools_msgs/PoseGraph.h>
