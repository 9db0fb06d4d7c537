#pragma once
#include <memory>
#include <vector>
#include <std_msgs/Header.h>
namespace std_msgs { struct UInt16MultiArray { std::vector<uint16_t> data; }; typedef std::shared_ptr<const UInt16MultiArray> UInt16MultiArrayConstPtr; }
