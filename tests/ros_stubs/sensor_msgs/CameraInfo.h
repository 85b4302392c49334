#pragma once
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct CameraInfo { std_msgs::Header header; uint32_t height = 0, width = 0; double K[9] = {}, R[9] = {}, P[12] = {}; };
typedef std::shared_ptr<CameraInfo const> CameraInfoConstPtr;
}
