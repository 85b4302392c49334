#pragma once
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct Image { std_msgs::Header header; uint32_t height = 0, width = 0; std::string encoding; uint8_t is_bigendian = 0; uint32_t step = 0; std::vector<uint8_t> data; };
typedef std::shared_ptr<Image const> ImageConstPtr;
}
