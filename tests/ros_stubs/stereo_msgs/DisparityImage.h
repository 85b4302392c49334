#pragma once
#include <sensor_msgs/Image.h>
namespace stereo_msgs {
struct DisparityImage { std_msgs::Header header; sensor_msgs::Image image; float f = 0, T = 0; float min_disparity = 0, max_disparity = 0, delta_d = 0; };
}
