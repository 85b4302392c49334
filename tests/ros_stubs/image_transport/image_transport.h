#pragma once
#include <ros/ros.h>
namespace image_transport {
class ImageTransport { public: explicit ImageTransport(const ros::NodeHandle &nh); };
std::string getCameraInfoTopic(const std::string &base_topic);
}
