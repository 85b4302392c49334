#pragma once
#include <ros/ros.h>
namespace dynamic_reconfigure {
template <class ConfigType> class Server { public: explicit Server(const ros::NodeHandle &); void setCallback(const std::function<void(ConfigType &, uint32_t)> &); };
}
