#pragma once
#include <ros/ros.h>
namespace message_filters {
template <class M> class Subscriber { public: void subscribe(ros::NodeHandle &nh, const std::string &topic, uint32_t queue_size); };
}
