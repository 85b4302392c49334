#pragma once
#include <ros/ros.h>
namespace nodelet {
class Nodelet { public: virtual ~Nodelet() {} virtual void onInit() = 0; protected: ros::NodeHandle &getNodeHandle() const; ros::NodeHandle &getPrivateNodeHandle() const; };
}
#define NODELET_FATAL(...) ((void)0)
