#pragma once
#include <ros/ros.h>
namespace nodelet {
class Nodelet { public: virtual ~Nodelet() {} virtual void onInit() = 0; protected: ros::NodeHandle &getNodeHandle() const; ros::NodeHandle &getPrivateNodeHandle() const; };
}
#define NODELET_FATAL(...) ((void)0)
#define NODELET_ERROR_STREAM(args) do { std::ostringstream nodelet_stub_stream_; nodelet_stub_stream_ << args; } while (0)
#define NODELET_INFO_STREAM(args) do { std::ostringstream nodelet_stub_stream_; nodelet_stub_stream_ << args; } while (0)
