#pragma once
#include <ros/ros.h>
namespace nodelet {
typedef std::map<std::string, std::string> M_string;
typedef std::vector<std::string> V_string;
class Loader { public: explicit Loader(bool provide_ros_api = true); bool load(const std::string &name, const std::string &type, const M_string &remappings, const V_string &my_argv); };
}
