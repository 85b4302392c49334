#pragma once
#include <geometry_msgs/Transform.h>
#include <std_msgs/Header.h>
namespace moving_object_msgs {
struct MovingObject { int32_t id = 0; geometry_msgs::Pose center; geometry_msgs::Vector3 velocity, bounding_box; };   // msg/MovingObject.msg:3-7
struct MovingObjectArray { std_msgs::Header header; std::vector<MovingObject> moving_object_array; };               // msg/MovingObjectArray.msg:1-2
}
