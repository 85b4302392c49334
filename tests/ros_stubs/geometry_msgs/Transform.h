#pragma once
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct Transform { Vector3 translation; Quaternion rotation; };
}
