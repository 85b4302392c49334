// messages.hpp — minimal stand-ins for the ROS message types on the hot path's boundary, so the host mirror compiles
// and runs without ROS.  Field names follow the ROS definitions; a ROS build maps the real messages onto these views
// (ros_adapter/) without copying pixels.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mod_sf.h"

namespace mod_host {

struct Header { uint32_t seq = 0; double stamp = 0.0; std::string frame_id; };

// stereo_msgs/DisparityImage: 32FC1 image + f, T, min/max_disparity (disparity_image_processor.cpp:5,25-27,41-42)
struct DisparityImage {
  Header header;
  int width = 0, height = 0;
  const float *data = nullptr;      // row-major, step == width * 4
  float f = 0.f, T = 0.f, min_disparity = 0.f, max_disparity = 0.f;
};

// sensor_msgs/CameraInfo: only the projection matrix P is used (image_geometry::PinholeCameraModel::fromCameraInfo)
struct CameraInfo { int width = 0, height = 0; double P[12] = {0}; };

// cv_bridge::CvImage with encoding 32FC2: optical flow, x then y
struct FlowImage { Header header; int width = 0, height = 0; const float *data = nullptr; };

// geometry_msgs/Transform
struct Transform { double translation[3] = {0, 0, 0}; double rotation[4] = {0, 0, 0, 1}; /* x y z w */ };

// sensor_msgs/PointCloud2 carrying pcl::PointXYZVelocity records (point_step 32; x@0 y@4 z@8 vx@16 vy@20 vz@24)
struct PointCloud2 {
  Header header;
  uint32_t width = 0, height = 0, point_step = 32, row_step = 0;
  bool is_dense = true;
  std::vector<uint8_t> data;
};

// moving_object_msgs/MovingObject(Array) (moving_object_msgs/msg/*.msg)
struct MovingObject {
  int32_t id = 0;
  struct { double position[3]; double orientation[4]; } center;
  double velocity[3];
  double bounding_box[3];
};
struct MovingObjectArray { Header header; std::vector<MovingObject> moving_object_array; };

inline MovingObject to_message(const ModObject &o) {
  MovingObject m;
  m.id = o.id;
  for (int i = 0; i < 3; i++) { m.center.position[i] = o.center[i]; m.velocity[i] = o.velocity[i]; m.bounding_box[i] = o.bounding_box[i]; }
  for (int i = 0; i < 4; i++) m.center.orientation[i] = o.orientation[i];
  return m;
}

}  // namespace mod_host
