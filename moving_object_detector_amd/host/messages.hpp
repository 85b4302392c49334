// messages.hpp — minimal stand-ins for the ROS message types on the hot path's boundary, so the host mirror compiles
// and runs without ROS.  Field names follow the ROS definitions; a ROS build maps the real messages onto these views
// (ros_adapter/) without copying pixels.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mod_sf.h"

namespace mod_host {

// ros::Time: seconds + nanoseconds.  The frame interval of the velocity (scene_flow_constructor.cpp:162-164,200-202) is
// `(stamp_now - stamp_previous).toSec()`: integer arithmetic on (sec, nsec), normalised to 0 <= nsec < 1e9, then
// `(double)sec + 1e-9 * (double)nsec` (roscpp_core: rostime/duration.h, impl/duration.h) — NOT the difference of two doubles,
// which would round differently; velocities are only bit-exact with the reference if dt is formed this way.
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() = default;
  Time(uint32_t s, uint32_t ns) : sec(s), nsec(ns) {}
  static Time fromSec(double t) {                       // ros::TimeBase::fromSec
    Time r;
    r.sec = (uint32_t)std::floor(t);
    r.nsec = (uint32_t)std::floor((t - (double)r.sec) * 1e9 + 0.5);   // boost::math::round for a non-negative value
    r.sec += r.nsec / 1000000000u; r.nsec %= 1000000000u;
    return r;
  }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
};
// (a - b).toSec() of two ros::Time values
inline double duration_sec(const Time &a, const Time &b) {
  int64_t sec = (int64_t)a.sec - (int64_t)b.sec, nsec = (int64_t)a.nsec - (int64_t)b.nsec;
  while (nsec >= 1000000000LL) { nsec -= 1000000000LL; ++sec; }     // normalizeSecNSecSigned
  while (nsec < 0) { nsec += 1000000000LL; --sec; }
  return (double)sec + 1e-9 * (double)nsec;
}

struct Header { uint32_t seq = 0; Time stamp; std::string frame_id; };

// stereo_msgs/DisparityImage: 32FC1 image + f, T, min/max_disparity (disparity_image_processor.cpp:5,25-27,41-42)
struct DisparityImage {
  Header header;
  int width = 0, height = 0;
  const float *data = nullptr;      // row-major, step == width * 4
  float f = 0.f, T = 0.f, min_disparity = 0.f, max_disparity = 0.f;
};

// sensor_msgs/CameraInfo: only the projection matrix P is used (image_geometry::PinholeCameraModel::fromCameraInfo)
struct CameraInfo { int width = 0, height = 0; double P[12] = {0}; };

// sensor_msgs/Image, encoding mono8 (what the disparity estimator consumes): row-major, step == width
struct Image { Header header; int width = 0, height = 0; const uint8_t *data = nullptr; };

// cv_bridge::CvImage with encoding 32FC2: optical flow, x then y
struct FlowImage { Header header; int width = 0, height = 0; const float *data = nullptr; };

// geometry_msgs/Transform
struct Transform { double translation[3] = {0, 0, 0}; double rotation[4] = {0, 0, 0, 1}; /* x y z w */ };

// The libviso2 hand-off (scene_flow_constructor.cpp:232-249): VisualOdometryStereo::getMotion() returns the 4x4 motion of the
// left camera from the previous to the current frame; the reference wraps it as tf2::Transform(Matrix3x3, Vector3) and stores
// tf2::toMsg() of it, i.e. the rotation goes through tf2::Matrix3x3::getRotation() (geometry2 0.6.x, LinearMath/Matrix3x3.h —
// not vendored by the reference; restated here, all in double) before construct() rebuilds a matrix from that quaternion.
// Feeding the kernels anything but this quaternion (e.g. the matrix itself) would differ from the reference in the last bits.
inline Transform transform_from_motion(const double m[4][4]) {
  Transform t;
  for (int i = 0; i < 3; i++) t.translation[i] = m[i][3];
  double q[4];
  const double trace = m[0][0] + m[1][1] + m[2][2];
  if (trace > 0.0) {
    double s = std::sqrt(trace + 1.0);
    q[3] = s * 0.5;
    s = 0.5 / s;
    q[0] = (m[2][1] - m[1][2]) * s;
    q[1] = (m[0][2] - m[2][0]) * s;
    q[2] = (m[1][0] - m[0][1]) * s;
  } else {
    const int i = m[0][0] < m[1][1] ? (m[1][1] < m[2][2] ? 2 : 1) : (m[0][0] < m[2][2] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[i] = s * 0.5;
    s = 0.5 / s;
    q[3] = (m[k][j] - m[j][k]) * s;
    q[j] = (m[j][i] + m[i][j]) * s;
    q[k] = (m[k][i] + m[i][k]) * s;
  }
  for (int i = 0; i < 4; i++) t.rotation[i] = q[i];     // x y z w
  return t;
}

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
