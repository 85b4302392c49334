// ros_wire.hpp — ROS-1 wire format (roscpp_serialization) of the messages on the hot path's boundary, without ROS:
//   stereo_msgs/DisparityImage   in   (scene_flow_constructor.cpp:258-277 fills it, disparity_image_processor.cpp:5-15 reads it)
//   sensor_msgs/PointCloud2      out / in: the ~scene_flow topic between the two nodes (scene_flow_constructor.cpp:351-362,
//                                clusterer_nodelet.cpp:221-226), pcl::PointXYZVelocity records (pcl_point_xyz_velocity.h:8-34)
//   moving_object_msgs/MovingObjectArray   out (clusterer_nodelet.cpp:324-343; moving_object_msgs/msg/*.msg)
//   geometry_msgs/Transform      in   (scene_flow_constructor.cpp:249)
// so that recorded traffic (bag payloads, TCPROS bodies) can be fed to / produced by the host mirror byte for byte.
// Format: little-endian scalars; string = uint32 length + bytes; T[] = uint32 count + elements; bool = uint8;
// time = uint32 sec + uint32 nsec.  Field order = the .msg definitions of ROS melodic (std_msgs, sensor_msgs, stereo_msgs,
// geometry_msgs) and the reference's own moving_object_msgs.
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "messages.hpp"

namespace ros_wire {

using Bytes = std::vector<uint8_t>;

class Writer {
 public:
  explicit Writer(Bytes &out) : out_(out) {}
  template <class T> void put(const T &v) { const uint8_t *p = (const uint8_t *)&v; out_.insert(out_.end(), p, p + sizeof(T)); }
  void str(const std::string &s) { put<uint32_t>((uint32_t)s.size()); out_.insert(out_.end(), s.begin(), s.end()); }
  void raw(const void *p, size_t n) { out_.insert(out_.end(), (const uint8_t *)p, (const uint8_t *)p + n); }
 private:
  Bytes &out_;
};

class Reader {
 public:
  Reader(const uint8_t *p, size_t n) : p_(p), end_(p + n) {}
  template <class T> T get() { need(sizeof(T)); T v; memcpy(&v, p_, sizeof(T)); p_ += sizeof(T); return v; }
  std::string str() { const uint32_t n = get<uint32_t>(); need(n); std::string s((const char *)p_, n); p_ += n; return s; }
  const uint8_t *raw(size_t n) { need(n); const uint8_t *q = p_; p_ += n; return q; }
  size_t left() const { return (size_t)(end_ - p_); }
 private:
  void need(size_t n) const { if ((size_t)(end_ - p_) < n) throw std::runtime_error("ros_wire: message truncated"); }
  const uint8_t *p_, *end_;
};

// ---- std_msgs/Header ----
inline void write(Writer &w, const mod_host::Header &h) { w.put(h.seq); w.put(h.stamp.sec); w.put(h.stamp.nsec); w.str(h.frame_id); }
inline void read(Reader &r, mod_host::Header &h) { h.seq = r.get<uint32_t>(); h.stamp.sec = r.get<uint32_t>(); h.stamp.nsec = r.get<uint32_t>(); h.frame_id = r.str(); }

// ---- sensor_msgs/PointCloud2 with the PointXYZVelocity field table pcl::toROSMsg emits ----
struct PointField { const char *name; uint32_t offset; };
constexpr uint8_t kFloat32 = 7;                                     // sensor_msgs/PointField::FLOAT32
constexpr PointField kXYZVelocityFields[6] = {{"x", 0}, {"y", 4}, {"z", 8}, {"vx", 16}, {"vy", 20}, {"vz", 24}};

inline Bytes serialize(const mod_host::PointCloud2 &m) {
  Bytes b;
  b.reserve(m.data.size() + 256);
  Writer w(b);
  write(w, m.header);
  w.put<uint32_t>(m.height); w.put<uint32_t>(m.width);
  w.put<uint32_t>(6);
  for (const PointField &f : kXYZVelocityFields) { w.str(f.name); w.put<uint32_t>(f.offset); w.put<uint8_t>(kFloat32); w.put<uint32_t>(1); }
  w.put<uint8_t>(0);                                                // is_bigendian
  w.put<uint32_t>(m.point_step); w.put<uint32_t>(m.row_step);
  w.put<uint32_t>((uint32_t)m.data.size()); w.raw(m.data.data(), m.data.size());
  w.put<uint8_t>(m.is_dense ? 1 : 0);
  return b;
}
// Accepts any field table that contains x,y,z,vx,vy,vz as FLOAT32 at the PointXYZVelocity offsets (what fromROSMsg maps
// without per-field copies); anything else is an error, not a silent reinterpretation.
inline void deserialize(const uint8_t *p, size_t n, mod_host::PointCloud2 &m) {
  Reader r(p, n);
  read(r, m.header);
  m.height = r.get<uint32_t>(); m.width = r.get<uint32_t>();
  const uint32_t nf = r.get<uint32_t>();
  int found = 0;
  for (uint32_t i = 0; i < nf; i++) {
    const std::string name = r.str();
    const uint32_t off = r.get<uint32_t>();
    const uint8_t type = r.get<uint8_t>();
    const uint32_t count = r.get<uint32_t>();
    for (const PointField &f : kXYZVelocityFields)
      if (name == f.name) {
        if (off != f.offset || type != kFloat32 || count != 1) throw std::runtime_error("ros_wire: field '" + name + "' is not a PointXYZVelocity float");
        found++;
      }
  }
  if (found != 6) throw std::runtime_error("ros_wire: cloud lacks the x,y,z,vx,vy,vz fields");
  if (r.get<uint8_t>() != 0) throw std::runtime_error("ros_wire: big-endian clouds are not supported");
  m.point_step = r.get<uint32_t>(); m.row_step = r.get<uint32_t>();
  const uint32_t len = r.get<uint32_t>();
  const uint8_t *d = r.raw(len);
  m.data.assign(d, d + len);
  m.is_dense = r.get<uint8_t>() != 0;
  if (m.point_step != 32 || (uint64_t)m.row_step * m.height != len) throw std::runtime_error("ros_wire: cloud geometry inconsistent");
}

// ---- stereo_msgs/DisparityImage (owning form: the pixels live in `pixels`, `view.data` points at them) ----
struct DisparityImageMsg {
  mod_host::DisparityImage view;          // header, width, height, f, T, min/max_disparity, data
  mod_host::Header image_header;          // sensor_msgs/Image has a header of its own
  std::vector<float> pixels;
  uint32_t roi[4] = {0, 0, 0, 0};         // valid_window: x_offset, y_offset, height, width
  bool roi_do_rectify = false;
  float delta_d = 0.f;
};
inline Bytes serialize(const DisparityImageMsg &m) {
  Bytes b;
  Writer w(b);
  write(w, m.view.header);
  write(w, m.image_header);
  w.put<uint32_t>((uint32_t)m.view.height); w.put<uint32_t>((uint32_t)m.view.width);
  w.str("32FC1");
  w.put<uint8_t>(0);
  w.put<uint32_t>((uint32_t)m.view.width * 4u);
  const size_t bytes = (size_t)m.view.width * m.view.height * 4;
  w.put<uint32_t>((uint32_t)bytes); w.raw(m.view.data, bytes);
  w.put<float>(m.view.f); w.put<float>(m.view.T);
  for (int i = 0; i < 4; i++) w.put<uint32_t>(m.roi[i]);
  w.put<uint8_t>(m.roi_do_rectify ? 1 : 0);
  w.put<float>(m.view.min_disparity); w.put<float>(m.view.max_disparity); w.put<float>(m.delta_d);
  return b;
}
inline void deserialize(const uint8_t *p, size_t n, DisparityImageMsg &m) {
  Reader r(p, n);
  read(r, m.view.header);
  read(r, m.image_header);
  m.view.height = (int)r.get<uint32_t>(); m.view.width = (int)r.get<uint32_t>();
  const std::string enc = r.str();
  const uint8_t big = r.get<uint8_t>();
  const uint32_t step = r.get<uint32_t>(), len = r.get<uint32_t>();
  // the reference wraps the bytes as cv::Mat_<float> without looking at the encoding (disparity_image_processor.cpp:5-15);
  // anything but packed little-endian 32FC1 would be misread there, so it is refused here
  if (enc != "32FC1" || big != 0 || step != (uint32_t)m.view.width * 4u || (uint64_t)step * (uint32_t)m.view.height != len)
    throw std::runtime_error("ros_wire: disparity image must be packed little-endian 32FC1");
  const uint8_t *d = r.raw(len);
  m.pixels.resize(len / 4);
  memcpy(m.pixels.data(), d, len);
  m.view.data = m.pixels.data();
  m.view.f = r.get<float>(); m.view.T = r.get<float>();
  for (int i = 0; i < 4; i++) m.roi[i] = r.get<uint32_t>();
  m.roi_do_rectify = r.get<uint8_t>() != 0;
  m.view.min_disparity = r.get<float>(); m.view.max_disparity = r.get<float>(); m.delta_d = r.get<float>();
}

// ---- geometry_msgs/Transform ----
inline Bytes serialize(const mod_host::Transform &t) {
  Bytes b;
  Writer w(b);
  for (int i = 0; i < 3; i++) w.put<double>(t.translation[i]);
  for (int i = 0; i < 4; i++) w.put<double>(t.rotation[i]);       // x y z w
  return b;
}
inline void deserialize(const uint8_t *p, size_t n, mod_host::Transform &t) {
  Reader r(p, n);
  for (int i = 0; i < 3; i++) t.translation[i] = r.get<double>();
  for (int i = 0; i < 4; i++) t.rotation[i] = r.get<double>();
}

// ---- moving_object_msgs/MovingObjectArray ----
inline Bytes serialize(const mod_host::MovingObjectArray &m) {
  Bytes b;
  Writer w(b);
  write(w, m.header);
  w.put<uint32_t>((uint32_t)m.moving_object_array.size());
  for (const mod_host::MovingObject &o : m.moving_object_array) {
    w.put<int32_t>(o.id);
    for (int i = 0; i < 3; i++) w.put<double>(o.center.position[i]);
    for (int i = 0; i < 4; i++) w.put<double>(o.center.orientation[i]);
    for (int i = 0; i < 3; i++) w.put<double>(o.velocity[i]);
    for (int i = 0; i < 3; i++) w.put<double>(o.bounding_box[i]);
  }
  return b;
}
inline void deserialize(const uint8_t *p, size_t n, mod_host::MovingObjectArray &m) {
  Reader r(p, n);
  read(r, m.header);
  const uint32_t cnt = r.get<uint32_t>();
  if ((uint64_t)cnt * 108 > r.left()) throw std::runtime_error("ros_wire: object array longer than the message");
  m.moving_object_array.resize(cnt);
  for (mod_host::MovingObject &o : m.moving_object_array) {
    o.id = r.get<int32_t>();
    for (int i = 0; i < 3; i++) o.center.position[i] = r.get<double>();
    for (int i = 0; i < 4; i++) o.center.orientation[i] = r.get<double>();
    for (int i = 0; i < 3; i++) o.velocity[i] = r.get<double>();
    for (int i = 0; i < 3; i++) o.bounding_box[i] = r.get<double>();
  }
}

}  // namespace ros_wire
