// CPU test of host/ros_wire.hpp: known-answer bytes written out by hand from the .msg definitions, round trips, and the
// rejects.  Also pins ros::Time / ros::Duration arithmetic of host/messages.hpp.  Prints "ok" and exits 0.
#include "../../moving_object_detector_amd/host/ros_wire.hpp"

#include <cmath>
#include <cstdio>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

static void le32(std::vector<uint8_t> &b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
static void f64(std::vector<uint8_t> &b, double v) { uint64_t u; memcpy(&u, &v, 8); for (int i = 0; i < 8; i++) b.push_back((uint8_t)(u >> (8 * i))); }
static void f32(std::vector<uint8_t> &b, float v) { uint32_t u; memcpy(&u, &v, 4); le32(b, u); }
static void str(std::vector<uint8_t> &b, const char *s) { le32(b, (uint32_t)strlen(s)); while (*s) b.push_back((uint8_t)*s++); }

int main() {
  using namespace mod_host;
  // ---- time ----
  CHECK(duration_sec(Time(100, 350000000u), Time(100, 250000000u)) == 0.1);
  CHECK(duration_sec(Time(101, 50000000u), Time(100, 950000000u)) == 0.1);          // borrow across the second
  CHECK(duration_sec(Time(100, 0), Time(100, 1)) == -1.0 + 1e-9 * 999999999.0);      // negative: sec = -1, nsec = 999999999
  CHECK(Time::fromSec(100.35).sec == 100 && Time::fromSec(100.35).nsec == 350000000u);
  CHECK(Time::fromSec(1.9999999999).sec == 2 && Time::fromSec(1.9999999999).nsec == 0);

  // ---- MovingObjectArray: known answer ----
  MovingObjectArray moa;
  moa.header.seq = 7; moa.header.stamp = Time(1600000000u, 123456789u); moa.header.frame_id = "left_camera";
  MovingObject o;
  o.id = 3;
  o.center.position[0] = 1.5; o.center.position[1] = -2.25; o.center.position[2] = 8.0;
  o.center.orientation[0] = 0; o.center.orientation[1] = 0; o.center.orientation[2] = 0; o.center.orientation[3] = 1;
  o.velocity[0] = 0.5; o.velocity[1] = 0.0; o.velocity[2] = -1.0;
  o.bounding_box[0] = 1.0; o.bounding_box[1] = 2.0; o.bounding_box[2] = 0.5;
  moa.moving_object_array.push_back(o);
  std::vector<uint8_t> want;
  le32(want, 7); le32(want, 1600000000u); le32(want, 123456789u); str(want, "left_camera");
  le32(want, 1); le32(want, 3);
  for (double v : {1.5, -2.25, 8.0, 0.0, 0.0, 0.0, 1.0, 0.5, 0.0, -1.0, 1.0, 2.0, 0.5}) f64(want, v);
  const ros_wire::Bytes got = ros_wire::serialize(moa);
  CHECK(got == want);
  CHECK(got.size() == 12 + 4 + 11 + 4 + 108);
  MovingObjectArray back;
  ros_wire::deserialize(got.data(), got.size(), back);
  CHECK(back.header.frame_id == "left_camera" && back.header.stamp.nsec == 123456789u && back.moving_object_array.size() == 1);
  CHECK(back.moving_object_array[0].id == 3 && back.moving_object_array[0].velocity[2] == -1.0 && back.moving_object_array[0].center.orientation[3] == 1.0);
  bool threw = false;
  try { ros_wire::deserialize(got.data(), got.size() - 5, back); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);

  // ---- PointCloud2 of PointXYZVelocity: 2 x 1 points, known answer ----
  PointCloud2 pc;
  pc.header.seq = 1; pc.header.stamp = Time(10, 20); pc.header.frame_id = "cam";
  pc.width = 2; pc.height = 1; pc.point_step = 32; pc.row_step = 64; pc.is_dense = true;
  const float pts[16] = {1, 2, 3, 0, 0.1f, 0.2f, 0.3f, 0, 4, 5, 6, 0, -0.1f, -0.2f, -0.3f, 0};
  pc.data.assign((const uint8_t *)pts, (const uint8_t *)pts + 64);
  std::vector<uint8_t> w2;
  le32(w2, 1); le32(w2, 10); le32(w2, 20); str(w2, "cam");
  le32(w2, 1); le32(w2, 2);                                   // height, width
  le32(w2, 6);
  const char *names[6] = {"x", "y", "z", "vx", "vy", "vz"};
  const uint32_t offs[6] = {0, 4, 8, 16, 20, 24};
  for (int i = 0; i < 6; i++) { str(w2, names[i]); le32(w2, offs[i]); w2.push_back(7); le32(w2, 1); }
  w2.push_back(0);                                            // is_bigendian
  le32(w2, 32); le32(w2, 64); le32(w2, 64);
  for (float v : pts) f32(w2, v);
  w2.push_back(1);                                            // is_dense
  const ros_wire::Bytes g2 = ros_wire::serialize(pc);
  CHECK(g2 == w2);
  PointCloud2 pcb;
  ros_wire::deserialize(g2.data(), g2.size(), pcb);
  CHECK(pcb.width == 2 && pcb.height == 1 && pcb.row_step == 64 && pcb.data == pc.data && pcb.is_dense && pcb.header.frame_id == "cam");
  {   // a cloud whose vx sits at another offset is refused
    std::vector<uint8_t> bad = w2;
    const size_t pos = 12 + 4 + 3 + 8 + 4 + 3 * (4 + 1 + 4 + 1 + 4) + 4 + 2;   // offset field of "vx"
    bad[pos] = 12;
    threw = false;
    try { ros_wire::deserialize(bad.data(), bad.size(), pcb); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
  }

  // ---- DisparityImage round trip + layout facts ----
  ros_wire::DisparityImageMsg dm;
  dm.view.header.seq = 5; dm.view.header.stamp = Time(3, 4); dm.view.header.frame_id = "left";
  dm.image_header = dm.view.header;
  dm.view.width = 3; dm.view.height = 2;
  dm.pixels = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
  dm.view.data = dm.pixels.data();
  dm.view.f = 700.f; dm.view.T = 0.12f; dm.view.min_disparity = 0.f; dm.view.max_disparity = 128.f; dm.delta_d = 0.0625f;
  dm.roi[2] = 2; dm.roi[3] = 3;
  const ros_wire::Bytes g3 = ros_wire::serialize(dm);
  const size_t hdr = 12 + 4 + 4;
  CHECK(g3.size() == 2 * hdr + 8 + (4 + 5) + 1 + 4 + 4 + 24 + 8 + 16 + 1 + 12);
  ros_wire::DisparityImageMsg db;
  ros_wire::deserialize(g3.data(), g3.size(), db);
  CHECK(db.view.width == 3 && db.view.height == 2 && db.pixels == dm.pixels && db.view.data == db.pixels.data());
  CHECK(db.view.f == 700.f && db.view.T == 0.12f && db.view.max_disparity == 128.f && db.delta_d == 0.0625f && db.roi[3] == 3);
  {   // encoding other than 32FC1 is refused
    std::vector<uint8_t> bad = g3;
    bad[2 * hdr + 8 + 4] = '1';                               // "12FC1"
    threw = false;
    try { ros_wire::deserialize(bad.data(), bad.size(), db); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
  }

  // ---- Transform ----
  Transform tf;
  tf.translation[0] = 0.01; tf.translation[2] = 0.08; tf.rotation[1] = 0.0034906514; tf.rotation[3] = 0.9999939;
  const ros_wire::Bytes g4 = ros_wire::serialize(tf);
  CHECK(g4.size() == 56);
  Transform tb;
  ros_wire::deserialize(g4.data(), g4.size(), tb);
  CHECK(memcmp(&tf, &tb, sizeof(tf)) == 0);
  // ---- viso2 motion matrix -> geometry_msgs/Transform (tf2::Matrix3x3::getRotation restated) ----
  {
    auto rot = [](const double q[4], double m[4][4]) {       // rotation matrix of a unit quaternion x y z w
      const double x = q[0], y = q[1], z = q[2], w = q[3];
      const double r[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                              {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                              {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
      for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m[i][j] = (i < 3 && j < 3) ? r[i][j] : (i == j ? 1.0 : 0.0);
    };
    double m[4][4];
    const double ident[4] = {0, 0, 0, 1};
    rot(ident, m); m[0][3] = 0.01; m[1][3] = -0.002; m[2][3] = 0.08;
    Transform t = transform_from_motion(m);
    CHECK(t.rotation[0] == 0 && t.rotation[1] == 0 && t.rotation[2] == 0 && t.rotation[3] == 1);
    CHECK(t.translation[0] == 0.01 && t.translation[1] == -0.002 && t.translation[2] == 0.08);
    // half turns have trace -1: one case per branch of the largest-diagonal selection
    const double hx[4] = {1, 0, 0, 0}, hy[4] = {0, 1, 0, 0}, hz[4] = {0, 0, 1, 0};
    rot(hx, m); t = transform_from_motion(m); CHECK(t.rotation[0] == 1 && t.rotation[1] == 0 && t.rotation[2] == 0 && t.rotation[3] == 0);
    rot(hy, m); t = transform_from_motion(m); CHECK(t.rotation[0] == 0 && t.rotation[1] == 1 && t.rotation[2] == 0 && t.rotation[3] == 0);
    rot(hz, m); t = transform_from_motion(m); CHECK(t.rotation[0] == 0 && t.rotation[1] == 0 && t.rotation[2] == 1 && t.rotation[3] == 0);
    // small ego-motion rotations (the case on the road) and arbitrary ones come back to within rounding, up to sign
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0; };
    for (int it = 0; it < 20000; it++) {
      double q[4] = {rnd(), rnd(), rnd(), rnd()};
      if (it % 2) { q[0] *= 0.01; q[1] *= 0.01; q[2] *= 0.01; q[3] = 1.0; }
      const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      if (nrm < 1e-3) continue;
      for (double &v : q) v /= nrm;
      rot(q, m);
      t = transform_from_motion(m);
      const double sgn = (t.rotation[0] * q[0] + t.rotation[1] * q[1] + t.rotation[2] * q[2] + t.rotation[3] * q[3]) < 0 ? -1.0 : 1.0;
      for (int i = 0; i < 4; i++) CHECK(std::fabs(sgn * t.rotation[i] - q[i]) < 1e-13);
    }
    // a pinned value: yaw 0.4 deg about y (the synthetic sequence's nominal ego rotation), computed by hand with the formulas above
    const double a = 0.4 * 3.14159265358979323846 / 180.0;
    const double qy[4] = {0, std::sin(a / 2), 0, std::cos(a / 2)};
    rot(qy, m);
    t = transform_from_motion(m);
    const double tr = m[0][0] + m[1][1] + m[2][2], s0 = std::sqrt(tr + 1.0);
    CHECK(t.rotation[3] == s0 * 0.5 && t.rotation[1] == (m[0][2] - m[2][0]) * (0.5 / s0) && t.rotation[0] == 0 && t.rotation[2] == 0);
  }
  printf("ok\n");
  return 0;
}
