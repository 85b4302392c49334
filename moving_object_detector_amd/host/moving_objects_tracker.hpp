// moving_objects_tracker.hpp — host-side mirror of the reference's tracker node, the consumer of the
// hot path's MovingObjectArray:
//   MovingObjectsTracker::movingObjectsCallback / predict / correct   moving_object_tracker/src/moving_objects_tracker.cpp:54-197
//   KalmanTracker (constant-velocity model, state x y vx vy)           moving_object_tracker/include/kalman_tracker.hpp:17-162
//   kkl::alg::KalmanFilter<double, 4, 2, 4>                            kkl/include/kkl/alg/kalman_filter.hpp:62-86
//   kkl::alg::NearestNeighborAssociation + gating distance<>           kkl/include/kkl/alg/nearest_neighbor_association.hpp:32-58,
//                                                                      moving_objects_tracker.cpp:14-31, kkl/math/gaussian.hpp:45-71
// Not data-parallel (tens of 4 x 4 matrix operations per frame): plain C++ on the host, no GPU involved.  ROS, TF lookup and
// boost::any are left out: the caller hands in the frame's transform to the odom frame (identity when the frames coincide) and
// gets the tracked objects back.  Eigen is not available here; the 4 x 4 algebra is written out (inverse by cofactors,
// determinant by expansion) — results agree with an Eigen build to rounding, not bit for bit (parity unpinned for this row).
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#include "messages.hpp"

namespace moving_object_tracker {

struct MovingObjectsTrackerConfig {      // cfg/MovingObjectTracker.cfg:8-10
  double covariance_trace_limit = 0.5;
  int correction_count_limit = 3;
  double object_radius = 0.5;
};

struct Mat4 {
  double m[4][4];
  static Mat4 identity(double s = 1.0) { Mat4 r{}; for (int i = 0; i < 4; i++) r.m[i][i] = s; return r; }
  Mat4 operator*(const Mat4 &o) const { Mat4 r{}; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += m[i][k] * o.m[k][j]; r.m[i][j] = s; } return r; }
  Mat4 operator+(const Mat4 &o) const { Mat4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = m[i][j] + o.m[i][j]; return r; }
  Mat4 operator-(const Mat4 &o) const { Mat4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = m[i][j] - o.m[i][j]; return r; }
  Mat4 transpose() const { Mat4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = m[j][i]; return r; }
  double minor3(int r, int c) const {
    int ri[3], ci[3];
    for (int i = 0, k = 0; i < 4; i++) if (i != r) ri[k++] = i;
    for (int j = 0, k = 0; j < 4; j++) if (j != c) ci[k++] = j;
    const double a = m[ri[0]][ci[0]], b = m[ri[0]][ci[1]], cc = m[ri[0]][ci[2]], d = m[ri[1]][ci[0]], e = m[ri[1]][ci[1]], f = m[ri[1]][ci[2]],
                 g = m[ri[2]][ci[0]], h = m[ri[2]][ci[1]], i2 = m[ri[2]][ci[2]];
    return a * (e * i2 - f * h) - b * (d * i2 - f * g) + cc * (d * h - e * g);
  }
  double determinant() const { double s = 0; for (int j = 0; j < 4; j++) s += ((j & 1) ? -1.0 : 1.0) * m[0][j] * minor3(0, j); return s; }
  Mat4 inverse() const {
    const double det = determinant();
    Mat4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[j][i] = (((i + j) & 1) ? -1.0 : 1.0) * minor3(i, j) / det;
    return r;
  }
};
struct Vec4 {
  double v[4];
  Vec4 operator-(const Vec4 &o) const { return {{v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2], v[3] - o.v[3]}}; }
  Vec4 operator+(const Vec4 &o) const { return {{v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2], v[3] + o.v[3]}}; }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]); }
};
inline Vec4 operator*(const Mat4 &a, const Vec4 &x) { Vec4 r{}; for (int i = 0; i < 4; i++) { double s = 0; for (int k = 0; k < 4; k++) s += a.m[i][k] * x.v[k]; r.v[i] = s; } return r; }
inline double quad_form(const Vec4 &d, const Mat4 &a) { const Vec4 t = a * d; return d.v[0] * t.v[0] + d.v[1] * t.v[1] + d.v[2] * t.v[2] + d.v[3] * t.v[3]; }

// kkl::math (gaussian.hpp:45-51,67-71)
inline double squaredMahalanobisDistance(const Vec4 &mean, const Mat4 &cov, const Vec4 &x) { return quad_form(x - mean, cov.inverse()); }
inline double gaussianProbMul(const Vec4 &mean, const Mat4 &cov, const Vec4 &x) {
  const double sqrtDet = std::sqrt(cov.determinant());
  const double lhs = 1.0 / (std::pow(2.0 * M_PI, 4 / 2.0) * sqrtDet);
  return lhs * std::exp(-0.5 * quad_form(x - mean, cov.inverse()));
}

// KalmanTracker (kalman_tracker.hpp:17-162) over kkl's KalmanFilter<double, 4, 2, 4> (control matrix zero, measurement = identity)
class KalmanTracker {
 public:
  KalmanTracker(long id, const mod_host::Time &time, double px, double py, double vx, double vy, const mod_host::MovingObject &associated)
      : id_(id), last_prediction_time(time), last_correction_time(time), last_associated(associated) {
    transition = Mat4::identity();
    process_noise = Mat4{};                                   // kalman_tracker.hpp:41-43
    process_noise.m[0][0] = process_noise.m[1][1] = 0.003;
    process_noise.m[2][2] = process_noise.m[3][3] = 0.01;
    measurement_noise = Mat4::identity(0.2);                  // :44
    mean_ = {{px, py, vx, vy}};
    cov_ = Mat4::identity(0.1);                               // :50
  }
  void predict(const mod_host::Time &time) {                  // :63-72
    double difftime = mod_host::duration_sec(time, last_prediction_time);
    difftime = std::max(0.001, difftime);
    transition.m[0][2] = difftime;
    transition.m[1][3] = difftime;
    mean_ = transition * mean_;                               // kalman_filter.hpp:62-72 with B u = 0
    cov_ = transition * cov_ * transition.transpose() + process_noise;
    last_prediction_time = time;
  }
  void correct(const mod_host::Time &time, double px, double py, double vx, double vy, const mod_host::MovingObject &associated) {   // :81-92
    const Vec4 z{{px, py, vx, vy}};
    const Mat4 gain = cov_ * (cov_ + measurement_noise).inverse();     // kalman_filter.hpp:78-86 with C = I
    mean_ = mean_ + gain * (z - mean_);
    cov_ = (Mat4::identity() - gain) * cov_;
    last_correction_time = time;
    last_associated = associated;
    correction_count_++;
  }
  long id() const { return id_; }
  int correction_count() const { return correction_count_; }
  const mod_host::Time &lastCorrectionTime() const { return last_correction_time; }
  const mod_host::MovingObject &lastAssociated() const { return last_associated; }
  const Vec4 &mean() const { return mean_; }
  const Mat4 &cov() const { return cov_; }
  double positionCovTrace() const { return cov_.m[0][0] + cov_.m[1][1]; }
  double velocityCovTrace() const { return cov_.m[2][2] + cov_.m[3][3]; }

 private:
  long id_;
  mod_host::Time last_prediction_time, last_correction_time;
  mod_host::MovingObject last_associated;
  int correction_count_ = 0;
  Mat4 transition, process_noise, measurement_noise, cov_;
  Vec4 mean_;
};

struct TrackerCovariance { int32_t id; double covariance[16]; };   // msg/TrackerCovariance.msg: column-major (Eigen storage order)

class MovingObjectsTracker {
 public:
  void reconfigureCallback(const MovingObjectsTrackerConfig &config) { cfg_ = config; }   // moving_objects_tracker.cpp:199-203

  // movingObjectsCallback (:54-134).  `to_odom`: the transform the reference looks up with tf2 (:56-64); objects are moved into
  // the odom frame (tf2::doTransform of the pose and of the velocity vector, :68-75), then predict + correct, then the tracked
  // objects of THIS stamp with at least correction_count_limit corrections are reported (:83-101).
  void movingObjectsCallback(const mod_host::MovingObjectArray &moving_objects, const mod_host::Transform &to_odom,
                             mod_host::MovingObjectArray *tracked, std::vector<TrackerCovariance> *covariances = nullptr) {
    std::vector<mod_host::MovingObject> transformed;
    transformed.reserve(moving_objects.moving_object_array.size());
    double R[3][3];
    rotation_matrix(to_odom.rotation, R);
    for (const auto &o : moving_objects.moving_object_array) {
      mod_host::MovingObject t = o;
      for (int i = 0; i < 3; i++) {
        t.center.position[i] = R[i][0] * o.center.position[0] + R[i][1] * o.center.position[1] + R[i][2] * o.center.position[2] + to_odom.translation[i];
        t.velocity[i] = R[i][0] * o.velocity[0] + R[i][1] * o.velocity[1] + R[i][2] * o.velocity[2];      // a Vector3 only rotates
      }
      quat_mul(to_odom.rotation, o.center.orientation, t.center.orientation);
      transformed.push_back(t);
    }
    const mod_host::Time stamp = moving_objects.header.stamp;
    for (auto &t : trackers_) t->predict(stamp);              // predict (:136-140)
    correct(stamp, transformed);
    if (tracked) {                                             // header: odom frame, stamp of the input (:83-84); seq is left at 0
      tracked->header = mod_host::Header();
      tracked->header.frame_id = odom_frame_id_;
      tracked->header.stamp = stamp;
      tracked->moving_object_array.clear();
    }
    if (covariances) covariances->clear();
    for (auto &t : trackers_) {
      if (t->correction_count() < cfg_.correction_count_limit) continue;
      if (t->lastCorrectionTime().sec != stamp.sec || t->lastCorrectionTime().nsec != stamp.nsec) continue;
      if (tracked) {
        mod_host::MovingObject m = t->lastAssociated();
        m.id = (int32_t)t->id();
        m.center.position[0] = t->mean().v[0]; m.center.position[1] = t->mean().v[1];
        m.velocity[0] = t->mean().v[2]; m.velocity[1] = t->mean().v[3];
        tracked->moving_object_array.push_back(m);
      }
      if (covariances) {
        TrackerCovariance c; c.id = (int32_t)t->id();
        for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) c.covariance[j * 4 + i] = t->cov().m[i][j];
        covariances->push_back(c);
      }
    }
  }
  size_t trackerCount() const { return trackers_.size(); }
  void setOdomFrame(const std::string &frame) { odom_frame_id_ = frame; }   // param "odom_frame", default "odom" (:44)

 private:
  struct Association { int tracker, observation; double distance; };

  // gating distance (:14-31): none when the squared Mahalanobis distance exceeds 3^2 or the state difference 1.5
  static bool distance(const KalmanTracker &t, const mod_host::MovingObject &o, double *d) {
    const Vec4 x{{o.center.position[0], o.center.position[1], o.velocity[0], o.velocity[1]}};
    const double sq = squaredMahalanobisDistance(t.mean(), t.cov(), x);
    if (sq > std::pow(3.0, 2) || (t.mean() - x).norm() > 1.5) return false;
    *d = -gaussianProbMul(t.mean(), t.cov(), x);
    return true;
  }

  void correct(const mod_host::Time &time, const std::vector<mod_host::MovingObject> &objs) {   // :142-197
    std::vector<bool> associated(objs.size(), false);
    // NearestNeighborAssociation::associate (nearest_neighbor_association.hpp:32-58): all gated pairs, sorted by distance, greedy
    std::vector<Association> all;
    if (!trackers_.empty() && !objs.empty())
      for (int i = 0; i < (int)trackers_.size(); i++)
        for (int j = 0; j < (int)objs.size(); j++) { double d; if (distance(*trackers_[i], objs[j], &d)) all.push_back({i, j, d}); }
    std::sort(all.begin(), all.end(), [](const Association &a, const Association &b) { return a.distance < b.distance; });
    while (!all.empty()) {
      const Association pick = all.front();
      associated[pick.observation] = true;
      const auto &o = objs[pick.observation];
      trackers_[pick.tracker]->correct(time, o.center.position[0], o.center.position[1], o.velocity[0], o.velocity[1], o);
      all.erase(std::remove_if(all.begin(), all.end(), [&](const Association &a) { return a.tracker == pick.tracker || a.observation == pick.observation; }), all.end());
    }
    for (size_t i = 0; i < objs.size(); i++) {                // new tracks for detections far from every track (:157-185)
      if (associated[i]) continue;
      bool close_to_tracker = false;
      for (const auto &t : trackers_) {
        const double dx = t->mean().v[0] - objs[i].center.position[0], dy = t->mean().v[1] - objs[i].center.position[1];
        if (std::sqrt(dx * dx + dy * dy) < cfg_.object_radius * 2.0) { close_to_tracker = true; break; }
      }
      if (close_to_tracker) continue;
      trackers_.push_back(std::make_shared<KalmanTracker>(id_gen_++, time, objs[i].center.position[0], objs[i].center.position[1],
                                                           objs[i].velocity[0], objs[i].velocity[1], objs[i]));
    }
    // tracks whose covariance grew too large go (:187-196): std::partition keeps the survivors' relative order unspecified in
    // general; with libstdc++'s forward partition of a random-access range the survivors keep their order only up to swaps — the
    // mirror uses std::partition as the reference does
    auto loc = std::partition(trackers_.begin(), trackers_.end(), [&](const std::shared_ptr<KalmanTracker> &t) { return t->positionCovTrace() < cfg_.covariance_trace_limit; });
    trackers_.erase(loc, trackers_.end());
    loc = std::partition(trackers_.begin(), trackers_.end(), [&](const std::shared_ptr<KalmanTracker> &t) { return t->velocityCovTrace() < cfg_.covariance_trace_limit; });
    trackers_.erase(loc, trackers_.end());
  }

  static void rotation_matrix(const double q[4], double R[3][3]) {   // x y z w, normalised like tf2::Matrix3x3::setRotation (s = 2 / |q|^2)
    const double x = q[0], y = q[1], z = q[2], w = q[3], d = x * x + y * y + z * z + w * w, s = 2.0 / d;
    const double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    R[0][0] = 1.0 - (yy + zz); R[0][1] = xy - wz; R[0][2] = xz + wy;
    R[1][0] = xy + wz; R[1][1] = 1.0 - (xx + zz); R[1][2] = yz - wx;
    R[2][0] = xz - wy; R[2][1] = yz + wx; R[2][2] = 1.0 - (xx + yy);
  }
  static void quat_mul(const double a[4], const double b[4], double out[4]) {   // a * b, x y z w
    out[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    out[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    out[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    out[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  }

  MovingObjectsTrackerConfig cfg_;
  std::string odom_frame_id_ = "odom";
  std::vector<std::shared_ptr<KalmanTracker>> trackers_;
  long id_gen_ = 0;
};

}  // namespace moving_object_tracker
