// Drives the host mirror of the reference's tracker (moving_object_detector_amd/host/moving_objects_tracker.hpp) over a scripted
// stream read from stdin and prints the tracked objects per frame; tests/test_tracker.py compares with oracle/tracker_numpy.py.
// input : <frames>; per frame: <sec> <nsec> <n>; per object: x y vx vy payload
// output: per frame one line: <count> then per tracked object: id x y vx vy payload (sorted by id), %.17g
#include <cstdio>
#include <algorithm>
#include "../../moving_object_detector_amd/host/moving_objects_tracker.hpp"

int main() {
  int frames;
  if (scanf("%d", &frames) != 1) return 2;
  moving_object_tracker::MovingObjectsTracker tracker;
  mod_host::Transform identity;
  for (int f = 0; f < frames; f++) {
    unsigned sec, nsec; int n;
    if (scanf("%u %u %d", &sec, &nsec, &n) != 3) return 2;
    mod_host::MovingObjectArray in, out;
    in.header.stamp = mod_host::Time(sec, nsec);
    in.header.frame_id = "zed_left_camera_optical_frame";
    for (int i = 0; i < n; i++) {
      mod_host::MovingObject o{};
      double payload;
      if (scanf("%lf %lf %lf %lf %lf", &o.center.position[0], &o.center.position[1], &o.velocity[0], &o.velocity[1], &payload) != 5) return 2;
      o.center.orientation[3] = 1.0;
      o.bounding_box[0] = payload;                      // rides along untouched: the tracked object must carry its last detection
      in.moving_object_array.push_back(o);
    }
    std::vector<moving_object_tracker::TrackerCovariance> cov;
    tracker.movingObjectsCallback(in, identity, &out, &cov);
    if (cov.size() != out.moving_object_array.size()) return 3;
    // tracked array: odom frame + the input's stamp (moving_objects_tracker.cpp:83-84), not the camera frame of the detections
    if (out.header.frame_id != "odom" || out.header.stamp.sec != sec || out.header.stamp.nsec != nsec) return 4;
    std::sort(out.moving_object_array.begin(), out.moving_object_array.end(), [](const mod_host::MovingObject &a, const mod_host::MovingObject &b) { return a.id < b.id; });
    printf("%zu", out.moving_object_array.size());
    for (const auto &o : out.moving_object_array)
      printf(" %d %.17g %.17g %.17g %.17g %.17g", o.id, o.center.position[0], o.center.position[1], o.velocity[0], o.velocity[1], o.bounding_box[0]);
    printf("\n");
  }
  return 0;
}
