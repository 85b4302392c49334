// host_mirror_test.cpp — drives the C++ host mirror (SceneFlowConstructor + ClustererNodelet) the way the reference's
// node and nodelet are driven: frame 0 carries no previous disparity (nothing is published), frame 1 produces the cloud,
// which is then fed to the clusterer mirror as a PointCloud2.  Inputs/outputs are raw binary files so that pytest can
// compare against the golden fixtures (tests/test_gpu_host_mirror.py).  Standalone process: no torch, no Python.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../moving_object_detector_amd/host/clusterer_nodelet.hpp"
#include "../../moving_object_detector_amd/host/ros_wire.hpp"
#include "../../moving_object_detector_amd/host/scene_flow_constructor.hpp"

template <class T>
static std::vector<T> read_all(const std::string &p, bool optional = false) {
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) { if (optional) return {}; fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<T> v(n / sizeof(T));
  if (fread(v.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return v;
}
static void write_all(const std::string &p, const void *d, size_t n) {
  FILE *f = fopen(p.c_str(), "wb");
  if (!f || fwrite(d, 1, n, f) != n) exit(3);
  fclose(f);
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: host_mirror_test <dir>\n"); return 1; }
  const std::string dir = argv[1];
  auto cam = read_all<double>(dir + "/cam.f64");     // W H fx fy cx cy Tx Ty f T min max
  auto prm = read_all<double>(dir + "/prm.f64");     // flow_diff cluster_size n depth_diff dynamic_speed
  auto d_now = read_all<float>(dir + "/d_now.f32");
  auto d_prev = read_all<float>(dir + "/d_prev.f32");
  auto flow = read_all<float>(dir + "/flow.f32");
  auto tq = read_all<double>(dir + "/tq.f64");       // t(3) q(4) dt
  const int W = (int)cam[0], H = (int)cam[1];

  ModConfig cfg{};
  cfg.device = 0; cfg.max_width = W; cfg.max_height = H; cfg.max_frames = 1;
  cfg.max_objects = W * H / (int)prm[1] + 1;   // every cluster that can survive the size filter fits
  ModContext *ctx = nullptr;
  if (mod_create(&cfg, &ctx) != MOD_OK) { fprintf(stderr, "mod_create failed\n"); return 4; }

  scene_flow_constructor::SceneFlowConstructor constructor(ctx);
  scene_flow_clusterer::ClustererNodelet clusterer(ctx);
  clusterer.reconfigureCB({(int)prm[1], prm[3], prm[4], (int)prm[2]});
  constructor.reconfigureCB({(int)prm[0], 1.0});

  mod_host::CameraInfo info;
  info.width = W; info.height = H;
  info.P[0] = cam[2]; info.P[5] = cam[3]; info.P[2] = cam[4]; info.P[6] = cam[5]; info.P[3] = cam[6]; info.P[7] = cam[7];
  mod_host::DisparityImage prev, now;
  prev.width = now.width = W; prev.height = now.height = H;
  prev.f = now.f = (float)cam[8]; prev.T = now.T = (float)cam[9];
  prev.min_disparity = now.min_disparity = (float)cam[10]; prev.max_disparity = now.max_disparity = (float)cam[11];
  prev.data = d_prev.data(); now.data = d_now.data();
  // stamps as a ROS driver produces them: (sec, nsec); the interval reaches the kernels as ros::Duration::toSec() forms it
  prev.header.stamp = mod_host::Time(100, 250000000u);
  now.header.stamp = mod_host::Time::fromSec(prev.header.stamp.toSec() + tq[7]);
  constructor.setCameraInfo(info, now);

  mod_host::FlowImage fl;
  fl.width = W; fl.height = H; fl.data = flow.data(); fl.header.stamp = now.header.stamp; fl.header.frame_id = "left_camera";
  mod_host::Transform tf;
  for (int i = 0; i < 3; i++) tf.translation[i] = tq[i];
  for (int i = 0; i < 4; i++) tf.rotation[i] = tq[3 + i];

  mod_host::PointCloud2 cloud;
  mod_host::MovingObjectArray objs_fused, objs_cluster;
  // frame 0: no previous image yet -> no flow, no previous disparity (scene_flow_constructor.cpp:380-384,397-398)
  const bool pub0 = constructor.stereoCallback(&prev, nullptr, nullptr, &cloud, &objs_fused);
  // frame 1
  const bool pub1 = constructor.stereoCallback(&now, &fl, &tf, &cloud, &objs_fused);
  // visual odometry failure on a later frame publishes nothing and keeps running
  mod_host::PointCloud2 dummy;
  const bool pub2 = constructor.stereoCallback(&now, &fl, nullptr, &dummy, nullptr);
  if (pub0 || !pub1 || pub2) { fprintf(stderr, "publish pattern wrong: %d %d %d\n", pub0, pub1, pub2); return 5; }

  // the two nodes talk through the ~scene_flow topic: the cloud crosses it in ROS-1 wire format (host/ros_wire.hpp)
  const ros_wire::Bytes on_the_wire = ros_wire::serialize(cloud);
  mod_host::PointCloud2 received;
  ros_wire::deserialize(on_the_wire.data(), on_the_wire.size(), received);
  if (received.data != cloud.data || received.header.frame_id != cloud.header.frame_id) { fprintf(stderr, "cloud changed on the wire\n"); return 14; }
  std::vector<int32_t> cluster_map;
  clusterer.dataCB(received, &objs_cluster, &cluster_map);
  {   // and the result leaves as a serialised MovingObjectArray
    const ros_wire::Bytes out = ros_wire::serialize(objs_cluster);
    mod_host::MovingObjectArray echoed;
    ros_wire::deserialize(out.data(), out.size(), echoed);
    if (echoed.moving_object_array.size() != objs_cluster.moving_object_array.size()) { fprintf(stderr, "objects changed on the wire\n"); return 15; }
  }
  if (objs_cluster.moving_object_array.size() != objs_fused.moving_object_array.size()) { fprintf(stderr, "object count differs\n"); return 6; }
  if (objs_cluster.header.frame_id != "left_camera") { fprintf(stderr, "header not propagated\n"); return 7; }

  // the pipelined form of the same sequence (submit / collect): frame 0 publishes nothing but its disparity becomes frame 1's
  // previous one; frame 1 gives the same bytes as the synchronous callback; a frame without transform publishes nothing and
  // the frame after it pairs with it, as in the reference (:397-398)
  {
    scene_flow_constructor::SceneFlowConstructor piped(ctx);
    piped.setCameraInfo(info, now);
    mod_host::PointCloud2 c1, c2, c3;
    mod_host::MovingObjectArray o1, o3;
    const int t0 = piped.submit(&prev, nullptr, nullptr, &c1, &o1);
    const int t1 = piped.submit(&now, &fl, &tf, &c1, &o1);
    const int t2 = piped.submit(&now, &fl, nullptr, &c2, nullptr);
    mod_host::DisparityImage later = now;
    later.header.stamp = mod_host::Time::fromSec(now.header.stamp.toSec() + tq[7]);
    const int t3 = piped.submit(&later, &fl, &tf, &c3, &o3);        // previous = `now` (parked by the skipped frame)
    if (t0 != -1 || t1 < 0 || t2 != -1 || t3 < 0) { fprintf(stderr, "ticket pattern wrong: %d %d %d %d\n", t0, t1, t2, t3); return 8; }
    piped.collect(t1);
    piped.collect(t3);
    if (c1.data != cloud.data) { fprintf(stderr, "pipelined cloud differs from the synchronous one\n"); return 9; }
    if (o1.moving_object_array.size() != objs_fused.moving_object_array.size()) { fprintf(stderr, "pipelined object count differs\n"); return 10; }
    for (size_t i = 0; i < o1.moving_object_array.size(); i++)
    {
      const mod_host::MovingObject &a = o1.moving_object_array[i], &b = objs_fused.moving_object_array[i];   // (the struct has padding)
      if (a.id != b.id || memcmp(&a.center, &b.center, sizeof(a.center)) || memcmp(a.velocity, b.velocity, sizeof(a.velocity)) ||
          memcmp(a.bounding_box, b.bounding_box, sizeof(a.bounding_box))) { fprintf(stderr, "pipelined object %zu differs\n", i); return 11; }
    }
    // frame 3 paired `now` with itself shifted by dt: must equal the synchronous path fed the same pair
    mod_host::PointCloud2 c3s;
    scene_flow_constructor::SceneFlowConstructor again(ctx);
    again.setCameraInfo(info, now);
    if (!again.construct(&later, &now, &fl, &tf, &c3s)) { fprintf(stderr, "reference pair not published\n"); return 12; }
    if (c3.data != c3s.data) { fprintf(stderr, "pipelined frame after a skipped one differs\n"); return 13; }
  }

  // SURVEY.md 8(f)2, the collapsed node: the objects of the FUSED path (constructor context, planes still in HBM) are the nodelet
  // path's (PointCloud2 through the wire, clustered again) byte for byte ...
  for (size_t i = 0; i < objs_cluster.moving_object_array.size(); i++) {
    const mod_host::MovingObject &a = objs_fused.moving_object_array[i], &b = objs_cluster.moving_object_array[i];
    if (a.id != b.id || memcmp(&a.center, &b.center, sizeof(a.center)) || memcmp(a.velocity, b.velocity, sizeof(a.velocity)) ||
        memcmp(a.bounding_box, b.bounding_box, sizeof(a.bounding_box))) { fprintf(stderr, "fused object %zu differs from the nodelet's\n", i); return 30; }
  }
  // ... and a frame submitted WITHOUT cluster outputs (nobody subscribes to ~moving_objects) does not launch the cluster stage at
  // all, while one with them launches it once: stage timers of the context
  {
    scene_flow_constructor::SceneFlowConstructor quiet(ctx);
    quiet.setCameraInfo(info, now);
    mod_host::PointCloud2 cq;
    mod_host::MovingObjectArray oq;
    double ms = 0.0; int64_t calls = -1;
    if (mod_set_profiling(ctx, MOD_PROFILE_ALL) != MOD_OK || mod_reset_stage_times(ctx) != MOD_OK) return 31;
    const int q0 = quiet.submit(&prev, nullptr, nullptr, nullptr, nullptr);
    const int q1 = quiet.submit(&now, &fl, &tf, &cq, nullptr);           // the cloud alone
    if (q0 != -1 || q1 < 0) { fprintf(stderr, "quiet ticket pattern wrong: %d %d\n", q0, q1); return 32; }
    quiet.collect(q1);
    if (mod_get_stage_time(ctx, MOD_STAGE_CCL_TILE, &ms, &calls) != MOD_OK || calls != 0) { fprintf(stderr, "cluster stage ran without cluster outputs: %lld calls\n", (long long)calls); return 33; }
    if (mod_get_stage_time(ctx, MOD_STAGE_SCENE_FLOW, &ms, &calls) != MOD_OK || calls != 1) { fprintf(stderr, "scene-flow stage calls: %lld\n", (long long)calls); return 34; }
    if (cq.data != cloud.data) { fprintf(stderr, "cloud of the scene-flow-only submit differs\n"); return 35; }
    mod_host::DisparityImage later = now;
    later.header.stamp = mod_host::Time::fromSec(now.header.stamp.toSec() + tq[7]);
    const int q2 = quiet.submit(&later, &fl, &tf, nullptr, &oq);          // the objects alone: no cloud is packed or copied
    if (q2 < 0) { fprintf(stderr, "objects-only submit refused\n"); return 36; }
    quiet.collect(q2);
    if (mod_get_stage_time(ctx, MOD_STAGE_CCL_TILE, &ms, &calls) != MOD_OK || calls != 1) { fprintf(stderr, "cluster stage calls with objects asked for: %lld\n", (long long)calls); return 37; }
    if (mod_set_profiling(ctx, 0) != MOD_OK) return 38;
  }

  write_all(dir + "/cloud.bin", cloud.data.data(), cloud.data.size());
  write_all(dir + "/labels.i32", cluster_map.data(), cluster_map.size() * 4);
  std::vector<double> o;
  for (const auto &m : objs_cluster.moving_object_array) {
    o.push_back(m.id);
    for (int i = 0; i < 3; i++) o.push_back(m.center.position[i]);
    for (int i = 0; i < 4; i++) o.push_back(m.center.orientation[i]);
    for (int i = 0; i < 3; i++) o.push_back(m.velocity[i]);
    for (int i = 0; i < 3; i++) o.push_back(m.bounding_box[i]);
  }
  write_all(dir + "/objects.f64", o.data(), o.size() * 8);
  mod_destroy(ctx);

  // estimateDisparity() (scene_flow_constructor.cpp:258-279) on a rectified mono8 pair, when the directory holds one
  auto sgm_dims = read_all<double>(dir + "/sgm_dims.f64", /*optional=*/true);
  if (sgm_dims.size() == 2) {
    const int SW = (int)sgm_dims[0], SH = (int)sgm_dims[1];
    auto limg = read_all<uint8_t>(dir + "/sgm_left.u8"), rimg = read_all<uint8_t>(dir + "/sgm_right.u8");
    ModConfig scfg{};
    scfg.device = 0; scfg.max_width = SW; scfg.max_height = SH; scfg.max_frames = 1; scfg.max_objects = 64;
    ModContext *sctx = nullptr;
    if (mod_create(&scfg, &sctx) != MOD_OK) { fprintf(stderr, "mod_create (sgm) failed\n"); return 20; }
    scene_flow_constructor::SceneFlowConstructor est(sctx);
    mod_host::CameraInfo li, ri;
    li.width = ri.width = SW; li.height = ri.height = SH;
    li.P[0] = ri.P[0] = 700.0; li.P[5] = ri.P[5] = 700.0; li.P[2] = ri.P[2] = SW / 2.0; li.P[6] = ri.P[6] = SH / 2.0;
    ri.P[3] = -700.0 * 0.12;                        // right camera: Tx = -fx * baseline
    mod_host::DisparityImage first;
    first.width = SW; first.height = SH; first.f = 700.f; first.T = 0.12f; first.min_disparity = 0.f; first.max_disparity = 127.f;
    est.setCameraInfo(li, first);
    est.reconfigureCB({5, 1.0});                     // (a context wants camera and parameters before any work)
    mod_host::Image left, right;
    left.width = right.width = SW; left.height = right.height = SH; left.data = limg.data(); right.data = rimg.data();
    left.header.stamp = mod_host::Time(7, 5u); left.header.frame_id = "left_camera";
    mod_host::DisparityImage disp;
    std::vector<float> pixels;
    if (!est.estimateDisparity(&left, &right, li, ri, &disp, &pixels)) { fprintf(stderr, "estimateDisparity failed\n"); return 21; }
    if (disp.f != 700.f || disp.T != 0.12f || disp.min_disparity != 0.f || disp.max_disparity != 127.f || disp.data != pixels.data() ||
        disp.header.frame_id != "left_camera") { fprintf(stderr, "disparity message fields wrong\n"); return 22; }
    if (est.estimateDisparity(nullptr, &right, li, ri, &disp, &pixels)) { fprintf(stderr, "a missing image must fail\n"); return 23; }
    write_all(dir + "/sgm_disparity.f32", pixels.data(), pixels.size() * 4);
    // submitStereo(): the same image pair twice (a camera that stands still), a flow that moves a region — the device-resident
    // path must give the bytes of estimateDisparity() + construct() with the disparity carried through the host
    {
      std::vector<float> sflow((size_t)SW * SH * 2, 0.0f);
      for (int y = SH / 4; y < SH / 2; y++) for (int x = SW / 4; x < SW / 2; x++) sflow[((size_t)y * SW + x) * 2] = -9.0f;
      mod_host::FlowImage sfl;
      sfl.width = SW; sfl.height = SH; sfl.data = sflow.data(); sfl.header.frame_id = "left_camera";
      mod_host::Transform still;                      // identity
      mod_host::Image left2 = left, right2 = right;
      left2.header.stamp = right2.header.stamp = mod_host::Time(7, 66666672u);
      sfl.header.stamp = left2.header.stamp;
      mod_host::PointCloud2 ca, cb;
      mod_host::MovingObjectArray oa, ob;
      const int s0 = est.submitStereo(&left, &right, nullptr, nullptr, &ca, &oa);       // first frame: no flow, no previous disparity
      const int s1 = est.submitStereo(&left2, &right2, &sfl, &still, &ca, &oa);
      if (s0 != -1 || s1 < 0) { fprintf(stderr, "submitStereo ticket pattern wrong: %d %d\n", s0, s1); return 24; }
      est.collect(s1);
      mod_host::DisparityImage d0 = disp, d1 = disp;
      d1.header.stamp = left2.header.stamp;
      scene_flow_constructor::SceneFlowConstructor ref(sctx);
      ref.setCameraInfo(li, first);
      if (!ref.construct(&d1, &d0, &sfl, &still, &cb, nullptr, &ob)) { fprintf(stderr, "reference stereo frame not published\n"); return 25; }
      if (ca.data != cb.data) { fprintf(stderr, "submitStereo cloud differs from estimateDisparity + construct\n"); return 26; }
      if (oa.moving_object_array.size() != ob.moving_object_array.size()) { fprintf(stderr, "submitStereo object count differs\n"); return 27; }
      const int s2 = est.submitStereo(nullptr, &right2, &sfl, &still, &ca, &oa);        // a missing image: nothing published ...
      const int s3 = est.submitStereo(&left2, &right2, &sfl, &still, &ca, &oa);         // ... and the next frame has no previous disparity
      if (s2 != -1 || s3 != -1) { fprintf(stderr, "submitStereo after a missing image: %d %d\n", s2, s3); return 28; }
    }
    mod_destroy(sctx);
  }
  printf("ok %zu objects\n", objs_cluster.moving_object_array.size());
  return 0;
}
