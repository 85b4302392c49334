// scene_flow_constructor_node.cpp — the `scene_flow_constructor` executable of the reference (scene_flow_constructor/CMakeLists.txt:45-48,
// src/scene_flow_constructor_node.cpp:3-9) with the hot path on an MI355X.  Same node name, same subscriptions (left_image /
// right_image + their camera_info, exact-time synchronised, queue 1 — scene_flow_constructor.cpp:51-62), same four private topics
// (~depth, ~optical_flow, ~scene_flow, ~synthetic_optical_flow, :45-48), same dynamic_reconfigure server (:40-42), every publish
// gated on getNumSubscribers() (:99,114,141,144).
//
// What runs where:
//   disparity        on the GPU (libmod_sf's SGM), replacing sgm_gpu::SgmGpu::computeDisparity (:35,267)
//   scene flow       on the GPU, overlapped with the next frame's estimators like construct_thread_ (:389-392): submitStereo() /
//                    collect(); the disparity plane stays in HBM as the next frame's previous one (:397-398)
//   optical flow     CALL-OUT estimateOpticalFlow(): the reference asks pwc_net (:281-291); not part of this package
//   camera motion    CALL-OUT estimateCameraMotion(): the reference runs libviso2 and a TF lookup (:214-256); not part of this package
//   clustering       on the GPU, IN THIS PROCESS, when `~publish_moving_objects` is true (default): the node then advertises
//                    `~moving_objects` itself (what the separate scene_flow_clusterer publishes, clusterer_nodelet.cpp:324-343) from the
//                    planes that are still in HBM — one clustering per frame, no 29.5 MB PointCloud2 hand-off between two processes
//                    (detect_moving_object.launch:19-33 wires constructor -> topic -> clusterer; INTEGRATION.md section 3 has the two
//                    launch-file lines that drop the separate clusterer).  The clusterer's four parameters are read from
//                    `~clusterer/{cluster_size, depth_diff, dynamic_speed, neighbor_distance}` (Clusterer.cfg's names, defaults and ranges) and served by a
//                    second dynamic_reconfigure server in that namespace from this package's own cfg/CollapsedClusterer.cfg — the
//                    reference's clusterer package build-depends on this one, so its generated header cannot be used here.  With the parameter false the node is the reference's
//                    constructor alone: no clustering runs here.
//   ~scene_flow      the 32-byte cloud is packed and copied to the host only while somebody subscribes (:141-142 gates exactly so)
// Not buildable in the image this was written in (no ROS): tests/test_ros_adapter_syntax.py compiles it against declaration-only
// stand-ins of the ROS headers; behaviour lives in the host mirror (moving_object_detector_amd/host/scene_flow_constructor.hpp) and the C ABI, which are tested.
#include <dynamic_reconfigure/server.h>
#include <geometry_msgs/Transform.h>
#include <image_transport/image_transport.h>
#include <image_transport/subscriber_filter.h>
#include <message_filters/subscriber.h>
#include <message_filters/time_synchronizer.h>
#include <moving_object_msgs/MovingObjectArray.h>
#include <ros/ros.h>
#include <scene_flow_constructor/CollapsedClustererConfig.h>   // this package's own cfg/CollapsedClusterer.cfg (no dependency on scene_flow_clusterer)
#include <scene_flow_constructor/SceneFlowConstructorConfig.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/PointCloud2.h>

#include <memory>
#include <string>
#include <vector>

#define MOD_HOST_ROS_CONFIG   // SceneFlowConstructorConfig is the generated one
#include "scene_flow_constructor.hpp"   // moving_object_detector_amd/host/ (an include directory of this package's CMakeLists.txt)

namespace scene_flow_constructor {

class SceneFlowConstructorNode {
 public:
  SceneFlowConstructorNode() : node_handle_(), private_node_handle_("~") {
    ModConfig cfg{};
    cfg.device = private_node_handle_.param("device", 0);
    cfg.max_width = private_node_handle_.param("max_width", 1920);
    cfg.max_height = private_node_handle_.param("max_height", 1080);
    cfg.max_frames = 1;
    if (mod_create(&cfg, &ctx_) != MOD_OK) { ROS_FATAL("mod_create failed: no MI355X visible or libmod_sf.so missing"); ros::shutdown(); return; }
    impl_.reset(new SceneFlowConstructor(ctx_));
    ModSgmParams sgm{};
    sgm.disparities = private_node_handle_.param("disparities", 128); sgm.p1 = private_node_handle_.param("p1", 6);
    sgm.p2 = private_node_handle_.param("p2", 96); sgm.paths = private_node_handle_.param("paths", 8); sgm.lr_check = 1; sgm.median = 1;
    impl_->setDisparityParams(sgm);
    max_disparity_ = (float)(sgm.disparities - 1);

    image_transport_.reset(new image_transport::ImageTransport(private_node_handle_));

    // dynamic reconfigure (scene_flow_constructor.cpp:40-42, :401-407)
    reconfigure_server_.reset(new dynamic_reconfigure::Server<SceneFlowConstructorConfig>(private_node_handle_));
    reconfigure_server_->setCallback([this](SceneFlowConstructorConfig &config, uint32_t) {
      ROS_INFO("Reconfigure Request: dynamic_flow_diff = %d, max_color_velocity = %f", config.dynamic_flow_diff, config.max_color_velocity);
      impl_->reconfigureCB(config);
    });

    // publishers (:45-48)
    depth_pub_ = private_node_handle_.advertise<sensor_msgs::Image>("depth", 1);
    optflow_pub_ = private_node_handle_.advertise<sensor_msgs::Image>("optical_flow", 1);
    pc_with_velocity_pub_ = private_node_handle_.advertise<sensor_msgs::PointCloud2>("scene_flow", 1);
    static_flow_pub_ = private_node_handle_.advertise<sensor_msgs::Image>("synthetic_optical_flow", 1);

    // collapsed mode: this process clusters too and publishes the moving objects (see the header comment)
    publish_moving_objects_ = private_node_handle_.param("publish_moving_objects", true);
    if (publish_moving_objects_) {
      clusterer_node_handle_ = ros::NodeHandle(private_node_handle_, "clusterer");
      clusterer_reconfigure_server_.reset(new dynamic_reconfigure::Server<CollapsedClustererConfig>(clusterer_node_handle_));
      clusterer_reconfigure_server_->setCallback([this](CollapsedClustererConfig &config, uint32_t) {
        impl_->reconfigureClusterer(config.cluster_size, config.depth_diff, config.dynamic_speed, config.neighbor_distance);
      });
      moving_objects_pub_ = private_node_handle_.advertise<moving_object_msgs::MovingObjectArray>("moving_objects", 1);
    }

    // subscribers + exact-time synchroniser, queue 1 (:51-62)
    const std::string left_image_topic = node_handle_.resolveName("left_image"), right_image_topic = node_handle_.resolveName("right_image");
    left_image_sub_.subscribe(*image_transport_, left_image_topic, 1);
    right_image_sub_.subscribe(*image_transport_, right_image_topic, 1);
    left_caminfo_sub_.subscribe(node_handle_, image_transport::getCameraInfoTopic(left_image_topic), 1);
    right_caminfo_sub_.subscribe(node_handle_, image_transport::getCameraInfoTopic(right_image_topic), 1);
    stereo_synchronizer_.reset(new StereoSynchronizer(left_image_sub_, right_image_sub_, left_caminfo_sub_, right_caminfo_sub_, 1));
    stereo_synchronizer_->registerCallback(&SceneFlowConstructorNode::stereoCallback, this);
  }
  ~SceneFlowConstructorNode() { impl_.reset(); if (ctx_) mod_destroy(ctx_); }

 private:
  using StereoSynchronizer = message_filters::TimeSynchronizer<sensor_msgs::Image, sensor_msgs::Image, sensor_msgs::CameraInfo, sensor_msgs::CameraInfo>;

  // ---- CALL-OUTS: the two estimators this package does not contain ------------------------------------------------------------
  // Optical flow previous -> now of the left image as 32FC2 (x then y, indexed at the NOW pixel), or null.  The reference calls
  // pwc_net_.estimateOpticalFlow(previous_left_image_, left_image, left_flow_) (:281-291).
  sensor_msgs::ImageConstPtr estimateOpticalFlow(const sensor_msgs::ImageConstPtr & /*previous_left*/, const sensor_msgs::ImageConstPtr & /*left*/) {
    return sensor_msgs::ImageConstPtr();
  }
  // Camera motion previous -> now in the left optical frame, or null when odometry failed (:251-255).  The reference feeds libviso2
  // and converts its 4 x 4 through tf2 (:214-249); mod_host::transform_from_motion() in the host mirror's messages.hpp restates that conversion.
  std::shared_ptr<geometry_msgs::Transform> estimateCameraMotion(const sensor_msgs::ImageConstPtr & /*left*/, const sensor_msgs::ImageConstPtr & /*right*/,
                                                                   const sensor_msgs::CameraInfoConstPtr & /*left_info*/, const sensor_msgs::CameraInfoConstPtr & /*right_info*/) {
    return std::shared_ptr<geometry_msgs::Transform>();
  }

  static mod_host::Header header_of(const std_msgs::Header &h) {
    mod_host::Header o;
    o.seq = h.seq; o.stamp = mod_host::Time(h.stamp.sec, h.stamp.nsec); o.frame_id = h.frame_id;
    return o;
  }

  // stereoCallback (:365-399)
  void stereoCallback(const sensor_msgs::ImageConstPtr &left_image, const sensor_msgs::ImageConstPtr &right_image,
                      const sensor_msgs::CameraInfoConstPtr &left_camera_info, const sensor_msgs::CameraInfoConstPtr &right_camera_info) {
    const ros::WallTime start_process = ros::WallTime::now();
    if (!camera_set_) {               // first frame: camera model from the left CameraInfo (:368-375); disparity fields as the estimator reports them
      mod_host::CameraInfo info;
      info.width = left_camera_info->width; info.height = left_camera_info->height;
      for (int i = 0; i < 12; i++) info.P[i] = left_camera_info->P[i];
      mod_host::DisparityImage d;
      d.f = (float)left_camera_info->P[0];
      d.T = (float)(-right_camera_info->P[3] / right_camera_info->P[0]);
      d.min_disparity = 0.0f; d.max_disparity = max_disparity_;
      impl_->setCameraInfo(info, d);
      camera_set_ = true;
    }
    // the estimators of THIS frame (the reference runs them on three threads, :378-388); the GPU is meanwhile busy with the
    // previous frame's scene flow
    sensor_msgs::ImageConstPtr left_flow = previous_left_image_ ? estimateOpticalFlow(previous_left_image_, left_image) : sensor_msgs::ImageConstPtr();
    std::shared_ptr<geometry_msgs::Transform> transform_prev2now = estimateCameraMotion(left_image, right_image, left_camera_info, right_camera_info);

    // the frame before has had a whole callback period: fetch it and publish (construct_thread_.join(), :389-390)
    publishPending();

    mod_host::Image l, r;
    l.header = header_of(left_image->header); l.width = left_image->width; l.height = left_image->height; l.data = left_image->data.data();
    r.header = header_of(right_image->header); r.width = right_image->width; r.height = right_image->height; r.data = right_image->data.data();
    mod_host::FlowImage f;
    if (left_flow) { f.header = header_of(left_flow->header); f.width = left_flow->width; f.height = left_flow->height;
                     f.data = reinterpret_cast<const float *>(left_flow->data.data()); }
    mod_host::Transform t;
    if (transform_prev2now) {
      t.translation[0] = transform_prev2now->translation.x; t.translation[1] = transform_prev2now->translation.y; t.translation[2] = transform_prev2now->translation.z;
      t.rotation[0] = transform_prev2now->rotation.x; t.rotation[1] = transform_prev2now->rotation.y;
      t.rotation[2] = transform_prev2now->rotation.z; t.rotation[3] = transform_prev2now->rotation.w;
    }
    if (left_flow && optflow_pub_.getNumSubscribers() > 0) optflow_pub_.publish(left_flow);                 // (:99-100)
    // ~depth and ~synthetic_optical_flow are debugging views the reference renders only for subscribers (:114,141); they need the
    // disparity on the host, which the streaming path avoids: when somebody listens, this frame's disparity is fetched as well
    const bool want_views = depth_pub_.getNumSubscribers() > 0 || static_flow_pub_.getNumSubscribers() > 0;
    // outputs asked for THIS frame, gated like the reference's publishes: the cloud only while ~scene_flow has subscribers
    // (:141-142), the objects while the collapsed mode is on and ~moving_objects has subscribers (clusterer_nodelet.cpp:237-238);
    // with neither, the frame still goes through the scene-flow stage so that its disparity is the next frame's previous one
    pending_cloud_.reset(pc_with_velocity_pub_.getNumSubscribers() > 0 ? new mod_host::PointCloud2() : nullptr);
    pending_objects_.reset(publish_moving_objects_ && moving_objects_pub_.getNumSubscribers() > 0 ? new mod_host::MovingObjectArray() : nullptr);
    pending_ticket_ = impl_->submitStereo(&l, &r, left_flow ? &f : nullptr, transform_prev2now ? &t : nullptr, pending_cloud_.get(), pending_objects_.get());
    pending_header_ = left_flow ? left_flow->header : left_image->header;
    pending_keep_ = {left_image, right_image, left_flow};           // the buffers stay alive until the frame is collected
    if (want_views) publishViews(l, r, *left_camera_info, *right_camera_info, left_image->header, transform_prev2now ? &t : nullptr);

    ROS_INFO("process time: %f", (ros::WallTime::now() - start_process).toSec());
    previous_left_image_ = left_image;
  }

  void publishPending() {
    if (pending_ticket_ < 0) return;
    impl_->collect(pending_ticket_);
    pending_ticket_ = -1;
    if (pending_objects_) {                                           // publishMovingObjects (clusterer_nodelet.cpp:324-343)
      moving_object_msgs::MovingObjectArray m;
      m.header = pending_header_;
      for (const auto &o : pending_objects_->moving_object_array) {
        moving_object_msgs::MovingObject mo;
        mo.id = o.id;
        mo.center.position.x = o.center.position[0]; mo.center.position.y = o.center.position[1]; mo.center.position.z = o.center.position[2];
        mo.center.orientation.x = 0; mo.center.orientation.y = 0; mo.center.orientation.z = 0; mo.center.orientation.w = 1;
        mo.velocity.x = o.velocity[0]; mo.velocity.y = o.velocity[1]; mo.velocity.z = o.velocity[2];
        mo.bounding_box.x = o.bounding_box[0]; mo.bounding_box.y = o.bounding_box[1]; mo.bounding_box.z = o.bounding_box[2];
        m.moving_object_array.push_back(mo);
      }
      moving_objects_pub_.publish(m);
      pending_objects_.reset();
    }
    if (pending_cloud_) {                                             // publishPointcloud (:351-362); asked for because somebody subscribed (:144)
      sensor_msgs::PointCloud2 msg;   // fields x, y, z, vx, vy, vz float32 at 0, 4, 8, 16, 20, 24 (pcl_point_xyz_velocity.h:27-34)
      msg.header = pending_header_; msg.width = pending_cloud_->width; msg.height = pending_cloud_->height;
      msg.point_step = 32; msg.row_step = pending_cloud_->row_step; msg.is_dense = true; msg.is_bigendian = false;
      msg.data = std::move(pending_cloud_->data);
      const char *names[6] = {"x", "y", "z", "vx", "vy", "vz"};
      const uint32_t offs[6] = {0, 4, 8, 16, 20, 24};
      for (int i = 0; i < 6; i++) { sensor_msgs::PointField pf; pf.name = names[i]; pf.offset = offs[i]; pf.datatype = sensor_msgs::PointField::FLOAT32; pf.count = 1; msg.fields.push_back(pf); }
      pc_with_velocity_pub_.publish(msg);
    }
    pending_keep_.clear();
  }

  // ~depth (toDepthImage, :114-121) and ~synthetic_optical_flow (:136-145): synchronous side path, only while somebody subscribes
  void publishViews(const mod_host::Image &l, const mod_host::Image &r, const sensor_msgs::CameraInfo &li, const sensor_msgs::CameraInfo &ri,
                    const std_msgs::Header &header, const mod_host::Transform *t) {
    mod_host::CameraInfo a, b;
    a.width = li.width; a.height = li.height; b.width = ri.width; b.height = ri.height;
    for (int i = 0; i < 12; i++) { a.P[i] = li.P[i]; b.P[i] = ri.P[i]; }
    mod_host::DisparityImage d;
    std::vector<float> pixels;
    if (!impl_->estimateDisparity(&l, &r, a, b, &d, &pixels)) return;
    const size_t n = (size_t)l.width * l.height;
    if (depth_pub_.getNumSubscribers() > 0) {
      sensor_msgs::Image m;
      m.header = header; m.width = l.width; m.height = l.height; m.encoding = "32FC1"; m.step = 4 * l.width; m.data.resize(4 * n);
      if (mod_depth_image_host(ctx_, pixels.data(), reinterpret_cast<float *>(m.data.data())) == MOD_OK) depth_pub_.publish(m);
    }
    if (static_flow_pub_.getNumSubscribers() > 0 && t && !view_prev_disparity_.empty()) {
      ModTransform tf{};
      for (int i = 0; i < 3; i++) tf.t[i] = t->translation[i];
      for (int i = 0; i < 4; i++) tf.q[i] = t->rotation[i];
      sensor_msgs::Image m;
      m.header = header; m.width = l.width; m.height = l.height; m.encoding = "32FC2"; m.step = 8 * l.width; m.data.resize(8 * n);
      if (mod_static_flow_host(ctx_, view_prev_disparity_.data(), &tf, reinterpret_cast<float *>(m.data.data())) == MOD_OK) static_flow_pub_.publish(m);
    }
    view_prev_disparity_.swap(pixels);
  }

  ros::NodeHandle node_handle_, private_node_handle_;
  ModContext *ctx_ = nullptr;
  std::unique_ptr<SceneFlowConstructor> impl_;
  std::unique_ptr<image_transport::ImageTransport> image_transport_;
  std::unique_ptr<dynamic_reconfigure::Server<SceneFlowConstructorConfig>> reconfigure_server_;
  image_transport::SubscriberFilter left_image_sub_, right_image_sub_;
  message_filters::Subscriber<sensor_msgs::CameraInfo> left_caminfo_sub_, right_caminfo_sub_;
  std::unique_ptr<StereoSynchronizer> stereo_synchronizer_;
  ros::Publisher depth_pub_, optflow_pub_, pc_with_velocity_pub_, static_flow_pub_, moving_objects_pub_;
  bool publish_moving_objects_ = true;
  ros::NodeHandle clusterer_node_handle_;
  std::unique_ptr<dynamic_reconfigure::Server<CollapsedClustererConfig>> clusterer_reconfigure_server_;
  std::unique_ptr<mod_host::MovingObjectArray> pending_objects_;
  bool camera_set_ = false;
  float max_disparity_ = 127.0f;
  sensor_msgs::ImageConstPtr previous_left_image_;
  int pending_ticket_ = -1;
  std::unique_ptr<mod_host::PointCloud2> pending_cloud_;
  std_msgs::Header pending_header_;
  std::vector<sensor_msgs::ImageConstPtr> pending_keep_;
  std::vector<float> view_prev_disparity_;
};

}  // namespace scene_flow_constructor

int main(int argc, char **argv) {
  ros::init(argc, argv, "scene_flow_constructor");
  scene_flow_constructor::SceneFlowConstructorNode node;
  ros::spin();
  return 0;
}
