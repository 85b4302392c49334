// clusterer_nodelet_ros.cpp — the scene_flow_clusterer nodelet of the reference (plugin scene_flow_clusterer/scene_flow_clusterer,
// base nodelet::Nodelet: nodelet_plugins.xml:3-4, clusterer_nodelet.cpp:5) with clustering() on an MI355X.  Same subscription
// (`scene_flow`, queue 10, on the nodelet's single-threaded callback queue, clusterer_nodelet.cpp:25,35), same `~moving_objects`
// topic gated on getNumSubscribers() (:237-238), same dynamic_reconfigure server (:27-29).  `~clusters` (MarkerArray) and
// `~clusters_image` are debugging views drawn from the label plane this call can return; they are not rendered here.
// Not buildable in the image this was written in (no ROS): tests/test_ros_adapter_syntax.py compiles it against declaration-only
// stand-ins; behaviour lives in the host mirror (moving_object_detector_amd/host/clusterer_nodelet.hpp) and the C ABI, which are tested.
#include <dynamic_reconfigure/server.h>
#include <moving_object_msgs/MovingObjectArray.h>
#include <nodelet/nodelet.h>
#include <pluginlib/class_list_macros.h>
#include <ros/ros.h>
#include <scene_flow_clusterer/ClustererConfig.h>
#include <sensor_msgs/PointCloud2.h>

#include <memory>

#define MOD_HOST_ROS_CONFIG   // ClustererConfig is the generated one
#include "clusterer_nodelet.hpp"   // moving_object_detector_amd/host/ (an include directory of this package's CMakeLists.txt)

namespace scene_flow_clusterer {

class ClustererNodeletRos : public nodelet::Nodelet {
 public:
  ~ClustererNodeletRos() override { impl_.reset(); if (ctx_) mod_destroy(ctx_); }

  void onInit() override {
    ros::NodeHandle &node_handle = getNodeHandle(), &private_node_handle = getPrivateNodeHandle();
    ModConfig cfg{};
    cfg.device = private_node_handle.param("device", 0);
    cfg.max_width = private_node_handle.param("max_width", 1920);
    cfg.max_height = private_node_handle.param("max_height", 1080);
    cfg.max_frames = 1;
    if (mod_create(&cfg, &ctx_) != MOD_OK) { NODELET_FATAL("mod_create failed: no MI355X visible or libmod_sf.so missing"); return; }
    impl_.reset(new ClustererNodelet(ctx_));
    // the first reconfigure callback (fired from setCallback with the .cfg defaults) gives the context its parameters; the image
    // size comes with every cloud (mod_cluster_cloud_host sizes a camera-less context from its arguments)
    reconfigure_server_.reset(new dynamic_reconfigure::Server<ClustererConfig>(private_node_handle));
    reconfigure_server_->setCallback([this](ClustererConfig &config, uint32_t) { impl_->reconfigureCB(config); });
    dynamic_objects_pub_ = private_node_handle.advertise<moving_object_msgs::MovingObjectArray>("moving_objects", 1);
    velocity_pc_sub_ = node_handle.subscribe<sensor_msgs::PointCloud2>("scene_flow", 10, &ClustererNodeletRos::dataCB, this);
  }

 private:
  // dataCB (clusterer_nodelet.cpp:221-242)
  void dataCB(const sensor_msgs::PointCloud2ConstPtr &input_pc_msg) {
    const ros::Time start = ros::Time::now();
    mod_host::PointCloud2 in;
    in.header.seq = input_pc_msg->header.seq; in.header.frame_id = input_pc_msg->header.frame_id;
    in.header.stamp = mod_host::Time(input_pc_msg->header.stamp.sec, input_pc_msg->header.stamp.nsec);
    in.width = input_pc_msg->width; in.height = input_pc_msg->height; in.point_step = input_pc_msg->point_step; in.row_step = input_pc_msg->row_step;
    in.data = input_pc_msg->data;      // (one host copy of the payload; the library reads it from here)
    mod_host::MovingObjectArray out;
    try {
      impl_->dataCB(in, &out);         // clustering() always runs (:231), whoever listens
    } catch (const std::exception &e) {
      NODELET_ERROR_STREAM("clustering failed: " << e.what());
      return;
    }
    if (dynamic_objects_pub_.getNumSubscribers() > 0) {               // publishMovingObjects (:324-343)
      moving_object_msgs::MovingObjectArray m;
      m.header = input_pc_msg->header;
      for (const auto &o : out.moving_object_array) {
        moving_object_msgs::MovingObject mo;
        mo.id = o.id;
        mo.center.position.x = o.center.position[0]; mo.center.position.y = o.center.position[1]; mo.center.position.z = o.center.position[2];
        mo.center.orientation.x = 0; mo.center.orientation.y = 0; mo.center.orientation.z = 0; mo.center.orientation.w = 1;
        mo.velocity.x = o.velocity[0]; mo.velocity.y = o.velocity[1]; mo.velocity.z = o.velocity[2];
        mo.bounding_box.x = o.bounding_box[0]; mo.bounding_box.y = o.bounding_box[1]; mo.bounding_box.z = o.bounding_box[2];
        m.moving_object_array.push_back(mo);
      }
      dynamic_objects_pub_.publish(m);
    }
    NODELET_INFO_STREAM("Process time: " << (ros::Time::now() - start).toSec() << " [s]");
  }

  ModContext *ctx_ = nullptr;
  std::unique_ptr<ClustererNodelet> impl_;
  std::unique_ptr<dynamic_reconfigure::Server<ClustererConfig>> reconfigure_server_;
  ros::Publisher dynamic_objects_pub_;
  ros::Subscriber velocity_pc_sub_;
};

}  // namespace scene_flow_clusterer

PLUGINLIB_EXPORT_CLASS(scene_flow_clusterer::ClustererNodeletRos, nodelet::Nodelet)
