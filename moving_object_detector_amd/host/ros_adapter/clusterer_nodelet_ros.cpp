// clusterer_nodelet_ros.cpp — ROS-1 nodelet shell around scene_flow_clusterer::ClustererNodelet (host mirror).
// Not built in this image (no ROS); tests/test_ros_adapter_syntax.py keeps it compiling against declaration-only stand-ins of
// the ROS types.  Field mapping only.  Plugin identity as in the reference (nodelet_plugins.xml beside this file):
// PLUGINLIB_EXPORT_CLASS(scene_flow_clusterer::ClustererNodeletRos, nodelet::Nodelet) with the class name
// "scene_flow_clusterer/scene_flow_clusterer" in nodelet_plugins.xml.
#include <dynamic_reconfigure/server.h>
#include <moving_object_msgs/MovingObjectArray.h>
#include <nodelet/nodelet.h>
#include <pluginlib/class_list_macros.h>
#include <ros/ros.h>
#include <scene_flow_clusterer/ClustererConfig.h>
#include <sensor_msgs/PointCloud2.h>

#define MOD_HOST_ROS_CONFIG   // ClustererConfig is the generated one
#include "../clusterer_nodelet.hpp"

namespace scene_flow_clusterer {

class ClustererNodeletRos : public nodelet::Nodelet {
 public:
  void onInit() override {
    ros::NodeHandle &nh = getNodeHandle(), &pnh = getPrivateNodeHandle();
    ModConfig cfg{};
    cfg.device = pnh.param("device", 0);
    cfg.max_width = pnh.param("max_width", 1920); cfg.max_height = pnh.param("max_height", 1080); cfg.max_frames = 1;
    if (mod_create(&cfg, &ctx_) != MOD_OK) { NODELET_FATAL("mod_create failed: no MI355X / library"); return; }
    impl_.reset(new ClustererNodelet(ctx_));
    server_.reset(new dynamic_reconfigure::Server<scene_flow_clusterer::ClustererConfig>(pnh));
    server_->setCallback([this](scene_flow_clusterer::ClustererConfig &c, uint32_t) {
      impl_->reconfigureCB(c);
    });
    pub_ = pnh.advertise<moving_object_msgs::MovingObjectArray>("moving_objects", 1);
    sub_ = nh.subscribe<sensor_msgs::PointCloud2>("scene_flow", 10, &ClustererNodeletRos::dataCB, this);
  }

 private:
  void dataCB(const sensor_msgs::PointCloud2ConstPtr &msg) {
    if (!camera_set_) {   // the clusterer only needs the image size; intrinsics are irrelevant for it
      ModCamera cam{}; cam.width = msg->width; cam.height = msg->height; cam.fx = cam.fy = 1.0; cam.disp_f = cam.disp_T = 1.f;
      cam.max_disparity = 1.f; mod_set_camera(ctx_, &cam); camera_set_ = true;
    }
    mod_host::PointCloud2 in;   // a production adapter passes msg->data by pointer; copied here for brevity
    in.width = msg->width; in.height = msg->height; in.point_step = msg->point_step; in.row_step = msg->row_step;
    in.data = msg->data;
    mod_host::MovingObjectArray out;
    impl_->dataCB(in, &out);
    if (pub_.getNumSubscribers() == 0) return;
    moving_object_msgs::MovingObjectArray m;
    m.header = msg->header;
    for (const auto &o : out.moving_object_array) {
      moving_object_msgs::MovingObject mo;
      mo.id = o.id;
      mo.center.position.x = o.center.position[0]; mo.center.position.y = o.center.position[1]; mo.center.position.z = o.center.position[2];
      mo.center.orientation.x = 0; mo.center.orientation.y = 0; mo.center.orientation.z = 0; mo.center.orientation.w = 1;
      mo.velocity.x = o.velocity[0]; mo.velocity.y = o.velocity[1]; mo.velocity.z = o.velocity[2];
      mo.bounding_box.x = o.bounding_box[0]; mo.bounding_box.y = o.bounding_box[1]; mo.bounding_box.z = o.bounding_box[2];
      m.moving_object_array.push_back(mo);
    }
    pub_.publish(m);
  }
  ModContext *ctx_ = nullptr;
  bool camera_set_ = false;
  std::unique_ptr<ClustererNodelet> impl_;
  std::unique_ptr<dynamic_reconfigure::Server<scene_flow_clusterer::ClustererConfig>> server_;
  ros::Publisher pub_;
  ros::Subscriber sub_;
};

}  // namespace scene_flow_clusterer

PLUGINLIB_EXPORT_CLASS(scene_flow_clusterer::ClustererNodeletRos, nodelet::Nodelet)
