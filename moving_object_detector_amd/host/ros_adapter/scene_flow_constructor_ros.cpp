// scene_flow_constructor_ros.cpp — ROS-1 node shell around scene_flow_constructor::SceneFlowConstructor (host mirror).
// Not built in this image (no ROS / pwc_net / viso2); tests/test_ros_adapter_syntax.py keeps it compiling against declaration-only
// stand-ins of the ROS types.  Shows where the three estimators hand their
// outputs to construct() (scene_flow_constructor.cpp:378-392) and how the results map back onto the reference's topics.
#include <geometry_msgs/Transform.h>
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/PointCloud2.h>
#include <stereo_msgs/DisparityImage.h>

#include "../scene_flow_constructor.hpp"

// Called from the reference's stereoCallback() in place of `construct_thread_ = std::thread(&construct, ...)`:
//   disparity : result of sgm_gpu_->computeDisparity          (scene_flow_constructor.cpp:267)
//   flow      : result of pwc_net_.estimateOpticalFlow, 32FC2  (:282)
//   motion    : tf2::toMsg(tf2_camera_motion) from libviso2    (:235-249); null when visual odometry failed (:251-255)
void publishSceneFlow(scene_flow_constructor::SceneFlowConstructor &impl, ros::Publisher &pc_with_velocity_pub,
                      const stereo_msgs::DisparityImage *disparity, const sensor_msgs::Image *flow,
                      const geometry_msgs::Transform *motion) {
  mod_host::DisparityImage d;
  mod_host::FlowImage f;
  mod_host::Transform t;
  if (disparity) {
    d.header.stamp = mod_host::Time(disparity->header.stamp.sec, disparity->header.stamp.nsec); d.width = disparity->image.width; d.height = disparity->image.height;
    d.data = reinterpret_cast<const float *>(disparity->image.data.data());
    d.f = disparity->f; d.T = disparity->T; d.min_disparity = disparity->min_disparity; d.max_disparity = disparity->max_disparity;
  }
  if (flow) { f.header.stamp = mod_host::Time(flow->header.stamp.sec, flow->header.stamp.nsec); f.width = flow->width; f.height = flow->height;
              f.data = reinterpret_cast<const float *>(flow->data.data()); }
  if (motion) { t.translation[0] = motion->translation.x; t.translation[1] = motion->translation.y; t.translation[2] = motion->translation.z;
                t.rotation[0] = motion->rotation.x; t.rotation[1] = motion->rotation.y; t.rotation[2] = motion->rotation.z; t.rotation[3] = motion->rotation.w; }
  mod_host::PointCloud2 cloud;
  if (!impl.stereoCallback(disparity ? &d : nullptr, flow ? &f : nullptr, motion ? &t : nullptr, &cloud)) return;
  if (pc_with_velocity_pub.getNumSubscribers() == 0) return;
  sensor_msgs::PointCloud2 msg;   // fields x,y,z,vx,vy,vz float32 at 0,4,8,16,20,24 (pcl_point_xyz_velocity.h:27-34)
  msg.header = flow->header; msg.width = cloud.width; msg.height = cloud.height; msg.point_step = 32; msg.row_step = cloud.row_step;
  msg.is_dense = true; msg.data = std::move(cloud.data);
  const char *names[6] = {"x", "y", "z", "vx", "vy", "vz"};
  const uint32_t offs[6] = {0, 4, 8, 16, 20, 24};
  for (int i = 0; i < 6; i++) { sensor_msgs::PointField pf; pf.name = names[i]; pf.offset = offs[i]; pf.datatype = sensor_msgs::PointField::FLOAT32; pf.count = 1; msg.fields.push_back(pf); }
  pc_with_velocity_pub.publish(msg);
}
