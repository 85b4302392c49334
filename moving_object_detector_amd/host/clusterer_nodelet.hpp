// clusterer_nodelet.hpp — host-side mirror of scene_flow_clusterer::ClustererNodelet
// (scene_flow_clusterer/include/clusterer_nodelet.h:37-108): dataCB() on a PointCloud2 of pcl::PointXYZVelocity,
// reconfigureCB(), and the MovingObjectArray it publishes.  clustering() and cluster2MovingObject() run on the GPU
// through the C ABI (include/mod_sf.h).  Markers / colour image (debug visualisation) are out of scope; the labels
// plane returned here is what they are drawn from.
#pragma once
#include <stdexcept>

#include "messages.hpp"

namespace scene_flow_clusterer {

#ifndef MOD_HOST_ROS_CONFIG   // a ROS build includes the dynamic_reconfigure-generated scene_flow_clusterer/ClustererConfig.h first
struct ClustererConfig {   // cfg/Clusterer.cfg:8-11
  int cluster_size = 2500;
  double depth_diff = 0.15;
  double dynamic_speed = 0.3;
  int neighbor_distance = 4;
};
#endif

class ClustererNodelet {
 public:
  explicit ClustererNodelet(ModContext *ctx) : ctx_(ctx) {}

  // reconfigureCB (clusterer_nodelet.cpp:345-352); the constructor's dynamic_flow_diff in the shared block is preserved.
  void reconfigureCB(const ClustererConfig &config) {
    ModParams p{};
    if (mod_get_params(ctx_, &p) != MOD_OK) p.dynamic_flow_diff = 5;
    p.cluster_size = config.cluster_size; p.depth_diff = config.depth_diff;
    p.dynamic_speed = config.dynamic_speed; p.neighbor_distance = config.neighbor_distance;
    check(mod_set_params(ctx_, &p));
  }

  // dataCB (clusterer_nodelet.cpp:221-242): clustering() + publishMovingObjects().  `cluster_map` receives the final
  // cluster_map_ (-1 = NOT_BELONGED_).  An unorganized / wrongly sized cloud is an error, as .at() would throw there.
  void dataCB(const mod_host::PointCloud2 &input_pc_msg, mod_host::MovingObjectArray *moving_objects,
              std::vector<int32_t> *cluster_map = nullptr) {
    const size_t n = (size_t)input_pc_msg.width * input_pc_msg.height;
    if (cluster_map) cluster_map->resize(n);
    std::vector<ModObject> objs(max_objects_);
    int32_t n_obj = 0;
    check(mod_cluster_cloud_host(ctx_, input_pc_msg.data.data(), (int32_t)input_pc_msg.width, (int32_t)input_pc_msg.height,
                                 (int32_t)input_pc_msg.point_step, (int32_t)input_pc_msg.row_step,
                                 cluster_map ? cluster_map->data() : nullptr, objs.data(), (int32_t)objs.size(), &n_obj));
    if (moving_objects) {              // publishMovingObjects (:324-343): header = input header, ids over accepted clusters
      moving_objects->header = input_pc_msg.header;
      moving_objects->moving_object_array.clear();
      for (int i = 0; i < n_obj && i < (int)objs.size(); i++) moving_objects->moving_object_array.push_back(mod_host::to_message(objs[i]));
    }
  }

  void setMaxObjects(int n) { max_objects_ = n; }

 private:
  void check(int rc) { if (rc < 0) throw std::runtime_error(std::string("libmod_sf: ") + mod_last_error(ctx_)); }
  ModContext *ctx_;
  int max_objects_ = 1024;
};

}  // namespace scene_flow_clusterer
