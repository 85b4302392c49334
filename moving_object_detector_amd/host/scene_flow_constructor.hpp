// scene_flow_constructor.hpp — host-side mirror of scene_flow_constructor::SceneFlowConstructor
// (scene_flow_constructor/include/scene_flow_constructor.h:33-275) for the part of the class that is on the hot path:
// construct(), the previous-frame state of stereoCallback() and reconfigureCB().  Same member names, same argument
// meaning, same "publish nothing when an input is missing" behaviour — the per-pixel work goes through the C ABI
// (include/mod_sf.h) to the HIP kernels.  estimateDisparity() is here as well (the on-GPU SGM of SURVEY.md section 8(f) row 3); the
// other estimators (PWC-Net / viso2), TF and ROS wiring are out of scope.
#pragma once
#include <functional>
#include <stdexcept>

#include "messages.hpp"

namespace scene_flow_constructor {

#ifndef MOD_HOST_ROS_CONFIG   // a ROS build includes the dynamic_reconfigure-generated scene_flow_constructor/SceneFlowConstructorConfig.h first
struct SceneFlowConstructorConfig {   // cfg/SceneFlowConstructor.cfg:8-9
  int dynamic_flow_diff = 5;
  double max_color_velocity = 1.0;    // unused by live code in the reference as well
};
#endif

class SceneFlowConstructor {
 public:
  using CloudCallback = std::function<void(const mod_host::PointCloud2 &)>;

  // One context per constructor, like one node per camera.  `ctx` may be shared with a ClustererNodelet so that the
  // cloud never leaves the GPU between the two stages.
  explicit SceneFlowConstructor(ModContext *ctx) : ctx_(ctx) {}

  // stereoCallback()'s first-frame block (scene_flow_constructor.cpp:368-375): camera model from the left CameraInfo.
  void setCameraInfo(const mod_host::CameraInfo &left, const mod_host::DisparityImage &disparity) {
    ModCamera cam{};
    cam.width = left.width; cam.height = left.height;
    cam.fx = left.P[0]; cam.fy = left.P[5]; cam.cx = left.P[2]; cam.cy = left.P[6]; cam.Tx = left.P[3]; cam.Ty = left.P[7];
    cam.disp_f = disparity.f; cam.disp_T = disparity.T;
    cam.min_disparity = disparity.min_disparity; cam.max_disparity = disparity.max_disparity;
    check(mod_set_camera(ctx_, &cam));
    image_width_ = left.width; image_height_ = left.height;
  }

  // reconfigureCB (scene_flow_constructor.cpp:401-407).  The clusterer's parameters live in the same ModParams block;
  // they are preserved.
  void reconfigureCB(const SceneFlowConstructorConfig &config) {
    ModParams p = currentParams();
    p.dynamic_flow_diff = config.dynamic_flow_diff;
    check(mod_set_params(ctx_, &p));
  }

  // The clusterer's reconfigureCB (clusterer_nodelet.cpp:345-352) for a process that runs both stages on one context (the collapsed
  // node: `~publish_moving_objects`): the four Clusterer.cfg values go into the shared ModParams block, dynamic_flow_diff is preserved.
  void reconfigureClusterer(int cluster_size, double depth_diff, double dynamic_speed, int neighbor_distance) {
    ModParams p = currentParams();
    p.cluster_size = cluster_size; p.depth_diff = depth_diff; p.dynamic_speed = dynamic_speed; p.neighbor_distance = neighbor_distance;
    check(mod_set_params(ctx_, &p));
  }

  // estimateDisparity() (scene_flow_constructor.cpp:258-279): sgm_gpu_->computeDisparity(left, right, infos, disparity).  Fills
  // `disparity` as that call does for its caller — the stereo_msgs/DisparityImage contract DisparityImageProcessor reads
  // (disparity_image_processor.cpp:25-27,41-45): 32FC1 pixels with -1 = min_disparity - 1 where no match was found, f from the
  // left projection matrix, T = the baseline -P_right[3] / P_right[0], min_disparity 0, max_disparity D - 1 — into `pixels`, which
  // the message then points at.  Returns false (and leaves `disparity` alone) when an image is missing or of the wrong size: the
  // reference then resets disparity_now_, i.e. the caller passes a null disparity on.
  void setDisparityParams(const ModSgmParams &p) { sgm_ = p; }
  bool estimateDisparity(const mod_host::Image *left_image, const mod_host::Image *right_image, const mod_host::CameraInfo &left_camera_info,
                         const mod_host::CameraInfo &right_camera_info, mod_host::DisparityImage *disparity, std::vector<float> *pixels) {
    if (!left_image || !right_image || !left_image->data || !right_image->data) return false;
    if (left_image->width != left_camera_info.width || left_image->height != left_camera_info.height ||
        right_image->width != left_image->width || right_image->height != left_image->height) return false;
    // the library copies the CONTEXT's width x height bytes from each image: an image of another size than the configured camera
    // would be over-read (setCameraInfo() ran on the first frame, scene_flow_constructor.cpp:368-375)
    if (image_width_ > 0 && (left_image->width != image_width_ || left_image->height != image_height_)) return false;
    pixels->resize((size_t)left_image->width * left_image->height);
    const int rc = mod_sgm_compute_host(ctx_, left_image->data, right_image->data, &sgm_, pixels->data());
    if (rc > 0) return false;
    check(rc);
    disparity->header = left_image->header;
    disparity->width = left_image->width; disparity->height = left_image->height;
    disparity->data = pixels->data();
    disparity->f = (float)left_camera_info.P[0];
    disparity->T = (float)(-right_camera_info.P[3] / right_camera_info.P[0]);
    disparity->min_disparity = 0.0f;
    disparity->max_disparity = (float)(sgm_.disparities - 1);
    return true;
  }

  // construct() (scene_flow_constructor.cpp:91-147).  Null pointers play the role of the reference's empty shared_ptrs.
  // Returns true when a cloud was produced ("published"); false when the reference would have published nothing.
  bool construct(const mod_host::DisparityImage *disparity_now, const mod_host::DisparityImage *disparity_previous,
                 const mod_host::FlowImage *left_flow, const mod_host::Transform *transform_prev2now,
                 mod_host::PointCloud2 *pc_with_velocity, std::vector<int32_t> *labels = nullptr,
                 mod_host::MovingObjectArray *moving_objects = nullptr, std::vector<float> *depth_image = nullptr) {
    // ~depth goes out whenever disparity_now exists, before the guards that end the frame (:110-123)
    if (depth_image) {
      depth_image->clear();
      if (disparity_now) {
        depth_image->resize((size_t)image_width_ * image_height_);
        const int drc = mod_depth_image_host(ctx_, disparity_now->data, depth_image->data());
        check(drc);
        if (drc > 0) depth_image->clear();   // a message without pixels: the reference publishes no depth image either
      }
    }
    ModTransform tf{};
    if (transform_prev2now) {
      for (int i = 0; i < 3; i++) tf.t[i] = transform_prev2now->translation[i];
      for (int i = 0; i < 4; i++) tf.q[i] = transform_prev2now->rotation[i];
    }
    // time_between_frames = stamp_now - stamp_previous (scene_flow_constructor.cpp:162-164)
    const double dt = (disparity_now && disparity_previous) ? mod_host::duration_sec(disparity_now->header.stamp, disparity_previous->header.stamp) : 0.0;
    const size_t n = (size_t)image_width_ * image_height_;
    if (pc_with_velocity) pc_with_velocity->data.resize(n * 32);
    if (labels) labels->resize(n);
    std::vector<ModObject> objs(max_objects_);
    int32_t n_obj = 0;
    const int rc = mod_process_frame_host(ctx_, disparity_now ? disparity_now->data : nullptr,
                                          disparity_previous ? disparity_previous->data : nullptr,
                                          left_flow ? left_flow->data : nullptr, transform_prev2now ? &tf : nullptr, dt,
                                          pc_with_velocity ? pc_with_velocity->data.data() : nullptr,
                                          labels ? labels->data() : nullptr, objs.data(), (int32_t)objs.size(), &n_obj);
    if (rc > 0) return false;          // an input is missing: nothing is published (:104,110,122,127,133)
    check(rc);
    if (pc_with_velocity) {            // publishPointcloud (:351-362): organized cloud, header of the flow image
      pc_with_velocity->header = left_flow->header;
      pc_with_velocity->width = image_width_; pc_with_velocity->height = image_height_;
      pc_with_velocity->point_step = 32; pc_with_velocity->row_step = 32 * image_width_;
      pc_with_velocity->is_dense = true;   // PCL's (width, height, value) constructor leaves is_dense = true
    }
    if (moving_objects) {
      moving_objects->header = left_flow->header;
      moving_objects->moving_object_array.clear();
      for (int i = 0; i < n_obj && i < (int)objs.size(); i++) moving_objects->moving_object_array.push_back(mod_host::to_message(objs[i]));
    }
    return true;
  }

  // The part of stereoCallback() that carries state across frames (:389-398): the caller hands in this frame's estimator
  // outputs; the previous disparity is remembered here.  Frame 0 has no previous disparity / flow -> returns false.
  bool stereoCallback(const mod_host::DisparityImage *disparity_now, const mod_host::FlowImage *left_flow,
                      const mod_host::Transform *transform_prev2now, mod_host::PointCloud2 *pc_with_velocity,
                      mod_host::MovingObjectArray *moving_objects = nullptr) {
    const mod_host::DisparityImage *prev = have_previous_ ? &disparity_previous_ : nullptr;
    const bool ok = construct(disparity_now, prev, left_flow, transform_prev2now, pc_with_velocity, nullptr, moving_objects);
    if (disparity_now) {               // disparity_previous_ = disparity_now_ (:398); the pixels are copied: the message dies
      previous_pixels_.assign(disparity_now->data, disparity_now->data + (size_t)disparity_now->width * disparity_now->height);
      disparity_previous_ = *disparity_now;
      disparity_previous_.data = previous_pixels_.data();
      have_previous_ = true;
    } else {
      have_previous_ = false;          // estimateDisparity failed: disparity_now_.reset() (:272-276)
    }
    return ok;
  }

  // Pipelined stereoCallback(): the reference runs construct() of frame t on construct_thread_ while the estimators of
  // frame t+1 run (:389-392).  submit() enqueues the frame (copies and kernels proceed on the GPU's streams) and returns a
  // ticket, or -1 where nothing will be published; collect() joins it and fills the messages handed to submit().  The
  // previous disparity stays resident in HBM (disparity_previous_ = disparity_now_, :397-398) instead of being copied on
  // the host.  Up to MOD_PIPELINE_DEPTH frames may be in flight; tickets are collected in order.  The message buffers
  // (disparity, flow, cloud) should be page-locked (mod_host_malloc) for the copies to overlap.
  int submit(const mod_host::DisparityImage *disparity_now, const mod_host::FlowImage *left_flow,
             const mod_host::Transform *transform_prev2now, mod_host::PointCloud2 *pc_with_velocity,
             mod_host::MovingObjectArray *moving_objects = nullptr) {
    ModTransform tf{};
    if (transform_prev2now) {
      for (int i = 0; i < 3; i++) tf.t[i] = transform_prev2now->translation[i];
      for (int i = 0; i < 4; i++) tf.q[i] = transform_prev2now->rotation[i];
    }
    const double dt = (disparity_now && have_stamp_) ? mod_host::duration_sec(disparity_now->header.stamp, previous_stamp_) : 0.0;
    const size_t n = (size_t)image_width_ * image_height_;
    int32_t ticket = -1;
    int rc = MOD_SKIP_NO_DISPARITY_NOW;
    if (disparity_now) {
      if (pc_with_velocity && pc_with_velocity->data.size() != n * 32) pc_with_velocity->data.resize(n * 32);
      Pending &p = pending_[next_slot_];
      p.objects.resize(max_objects_);
      // previous disparity: resident in HBM from the last enqueued frame, or the host copy parked by a skipped frame
      // moving_objects == nullptr: no cluster output is asked for and the library runs the scene-flow stage alone
      rc = mod_submit_frame_host(ctx_, disparity_now->data, have_parked_ ? parked_.data() : nullptr,
                                 left_flow ? left_flow->data : nullptr, transform_prev2now ? &tf : nullptr, dt,
                                 pc_with_velocity ? pc_with_velocity->data.data() : nullptr, nullptr, moving_objects ? p.objects.data() : nullptr,
                                 moving_objects ? (int32_t)p.objects.size() : 0, &ticket);
      if (rc == MOD_OK) {
        p.ticket = ticket; p.cloud = pc_with_velocity; p.objs = moving_objects; p.header = left_flow->header;
        next_slot_ = (next_slot_ + 1) % MOD_PIPELINE_DEPTH;
        have_parked_ = false;
      } else if (rc > 0) {
        // nothing was enqueued, so this disparity never reached HBM — yet it IS the next frame's previous one
        // (disparity_previous_ = disparity_now_ runs whatever construct() did, :398): park a host copy
        check(mod_forget_previous(ctx_));
        parked_.assign(disparity_now->data, disparity_now->data + n);
        have_parked_ = true;
      }
      previous_stamp_ = disparity_now->header.stamp; have_stamp_ = true;
    } else {
      check(mod_forget_previous(ctx_));   // estimateDisparity failed: disparity_now_.reset() (:272-276)
      have_stamp_ = false; have_parked_ = false;
    }
    if (rc > 0) return -1;
    check(rc);
    return ticket;
  }
  // Pipelined stereoCallback() with the estimator on the GPU (BASELINE config 5): estimateDisparity() + construct() of one frame
  // in a single submission — images go in, the disparity plane is produced in HBM, serves as `now` here and as `previous` of the
  // next frame (disparity_previous_ = disparity_now_, :397-398) and never crosses the host link (estimateDisparity() + submit()
  // carry it to the host and back).  Returns a ticket for collect(), or -1 where nothing will be published; a missing or
  // mis-sized image plays the part of a failed estimateDisparity() (:272-276).
  int submitStereo(const mod_host::Image *left_image, const mod_host::Image *right_image, const mod_host::FlowImage *left_flow,
                   const mod_host::Transform *transform_prev2now, mod_host::PointCloud2 *pc_with_velocity,
                   mod_host::MovingObjectArray *moving_objects = nullptr) {
    const bool images = left_image && right_image && left_image->data && right_image->data && left_image->width == image_width_ &&
                        left_image->height == image_height_ && right_image->width == image_width_ && right_image->height == image_height_;
    ModTransform tf{};
    if (transform_prev2now) {
      for (int i = 0; i < 3; i++) tf.t[i] = transform_prev2now->translation[i];
      for (int i = 0; i < 4; i++) tf.q[i] = transform_prev2now->rotation[i];
    }
    const double dt = (images && have_stamp_) ? mod_host::duration_sec(left_image->header.stamp, previous_stamp_) : 0.0;
    const size_t n = (size_t)image_width_ * image_height_;
    if (images && pc_with_velocity && pc_with_velocity->data.size() != n * 32) pc_with_velocity->data.resize(n * 32);
    Pending &p = pending_[next_slot_];
    p.objects.resize(max_objects_);
    int32_t ticket = -1;
    // pc_with_velocity == nullptr: the 32-byte cloud is neither packed nor copied to the host (nobody subscribes to ~scene_flow);
    // moving_objects == nullptr: no cluster output is asked for and the library runs the scene-flow stage alone
    const int rc = mod_submit_stereo_host(ctx_, images ? left_image->data : nullptr, images ? right_image->data : nullptr, &sgm_,
                                          left_flow ? left_flow->data : nullptr, transform_prev2now ? &tf : nullptr, dt,
                                          pc_with_velocity ? pc_with_velocity->data.data() : nullptr, nullptr, moving_objects ? p.objects.data() : nullptr,
                                          moving_objects ? (int32_t)p.objects.size() : 0, nullptr, &ticket);
    have_parked_ = false;                         // the previous disparity lives in HBM (or is gone with the images)
    if (images) { previous_stamp_ = left_image->header.stamp; have_stamp_ = true; }
    else have_stamp_ = false;
    if (rc > 0) return -1;
    check(rc);
    p.ticket = ticket; p.cloud = pc_with_velocity; p.objs = moving_objects; p.header = left_flow->header;
    next_slot_ = (next_slot_ + 1) % MOD_PIPELINE_DEPTH;
    return ticket;
  }
  void collect(int ticket) {
    for (Pending &p : pending_) {
      if (p.ticket != ticket) continue;
      int32_t n_obj = 0;
      check(mod_collect_frame_host(ctx_, ticket, &n_obj));
      if (p.cloud) {
        p.cloud->header = p.header;
        p.cloud->width = image_width_; p.cloud->height = image_height_;
        p.cloud->point_step = 32; p.cloud->row_step = 32 * image_width_;
        p.cloud->is_dense = true;
      }
      if (p.objs) {
        p.objs->header = p.header;
        p.objs->moving_object_array.clear();
        for (int i = 0; i < n_obj && i < (int)p.objects.size(); i++) p.objs->moving_object_array.push_back(mod_host::to_message(p.objects[i]));
      }
      p.ticket = -1;
      return;
    }
    throw std::runtime_error("collect(): unknown ticket");
  }

  void setMaxObjects(int n) { max_objects_ = n; }

 private:
  struct Pending {
    int ticket = -1;
    mod_host::PointCloud2 *cloud = nullptr;
    mod_host::MovingObjectArray *objs = nullptr;
    mod_host::Header header;
    std::vector<ModObject> objects;
  };
  Pending pending_[MOD_PIPELINE_DEPTH];
  int next_slot_ = 0;
  bool have_stamp_ = false, have_parked_ = false;
  mod_host::Time previous_stamp_;
  std::vector<float> parked_;

  ModParams currentParams() {
    ModParams p{};
    if (mod_get_params(ctx_, &p) != MOD_OK) {   // reference defaults (SceneFlowConstructor.cfg:8, Clusterer.cfg:8-11)
      p.dynamic_flow_diff = 5; p.cluster_size = 2500; p.neighbor_distance = 4; p.depth_diff = 0.15; p.dynamic_speed = 0.3;
    }
    return p;
  }
  void check(int rc) { if (rc < 0) throw std::runtime_error(std::string("libmod_sf: ") + mod_last_error(ctx_)); }

  ModContext *ctx_;
  ModSgmParams sgm_{128, 6, 96, 8, 1, 1};   // the published configuration of the estimator (oracle/sgm_ref.cpp)
  int image_width_ = 0, image_height_ = 0;
  int max_objects_ = 1024;
  bool have_previous_ = false;
  mod_host::DisparityImage disparity_previous_;
  std::vector<float> previous_pixels_;
};

}  // namespace scene_flow_constructor
