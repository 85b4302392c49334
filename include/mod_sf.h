/*
 * mod_sf.h — C ABI of the MI355X-native scene-flow + moving-point clustering path.
 *
 * The reference (ActiveIntelligentSystemsLab/moving_object_detector) has no FFI: the path is a ROS-1 node
 * (scene_flow_constructor) plus a nodelet plugin (scene_flow_clusterer/scene_flow_clusterer,
 * scene_flow_clusterer/nodelet_plugins.xml:3-4) joined by the ~scene_flow PointCloud2 topic.  This header is the
 * boundary a maintainer binds instead of the CPU loops; every entry point cites the reference code it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the node/nodelet side of the binding.
 *
 * Conventions
 *   - plain C, POD structs, no exceptions cross the boundary;
 *   - return value: 0 = ok, >0 = "skipped, input missing" (the reference silently publishes nothing,
 *     scene_flow_constructor/src/scene_flow_constructor.cpp:104,110,122,127,133), <0 = error
 *     (mod_last_error() gives the text);
 *   - a context is single-caller (the reference runs construct() on one thread at a time, :389-392, and the
 *     clusterer on a single-threaded callback queue, clusterer_nodelet.cpp:25,35); distinct contexts are independent;
 *   - "_dev" entry points take DEVICE pointers (HBM-resident planes) and only enqueue work on the context's
 *     stream; "_host" entry points take host pointers, stage through the context and synchronise;
 *   - image planes are row-major, pitch == width, frames contiguous ([frames][H][W]).
 */
#ifndef MOD_SF_H_
#define MOD_SF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libmod_sf.so is built with -fvisibility=hidden: the declarations of this header are its whole dynamic symbol table
 * (tests/test_abi.py checks `nm -D --defined-only`). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define MOD_ABI_VERSION 2   /* 2: ModConfig.batch_chunks (was reserved), stage "select" folded into MOD_STAGE_CCL_MERGE, MOD_STAGE_CLUSTER_GROUP */

/* status codes */
#define MOD_OK                      0
#define MOD_SKIP_NO_DISPARITY_NOW   1  /* construct(): `if (disparity_now)` guard, scene_flow_constructor.cpp:110 */
#define MOD_SKIP_NO_DISPARITY_PREV  2  /* :104,127 */
#define MOD_SKIP_NO_FLOW            3  /* :122-123 */
#define MOD_SKIP_NO_TRANSFORM       4  /* :127 (visual odometry failed, :251-255) */
#define MOD_ERR_INVALID_ARGUMENT   -1
#define MOD_ERR_NOT_CONFIGURED     -2  /* camera/params not set */
#define MOD_ERR_CAPACITY           -3  /* frames / image larger than the context was created for */
#define MOD_ERR_DEVICE             -4  /* HIP runtime error */
#define MOD_ERR_NO_DEVICE          -5  /* no gfx950 device / library built without a usable device */

typedef struct ModContext ModContext;

#define MOD_MAX_WIDTH 16384   /* widest image a context accepts (one image row of census words must fit LDS; tested up to 8256) */

/* Context creation parameters. */
typedef struct ModConfig {
  int32_t device;       /* HIP device ordinal */
  int32_t max_width;    /* largest image width  the scratch is sized for, <= MOD_MAX_WIDTH */
  int32_t max_height;   /* largest image height the scratch is sized for */
  int32_t max_frames;   /* largest batch (frames per call), <= 65535; max_width * max_height < 2^27 */
  int32_t max_objects;  /* per-frame capacity of the ModObject output; 0 -> max_width*max_height/100 (Clusterer.cfg:8 lower bound).
                           mod_set_params rejects a cluster_size with max_width*max_height/cluster_size > max_objects, so no
                           cluster can ever be dropped */
  int32_t batch_chunks; /* mod_process_dev on a large batch: 2..4 -> the cluster stage runs in that many chunks of frames, side by side
                           on streams of the context's own, so that its waiting kernels (cross-tile links, root merge, median
                           selection, tie replay) share the GPU with the streaming kernels of another chunk; forked from and joined
                           to `stream` with events — the call is ordered on `stream` like any other; a chunk has at least 32 frames.
                           0 or 1 -> one piece (the default).  Results do not depend on it.  Worth 1 - 3 % of a 512-pair step in a
                           long-running process and nothing in a process's first calls (csrc/mod_sf.hip process_chunked). */
  void   *stream;       /* hipStream_t to enqueue on; NULL -> the context creates its own */
} ModConfig;

/*
 * Camera + disparity-message constants.
 *   fx..Ty : left CameraInfo projection matrix P (P[0],P[5],P[2],P[6],P[3],P[7]) as used by
 *            image_geometry::PinholeCameraModel::projectPixelTo3dRay / project3dToPixel
 *            (call sites disparity_image_processor.cpp:45, scene_flow_constructor.cpp:84);
 *   disp_* : stereo_msgs/DisparityImage f, T, min_disparity, max_disparity
 *            (disparity_image_processor.cpp:25-27,41-42).
 */
typedef struct ModCamera {
  int32_t width, height;
  double  fx, fy, cx, cy, Tx, Ty;
  float   disp_f, disp_T, min_disparity, max_disparity;
} ModCamera;

/*
 * dynamic_reconfigure parameters, read by value inside the hot loops
 * (scene_flow_constructor/cfg/SceneFlowConstructor.cfg:8, scene_flow_clusterer/cfg/Clusterer.cfg:8-11).
 * cluster_size must be >= 1 (the reference's range is [100,10000]).
 */
typedef struct ModParams {
  int32_t dynamic_flow_diff;   /* default 5   [px]  */
  int32_t cluster_size;        /* default 2500 [px] */
  int32_t neighbor_distance;   /* default 4   [px], 1..MOD_MAX_NEIGHBOR_DISTANCE */
  int32_t reserved;
  double  depth_diff;          /* default 0.15 [m]  */
  double  dynamic_speed;       /* default 0.3 [m/s] */
} ModParams;
#define MOD_MAX_NEIGHBOR_DISTANCE 16

/* geometry_msgs/Transform previous->now (scene_flow_constructor.cpp:248-249,411): translation + quaternion x,y,z,w. */
typedef struct ModTransform {
  double t[3];
  double q[4];
} ModTransform;

/* moving_object_msgs/MovingObject (moving_object_msgs/msg/MovingObject.msg:3-7); orientation is always (0,0,0,1). */
typedef struct ModObject {
  int32_t id;
  int32_t n_points;       /* cluster size (not in the message; handy for tests/tracing) */
  double  center[3];
  double  orientation[4];
  double  velocity[3];
  double  bounding_box[3];
} ModObject;

/* The four values handed to SceneFlowConstructor::construct (scene_flow_constructor.cpp:91-97,392), batched. */
typedef struct ModFrameBatch {
  int32_t      frames;
  int32_t      reserved;
  const float *disparity_now;    /* dev [frames][H][W] 32FC1; NULL -> MOD_SKIP_NO_DISPARITY_NOW  */
  const float *disparity_prev;   /* dev [frames][H][W] 32FC1; NULL -> MOD_SKIP_NO_DISPARITY_PREV; for a sequence D[0..F]
                                    pass prev = D, now = D + H*W */
  const float *flow;             /* dev [frames][H][W][2] 32FC2 (x then y, scene_flow_constructor/README.md:35-40); NULL -> skip */
  const ModTransform *transforms;/* HOST [frames]; NULL -> MOD_SKIP_NO_TRANSFORM */
  const double *dt;              /* HOST [frames]: stamp_now - stamp_previous in seconds (scene_flow_constructor.cpp:162-164) */
} ModFrameBatch;

/* ~scene_flow as SoA planes (+ optional reference-layout outputs). All device pointers, [frames][H][W]. */
typedef struct ModSceneFlowPlanes {
  float    *x, *y, *z, *vx, *vy, *vz;  /* required — except x and y in mod_process_dev (both or neither): the reference builds and ships
                                          the cloud only for subscribers (scene_flow_constructor.cpp:141-142); a caller that serves the
                                          moving objects alone passes x = y = NULL, the scene-flow kernel then writes 16 instead of 24
                                          bytes per pixel and the cluster stage recomputes its members' x, y from z (bit-identical) */
  uint64_t *dynamic_mask;  /* optional [frames][H][mod_mask_words(W)]: bit b of word k of a row = pixel 64k+b is dynamic
                              (calculateDynamicMap, clusterer_nodelet.cpp:40-54) */
  void     *cloud_aos;     /* optional [frames][H][W] 32-byte pcl::PointXYZVelocity records
                              (scene_flow_constructor/pcl_point_xyz_velocity.h:8-34: x@0 y@4 z@8 vx@16 vy@20 vz@24) */
  float    *depth;         /* optional ~depth image (disparity_image_processor.cpp:105-120) */
  float    *static_flow;   /* optional ~synthetic_optical_flow 32FC2 (scene_flow_constructor.cpp:65-89) */
} ModSceneFlowPlanes;

/* Output of the clusterer (clusterer_nodelet.cpp:85-95,324-343). Device pointers. */
typedef struct ModClusterOut {
  int32_t   *labels;      /* optional [frames][H][W]: -1 = none, 0..K-1 in the reference's order (removeSmallClusters, :354-393).
                             The reference renders its cluster image only while somebody subscribes to it
                             (publishClustersImage, :235-236,292-322): pass NULL and the 4 B/px plane is never written. */
  ModObject *objects;     /* [frames][max_objects] */
  int32_t   *n_objects;   /* [frames] accepted objects (publishMovingObjects, :324-343) */
  int32_t   *n_clusters;  /* [frames] K = clusters surviving the size filter (may be NULL) */
} ModClusterOut;

static inline int32_t mod_mask_words(int32_t width) { return (width + 63) / 64; }

/* ---- context ---------------------------------------------------------------------------------------------- */
int  mod_abi_version(void);
int  mod_create(const ModConfig *cfg, ModContext **out_ctx);
void mod_destroy(ModContext *ctx);
const char *mod_last_error(const ModContext *ctx);          /* never NULL */
int  mod_set_camera(ModContext *ctx, const ModCamera *cam); /* stereoCallback first-frame block, scene_flow_constructor.cpp:368-375 */
int  mod_set_params(ModContext *ctx, const ModParams *prm); /* reconfigureCB, scene_flow_constructor.cpp:401-407, clusterer_nodelet.cpp:345-352 */
int  mod_get_camera(const ModContext *ctx, ModCamera *cam);
int  mod_get_params(const ModContext *ctx, ModParams *prm);
int  mod_synchronize(ModContext *ctx);                      /* wait for everything enqueued on the context's stream */

/* ---- hot path, device-resident ---------------------------------------------------------------------------- */
/* construct(): toPointCloud x2 + transformPCPreviousToNow + calculateStaticOpticalFlow + constructVelocityPC
 * (scene_flow_constructor.cpp:91-147) as one fused kernel; also emits the dynamic mask when requested. */
int  mod_scene_flow_dev(ModContext *ctx, const ModFrameBatch *in, const ModSceneFlowPlanes *out);

/* ~depth alone: toDepthImage (disparity_image_proc/src/disparity_image_processor.cpp:105-120).  construct() publishes it whenever
 * disparity_now exists — also on frames that end at one of its guards without scene flow (scene_flow_constructor.cpp:110-123).
 * mod_scene_flow_dev / mod_process_dev follow that: when they return a skip code other than MOD_SKIP_NO_DISPARITY_NOW and
 * out->depth was requested, the depth plane HAS been enqueued (all other outputs are untouched).  These two entry points
 * give the depth image without a batch (device planes [frames][H][W]; host: one frame).  NULL disparity -> MOD_SKIP_NO_DISPARITY_NOW. */
int  mod_depth_image_dev(ModContext *ctx, int32_t frames, const float *disparity_now, float *depth);
int  mod_depth_image_host(ModContext *ctx, const float *disparity_now, float *depth);

/* ~synthetic_optical_flow alone: calculateStaticOpticalFlow (scene_flow_constructor.cpp:65-89) — the flow a static scene would
 * show under the camera motion `transform`: previous cloud reprojected, moved, projected (32FC2, NaN where the previous point is
 * invalid).  construct() publishes it only for subscribers (:141-145); the batch entry points give it through
 * ModSceneFlowPlanes.static_flow, this one for a node that holds host buffers.  One frame, synchronous.
 * NULL disparity_prev -> MOD_SKIP_NO_DISPARITY_PREV, NULL transform -> MOD_SKIP_NO_TRANSFORM. */
int  mod_static_flow_host(ModContext *ctx, const float *disparity_prev, const ModTransform *transform, float *static_flow);

/* calculateDynamicMap (clusterer_nodelet.cpp:40-54) for a cloud that did not come from mod_scene_flow_dev. */
int  mod_dynamic_mask_dev(ModContext *ctx, int32_t frames, const float *vx, const float *vy, const float *vz,
                          uint64_t *dynamic_mask);

/* clustering() + publishMovingObjects() (clusterer_nodelet.cpp:85-95,324-343) on SoA planes.
 * planes->dynamic_mask may be NULL (computed internally). */
int  mod_cluster_dev(ModContext *ctx, int32_t frames, const ModSceneFlowPlanes *planes, const ModClusterOut *out);

/* Both stages back to back without the PointCloud2 round trip (the mask is produced by the scene-flow kernel).
 * planes->x and planes->y may both be NULL (objects-only: see ModSceneFlowPlanes). */
int  mod_process_dev(ModContext *ctx, const ModFrameBatch *in, const ModSceneFlowPlanes *planes, const ModClusterOut *out);

/* pcl::toROSMsg / pcl::fromROSMsg payload conversion (scene_flow_constructor.cpp:358-361, clusterer_nodelet.cpp:226). */
int  mod_pack_cloud_dev(ModContext *ctx, int32_t frames, const ModSceneFlowPlanes *planes, void *cloud_aos);
int  mod_unpack_cloud_dev(ModContext *ctx, int32_t frames, const void *cloud_aos, const ModSceneFlowPlanes *planes);

/* ---- on-GPU disparity (BASELINE config 5) ------------------------------------------------------------ */
/* The reference obtains disparity_now from sgm_gpu::SgmGpu::computeDisparity(left, right, left_info, right_info, disparity)
 * (scene_flow_constructor/src/scene_flow_constructor.cpp:35,267; package sgm_gpu of sgm_gpu_ros, not vendored).  Algorithm,
 * parameters and every choice the publication leaves open: oracle/sgm_ref.cpp (semi-global matching: centre-symmetric 9 x 7
 * census, Hamming cost, 8 paths with P1 / P2, winner-take-all, 3 x 3 median, left-right check).  Images are 8-bit, rectified,
 * [frames][H][W] of the configured camera size; the result is the `image` of a stereo_msgs/DisparityImage (32FC1, invalid
 * pixels -1 = min_disparity - 1) whose other fields are f, T of the camera, min_disparity 0, max_disparity disparities - 1 —
 * exactly what mod_set_camera takes as disp_f, disp_T, min_disparity, max_disparity
 * (disparity_image_proc/src/disparity_image_processor.cpp:25-27,41-42). */
#define MOD_SGM_MAX_DISPARITIES 128
typedef struct ModSgmParams {
  int32_t disparities;   /* D <= 128; default 128 */
  int32_t p1, p2;        /* smoothness penalties; defaults 6, 96; 31 + p2 must fit uint8 */
  int32_t paths;         /* 8, or 4 (the horizontal and vertical ones) */
  int32_t lr_check;      /* left-right consistency check, tolerance 1 */
  int32_t median;        /* 3 x 3 median of the winner-take-all maps */
} ModSgmParams;
/* computeDisparity: device images in, device disparity plane out.  Ordered like any other work on the context's stream (the
 * aggregation paths run on streams of the context's own, forked from and joined to it with events).  Scratch — two sets of, per
 * frame of a group of up to 8 frames, one W*H*disparities uint8 cost volume per path, at most 24 GB — is allocated on first use.
 * Images at least 2 pixels wide.  NULL image -> MOD_SKIP_NO_DISPARITY_NOW, as a failed estimateDisparity. */
int  mod_sgm_compute_dev(ModContext *ctx, int32_t frames, const uint8_t *left, const uint8_t *right, const ModSgmParams *params,
                         float *disparity);
/* the same for one frame in host memory (what a ROS node holding sensor_msgs/Image buffers calls); synchronous */
int  mod_sgm_compute_host(ModContext *ctx, const uint8_t *left, const uint8_t *right, const ModSgmParams *params, float *disparity);
/* stages, for tests and tracing: centre-symmetric census (31 bits per pixel, 0 where the window leaves the image) ... */
int  mod_sgm_census_dev(ModContext *ctx, int32_t frames, const uint8_t *image, uint32_t *census);
/* ... and one aggregation path L_r [frames][H][W][disparities] uint8 over the Hamming cost of the census words; direction 0..7 =
 * (+1,0) (-1,0) (0,+1) (0,-1) (+1,+1) (-1,-1) (-1,+1) (+1,-1).  matching_cost (optional) receives C itself. */
int  mod_sgm_path_dev(ModContext *ctx, int32_t frames, const uint32_t *census_left, const uint32_t *census_right,
                      const ModSgmParams *params, int32_t direction, uint8_t *path_cost, uint8_t *matching_cost);

/* ---- host-pointer convenience (what a ROS node with host-side messages calls) ----------------------------- */
/* One frame, host buffers in/out; any output pointer may be NULL.  Returns a skip code exactly where construct()
 * would publish nothing.  cloud_aos: W*H*32 bytes; labels: W*H int32; objects: capacity `max_objects`.
 * With labels == NULL, objects == NULL AND n_objects == NULL the clustering stage does not run at all (the reference's constructor
 * node does not cluster; its clusterer runs whenever a cloud arrives, clusterer_nodelet.cpp:231): a node that serves ~scene_flow
 * alone pays for the scene-flow stage only.  A caller that passes n_objects alone still gets the real count.  mod_submit_frame_host
 * and mod_submit_stereo_host have no count pointer at submit time: there labels == NULL and objects == NULL select the scene-flow
 * stage alone, and mod_collect_frame_host reports 0 objects for such a ticket. */
int  mod_process_frame_host(ModContext *ctx,
                            const float *disparity_now, const float *disparity_prev, const float *flow,
                            const ModTransform *transform, double dt,
                            void *cloud_aos, int32_t *labels,
                            ModObject *objects, int32_t max_objects, int32_t *n_objects);

/* Clusterer alone on a host PointCloud2 payload (ClustererNodelet::dataCB, clusterer_nodelet.cpp:221-242).
 * point_step/row_step as in sensor_msgs/PointCloud2; fields x,y,z,vx,vy,vz at offsets 0,4,8,16,20,24.
 * A context without a camera (a clusterer-only process: the nodelet) takes the image size from this call — the clusterer needs
 * nothing else of the camera; parameters must have been set.  With a camera set, a cloud of another size is an error. */
int  mod_cluster_cloud_host(ModContext *ctx, const void *cloud, int32_t width, int32_t height,
                            int32_t point_step, int32_t row_step,
                            int32_t *labels, ModObject *objects, int32_t max_objects, int32_t *n_objects);

/* ---- host streaming: frames in flight, copies overlapped with the kernels --------------------------------------- */
/* The reference overlaps construct() of frame t with the estimators of frame t+1 (construct_thread_,
 * scene_flow_constructor.cpp:389-392) and keeps disparity_previous_ from the last callback (:397-398).  Here up to
 * MOD_PIPELINE_DEPTH frames are in flight: the input copy of frame t+1 and the result copy of frame t-1 run on their own
 * HIP streams beside the kernels of frame t.
 *   mod_submit_frame_host  enqueues one frame and returns at once with a ticket.  Arguments as mod_process_frame_host,
 *       except: disparity_prev may be NULL, then the disparity_now of the previous submit (still resident in HBM) is used
 *       (MOD_SKIP_NO_DISPARITY_PREV if there was none); outputs are written asynchronously — every pointer must stay valid
 *       and untouched until the ticket is collected.  Skip codes as construct(); MOD_ERR_CAPACITY when MOD_PIPELINE_DEPTH
 *       frames are already in flight.  The large buffers (inputs, cloud, labels) should come from mod_host_malloc: a copy
 *       to or from pageable memory makes the call wait for that copy; `objects` may be ordinary memory (filled at collect time).
 *   mod_collect_frame_host waits for a ticket (tickets complete in submission order; the oldest one must be collected
 *       first) and reports its object count.
 * Reconfiguration while frames are in flight: the PARAMETERS may change between two submits (mod_set_params; the reference's
 * reconfigureCB runs between two stereoCallbacks, scene_flow_constructor.cpp:401-407) — every kernel takes them by value at
 * submit time, so a frame in flight completes with the parameters of ITS submit and the next submit uses the new ones
 * (tests/test_gpu_host_stream.py).  The CAMERA must not change while frames are in flight (its ray tables live in HBM and are
 * read by the kernels of the frames in flight): collect every ticket first. */
#define MOD_PIPELINE_DEPTH 3
int  mod_submit_frame_host(ModContext *ctx,
                           const float *disparity_now, const float *disparity_prev, const float *flow,
                           const ModTransform *transform, double dt,
                           void *cloud_aos, int32_t *labels, ModObject *objects, int32_t max_objects,
                           int32_t *ticket);
int  mod_collect_frame_host(ModContext *ctx, int32_t ticket, int32_t *n_objects);

/* Stereo images in, moving objects out: the device-resident form of stereoCallback() for BASELINE config 5
 * (scene_flow_constructor.cpp:365-399: estimateDisparity, then construct() on construct_thread_ beside the next frame's
 * estimators, then disparity_previous_ = disparity_now_).  The two 8-bit images go to the GPU on the copy stream, the on-GPU
 * estimator (mod_sgm_compute_dev) writes the disparity plane straight into the pipe's ring — where it serves as `now` of this frame
 * and as `previous` of the next — and scene flow + clustering follow on the context's stream: the disparity never crosses PCIe
 * (mod_sgm_compute_host + mod_submit_frame_host move it there and back: two trips of 4 bytes per pixel and frame).
 *   left / right    W*H bytes each, row-major (sensor_msgs/Image mono8); NULL = the estimator has nothing to work on:
 *                   MOD_SKIP_NO_DISPARITY_NOW, and the next frame has no previous disparity (disparity_now_.reset(), :272-276)
 *   sgm             estimator parameters; the camera's min / max_disparity must describe its output (0 and disparities - 1)
 *   flow, transform, dt, cloud_aos, labels, objects, max_objects, ticket: as mod_submit_frame_host.  A frame that ends at one
 *                   of construct()'s guards (no flow, no previous disparity, no transform) still had its disparity estimated:
 *                   it IS the next frame's previous one (:397-398)
 *   disparity       optional host copy of the disparity image (32FC1, -1 where no match was found), valid after collect
 * Collected with mod_collect_frame_host like any other ticket. */
int  mod_submit_stereo_host(ModContext *ctx, const uint8_t *left, const uint8_t *right, const ModSgmParams *sgm,
                            const float *flow, const ModTransform *transform, double dt,
                            void *cloud_aos, int32_t *labels, ModObject *objects, int32_t max_objects,
                            float *disparity, int32_t *ticket);
/* disparity_now_.reset() of a failed estimateDisparity (scene_flow_constructor.cpp:272-276): the next submit without an
 * explicit disparity_prev reports MOD_SKIP_NO_DISPARITY_PREV instead of pairing with a stale frame. */
int  mod_forget_previous(ModContext *ctx);
int  mod_host_malloc(ModContext *ctx, uint64_t bytes, void **host_ptr);   /* page-locked host memory */
int  mod_host_free(ModContext *ctx, void *host_ptr);

/* ---- device memory helpers (so a non-HIP host language can own HBM buffers) ------------------------------- */
int  mod_malloc(ModContext *ctx, uint64_t bytes, void **dev_ptr);
int  mod_free(ModContext *ctx, void *dev_ptr);
int  mod_memcpy_h2d(ModContext *ctx, void *dev_dst, const void *host_src, uint64_t bytes);
int  mod_memcpy_d2h(ModContext *ctx, void *host_dst, const void *dev_src, uint64_t bytes);

/* ---- measurement ------------------------------------------------------------------------------------------ */
/* Stage timers: each stage selected by `stage_mask` (bit i = stage i; MOD_PROFILE_ALL for all of them, 0 = off) is
 * bracketed by HIP events on the context's stream in the calls that follow.  Every event pair costs a few microseconds of
 * stream time, so a throughput measurement should select only the stage it prices. */
#define MOD_STAGE_SCENE_FLOW  0   /* k_scene_flow_v4 / _v1: fused scene-flow kernel (+ dynamic mask)            */
#define MOD_STAGE_CCL_TILE    1   /* tile stage: k_ccl_bits<n> + k_ccl_tile_list (k_ccl_tile for n > 10): tile-local
                                     connected components (+ k_dynamic_mask / k_tile_flags for a caller's cloud)  */
#define MOD_STAGE_CCL_LINK    2   /* k_ccl_link: cross-tile unions                                              */
#define MOD_STAGE_CCL_MERGE   3   /* k_ccl_merge: root-level flatten + record folding, and the size filter with the
                                     reference numbering: in k_ccl_merge's last workgroup per frame for batches of up
                                     to 16 frames, as k_select behind it for larger ones (a stage of its own, "select",
                                     until ABI version 1)                                                        */
#define MOD_STAGE_FINAL       4   /* k_final: labels plane + member compaction + cluster boxes                  */
#define MOD_STAGE_MEDIAN      5   /* k_median + k_median_ties: median-velocity member; object ids               */
#define MOD_STAGE_CLUSTER_GROUP 6 /* the whole cluster stage of a call, first launch to last (stages 1..5 and, in a chunked
                                     mod_process_dev, the overlap of its chunks: see ModConfig.batch_chunks).  While a timer
                                     of one of the stages 1..5 is on, mod_process_dev runs un-chunked: side by side the kernels
                                     of different chunks would be priced with each other's load */
#define MOD_STAGE_COUNT       7
#define MOD_PROFILE_ALL        0x7f
int  mod_set_profiling(ModContext *ctx, int32_t stage_mask);
/* Accumulated milliseconds and launch count of a stage since the last reset (synchronises the stream). */
int  mod_get_stage_time(ModContext *ctx, int32_t stage, double *total_ms, int64_t *calls);
int  mod_reset_stage_times(ModContext *ctx);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MOD_SF_H_ */
