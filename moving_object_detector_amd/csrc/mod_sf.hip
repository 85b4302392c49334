// mod_sf.hip — C ABI (include/mod_sf.h) over the gfx950 kernels: context, scratch, parameter folding, stage timers.
// Host-side only; the kernels live in sceneflow.hip and cluster.hip.
#include "../../include/mod_sf.h"
#include "exact_div.h"
#include "mod_launch.h"
#include "mod_sf_debug.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace {

constexpr int kRing = 4;   // pinned staging slots for the per-frame constants
constexpr int kMaxChunks = 4;       // mod_process_dev cuts a large batch into at most this many chunks (ModConfig.batch_chunks)
constexpr int kChunkMinFrames = 32; // ... of at least this many frames each (a chunk must still fill the GPU on its own)
constexpr int kAutoChunks = 1;      // ModConfig.batch_chunks == 0: one piece (see process_chunked for what two chunks gain, and when)

struct EventPair { hipEvent_t a, b; };

struct Buffers {
  double *rayx = nullptr, *rayy = nullptr;
  FrameConst *fc = nullptr;                 // [maxF]
  uint64_t *mask = nullptr, *lroot = nullptr;
  float2 *zrange = nullptr;                 // [maxF][H][mask_words] depth range of the dynamic pixels of each mask word (fused path)
  int32_t *parent = nullptr;
  int32_t *rsize = nullptr, *rkey = nullptr;
  ClusterBox *cbox = nullptr;
  int32_t *counters = nullptr;
  ClusterInfo *clusters = nullptr;          // 2 x [maxF][max_objects]
  uint32_t *mbits = nullptr, *mpix = nullptr;
  uint32_t *worklist = nullptr;   // [2][F * max_objects]: all clusters of a launch, then the ambiguous ones
  unsigned long long *dbg = nullptr;
  uint2 *requests = nullptr;
  int32_t *tilehdr = nullptr;
  uint32_t *tilelist = nullptr;
  size_t req_alloc = 0;                     // entries currently allocated for `requests`
  // one-frame staging for the *_host entry points (allocated on first use)
  float *h_dnow = nullptr, *h_dprev = nullptr, *h_flow = nullptr, *h_planes = nullptr;
  void *h_aos = nullptr;
  int32_t *h_labels = nullptr, *h_nobj = nullptr;
  ModObject *h_objects = nullptr;
  // on-GPU disparity (allocated on first use)
  uint32_t *sgm_census = nullptr;
  uint8_t *sgm_maps = nullptr;
  uint8_t *sgm_S = nullptr;                 // [paths][group][H][W][D] path cost volumes
  int sgm_D = 0, sgm_G = 0;                 // disparities / frames per group the scratch is sized for
  // the aggregation paths are independent of each other: they run side by side on these streams (forked from / joined to the
  // context's stream with events), so that the waves of one path fill the SIMD slots another leaves idle
  hipStream_t sgm_side[8] = {};
  hipEvent_t sgm_fork[2] = {}, sgm_join[2][8] = {};
};

}  // namespace

struct ModContext {
  ModConfig cfg{};
  ModCamera cam{};
  ModParams prm{};
  bool has_cam = false, has_prm = false;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  DevCam dc{};
  Buffers b;
  int max_objects = 0;
  size_t maxN = 0;
  int max_mask_words = 0;
  // host streaming (mod_submit_frame_host): per-slot device buffers, a ring of MOD_PIPELINE_DEPTH + 1 disparity planes
  // (frame t's plane is frame t+1's "previous"), two copy streams and the events that order them with the kernels
  struct Pipe {
    bool ready = false;
    hipStream_t h2d = nullptr, d2h = nullptr;
    float *dnow[MOD_PIPELINE_DEPTH + 1] = {}, *dprev[MOD_PIPELINE_DEPTH] = {}, *flow[MOD_PIPELINE_DEPTH] = {};
    float *planes[MOD_PIPELINE_DEPTH] = {};
    void *aos[MOD_PIPELINE_DEPTH] = {};
    int32_t *labels[MOD_PIPELINE_DEPTH] = {}, *nobj[MOD_PIPELINE_DEPTH] = {};
    ModObject *objects[MOD_PIPELINE_DEPTH] = {};
    int32_t *h_n[MOD_PIPELINE_DEPTH] = {};         // pinned: object count of the slot's frame
    ModObject *h_obj[MOD_PIPELINE_DEPTH] = {};     // pinned: its objects (handed to the caller's array at collect time)
    ModObject *user_obj[MOD_PIPELINE_DEPTH] = {};
    int32_t user_cap[MOD_PIPELINE_DEPTH] = {};
    hipEvent_t ev_in[MOD_PIPELINE_DEPTH] = {}, ev_done[MOD_PIPELINE_DEPTH] = {}, ev_out[MOD_PIPELINE_DEPTH] = {};
    uint8_t *img[MOD_PIPELINE_DEPTH] = {};         // mod_submit_stereo_host: the slot's two 8-bit images
    hipEvent_t ev_img[MOD_PIPELINE_DEPTH] = {};    // ... the estimator has been enqueued behind them (context stream)
    bool img_used[MOD_PIPELINE_DEPTH] = {};
    hipEvent_t ev_ring = nullptr;                  // last disparity plane written by kernels (stereo path)
    // a ring plane may still be on its way to a caller's `disparity` buffer (result stream) when a frame that ended at a guard —
    // it takes a plane but no ticket — has advanced the ring back to it: the plane's next writer waits for that copy
    hipEvent_t ev_plane_read[MOD_PIPELINE_DEPTH + 1] = {};
    bool plane_read_pending[MOD_PIPELINE_DEPTH + 1] = {};
    bool ring_by_kernels = false;
    int64_t dring = 0;                             // disparity planes handed out so far: plane of the next frame = dring % (DEPTH + 1)
    int64_t seq = 0;                               // frames submitted so far
    int in_flight = 0;
    bool have_prev = false;                        // dnow[(seq - 1) % (DEPTH + 1)] holds the previous frame's disparity
  } pipe;
  FrameConst *pinned[kRing] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t pinned_ev[kRing] = {nullptr, nullptr, nullptr, nullptr};
  int ring_pos = 0;
  // chunks of a large batch (process_chunked): chunk 0 runs on the context's stream, chunk k > 0 on chunk_stream[k - 1], forked from
  // and joined to the context's stream with events, so the call keeps the stream semantics of every other entry point
  hipStream_t chunk_stream[kMaxChunks - 1] = {};
  hipEvent_t ev_fork = nullptr, ev_join[kMaxChunks - 1] = {};
  // The tile headers (word 0) and the cluster counters are ZERO between calls: the context's first call clears them, and the cluster stage's
  // last readers (k_final; k_median_ties' last workgroup) clear what a call has set — two memsets less in front of every call, which
  // a small batch feels (a launch costs it ~8 us of GPU time whatever it does).  False while a call is being enqueued; a call that
  // failed half-way leaves it false and the next one clears the scratch itself.
  bool scratch_clean = false;
  int profiling = 0;                          // stage mask of mod_set_profiling
  std::vector<EventPair> pending[MOD_STAGE_COUNT];
  std::vector<EventPair> free_events;
  double stage_ms[MOD_STAGE_COUNT] = {};
  int64_t stage_calls[MOD_STAGE_COUNT] = {};
  std::string err;
};

namespace {

int fail(ModContext *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                              \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return fail(ctx, MOD_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)

// smallest float t with (double)t >= th: for float a, ((double)a >= th) <=> (a >= t)
float ceil_to_f32(double th) {
  if (std::isnan(th)) return std::nanf("");
  float t = (float)th;
  if ((double)t < th) t = std::nextafterf(t, INFINITY);
  return t;
}
// largest float t with (double)t <= th: for float a, ((double)a > th) <=> (a > t)
float floor_to_f32(double th) {
  if (std::isnan(th)) return std::nanf("");
  float t = (float)th;
  if ((double)t > th) t = std::nextafterf(t, -INFINITY);
  return t;
}

// Eigen::Quaterniond::toRotationMatrix operation order (tf2::transformToEigen, scene_flow_constructor.cpp:411);
// the quaternion is used as given, without normalisation.
void transform_to_rows(const ModTransform &tf, double m[12]) {
  const double x = tf.q[0], y = tf.q[1], z = tf.q[2], w = tf.q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  m[0] = 1.0 - (tyy + tzz); m[1] = txy - twz;         m[2] = txz + twy;          m[3] = tf.t[0];
  m[4] = txy + twz;         m[5] = 1.0 - (txx + tzz); m[6] = tyz - twx;          m[7] = tf.t[1];
  m[8] = txz - twy;         m[9] = tyz + twx;         m[10] = 1.0 - (txx + tyy); m[11] = tf.t[2];
}

// Smallest F32 a >= 0 with sqrtf(a) >= th, so that the kernels test a sum of squares instead of taking its root:
// sqrtf is correctly rounded (host libm and the device build alike) and therefore monotone, which makes the two tests
// agree on every input (a NaN fails both; th <= 0 accepts every sum of squares, which is >= +0).
float sqrt_threshold_sq(float th) {
  if (std::isnan(th)) return th;                            // never true, like the original comparison
  if (!(th > 0.0f)) return 0.0f;
  const float inf = std::numeric_limits<float>::infinity();
  if (std::isinf(th)) return inf;
  float a = (float)((double)th * (double)th);
  while (a > 0.0f && sqrtf(std::nextafterf(a, 0.0f)) >= th) a = std::nextafterf(a, 0.0f);
  while (a < inf && !(sqrtf(a) >= th)) a = std::nextafterf(a, inf);
  return a;
}

void refresh_devcam(ModContext *c) {
  DevCam &d = c->dc;
  d.W = c->cam.width; d.H = c->cam.height;
  d.mask_words = mod_mask_words(d.W);
  d.n = c->prm.neighbor_distance;
  d.cluster_size = c->prm.cluster_size;
#if defined(MOD_PHASE_COUNTERS) || defined(MOD_ABLATION)
  { const char *e = getenv("MOD_DEBUG"); d.debug = e ? atoi(e) : 0; }   // diagnostic builds only (mod_sf_debug.h)
#else
  d.debug = 0;
#endif
  d.fT = c->cam.disp_f * c->cam.disp_T;               // F32 product, exactly the reference's `focal_length * baseline`
  d.dmin = c->cam.min_disparity; d.dmax = c->cam.max_disparity;
  d.flow_th_sq = sqrt_threshold_sq((float)c->prm.dynamic_flow_diff);
  d.speed_th_sq = sqrt_threshold_sq(ceil_to_f32(c->prm.dynamic_speed));
  d.depth_th = floor_to_f32(c->prm.depth_diff);
  d.speed_th_d = c->prm.dynamic_speed;
  d.fx = c->cam.fx; d.fy = c->cam.fy; d.cx = c->cam.cx; d.cy = c->cam.cy; d.Tx = c->cam.Tx; d.Ty = c->cam.Ty;
  d.rayx = c->b.rayx; d.rayy = c->b.rayy;
}

int check_ready(ModContext *c, int frames) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  if (!c->has_cam || !c->has_prm) return fail(c, MOD_ERR_NOT_CONFIGURED, "camera and parameters must be set first");
  if (frames < 1) return fail(c, MOD_ERR_INVALID_ARGUMENT, "frames must be >= 1");
  if (frames > c->cfg.max_frames) return fail(c, MOD_ERR_CAPACITY, "frames exceeds ModConfig.max_frames");
  return MOD_OK;
}

struct StageTimer {
  ModContext *c; int stage; hipStream_t s; EventPair ev{}; bool on;
  StageTimer(ModContext *ctx, int st, hipStream_t stream) : c(ctx), stage(st), s(stream), on((ctx->profiling >> st) & 1) {
    if (!on) return;
    if (!c->free_events.empty()) { ev = c->free_events.back(); c->free_events.pop_back(); }
    else { (void)hipEventCreate(&ev.a); (void)hipEventCreate(&ev.b); }
    (void)hipEventRecord(ev.a, s);
  }
  ~StageTimer() {
    if (!on) return;
    (void)hipEventRecord(ev.b, s);
    c->pending[stage].push_back(ev);
  }
};

void drain_timers(ModContext *c) {
  for (int s = 0; s < MOD_STAGE_COUNT; s++) {
    for (EventPair &ev : c->pending[s]) {
      (void)hipEventSynchronize(ev.b);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) { c->stage_ms[s] += ms; c->stage_calls[s] += 1; }
      c->free_events.push_back(ev);
    }
    c->pending[s].clear();
  }
}

void fill_frame_const(FrameConst &h, const ModTransform &tf, double dt) {
  transform_to_rows(tf, h.m);
  h.dt = dt;
  // |t_i + (m_i0 x + (m_i1 y + m_i2 z))| <= tmax + 3 mmax B stays below FLT_MAX / 2 (so the F32 cast is finite, and no
  // intermediate can overflow or turn NaN) for every |x|,|y|,|z| <= B.  Non-finite transforms get B = 0: always compute.
  double mmax = 0.0, tmax = 0.0;
  bool finite = true;
  for (int i = 0; i < 12; i++) {
    const double v = std::fabs(h.m[i]);
    finite = finite && std::isfinite(v);
    if (i % 4 == 3) tmax = std::max(tmax, v); else mmax = std::max(mmax, v);
  }
  double B = 0.0;
  if (finite && tmax < 1e37) B = std::min((1.7e38 - tmax) / (3.0 * std::max(mmax, 1e-30)), 1e30);
  h.pad[0] = B;
  // velocity = difference / dt through the correctly rounded reciprocal (exact_div.h) when dt is an ordinary number
  const bool usable = exact_div::reciprocal_usable(h.dt);
  h.pad[1] = usable ? 1.0 / h.dt : 0.0;
  h.pad[2] = usable ? 1.0 : 0.0;
}

// The per-frame constants of a batch.  Up to MOD_SF_INLINE_FRAMES frames: into `inl`, which the scene-flow launch passes in its
// kernel arguments (inl->used) — nothing is copied, nothing waited for.  Larger batches: through the pinned ring into c->b.fc.
struct InlineConsts { FrameConst v[MOD_SF_INLINE_FRAMES]; bool used = false; };

int upload_frame_consts(ModContext *c, const ModFrameBatch *in, InlineConsts *inl) {
  if (inl && in->frames <= MOD_SF_INLINE_FRAMES) {
    for (int f = 0; f < in->frames; f++) fill_frame_const(inl->v[f], in->transforms[f], in->dt[f]);
    inl->used = true;
    return MOD_OK;
  }
  const int slot = c->ring_pos;
  c->ring_pos = (c->ring_pos + 1) % kRing;
  HIP_TRY(c, hipEventSynchronize(c->pinned_ev[slot]));   // the slot's previous copy has left the host buffer
  FrameConst *h = c->pinned[slot];
  for (int f = 0; f < in->frames; f++) fill_frame_const(h[f], in->transforms[f], in->dt[f]);
  // a kernel of ours reads the pinned slot over the host link (hipHostMalloc memory is mapped into the device's address space):
  // the runtime's own host-to-device copy is a blit kernel too, and the kernel behind it started 5 us after it had ended
  launch_copy_words((const unsigned long long *)h, (unsigned long long *)c->b.fc, sizeof(FrameConst) / 8 * (size_t)in->frames, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(c->pinned_ev[slot], c->stream));
  return MOD_OK;
}

int check_batch(ModContext *c, const ModFrameBatch *in) {
  if (!in) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null batch");
  int rc = check_ready(c, in->frames);
  if (rc) return rc;
  // construct()'s guards: nothing is published when an input is missing (scene_flow_constructor.cpp:104,110,122,127,133)
  if (!in->flow) return MOD_SKIP_NO_FLOW;
  if (!in->disparity_prev) return MOD_SKIP_NO_DISPARITY_PREV;
  if (!in->transforms || !in->dt) return MOD_SKIP_NO_TRANSFORM;
  if (!in->disparity_now) return MOD_SKIP_NO_DISPARITY_NOW;
  return MOD_OK;
}

int begin_cluster_scratch(ModContext *c) {
  if (!c->scratch_clean) {
    const size_t tiles = (size_t)c->max_mask_words * ((c->cfg.max_height + ccl_tile_rows() - 1) / ccl_tile_rows());
    HIP_TRY(c, hipMemsetAsync(c->b.tilehdr, 0, sizeof(int32_t) * 2 * tiles * c->cfg.max_frames, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->b.counters, 0, sizeof(int32_t) * 8 * c->cfg.max_frames, c->stream));
  }
  c->scratch_clean = false;
  return MOD_OK;
}

// frames [f0, f0 + n) of a batch: every per-frame pointer of the launch moves to the chunk's first frame, so that "frame 0 of the
// launch" — where the clustering kernels keep their launch-wide counters and lists — is the chunk's own
struct Chunk { int f0, n; };

int check_scene_flow_out(ModContext *c, const ModSceneFlowPlanes *out, bool xy_optional) {
  if (!out || !out->z || !out->vx || !out->vy || !out->vz)
    return fail(c, MOD_ERR_INVALID_ARGUMENT, "scene-flow output planes z,vx,vy,vz are required");
  // x and y: both or neither; neither only where the call says so (mod_process_dev: the cluster stage recomputes them from z)
  if ((out->x == nullptr) != (out->y == nullptr) || (!out->x && !xy_optional))
    return fail(c, MOD_ERR_INVALID_ARGUMENT, xy_optional ? "scene-flow output planes x and y: pass both or neither" : "scene-flow output planes x,y,z,vx,vy,vz are required");
  return MOD_OK;
}

// the scene-flow kernel over one chunk of the batch (the per-frame constants of the WHOLE batch are in c->b.fc by now)
void enqueue_scene_flow(ModContext *c, const ModFrameBatch *in, const ModSceneFlowPlanes *out, uint64_t *mask, bool tile_flags, Chunk ch,
                        hipStream_t s, const InlineConsts *inl = nullptr) {
  const size_t N = (size_t)c->dc.W * c->dc.H, f0 = (size_t)ch.f0, MWH = (size_t)c->dc.mask_words * c->dc.H;
  SfArgs a;
  a.dnow = in->disparity_now + f0 * N; a.dprev = in->disparity_prev + f0 * N; a.flow = in->flow + f0 * N * 2;
  a.x = out->x ? out->x + f0 * N : nullptr; a.y = out->y ? out->y + f0 * N : nullptr;
  a.z = out->z + f0 * N; a.vx = out->vx + f0 * N; a.vy = out->vy + f0 * N; a.vz = out->vz + f0 * N;
  a.mask = mask ? mask + f0 * MWH : nullptr;
  a.aos = out->cloud_aos ? (float4 *)out->cloud_aos + f0 * N * 2 : nullptr;
  a.depth = out->depth ? out->depth + f0 * N : nullptr;
  a.sflow = out->static_flow ? out->static_flow + f0 * N * 2 : nullptr;
  a.fc = c->b.fc + f0;
  a.tilehdr = nullptr; a.zrange = nullptr; a.tile_rows = ccl_tile_rows(); a.tiles_x = c->dc.mask_words;
  a.tiles_per_frame = c->dc.mask_words * ((c->dc.H + ccl_tile_rows() - 1) / ccl_tile_rows());
  a.dbg = c->b.dbg;
  if (tile_flags && mask) {      // the clustering follows: the kernel's epilogue also marks the cluster tiles that hold a dynamic pixel
    a.tilehdr = c->b.tilehdr + f0 * 2 * (size_t)a.tiles_per_frame;   // (all zero: ModContext::scratch_clean)
    a.zrange = c->b.zrange + f0 * MWH; // ... and leaves the depth range of every non-zero mask word's dynamic pixels for the tile stage
  }
  StageTimer t(c, MOD_STAGE_SCENE_FLOW, s);
  launch_scene_flow(c->dc, a, ch.n, (inl && inl->used) ? inl->v : nullptr, s);
}

int run_scene_flow(ModContext *c, const ModFrameBatch *in, const ModSceneFlowPlanes *out, uint64_t *mask, bool tile_flags, bool xy_optional) {
  int rc = check_scene_flow_out(c, out, xy_optional);
  if (rc) return rc;
  InlineConsts inl;
  if ((rc = upload_frame_consts(c, in, &inl))) return rc;
  enqueue_scene_flow(c, in, out, mask, tile_flags, Chunk{0, in->frames}, c->stream, &inl);   // (tile_flags: the caller has called begin_cluster_scratch)
  HIP_TRY(c, hipGetLastError());
  return MOD_OK;
}

int check_cluster_io(ModContext *c, const ModSceneFlowPlanes *pl, bool flags_ready, const ModClusterOut *out) {
  // flags_ready: the planes are this call's own scene-flow output (mod_process_dev), where x, y are functions of z and may be absent
  if (!pl || !pl->z || !pl->vx || !pl->vy || !pl->vz || (!flags_ready && (!pl->x || !pl->y)))
    return fail(c, MOD_ERR_INVALID_ARGUMENT, "cluster input planes x,y,z,vx,vy,vz are required");
  if (!out || !out->objects || !out->n_objects)
    return fail(c, MOD_ERR_INVALID_ARGUMENT, "cluster outputs objects, n_objects are required");
  return MOD_OK;
}

// The clustering of one chunk on stream s.
int enqueue_cluster(ModContext *c, Chunk ch, const ModSceneFlowPlanes *pl, const uint64_t *mask, bool mask_ready, bool flags_ready,
                    const ModClusterOut *out, hipStream_t s) {
  const size_t N = (size_t)c->dc.W * c->dc.H, f0 = (size_t)ch.f0, MWH = (size_t)c->dc.mask_words * c->dc.H, MO = (size_t)c->max_objects;
  const size_t tiles = (size_t)c->dc.mask_words * ((c->dc.H + ccl_tile_rows() - 1) / ccl_tile_rows());
  const int frames = ch.n;
  ClArgs a;
  a.x = pl->x ? pl->x + f0 * N : nullptr; a.y = pl->y ? pl->y + f0 * N : nullptr;
  a.z = pl->z + f0 * N; a.vx = pl->vx + f0 * N; a.vy = pl->vy + f0 * N; a.vz = pl->vz + f0 * N;
  a.mask = mask + f0 * MWH; a.zrange = flags_ready ? c->b.zrange + f0 * MWH : nullptr; a.lroot = c->b.lroot + f0 * MWH;
  a.parent = c->b.parent + f0 * N; a.rootlist = (int32_t *)c->b.mpix + f0 * N;
  a.labels = out->labels ? out->labels + f0 * N : nullptr; a.rsize = c->b.rsize + f0 * N; a.rkey = c->b.rkey + f0 * N;
  a.cbox = c->b.cbox + f0 * MO; a.counters = c->b.counters + f0 * 8; a.clusters = c->b.clusters + f0 * MO;
  a.mbits = c->b.mbits + f0 * N; a.mpix = c->b.mpix + f0 * N;
  a.worklist = c->b.worklist + f0 * MO; a.tielist = c->b.worklist + (size_t)c->cfg.max_frames * MO + f0 * MO;
  a.objects = (ModObject *)out->objects + f0 * MO; a.n_objects = out->n_objects + f0;
  a.n_clusters = out->n_clusters ? out->n_clusters + f0 : nullptr; a.max_objects = c->max_objects; a.dbg = c->b.dbg;
  a.xy_from_z = flags_ready ? 1 : 0;                  // only mod_process_dev's fused path hands over its own scene-flow planes
  const int req_cap = ccl_request_capacity(c->prm.neighbor_distance);
  a.requests = c->b.requests + f0 * tiles * (size_t)req_cap; a.tilehdr = c->b.tilehdr + f0 * tiles * 2; a.tilelist = c->b.tilelist + f0 * tiles;
  a.req_cap = req_cap;
  ClusterInfo *const rank_scratch = c->b.clusters + (size_t)c->cfg.max_frames * MO + f0 * MO;   // second half of the allocation
  {
    StageTimer t(c, MOD_STAGE_CCL_TILE, s);
    if (!mask_ready) launch_dynamic_mask(c->dc, frames, a.vx, a.vy, a.vz, (uint64_t *)a.mask, s);
    if (!flags_ready) launch_tile_flags(c->dc, a, frames, s);
    launch_ccl_tile(c->dc, a, frames, s);                 // (the counters are zero: ModContext::scratch_clean)
  }
  { StageTimer t(c, MOD_STAGE_CCL_LINK, s); launch_ccl_link(c->dc, a, frames, s); }
  { StageTimer t(c, MOD_STAGE_CCL_MERGE, s); launch_ccl_merge(c->dc, a, frames, rank_scratch, s); }
  { StageTimer t(c, MOD_STAGE_FINAL, s); launch_final(c->dc, a, frames, s); }
  { StageTimer t(c, MOD_STAGE_MEDIAN, s); launch_median(c->dc, a, frames, s); }
  return MOD_OK;
}

int run_cluster(ModContext *c, int frames, const ModSceneFlowPlanes *pl, const uint64_t *mask, bool mask_ready, bool flags_ready,
                const ModClusterOut *out) {
  int rc = check_cluster_io(c, pl, flags_ready, out);
  if (rc) return rc;
  StageTimer t(c, MOD_STAGE_CLUSTER_GROUP, c->stream);
  if ((rc = enqueue_cluster(c, Chunk{0, frames}, pl, mask, mask_ready, flags_ready, out, c->stream))) return rc;
  HIP_TRY(c, hipGetLastError());
  return MOD_OK;
}

int ensure_chunk_streams(ModContext *c) {
  if (c->ev_fork) return MOD_OK;
  for (hipStream_t &q : c->chunk_stream) if (!q) HIP_TRY(c, hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
  for (hipEvent_t &e : c->ev_join) if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  // a few fork / join rounds, once: the runtime builds the cross-stream signalling of the new streams here, not inside the first call
  for (int i = 0; i < 16; i++) {
    HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
    for (int k = 0; k < kMaxChunks - 1; k++) {
      HIP_TRY(c, hipStreamWaitEvent(c->chunk_stream[k], c->ev_fork, 0));
      HIP_TRY(c, hipEventRecord(c->ev_join[k], c->chunk_stream[k]));
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join[k], 0));
    }
  }
  return MOD_OK;
}

// How many chunks a fused call of `frames` frames runs in (ModConfig.batch_chunks; mod_sf.h).  One chunk while a per-kernel
// cluster timer is on: the kernels of different chunks run side by side, and a timer would price its kernel with its neighbours' load.
int chunk_count(const ModContext *c, int frames) {
  const int want = c->cfg.batch_chunks;
  const int per_kernel = ((1 << MOD_STAGE_CCL_TILE) | (1 << MOD_STAGE_CCL_LINK) | (1 << MOD_STAGE_CCL_MERGE) | (1 << MOD_STAGE_FINAL) |
                          (1 << MOD_STAGE_MEDIAN));
  if (c->profiling & per_kernel) return 1;
  int n = want ? want : kAutoChunks;
  n = std::min(n, kMaxChunks);
  while (n > 1 && frames / n < kChunkMinFrames) n--;
  return std::max(n, 1);
}

// mod_process_dev on a large batch, cluster stage in chunks (ModConfig.batch_chunks >= 2; opt-in).  The cluster stage ends in kernels
// that wait instead of moving bytes (cross-tile links, the root merge + size filter, the median selection, the tie replay: a few
// workgroups chasing pointers) and begins with one that is bound by workgroup dispatch (three quarters of the tiles are empty).  Cut
// into chunks of frames whose kernel chains run side by side on streams of their own, the waiting kernels of one chunk share the GPU
// with the streaming kernels of another; the hardware interleaves them as their workgroups come.  Measured (round 5, 512 pairs,
// profiles/README.md): in a process that has been running for a second or more, 2 chunks take 0.7 - 2.9 % off the step (in-process
// A/B on four boxes: tools/chunk_ab.py, tools/step_trace.py; bench.py --steps 200 --warmup 50: 5.11 - 5.18 vs 5.23 ms); in the first
// ~25 calls of a fresh process they ADD 2 % (bench.py --steps 20 --warmup 5: 5.28 - 5.31 vs 5.15 - 5.21 ms) — the first calls show
// hitches of ~0.9 ms each (the host falls behind while the runtime grows what the new streams need; a burst of fills and fork /
// join rounds at stream creation did not remove them).  Hence not the default.  Also measured with switches that have left the
// code again (commit d9702f0 of round 5 has them: ModConfig.batch_chunks bits 8 and 9): chains held one kernel apart by events — no better than free-running ones; 3 or
// 4 chunks like 2; the scene-flow kernel cut into the chunks too — +0.5 ... +3 % (it is bandwidth-bound throughout and gains
// nothing from company), so it stays ONE launch over the whole batch ahead of the chunks.
int process_chunked(ModContext *c, const ModFrameBatch *in, const ModSceneFlowPlanes *pl, uint64_t *mask, const ModClusterOut *out, int C) {
  int rc = ensure_chunk_streams(c);
  if (rc) return rc;
  if ((rc = upload_frame_consts(c, in, nullptr))) return rc;
  enqueue_scene_flow(c, in, pl, mask, true, Chunk{0, in->frames}, c->stream);
  StageTimer group(c, MOD_STAGE_CLUSTER_GROUP, c->stream);
  HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
  for (int k = 0; k < C; k++) {
    const Chunk ch{(int)((int64_t)in->frames * k / C), (int)((int64_t)in->frames * (k + 1) / C - (int64_t)in->frames * k / C)};
    hipStream_t s = k ? c->chunk_stream[k - 1] : c->stream;
    if (k) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_fork, 0));
    if ((rc = enqueue_cluster(c, ch, pl, mask, true, true, out, s))) return rc;
    if (k) HIP_TRY(c, hipEventRecord(c->ev_join[k - 1], s));
  }
  for (int k = 1; k < C; k++) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join[k - 1], 0));
  HIP_TRY(c, hipGetLastError());
  return MOD_OK;
}

// allocates unless *p already points at a buffer (lazily built buffer sets can be resumed after a failed attempt without leaking)
template <class T>
hipError_t dalloc(T **p, size_t count) { return *p ? hipSuccess : hipMalloc((void **)p, count * sizeof(T)); }

}  // namespace

extern "C" {

int mod_abi_version(void) { return MOD_ABI_VERSION; }

int mod_create(const ModConfig *cfg, ModContext **out_ctx) {
  if (!cfg || !out_ctx) return MOD_ERR_INVALID_ARGUMENT;
  *out_ctx = nullptr;
  if (cfg->max_width < 1 || cfg->max_height < 1 || cfg->max_frames < 1) return MOD_ERR_INVALID_ARGUMENT;
  // launch geometry and index widths: frames ride in grid.y / grid.z (<= 65535), the scene-flow kernel addresses a frame's
  // planes with 32-bit byte offsets (32 B/px for the AoS cloud), cluster work items are frame * max_objects + cluster in 32 bits
  if (cfg->max_frames > 65535 || cfg->max_width > MOD_MAX_WIDTH) return MOD_ERR_INVALID_ARGUMENT;
  if (cfg->batch_chunks < 0 || cfg->batch_chunks > kMaxChunks) return MOD_ERR_INVALID_ARGUMENT;
  if ((uint64_t)cfg->max_width * (uint64_t)cfg->max_height >= (1ull << 27)) return MOD_ERR_INVALID_ARGUMENT;
  if (cfg->max_objects > 0 && (uint64_t)cfg->max_objects * (uint64_t)cfg->max_frames >= (1ull << 31)) return MOD_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device || cfg->device < 0) return MOD_ERR_NO_DEVICE;
  if (hipSetDevice(cfg->device) != hipSuccess) return MOD_ERR_NO_DEVICE;
  ModContext *c = new ModContext();
  c->cfg = *cfg;
  const size_t N = (size_t)cfg->max_width * cfg->max_height;
  const int F = cfg->max_frames;
  c->maxN = N;
  c->max_mask_words = mod_mask_words(cfg->max_width);
  c->max_objects = cfg->max_objects > 0 ? cfg->max_objects : (int)std::max<size_t>(1, N / 100);
  if (cfg->stream) c->stream = (hipStream_t)cfg->stream;
  else { if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return MOD_ERR_DEVICE; } c->own_stream = true; }
  const size_t mw = (size_t)F * cfg->max_height * c->max_mask_words;
  bool ok = true;
  ok &= dalloc(&c->b.rayx, cfg->max_width + 4) == hipSuccess;
  ok &= dalloc(&c->b.rayy, cfg->max_height + 4) == hipSuccess;
  ok &= dalloc(&c->b.fc, F) == hipSuccess;
  ok &= dalloc(&c->b.mask, mw) == hipSuccess;
  ok &= dalloc(&c->b.lroot, mw) == hipSuccess;
  ok &= dalloc(&c->b.zrange, mw) == hipSuccess;
  ok &= dalloc(&c->b.parent, (size_t)F * N) == hipSuccess;
  ok &= dalloc(&c->b.rsize, (size_t)F * N) == hipSuccess;   // 4 + 4 B per pixel of address space, touched only at roots
  ok &= dalloc(&c->b.rkey, (size_t)F * N) == hipSuccess;
  ok &= dalloc(&c->b.cbox, (size_t)F * c->max_objects) == hipSuccess;
  ok &= dalloc(&c->b.counters, (size_t)F * 8) == hipSuccess;
  ok &= dalloc(&c->b.clusters, (size_t)2 * F * c->max_objects) == hipSuccess;
  ok &= dalloc(&c->b.mbits, (size_t)F * N) == hipSuccess;
  ok &= dalloc(&c->b.mpix, (size_t)F * N) == hipSuccess;
  ok &= dalloc(&c->b.worklist, (size_t)2 * F * c->max_objects) == hipSuccess;
  {
    const size_t tiles = (size_t)c->max_mask_words * ((cfg->max_height + ccl_tile_rows() - 1) / ccl_tile_rows());
    ok &= dalloc(&c->b.tilehdr, (size_t)F * tiles * 2) == hipSuccess;
    ok &= dalloc(&c->b.tilelist, (size_t)F * tiles) == hipSuccess;
  }
  ok &= dalloc(&c->b.dbg, kDbgWords) == hipSuccess;
  if (ok) ok &= hipMemset(c->b.dbg, 0, kDbgWords * 8) == hipSuccess;
  // (tile headers and counters are cleared by the first call: scratch_clean starts false)
  if (ok) ok &= hipMemset((char *)c->b.dbg + 42 * 8, 0xFF, 8) == hipSuccess;   // slot 42 is a minimum
  for (int i = 0; i < kRing && ok; i++) {
    ok &= hipHostMalloc((void **)&c->pinned[i], sizeof(FrameConst) * F, hipHostMallocDefault) == hipSuccess;
    ok &= hipEventCreateWithFlags(&c->pinned_ev[i], hipEventDisableTiming) == hipSuccess;
    if (ok) ok &= hipEventRecord(c->pinned_ev[i], c->stream) == hipSuccess;
  }
  if (!ok) { mod_destroy(c); return MOD_ERR_DEVICE; }
  *out_ctx = c;
  return MOD_OK;
}

void mod_destroy(ModContext *c) {
  if (!c) return;
  (void)hipStreamSynchronize(c->stream);
  Buffers &b = c->b;
  void *dev[] = {b.rayx, b.rayy, b.fc, b.mask, b.lroot, b.zrange, b.parent, b.rsize, b.rkey, b.cbox, b.counters, b.clusters, b.mbits, b.mpix,
                 b.worklist, b.dbg, b.requests, b.tilehdr, b.tilelist, b.h_dnow, b.h_dprev, b.h_flow, b.h_planes, b.h_aos, b.h_labels, b.h_nobj, b.h_objects, b.sgm_census, b.sgm_maps, b.sgm_S};
  for (void *p : dev) if (p) (void)hipFree(p);
  for (int i = 0; i < kRing; i++) {
    if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
    if (c->pinned_ev[i]) (void)hipEventDestroy(c->pinned_ev[i]);
  }
  for (int s = 0; s < MOD_STAGE_COUNT; s++) for (EventPair &e : c->pending[s]) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (EventPair &e : c->free_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  {
    ModContext::Pipe &p = c->pipe;
    if (p.h2d) { (void)hipStreamSynchronize(p.h2d); (void)hipStreamDestroy(p.h2d); }
    if (p.d2h) { (void)hipStreamSynchronize(p.d2h); (void)hipStreamDestroy(p.d2h); }
    for (int i = 0; i <= MOD_PIPELINE_DEPTH; i++) if (p.dnow[i]) (void)hipFree(p.dnow[i]);
    for (int i = 0; i < MOD_PIPELINE_DEPTH; i++) {
      void *dv[] = {p.dprev[i], p.flow[i], p.planes[i], p.aos[i], p.labels[i], p.nobj[i], p.objects[i], p.img[i]};
      for (void *q : dv) if (q) (void)hipFree(q);
      if (p.h_n[i]) (void)hipHostFree(p.h_n[i]);
      if (p.h_obj[i]) (void)hipHostFree(p.h_obj[i]);
      hipEvent_t ev[] = {p.ev_in[i], p.ev_done[i], p.ev_out[i], p.ev_img[i]};
      for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    }
    if (p.ev_ring) (void)hipEventDestroy(p.ev_ring);
    for (hipEvent_t e : p.ev_plane_read) if (e) (void)hipEventDestroy(e);
  }
  for (hipStream_t q : c->b.sgm_side) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
  for (int k = 0; k < 2; k++) {
    if (c->b.sgm_fork[k]) (void)hipEventDestroy(c->b.sgm_fork[k]);
    for (hipEvent_t e : c->b.sgm_join[k]) if (e) (void)hipEventDestroy(e);
  }
  for (hipStream_t q : c->chunk_stream) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
  for (hipEvent_t e : c->ev_join) if (e) (void)hipEventDestroy(e);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *mod_last_error(const ModContext *c) { return c ? c->err.c_str() : "null context"; }

int mod_set_camera(ModContext *c, const ModCamera *cam) {
  if (!c || !cam) return MOD_ERR_INVALID_ARGUMENT;
  if (cam->width < 1 || cam->height < 1) return fail(c, MOD_ERR_INVALID_ARGUMENT, "camera size must be positive");
  if (cam->width > c->cfg.max_width || cam->height > c->cfg.max_height || (size_t)cam->width * cam->height > c->maxN)
    return fail(c, MOD_ERR_CAPACITY, "camera larger than ModConfig.max_width/max_height");
  c->cam = *cam;
  // projectPixelTo3dRay (image_geometry, melodic): ((u - cx - Tx)/fx, (v - cy - Ty)/fy, 1) in F64 — only a function of
  // the column / the row, so the two F64 divides per pixel of the reference become two table reads.
  std::vector<double> rx(cam->width + 4, 0.0), ry(cam->height + 4, 0.0);
  for (int u = 0; u < cam->width; u++) rx[u] = ((double)u - cam->cx - cam->Tx) / cam->fx;
  for (int v = 0; v < cam->height; v++) ry[v] = ((double)v - cam->cy - cam->Ty) / cam->fy;
  HIP_TRY(c, hipMemcpyAsync(c->b.rayx, rx.data(), sizeof(double) * rx.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->b.rayy, ry.data(), sizeof(double) * ry.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // rx/ry are stack-owned
  c->has_cam = true;
  refresh_devcam(c);
  return MOD_OK;
}

int mod_set_params(ModContext *c, const ModParams *p) {
  if (!c || !p) return MOD_ERR_INVALID_ARGUMENT;
  if (p->neighbor_distance < 1 || p->neighbor_distance > MOD_MAX_NEIGHBOR_DISTANCE)
    return fail(c, MOD_ERR_INVALID_ARGUMENT, "neighbor_distance must be in 1..16");
  if (p->cluster_size < 1) return fail(c, MOD_ERR_INVALID_ARGUMENT, "cluster_size must be >= 1");
  // at most N / cluster_size clusters can survive the size filter: the object arrays must hold them all, so that no
  // cluster is ever dropped (the default capacity N/100 covers the reference's whole range cluster_size >= 100)
  if (c->maxN / (size_t)p->cluster_size > (size_t)c->max_objects)
    return fail(c, MOD_ERR_CAPACITY, "ModConfig.max_objects is smaller than max_width*max_height / cluster_size");
  {   // link-request scratch grows with neighbor_distance
    const size_t tiles = (size_t)c->max_mask_words * ((c->cfg.max_height + ccl_tile_rows() - 1) / ccl_tile_rows());
    const size_t need = (size_t)c->cfg.max_frames * tiles * ccl_request_capacity(p->neighbor_distance);
    if (need > c->b.req_alloc) {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if (c->b.requests) HIP_TRY(c, hipFree(c->b.requests));
      c->b.requests = nullptr; c->b.req_alloc = 0;
      HIP_TRY(c, dalloc(&c->b.requests, need));
      c->b.req_alloc = need;
    }
  }
  c->prm = *p;
  c->has_prm = true;
  refresh_devcam(c);
  return MOD_OK;
}

int mod_get_camera(const ModContext *c, ModCamera *cam) {
  if (!c || !cam || !c->has_cam) return MOD_ERR_NOT_CONFIGURED;
  *cam = c->cam;
  return MOD_OK;
}
int mod_get_params(const ModContext *c, ModParams *p) {
  if (!c || !p || !c->has_prm) return MOD_ERR_NOT_CONFIGURED;
  *p = c->prm;
  return MOD_OK;
}

int mod_synchronize(ModContext *c) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MOD_OK;
}

// construct() publishes ~depth as soon as disparity_now exists, before the guards that end a frame without scene flow
// (scene_flow_constructor.cpp:110-123): on a skipped frame the depth plane is still produced when the caller asked for it.
static int depth_on_skip(ModContext *c, int skip, const ModFrameBatch *in, const ModSceneFlowPlanes *out) {
  if (skip <= 0 || skip == MOD_SKIP_NO_DISPARITY_NOW || !out || !out->depth || !in->disparity_now) return skip;
  launch_depth(c->dc, in->frames, in->disparity_now, out->depth, c->stream);
  HIP_TRY(c, hipGetLastError());
  return skip;
}

int mod_scene_flow_dev(ModContext *c, const ModFrameBatch *in, const ModSceneFlowPlanes *out) {
  int rc = check_batch(c, in);
  if (rc) return depth_on_skip(c, rc, in, out);
  return run_scene_flow(c, in, out, out ? out->dynamic_mask : nullptr, false, false);
}

// the scene-flow stage of the host entry points: their SoA planes are internal staging that no caller sees (the cloud leaves as
// 32-byte records straight from the kernel's registers), so the x and y planes are not written at all
static int scene_flow_staged(ModContext *c, const ModFrameBatch *in, const ModSceneFlowPlanes *out) {
  int rc = check_batch(c, in);
  if (rc) return depth_on_skip(c, rc, in, out);
  return run_scene_flow(c, in, out, out->dynamic_mask, false, true);
}

int mod_depth_image_dev(ModContext *c, int32_t frames, const float *disparity_now, float *depth) {
  int rc = check_ready(c, frames);
  if (rc) return rc;
  if (!disparity_now) return MOD_SKIP_NO_DISPARITY_NOW;
  if (!depth) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null depth plane");
  launch_depth(c->dc, frames, disparity_now, depth, c->stream);
  HIP_TRY(c, hipGetLastError());
  return MOD_OK;
}

int mod_dynamic_mask_dev(ModContext *c, int32_t frames, const float *vx, const float *vy, const float *vz, uint64_t *mask) {
  int rc = check_ready(c, frames);
  if (rc) return rc;
  if (!vx || !vy || !vz || !mask) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null plane");
  launch_dynamic_mask(c->dc, frames, vx, vy, vz, mask, c->stream);
  HIP_TRY(c, hipGetLastError());
  return MOD_OK;
}

int mod_cluster_dev(ModContext *c, int32_t frames, const ModSceneFlowPlanes *pl, const ModClusterOut *out) {
  int rc = check_ready(c, frames);
  if (rc) return rc;
  const bool have = pl && pl->dynamic_mask;
  if ((rc = check_cluster_io(c, pl, false, out)) || (rc = begin_cluster_scratch(c))) return rc;
  if ((rc = run_cluster(c, frames, pl, have ? pl->dynamic_mask : c->b.mask, have, false, out))) return rc;
  c->scratch_clean = true;
  return MOD_OK;
}

int mod_process_dev(ModContext *c, const ModFrameBatch *in, const ModSceneFlowPlanes *pl, const ModClusterOut *out) {
  int rc = check_batch(c, in);
  if (rc) return depth_on_skip(c, rc, in, pl);
  uint64_t *mask = (pl && pl->dynamic_mask) ? pl->dynamic_mask : c->b.mask;
  const int chunks = chunk_count(c, in->frames);
  if ((rc = check_scene_flow_out(c, pl, true)) || (rc = check_cluster_io(c, pl, true, out)) || (rc = begin_cluster_scratch(c))) return rc;
  if (chunks > 1) rc = process_chunked(c, in, pl, mask, out, chunks);
  else if (!(rc = run_scene_flow(c, in, pl, mask, true, true))) rc = run_cluster(c, in->frames, pl, mask, true, true, out);
  if (rc) return rc;
  c->scratch_clean = true;
  return MOD_OK;
}

int mod_pack_cloud_dev(ModContext *c, int32_t frames, const ModSceneFlowPlanes *pl, void *aos) {
  int rc = check_ready(c, frames);
  if (rc) return rc;
  if (!pl || !aos || !pl->x || !pl->y || !pl->z || !pl->vx || !pl->vy || !pl->vz) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null plane");
  launch_pack((size_t)frames * c->dc.W * c->dc.H, pl->x, pl->y, pl->z, pl->vx, pl->vy, pl->vz, aos, c->stream);
  HIP_TRY(c, hipGetLastError());
  return MOD_OK;
}

int mod_unpack_cloud_dev(ModContext *c, int32_t frames, const void *aos, const ModSceneFlowPlanes *pl) {
  int rc = check_ready(c, frames);
  if (rc) return rc;
  if (!pl || !aos || !pl->x || !pl->y || !pl->z || !pl->vx || !pl->vy || !pl->vz) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null plane");
  launch_unpack((size_t)frames * c->dc.W * c->dc.H, aos, pl->x, pl->y, pl->z, pl->vx, pl->vy, pl->vz, c->stream);
  HIP_TRY(c, hipGetLastError());
  return MOD_OK;
}

// ---- on-GPU disparity, first stages ---------------------------------------------------------------
int mod_sgm_census_dev(ModContext *c, int32_t frames, const uint8_t *image, uint32_t *census) {
  int rc = check_ready(c, frames);
  if (rc) return rc;
  if (!image || !census) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null image / census plane");
  launch_sgm_census(c->dc.W, c->dc.H, frames, image, census, c->stream);
  HIP_TRY(c, hipGetLastError());
  return MOD_OK;
}

static int check_sgm_params(ModContext *c, const ModSgmParams *p) {
  if (!p) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null SGM parameters");
  if (p->disparities < 1 || p->disparities > MOD_SGM_MAX_DISPARITIES) return fail(c, MOD_ERR_INVALID_ARGUMENT, "disparities must be in 1..128");
  if (p->p1 < 0 || p->p2 < p->p1 || 31 + p->p2 > 255) return fail(c, MOD_ERR_INVALID_ARGUMENT, "need 0 <= P1 <= P2 <= 224 (path costs are uint8)");
  if (p->paths != 4 && p->paths != 8) return fail(c, MOD_ERR_INVALID_ARGUMENT, "paths must be 4 or 8");
  if (c->dc.W < 2) return fail(c, MOD_ERR_INVALID_ARGUMENT, "the disparity estimator needs images at least 2 pixels wide");
  if ((size_t)c->dc.W * 8 + 4 > 64 * 1024) return fail(c, MOD_ERR_CAPACITY, "image row does not fit the census row buffer in LDS");
  return MOD_OK;
}

int mod_sgm_path_dev(ModContext *c, int32_t frames, const uint32_t *census_left, const uint32_t *census_right, const ModSgmParams *p,
                     int32_t direction, uint8_t *path_cost, uint8_t *matching_cost) {
  int rc = check_ready(c, frames);
  if (rc) return rc;
  if (!census_left || !census_right || !path_cost) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null plane");
  if ((rc = check_sgm_params(c, p))) return rc;
  if (direction < 0 || direction > 7) return fail(c, MOD_ERR_INVALID_ARGUMENT, "direction must be 0..7");
  // (stage entry point, tests and tracing) the D == 128 kernels read up to 127 words before the right plane: give them a padded copy
  const size_t words = (size_t)frames * c->dc.W * c->dc.H;
  uint32_t *padded = nullptr;
  hipError_t e = hipSuccess;
  if (p->disparities == 128) {
    e = hipMalloc((void **)&padded, (words + 128) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(padded, 0, 128 * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(padded + 128, census_right, words * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream);
  }
  if (e == hipSuccess) {
    launch_sgm_path(c->dc.W, c->dc.H, frames, p->disparities, p->p1, p->p2, direction, census_left, padded ? padded + 128 : census_right, path_cost,
                    matching_cost, padded != nullptr, c->stream);
    e = hipGetLastError();
  }
  if (padded) { (void)hipStreamSynchronize(c->stream); (void)hipFree(padded); }   // on every path, the failed ones included
  HIP_TRY(c, e);
  return MOD_OK;
}

// scratch of the complete estimator for a GROUP of frames (one wave walks a path line, so a single frame cannot fill the GPU; the
// frames of a group run side by side): per frame two census planes, one uint8 cost volume PER PATH (written once, never read
// back by the path kernels: a running sum would put its load latency into every step of a path), four disparity maps.  Census
// planes and volumes exist twice: consecutive groups overlap (mod_sgm_compute_dev).
constexpr int kSgmPaths = 8;
constexpr int kSgmGroup = 8;                             // frames per group (4 .. 16 measured in round 3: 8 is the knee)
constexpr size_t kSgmVolumeBudget = (size_t)24 << 30;    // bytes of cost volumes a context may hold

static int ensure_sgm_scratch(ModContext *c, int D, int frames, int *group) {
  Buffers &b = c->b;
  const size_t N = c->maxN;
  const int even = (frames + kSgmGroup - 1) / kSgmGroup;          // groups of equal size: 11 frames go as 6 + 5, not 8 + 3
  int g = (frames + even - 1) / even;
  while (g > 1 && 2 * (size_t)g * N * D * kSgmPaths > kSgmVolumeBudget) g--;
  *group = g;
  if (!b.sgm_fork[0]) {
    for (int k = 0; k < 2; k++) {
      HIP_TRY(c, hipEventCreateWithFlags(&b.sgm_fork[k], hipEventDisableTiming));
      for (int i = 0; i < 8; i++) HIP_TRY(c, hipEventCreateWithFlags(&b.sgm_join[k][i], hipEventDisableTiming));
    }
    // (a CU-masked path stream that kept one CU in 8 / 4 / 3 free for the winner-take-all of the group before was measured in
    // round 3 and changed nothing: plain non-blocking side streams)
    for (int i = 0; i < 8; i++) HIP_TRY(c, hipStreamCreateWithFlags(&b.sgm_side[i], hipStreamNonBlocking));
  }
  if (b.sgm_S && b.sgm_D >= D && b.sgm_G >= g) return MOD_OK;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  // grow-only in BOTH dimensions: calls that alternate between (few disparities, large group) and (many, small) settle on the
  // maxima after one reallocation each instead of freeing and allocating gigabytes on every call
  const int D2 = std::max(D, b.sgm_D), g2 = std::max(g, b.sgm_G);
  void *old[] = {b.sgm_S, b.sgm_census, b.sgm_maps};
  for (void *q : old) if (q) HIP_TRY(c, hipFree(q));
  b.sgm_S = nullptr; b.sgm_census = nullptr; b.sgm_maps = nullptr; b.sgm_D = 0; b.sgm_G = 0;
  // two sets (see mod_sgm_compute_dev) behind 128 words of lead: the D == 128 path kernels read up to 127 words to the left of a
  // right census plane unconditionally (discarded: disparities that do not exist) — inside the allocation even for tiny images
  HIP_TRY(c, dalloc(&b.sgm_census, 2 * 2 * N * g2 + 128));
  HIP_TRY(c, dalloc(&b.sgm_maps, 4 * N * g2));
  HIP_TRY(c, dalloc(&b.sgm_S, 2 * N * (size_t)D2 * g2 * kSgmPaths));
  b.sgm_D = D2; b.sgm_G = g2;
  return MOD_OK;
}

int mod_sgm_compute_dev(ModContext *c, int32_t frames, const uint8_t *left, const uint8_t *right, const ModSgmParams *p, float *disparity) {
  int rc = check_ready(c, frames);
  if (rc) return rc;
  if (!left || !right) return MOD_SKIP_NO_DISPARITY_NOW;     // no image pair: no disparity (estimateDisparity fails, :272-276)
  if (!disparity) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null disparity plane");
  if ((rc = check_sgm_params(c, p))) return rc;
  int group = 1;
  if ((rc = ensure_sgm_scratch(c, p->disparities, frames, &group))) return rc;
  const int W = c->dc.W, H = c->dc.H, D = p->disparities;
  const size_t N = (size_t)W * H;
  Buffers &b = c->b;
  static const int order4[4] = {0, 1, 2, 3};
  // Groups of frames go through two sets of census planes and cost volumes: while the winner-take-all of group k streams its
  // volumes (HBM-bound, context stream), the aggregation paths of group k + 1 (instruction-bound, one side stream per path) already
  // run.  Order on the context stream: census(0) fork(0) | census(1) fork(1) join(0) finish(0) | census(2) fork(2) join(1) finish(1) ...
  // — set s is written again (census(k + 2), paths(k + 2) behind fork(k + 2)) only after finish(k) has been enqueued before it.
  const int ngroups = (frames + group - 1) / group;
  const size_t set_census = 2 * N * group, set_volumes = N * (size_t)D * group * kSgmPaths;
  uint8_t *dl = b.sgm_maps, *dr = dl + N * group, *dlm = dr + N * group, *drm = dlm + N * group;
  bool all_in_one[2] = {false, false};
  auto start = [&](int k) -> int {
    const int f0 = k * group, g = std::min(group, frames - f0), s = k & 1;
    uint32_t *cl = b.sgm_census + 128 + s * set_census, *cr = cl + N * g;
    launch_sgm_census(W, H, g, left + (size_t)f0 * N, cl, c->stream);
    launch_sgm_census(W, H, g, right + (size_t)f0 * N, cr, c->stream);
    HIP_TRY(c, hipEventRecord(b.sgm_fork[s], c->stream));
    const size_t path_stride = N * (size_t)D * g;        // one volume [g][H][W][D] per path
    // the published configuration: all paths in ONE grid on one side stream (sgm.hip k_sgm_paths_all)
    HIP_TRY(c, hipStreamWaitEvent(b.sgm_side[0], b.sgm_fork[s], 0));
    if (launch_sgm_paths_all(W, H, g, D, p->p1, p->p2, p->paths, path_stride, cl, cr, b.sgm_S + s * set_volumes, b.sgm_side[0])) {
      HIP_TRY(c, hipEventRecord(b.sgm_join[s][0], b.sgm_side[0]));
      all_in_one[s] = true;
      return MOD_OK;
    }
    all_in_one[s] = false;
    for (int i = 0; i < p->paths; i++) {
      HIP_TRY(c, hipStreamWaitEvent(b.sgm_side[i], b.sgm_fork[s], 0));   // a failed wait would let a path read census planes in flight
      launch_sgm_path(W, H, g, D, p->p1, p->p2, p->paths == 4 ? order4[i] : i, cl, cr, b.sgm_S + s * set_volumes + (size_t)i * path_stride,
                      nullptr, /*right_plane_padded=*/true, b.sgm_side[i]);   // cr follows cl inside the scratch allocation
      HIP_TRY(c, hipEventRecord(b.sgm_join[s][i], b.sgm_side[i]));
    }
    return MOD_OK;
  };
  auto finish = [&](int k) -> int {
    const int f0 = k * group, g = std::min(group, frames - f0), s = k & 1;
    // a failed wait would let the winner-take-all read volumes the path kernels are still writing: surface it
    for (int i = 0; i < (all_in_one[s] ? 1 : p->paths); i++) HIP_TRY(c, hipStreamWaitEvent(c->stream, b.sgm_join[s][i], 0));
    launch_sgm_finish(W, H, g, D, p->paths, N * (size_t)D * g, p->median, p->lr_check, b.sgm_S + s * set_volumes, dl, dr, dlm, drm,
                      disparity + (size_t)f0 * N, c->stream);
    return MOD_OK;
  };
  if ((rc = start(0))) return rc;
  for (int k = 0; k < ngroups; k++) {
    if (k + 1 < ngroups && (rc = start(k + 1))) return rc;
    if ((rc = finish(k))) return rc;
  }
  HIP_TRY(c, hipGetLastError());
  return MOD_OK;
}

// ---- host-pointer convenience --------------------------------------------------------------------------------------
static int ensure_host_staging(ModContext *c);

int mod_sgm_compute_host(ModContext *c, const uint8_t *left, const uint8_t *right, const ModSgmParams *p, float *disparity) {
  int rc = check_ready(c, 1);
  if (rc) return rc;
  if (!left || !right) return MOD_SKIP_NO_DISPARITY_NOW;
  if (!disparity) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null disparity image");
  if ((rc = ensure_host_staging(c))) return rc;
  const size_t N = (size_t)c->dc.W * c->dc.H;
  Buffers &b = c->b;
  uint8_t *dimg = reinterpret_cast<uint8_t *>(b.h_flow);          // staging: the 8 N bytes of the flow slot hold both images
  HIP_TRY(c, hipMemcpyAsync(dimg, left, N, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(dimg + N, right, N, hipMemcpyHostToDevice, c->stream));
  if ((rc = mod_sgm_compute_dev(c, 1, dimg, dimg + N, p, b.h_dnow))) return rc;
  HIP_TRY(c, hipMemcpyAsync(disparity, b.h_dnow, 4 * N, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MOD_OK;
}

static int ensure_host_staging(ModContext *c) {
  Buffers &b = c->b;
  if (b.h_objects) return MOD_OK;                    // the last buffer of the set exists: all do
  const size_t N = c->maxN;
  HIP_TRY(c, dalloc(&b.h_dnow, N));
  HIP_TRY(c, dalloc(&b.h_dprev, N));
  HIP_TRY(c, dalloc(&b.h_flow, 2 * N));
  HIP_TRY(c, dalloc(&b.h_planes, 6 * N));
  if (!b.h_aos) HIP_TRY(c, hipMalloc(&b.h_aos, 32 * N));
  HIP_TRY(c, dalloc(&b.h_labels, N));
  HIP_TRY(c, dalloc(&b.h_nobj, 8));
  HIP_TRY(c, dalloc(&b.h_objects, (size_t)c->max_objects));
  return MOD_OK;
}

// xy: the x and y planes too (a caller's cloud unpacked for the clusterer); the fused host paths leave them out (scene_flow_staged)
static void staged_planes(ModContext *c, ModSceneFlowPlanes *pl, bool xy) {
  const size_t N = (size_t)c->dc.W * c->dc.H;
  float *p = c->b.h_planes;
  memset(pl, 0, sizeof(*pl));
  if (xy) { pl->x = p; pl->y = p + N; }
  pl->z = p + 2 * N; pl->vx = p + 3 * N; pl->vy = p + 4 * N; pl->vz = p + 5 * N;
}

static int fetch_cluster_results(ModContext *c, int32_t *labels, ModObject *objects, int32_t max_objects, int32_t *n_objects) {
  const size_t N = (size_t)c->dc.W * c->dc.H;
  int32_t n = 0;
  HIP_TRY(c, hipMemcpyAsync(&n, c->b.h_nobj, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (labels) HIP_TRY(c, hipMemcpyAsync(labels, c->b.h_labels, sizeof(int32_t) * N, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (n_objects) *n_objects = n;
  const int32_t ncopy = std::min(n, std::min(max_objects, (int32_t)c->max_objects));
  if (objects && ncopy > 0) {
    HIP_TRY(c, hipMemcpyAsync(objects, c->b.h_objects, sizeof(ModObject) * ncopy, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return MOD_OK;
}

int mod_process_frame_host(ModContext *c, const float *disparity_now, const float *disparity_prev, const float *flow,
                           const ModTransform *transform, double dt, void *cloud_aos, int32_t *labels, ModObject *objects,
                           int32_t max_objects, int32_t *n_objects) {
  int rc = check_ready(c, 1);
  if (rc) return rc;
  if (n_objects) *n_objects = 0;
  if (!flow) return MOD_SKIP_NO_FLOW;
  if (!disparity_prev) return MOD_SKIP_NO_DISPARITY_PREV;
  if (!transform) return MOD_SKIP_NO_TRANSFORM;
  if (!disparity_now) return MOD_SKIP_NO_DISPARITY_NOW;
  rc = ensure_host_staging(c);
  if (rc) return rc;
  const size_t N = (size_t)c->dc.W * c->dc.H;
  Buffers &b = c->b;
  HIP_TRY(c, hipMemcpyAsync(b.h_dnow, disparity_now, 4 * N, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(b.h_dprev, disparity_prev, 4 * N, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(b.h_flow, flow, 8 * N, hipMemcpyHostToDevice, c->stream));
  ModFrameBatch in{};
  in.frames = 1; in.disparity_now = b.h_dnow; in.disparity_prev = b.h_dprev; in.flow = b.h_flow;
  in.transforms = transform; in.dt = &dt;
  ModSceneFlowPlanes pl;
  staged_planes(c, &pl, false);
  pl.cloud_aos = cloud_aos ? b.h_aos : nullptr;
  ModClusterOut out{};
  out.labels = labels ? b.h_labels : nullptr; out.objects = b.h_objects; out.n_objects = b.h_nobj; out.n_clusters = b.h_nobj + 1;
  // no cluster output asked for (neither labels nor objects nor their count): the scene-flow stage alone — a constructor whose
  // moving objects nobody takes does not cluster (the reference's constructor never does; its clusterer is a node of its own)
  const bool cluster = labels || objects || n_objects;
  rc = cluster ? mod_process_dev(c, &in, &pl, &out) : scene_flow_staged(c, &in, &pl);
  if (rc) return rc;
  if (cloud_aos) HIP_TRY(c, hipMemcpyAsync(cloud_aos, b.h_aos, 32 * N, hipMemcpyDeviceToHost, c->stream));
  if (!cluster) { HIP_TRY(c, hipStreamSynchronize(c->stream)); return MOD_OK; }
  return fetch_cluster_results(c, labels, objects, max_objects, n_objects);
}

int mod_depth_image_host(ModContext *c, const float *disparity_now, float *depth) {
  int rc = check_ready(c, 1);
  if (rc) return rc;
  if (!disparity_now) return MOD_SKIP_NO_DISPARITY_NOW;
  if (!depth) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null depth image");
  if ((rc = ensure_host_staging(c))) return rc;
  const size_t N = (size_t)c->dc.W * c->dc.H;
  Buffers &b = c->b;
  HIP_TRY(c, hipMemcpyAsync(b.h_dnow, disparity_now, 4 * N, hipMemcpyHostToDevice, c->stream));
  launch_depth(c->dc, 1, b.h_dnow, b.h_planes, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(depth, b.h_planes, 4 * N, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MOD_OK;
}

int mod_static_flow_host(ModContext *c, const float *disparity_prev, const ModTransform *transform, float *static_flow) {
  int rc = check_ready(c, 1);
  if (rc) return rc;
  if (!disparity_prev) return MOD_SKIP_NO_DISPARITY_PREV;
  if (!transform) return MOD_SKIP_NO_TRANSFORM;
  if (!static_flow) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null static-flow image");
  if ((rc = ensure_host_staging(c))) return rc;
  const size_t N = (size_t)c->dc.W * c->dc.H;
  Buffers &b = c->b;
  HIP_TRY(c, hipMemcpyAsync(b.h_dprev, disparity_prev, 4 * N, hipMemcpyHostToDevice, c->stream));
  // the static flow depends on the previous disparity and the transform only (sceneflow.hip sf_stage1): the kernel's other
  // inputs are fed the same plane / a zeroed flow, and its cloud goes to the staging planes nobody reads
  HIP_TRY(c, hipMemsetAsync(b.h_flow, 0, 8 * N, c->stream));
  ModFrameBatch in{};
  const double dt = 1.0;
  in.frames = 1; in.disparity_now = b.h_dprev; in.disparity_prev = b.h_dprev; in.flow = b.h_flow; in.transforms = transform; in.dt = &dt;
  ModSceneFlowPlanes pl;
  staged_planes(c, &pl, false);
  pl.static_flow = reinterpret_cast<float *>(b.h_aos);      // 8 of the staging cloud's 32 bytes per pixel
  if ((rc = scene_flow_staged(c, &in, &pl))) return rc;
  HIP_TRY(c, hipMemcpyAsync(static_flow, b.h_aos, 8 * N, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MOD_OK;
}

int mod_cluster_cloud_host(ModContext *c, const void *cloud, int32_t width, int32_t height, int32_t point_step,
                           int32_t row_step, int32_t *labels, ModObject *objects, int32_t max_objects, int32_t *n_objects) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  int rc;
  if (n_objects) *n_objects = 0;
  if (!c->has_cam) {
    // A clusterer-only context (the nodelet lives in its own process, clusterer_nodelet.cpp:221-242): the clusterer reads the
    // image size from the cloud it is handed and needs nothing else of the camera — the context takes the size from the call.
    if (!c->has_prm) return fail(c, MOD_ERR_NOT_CONFIGURED, "parameters must be set first");
    if (width < 1 || height < 1) return fail(c, MOD_ERR_INVALID_ARGUMENT, "cloud size must be positive");
    if (width > c->cfg.max_width || height > c->cfg.max_height || (size_t)width * height > c->maxN)
      return fail(c, MOD_ERR_CAPACITY, "cloud larger than ModConfig.max_width/max_height");
    if (c->dc.W != width || c->dc.H != height) {
      c->cam = ModCamera{};
      c->cam.width = width; c->cam.height = height; c->cam.fx = c->cam.fy = 1.0;
      refresh_devcam(c);
    }
  } else if ((rc = check_ready(c, 1))) return rc;
  if (!cloud) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null cloud");
  // an unorganized / mis-sized cloud is an error (the reference would throw from .at(), clusterer_nodelet.h:99-102)
  if (width != c->dc.W || height != c->dc.H) return fail(c, MOD_ERR_INVALID_ARGUMENT, "cloud size differs from the configured camera");
  if (point_step != 32 || row_step < 32 * width) return fail(c, MOD_ERR_INVALID_ARGUMENT, "expected PointXYZVelocity records (point_step 32)");
  rc = ensure_host_staging(c);
  if (rc) return rc;
  Buffers &b = c->b;
  HIP_TRY(c, hipMemcpy2DAsync(b.h_aos, (size_t)32 * width, cloud, (size_t)row_step, (size_t)32 * width, (size_t)height,
                              hipMemcpyHostToDevice, c->stream));
  ModSceneFlowPlanes pl;
  staged_planes(c, &pl, true);
  launch_unpack((size_t)width * height, b.h_aos, pl.x, pl.y, pl.z, pl.vx, pl.vy, pl.vz, c->stream);
  HIP_TRY(c, hipGetLastError());
  ModClusterOut out{};
  out.labels = labels ? b.h_labels : nullptr; out.objects = b.h_objects; out.n_objects = b.h_nobj; out.n_clusters = b.h_nobj + 1;
  if ((rc = begin_cluster_scratch(c)) || (rc = run_cluster(c, 1, &pl, c->b.mask, false, false, &out))) return rc;
  c->scratch_clean = true;
  return fetch_cluster_results(c, labels, objects, max_objects, n_objects);
}

// ---- host streaming ----------------------------------------------------------------------------------------------------
static int ensure_pipe(ModContext *c) {
  ModContext::Pipe &p = c->pipe;
  if (p.ready) return MOD_OK;
  const size_t N = c->maxN;
  if (!p.h2d) HIP_TRY(c, hipStreamCreateWithFlags(&p.h2d, hipStreamNonBlocking));
  if (!p.d2h) HIP_TRY(c, hipStreamCreateWithFlags(&p.d2h, hipStreamNonBlocking));
  for (int i = 0; i <= MOD_PIPELINE_DEPTH; i++) HIP_TRY(c, dalloc(&p.dnow[i], N));
  for (int i = 0; i < MOD_PIPELINE_DEPTH; i++) {
    HIP_TRY(c, dalloc(&p.dprev[i], N));
    HIP_TRY(c, dalloc(&p.flow[i], 2 * N));
    HIP_TRY(c, dalloc(&p.planes[i], 4 * N));
    if (!p.aos[i]) HIP_TRY(c, hipMalloc(&p.aos[i], 32 * N));
    HIP_TRY(c, dalloc(&p.labels[i], N));
    HIP_TRY(c, dalloc(&p.nobj[i], 8));
    HIP_TRY(c, dalloc(&p.objects[i], (size_t)c->max_objects));
    if (!p.h_n[i]) HIP_TRY(c, hipHostMalloc((void **)&p.h_n[i], 64, hipHostMallocDefault));
    if (!p.h_obj[i]) HIP_TRY(c, hipHostMalloc((void **)&p.h_obj[i], sizeof(ModObject) * (size_t)c->max_objects, hipHostMallocDefault));
    if (!p.ev_in[i]) HIP_TRY(c, hipEventCreateWithFlags(&p.ev_in[i], hipEventDisableTiming));
    if (!p.ev_done[i]) HIP_TRY(c, hipEventCreateWithFlags(&p.ev_done[i], hipEventDisableTiming));
    if (!p.ev_out[i]) HIP_TRY(c, hipEventCreateWithFlags(&p.ev_out[i], hipEventDisableTiming));
    if (!p.ev_img[i]) HIP_TRY(c, hipEventCreateWithFlags(&p.ev_img[i], hipEventDisableTiming));
  }
  if (!p.ev_ring) HIP_TRY(c, hipEventCreateWithFlags(&p.ev_ring, hipEventDisableTiming));
  for (hipEvent_t &e : p.ev_plane_read) if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  p.ready = true;
  return MOD_OK;
}

int mod_submit_frame_host(ModContext *c, const float *disparity_now, const float *disparity_prev, const float *flow,
                          const ModTransform *transform, double dt, void *cloud_aos, int32_t *labels, ModObject *objects,
                          int32_t max_objects, int32_t *ticket) {
  int rc = check_ready(c, 1);
  if (rc) return rc;
  if (!ticket) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null ticket");
  *ticket = -1;
  ModContext::Pipe &p = c->pipe;
  // the guards of construct() (scene_flow_constructor.cpp:104,110,122,127,133), in its order
  if (!flow) return MOD_SKIP_NO_FLOW;
  if (!disparity_prev && !p.have_prev) return MOD_SKIP_NO_DISPARITY_PREV;
  if (!transform) return MOD_SKIP_NO_TRANSFORM;
  if (!disparity_now) return MOD_SKIP_NO_DISPARITY_NOW;
  if (p.in_flight >= MOD_PIPELINE_DEPTH) return fail(c, MOD_ERR_CAPACITY, "MOD_PIPELINE_DEPTH frames are already in flight");
  if ((rc = ensure_pipe(c))) return rc;
  constexpr int R = MOD_PIPELINE_DEPTH + 1;
  const int slot = (int)(p.seq % MOD_PIPELINE_DEPTH), nowi = (int)(p.dring % R), previ = (int)((p.dring + R - 1) % R);
  const size_t N = (size_t)c->dc.W * c->dc.H;
  // inputs: their own stream.  dnow[nowi] was last read by the frame R - 1 planes ago (as its "previous"), which has been collected:
  // at most MOD_PIPELINE_DEPTH - 1 frames are in flight at this point.  (Planes the stereo entry filled were written by kernels,
  // and a frame it skipped took a plane without a ticket: the copy then also waits for the last of those kernels.)
  if (p.ring_by_kernels) { HIP_TRY(c, hipStreamWaitEvent(p.h2d, p.ev_ring, 0)); p.ring_by_kernels = false; }
  if (p.plane_read_pending[nowi]) { HIP_TRY(c, hipStreamWaitEvent(p.h2d, p.ev_plane_read[nowi], 0)); p.plane_read_pending[nowi] = false; }
  HIP_TRY(c, hipMemcpyAsync(p.dnow[nowi], disparity_now, 4 * N, hipMemcpyHostToDevice, p.h2d));
  if (disparity_prev) HIP_TRY(c, hipMemcpyAsync(p.dprev[slot], disparity_prev, 4 * N, hipMemcpyHostToDevice, p.h2d));
  HIP_TRY(c, hipMemcpyAsync(p.flow[slot], flow, 8 * N, hipMemcpyHostToDevice, p.h2d));
  HIP_TRY(c, hipEventRecord(p.ev_in[slot], p.h2d));
  // kernels: the context's stream
  HIP_TRY(c, hipStreamWaitEvent(c->stream, p.ev_in[slot], 0));
  ModFrameBatch in{};
  in.frames = 1; in.disparity_now = p.dnow[nowi]; in.disparity_prev = disparity_prev ? p.dprev[slot] : p.dnow[previ];
  in.flow = p.flow[slot]; in.transforms = transform; in.dt = &dt;
  ModSceneFlowPlanes pl;
  memset(&pl, 0, sizeof(pl));
  float *q = p.planes[slot];                 // z, vx, vy, vz for the cluster stage; no x, y planes (see scene_flow_staged)
  pl.z = q; pl.vx = q + N; pl.vy = q + 2 * N; pl.vz = q + 3 * N;
  pl.cloud_aos = cloud_aos ? p.aos[slot] : nullptr;
  ModClusterOut out{};
  out.labels = labels ? p.labels[slot] : nullptr; out.objects = p.objects[slot]; out.n_objects = p.nobj[slot]; out.n_clusters = p.nobj[slot] + 1;
  const bool cluster = labels || objects;     // neither asked for: the scene-flow stage alone (see mod_process_frame_host)
  if ((rc = cluster ? mod_process_dev(c, &in, &pl, &out) : scene_flow_staged(c, &in, &pl))) return rc;
  HIP_TRY(c, hipEventRecord(p.ev_done[slot], c->stream));
  // results: their own stream
  HIP_TRY(c, hipStreamWaitEvent(p.d2h, p.ev_done[slot], 0));
  if (cluster) HIP_TRY(c, hipMemcpyAsync(p.h_n[slot], p.nobj[slot], sizeof(int32_t), hipMemcpyDeviceToHost, p.d2h));
  else *p.h_n[slot] = 0;
  if (labels) HIP_TRY(c, hipMemcpyAsync(labels, p.labels[slot], sizeof(int32_t) * N, hipMemcpyDeviceToHost, p.d2h));
  // the count is not known yet: the caller's capacity goes to a pinned staging array (a pageable destination would make this
  // call wait for the kernels); mod_collect_frame_host hands the objects over
  const int32_t ncopy = objects ? std::max(0, std::min(max_objects, (int32_t)c->max_objects)) : 0;
  if (ncopy > 0) HIP_TRY(c, hipMemcpyAsync(p.h_obj[slot], p.objects[slot], sizeof(ModObject) * ncopy, hipMemcpyDeviceToHost, p.d2h));
  p.user_obj[slot] = objects; p.user_cap[slot] = ncopy;
  if (cloud_aos) HIP_TRY(c, hipMemcpyAsync(cloud_aos, p.aos[slot], 32 * N, hipMemcpyDeviceToHost, p.d2h));
  HIP_TRY(c, hipEventRecord(p.ev_out[slot], p.d2h));
  *ticket = (int32_t)(p.seq & 0x7fffffff);
  p.seq++; p.dring++; p.in_flight++; p.have_prev = true;
  return MOD_OK;
}

int mod_submit_stereo_host(ModContext *c, const uint8_t *left, const uint8_t *right, const ModSgmParams *sgm, const float *flow,
                           const ModTransform *transform, double dt, void *cloud_aos, int32_t *labels, ModObject *objects,
                           int32_t max_objects, float *disparity, int32_t *ticket) {
  int rc = check_ready(c, 1);
  if (rc) return rc;
  if (!ticket) return fail(c, MOD_ERR_INVALID_ARGUMENT, "null ticket");
  *ticket = -1;
  ModContext::Pipe &p = c->pipe;
  if (!left || !right) {            // estimateDisparity() has nothing to work on: disparity_now_.reset() (scene_flow_constructor.cpp:272-276)
    p.have_prev = false;            // ... which becomes the next frame's (missing) previous disparity (:397-398)
    return MOD_SKIP_NO_DISPARITY_NOW;
  }
  if ((rc = check_sgm_params(c, sgm))) return rc;
  if (p.in_flight >= MOD_PIPELINE_DEPTH) return fail(c, MOD_ERR_CAPACITY, "MOD_PIPELINE_DEPTH frames are already in flight");
  if ((rc = ensure_pipe(c))) return rc;
  constexpr int R = MOD_PIPELINE_DEPTH + 1;
  const int slot = (int)(p.seq % MOD_PIPELINE_DEPTH), nowi = (int)(p.dring % R), previ = (int)((p.dring + R - 1) % R);
  const size_t N = (size_t)c->dc.W * c->dc.H;
  if (!p.img[slot]) HIP_TRY(c, dalloc(&p.img[slot], 2 * c->maxN));
  // images (and flow) on the copy stream; the slot's image buffer may still be read by the estimator of a frame that ended at a
  // guard (it took no ticket, so nobody waited for it): the copy queues behind that estimator
  if (p.img_used[slot]) HIP_TRY(c, hipStreamWaitEvent(p.h2d, p.ev_img[slot], 0));
  HIP_TRY(c, hipMemcpyAsync(p.img[slot], left, N, hipMemcpyHostToDevice, p.h2d));
  HIP_TRY(c, hipMemcpyAsync(p.img[slot] + N, right, N, hipMemcpyHostToDevice, p.h2d));
  if (flow) HIP_TRY(c, hipMemcpyAsync(p.flow[slot], flow, 8 * N, hipMemcpyHostToDevice, p.h2d));
  HIP_TRY(c, hipEventRecord(p.ev_in[slot], p.h2d));
  HIP_TRY(c, hipStreamWaitEvent(c->stream, p.ev_in[slot], 0));
  // estimateDisparity (:258-279) on the GPU, straight into the ring: this plane is `now` here and `previous` of the next frame.
  // Kernels of older frames that read the plane being replaced are ahead of the estimator on the same stream.
  if (p.plane_read_pending[nowi]) { HIP_TRY(c, hipStreamWaitEvent(c->stream, p.ev_plane_read[nowi], 0)); p.plane_read_pending[nowi] = false; }
  if ((rc = mod_sgm_compute_dev(c, 1, p.img[slot], p.img[slot] + N, sgm, p.dnow[nowi]))) return rc;
  HIP_TRY(c, hipEventRecord(p.ev_img[slot], c->stream));
  HIP_TRY(c, hipEventRecord(p.ev_ring, c->stream));
  p.img_used[slot] = true; p.ring_by_kernels = true;
  const bool had_prev = p.have_prev;
  p.dring++; p.have_prev = true;    // disparity_previous_ = disparity_now_, whatever construct() does with the frame (:397-398)
  // the guards of construct() (:104,110,122,127,133), in its order; disparity_now exists by now
  if (!flow) return MOD_SKIP_NO_FLOW;
  if (!had_prev) return MOD_SKIP_NO_DISPARITY_PREV;
  if (!transform) return MOD_SKIP_NO_TRANSFORM;
  ModFrameBatch in{};
  in.frames = 1; in.disparity_now = p.dnow[nowi]; in.disparity_prev = p.dnow[previ];
  in.flow = p.flow[slot]; in.transforms = transform; in.dt = &dt;
  ModSceneFlowPlanes pl;
  memset(&pl, 0, sizeof(pl));
  float *q = p.planes[slot];                 // z, vx, vy, vz for the cluster stage; no x, y planes (see scene_flow_staged)
  pl.z = q; pl.vx = q + N; pl.vy = q + 2 * N; pl.vz = q + 3 * N;
  pl.cloud_aos = cloud_aos ? p.aos[slot] : nullptr;
  ModClusterOut out{};
  out.labels = labels ? p.labels[slot] : nullptr; out.objects = p.objects[slot]; out.n_objects = p.nobj[slot]; out.n_clusters = p.nobj[slot] + 1;
  const bool cluster = labels || objects;     // neither asked for: the scene-flow stage alone (see mod_process_frame_host)
  if ((rc = cluster ? mod_process_dev(c, &in, &pl, &out) : scene_flow_staged(c, &in, &pl))) return rc;
  HIP_TRY(c, hipEventRecord(p.ev_done[slot], c->stream));
  HIP_TRY(c, hipStreamWaitEvent(p.d2h, p.ev_done[slot], 0));
  if (cluster) HIP_TRY(c, hipMemcpyAsync(p.h_n[slot], p.nobj[slot], sizeof(int32_t), hipMemcpyDeviceToHost, p.d2h));
  else *p.h_n[slot] = 0;
  if (labels) HIP_TRY(c, hipMemcpyAsync(labels, p.labels[slot], sizeof(int32_t) * N, hipMemcpyDeviceToHost, p.d2h));
  if (disparity) {
    HIP_TRY(c, hipMemcpyAsync(disparity, p.dnow[nowi], sizeof(float) * N, hipMemcpyDeviceToHost, p.d2h));
    HIP_TRY(c, hipEventRecord(p.ev_plane_read[nowi], p.d2h));
    p.plane_read_pending[nowi] = true;
  }
  const int32_t ncopy = objects ? std::max(0, std::min(max_objects, (int32_t)c->max_objects)) : 0;
  if (ncopy > 0) HIP_TRY(c, hipMemcpyAsync(p.h_obj[slot], p.objects[slot], sizeof(ModObject) * ncopy, hipMemcpyDeviceToHost, p.d2h));
  p.user_obj[slot] = objects; p.user_cap[slot] = ncopy;
  if (cloud_aos) HIP_TRY(c, hipMemcpyAsync(cloud_aos, p.aos[slot], 32 * N, hipMemcpyDeviceToHost, p.d2h));
  HIP_TRY(c, hipEventRecord(p.ev_out[slot], p.d2h));
  *ticket = (int32_t)(p.seq & 0x7fffffff);
  p.seq++; p.in_flight++;
  return MOD_OK;
}

int mod_collect_frame_host(ModContext *c, int32_t ticket, int32_t *n_objects) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  ModContext::Pipe &p = c->pipe;
  if (n_objects) *n_objects = 0;
  if (p.in_flight < 1) return fail(c, MOD_ERR_INVALID_ARGUMENT, "no frame in flight");
  const int64_t oldest = p.seq - p.in_flight;
  if (ticket != (int32_t)(oldest & 0x7fffffff)) return fail(c, MOD_ERR_INVALID_ARGUMENT, "tickets are collected in submission order");
  const int slot = (int)(oldest % MOD_PIPELINE_DEPTH);
  HIP_TRY(c, hipEventSynchronize(p.ev_out[slot]));
  const int32_t n = *p.h_n[slot];
  if (n_objects) *n_objects = n;
  const int32_t ncopy = std::min(n, p.user_cap[slot]);
  if (p.user_obj[slot] && ncopy > 0) memcpy(p.user_obj[slot], p.h_obj[slot], sizeof(ModObject) * (size_t)ncopy);
  p.in_flight--;
  return MOD_OK;
}

int mod_forget_previous(ModContext *c) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  c->pipe.have_prev = false;
  return MOD_OK;
}

int mod_host_malloc(ModContext *c, uint64_t bytes, void **p) {
  if (!c || !p) return MOD_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipHostMalloc(p, bytes, hipHostMallocDefault));
  return MOD_OK;
}
int mod_host_free(ModContext *c, void *p) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipHostFree(p));
  return MOD_OK;
}

// ---- memory helpers --------------------------------------------------------------------------------------------------
int mod_malloc(ModContext *c, uint64_t bytes, void **p) {
  if (!c || !p) return MOD_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipMalloc(p, bytes));
  return MOD_OK;
}
int mod_free(ModContext *c, void *p) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipFree(p));
  return MOD_OK;
}
int mod_memcpy_h2d(ModContext *c, void *d, const void *h, uint64_t bytes) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MOD_OK;
}
int mod_memcpy_d2h(ModContext *c, void *h, const void *d, uint64_t bytes) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MOD_OK;
}

#if defined(MOD_PHASE_COUNTERS) || defined(MOD_ABLATION) || defined(MOD_CHECKED)
// diagnostic builds only (declared in mod_sf_debug.h; a product build does not export them)
// copy an internal buffer to the host (0 member norms, 4 member pixels, 1 clusters, 2 counters)
int mod_debug_read(ModContext *c, int which, void *dst, unsigned long long bytes) {
  if (!c || !dst) return MOD_ERR_INVALID_ARGUMENT;
  const void *src = which == 0 ? (const void *)c->b.mbits : which == 4 ? (const void *)c->b.mpix : which == 1 ? (const void *)c->b.clusters
                  : (const void *)c->b.counters;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return MOD_OK;
}

int mod_debug_counters(ModContext *c, unsigned long long *out32) {
  if (!c || !out32) return MOD_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(out32, c->b.dbg, kDbgWords * 8, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemset(c->b.dbg, 0, kDbgWords * 8));
  HIP_TRY(c, hipMemset((char *)c->b.dbg + 42 * 8, 0xFF, 8));
  return MOD_OK;
}
#endif

// ---- measurement -------------------------------------------------------------------------------------------------------
int mod_set_profiling(ModContext *c, int32_t stage_mask) {
  if (!c || (stage_mask & ~MOD_PROFILE_ALL)) return MOD_ERR_INVALID_ARGUMENT;
  c->profiling = stage_mask;
  // events for the next 64 calls are created here, not inside the calls that are being timed (an event pair costs tens of
  // microseconds to create; later calls create what they lack)
  const size_t want = (size_t)64 * (size_t)__builtin_popcount((unsigned)stage_mask);
  while (c->free_events.size() < want) {
    EventPair ev{};
    if (hipEventCreate(&ev.a) != hipSuccess) break;
    if (hipEventCreate(&ev.b) != hipSuccess) { (void)hipEventDestroy(ev.a); break; }
    c->free_events.push_back(ev);
  }
  return MOD_OK;
}
int mod_get_stage_time(ModContext *c, int32_t stage, double *total_ms, int64_t *calls) {
  if (!c || stage < 0 || stage >= MOD_STAGE_COUNT) return MOD_ERR_INVALID_ARGUMENT;
  drain_timers(c);
  if (total_ms) *total_ms = c->stage_ms[stage];
  if (calls) *calls = c->stage_calls[stage];
  return MOD_OK;
}
int mod_reset_stage_times(ModContext *c) {
  if (!c) return MOD_ERR_INVALID_ARGUMENT;
  drain_timers(c);
  for (int s = 0; s < MOD_STAGE_COUNT; s++) { c->stage_ms[s] = 0; c->stage_calls[s] = 0; }
  return MOD_OK;
}

}  // extern "C"
