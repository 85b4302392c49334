// oracle.cpp — CPU restatement of the reference's scene-flow + clustering hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library, and only as the checker / the timed CPU baseline.
//
// PARITY PIN STATUS: the reference ships no tests, goldens or fixtures for this path and cannot be built here
// (needs ROS melodic, PCL, OpenCV, Eigen, tf2 — none installed), so this restatement is "parity unpinned" except
// for the union-find LookupTable, which is checked against the reference's own lookup_table.cpp compiled into
// oracle/_ref (see oracle/Makefile, tests/test_oracle_lookup_ref.py).  A second, independently written numpy
// restatement (oracle/numpy_ref.py) cross-checks everything else.
//
// Each function cites the reference file:line it follows (paths relative to the reference root).  Arithmetic
// precision is annotated F32/F64 exactly as the reference expression evaluates on x86-64/SSE2
// (FLT_EVAL_METHOD == 0, no FMA contraction: build with -ffp-contract=off and no -march flags).
//
// Two execution modes with identical results:
//   faithful — the reference's data layouts and loop orders (16-byte / 32-byte AoS clouds, column-major walks,
//              per-frame heap allocation, NaN pre-fill, std::vector<bool> dynamic map);
//   tidy     — row-major SoA planes, caller-provided workspace, no allocation.
// Both are timed by bench.py so that the GPU/CPU ratio is not inflated by the reference's incidental inefficiencies.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

extern "C" {

struct OrcCamera {            // mirrors ModCamera (include/mod_sf.h)
  int32_t width, height;
  double fx, fy, cx, cy, Tx, Ty;
  float disp_f, disp_T, min_disparity, max_disparity;
};

struct OrcParams {            // mirrors ModParams
  int32_t dynamic_flow_diff;
  int32_t cluster_size;
  int32_t neighbor_distance;
  int32_t reserved;
  double depth_diff;
  double dynamic_speed;
};

struct OrcTransform {         // geometry_msgs/Transform: translation, quaternion x y z w
  double t[3];
  double q[4];
};

struct OrcObject {            // mirrors ModObject
  int32_t id;
  int32_t n_points;
  double center[3];
  double orientation[4];
  double velocity[3];
  double bounding_box[3];
};

}  // extern "C"

namespace {

struct Pt3 {  // pcl::PointXYZ: 16 bytes, 4th float is padding
  float x, y, z, pad;
};
struct PtV {  // pcl::PointXYZVelocity (scene_flow_constructor/pcl_point_xyz_velocity.h:8-34): 32 bytes
  float x, y, z, pad0;
  float vx, vy, vz, pad1;
};
static_assert(sizeof(Pt3) == 16 && sizeof(PtV) == 32, "reference point sizes");

const float kNaN = std::numeric_limits<float>::quiet_NaN();

// disparity_image_processor.cpp:17-31 — bounds, then range gate.  NaN fails neither comparison and passes.
inline bool get_disparity(const OrcCamera &c, const float *D, int u, int v, float &d) {
  if (u < 0 || u >= c.width) return false;
  if (v < 0 || v >= c.height) return false;
  d = D[(size_t)v * c.width + u];
  if (c.max_disparity < d) return false;
  if (c.min_disparity > d) return false;
  return true;
}

// disparity_image_processor.cpp:33-50 with image_geometry::PinholeCameraModel::projectPixelTo3dRay
// (vision_opencv melodic): ray = ((u - cx - Tx)/fx, (v - cy - Ty)/fy, 1) in F64.
inline bool get_point3d(const OrcCamera &c, const float *D, int u, int v, Pt3 &p) {
  float d;
  if (!get_disparity(c, D, u, v, d)) return false;
  if (d == 0.0) return false;
  float z = c.disp_f * c.disp_T / d;                       // F32: F(F(f*T)/d)
  double rx = ((double)u - c.cx - c.Tx) / c.fx;            // F64
  double ry = ((double)v - c.cy - c.Ty) / c.fy;            // F64
  p.z = z;
  p.x = (float)(rx * (double)z);                           // F64 product, rounded to F32
  p.y = (float)(ry * (double)z);
  return true;
}

// Eigen::Quaterniond::toRotationMatrix (Eigen 3.3, un-vendored) as used by tf2::transformToEigen
// (scene_flow_constructor.cpp:411).  The quaternion is not renormalised.
struct Iso {
  double r[3][3];
  double t[3];
};
inline Iso make_iso(const OrcTransform &tf) {
  const double x = tf.q[0], y = tf.q[1], z = tf.q[2], w = tf.q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Iso m;
  m.r[0][0] = 1.0 - (tyy + tzz); m.r[0][1] = txy - twz;         m.r[0][2] = txz + twy;
  m.r[1][0] = txy + twz;         m.r[1][1] = 1.0 - (txx + tzz); m.r[1][2] = tyz - twx;
  m.r[2][0] = txz - twy;         m.r[2][1] = tyz + twx;         m.r[2][2] = 1.0 - (txx + tyy);
  m.t[0] = tf.t[0]; m.t[1] = tf.t[1]; m.t[2] = tf.t[2];
  return m;
}

// Eigen::Isometry3d * Vector3d (scene_flow_constructor.cpp:425): res = translation; res += linear * v, each
// coefficient of the 3x3 * 3x1 lazy product reduced as p0 + (p1 + p2) (Eigen 3.3 redux_novec_unroller, length 3).
// Eigen is un-vendored: the association only matters below one F64 ulp and is recorded as a parity hazard in DESIGN.md.
inline void iso_apply(const Iso &m, double x, double y, double z, double out[3]) {
  for (int i = 0; i < 3; i++) {
    double p0 = m.r[i][0] * x, p1 = m.r[i][1] * y, p2 = m.r[i][2] * z;
    out[i] = m.t[i] + (p0 + (p1 + p2));
  }
}

// scene_flow_constructor.h:240-249 — tests x only.
inline bool is_valid(const Pt3 &p) { return !std::isnan(p.x) && !std::isinf(p.x); }

// std::round on F32 then int conversion (scene_flow_constructor.h:210-211).  Out-of-int-range values are UB in the
// reference; on x86-64 cvttss2si yields INT_MIN which the bounds check then rejects.  Defined here as "reject".
inline bool round_to_int(float r, int &o) {
  float q = roundf(r);
  if (!(q >= -2147483648.0f && q < 2147483648.0f)) return false;
  o = (int)q;
  return true;
}

// scene_flow_constructor.h:215-227 (the right pixel it also computes is never used).
inline bool right_point_ok(const OrcCamera &c, const float *D, int u, int v) {
  float d;
  if (!get_disparity(c, D, u, v, d)) return false;
  if (std::isnan(d) || std::isinf(d) || d < 0) return false;
  return true;
}

// Eigen::Vector3f::norm(): sqrt of x^2 + (y^2 + z^2) in F32 (Eigen 3.3 redux_novec_unroller, length 3).
inline float norm3(float x, float y, float z) {
  float xx = x * x, yy = y * y, zz = z * z;
  float s = yy + zz;
  s = xx + s;
  return sqrtf(s);
}

// ---- union-find restating scene_flow_clusterer/include/lookup_table.h:10-33 ---------------------------------
struct Lut {
  std::vector<int> table;
  int max_label = -1;
  void resize(size_t n) { table.resize(n); }
  void reset() { max_label = -1; }
  int add_label() { ++max_label; table.at(max_label) = max_label; return max_label; }
  int lookup(int s) {
    while (s != table.at(s)) { table.at(s) = table.at(table.at(s)); s = table.at(s); }
    return s;
  }
  void link(int a, int b) {
    int ra = lookup(a), rb = lookup(b);
    if (ra > rb) table.at(ra) = rb; else table.at(rb) = ra;
  }
};

// ---- scene flow, faithful mode --------------------------------------------------------------------------------

// disparity_image_processor.cpp:86-103 — NaN-filled organized cloud, column-major walk.
void to_point_cloud(const OrcCamera &c, const float *D, std::vector<Pt3> &pc) {
  Pt3 def{kNaN, kNaN, kNaN, 1.0f};
  pc.assign((size_t)c.width * c.height, def);
  for (int u = 0; u < c.width; u++)
    for (int v = 0; v < c.height; v++) {
      Pt3 p{0, 0, 0, 1.0f};
      if (get_point3d(c, D, u, v, p)) pc[(size_t)v * c.width + u] = p;
    }
}

// scene_flow_constructor.cpp:409-429 — column-major walk; NaN x copies through.
void transform_prev(const OrcCamera &c, const std::vector<Pt3> &in, const OrcTransform &tf, std::vector<Pt3> &out) {
  Iso m = make_iso(tf);
  out.assign((size_t)c.width * c.height, Pt3{0, 0, 0, 1.0f});
  for (int u = 0; u < c.width; u++)
    for (int v = 0; v < c.height; v++) {
      const Pt3 &p = in[(size_t)v * c.width + u];
      if (std::isnan(p.x)) { out[(size_t)v * c.width + u] = p; continue; }
      double q[3];
      iso_apply(m, p.x, p.y, p.z, q);
      out[(size_t)v * c.width + u] = Pt3{(float)q[0], (float)q[1], (float)q[2], 1.0f};
    }
}

// scene_flow_constructor.cpp:65-89 with PinholeCameraModel::project3dToPixel:
// (u,v) = ((fx*X + Tx)/Z + cx, (fy*Y + Ty)/Z + cy) in F64; flow = F32(u - x), F32(v - y).
void static_flow(const OrcCamera &c, const std::vector<Pt3> &prevT, std::vector<float> &sf) {
  sf.resize((size_t)c.width * c.height * 2);
  for (int y = 0; y < c.height; y++)
    for (int x = 0; x < c.width; x++) {
      const Pt3 &p = prevT[(size_t)y * c.width + x];
      float *o = &sf[((size_t)y * c.width + x) * 2];
      if (std::isnan(p.x)) { o[0] = kNaN; o[1] = kNaN; continue; }
      double X = p.x, Y = p.y, Z = p.z;
      double u = (c.fx * X + c.Tx) / Z + c.cx;
      double v = (c.fy * Y + c.Ty) / Z + c.cy;
      o[0] = (float)(u - x);
      o[1] = (float)(v - y);
    }
}

// scene_flow_constructor.cpp:149-212 (+ :293-303 NaN initialisation, + header helpers :173-249).
void construct_velocity_pc(const OrcCamera &c, const OrcParams &prm, const std::vector<Pt3> &pc_now,
                           const std::vector<Pt3> &pc_prevT, const float *flow, const std::vector<float> &sflow,
                           const float *D_now, const float *D_prev, double dt, PtV *out) {
  const int W = c.width, H = c.height;
  PtV def{kNaN, kNaN, kNaN, 0.0f, kNaN, kNaN, kNaN, 0.0f};  // pads are uninitialised in the reference
  for (size_t i = 0; i < (size_t)W * H; i++) out[i] = def;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      PtV &o = out[(size_t)y * W + x];
      const Pt3 pn = pc_now[(size_t)y * W + x];
      if (!is_valid(pn)) continue;
      o.x = pn.x; o.y = pn.y; o.z = pn.z;
      // getMatchPoints: getPreviousPoint
      const float *fl = &flow[((size_t)y * W + x) * 2];
      if (std::isnan(fl[0]) || std::isnan(fl[1])) continue;
      int px = 0, py = 0;
      bool okx = round_to_int((float)x - fl[0], px);
      bool oky = round_to_int((float)y - fl[1], py);
      // getRightPoint(now), getRightPoint(previous)
      if (!right_point_ok(c, D_now, x, y)) continue;
      if (!okx || !oky) continue;  // out-of-int-range warp == out of image
      if (!right_point_ok(c, D_prev, px, py)) continue;
      const Pt3 pp = pc_prevT[(size_t)py * W + px];
      if (!is_valid(pp)) continue;
      const float *s = &sflow[((size_t)y * W + x) * 2];
      if (std::isnan(s[0])) continue;
      float r0 = fl[0] - s[0], r1 = fl[1] - s[1];       // cv::Vec2f operator-
      float acc = 0.0f;                                 // cv::Matx::dot: s = 0; s += a_i*b_i
      acc = acc + r0 * r0;
      acc = acc + r1 * r1;
      if (sqrtf(acc) >= (float)prm.dynamic_flow_diff) {
        o.vx = (float)((double)(pn.x - pp.x) / dt);
        o.vy = (float)((double)(pn.y - pp.y) / dt);
        o.vz = (float)((double)(pn.z - pp.z) / dt);
      } else {
        o.vx = 0.0f; o.vy = 0.0f; o.vz = 0.0f;
      }
    }
}

// ---- scene flow, tidy mode: one fused per-pixel function (Appendix A of SURVEY.md) -----------------------------
inline void prev_transformed(const OrcCamera &c, const Iso &m, const float *D_prev, int x, int y, Pt3 &p) {
  p = Pt3{kNaN, kNaN, kNaN, 1.0f};
  Pt3 q{0, 0, 0, 1.0f};
  if (!get_point3d(c, D_prev, x, y, q)) return;
  if (std::isnan(q.x)) { p = q; return; }
  double r[3];
  iso_apply(m, q.x, q.y, q.z, r);
  p = Pt3{(float)r[0], (float)r[1], (float)r[2], 1.0f};
}

void scene_flow_tidy(const OrcCamera &c, const OrcParams &prm, const float *D_now, const float *D_prev,
                     const float *flow, const OrcTransform &tf, double dt, float *X, float *Y, float *Z, float *VX,
                     float *VY, float *VZ, float *sflow_out) {
  const int W = c.width, H = c.height;
  const Iso m = make_iso(tf);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const size_t i = (size_t)y * W + x;
      float ox = kNaN, oy = kNaN, oz = kNaN, vx = kNaN, vy = kNaN, vz = kNaN;
      // static flow at the own pixel (needed for the optional output even when the point is invalid)
      Pt3 own;
      prev_transformed(c, m, D_prev, x, y, own);
      float s0 = kNaN, s1 = kNaN;
      if (!std::isnan(own.x)) {
        double u = (c.fx * (double)own.x + c.Tx) / (double)own.z + c.cx;
        double v = (c.fy * (double)own.y + c.Ty) / (double)own.z + c.cy;
        s0 = (float)(u - x);
        s1 = (float)(v - y);
      }
      if (sflow_out) { sflow_out[2 * i] = s0; sflow_out[2 * i + 1] = s1; }
      Pt3 pn{kNaN, kNaN, kNaN, 1.0f}, t{0, 0, 0, 1.0f};
      if (get_point3d(c, D_now, x, y, t)) pn = t;
      do {
        if (!is_valid(pn)) break;
        ox = pn.x; oy = pn.y; oz = pn.z;
        const float f0 = flow[2 * i], f1 = flow[2 * i + 1];
        if (std::isnan(f0) || std::isnan(f1)) break;
        int px = 0, py = 0;
        bool okx = round_to_int((float)x - f0, px), oky = round_to_int((float)y - f1, py);
        if (!right_point_ok(c, D_now, x, y)) break;
        if (!okx || !oky) break;
        if (!right_point_ok(c, D_prev, px, py)) break;
        Pt3 pp;
        prev_transformed(c, m, D_prev, px, py, pp);
        if (!is_valid(pp)) break;
        if (std::isnan(s0)) break;
        float r0 = f0 - s0, r1 = f1 - s1;
        float acc = 0.0f;
        acc = acc + r0 * r0;
        acc = acc + r1 * r1;
        if (sqrtf(acc) >= (float)prm.dynamic_flow_diff) {
          vx = (float)((double)(pn.x - pp.x) / dt);
          vy = (float)((double)(pn.y - pp.y) / dt);
          vz = (float)((double)(pn.z - pp.z) / dt);
        } else {
          vx = vy = vz = 0.0f;
        }
      } while (0);
      X[i] = ox; Y[i] = oy; Z[i] = oz; VX[i] = vx; VY[i] = vy; VZ[i] = vz;
    }
}

// ---- clusterer ------------------------------------------------------------------------------------------------

struct ClusterWork {  // tidy-mode workspace, reused across frames
  std::vector<uint8_t> dyn;
  std::vector<int> cmap;
  Lut lut;
  std::vector<size_t> csize;
  std::vector<int> old2new;
  std::vector<int> order;      // member pixel indices, grouped per cluster, column-major within a cluster
  std::vector<int> offsets;
  std::vector<std::pair<float, int>> keyed;
};

template <class GetZ, class IsDyn>
inline void scan_edges(int W, int H, int n, double depth_diff, GetZ getz, IsDyn isdyn, std::vector<int> &cmap, Lut &lut) {
  // calculateInitialClusterMap + comparePoints (clusterer_nodelet.cpp:56-83,186-219)
  for (int v = 0; v < H; v++)
    for (int u = 0; u < W; u++) {
      if (!isdyn(u, v)) continue;
      for (int dv = -n; dv <= 0; dv++)
        for (int du = -n; du <= 0; du++) {
          if (dv == 0 && du == 0) continue;
          int cu = u + du, cv = v + dv;
          if (cu < 0 || cu >= W || cv < 0 || cv >= H) continue;            // isInRange (clusterer_nodelet.h:93-98)
          if (!isdyn(cu, cv)) continue;
          float dd = std::abs(getz(u, v) - getz(cu, cv));                  // depthDiff (clusterer_nodelet.h:83-86), F32
          if ((double)dd > depth_diff) continue;                           // F64 compare; NaN links
          int &a = cmap[(size_t)v * W + u];
          int &b = cmap[(size_t)cv * W + cu];
          if (a == -1 && b == -1) { int l = lut.add_label(); a = l; b = l; }
          else if (a != -1 && b == -1) b = a;
          else if (a == -1 && b != -1) a = b;
          else if (a != b) lut.link(a, b);
        }
    }
}

// integrateConnectedClusters (:253-267) + removeSmallClusters (:354-393); returns the number of surviving clusters.
inline int integrate_and_filter(std::vector<int> &cmap, Lut &lut, int cluster_size_th, std::vector<size_t> &csize,
                                std::vector<int> &old2new) {
  int n_clusters = 0;
  for (size_t i = 0; i < cmap.size(); i++) {
    int c0 = cmap[i];
    if (c0 == -1) continue;
    int r = lut.lookup(c0);
    cmap[i] = r;
    if (r > n_clusters - 1) n_clusters = r + 1;
  }
  if (n_clusters <= 0) return 0;
  csize.assign(n_clusters, 0);
  for (size_t i = 0; i < cmap.size(); i++) if (cmap[i] != -1) csize[cmap[i]] += 1;
  old2new.assign(n_clusters, 0);
  const size_t total = csize.size();
  for (size_t i = 0; i < total; i++) {
    if (csize[i] < (size_t)cluster_size_th) { old2new[i] = -1; n_clusters--; }
    else old2new[i] = (int)(i - (total - n_clusters));
  }
  for (size_t i = 0; i < cmap.size(); i++) if (cmap[i] != -1) cmap[i] = old2new[cmap[i]];
  return n_clusters;
}

// cluster2MovingObject (clusterer_nodelet.cpp:147-184) on gathered members (initial order = column-major pixel order).
template <class GetP>
inline bool make_object(const int *idx, int count, GetP getp, double dynamic_speed, std::vector<std::pair<float, int>> &keyed,
                        OrcObject &ob, int *ambiguous) {
  // pcl::getMinMax3D, dense path (is_dense stays true through toROSMsg/fromROSMsg): no NaN test.
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  for (int k = 0; k < count; k++) {
    PtV p = getp(idx[k]);
    float c[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; a++) { mn[a] = (mn[a] < c[a]) ? mn[a] : c[a]; mx[a] = (mx[a] > c[a]) ? mx[a] : c[a]; }
  }
  for (int a = 0; a < 3; a++) {
    float bb = mx[a] - mn[a];            // Vector4f max - min, F32
    float ce = (mn[a] + mx[a]) / 2;      // (min + max) / 2, F32
    ob.bounding_box[a] = bb;
    ob.center[a] = ce;
  }
  ob.orientation[0] = 0; ob.orientation[1] = 0; ob.orientation[2] = 0; ob.orientation[3] = 1;
  // std::sort by ||v|| descending (libstdc++ introsort, unstable); comparator looks at the norm only, so sorting
  // (norm, member position) pairs with the same comparator performs the identical sequence of moves.
  keyed.resize(count);
  for (int k = 0; k < count; k++) { PtV p = getp(idx[k]); keyed[k] = {norm3(p.vx, p.vy, p.vz), idx[k]}; }
  std::sort(keyed.begin(), keyed.end(),
            [](const std::pair<float, int> &a, const std::pair<float, int> &b) { return a.first > b.first; });
  const std::pair<float, int> med = keyed[count / 2];
  PtV pm = getp(med.second);
  if (ambiguous) {
    // the selected member is implementation-defined when another member ties on the norm with a different vector
    int amb = 0;
    for (int k = 0; k < count; k++)
      if (keyed[k].first == med.first) {
        PtV q = getp(keyed[k].second);
        if (memcmp(&q.vx, &pm.vx, 12) != 0) { amb = 1; break; }
      }
    *ambiguous = amb;
  }
  ob.n_points = count;
  if ((double)norm3(pm.vx, pm.vy, pm.vz) < dynamic_speed) return false;
  ob.velocity[0] = pm.vx; ob.velocity[1] = pm.vy; ob.velocity[2] = pm.vz;
  return true;
}

}  // namespace

extern "C" {

// ---- single-pixel probes (unit tests) ---------------------------------------------------------------------------
int orc_get_point3d(const OrcCamera *c, const float *D, int u, int v, float out[3]) {
  Pt3 p{kNaN, kNaN, kNaN, 1.0f};
  int ok = get_point3d(*c, D, u, v, p) ? 1 : 0;
  out[0] = p.x; out[1] = p.y; out[2] = p.z;
  return ok;
}

void orc_rotation_from_transform(const OrcTransform *tf, double out12[12]) {
  Iso m = make_iso(*tf);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) out12[i * 4 + j] = m.r[i][j]; out12[i * 4 + 3] = m.t[i]; }
}

float orc_norm3(float x, float y, float z) { return norm3(x, y, z); }

// ---- construct(), faithful: scene_flow_constructor.cpp:91-147 ---------------------------------------------------
// cloud_out: W*H 32-byte PointXYZVelocity records; static_flow_out / depth_out optional.
// Returns 0, or the skip code (same numbering as include/mod_sf.h) when an input is missing.
int orc_construct_faithful(const OrcCamera *c, const OrcParams *prm, const float *D_now, const float *D_prev,
                           const float *flow, const OrcTransform *tf, double dt, void *cloud_out,
                           float *static_flow_out, float *depth_out) {
  std::vector<Pt3> pc_now, pc_prev, pc_prevT;
  if (D_prev) to_point_cloud(*c, D_prev, pc_prev);
  if (D_now) {
    to_point_cloud(*c, D_now, pc_now);
    if (depth_out)  // toDepthImage, disparity_image_processor.cpp:105-120
      for (int u = 0; u < c->width; u++)
        for (int v = 0; v < c->height; v++) {
          Pt3 p{0, 0, 0, 1.0f};
          depth_out[(size_t)v * c->width + u] = get_point3d(*c, D_now, u, v, p) ? p.z : kNaN;
        }
  }
  if (!flow) return 3;
  if (!D_prev) return 2;
  if (!tf) return 4;
  transform_prev(*c, pc_prev, *tf, pc_prevT);
  if (!D_now) return 1;
  std::vector<float> sf;
  static_flow(*c, pc_prevT, sf);
  std::vector<PtV> cloud((size_t)c->width * c->height);  // per-frame allocation as in the reference (:138,293-303)
  construct_velocity_pc(*c, *prm, pc_now, pc_prevT, flow, sf, D_now, D_prev, dt, cloud.data());
  if (cloud_out) memcpy(cloud_out, cloud.data(), cloud.size() * sizeof(PtV));  // pcl::toROSMsg payload copy (:358-361)
  if (static_flow_out) memcpy(static_flow_out, sf.data(), sf.size() * sizeof(float));
  return 0;
}

// ---- construct(), tidy: same results, SoA planes, no allocation ---------------------------------------------------
int orc_construct_tidy(const OrcCamera *c, const OrcParams *prm, const float *D_now, const float *D_prev, const float *flow,
                       const OrcTransform *tf, double dt, float *X, float *Y, float *Z, float *VX, float *VY, float *VZ,
                       float *static_flow_out) {
  if (!flow) return 3;
  if (!D_prev) return 2;
  if (!tf) return 4;
  if (!D_now) return 1;
  scene_flow_tidy(*c, *prm, D_now, D_prev, flow, *tf, dt, X, Y, Z, VX, VY, VZ, static_flow_out);
  return 0;
}

// ---- ClustererNodelet::dataCB, faithful: clusterer_nodelet.cpp:221-242 --------------------------------------------
// cloud: W*H 32-byte records.  labels_out: W*H int32 (final cluster_map_).  objects_out: capacity max_objects.
// ambiguous_out (optional, per accepted object): 1 when the median member is tied on ||v|| with a different vector.
// Returns the number of clusters K surviving the size filter (>= n_objects).
int orc_cluster_faithful(const void *cloud, int W, int H, const OrcParams *prm, int32_t *labels_out, OrcObject *objects_out,
                         int max_objects, int32_t *n_objects, int32_t *ambiguous_out) {
  const size_t N = (size_t)W * H;
  std::vector<PtV> pc(N);
  memcpy(pc.data(), cloud, N * sizeof(PtV));  // pcl::fromROSMsg copy (:225-226)
  // calculateDynamicMap (:40-54)
  std::vector<bool> dyn(N, false);
  for (size_t i = 0; i < N; i++)
    if ((double)norm3(pc[i].vx, pc[i].vy, pc[i].vz) >= prm->dynamic_speed) dyn[i] = true;
  // initClusterMap (:244-251)
  std::vector<int> cmap(N, -1);
  Lut lut;
  lut.resize(N);
  lut.reset();
  scan_edges(W, H, prm->neighbor_distance, prm->depth_diff,
             [&](int u, int v) { return pc[(size_t)v * W + u].z; },
             [&](int u, int v) { return (bool)dyn[(size_t)v * W + u]; }, cmap, lut);
  std::vector<size_t> csize;
  std::vector<int> old2new;
  int K = integrate_and_filter(cmap, lut, prm->cluster_size, csize, old2new);
  if (labels_out) for (size_t i = 0; i < N; i++) labels_out[i] = cmap[i];
  // clusterMap2IndicesCluster (:97-117): column-major
  std::vector<std::vector<int>> clusters;
  if (K > 0) {
    clusters.resize(K);
    for (int u = 0; u < W; u++)
      for (int v = 0; v < H; v++) {
        int cn = cmap[(size_t)v * W + u];
        if (cn == -1) continue;
        clusters[cn].push_back(W * v + u);
      }
  }
  // publishMovingObjects (:324-343)
  int id = 0;
  std::vector<std::pair<float, int>> keyed;
  for (int k = 0; k < K; k++) {
    OrcObject ob;
    memset(&ob, 0, sizeof(ob));
    int amb = 0;
    // the reference copies the sub-cloud (:149) before sorting it
    std::vector<PtV> sub(clusters[k].size());
    for (size_t j = 0; j < sub.size(); j++) sub[j] = pc[clusters[k][j]];
    std::vector<int> local(sub.size());
    for (size_t j = 0; j < local.size(); j++) local[j] = (int)j;
    if (make_object(local.data(), (int)local.size(), [&](int j) { return sub[j]; }, prm->dynamic_speed, keyed, ob, &amb)) {
      ob.id = id;
      if (id < max_objects) { if (objects_out) objects_out[id] = ob; if (ambiguous_out) ambiguous_out[id] = amb; }
      id++;
    }
  }
  if (n_objects) *n_objects = id;
  return K;
}

// ---- clusterer, tidy: SoA planes, reusable workspace ----------------------------------------------------------------
void *orc_cluster_workspace_create(void) { return new ClusterWork(); }
void orc_cluster_workspace_destroy(void *w) { delete (ClusterWork *)w; }

int orc_cluster_tidy(void *work, const float *X, const float *Y, const float *Z, const float *VX, const float *VY,
                     const float *VZ, int W, int H, const OrcParams *prm, int32_t *labels_out, OrcObject *objects_out,
                     int max_objects, int32_t *n_objects, int32_t *ambiguous_out) {
  ClusterWork &wk = *(ClusterWork *)work;
  const size_t N = (size_t)W * H;
  wk.dyn.assign(N, 0);
  for (size_t i = 0; i < N; i++) wk.dyn[i] = ((double)norm3(VX[i], VY[i], VZ[i]) >= prm->dynamic_speed) ? 1 : 0;
  wk.cmap.assign(N, -1);
  if (wk.lut.table.size() != N) wk.lut.resize(N);
  wk.lut.reset();
  scan_edges(W, H, prm->neighbor_distance, prm->depth_diff, [&](int u, int v) { return Z[(size_t)v * W + u]; },
             [&](int u, int v) { return wk.dyn[(size_t)v * W + u] != 0; }, wk.cmap, wk.lut);
  int K = integrate_and_filter(wk.cmap, wk.lut, prm->cluster_size, wk.csize, wk.old2new);
  if (labels_out) for (size_t i = 0; i < N; i++) labels_out[i] = wk.cmap[i];
  // group members per cluster, column-major within a cluster (counting sort instead of K push_back vectors)
  wk.offsets.assign(K + 1, 0);
  for (size_t i = 0; i < N; i++) if (wk.cmap[i] != -1) wk.offsets[wk.cmap[i] + 1]++;
  for (int k = 0; k < K; k++) wk.offsets[k + 1] += wk.offsets[k];
  wk.order.resize(K > 0 ? wk.offsets[K] : 0);
  std::vector<int> cursor(wk.offsets.begin(), wk.offsets.end());
  for (int u = 0; u < W; u++)
    for (int v = 0; v < H; v++) {
      int cn = wk.cmap[(size_t)v * W + u];
      if (cn != -1) wk.order[cursor[cn]++] = W * v + u;
    }
  int id = 0;
  for (int k = 0; k < K; k++) {
    OrcObject ob;
    memset(&ob, 0, sizeof(ob));
    int amb = 0;
    auto getp = [&](int i) { return PtV{X[i], Y[i], Z[i], 0.0f, VX[i], VY[i], VZ[i], 0.0f}; };
    if (make_object(&wk.order[wk.offsets[k]], wk.offsets[k + 1] - wk.offsets[k], getp, prm->dynamic_speed, wk.keyed, ob, &amb)) {
      ob.id = id;
      if (id < max_objects) { if (objects_out) objects_out[id] = ob; if (ambiguous_out) ambiguous_out[id] = amb; }
      id++;
    }
  }
  if (n_objects) *n_objects = id;
  return K;
}

// ---- AoS <-> SoA helpers for tests ------------------------------------------------------------------------------------
void orc_unpack_cloud(const void *cloud, int64_t n, float *X, float *Y, float *Z, float *VX, float *VY, float *VZ) {
  const PtV *p = (const PtV *)cloud;
  for (int64_t i = 0; i < n; i++) { X[i] = p[i].x; Y[i] = p[i].y; Z[i] = p[i].z; VX[i] = p[i].vx; VY[i] = p[i].vy; VZ[i] = p[i].vz; }
}
void orc_pack_cloud(void *cloud, int64_t n, const float *X, const float *Y, const float *Z, const float *VX, const float *VY,
                    const float *VZ) {
  PtV *p = (PtV *)cloud;
  for (int64_t i = 0; i < n; i++) p[i] = PtV{X[i], Y[i], Z[i], 0.0f, VX[i], VY[i], VZ[i], 0.0f};
}

// ---- union-find probes (fuzzed against the reference's LookupTable in oracle/_ref) --------------------------------------
void *orc_lut_create(int64_t size) { Lut *l = new Lut(); l->resize((size_t)size); l->reset(); return l; }
void orc_lut_destroy(void *l) { delete (Lut *)l; }
void orc_lut_reset(void *l) { ((Lut *)l)->reset(); }
int orc_lut_add_label(void *l) { return ((Lut *)l)->add_label(); }
void orc_lut_link(void *l, int a, int b) { ((Lut *)l)->link(a, b); }
int orc_lut_lookup(void *l, int s) { return ((Lut *)l)->lookup(s); }

// ---- std::sort probe: position-of-median under libstdc++ introsort for a key array (norm descending) -------------------
// Used to validate the product's introsort emulation.  Returns the ORIGINAL index of the element that ends at n/2.
int orc_sorted_median_index(const float *keys, int n) {
  std::vector<std::pair<float, int>> v(n);
  for (int i = 0; i < n; i++) v[i] = {keys[i], i};
  std::sort(v.begin(), v.end(), [](const std::pair<float, int> &a, const std::pair<float, int> &b) { return a.first > b.first; });
  return v[n / 2].second;
}

}  // extern "C"
