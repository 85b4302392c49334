// sceneflow.hip — fused scene-flow construction kernel for gfx950 (MI355X).
//
// One kernel replaces the five full-image CPU passes of SceneFlowConstructor::construct
// (scene_flow_constructor/src/scene_flow_constructor.cpp:91-147):
//   toPointCloud(now), toPointCloud(prev)   disparity_image_proc/src/disparity_image_processor.cpp:86-103,33-50
//   transformPCPreviousToNow                scene_flow_constructor.cpp:409-429
//   calculateStaticOpticalFlow              scene_flow_constructor.cpp:65-89
//   constructVelocityPC (+ helpers)         scene_flow_constructor.cpp:149-212, scene_flow_constructor.h:173-249
// and, as an epilogue, calculateDynamicMap  scene_flow_clusterer/src/clusterer_nodelet.cpp:40-54.
//
// Memory plan (HBM-bound): per pixel 16 B are read once (disp_now 4, disp_prev 4, flow 8) with
// 16-byte-per-lane row-contiguous loads, 24 B are written as six SoA planes with 16-byte stores, plus 1 bit of mask.
// The previous cloud is never materialised: the point at the flow-warped pixel is re-derived from one 4-byte gather
// of disp_prev (served by L2 — the warp target is a few rows away) and two F64 ray-table entries.
// Arithmetic follows the reference expression by expression (F32 vs F64 as annotated); FP contraction is off so
// that every product and sum rounds exactly like the SSE2 build of the reference.
#include "mod_launch.h"

#pragma clang fp contract(off)

namespace {

struct Px {          // result of one pixel
  float x, y, z, vx, vy, vz, s0, s1, depth;
  bool dyn;
};

__device__ __forceinline__ bool disp_in_range(const DevCam &c, float d) {
  // getDisparity range gate (disparity_image_processor.cpp:25-28): a NaN passes both tests.
  return !(c.dmax < d) && !(c.dmin > d);
}

// Rigid transform, Eigen association t + (p0 + (p1 + p2)) per row (see oracle/oracle.cpp iso_apply), result cast to F32.
__device__ __forceinline__ void iso_apply(const FrameConst &fc, float X, float Y, float Z, float &ox, float &oy, float &oz) {
  const double x = (double)X, y = (double)Y, z = (double)Z;
  ox = (float)(fc.m[3] + (fc.m[0] * x + (fc.m[1] * y + fc.m[2] * z)));
  oy = (float)(fc.m[7] + (fc.m[4] * x + (fc.m[5] * y + fc.m[6] * z)));
  oz = (float)(fc.m[11] + (fc.m[8] * x + (fc.m[9] * y + fc.m[10] * z)));
}

// One pixel of constructVelocityPC with everything it depends on, in two stages so that a thread
// can issue the data-dependent gathers of all its pixels together (one extra memory round trip per thread, not per pixel).
//   stage 1: static flow at the own pixel, now point, warp target        (needs dn, dpo, flow, own rays)
//   stage 2: previous point at the warp target, residual test, velocity   (needs disp_prev / rays AT the warp target)
// The kernel is VALU-bound, not HBM-bound, so the stages are written without control flow: every expression is evaluated
// for every lane (IEEE special values flow through the arithmetic without traps) and the reference's early-outs become
// predicates that select between the computed value and NaN.  Tests that a later NaN test subsumes are noted where dropped.
struct PxState {
  float Xn, Yn, zn, f0, f1;
  int px, py;          // warp target, (0,0) when `go` is false
  bool go;             // every test before the gather passed
};

__device__ __forceinline__ bool is_nan_or_inf(float v) { return __builtin_isfpclass(v, 0x0207); }   // snan|qnan|-inf|+inf
// getRightPoint's rejects (scene_flow_constructor.h:211-227): NaN, +-inf or negative (a -0 passes, as `d < 0` is false)
__device__ __forceinline__ bool is_nan_inf_or_negative(float v) { return __builtin_isfpclass(v, 0x0207 | 0x0018); }

__device__ __forceinline__ void sf_stage1(const DevCam &c, const FrameConst &fc, int x, int y, float dn, float dpo, float f0,
                                          float f1, double rx, double ry, Px &o, PxState &st) {
  const float nan = __uint_as_float(0x7fc00000u);
  st.f0 = f0; st.f1 = f1;

  // ---- static flow at the own pixel: reproject prev, transform, project (always needed by the residual test) ----
  {
    const float z = c.fT / dpo;                             // F32
    const double zd = (double)z;
    const float X = (float)(rx * zd);                       // F64 product -> F32
    const float Y = (float)(ry * zd);
    float tx, ty, tz;
    iso_apply(fc, X, Y, z, tx, ty, tz);
    const double u = (c.fx * (double)tx + c.Tx) / (double)tz + c.cx;     // project3dToPixel, F64
    const double v = (c.fy * (double)ty + c.Ty) / (double)tz + c.cy;
    // the reference skips NaN points before and after the transform; a NaN X makes tx NaN, so one test covers both
    const bool ok = disp_in_range(c, dpo) & !(dpo == 0.0f) & !isnan(tx);
    o.s0 = ok ? (float)(u - (double)x) : nan;
    o.s1 = ok ? (float)(v - (double)y) : nan;
  }

  // ---- now point ----
  const bool okn = disp_in_range(c, dn) & !(dn == 0.0f);
  const float zn = c.fT / dn;
  const double znd = (double)zn;
  const float Xn = (float)(rx * znd);
  const float Yn = (float)(ry * znd);
  st.Xn = Xn; st.Yn = Yn; st.zn = zn;
  o.depth = okn ? zn : nan;                                 // toDepthImage (disparity_image_processor.cpp:105-120)
  const bool valid = okn & !is_nan_or_inf(Xn);              // isValid tests x only (scene_flow_constructor.h:240-249)
  o.x = valid ? Xn : nan; o.y = valid ? Yn : nan; o.z = valid ? zn : nan;
  o.vx = o.vy = o.vz = nan;
  o.dyn = false;

  // ---- getMatchPoints (scene_flow_constructor.h:173-227) ----
  const float rxf = roundf((float)x - f0);                  // std::round, F32, half away from zero
  const float ryf = roundf((float)y - f1);
  // a NaN flow gives a NaN target, which fails the bounds test below; out-of-int-range warps are UB in the reference
  // (x86 yields INT_MIN -> rejected by its bounds test): reject.
  const bool inimg = (rxf >= 0.0f) & (rxf < (float)c.W) & (ryf >= 0.0f) & (ryf < (float)c.H);
  const bool go = valid & inimg & !is_nan_inf_or_negative(dn)            // getRightPoint(now)
                & !isnan(o.s0);                                           // static flow NaN (scene_flow_constructor.cpp:193)
  st.px = go ? (int)rxf : 0; st.py = go ? (int)ryf : 0;
  st.go = go;
}

//   dpw = disparity_prev at the warp target, rpx / rpy = F64 rays of its column / row
// The residual test only needs the static flow of stage 1; when it says "static" the velocity is 0 as long as the previous
// point is valid, and validity (x neither NaN nor inf after the rigid transform) is certain whenever the untransformed
// point is within FrameConst.pad[0] (a bound derived from the transform on the host).  Stage 2a settles those pixels;
// only moving pixels and extreme coordinates go on to stage 2b (second rigid transform + velocity), which the kernels
// guard with a wave-uniform branch — most waves of a street scene skip it entirely.
struct PxWarp { float Xp, Yp, zp; bool todo, moving; };

__device__ __forceinline__ void sf_stage2a(const DevCam &c, const FrameConst &fc, const PxState &st, float dpw, double rpx,
                                           double rpy, Px &o, PxWarp &wp) {
  const float zp = c.fT / dpw;
  const double zpd = (double)zp;
  const float Xp = (float)(rpx * zpd);
  const float Yp = (float)(rpy * zpd);
  // getRightPoint(previous): getDisparity range gate, then NaN / inf / negative; a 0 disparity leaves NaN in the previous
  // cloud (!isValid); a NaN x passes through the transform untouched -> invalid
  const bool ok = st.go & disp_in_range(c, dpw) & !is_nan_inf_or_negative(dpw) & !(dpw == 0.0f) & !isnan(Xp);
  // ---- residual test (scene_flow_constructor.cpp:196-198): sqrtf(acc) >= flow_th  <=>  acc >= flow_th_sq (host-derived) ----
  const float r0 = st.f0 - o.s0, r1 = st.f1 - o.s1;
  float acc = 0.0f;
  acc = acc + r0 * r0;
  acc = acc + r1 * r1;
  const bool moving = acc >= c.flow_th_sq;
  const float safe = (float)fc.pad[0];
  const bool settled = ok & !moving & (fabsf(Xp) <= safe) & (fabsf(Yp) <= safe) & (fabsf(zp) <= safe);
  // settled: the transformed point is certainly finite -> valid and static
  o.vx = settled ? 0.0f : o.vx; o.vy = settled ? 0.0f : o.vy; o.vz = settled ? 0.0f : o.vz;
  o.dyn = settled & (0.0f >= c.speed_th_sq);
  wp.Xp = Xp; wp.Yp = Yp; wp.zp = zp; wp.todo = ok & !settled; wp.moving = moving;
}

__device__ __forceinline__ void sf_stage2b(const DevCam &c, const FrameConst &fc, const PxState &st, const PxWarp &wp, Px &o) {
  float tx, ty, tz;
  iso_apply(fc, wp.Xp, wp.Yp, wp.zp, tx, ty, tz);
  const bool ok = wp.todo & !is_nan_or_inf(tx);
  // velocity (scene_flow_constructor.cpp:200-202); 0 when the residual test said "static"
  const float vx = wp.moving ? (float)((double)(st.Xn - tx) / fc.dt) : 0.0f;
  const float vy = wp.moving ? (float)((double)(st.Yn - ty) / fc.dt) : 0.0f;
  const float vz = wp.moving ? (float)((double)(st.zn - tz) / fc.dt) : 0.0f;
  o.vx = ok ? vx : o.vx; o.vy = ok ? vy : o.vy; o.vz = ok ? vz : o.vz;
  // calculateDynamicMap: (double)||v|| >= dynamic_speed, folded on the host into an F32 threshold on x^2 + (y^2 + z^2)
  o.dyn = ok ? (sumsq3_f32(vx, vy, vz) >= c.speed_th_sq) : o.dyn;
}

__device__ __forceinline__ void sf_pixel(const DevCam &c, const FrameConst &fc, const float *__restrict__ dprev, int x, int y,
                                         float dn, float dpo, float f0, float f1, double rx, double ry, Px &o) {
  PxState st;
  sf_stage1(c, fc, x, y, dn, dpo, f0, f1, rx, ry, o, st);
  PxWarp wp;
  sf_stage2a(c, fc, st, dprev[(size_t)st.py * c.W + st.px], c.rayx[st.px], c.rayy[st.py], o, wp);
  if (__any(wp.todo)) sf_stage2b(c, fc, st, wp, o);
}

// OR-combine the 4-bit nibbles of 16 consecutive lanes into one 64-bit word (lane 16k -> word k of the wave).
__device__ __forceinline__ uint64_t nibbles_to_word(uint32_t nib, int lane) {
  uint32_t v = nib << (4 * (lane & 7));
  v |= __shfl_xor(v, 1);
  v |= __shfl_xor(v, 2);
  v |= __shfl_xor(v, 4);
  const uint32_t hi = __shfl_down(v, 8);
  return (uint64_t)v | ((uint64_t)hi << 32);
}

// Vector kernel: W % 4 == 0.  Block = 64 x 4 threads, thread = 4 consecutive pixels of a row, wave = 256 px of one row.
__global__ __launch_bounds__(256) void k_scene_flow_v4(DevCam c, SfArgs a) {
  const int lane = threadIdx.x;                      // 0..63
  const int x0 = (blockIdx.x * 64 + lane) * 4;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  const bool inb = (x0 < c.W) && (y < c.H);
  const size_t N = (size_t)c.W * c.H;
  const size_t base = (size_t)f * N + (size_t)y * c.W + x0;
  const FrameConst fc = a.fc[f];
  uint32_t nib = 0;
  if (inb) {
    const float4 dn = *reinterpret_cast<const float4 *>(a.dnow + base);
    const float4 dp = *reinterpret_cast<const float4 *>(a.dprev + base);
    const float4 fa = *reinterpret_cast<const float4 *>(a.flow + 2 * base);
    const float4 fb = *reinterpret_cast<const float4 *>(a.flow + 2 * base + 4);
    const double ry = c.rayy[y];
    const double2 rxa = *reinterpret_cast<const double2 *>(c.rayx + x0);
    const double2 rxb = *reinterpret_cast<const double2 *>(c.rayx + x0 + 2);
    const float *dprev_f = a.dprev + (size_t)f * N;
    Px p0, p1, p2, p3;
    PxState s0, s1, s2, s3;
    sf_stage1(c, fc, x0 + 0, y, dn.x, dp.x, fa.x, fa.y, rxa.x, ry, p0, s0);
    sf_stage1(c, fc, x0 + 1, y, dn.y, dp.y, fa.z, fa.w, rxa.y, ry, p1, s1);
    sf_stage1(c, fc, x0 + 2, y, dn.z, dp.z, fb.x, fb.y, rxb.x, ry, p2, s2);
    sf_stage1(c, fc, x0 + 3, y, dn.w, dp.w, fb.z, fb.w, rxb.y, ry, p3, s3);
    // the four gathers (and their ray-table reads) leave together: unconditional loads at in-image (clamped) targets
    const float g0 = dprev_f[(size_t)s0.py * c.W + s0.px], g1 = dprev_f[(size_t)s1.py * c.W + s1.px];
    const float g2 = dprev_f[(size_t)s2.py * c.W + s2.px], g3 = dprev_f[(size_t)s3.py * c.W + s3.px];
    const double ax0 = c.rayx[s0.px], ax1 = c.rayx[s1.px], ax2 = c.rayx[s2.px], ax3 = c.rayx[s3.px];
    const double ay0 = c.rayy[s0.py], ay1 = c.rayy[s1.py], ay2 = c.rayy[s2.py], ay3 = c.rayy[s3.py];
    PxWarp w0, w1, w2, w3;
    sf_stage2a(c, fc, s0, g0, ax0, ay0, p0, w0);
    sf_stage2a(c, fc, s1, g1, ax1, ay1, p1, w1);
    sf_stage2a(c, fc, s2, g2, ax2, ay2, p2, w2);
    sf_stage2a(c, fc, s3, g3, ax3, ay3, p3, w3);
    if (__any(w0.todo || w1.todo || w2.todo || w3.todo)) {   // wave-uniform: some pixel moves (or has extreme coordinates)
      sf_stage2b(c, fc, s0, w0, p0);
      sf_stage2b(c, fc, s1, w1, p1);
      sf_stage2b(c, fc, s2, w2, p2);
      sf_stage2b(c, fc, s3, w3, p3);
    }
    *reinterpret_cast<float4 *>(a.x + base) = make_float4(p0.x, p1.x, p2.x, p3.x);
    *reinterpret_cast<float4 *>(a.y + base) = make_float4(p0.y, p1.y, p2.y, p3.y);
    *reinterpret_cast<float4 *>(a.z + base) = make_float4(p0.z, p1.z, p2.z, p3.z);
    *reinterpret_cast<float4 *>(a.vx + base) = make_float4(p0.vx, p1.vx, p2.vx, p3.vx);
    *reinterpret_cast<float4 *>(a.vy + base) = make_float4(p0.vy, p1.vy, p2.vy, p3.vy);
    *reinterpret_cast<float4 *>(a.vz + base) = make_float4(p0.vz, p1.vz, p2.vz, p3.vz);
    if (a.aos) {   // pcl::PointXYZVelocity records, 32 B each (pads written as 0)
      float4 *q = a.aos + 2 * base;
      q[0] = make_float4(p0.x, p0.y, p0.z, 0.f); q[1] = make_float4(p0.vx, p0.vy, p0.vz, 0.f);
      q[2] = make_float4(p1.x, p1.y, p1.z, 0.f); q[3] = make_float4(p1.vx, p1.vy, p1.vz, 0.f);
      q[4] = make_float4(p2.x, p2.y, p2.z, 0.f); q[5] = make_float4(p2.vx, p2.vy, p2.vz, 0.f);
      q[6] = make_float4(p3.x, p3.y, p3.z, 0.f); q[7] = make_float4(p3.vx, p3.vy, p3.vz, 0.f);
    }
    if (a.depth) *reinterpret_cast<float4 *>(a.depth + base) = make_float4(p0.depth, p1.depth, p2.depth, p3.depth);
    if (a.sflow) {
      *reinterpret_cast<float4 *>(a.sflow + 2 * base) = make_float4(p0.s0, p0.s1, p1.s0, p1.s1);
      *reinterpret_cast<float4 *>(a.sflow + 2 * base + 4) = make_float4(p2.s0, p2.s1, p3.s0, p3.s1);
    }
    nib = (p0.dyn ? 1u : 0u) | (p1.dyn ? 2u : 0u) | (p2.dyn ? 4u : 0u) | (p3.dyn ? 8u : 0u);
  }
  if (a.mask) {   // wave-uniform branch; all 64 lanes take part in the shuffles
    const uint64_t w = nibbles_to_word(nib, lane);
    const int word = (blockIdx.x * 64 + lane) / 16;          // (x0 / 64)
    if ((lane & 15) == 0 && y < c.H && word < c.mask_words)
      a.mask[((size_t)f * c.H + y) * c.mask_words + word] = w;
  }
}

// Scalar kernel for widths that are not a multiple of 4: thread = 1 pixel, wave = 64 px of one row.
__global__ __launch_bounds__(256) void k_scene_flow_v1(DevCam c, SfArgs a) {
  const int lane = threadIdx.x;
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  const bool inb = (x < c.W) && (y < c.H);
  const size_t N = (size_t)c.W * c.H;
  const size_t i = (size_t)f * N + (size_t)y * c.W + x;
  const FrameConst fc = a.fc[f];
  bool dyn = false;
  if (inb) {
    Px p;
    sf_pixel(c, fc, a.dprev + (size_t)f * N, x, y, a.dnow[i], a.dprev[i], a.flow[2 * i], a.flow[2 * i + 1], c.rayx[x], c.rayy[y], p);
    a.x[i] = p.x; a.y[i] = p.y; a.z[i] = p.z; a.vx[i] = p.vx; a.vy[i] = p.vy; a.vz[i] = p.vz;
    if (a.aos) { a.aos[2 * i] = make_float4(p.x, p.y, p.z, 0.f); a.aos[2 * i + 1] = make_float4(p.vx, p.vy, p.vz, 0.f); }
    if (a.depth) a.depth[i] = p.depth;
    if (a.sflow) { a.sflow[2 * i] = p.s0; a.sflow[2 * i + 1] = p.s1; }
    dyn = p.dyn;
  }
  if (a.mask) {
    const uint64_t w = __ballot(dyn);
    if (lane == 0 && y < c.H && blockIdx.x < (unsigned)c.mask_words) a.mask[((size_t)f * c.H + y) * c.mask_words + blockIdx.x] = w;
  }
}

// calculateDynamicMap alone (clusterer_nodelet.cpp:40-54) for clouds that did not come from the fused kernel.
__global__ __launch_bounds__(256) void k_dynamic_mask(DevCam c, const float *__restrict__ vx, const float *__restrict__ vy,
                                                      const float *__restrict__ vz, uint64_t *__restrict__ mask) {
  const int lane = threadIdx.x;
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  bool dyn = false;
  if (x < c.W && y < c.H) {
    const size_t i = ((size_t)f * c.H + y) * c.W + x;
    dyn = sumsq3_f32(vx[i], vy[i], vz[i]) >= c.speed_th_sq;
  }
  const uint64_t w = __ballot(dyn);
  if (lane == 0 && y < c.H && blockIdx.x < (unsigned)c.mask_words) mask[((size_t)f * c.H + y) * c.mask_words + blockIdx.x] = w;
}

// SoA <-> 32-byte AoS (pcl::toROSMsg / fromROSMsg payloads, scene_flow_constructor.cpp:358-361, clusterer_nodelet.cpp:226).
__global__ __launch_bounds__(256) void k_pack(size_t n, const float *x, const float *y, const float *z, const float *vx,
                                              const float *vy, const float *vz, float4 *aos) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  aos[2 * i] = make_float4(x[i], y[i], z[i], 0.f);
  aos[2 * i + 1] = make_float4(vx[i], vy[i], vz[i], 0.f);
}
__global__ __launch_bounds__(256) void k_unpack(size_t n, const float4 *aos, float *x, float *y, float *z, float *vx, float *vy,
                                                float *vz) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 a = aos[2 * i], b = aos[2 * i + 1];
  x[i] = a.x; y[i] = a.y; z[i] = a.z; vx[i] = b.x; vy[i] = b.y; vz[i] = b.z;
}

}  // namespace

void launch_scene_flow(const DevCam &c, const SfArgs &a, int frames, hipStream_t s) {
  dim3 block(64, 4, 1);
  if ((c.W & 3) == 0) {
    dim3 grid((c.W / 4 + 63) / 64, (c.H + 3) / 4, frames);
    hipLaunchKernelGGL(k_scene_flow_v4, grid, block, 0, s, c, a);
  } else {
    dim3 grid((c.W + 63) / 64, (c.H + 3) / 4, frames);
    hipLaunchKernelGGL(k_scene_flow_v1, grid, block, 0, s, c, a);
  }
}

void launch_dynamic_mask(const DevCam &c, int frames, const float *vx, const float *vy, const float *vz, uint64_t *mask,
                         hipStream_t s) {
  dim3 block(64, 4, 1), grid((c.W + 63) / 64, (c.H + 3) / 4, frames);
  hipLaunchKernelGGL(k_dynamic_mask, grid, block, 0, s, c, vx, vy, vz, mask);
}

void launch_pack(size_t n, const float *x, const float *y, const float *z, const float *vx, const float *vy, const float *vz,
                 void *aos, hipStream_t s) {
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, x, y, z, vx, vy, vz, (float4 *)aos);
}
void launch_unpack(size_t n, const void *aos, float *x, float *y, float *z, float *vx, float *vy, float *vz, hipStream_t s) {
  hipLaunchKernelGGL(k_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, (const float4 *)aos, x, y, z, vx, vy, vz);
}
