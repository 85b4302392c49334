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
#include <cstddef>
#include <type_traits>

#include "exact_div.h"
#include "mod_launch.h"
#include <algorithm>

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

__device__ __forceinline__ bool is_ordinary(float v) { return __builtin_isfpclass(v, 0x0198); }      // +-normal, +-subnormal
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
    // project3dToPixel, F64: (fx X + Tx) / Z + cx and (fy Y + Ty) / Z + cy share one refined reciprocal of Z (exact_div.h)
    const double A = c.fx * (double)tx + c.Tx, B = c.fy * (double)ty + c.Ty, D = (double)tz;
    // the reference skips NaN points before and after the transform; a NaN X makes tx NaN, so one test covers both
    const bool ok = disp_in_range(c, dpo) & !(dpo == 0.0f) & !isnan(tx);
    double qa, qb;
    const bool fast = exact_div::div2_shared(A, B, D, qa, qb);
    if (__any(ok & !fast)) {                                // operands outside the window: IEEE divisions (rare, wave-uniform)
      qa = fast ? qa : A / D;
      qb = fast ? qb : B / D;
    }
    const double u = qa + c.cx, v = qb + c.cy;
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
  // velocity (scene_flow_constructor.cpp:200-202); 0 when the residual test said "static".  The F64 division by the frame's dt
  // uses the host's correctly rounded 1/dt (FrameConst.pad[1], exact_div.h) unless dt is outside the usable window.
  const float dx = st.Xn - tx, dy = st.Yn - ty, dz = st.zn - tz;
  float vx, vy, vz;
  if (fc.pad[2] != 0.0) {
    vx = (float)exact_div::div_by_known((double)dx, is_ordinary(dx), fc.dt, fc.pad[1]);
    vy = (float)exact_div::div_by_known((double)dy, is_ordinary(dy), fc.dt, fc.pad[1]);
    vz = (float)exact_div::div_by_known((double)dz, is_ordinary(dz), fc.dt, fc.pad[1]);
  } else {
    vx = (float)((double)dx / fc.dt); vy = (float)((double)dy / fc.dt); vz = (float)((double)dz / fc.dt);
  }
  vx = wp.moving ? vx : 0.0f; vy = wp.moving ? vy : 0.0f; vz = wp.moving ? vz : 0.0f;
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
  // DPP moves (quad swaps, then the mirror inside each group of 8: after two steps the four lanes of a quad agree, so the mirror
  // brings in the other quad), not __shfl's ds_bpermute — those are LDS-pipeline round trips
  uint32_t v = nib << (4 * (lane & 7));
  v |= MOD_DPP(v, 0xB1);                             // quad_perm [1,0,3,2]
  v |= MOD_DPP(v, 0x4E);                             // quad_perm [2,3,0,1]
  v |= MOD_DPP(v, 0x141);                            // row_half_mirror
  const uint32_t hi = MOD_DPP(v, 0x108);             // row_shl:8: lane 16k reads lane 16k + 8
  return (uint64_t)v | ((uint64_t)hi << 32);
}

// Depth range of the DYNAMIC pixels of a mask word, stored next to the word (SfArgs.zrange): a dynamic pixel has a finite z (it is
// valid: x = ray * z is neither NaN nor inf), everything else enters as a quiet NaN, which v_min / v_max pass over.  The plain
// instructions: fminf / fmaxf would put a canonicalising v_max x, x in front of every operand.
__device__ __forceinline__ float sf_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sf_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define SF_DPPF(v, ctrl) __uint_as_float(MOD_DPP(__float_as_uint(v), ctrl))
// min / max over the 16 lanes of a DPP row (= the 64 pixels of one mask word in k_scene_flow_v4): every lane of the row ends with the result
__device__ __forceinline__ void row16_minmax(float &mn, float &mx) {
  mn = sf_min(mn, SF_DPPF(mn, 0xB1)); mx = sf_max(mx, SF_DPPF(mx, 0xB1));      // quad_perm [1,0,3,2]
  mn = sf_min(mn, SF_DPPF(mn, 0x4E)); mx = sf_max(mx, SF_DPPF(mx, 0x4E));      // quad_perm [2,3,0,1]
  mn = sf_min(mn, SF_DPPF(mn, 0x141)); mx = sf_max(mx, SF_DPPF(mx, 0x141));    // row_half_mirror
  mn = sf_min(mn, SF_DPPF(mn, 0x140)); mx = sf_max(mx, SF_DPPF(mx, 0x140));    // row_mirror
}
// the same over groups of G consecutive lanes (the narrower kernels: 32 lanes or the whole wave per mask word); not on the hot path
template <int G> __device__ __forceinline__ void group_minmax(float &mn, float &mx) {
#pragma unroll
  for (int o = 1; o < G; o <<= 1) { mn = sf_min(mn, __shfl_xor(mn, o)); mx = sf_max(mx, __shfl_xor(mx, o)); }
}

// Addressing: a wave-uniform 64-bit base (frame plane, in SGPRs) plus a 32-bit byte offset per lane, which is the
// `global_load/store v, v_off, s[base]` form — no 64-bit address arithmetic on the vector unit (the kernel is VALU-bound).
// mod_set_camera guarantees W*H*32 < 2^32.
template <class T> __device__ __forceinline__ T ld(const void *base, uint32_t byte_off) {
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <class T> __device__ __forceinline__ void st(void *base, uint32_t byte_off, const T &v) {
  *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
// Kernel arguments that are needed late are read from the kernarg segment after the register-heavy first stage (one scalar load
// each, issued together with the gathers so that their latency hides behind those) instead of being held in SGPRs across the
// arithmetic: with them the kernel wants more than the 102 SGPRs a wave has and spills into VGPR lanes (v_writelane / v_readlane
// pairs, which take VALU issue slots of a VALU-co-limited kernel).  Read through the kernarg segment pointer: taking the address of
// a by-value parameter would make the compiler copy it to scratch.
// Plane stores are WRITE-THROUGH streaming stores (`sc0 sc1 nt`): a step writes 11 GB of planes that nobody on this GPU reads
// before 20 GB of other traffic has passed, so a line kept dirty in the XCD's L2 only waits for its eviction — written through, it
// leaves as it is produced.  In-process A/B on one set of buffers (tools/ab_inproc.py, 512 pairs, four alternations, +-0.01 ms):
// plain 3.712 ms, `nt sc1` 3.621, `sc0 sc1 nt` 3.606; x, y, z alone 3.673, vx, vy, vz alone 3.660; `nt` alone (round 3) and
// non-temporal LOADS of the disparity / flow planes (+4 %) do not help.  The builtin offers `nt` only, hence the asm; the
// instruction is the very global_store_dwordx4 (SGPR base + 32-bit lane offset) the compiler emits for st().
__device__ __forceinline__ void st_plane(float *base, uint32_t byte_off, float a0, float a1, float a2, float a3) {
  typedef float sf_w4 __attribute__((ext_vector_type(4)));
  const sf_w4 v = {a0, a1, a2, a3};
  asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt" :: "v"(byte_off), "v"(v), "s"(base) : "memory");
}
#define SF_K4 __attribute__((address_space(4)))
template <class T> __device__ __forceinline__ T karg(size_t off) {
  const SF_K4 char *kp = (const SF_K4 char *)__builtin_amdgcn_kernarg_segment_ptr();
  return *(const volatile SF_K4 T *)(kp + off);
}
constexpr size_t kSfArgsAt = (sizeof(DevCam) + alignof(SfArgs) - 1) / alignof(SfArgs) * alignof(SfArgs);   // k(DevCam c, SfArgs a)
#define LATE_A(field) karg<decltype(SfArgs::field)>(kSfArgsAt + offsetof(SfArgs, field))
// karg()/LATE_A are tied to the signature `k_scene_flow_v4<XY>(DevCam c, SfArgs a)`: by-value aggregates are laid out in the kernarg
// segment in declaration order at their natural alignment (code object v5: explicit arguments first, hidden ones after), i.e.
// exactly like the members of the struct below.  A change of the signature or of either struct must change this too — a wrong
// offset would store through garbage pointers.  The checked build also compares LATE_A(vx) with a.vx at run time (code 16).
struct SfKernargLayout { DevCam c; SfArgs a; };
static_assert(std::is_trivially_copyable<DevCam>::value && std::is_trivially_copyable<SfArgs>::value, "kernarg structs are copied bytewise");
static_assert(alignof(DevCam) <= 8 && alignof(SfArgs) <= 8, "kernarg segment is 8-byte aligned per argument here");
static_assert(kSfArgsAt == offsetof(SfKernargLayout, a), "SfArgs does not sit where karg() reads it");
// Vector kernel: W % 4 == 0.  Block = 64 x 4 threads, thread = 4 consecutive pixels of a row, wave = 256 px of one row.
// XY: the x and y planes are written (ModSceneFlowPlanes.x / .y given).  Two instances rather than a branch: the six-plane kernel's
// instruction stream stays what it was (the kernel is sensitive to it: a wave-uniform `if (a.x)` around the two stores cost +1.6 %).
// INL (small batches): the per-frame constants ride in the kernel arguments (SfConsts behind SfArgs: no copy from the host's pinned
// ring in front of the kernel — a blit kernel plus the wait for it, 9 of a one-pair step's 100 us) and are read through the kernarg
// pointer like the late arguments.  The kernels of large batches keep their signature and their instruction stream.
constexpr int kSfInlineFrames = MOD_SF_INLINE_FRAMES;
struct SfConsts { FrameConst v[kSfInlineFrames]; };
struct SfKernargLayoutInl { DevCam c; SfArgs a; SfConsts k; };
static_assert(std::is_trivially_copyable<SfConsts>::value && alignof(SfConsts) <= 8, "kernarg struct");
template <bool INL> __device__ __forceinline__ FrameConst frame_const(const SfArgs &a, int f) {
  if (INL) {
    const SF_K4 char *kp = (const SF_K4 char *)__builtin_amdgcn_kernarg_segment_ptr();
    const SF_K4 double *q = (const SF_K4 double *)(kp + offsetof(SfKernargLayoutInl, k) + (size_t)f * sizeof(FrameConst));
    static_assert(sizeof(FrameConst) == 16 * sizeof(double), "FrameConst is 16 doubles");
    FrameConst r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.m[i] = q[i];
    r.dt = q[12]; r.pad[0] = q[13]; r.pad[1] = q[14]; r.pad[2] = q[15];
    return r;
  }
  return a.fc[f];
}

template <bool XY, bool INL>
__device__ __forceinline__ void sf_v4_body(const DevCam &c, const SfArgs &a) {
  const int lane = threadIdx.x;                      // 0..63
  // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs in dispatch order (x fastest), and each XCD has its
  // own L2.  Here XCD k takes every 8th ROW of blocks (4 image rows, all of their x blocks): the blocks in flight still cover ONE
  // contiguous window of each plane (about 1.4 frames), as in plain dispatch order, and the rows a block's flow-warp gathers touch
  // are in the L2 that fetched them for its x neighbours.  In-process A/B (tools/ab_inproc.py, 512 pairs, boxes A / B / C): XCD k on
  // the k-th contiguous EIGHTH of the batch (rounds 1-3: eight windows per plane) 3.387 / 3.611 / 3.561 ms; this order 3.331 /
  // 3.507 / 3.475 (-1.7 ... -2.9 %); plain dispatch order 3.346 (A), 3.471 (C); runs of 2 / 4 / 9 / 45 block rows per XCD
  // 3.523 / 3.521 / 3.564 / 3.708 (B); XCD k on frames = k (mod 8) 3.408 (A); all XCDs inside one pair of frames, an eighth each,
  // 3.552 (A); frame-linear blocks (4 consecutive 256-px segments) 3.631 (C).  The coarser the interleave, the slower.  It also
  // makes the kernel insensitive to where the planes sit (tools/ab_skew.py: 3.37-3.38 ms for every stagger from 4 KiB to 2 MiB,
  // 3.40 back to back).
  uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const uint32_t rows = gridDim.y * gridDim.z;             // block rows of the batch
    if ((rows & 7u) == 0u && !MOD_ABLATE(c, 64)) {
      const uint32_t lin = bx + gridDim.x * (by + gridDim.y * bz);
      const uint32_t j = lin >> 3, k = lin & 7u;             // k: the XCD this block was dealt to, j: its turn there
      const uint32_t r = j / gridDim.x * 8u + k;             // block row
      bx = j % gridDim.x;
      by = r % gridDim.y; bz = r / gridDim.y;
    }
  }
  const int x0 = (bx * 64 + lane) * 4;
  const int y = by * 4 + threadIdx.y;
  const int f = bz;
  const bool inb = (x0 < c.W) && (y < c.H);
  const size_t fN = (size_t)f * ((size_t)c.W * c.H);         // wave-uniform
  const uint32_t pix = (uint32_t)y * (uint32_t)c.W + (uint32_t)x0;
  const uint32_t o4 = pix * 4u, o8 = pix * 8u;
  const FrameConst fc = frame_const<INL>(a, f);
  uint32_t nib = 0;
  uint64_t *out_mask = nullptr;
  int32_t *out_hdr = nullptr;
  float2 *out_zr = nullptr;
  const float qnan = __uint_as_float(0x7fc00000u);
  float zmn = qnan, zmx = qnan;                       // depth range of this thread's dynamic pixels
  if (inb) {
    const float *dprev_f = a.dprev + fN;
    const float4 dn = ld<float4>(a.dnow + fN, o4);
    const float4 dp = ld<float4>(dprev_f, o4);
    const float4 fa = ld<float4>(a.flow + 2 * fN, o8);
    const float4 fb = ld<float4>(a.flow + 2 * fN, o8 + 16u);
    const double ry = c.rayy[y];
    const double2 rxa = ld<double2>(c.rayx, (uint32_t)x0 * 8u);
    const double2 rxb = ld<double2>(c.rayx, (uint32_t)x0 * 8u + 16u);
    Px p0, p1, p2, p3;
    PxState s0, s1, s2, s3;
    sf_stage1(c, fc, x0 + 0, y, dn.x, dp.x, fa.x, fa.y, rxa.x, ry, p0, s0);
    sf_stage1(c, fc, x0 + 1, y, dn.y, dp.y, fa.z, fa.w, rxa.y, ry, p1, s1);
    sf_stage1(c, fc, x0 + 2, y, dn.z, dp.z, fb.x, fb.y, rxb.x, ry, p2, s2);
    sf_stage1(c, fc, x0 + 3, y, dn.w, dp.w, fb.z, fb.w, rxb.y, ry, p3, s3);
    // x, y, z are final after stage 1: their stores leave before the gathers come back.  The x and y planes are optional (both or
    // neither; wave-uniform): a caller that serves the moving objects alone — no ~scene_flow subscriber, scene_flow_constructor.cpp:141-142
    // — does not pay their 8 B/px, and the cluster stage recomputes a member's x, y from z (k_final XY_FROM_Z)
    if (XY) {
      st_plane(a.x + fN, o4, p0.x, p1.x, p2.x, p3.x);
      st_plane(a.y + fN, o4, p0.y, p1.y, p2.y, p3.y);
    }
    st_plane(a.z + fN, o4, p0.z, p1.z, p2.z, p3.z);
    // the four gathers (and their ray-table reads) leave together: unconditional loads at in-image targets; so do the scalar
    // loads of the output pointers that are needed from here on
    float *const out_vx = LATE_A(vx), *const out_vy = LATE_A(vy), *const out_vz = LATE_A(vz);
#ifdef MOD_CHECKED
    if (out_vx != a.vx || LATE_A(tilehdr) != a.tilehdr || LATE_A(tiles_x) != a.tiles_x) {   // the late loads read what the signature passes
      if (a.dbg) atomicAdd(&a.dbg[kDbgCheckBase + 16], 1ull);
      return;
    }
#endif
    float4 *const out_aos = LATE_A(aos);
    float *const out_depth = LATE_A(depth), *const out_sflow = LATE_A(sflow);
    out_mask = LATE_A(mask); out_hdr = LATE_A(tilehdr); out_zr = LATE_A(zrange);
    const uint32_t W = (uint32_t)c.W;
    const float g0 = ld<float>(dprev_f, ((uint32_t)s0.py * W + (uint32_t)s0.px) * 4u);
    const float g1 = ld<float>(dprev_f, ((uint32_t)s1.py * W + (uint32_t)s1.px) * 4u);
    const float g2 = ld<float>(dprev_f, ((uint32_t)s2.py * W + (uint32_t)s2.px) * 4u);
    const float g3 = ld<float>(dprev_f, ((uint32_t)s3.py * W + (uint32_t)s3.px) * 4u);
    const double ax0 = ld<double>(c.rayx, (uint32_t)s0.px * 8u), ax1 = ld<double>(c.rayx, (uint32_t)s1.px * 8u);
    const double ax2 = ld<double>(c.rayx, (uint32_t)s2.px * 8u), ax3 = ld<double>(c.rayx, (uint32_t)s3.px * 8u);
    const double ay0 = ld<double>(c.rayy, (uint32_t)s0.py * 8u), ay1 = ld<double>(c.rayy, (uint32_t)s1.py * 8u);
    const double ay2 = ld<double>(c.rayy, (uint32_t)s2.py * 8u), ay3 = ld<double>(c.rayy, (uint32_t)s3.py * 8u);
    PxWarp w0, w1, w2, w3;
    sf_stage2a(c, fc, s0, g0, ax0, ay0, p0, w0);
    sf_stage2a(c, fc, s1, g1, ax1, ay1, p1, w1);
    sf_stage2a(c, fc, s2, g2, ax2, ay2, p2, w2);
    sf_stage2a(c, fc, s3, g3, ax3, ay3, p3, w3);
    if (__any(w0.todo | w1.todo | w2.todo | w3.todo)) {      // wave-uniform: some pixel moves (or has extreme coordinates)
      sf_stage2b(c, fc, s0, w0, p0);
      sf_stage2b(c, fc, s1, w1, p1);
      sf_stage2b(c, fc, s2, w2, p2);
      sf_stage2b(c, fc, s3, w3, p3);
    }
    st_plane(out_vx + fN, o4, p0.vx, p1.vx, p2.vx, p3.vx);
    st_plane(out_vy + fN, o4, p0.vy, p1.vy, p2.vy, p3.vy);
    st_plane(out_vz + fN, o4, p0.vz, p1.vz, p2.vz, p3.vz);
    if (out_aos) {   // pcl::PointXYZVelocity records, 32 B each (pads written as 0)
      float4 *q = out_aos + 2 * fN;
      const uint32_t o32 = pix * 32u;
      st(q, o32, make_float4(p0.x, p0.y, p0.z, 0.f)); st(q, o32 + 16u, make_float4(p0.vx, p0.vy, p0.vz, 0.f));
      st(q, o32 + 32u, make_float4(p1.x, p1.y, p1.z, 0.f)); st(q, o32 + 48u, make_float4(p1.vx, p1.vy, p1.vz, 0.f));
      st(q, o32 + 64u, make_float4(p2.x, p2.y, p2.z, 0.f)); st(q, o32 + 80u, make_float4(p2.vx, p2.vy, p2.vz, 0.f));
      st(q, o32 + 96u, make_float4(p3.x, p3.y, p3.z, 0.f)); st(q, o32 + 112u, make_float4(p3.vx, p3.vy, p3.vz, 0.f));
    }
    if (out_depth) st(out_depth + fN, o4, make_float4(p0.depth, p1.depth, p2.depth, p3.depth));
    if (out_sflow) {
      st(out_sflow + 2 * fN, o8, make_float4(p0.s0, p0.s1, p1.s0, p1.s1));
      st(out_sflow + 2 * fN, o8 + 16u, make_float4(p2.s0, p2.s1, p3.s0, p3.s1));
    }
    nib = (p0.dyn ? 1u : 0u) | (p1.dyn ? 2u : 0u) | (p2.dyn ? 4u : 0u) | (p3.dyn ? 8u : 0u);
    if (__any(nib != 0u)) {                                  // wave-uniform: few waves of a street scene hold a dynamic pixel
      // a dynamic pixel is valid, so its z IS st.zn — which stage 2b kept alive anyway (p.z would be four more registers held since stage 1)
      const float z0 = p0.dyn ? s0.zn : qnan, z1 = p1.dyn ? s1.zn : qnan, z2 = p2.dyn ? s2.zn : qnan, z3 = p3.dyn ? s3.zn : qnan;
      zmn = sf_min(sf_min(z0, z1), sf_min(z2, z3)); zmx = sf_max(sf_max(z0, z1), sf_max(z2, z3));
    }
  }
  if (!inb) { out_mask = LATE_A(mask); out_hdr = LATE_A(tilehdr); out_zr = LATE_A(zrange); }
  if (out_mask) {   // wave-uniform branch; all 64 lanes take part in the shuffles
    const uint64_t w = nibbles_to_word(nib, lane);
    const int word = (bx * 64 + lane) / 16;                  // (x0 / 64)
    // In-process A/B of this epilogue addition (round 5, 512 pairs, 8 alternations): no ranges 3.424 ms, the store alone 3.417, the
    // min / max alone 3.443, both 3.441 against 3.434 for round 4's kernel — inside the +-0.01 ms of the method; the tile stage saves 0.033 ms.
    if (out_zr && __any(nib != 0u)) row16_minmax(zmn, zmx);  // wave-uniform: the clustering follows and the wave holds a dynamic pixel
    if ((lane & 15) == 0 && y < c.H && word < c.mask_words) {
      const size_t wo = ((size_t)f * c.H + y) * c.mask_words + word;
      out_mask[wo] = w;
      if (out_hdr && w) out_hdr[((size_t)f * LATE_A(tiles_per_frame) + (size_t)(y / LATE_A(tile_rows)) * LATE_A(tiles_x) + word) * 2] = 1;   // benign race: every writer stores 1
      if (out_zr && w) out_zr[wo] = make_float2(zmn, zmx);   // only non-zero words have a range; k_ccl_bits reads no other
    }
  }
}


template <bool XY> __global__ __launch_bounds__(256) void k_scene_flow_v4(DevCam c, SfArgs a) { sf_v4_body<XY, false>(c, a); }
template <bool XY> __global__ __launch_bounds__(256) void k_scene_flow_v4i(DevCam c, SfArgs a, SfConsts k) { sf_v4_body<XY, true>(c, a); }

// Even widths that are not a multiple of 4 (the reference's own working resolution is 1242 x 376, detect_with_zed.launch:10):
// rows are 8-byte aligned, so thread = 2 consecutive pixels (float2 loads / stores, one float4 of flow), wave = 128 px.
template <bool INL>
__device__ __forceinline__ void sf_v2_body(const DevCam &c, const SfArgs &a) {
  const int lane = threadIdx.x;
  const int x0 = (blockIdx.x * 64 + lane) * 2;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  const bool inb = (x0 < c.W) && (y < c.H);
  const size_t fN = (size_t)f * ((size_t)c.W * c.H);
  const uint32_t pix = (uint32_t)y * (uint32_t)c.W + (uint32_t)x0;
  const uint32_t o4 = pix * 4u, o8 = pix * 8u;
  const FrameConst fc = frame_const<INL>(a, f);
  uint32_t two = 0;
  const float qnan = __uint_as_float(0x7fc00000u);
  float zmn = qnan, zmx = qnan;
  if (inb) {
    const float *dprev_f = a.dprev + fN;
    const float2 dn = ld<float2>(a.dnow + fN, o4);
    const float2 dp = ld<float2>(dprev_f, o4);
    const float4 fl = ld<float4>(a.flow + 2 * fN, o8);
    const double ry = c.rayy[y];
    const double2 rx = ld<double2>(c.rayx, (uint32_t)x0 * 8u);
    Px p0, p1;
    PxState s0, s1;
    sf_stage1(c, fc, x0 + 0, y, dn.x, dp.x, fl.x, fl.y, rx.x, ry, p0, s0);
    sf_stage1(c, fc, x0 + 1, y, dn.y, dp.y, fl.z, fl.w, rx.y, ry, p1, s1);
    if (a.x) {                                       // final after stage 1: out before the gathers come back; x, y optional
      st(a.x + fN, o4, make_float2(p0.x, p1.x));
      st(a.y + fN, o4, make_float2(p0.y, p1.y));
    }
    st(a.z + fN, o4, make_float2(p0.z, p1.z));
    const uint32_t W = (uint32_t)c.W;
    const float g0 = ld<float>(dprev_f, ((uint32_t)s0.py * W + (uint32_t)s0.px) * 4u);
    const float g1 = ld<float>(dprev_f, ((uint32_t)s1.py * W + (uint32_t)s1.px) * 4u);
    const double ax0 = ld<double>(c.rayx, (uint32_t)s0.px * 8u), ax1 = ld<double>(c.rayx, (uint32_t)s1.px * 8u);
    const double ay0 = ld<double>(c.rayy, (uint32_t)s0.py * 8u), ay1 = ld<double>(c.rayy, (uint32_t)s1.py * 8u);
    PxWarp w0, w1;
    sf_stage2a(c, fc, s0, g0, ax0, ay0, p0, w0);
    sf_stage2a(c, fc, s1, g1, ax1, ay1, p1, w1);
    if (__any(w0.todo | w1.todo)) {
      sf_stage2b(c, fc, s0, w0, p0);
      sf_stage2b(c, fc, s1, w1, p1);
    }
    st(a.vx + fN, o4, make_float2(p0.vx, p1.vx));
    st(a.vy + fN, o4, make_float2(p0.vy, p1.vy));
    st(a.vz + fN, o4, make_float2(p0.vz, p1.vz));
    if (a.aos) {
      float4 *q = a.aos + 2 * fN;
      const uint32_t o32 = pix * 32u;
      st(q, o32, make_float4(p0.x, p0.y, p0.z, 0.f)); st(q, o32 + 16u, make_float4(p0.vx, p0.vy, p0.vz, 0.f));
      st(q, o32 + 32u, make_float4(p1.x, p1.y, p1.z, 0.f)); st(q, o32 + 48u, make_float4(p1.vx, p1.vy, p1.vz, 0.f));
    }
    if (a.depth) st(a.depth + fN, o4, make_float2(p0.depth, p1.depth));
    if (a.sflow) st(a.sflow + 2 * fN, o8, make_float4(p0.s0, p0.s1, p1.s0, p1.s1));
    two = (p0.dyn ? 1u : 0u) | (p1.dyn ? 2u : 0u);
    zmn = sf_min(p0.dyn ? p0.z : qnan, p1.dyn ? p1.z : qnan); zmx = sf_max(p0.dyn ? p0.z : qnan, p1.dyn ? p1.z : qnan);
  }
  if (a.mask) {   // wave-uniform branch; lanes 0..31 build word 2*blockIdx.x, lanes 32..63 the next one
    uint32_t v = two << (2 * (lane & 15));
    v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 4); v |= __shfl_xor(v, 8);
    const uint32_t hi = __shfl_down(v, 16);
    const int word = blockIdx.x * 2 + (lane >> 5);
    if (a.zrange) group_minmax<32>(zmn, zmx);
    if ((lane & 31) == 0 && y < c.H && word < c.mask_words) {
      const uint64_t wd = (uint64_t)v | ((uint64_t)hi << 32);
      const size_t wo = ((size_t)f * c.H + y) * c.mask_words + word;
      a.mask[wo] = wd;
      if (a.tilehdr && wd) a.tilehdr[((size_t)f * a.tiles_per_frame + (size_t)(y / a.tile_rows) * a.tiles_x + word) * 2] = 1;
      if (a.zrange && wd) a.zrange[wo] = make_float2(zmn, zmx);
    }
  }
}

__global__ __launch_bounds__(256) void k_scene_flow_v2(DevCam c, SfArgs a) { sf_v2_body<false>(c, a); }
__global__ __launch_bounds__(256) void k_scene_flow_v2i(DevCam c, SfArgs a, SfConsts k) { sf_v2_body<true>(c, a); }

// Scalar kernel for odd widths: thread = 1 pixel, wave = 64 px of one row.
template <bool INL>
__device__ __forceinline__ void sf_v1_body(const DevCam &c, const SfArgs &a) {
  const int lane = threadIdx.x;
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  const bool inb = (x < c.W) && (y < c.H);
  const size_t N = (size_t)c.W * c.H;
  const size_t i = (size_t)f * N + (size_t)y * c.W + x;
  const FrameConst fc = frame_const<INL>(a, f);
  bool dyn = false;
  const float qnan = __uint_as_float(0x7fc00000u);
  float zmn = qnan, zmx = qnan;
  if (inb) {
    Px p;
    sf_pixel(c, fc, a.dprev + (size_t)f * N, x, y, a.dnow[i], a.dprev[i], a.flow[2 * i], a.flow[2 * i + 1], c.rayx[x], c.rayy[y], p);
    if (a.x) { a.x[i] = p.x; a.y[i] = p.y; }
    a.z[i] = p.z; a.vx[i] = p.vx; a.vy[i] = p.vy; a.vz[i] = p.vz;
    if (a.aos) { a.aos[2 * i] = make_float4(p.x, p.y, p.z, 0.f); a.aos[2 * i + 1] = make_float4(p.vx, p.vy, p.vz, 0.f); }
    if (a.depth) a.depth[i] = p.depth;
    if (a.sflow) { a.sflow[2 * i] = p.s0; a.sflow[2 * i + 1] = p.s1; }
    dyn = p.dyn;
    zmn = zmx = dyn ? p.z : qnan;
  }
  if (a.mask) {
    const uint64_t w = __ballot(dyn);
    if (a.zrange) group_minmax<64>(zmn, zmx);
    if (lane == 0 && y < c.H && blockIdx.x < (unsigned)c.mask_words) {
      const size_t wo = ((size_t)f * c.H + y) * c.mask_words + blockIdx.x;
      a.mask[wo] = w;
      if (a.tilehdr && w) a.tilehdr[((size_t)f * a.tiles_per_frame + (size_t)(y / a.tile_rows) * a.tiles_x + blockIdx.x) * 2] = 1;
      if (a.zrange && w) a.zrange[wo] = make_float2(zmn, zmx);
    }
  }
}
__global__ __launch_bounds__(256) void k_scene_flow_v1(DevCam c, SfArgs a) { sf_v1_body<false>(c, a); }
__global__ __launch_bounds__(256) void k_scene_flow_v1i(DevCam c, SfArgs a, SfConsts k) { sf_v1_body<true>(c, a); }

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

// toDepthImage alone (disparity_image_processor.cpp:105-120): construct() publishes ~depth whenever disparity_now exists, also on
// frames where the flow / the previous disparity / the transform is missing and nothing else is published
// (scene_flow_constructor.cpp:110-123) — the fused kernel does not run on those.  depth = z of getPoint3D, NaN where it fails.
__global__ __launch_bounds__(256) void k_depth(DevCam c, size_t n, const float *__restrict__ dnow, float *__restrict__ depth) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float d = dnow[i];
  const bool ok = disp_in_range(c, d) & !(d == 0.0f);
  depth[i] = ok ? c.fT / d : __uint_as_float(0x7fc00000u);
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

void launch_scene_flow(const DevCam &c, const SfArgs &a, int frames, const FrameConst *inline_consts, hipStream_t s) {
  dim3 block(64, 4, 1);
  SfConsts k;
  if (inline_consts) for (int f = 0; f < frames && f < kSfInlineFrames; f++) k.v[f] = inline_consts[f];   // (frames <= kSfInlineFrames)
  if ((c.W & 3) == 0) {
    dim3 grid((c.W / 4 + 63) / 64, (c.H + 3) / 4, frames);
    if (inline_consts) {
      if (a.x) hipLaunchKernelGGL(k_scene_flow_v4i<true>, grid, block, 0, s, c, a, k);
      else hipLaunchKernelGGL(k_scene_flow_v4i<false>, grid, block, 0, s, c, a, k);
    } else if (a.x) hipLaunchKernelGGL(k_scene_flow_v4<true>, grid, block, 0, s, c, a);
    else hipLaunchKernelGGL(k_scene_flow_v4<false>, grid, block, 0, s, c, a);
  } else if ((c.W & 1) == 0) {
    dim3 grid((c.W / 2 + 63) / 64, (c.H + 3) / 4, frames);
    if (inline_consts) hipLaunchKernelGGL(k_scene_flow_v2i, grid, block, 0, s, c, a, k);
    else hipLaunchKernelGGL(k_scene_flow_v2, grid, block, 0, s, c, a);
  } else {
    dim3 grid((c.W + 63) / 64, (c.H + 3) / 4, frames);
    if (inline_consts) hipLaunchKernelGGL(k_scene_flow_v1i, grid, block, 0, s, c, a, k);
    else hipLaunchKernelGGL(k_scene_flow_v1, grid, block, 0, s, c, a);
  }
}

void launch_dynamic_mask(const DevCam &c, int frames, const float *vx, const float *vy, const float *vz, uint64_t *mask,
                         hipStream_t s) {
  dim3 block(64, 4, 1), grid((c.W + 63) / 64, (c.H + 3) / 4, frames);
  hipLaunchKernelGGL(k_dynamic_mask, grid, block, 0, s, c, vx, vy, vz, mask);
}

__global__ __launch_bounds__(256) void k_copy_words(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
void launch_copy_words(const unsigned long long *src, unsigned long long *dst, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_words, dim3((unsigned)std::min<size_t>((n + 255) / 256, 64)), dim3(256), 0, s, src, dst, n);
}

void launch_depth(const DevCam &c, int frames, const float *dnow, float *depth, hipStream_t s) {
  const size_t n = (size_t)frames * c.W * c.H;
  hipLaunchKernelGGL(k_depth, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c, n, dnow, depth);
}

void launch_pack(size_t n, const float *x, const float *y, const float *z, const float *vx, const float *vy, const float *vz,
                 void *aos, hipStream_t s) {
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, x, y, z, vx, vy, vz, (float4 *)aos);
}
void launch_unpack(size_t n, const void *aos, float *x, float *y, float *z, float *vx, float *vy, float *vz, hipStream_t s) {
  hipLaunchKernelGGL(k_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, (const float4 *)aos, x, y, z, vx, vy, vz);
}
