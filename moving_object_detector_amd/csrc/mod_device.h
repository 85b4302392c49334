// mod_device.h — device-side constants and small helpers shared by the gfx950 kernels.
// Written for CDNA4 only: 64-lane wavefronts are assumed throughout (ballots are 64-bit, tiles are 64 px wide).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MOD_WAVE 64

// Ablation switches for timing experiments (tools/ablate.sh): compiled in only with `make ABLATE=1`; a product build carries
// none of them.  MOD_ABLATE(c, bits) is true when the diagnostic build runs with one of `bits` set in MOD_DEBUG — the results
// are then wrong on purpose (a phase is skipped).
#ifdef MOD_ABLATION
#define MOD_ABLATE(c, bits) (((c).debug & (bits)) != 0)
#else
#define MOD_ABLATE(c, bits) false
#endif

// Index assertions of the diagnostic build `make CHECKED=1` (libmod_sf_checked.so, tests/test_gpu_checked_build.py): every index
// the cluster kernels derive from DATA IN MEMORY (parent entries, link requests, member slots, cluster tables) is checked
// before it is used; a violation is counted in ClArgs.dbg[kDbgCheckBase + code] and the access is skipped instead of faulting the GPU.
// A product build compiles the conditions away.  Codes: 0 link request pixel, 1 link roots, 2 merge root, 3 root list slot,
// 4 select root, 5 final: parent entry, 6 final: tile-root cell, 7 final: label, 8 final: member slot, 9 median: segment,
// 10 median: member pixel, 11 ties: member slot, 12 tile: link-request slot, 13 tile: root pixel, 14 ties: members placed != cluster size,
// 15 select: more clusters than max_objects, 16 scene flow: a late kernarg load disagrees with the argument it stands for.
// dbg layout: [0, 64) cycle counters of the PHASE_COUNTERS build, [64, 96) one counter per code.
constexpr int kDbgCheckBase = 64, kDbgWords = 96;
#ifdef MOD_CHECKED
#define MOD_CHECK(a, cond, code) ((cond) ? true : (atomicAdd(&(a).dbg[kDbgCheckBase + (code)], 1ull), false))
#else
#define MOD_CHECK(a, cond, code) true
#endif

// Camera / parameter block as the kernels see it.  Everything that the reference computes per pixel but that only
// depends on the camera (F32 product f*T, pixel rays, threshold conversions) is computed ONCE on the host with the same
// IEEE operations and uploaded, so the per-pixel results stay bit-identical to the reference expressions.
struct DevCam {
  int32_t W, H;
  int32_t mask_words;      // ceil(W/64)
  int32_t n;               // neighbor_distance
  int32_t cluster_size;
  int32_t debug;           // MOD_DEBUG bits of the diagnostic builds (ABLATE=1 / PHASE_COUNTERS=1); ignored by a product build
  float fT;                // F32(f * T)                       disparity_image_processor.cpp:44
  float dmin, dmax;        // min/max_disparity                disparity_image_processor.cpp:25-27
  float flow_th_sq;        // smallest F32 a with sqrtf(a) >= (float)dynamic_flow_diff   scene_flow_constructor.cpp:198
  float speed_th_sq;       // smallest F32 a with (double)sqrtf(a) >= dynamic_speed      clusterer_nodelet.cpp:51
  float depth_th;          // largest  F32 t with (double)t <= depth_diff      (clusterer_nodelet.cpp:194)
  double speed_th_d;       // dynamic_speed itself (object acceptance test, clusterer_nodelet.cpp:176)
  double fx, fy, cx, cy, Tx, Ty;  // project3dToPixel, scene_flow_constructor.cpp:84
  const double *rayx;      // [W] (u - cx - Tx)/fx in F64       projectPixelTo3dRay, disparity_image_processor.cpp:45
  const double *rayy;      // [H] (v - cy - Ty)/fy
};

// Per-frame constants: previous->now isometry (row-major 3x4: rotation | translation) and dt.
struct FrameConst {
  double m[12];
  double dt;
  double pad[3];           // [0]: |coordinate| bound below which the transformed point is certainly finite (0 = never)
                           // [1]: RN(1/dt), [2]: 1 when [1] may replace the division by dt (exact_div.h), else 0
};

// Per-component statistics live in two sparse planes indexed by the root's pixel index (ClArgs.rsize / rkey): member count
// and first_edge_key (min raster index of a member with an up-left edge; after k_select: new label or -1).  RootRec is the
// LDS form of one such record while a tile reduces its components.
struct RootRec {
  int32_t size;
  int32_t key;
};

// Bounding box of a surviving cluster (ordered-uint encodings, f2ord): accumulated by k_final (one 6-lane atomic per wave), turned into the object's
// bounding_box / center by k_median.  32 bytes.
struct ClusterBox {
  uint32_t w[8];       // [0..2] min x, y, z; [3..5] COMPLEMENT of max x, y, z (so that every word is folded with atomicMin); [6..7] pad
};

// Per surviving cluster (after the size filter), in label order.
struct ClusterInfo {
  int32_t comp;        // component id = pixel index of its final root
  int32_t size;
  int32_t offset;      // start of its member segment
  int32_t med_pix;     // pixel index of the member chosen as the median-velocity element
  uint32_t med_bits;   // F32 bits of that member's ||v||
  int32_t ambiguous;   // 1 when members tie on ||v|| with different vectors (introsort-defined choice)
  int32_t pad[2];
};

__device__ __forceinline__ uint32_t f2ord(float f) {   // monotone float -> uint
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Wave-wide min/max: four DPP steps reduce each row of 16 lanes (quad swaps, half-row mirror, row mirror), four
// v_readlane + scalar ops combine the rows.  All 64 lanes must be active.  ~11 instructions, no LDS traffic.
#define MOD_DPP(v, ctrl) (uint32_t) __builtin_amdgcn_update_dpp((int)(v), (int)(v), ctrl, 0xF, 0xF, false)
// value of the lane below (lane 0 keeps its own): one DPP move instead of __shfl_up's ds_bpermute, which is an LDS-pipeline
// round trip.  All 64 lanes must be active.
__device__ __forceinline__ int wave_prev_i32(int v) { return (int)MOD_DPP(v, 0x138); }                    // wave_shr:1
__device__ __forceinline__ float wave_prev_f32(float v) { return __uint_as_float(MOD_DPP(__float_as_uint(v), 0x138)); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  uint32_t t;
  t = MOD_DPP(v, 0xB1); v = t < v ? t : v;     // quad_perm [1,0,3,2]
  t = MOD_DPP(v, 0x4E); v = t < v ? t : v;     // quad_perm [2,3,0,1]
  t = MOD_DPP(v, 0x141); v = t < v ? t : v;    // row_half_mirror
  t = MOD_DPP(v, 0x140); v = t < v ? t : v;    // row_mirror
  const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  uint32_t t;
  t = MOD_DPP(v, 0xB1); v = t > v ? t : v;
  t = MOD_DPP(v, 0x4E); v = t > v ? t : v;
  t = MOD_DPP(v, 0x141); v = t > v ? t : v;
  t = MOD_DPP(v, 0x140); v = t > v ? t : v;
  const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}


// Eigen Vector3f::norm(): sqrt(x^2 + (y^2 + z^2)), every step rounded to F32.
// Plain operators + sqrtf on purpose: the build uses -ffp-contract=off and -fhip-fp32-correctly-rounded-divide-sqrt, whereas
// HIP's __fsqrt_rn() lowers to the *native* (approximate) square root unless OCML_BASIC_ROUNDED_OPERATIONS is defined.
#pragma clang fp contract(off)
__device__ __forceinline__ float sumsq3_f32(float vx, float vy, float vz) {
  const float xx = vx * vx, yy = vy * vy, zz = vz * vz;
  const float s = yy + zz;
  return xx + s;
}
__device__ __forceinline__ float norm3_f32(float vx, float vy, float vz) { return sqrtf(sumsq3_f32(vx, vy, vz)); }
