// sgm.hip — on-GPU disparity estimation, the complete estimator (BASELINE config 5).
//
// Replaces sgm_gpu::SgmGpu::computeDisparity, the un-vendored CUDA estimator the reference calls at
// scene_flow_constructor/src/scene_flow_constructor.cpp:35,267; its output contract is the stereo_msgs/DisparityImage consumed at
// disparity_image_proc/src/disparity_image_processor.cpp:25-27,41-42.  The algorithm (centre-symmetric 9 x 7 census, Hamming
// cost over D <= 128 disparities, 8-path semi-global aggregation with P1 / P2, winner-take-all, median, left-right check) and
// every choice the publication leaves open are stated in oracle/sgm_ref.cpp, which these kernels match bit for bit.
// Kernels: census transform, the eight aggregation paths (matching cost computed on the fly, never stored unless asked),
// winner-take-all over the summed paths (left and right disparity), 3 x 3 median, left-right check.
#include "mod_launch.h"
#include <cstdlib>
#include <string>

namespace {

// Census: block 64 x 4 output pixels, the 72 x 10 input patch staged in LDS (each input byte is used by up to 62 windows).
__global__ __launch_bounds__(256) void k_sgm_census(int W, int H, const uint8_t *__restrict__ img, uint32_t *__restrict__ out) {
  constexpr int TW = 64, TH = 4, PW = TW + 8, PH = TH + 6;
  __shared__ uint8_t t[PH][PW];
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, f = blockIdx.z;
  const size_t fN = (size_t)f * W * H;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int i = tid; i < PW * PH; i += 256) {
    const int py = i / PW, px = i - py * PW, gx = x0 - 4 + px, gy = y0 - 3 + py;
    t[py][px] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? img[fN + (size_t)gy * W + gx] : (uint8_t)0;
  }
  __syncthreads();
  const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
  if (x >= W || y >= H) return;
  uint32_t c = 0;
  if (x >= 4 && x < W - 4 && y >= 3 && y < H - 3) {           // border choice of the oracle: windows that leave the image give 0
    const int cx = threadIdx.x + 4, cy = threadIdx.y + 3;
#pragma unroll
    for (int dy = -3; dy <= 0; dy++)
#pragma unroll
      for (int dx = -4; dx <= 4; dx++) {
        if (dy == 0 && dx >= 0) break;
        c = (c << 1) | (t[cy + dy][cx + dx] >= t[cy - dy][cx - dx] ? 1u : 0u);
      }
  }
  out[fN + (size_t)y * W + x] = c;
}

// One horizontal aggregation path over one image row: ONE wave, lane l owns disparities 2l and 2l + 1; the row's census words sit
// in LDS (the right word of disparity d at column x is word x - d: consecutive lanes read consecutive addresses), the previous
// column's path costs in registers: neighbours d +- 1 come through DPP wave shifts, the minimum over d through a DPP reduction.
//   L(x, d) = C(x, d) + min(L(xp, d), L(xp, d-1) + P1, L(xp, d+1) + P1, min_k L(xp, k) + P2) - min_k L(xp, k)
// One step of a path for the lane's two disparities, held as a PACKED pair of 16-bit numbers (low half: disparity 2 * lane, high
// half: disparity 2 * lane + 1; every quantity of the recurrence stays below 2^14): v_pk_min_u16 / v_pk_add_u16 do both at once,
// the neighbours d - 1 / d + 1 are one DPP wave shift + one v_alignbit each, and the pair leaves as ONE 2-byte store.
// `lp` previous path costs (kSgmNone where the disparity does not exist), `c` matching costs -> new path costs.  All 64 lanes active.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kSgmNone = 0x3fffu;               // "no such disparity": larger than any cost + penalty, and twice it fits 16 bits
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)); }
// minimum over the wave's 128 packed costs: the two halves, then a DPP butterfly inside each row of 16 lanes, then the rows through
// row_bcast (8 VALU instructions; the s_nop are the two wait states a DPP read of a just-written VGPR needs).  Wave-uniform (SGPR).
__device__ __forceinline__ uint32_t sgm_min128(uint32_t lp) {
  uint32_t v = min(lp & 0xffffu, lp >> 16);
  asm("s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// `prev` / `next` are the DPP shift registers, carried from step to step: a wave shift leaves the lane without a source (lane 0 /
// lane 63) untouched, so those lanes keep the kSgmNone they were initialised with and no constant is re-materialised per step.
struct SgmShift { uint32_t prev, next; };
__device__ __forceinline__ SgmShift sgm_shift_init() { return SgmShift{kSgmNone << 16, kSgmNone}; }
__device__ __forceinline__ uint32_t sgm_step(uint32_t lp, uint32_t c, uint32_t p1pk, uint32_t p2, SgmShift &sh) {
  const uint32_t m = sgm_min128(lp);                                  // min over all disparities
  const uint32_t mpk = m | (m << 16), mp2pk = (m + p2) | ((m + p2) << 16);   // scalar unit
  // d - 1: (high half of lane - 1, own low half); d + 1: (own high half, low half of lane + 1); the ends see kSgmNone
  sh.prev = (uint32_t)__builtin_amdgcn_update_dpp((int)sh.prev, (int)lp, 0x138, 0xF, 0xF, false);   // wave_shr:1
  sh.next = (uint32_t)__builtin_amdgcn_update_dpp((int)sh.next, (int)lp, 0x130, 0xF, 0xF, false);   // wave_shl:1
  const uint32_t a = __builtin_amdgcn_alignbit(lp, sh.prev, 16), b = __builtin_amdgcn_alignbit(sh.next, lp, 16);
  const uint32_t best = pk_min(pk_min(lp, mp2pk), pk_add(pk_min(a, b), p1pk));
  return pk_sub(pk_add(c, best), mpk);
}
// A wave-uniform GLOBAL pointer the compiler must keep in SGPRs, so that loads / stores take the "SGPR base + 32-bit lane offset"
// form instead of a 64-bit add per lane (the readfirstlane is a no-op on a value that is already scalar; it stops the optimiser
// from re-associating the lane offset into the base).
#define SGM_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ SGM_GLOBAL uint8_t *sgm_uniform(const void *p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (SGM_GLOBAL uint8_t *)(((uint64_t)hi << 32) | lo);
}

// where a path's costs go: the path's volume, and (stage tests) the matching cost
struct SgmOut {
  uint8_t *L, *C;       // [H][W][D]; C may be null
};
// lanes whose disparities do not exist (d >= D) keep kSgmNone in that half and store nothing
struct SgmLane {
  uint32_t keep, none;  // masks: halves that exist / kSgmNone in the halves that do not
  bool has0, has1;
};
__device__ __forceinline__ SgmLane sgm_lane(int lane, int D) {
  SgmLane s;
  s.has0 = 2 * lane < D; s.has1 = 2 * lane + 1 < D;
  s.keep = (s.has0 ? 0xffffu : 0u) | (s.has1 ? 0xffff0000u : 0u);
  s.none = (s.has0 ? 0u : kSgmNone) | (s.has1 ? 0u : (kSgmNone << 16));
  return s;
}
// FULL: D == 128, every lane owns two disparities: one unconditional 2-byte store at (wave-uniform pixel base) + 2 * lane
template <bool FULL>
__device__ __forceinline__ void sgm_emit(const SgmOut &o, const SgmLane &ln, size_t at, uint32_t lane, uint32_t l, uint32_t c, bool even_d, uint32_t sel) {
  SGM_GLOBAL uint8_t *pl = sgm_uniform(o.L + at), *pc = o.C ? sgm_uniform(o.C + at) : nullptr;
  const uint32_t q = 2 * lane;
  if (FULL || (ln.has1 && even_d)) {                                  // both disparities exist and the pair is 2-byte aligned
    *(SGM_GLOBAL uint16_t *)(pl + q) = (uint16_t)__builtin_amdgcn_perm(0u, l, sel);
    if (pc) *(SGM_GLOBAL uint16_t *)(pc + q) = (uint16_t)__builtin_amdgcn_perm(0u, c, sel);
  } else {
    if (ln.has0) { pl[q] = (uint8_t)(l & 0xffu); if (pc) pc[q] = (uint8_t)(c & 0xffu); }
    if (ln.has1) { pl[q + 1] = (uint8_t)((l >> 16) & 0xffu); if (pc) pc[q + 1] = (uint8_t)((c >> 16) & 0xffu); }
  }
}

// The right census words of a lane's two disparities are neighbours in memory: word x - d1 (d1 = 2 lane + 1) and the one after it
// (x - d0).  They are read as ONE pair at max(x - d1, 0) — never left of the row (a saturating subtraction) — so near the left
// border (x < d1) the pair is (word 0, word 1) and the masked branch below picks word 0 for d0 == x.  Needs W >= 2.
struct SgmPair { uint32_t a, b; };
// matching costs of the lane's two disparities at column x (31 where the disparity leaves the right image); dlast = D - 1
__device__ __forceinline__ uint32_t sgm_cost_pk(uint32_t wl, SgmPair r, int x, int d0, int dlast) {
  uint32_t c = (uint32_t)__popc(wl ^ r.b) + ((uint32_t)__popc(wl ^ r.a) << 16);
  if (x < dlast) {                                                    // wave-uniform: only the first D - 1 columns
    asm volatile("" ::: "memory");                                    // keep this a branch: the other columns skip it
    const uint32_t c0 = x > d0 ? (c & 0xffffu) : x == d0 ? (c >> 16) : 31u;
    const uint32_t c1 = x > d0 ? (c >> 16) : 31u;
    c = c0 | (c1 << 16);
  }
  return c;
}

template <bool RTL, bool FULL>
__global__ __launch_bounds__(64) void k_sgm_path_h(int W, int H, int D, int P1, int P2, const uint32_t *__restrict__ cl,
                                                   const uint32_t *__restrict__ cr, SgmOut out) {
  extern __shared__ uint32_t srow[];                 // [W + 1] right census row, [W] left census row
  const int lane = threadIdx.x, y = blockIdx.x, f = blockIdx.y;
  const size_t row = ((size_t)f * H + y) * W;
  uint32_t *sr = srow, *sl = srow + W + 1;
  for (int i = lane; i < W; i += 64) { sl[i] = cl[row + i]; sr[i] = cr[row + i]; }
  if (lane == 0) sr[W] = 0;                          // W == 1: the pair read at x = 0
  __syncthreads();
  const int d0 = 2 * lane, d1 = 2 * lane + 1, dlast = D - 1;
  const bool even_d = !(D & 1);
  const SgmLane ln = sgm_lane(lane, D);
  const uint32_t p1pk = (uint32_t)P1 | ((uint32_t)P1 << 16);
  uint32_t sel = 0x0c0c0200u;                        // v_perm selector "bytes 0 and 2", pinned to an SGPR outside the loop
  asm volatile("" : "+s"(sel));
  const size_t vol = (size_t)f * H * W * D;
  out.L += vol;
  if (out.C) out.C += vol;
  uint32_t lp = kSgmNone | (kSgmNone << 16);
  // the census words of a step are read one step ahead, unconditionally: their LDS latency hides behind the recurrence of the
  // step before
  auto pair = [&](int xx) { const uint32_t i = __builtin_elementwise_sub_sat((uint32_t)xx, (uint32_t)d1); SgmPair r; r.a = sr[i]; r.b = sr[i + 1]; return r; };
  SgmShift sh = sgm_shift_init();
  int x = RTL ? W - 1 : 0;
  uint32_t wl = sl[x];
  SgmPair r = pair(x);
  for (int step = 0; step < W; step++) {
    const int xn = min(max(RTL ? x - 1 : x + 1, 0), W - 1);
    const uint32_t wln = sl[xn];
    const SgmPair rn = pair(xn);
    const uint32_t c = sgm_cost_pk(wl, r, x, d0, dlast);
    uint32_t l = step > 0 ? sgm_step(lp, c, p1pk, (uint32_t)P2, sh) : c;
    if (!FULL) l = (l & ln.keep) | ln.none;
    sgm_emit<FULL>(out, ln, ((size_t)y * W + x) * D, lane, l, c, even_d, sel);
    lp = l;
    x = RTL ? x - 1 : x + 1;
    wl = wln; r = rn;
  }
}

// Vertical and diagonal paths: one wave per path LINE (direction (RX, RY), RY != 0).  Lines enter through the first row in path
// order (W of them) and, for the diagonals, through the first column in path order (H - 1 more).  Census words come straight from
// L2 / HBM (one 8-byte load per lane, see SgmPair), kPF steps ahead of their use: the recurrence is a dependent chain per step, a
// memory round trip per step would dominate it.
constexpr int kSgmPF = 4;
template <int RX, int RY, bool FULL>
__global__ __launch_bounds__(64) void k_sgm_path_line(int W, int H, int D, int P1, int P2, const uint32_t *__restrict__ cl,
                                                      const uint32_t *__restrict__ cr, SgmOut out) {
  const int lane = threadIdx.x, line = blockIdx.x, f = blockIdx.y;
  int x, y;
  if (line < W) { x = RX < 0 ? W - 1 - line : line; y = RY > 0 ? 0 : H - 1; }   // long lines first (they enter at the far end when RX < 0)
  else { x = RX > 0 ? 0 : W - 1; y = RY > 0 ? line - W + 1 : H - 2 - (line - W); }   // the entry row's own pixel is a row line
  const size_t plane = (size_t)f * H * W, vol = plane * D;
  cl += plane; cr += plane;
  out.L += vol;
  if (out.C) out.C += vol;
  const int d0 = 2 * lane, d1 = 2 * lane + 1, dlast = D - 1;
  const bool even_d = !(D & 1);
  const SgmLane ln = sgm_lane(lane, D);
  const uint32_t p1pk = (uint32_t)P1 | ((uint32_t)P1 << 16);
  uint32_t sel = 0x0c0c0200u;                        // v_perm selector "bytes 0 and 2", pinned to an SGPR outside the loop
  asm volatile("" : "+s"(sel));
  // number of pixels on the line
  int len = RY > 0 ? H - y : y + 1;
  if (RX > 0) len = min(len, W - x);
  if (RX < 0) len = min(len, x + 1);
  auto fetch = [&](int i, uint32_t &wl, SgmPair &r) {                  // census words of the line's i-th pixel (clamped to the line)
    const int k = min(i, len - 1), xx = x + RX * k, yy = y + RY * k;
    const size_t rowp = (size_t)yy * W;
    wl = cl[rowp + xx];
    const SGM_GLOBAL uint32_t *q = (const SGM_GLOBAL uint32_t *)(sgm_uniform(cr + rowp) + 4u * __builtin_elementwise_sub_sat((uint32_t)xx, (uint32_t)d1));
    r.a = q[0]; r.b = q[1];
  };
  uint32_t wl[kSgmPF];
  SgmPair r[kSgmPF];
#pragma unroll
  for (int i = 0; i < kSgmPF; i++) fetch(i, wl[i], r[i]);
  uint32_t lp = kSgmNone | (kSgmNone << 16);
  SgmShift sh = sgm_shift_init();
  for (int i0 = 0; i0 < len; i0 += kSgmPF) {         // wave-uniform
#pragma unroll
    for (int u = 0; u < kSgmPF; u++) {
      const int i = i0 + u;
      if (i >= len) break;                           // wave-uniform
      const int xx = x + RX * i, yy = y + RY * i;
      const uint32_t c = sgm_cost_pk(wl[u], r[u], xx, d0, dlast);
      fetch(i + kSgmPF, wl[u], r[u]);                // slot u is free again: the words of pixel i + kPF
      uint32_t l = i > 0 ? sgm_step(lp, c, p1pk, (uint32_t)P2, sh) : c;
      if (!FULL) l = (l & ln.keep) | ln.none;
      sgm_emit<FULL>(out, ln, ((size_t)yy * W + xx) * D, lane, l, c, even_d, sel);
      lp = l;
    }
  }
}

// FOUR path lines per wave (D == 128): lane = (line q of the wave's four, slot t of 16), slot t owns the EIGHT disparities
// 8t .. 8t + 7 as four packed pairs.  Against one line per wave (lane = two disparities, above):
//   * the minimum over the 128 disparities is 3 packed minima inside the lane + 4 DPP steps inside the 16-lane row (quad swaps, half
//     mirror, mirror) and stays in a vector register — no row_bcast steps, no v_readlane, no scalar round trip per step;
//   * six of the eight d - 1 / d + 1 neighbours are register moves (v_alignbit between the lane's own pairs); only the pair ends
//     cross a lane (one row_shr:1 and one row_shl:1 per step);
//   * one 8-byte store per lane and step; the lane's eight right census words are eight CONSECUTIVE words (two 16-byte loads).
// ~70 vector instructions per step of four lines (17 per line step, one-line kernels: ~30) and a shorter dependent chain.
// The four lines of a wave are neighbours (rows / columns / diagonals next to each other); they may differ in length by a few
// pixels: a line that has ended keeps computing on its last pixel and stores nothing.
constexpr int kSgmPrefetch = 2;   // census slots in flight per line (steps of prefetch; 1 .. 4 measured, no difference)
typedef uint32_t sgm_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t sgm_u2 __attribute__((ext_vector_type(2)));
// UNIFORM: the wave's four lines exist and have one length (rows / columns of an image whose height / width is a multiple of 4):
// every step stores unconditionally, so the compiler can count the memory operations in flight and the wait in front of a census
// slot leaves the younger slot's loads outstanding (a conditional store makes it wait for everything).  COST: also emit the
// matching cost (stage entry point only).
template <int RX, int RY, bool UNIFORM, bool COST>
__device__ __forceinline__ void sgm_path_q_body(int W, int H, int P1, int P2, const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr,
                                                SgmOut out, int blk, int f) {
  constexpr int D = 128, kPF = kSgmPrefetch;
  const int lane = threadIdx.x, q = lane >> 4, t = lane & 15;
  const int nlines = RY == 0 ? H : (RX == 0 ? W : W + H - 1);
  const int line = blk * 4 + q;
  int x = 0, y = 0, len = 0;
  if (line < nlines) {
    if (RY == 0) { y = line; x = RX > 0 ? 0 : W - 1; len = W; }
    else {
      if (line < W) { x = RX < 0 ? W - 1 - line : line; y = RY > 0 ? 0 : H - 1; }
      else { x = RX > 0 ? 0 : W - 1; y = RY > 0 ? line - W + 1 : H - 2 - (line - W); }
      len = RY > 0 ? H - y : y + 1;
      if (RX > 0) len = min(len, W - x);
      if (RX < 0) len = min(len, x + 1);
    }
  }
  const int maxlen = max(max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 16)),
                         max(__builtin_amdgcn_readlane(len, 32), __builtin_amdgcn_readlane(len, 48)));
  const size_t plane = (size_t)f * H * W, vol = plane * D;
  SGM_GLOBAL uint8_t *cl_b = sgm_uniform(cl + plane), *cr_b = sgm_uniform(cr + plane);
  SGM_GLOBAL uint8_t *outL = sgm_uniform(out.L + vol), *outC = COST ? sgm_uniform(out.C + vol) : nullptr;
  const uint32_t p1pk = (uint32_t)P1 | ((uint32_t)P1 << 16), p2pk = (uint32_t)P2 | ((uint32_t)P2 << 16);
  const int dbase = 8 * t;
  // census words of the line's k-th pixel (clamped to the line): the left word, and the eight right words x - dbase - 7 .. x - dbase
  // (word j of the window belongs to disparity dbase + 7 - j).  Where a window would start left of the row (x < 127 somewhere in the
  // wave) every word is fetched on its own at max(x - d, 0): such disparities do not exist and get the border cost below.
  struct Slot { uint32_t wl; uint32_t r[8]; };
  auto fetch = [&](int k, Slot &sl) {
    const int kk = max(min(k, len - 1), 0), xx = x + RX * kk, yy = y + RY * kk;
    const uint32_t rowo = (uint32_t)(yy * W) * 4u;
    sl.wl = *(const SGM_GLOBAL uint32_t *)(cl_b + (rowo + (uint32_t)xx * 4u));
    // ALWAYS the same three loads (the compiler can then count them: the wait before a slot's use leaves the younger slot's loads
    // in flight).  Near the left border the window starts left of the row — in the row above, or (first row of the plane) up to
    // 127 words before the plane: the caller guarantees those bytes are readable; the words read there belong to disparities
    // that do not exist and are replaced by the border cost.
    const SGM_GLOBAL sgm_u4 *w = (const SGM_GLOBAL sgm_u4 *)(cr_b + (int32_t)(rowo + (uint32_t)(xx - dbase - 7) * 4u));
    const sgm_u4 a = w[0], b = w[1];
    sl.r[0] = a.x; sl.r[1] = a.y; sl.r[2] = a.z; sl.r[3] = a.w; sl.r[4] = b.x; sl.r[5] = b.y; sl.r[6] = b.z; sl.r[7] = b.w;
  };
  Slot slot[kPF];
#pragma unroll
  for (int u = 0; u < kPF; u++) fetch(u, slot[u]);
  uint32_t lp[4] = {0, 0, 0, 0};
  uint32_t shp = kSgmNone << 16, shn = kSgmNone;       // DPP shift registers: the row's end lanes keep "no such disparity"
  for (int i0 = 0; i0 < maxlen; i0 += kPF) {            // wave-uniform
#pragma unroll
    for (int u = 0; u < kPF; u++) {
      const int i = i0 + u;
      if (i >= maxlen) break;                           // wave-uniform
      const int kk = max(min(i, len - 1), 0), xx = x + RX * kk, yy = y + RY * kk;
      // matching costs of the lane's eight disparities (31 where the disparity leaves the right image)
      uint32_t c[4];
#pragma unroll
      for (int j = 0; j < 4; j++)
        c[j] = (uint32_t)__popc(slot[u].wl ^ slot[u].r[7 - 2 * j]) + ((uint32_t)__popc(slot[u].wl ^ slot[u].r[6 - 2 * j]) << 16);
      if (!__all(xx >= D - 1)) {                        // wave-uniform: only near the left border
        const int nvalid = xx - dbase + 1;              // disparities dbase .. dbase + nvalid - 1 exist (d <= x)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t lo = 2 * j < nvalid ? (c[j] & 0xffffu) : 31u, hi = 2 * j + 1 < nvalid ? (c[j] >> 16) : 31u;
          c[j] = lo | (hi << 16);
        }
      }
      fetch(i + kPF, slot[u]);                          // the slot is free again
      uint32_t l[4];
      if (i > 0) {
        // minimum over the line's 128 path costs: inside the lane, then a butterfly over the row's 16 lanes
        uint32_t m = pk_min(pk_min(lp[0], lp[1]), pk_min(lp[2], lp[3]));
        m = min(m & 0xffffu, m >> 16);
        asm("s_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
            : "+v"(m));
        const uint32_t mpk = m | (m << 16), mp2pk = pk_add(mpk, p2pk);
        shp = (uint32_t)__builtin_amdgcn_update_dpp((int)shp, (int)lp[3], 0x111, 0xF, 0xF, false);   // row_shr:1: the slot below's last pair
        shn = (uint32_t)__builtin_amdgcn_update_dpp((int)shn, (int)lp[0], 0x101, 0xF, 0xF, false);   // row_shl:1: the slot above's first pair
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t below = j == 0 ? shp : lp[j - 1], above = j == 3 ? shn : lp[j + 1];
          const uint32_t a = __builtin_amdgcn_alignbit(lp[j], below, 16), b = __builtin_amdgcn_alignbit(above, lp[j], 16);   // d - 1, d + 1
          const uint32_t best = pk_min(pk_min(lp[j], mp2pk), pk_add(pk_min(a, b), p1pk));
          l[j] = pk_sub(pk_add(c[j], best), mpk);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) l[j] = c[j];
      }
      if (UNIFORM || i < len) {                         // lines that have ended store nothing
        const uint32_t at = (uint32_t)(yy * W + xx) * (uint32_t)D + (uint32_t)dbase;
        sgm_u2 v;
        v.x = __builtin_amdgcn_perm(l[1], l[0], 0x06040200u); v.y = __builtin_amdgcn_perm(l[3], l[2], 0x06040200u);
        __builtin_nontemporal_store(v, (SGM_GLOBAL sgm_u2 *)(outL + at));
        if (COST) {
          v.x = __builtin_amdgcn_perm(c[1], c[0], 0x06040200u); v.y = __builtin_amdgcn_perm(c[3], c[2], 0x06040200u);
          *(SGM_GLOBAL sgm_u2 *)(outC + at) = v;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) lp[j] = l[j];
    }
  }
}

template <int RX, int RY, bool UNIFORM, bool COST>
__global__ __launch_bounds__(64) void k_sgm_path_q(int W, int H, int P1, int P2, const uint32_t *__restrict__ cl,
                                                   const uint32_t *__restrict__ cr, SgmOut out) {
  sgm_path_q_body<RX, RY, UNIFORM, COST>(W, H, P1, P2, cl, cr, out, (int)blockIdx.x, (int)blockIdx.y);
}

// All aggregation paths of a group of frames in ONE launch: blockIdx.x runs over the four-line blocks of direction 0, then 1, ...
// (the rows first: they are the longest lines, 1280 steps at 720p).  One grid instead of eight kernels on eight streams: the HIP
// runtime maps streams onto 4 hardware queues by default, so only four of the eight path kernels ever ran side by side (kernel
// trace of round 3) and the short ones queued behind the long ones; inside one grid the dispatcher fills every free wave slot
// with whatever block is next.  Path i writes its volume at out.L + i * path_stride.
template <bool UH, bool UV>
__global__ __launch_bounds__(64) void k_sgm_paths_all(int W, int H, int P1, int P2, int paths, size_t path_stride, const uint32_t *__restrict__ cl,
                                                      const uint32_t *__restrict__ cr, uint8_t *__restrict__ L) {
  const int nh = (H + 3) / 4, nv = (W + 3) / 4, nd = (W + H - 1 + 3) / 4;
  int blk = (int)blockIdx.x;
  const int f = (int)blockIdx.y;
  SgmOut o{L, nullptr};
  // numbering of oracle/sgm_ref.cpp: 0 (+1,0) 1 (-1,0) 2 (0,+1) 3 (0,-1) 4 (+1,+1) 5 (-1,-1) 6 (-1,+1) 7 (+1,-1)
  if (blk < nh) { sgm_path_q_body<1, 0, UH, false>(W, H, P1, P2, cl, cr, o, blk, f); return; }
  blk -= nh; o.L += path_stride;
  if (blk < nh) { sgm_path_q_body<-1, 0, UH, false>(W, H, P1, P2, cl, cr, o, blk, f); return; }
  blk -= nh; o.L += path_stride;
  if (blk < nv) { sgm_path_q_body<0, 1, UV, false>(W, H, P1, P2, cl, cr, o, blk, f); return; }
  blk -= nv; o.L += path_stride;
  if (blk < nv) { sgm_path_q_body<0, -1, UV, false>(W, H, P1, P2, cl, cr, o, blk, f); return; }
  if (paths <= 4) return;
  blk -= nv; o.L += path_stride;
  if (blk < nd) { sgm_path_q_body<1, 1, false, false>(W, H, P1, P2, cl, cr, o, blk, f); return; }
  blk -= nd; o.L += path_stride;
  if (blk < nd) { sgm_path_q_body<-1, -1, false, false>(W, H, P1, P2, cl, cr, o, blk, f); return; }
  blk -= nd; o.L += path_stride;
  if (blk < nd) { sgm_path_q_body<-1, 1, false, false>(W, H, P1, P2, cl, cr, o, blk, f); return; }
  blk -= nd; o.L += path_stride;
  sgm_path_q_body<1, -1, false, false>(W, H, P1, P2, cl, cr, o, blk, f);
}

// Winner-take-all over the sum of the path volumes, one workgroup (4 waves) per image row.  A wave takes every fourth pixel:
// lane l owns disparities 2l and 2l + 1 (one 2-byte load per path volume), sums the paths, and
//   left  disparity of x : first minimum of S(x, .)              -> DPP minimum over keys (S << 8 | d): the smallest d wins ties
//   right disparity of x': first minimum of S(x' + d, d), x' + d < W -> the pixel x contributes its S(x, d) to x' = x - d with an LDS
//                          atomicMin on the same keys (one row of keys in LDS): the cost volume is read ONCE, coalesced, instead of
//                          gathering a diagonal per pixel
__global__ __launch_bounds__(256) void k_sgm_wta(int W, int H, int D, int paths, size_t path_stride, const uint8_t *__restrict__ Lv,
                                                 uint8_t *__restrict__ dl, uint8_t *__restrict__ dr) {
  extern __shared__ uint32_t rkey[];                 // [W]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, y = blockIdx.x, f = blockIdx.y;
  const size_t plane = (size_t)f * W * H;
  Lv += plane * D; dl += plane; dr += plane;
  for (int i = threadIdx.x; i < W; i += 256) rkey[i] = 0xffffffffu;
  __syncthreads();
  const int d0 = 2 * lane, d1 = d0 + 1;
  for (int x = wv; x < W; x += 4) {
    const size_t at = ((size_t)y * W + x) * D;
    uint32_t s0 = 0, s1 = 0;
    if (d0 < D) {
      for (int p = 0; p < paths; p++) {
        const uint8_t *q = Lv + (size_t)p * path_stride + at + d0;
        if (d1 < D && !(D & 1)) { const uint32_t v = *reinterpret_cast<const uint16_t *>(q); s0 += v & 255u; s1 += v >> 8; }   // even D: 2-byte aligned
        else { s0 += q[0]; if (d1 < D) s1 += q[1]; }
      }
    }
    const uint32_t k0 = d0 < D ? ((s0 << 8) | (uint32_t)d0) : 0xffffffffu, k1 = d1 < D ? ((s1 << 8) | (uint32_t)d1) : 0xffffffffu;
    const uint32_t kl = wave_min_u32(min(k0, k1));
    if (lane == 0) dl[(size_t)y * W + x] = (uint8_t)(kl & 255u);
    if (d0 < D && x - d0 >= 0) atomicMin(&rkey[x - d0], k0);
    if (d1 < D && x - d1 >= 0) atomicMin(&rkey[x - d1], k1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < W; i += 256) dr[(size_t)y * W + i] = (uint8_t)(rkey[i] & 255u);
}

// The same for disparity counts that are multiples of 16 (the default 128 is): lane = (pixel of the wave's group of 8, 16 consecutive
// disparities) — one 16-byte load per path and lane, a wave reads 1 KB contiguous per path; the path costs are summed as packed pairs
// of 16-bit numbers (sums stay below 8 x 128), the minimum over a pixel's 8 lanes is three DPP steps.
__global__ __launch_bounds__(256) void k_sgm_wta16(int W, int H, int D, int paths, size_t path_stride, const uint8_t *__restrict__ Lv,
                                                   uint8_t *__restrict__ dl, uint8_t *__restrict__ dr) {
  extern __shared__ uint32_t rkey[];                 // [W]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, y = blockIdx.x, f = blockIdx.y;
  const size_t plane = (size_t)f * W * H;
  Lv += plane * D; dl += plane; dr += plane;
  for (int i = threadIdx.x; i < W; i += 256) rkey[i] = 0xffffffffu;
  __syncthreads();
  const int sub = lane & 7, px = lane >> 3, dbase = sub * 16;
  const bool has = dbase < D;                        // D / 16 lanes of a pixel carry disparities
  for (int x0 = wv * 8; x0 < W; x0 += 32) {
    const int x = x0 + px;
    const bool live = has && x < W;
    uint32_t e[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};          // packed sums: e[j] = (byte 0 | byte 2 << 16) of word j, o[j] = (byte 1 | byte 3 << 16)
    if (live) {
      const uint8_t *q = Lv + ((size_t)y * W + x) * D + dbase;
      for (int p = 0; p < paths; p++) {
        typedef uint32_t wta_u4 __attribute__((ext_vector_type(4)));
        const wta_u4 wv_ = __builtin_nontemporal_load(reinterpret_cast<const wta_u4 *>(q + (size_t)p * path_stride));
        uint4 w; w.x = wv_.x; w.y = wv_.y; w.z = wv_.z; w.w = wv_.w;
        e[0] += w.x & 0x00ff00ffu; o[0] += (w.x >> 8) & 0x00ff00ffu;
        e[1] += w.y & 0x00ff00ffu; o[1] += (w.y >> 8) & 0x00ff00ffu;
        e[2] += w.z & 0x00ff00ffu; o[2] += (w.z >> 8) & 0x00ff00ffu;
        e[3] += w.w & 0x00ff00ffu; o[3] += (w.w >> 8) & 0x00ff00ffu;
      }
    }
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t d = (uint32_t)(dbase + 4 * j);
      const uint32_t k0 = ((e[j] & 0xffffu) << 8) | d, k1 = ((o[j] & 0xffffu) << 8) | (d + 1), k2 = ((e[j] >> 16) << 8) | (d + 2), k3 = ((o[j] >> 16) << 8) | (d + 3);
      if (live) {
        best = min(best, min(min(k0, k1), min(k2, k3)));
        const int xr = x - (int)d;
        if (xr >= 0) atomicMin(&rkey[xr], k0);
        if (xr >= 1) atomicMin(&rkey[xr - 1], k1);
        if (xr >= 2) atomicMin(&rkey[xr - 2], k2);
        if (xr >= 3) atomicMin(&rkey[xr - 3], k3);
      }
    }
    // minimum over the 8 lanes of the pixel: quad swaps, then the mirror of the half row (lanes 0..7 <-> 7..0)
    uint32_t t;
    t = MOD_DPP(best, 0xB1); best = min(best, t);
    t = MOD_DPP(best, 0x4E); best = min(best, t);
    t = MOD_DPP(best, 0x141); best = min(best, t);
    if (sub == 0 && x < W) dl[(size_t)y * W + x] = (uint8_t)(best & 255u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < W; i += 256) dr[(size_t)y * W + i] = (uint8_t)(rkey[i] & 255u);
}

// 3 x 3 median of a uint8 map (border pixels keep their value): exact 9-element selection network
__device__ __forceinline__ void srt(int &a, int &b) { const int t = min(a, b); b = max(a, b); a = t; }
__global__ __launch_bounds__(256) void k_sgm_median3(int W, int H, const uint8_t *__restrict__ in, uint8_t *__restrict__ out) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= W || y >= H) return;
  in += (size_t)blockIdx.z * W * H; out += (size_t)blockIdx.z * W * H;
  const size_t p = (size_t)y * W + x;
  if (x == 0 || y == 0 || x == W - 1 || y == H - 1) { out[p] = in[p]; return; }
  int v0 = in[p - W - 1], v1 = in[p - W], v2 = in[p - W + 1], v3 = in[p - 1], v4 = in[p], v5 = in[p + 1], v6 = in[p + W - 1], v7 = in[p + W], v8 = in[p + W + 1];
  srt(v1, v2); srt(v4, v5); srt(v7, v8); srt(v0, v1); srt(v3, v4); srt(v6, v7); srt(v1, v2); srt(v4, v5); srt(v7, v8);
  srt(v0, v3); srt(v5, v8); srt(v4, v7); srt(v3, v6); srt(v1, v4); srt(v2, v5); srt(v4, v7); srt(v4, v2); srt(v6, v4); srt(v4, v2);
  out[p] = (uint8_t)v4;
}

// left-right consistency check -> DisparityImage pixels (float, -1 = invalid)
__global__ __launch_bounds__(256) void k_sgm_lr(int W, int H, int lr_check, const uint8_t *__restrict__ dl, const uint8_t *__restrict__ dr,
                                                float *__restrict__ disp) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= W || y >= H) return;
  const size_t plane = (size_t)blockIdx.z * W * H;
  dl += plane; dr += plane; disp += plane;
  const size_t p = (size_t)y * W + x;
  const int d = dl[p];
  bool ok = true;
  if (lr_check) ok = x - d >= 0 && abs((int)dr[p - min(d, x)] - d) <= 1;
  disp[p] = ok ? (float)d : -1.0f;
}

}  // namespace

void launch_sgm_census(int W, int H, int frames, const uint8_t *img, uint32_t *out, hipStream_t s) {
  hipLaunchKernelGGL(k_sgm_census, dim3((W + 63) / 64, (H + 3) / 4, frames), dim3(64, 4, 1), 0, s, W, H, img, out);
}

void launch_sgm_path(int W, int H, int frames, int D, int P1, int P2, int direction, const uint32_t *cl, const uint32_t *cr,
                     uint8_t *L, uint8_t *cost, bool right_plane_padded, hipStream_t s) {
  SgmOut o{L, cost};
  const size_t lds = ((size_t)2 * W + 1) * sizeof(uint32_t);
  const dim3 b(64);
  const dim3 gh(H, frames), gv(W, frames), gd(W + H - 1, frames);
#define SGM_PATH(FULL)                                                                                                              \
  switch (direction) { /* numbering of oracle/sgm_ref.cpp: 0 (+1,0) 1 (-1,0) 2 (0,+1) 3 (0,-1) 4 (+1,+1) 5 (-1,-1) 6 (-1,+1) 7 (+1,-1) */ \
    case 0: hipLaunchKernelGGL((k_sgm_path_h<false, FULL>), gh, b, lds, s, W, H, D, P1, P2, cl, cr, o); break;                      \
    case 1: hipLaunchKernelGGL((k_sgm_path_h<true, FULL>), gh, b, lds, s, W, H, D, P1, P2, cl, cr, o); break;                       \
    case 2: hipLaunchKernelGGL((k_sgm_path_line<0, 1, FULL>), gv, b, 0, s, W, H, D, P1, P2, cl, cr, o); break;                      \
    case 3: hipLaunchKernelGGL((k_sgm_path_line<0, -1, FULL>), gv, b, 0, s, W, H, D, P1, P2, cl, cr, o); break;                     \
    case 4: hipLaunchKernelGGL((k_sgm_path_line<1, 1, FULL>), gd, b, 0, s, W, H, D, P1, P2, cl, cr, o); break;                      \
    case 5: hipLaunchKernelGGL((k_sgm_path_line<-1, -1, FULL>), gd, b, 0, s, W, H, D, P1, P2, cl, cr, o); break;                    \
    case 6: hipLaunchKernelGGL((k_sgm_path_line<-1, 1, FULL>), gd, b, 0, s, W, H, D, P1, P2, cl, cr, o); break;                     \
    default: hipLaunchKernelGGL((k_sgm_path_line<1, -1, FULL>), gd, b, 0, s, W, H, D, P1, P2, cl, cr, o); break;                    \
  }
  // D == 128 (the published configuration): four lines per wave; the one-line kernels serve every other disparity count
  if (D == 128 && right_plane_padded) {
    const dim3 qh((H + 3) / 4, frames), qv((W + 3) / 4, frames), qd((W + H - 1 + 3) / 4, frames);
    const bool uh = (H & 3) == 0, uv = (W & 3) == 0;
#define SGM_Q(RX, RY, UNI, GRID)                                                                                            \
    do {                                                                                                                    \
      if (cost) { if (UNI) hipLaunchKernelGGL((k_sgm_path_q<RX, RY, true, true>), GRID, b, 0, s, W, H, P1, P2, cl, cr, o);   \
                  else hipLaunchKernelGGL((k_sgm_path_q<RX, RY, false, true>), GRID, b, 0, s, W, H, P1, P2, cl, cr, o); }     \
      else { if (UNI) hipLaunchKernelGGL((k_sgm_path_q<RX, RY, true, false>), GRID, b, 0, s, W, H, P1, P2, cl, cr, o);        \
             else hipLaunchKernelGGL((k_sgm_path_q<RX, RY, false, false>), GRID, b, 0, s, W, H, P1, P2, cl, cr, o); }         \
    } while (0)
    switch (direction) {
      case 0: SGM_Q(1, 0, uh, qh); break;
      case 1: SGM_Q(-1, 0, uh, qh); break;
      case 2: SGM_Q(0, 1, uv, qv); break;
      case 3: SGM_Q(0, -1, uv, qv); break;
      case 4: SGM_Q(1, 1, false, qd); break;
      case 5: SGM_Q(-1, -1, false, qd); break;
      case 6: SGM_Q(-1, 1, false, qd); break;
      default: SGM_Q(1, -1, false, qd); break;
    }
#undef SGM_Q
    return;
  }
  if (D == 128) { SGM_PATH(true) } else { SGM_PATH(false) }
#undef SGM_PATH
}

// every path of the published configuration (D == 128) in one grid; returns false when the combination is not covered (other D,
// other path counts): the caller then launches the paths one by one
bool launch_sgm_paths_all(int W, int H, int frames, int D, int P1, int P2, int paths, size_t path_stride, const uint32_t *cl, const uint32_t *cr,
                          uint8_t *L, hipStream_t s) {
  if (D != 128 || (paths != 8 && paths != 4)) return false;
  const int nh = (H + 3) / 4, nv = (W + 3) / 4, nd = (W + H - 1 + 3) / 4;
  const dim3 g(2 * nh + 2 * nv + (paths == 8 ? 4 * nd : 0), frames), b(64);
  const bool uh = (H & 3) == 0, uv = (W & 3) == 0;
  if (uh && uv) hipLaunchKernelGGL((k_sgm_paths_all<true, true>), g, b, 0, s, W, H, P1, P2, paths, path_stride, cl, cr, L);
  else if (uh) hipLaunchKernelGGL((k_sgm_paths_all<true, false>), g, b, 0, s, W, H, P1, P2, paths, path_stride, cl, cr, L);
  else if (uv) hipLaunchKernelGGL((k_sgm_paths_all<false, true>), g, b, 0, s, W, H, P1, P2, paths, path_stride, cl, cr, L);
  else hipLaunchKernelGGL((k_sgm_paths_all<false, false>), g, b, 0, s, W, H, P1, P2, paths, path_stride, cl, cr, L);
  return true;
}

void launch_sgm_finish(int W, int H, int frames, int D, int paths, size_t path_stride, int median, int lr_check, const uint8_t *Lv,
                       uint8_t *dl, uint8_t *dr, uint8_t *dlm, uint8_t *drm, float *disparity, hipStream_t s) {
  if (D % 16 == 0) hipLaunchKernelGGL(k_sgm_wta16, dim3(H, frames), dim3(256), (size_t)W * sizeof(uint32_t), s, W, H, D, paths, path_stride, Lv, dl, dr);
  else hipLaunchKernelGGL(k_sgm_wta, dim3(H, frames), dim3(256), (size_t)W * sizeof(uint32_t), s, W, H, D, paths, path_stride, Lv, dl, dr);
  const dim3 g((W + 63) / 64, (H + 3) / 4, frames), b(64, 4);
  if (median) {
    hipLaunchKernelGGL(k_sgm_median3, g, b, 0, s, W, H, dl, dlm);
    hipLaunchKernelGGL(k_sgm_median3, g, b, 0, s, W, H, dr, drm);
  }
  hipLaunchKernelGGL(k_sgm_lr, g, b, 0, s, W, H, lr_check, median ? dlm : dl, median ? drm : dr, disparity);
}
