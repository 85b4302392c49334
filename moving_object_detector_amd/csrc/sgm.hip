// sgm.hip — on-GPU disparity estimation, first stages (BASELINE config 5).
//
// Replaces (when finished) sgm_gpu::SgmGpu::computeDisparity, the un-vendored CUDA estimator the reference calls at
// scene_flow_constructor/src/scene_flow_constructor.cpp:35,267; its output contract is the stereo_msgs/DisparityImage consumed at
// disparity_image_proc/src/disparity_image_processor.cpp:25-27,41-42.  The algorithm (centre-symmetric 9 x 7 census, Hamming
// cost over D <= 128 disparities, 8-path semi-global aggregation with P1 / P2, winner-take-all, median, left-right check) and
// every choice the publication leaves open are stated in oracle/sgm_ref.cpp, which these kernels match bit for bit.
// This round: the census transform and the two HORIZONTAL aggregation paths (cost computed on the fly, never stored unless asked).
#include "mod_launch.h"

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

// One horizontal aggregation path over one image row: ONE wave, lane l owns disparities l and l + 64; the row's census words sit
// in LDS (the right word of disparity d at column x is word x - d: consecutive lanes read consecutive addresses), the previous
// column's path costs in registers: neighbours d +- 1 come through DPP wave shifts, the minimum over d through a DPP reduction.
//   L(x, d) = C(x, d) + min(L(xp, d), L(xp, d-1) + P1, L(xp, d+1) + P1, min_k L(xp, k) + P2) - min_k L(xp, k)
// One step of a path for the lane's two disparities: previous path costs lp0 / lp1 (kNone where the disparity does not exist),
// matching costs c0 / c1 -> new path costs.  All 64 lanes must be active.
constexpr int kSgmNone = 1 << 20;                    // "no such disparity": larger than any cost + penalty
__device__ __forceinline__ void sgm_step(int lp0, int lp1, int c0, int c1, int P1, int P2, int &l0, int &l1) {
  const int m = (int)wave_min_u32((uint32_t)min(lp0, lp1));
  // d - 1: lane - 1 (disparity 64's left neighbour is lane 63's first); d + 1: lane + 1 (disparity 63's right neighbour is lane 0's second)
  const int lo63 = __builtin_amdgcn_readlane(lp0, 63), hi0 = __builtin_amdgcn_readlane(lp1, 0);
  const int a0 = __builtin_amdgcn_update_dpp(kSgmNone, lp0, 0x138, 0xF, 0xF, false);     // wave_shr:1, lane 0 keeps kSgmNone
  const int a1 = __builtin_amdgcn_update_dpp(lo63, lp1, 0x138, 0xF, 0xF, false);         // lane 0 <- lane 63's first
  const int b0 = __builtin_amdgcn_update_dpp(hi0, lp0, 0x130, 0xF, 0xF, false);          // wave_shl:1, lane 63 <- lane 0's second
  const int b1 = __builtin_amdgcn_update_dpp(kSgmNone, lp1, 0x130, 0xF, 0xF, false);
  l0 = c0 + min(min(lp0, m + P2), min(a0, b0) + P1) - m;
  l1 = c1 + min(min(lp1, m + P2), min(a1, b1) + P1) - m;
}

// where a path's costs go: its own volume (stage tests), the matching cost, and / or the running sum over the paths
struct SgmOut {
  uint8_t *L, *C;       // [H][W][D] or null
  uint16_t *S;          // [H][W][D] or null
  int first;            // S = l instead of S += l (first path of a frame)
};
__device__ __forceinline__ void sgm_emit(const SgmOut &o, size_t at, int d, int l, int c) {
  if (o.L) o.L[at + d] = (uint8_t)l;
  if (o.C) o.C[at + d] = (uint8_t)c;
  if (o.S) o.S[at + d] = (uint16_t)(o.first ? l : (int)o.S[at + d] + l);
}

template <bool RTL>
__global__ __launch_bounds__(64) void k_sgm_path_h(int W, int H, int D, int P1, int P2, const uint32_t *__restrict__ cl,
                                                   const uint32_t *__restrict__ cr, SgmOut out) {
  extern __shared__ uint32_t srow[];                 // [W] left census row, [W] right census row
  const int lane = threadIdx.x, y = blockIdx.x, f = blockIdx.y;
  const size_t row = ((size_t)f * H + y) * W;
  for (int i = lane; i < W; i += 64) { srow[i] = cl[row + i]; srow[W + i] = cr[row + i]; }
  __syncthreads();
  const int d0 = lane, d1 = lane + 64;
  const bool has0 = d0 < D, has1 = d1 < D;
  constexpr int kNone = kSgmNone;
  int lp0 = kNone, lp1 = kNone;
  const size_t vol = (size_t)f * H * W * D;
  if (out.L) out.L += vol;
  if (out.C) out.C += vol;
  if (out.S) out.S += vol;
  for (int step = 0; step < W; step++) {
    const int x = RTL ? W - 1 - step : step;
    const uint32_t wl = srow[x];
    const int c0 = x - d0 >= 0 ? __popc(wl ^ srow[W + max(x - d0, 0)]) : 31;
    const int c1 = x - d1 >= 0 ? __popc(wl ^ srow[W + max(x - d1, 0)]) : 31;
    int l0 = c0, l1 = c1;
    if (step > 0) sgm_step(lp0, lp1, c0, c1, P1, P2, l0, l1);
    const size_t o = ((size_t)y * W + x) * D;
    if (has0) sgm_emit(out, o, d0, l0, c0);
    if (has1) sgm_emit(out, o, d1, l1, c1);
    lp0 = has0 ? l0 : kNone;
    lp1 = has1 ? l1 : kNone;
  }
}

// Vertical and diagonal paths: one wave per path LINE (direction (RX, RY), RY != 0).  Lines enter through the first row in path
// order (W of them) and, for the diagonals, through the first column in path order (H - 1 more).  Census words come straight from
// HBM / L2 (the right word of disparity d at (x, y) is word (x - d, y): lanes read a descending run of addresses).
template <int RX, int RY>
__global__ __launch_bounds__(64) void k_sgm_path_line(int W, int H, int D, int P1, int P2, const uint32_t *__restrict__ cl,
                                                      const uint32_t *__restrict__ cr, SgmOut out) {
  const int lane = threadIdx.x, line = blockIdx.x, f = blockIdx.y;
  int x, y;
  if (line < W) { x = line; y = RY > 0 ? 0 : H - 1; }
  else { x = RX > 0 ? 0 : W - 1; y = RY > 0 ? line - W + 1 : H - 2 - (line - W); }   // the entry row's own pixel is a row line
  const size_t plane = (size_t)f * H * W, vol = plane * D;
  cl += plane; cr += plane;
  if (out.L) out.L += vol;
  if (out.C) out.C += vol;
  if (out.S) out.S += vol;
  const int d0 = lane, d1 = lane + 64;
  const bool has0 = d0 < D, has1 = d1 < D;
  int lp0 = kSgmNone, lp1 = kSgmNone;
  bool first = true;
  while (x >= 0 && x < W && y >= 0 && y < H) {       // wave-uniform
    const size_t p = (size_t)y * W + x;
    const uint32_t wl = cl[p];
    const int c0 = x - d0 >= 0 ? __popc(wl ^ cr[p - min(d0, x)]) : 31;
    const int c1 = x - d1 >= 0 ? __popc(wl ^ cr[p - min(d1, x)]) : 31;
    int l0 = c0, l1 = c1;
    if (!first) sgm_step(lp0, lp1, c0, c1, P1, P2, l0, l1);
    if (has0) sgm_emit(out, p * D, d0, l0, c0);
    if (has1) sgm_emit(out, p * D, d1, l1, c1);
    lp0 = has0 ? l0 : kSgmNone;
    lp1 = has1 ? l1 : kSgmNone;
    first = false;
    x += RX; y += RY;
  }
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
                     uint8_t *L, uint8_t *cost, uint16_t *S, bool first, hipStream_t s) {
  SgmOut o{L, cost, S, first ? 1 : 0};
  const size_t lds = (size_t)2 * W * sizeof(uint32_t);
  const dim3 b(64);
  switch (direction) {   // numbering of oracle/sgm_ref.cpp: 0 (+1,0) 1 (-1,0) 2 (0,+1) 3 (0,-1) 4 (+1,+1) 5 (-1,-1) 6 (-1,+1) 7 (+1,-1)
    case 0: hipLaunchKernelGGL(k_sgm_path_h<false>, dim3(H, frames), b, lds, s, W, H, D, P1, P2, cl, cr, o); break;
    case 1: hipLaunchKernelGGL(k_sgm_path_h<true>, dim3(H, frames), b, lds, s, W, H, D, P1, P2, cl, cr, o); break;
    case 2: hipLaunchKernelGGL((k_sgm_path_line<0, 1>), dim3(W, frames), b, 0, s, W, H, D, P1, P2, cl, cr, o); break;
    case 3: hipLaunchKernelGGL((k_sgm_path_line<0, -1>), dim3(W, frames), b, 0, s, W, H, D, P1, P2, cl, cr, o); break;
    case 4: hipLaunchKernelGGL((k_sgm_path_line<1, 1>), dim3(W + H - 1, frames), b, 0, s, W, H, D, P1, P2, cl, cr, o); break;
    case 5: hipLaunchKernelGGL((k_sgm_path_line<-1, -1>), dim3(W + H - 1, frames), b, 0, s, W, H, D, P1, P2, cl, cr, o); break;
    case 6: hipLaunchKernelGGL((k_sgm_path_line<-1, 1>), dim3(W + H - 1, frames), b, 0, s, W, H, D, P1, P2, cl, cr, o); break;
    default: hipLaunchKernelGGL((k_sgm_path_line<1, -1>), dim3(W + H - 1, frames), b, 0, s, W, H, D, P1, P2, cl, cr, o); break;
  }
}

void launch_sgm_finish(int W, int H, int frames, int D, int paths, size_t path_stride, int median, int lr_check, const uint8_t *Lv,
                       uint8_t *dl, uint8_t *dr, uint8_t *dlm, uint8_t *drm, float *disparity, hipStream_t s) {
  hipLaunchKernelGGL(k_sgm_wta, dim3(H, frames), dim3(256), (size_t)W * sizeof(uint32_t), s, W, H, D, paths, path_stride, Lv, dl, dr);
  const dim3 g((W + 63) / 64, (H + 3) / 4, frames), b(64, 4);
  if (median) {
    hipLaunchKernelGGL(k_sgm_median3, g, b, 0, s, W, H, dl, dlm);
    hipLaunchKernelGGL(k_sgm_median3, g, b, 0, s, W, H, dr, drm);
  }
  hipLaunchKernelGGL(k_sgm_lr, g, b, 0, s, W, H, lr_check, median ? dlm : dl, median ? drm : dr, disparity);
}
