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
template <bool RTL>
__global__ __launch_bounds__(64) void k_sgm_path_h(int W, int H, int D, int P1, int P2, const uint32_t *__restrict__ cl,
                                                   const uint32_t *__restrict__ cr, uint8_t *__restrict__ L, uint8_t *__restrict__ Cout) {
  extern __shared__ uint32_t srow[];                 // [W] left census row, [W] right census row
  const int lane = threadIdx.x, y = blockIdx.x, f = blockIdx.y;
  const size_t row = ((size_t)f * H + y) * W;
  for (int i = lane; i < W; i += 64) { srow[i] = cl[row + i]; srow[W + i] = cr[row + i]; }
  __syncthreads();
  const int d0 = lane, d1 = lane + 64;
  const bool has0 = d0 < D, has1 = d1 < D;
  constexpr int kNone = 1 << 20;                     // "no such disparity": larger than any cost + penalty
  int lp0 = kNone, lp1 = kNone;
  for (int step = 0; step < W; step++) {
    const int x = RTL ? W - 1 - step : step;
    const uint32_t wl = srow[x];
    const int c0 = x - d0 >= 0 ? __popc(wl ^ srow[W + max(x - d0, 0)]) : 31;
    const int c1 = x - d1 >= 0 ? __popc(wl ^ srow[W + max(x - d1, 0)]) : 31;
    int l0 = c0, l1 = c1;
    if (step > 0) {
      const int m = (int)wave_min_u32((uint32_t)min(lp0, lp1));
      // d - 1: lane - 1 (disparity 64's left neighbour is lane 63's first); d + 1: lane + 1 (disparity 63's right neighbour is lane 0's second)
      const int lo63 = __builtin_amdgcn_readlane(lp0, 63), hi0 = __builtin_amdgcn_readlane(lp1, 0);
      const int a0 = __builtin_amdgcn_update_dpp(kNone, lp0, 0x138, 0xF, 0xF, false);      // wave_shr:1, lane 0 keeps kNone
      const int a1 = __builtin_amdgcn_update_dpp(lo63, lp1, 0x138, 0xF, 0xF, false);       // lane 0 <- lane 63's first
      const int b0 = __builtin_amdgcn_update_dpp(hi0, lp0, 0x130, 0xF, 0xF, false);        // wave_shl:1, lane 63 <- lane 0's second
      const int b1 = __builtin_amdgcn_update_dpp(kNone, lp1, 0x130, 0xF, 0xF, false);
      l0 = c0 + min(min(lp0, m + P2), min(a0, b0) + P1) - m;
      l1 = c1 + min(min(lp1, m + P2), min(a1, b1) + P1) - m;
    }
    const size_t o = (row + x) * D;
    if (has0) { L[o + d0] = (uint8_t)l0; if (Cout) Cout[o + d0] = (uint8_t)c0; }
    if (has1) { L[o + d1] = (uint8_t)l1; if (Cout) Cout[o + d1] = (uint8_t)c1; }
    lp0 = has0 ? l0 : kNone;
    lp1 = has1 ? l1 : kNone;
  }
}

}  // namespace

void launch_sgm_census(int W, int H, int frames, const uint8_t *img, uint32_t *out, hipStream_t s) {
  hipLaunchKernelGGL(k_sgm_census, dim3((W + 63) / 64, (H + 3) / 4, frames), dim3(64, 4, 1), 0, s, W, H, img, out);
}

void launch_sgm_path_h(int W, int H, int frames, int D, int P1, int P2, bool right_to_left, const uint32_t *cl, const uint32_t *cr,
                       uint8_t *L, uint8_t *cost, hipStream_t s) {
  const size_t lds = (size_t)2 * W * sizeof(uint32_t);
  if (right_to_left) hipLaunchKernelGGL(k_sgm_path_h<true>, dim3(H, frames), dim3(64), lds, s, W, H, D, P1, P2, cl, cr, L, cost);
  else hipLaunchKernelGGL(k_sgm_path_h<false>, dim3(H, frames), dim3(64), lds, s, W, H, D, P1, P2, cl, cr, L, cost);
}
