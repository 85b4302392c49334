// sgm_ref.cpp — CPU restatement of the disparity estimator the reference calls, sgm_gpu::SgmGpu::computeDisparity
// (scene_flow_constructor/src/scene_flow_constructor.cpp:35,267), for SURVEY.md §8(f) row 3 (on-GPU SGM, BASELINE config 5).
//
// TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product never does.
//
// PARITY UNPINNED.  The estimator is NOT in /root/reference: it is the un-vendored package `sgm_gpu` of
// ActiveIntelligentSystemsLab/sgm_gpu_ros, cloned at HEAD by docker/dockerfile:113-114 (no version pinned), a ROS wrapper of
// D. Hernandez-Juarez et al., "Embedded real-time stereo estimation via Semi-Global Matching on the GPU" (ICCS 2016).  What is
// restated here is that published algorithm with the parameters of its public implementation, as far as they are documented:
//   1. centre-symmetric census transform over a 9 x 7 window -> 31 bits per pixel;
//   2. matching cost = Hamming distance of the census words, D = 128 disparities (uint8);
//   3. semi-global path aggregation (Hirschmueller 2008) along 8 directions with P1 = 6, P2 = 96 (uint8 per path):
//        L_r(p, d) = C(p, d) + min(L_r(p-r, d), L_r(p-r, d-1) + P1, L_r(p-r, d+1) + P1, min_k L_r(p-r, k) + P2) - min_k L_r(p-r, k)
//   4. winner-take-all over the summed path costs (first minimum), 3 x 3 median, left-right consistency check (|dl - dr| <= 1);
//   5. stereo_msgs/DisparityImage contract consumed downstream (disparity_image_proc/src/disparity_image_processor.cpp:25-27,41-42):
//      32FC1 image, invalid pixels = min_disparity - 1 = -1, min_disparity = 0, max_disparity = D - 1, f and T from the camera.
// Everything the paper leaves open is fixed here and documented at the function that fixes it (image border, disparities that
// leave the right image, tie-breaking); a later pin against the real package may move these choices.
//
// Build: make -C oracle libsgm_ref.so   (plain C++14, integer arithmetic only)
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

struct SgmParams {
  int32_t disparities;   // D: 128 (a multiple of 64, <= 128)
  int32_t p1, p2;        // 6, 96
  int32_t paths;         // 8 (or 4: the horizontal and vertical ones)
  int32_t lr_check;      // 1: left-right consistency check with tolerance 1
  int32_t median;        // 1: 3 x 3 median of the winner-take-all map
};

// Centre-symmetric census, 9 wide x 7 high.  Bit order: the 31 pixel pairs (p, -p) in raster order of p over the upper half of
// the window — rows dy = -3..-1 with dx = -4..4, then row dy = 0 with dx = -4..-1 — first pair in the most significant of the 31
// bits; bit = 1 iff I(x+dx, y+dy) >= I(x-dx, y-dy).  Border choice: pixels whose window leaves the image get the word 0.
void sgm_census(const uint8_t *img, int W, int H, uint32_t *out) {
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      uint32_t c = 0;
      if (x >= 4 && x < W - 4 && y >= 3 && y < H - 3) {
        for (int dy = -3; dy <= 0; dy++)
          for (int dx = -4; dx <= (dy < 0 ? 4 : -1); dx++)
            c = (c << 1) | (img[(size_t)(y + dy) * W + x + dx] >= img[(size_t)(y - dy) * W + x - dx] ? 1u : 0u);
      }
      out[(size_t)y * W + x] = c;
    }
}

// Matching cost, layout [H][W][D].  Choice for disparities that leave the right image (x - d < 0): the cost of the nearest valid
// disparity is NOT reused — such entries get the constant 31 (the largest Hamming distance of 31-bit words), so that they can never
// win against a real match but keep every sum inside uint8 / uint16.
void sgm_cost(const uint32_t *cl, const uint32_t *cr, int W, int H, int D, uint8_t *C) {
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++)
      for (int d = 0; d < D; d++)
        C[((size_t)y * W + x) * D + d] = x - d >= 0 ? (uint8_t)__builtin_popcount(cl[(size_t)y * W + x] ^ cr[(size_t)y * W + x - d]) : (uint8_t)31;
}

// One aggregation path.  dir: 0 left->right (r = (+1, 0)), 1 right->left, 2 top->bottom, 3 bottom->top, 4 (+1,+1), 5 (-1,-1),
// 6 (-1,+1) [towards left-down], 7 (+1,-1).  The first pixel of every path line takes L = C.  d-1 / d+1 outside [0, D) are ignored.
void sgm_aggregate_path(const uint8_t *C, int W, int H, int D, int P1, int P2, int dir, uint8_t *L) {
  static const int RX[8] = {1, -1, 0, 0, 1, -1, -1, 1}, RY[8] = {0, 0, 1, -1, 1, -1, 1, -1};
  const int rx = RX[dir], ry = RY[dir];
  std::vector<int> order_y(H), order_x(W);
  for (int i = 0; i < H; i++) order_y[i] = ry >= 0 ? i : H - 1 - i;
  for (int i = 0; i < W; i++) order_x[i] = rx >= 0 ? i : W - 1 - i;
  for (int yi = 0; yi < H; yi++)
    for (int xi = 0; xi < W; xi++) {
      const int y = order_y[yi], x = order_x[xi], px = x - rx, py = y - ry;
      const uint8_t *c = C + ((size_t)y * W + x) * D;
      uint8_t *l = L + ((size_t)y * W + x) * D;
      if (px < 0 || px >= W || py < 0 || py >= H) { memcpy(l, c, D); continue; }
      const uint8_t *lp = L + ((size_t)py * W + px) * D;
      int mn = 255;
      for (int d = 0; d < D; d++) mn = std::min(mn, (int)lp[d]);
      for (int d = 0; d < D; d++) {
        int best = std::min((int)lp[d], mn + P2);
        if (d > 0) best = std::min(best, lp[d - 1] + P1);
        if (d < D - 1) best = std::min(best, lp[d + 1] + P1);
        l[d] = (uint8_t)(c[d] + best - mn);          // <= 31 + P2 < 256
      }
    }
}

static inline uint8_t median9(uint8_t *v) { std::nth_element(v, v + 4, v + 9); return v[4]; }

// Whole estimator: left / right 8-bit images -> disparity (float32, -1 where invalid).  Scratch is allocated inside.
// S = sum of the path costs (uint16); winner-take-all = first minimum over d; right disparity from the same S:
// dr(x) = argmin_d S(x + d, d) (first minimum, x + d < W); 3 x 3 median on both maps (border pixels keep their value);
// left-right check: dl(x) is kept iff |dl(x) - dr(x - dl(x))| <= 1.
int sgm_compute(const uint8_t *left, const uint8_t *right, int W, int H, const SgmParams *p, float *disparity, uint16_t *S_out) {
  const int D = p->disparities;
  if (D < 1 || D > 128 || W < 1 || H < 1) return -1;
  const size_t N = (size_t)W * H;
  std::vector<uint32_t> cl(N), cr(N);
  sgm_census(left, W, H, cl.data());
  sgm_census(right, W, H, cr.data());
  std::vector<uint8_t> C(N * D), L(N * D);
  sgm_cost(cl.data(), cr.data(), W, H, D, C.data());
  std::vector<uint16_t> S(N * D, 0);
  static const int order4[4] = {0, 1, 2, 3};
  for (int i = 0; i < p->paths; i++) {
    const int dir = p->paths == 4 ? order4[i] : i;
    sgm_aggregate_path(C.data(), W, H, D, p->p1, p->p2, dir, L.data());
    for (size_t k = 0; k < N * D; k++) S[k] = (uint16_t)(S[k] + L[k]);
  }
  if (S_out) memcpy(S_out, S.data(), N * D * sizeof(uint16_t));
  std::vector<uint8_t> dl(N), dr(N), t(N);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const uint16_t *s = S.data() + ((size_t)y * W + x) * D;
      int best = 0;
      for (int d = 1; d < D; d++) if (s[d] < s[best]) best = d;
      dl[(size_t)y * W + x] = (uint8_t)best;
      int bd = 0;
      uint16_t bv = S[((size_t)y * W + x) * D];
      for (int d = 1; d < D && x + d < W; d++) { const uint16_t v = S[((size_t)y * W + x + d) * D + d]; if (v < bv) { bv = v; bd = d; } }
      dr[(size_t)y * W + x] = (uint8_t)bd;
    }
  if (p->median)
    for (std::vector<uint8_t> *m : {&dl, &dr}) {
      t = *m;
      for (int y = 1; y < H - 1; y++)
        for (int x = 1; x < W - 1; x++) {
          uint8_t v[9];
          int k = 0;
          for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) v[k++] = t[(size_t)(y + dy) * W + x + dx];
          (*m)[(size_t)y * W + x] = median9(v);
        }
    }
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const int d = dl[(size_t)y * W + x];
      bool ok = true;
      if (p->lr_check) ok = x - d >= 0 && std::abs((int)dr[(size_t)y * W + x - d] - d) <= 1;
      disparity[(size_t)y * W + x] = ok ? (float)d : -1.0f;
    }
  return 0;
}

}  // extern "C"
