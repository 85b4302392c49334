// cluster.hip — moving-point clustering on gfx950 (MI355X).
//
// Replaces ClustererNodelet::clustering + publishMovingObjects
// (scene_flow_clusterer/src/clusterer_nodelet.cpp:85-95,324-343):
//   calculateInitialClusterMap + comparePoints + LookupTable   :56-83,186-219, include/lookup_table.h:10-33
//   integrateConnectedClusters                                  :253-267
//   removeSmallClusters                                         :354-393
//   clusterMap2IndicesCluster + cluster2MovingObject            :97-117,147-184
//
// The reference is a serial raster scan with a union-find; what it computes is order-independent (SURVEY.md
// Appendix A): the connected components of the graph whose edges join a dynamic pixel p to each dynamic pixel q in
// its up-left (n+1)x(n+1) window with !(|z_p - z_q| > depth_diff); components without an edge stay unlabelled;
// survivors of the size filter are numbered by ascending first_edge_key = the smallest raster index of a member that
// has an up-left edge (that is where the serial scan creates the component's first — hence smallest — label).
//
// GPU formulation (one launcher per stage, include/mod_sf.h MOD_STAGE_*)
//   k_ccl_tile     one workgroup per 64 x 16 tile + n-pixel halo (up / left): masked depth + parents live in LDS; rows are
//                  pre-linked into runs with wave ballots, vertically with one union per run pair, the remaining window edges
//                  are united with LDS atomicMin hooks; interior pixels publish parent[p] = tile root with plain stores, halo
//                  pixels that were reached leave link requests; one partial statistics record per tile root.
//   k_ccl_link     requests -> unions between tile roots (device-scope atomicMin hooks), one wave per tile.
//   k_ccl_merge    every tile root finds its final root, folds its record into it; final roots are listed; in a small batch the
//                  frame's last workgroup goes on to the size filter;
//   k_select       (large batches: a kernel of its own) size filter + ordering by first_edge_key (the reference's numbering), work list.
//   k_final        labels plane + per-cluster member lists (||v|| bits, pixel).
//   k_median       exact selection of the member at size/2 by ||v|| (norms held in registers, LDS histogram rounds);
//   k_median_ties  replay of libstdc++'s introsort for clusters whose median ties between different vectors; the launch's last
//                  workgroup assigns the object ids over the accepted clusters and zeroes the counters for the next call.
#include "mod_launch.h"
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <string>
#include "introsort_emul.h"
#include "../../include/mod_sf.h"

#pragma clang fp contract(off)

namespace {

constexpr int kKeyNone = 0x7fffffff;

// Workgroup barrier for phases that only exchange data through LDS: waits for this wave's LDS operations, not for its
// outstanding global loads/stores (__syncthreads() drains vmcnt(0) and would serialise every HBM round trip).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------
// Union-find helpers.  parent[i] <= i always; roots satisfy parent[r] == r; a larger root is hooked under a smaller
// one with atomicMin, whose return value tells whether the node was still a root.  Reads are relaxed atomic loads so
// the compiler re-reads memory; values that are stale in a CU's L1 are harmless: every value ever stored in parent[a]
// is a member of a's set and smaller than a, so a stale chain still ends inside the same set.
__device__ __forceinline__ int ld_relaxed(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// device-scope store / load: to / from memory past the (per-XCD, mutually incoherent) L2s — how a workgroup reads, in the SAME kernel,
// what a workgroup on another XCD has written (k_ccl_merge's last workgroup per frame)
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(const int *parent, int a) {
  int p = ld_relaxed(parent + a);
  while (p != a) { a = p; p = ld_relaxed(parent + a); }
  return a;
}

// unite the sets of a and b; returns the (current) root of the merged set
__device__ __forceinline__ int uf_unite(int *parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return a;
    if (a < b) { const int t = a; a = b; b = t; }   // hook the larger root a under the smaller b
    const int old = atomicMin(&parent[a], b);
    if (old == a) return b;                          // a was still a root
    a = old;                                         // a had been hooked meanwhile: unite its parent with b instead
  }
}

// Adds the members of one wave (grouped by the record they belong to) into the statistics records with one set of
// atomics per (wave, record).  `rec_idx` < 0 marks a lane without contribution.  Works on LDS slots and on the global planes.
__device__ __forceinline__ void wave_accumulate(int *sizes, int *keys, int stride, int rec_idx, uint32_t key, int lane) {
  uint64_t todo = __ballot(rec_idx >= 0);
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int lid = __builtin_amdgcn_readlane(rec_idx, leader);
    const bool mine = rec_idx == lid;
    const uint64_t grp = __ballot(mine);
    const uint32_t k = wave_min_u32(mine ? key : (uint32_t)kKeyNone);
    if (lane == leader) {
      atomicAdd(&sizes[(size_t)lid * stride], __popcll((unsigned long long)grp));
      if (k != (uint32_t)kKeyNone) atomicMin(&keys[(size_t)lid * stride], (int)k);
    }
    todo &= ~grp;
  }
}

// LDS union-find over node ids: interior cells use their grid index, halo cells carry bit 15 so that they compare
// larger than every interior cell — the root (minimum id) of any set that contains an interior pixel is interior.
constexpr int kHaloBit = 0x8000;
__device__ __forceinline__ int lds_find(int *L, int a) {
  // path halving with plain stores: only non-root cells are rewritten, and only with one of their ancestors, so a
  // racing atomicMin hook (which re-examines the value it displaced) never loses a link
  int p = ld_relaxed(L + (a & 0x7fff));
  while (p != a) {
    const int g = ld_relaxed(L + (p & 0x7fff));
    if (g != p) L[a & 0x7fff] = g;
    a = p; p = g;
  }
  return a;
}
__device__ __forceinline__ int lds_unite(int *L, int a, int b) {
  while (true) {
    a = lds_find(L, a);
    b = lds_find(L, b);
    if (a == b) return a;
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&L[a & 0x7fff], b);
    if (old == a) return b;
    a = old;
  }
}

// Unions of one window position for the whole wave.  Lanes of one run looking at one run above all hold the same
// (cur, other) pair, so only the first lane of each contiguous group of equal pairs performs the union; all such
// representatives run concurrently (LDS atomicMin hooks), then every lane that needed the union refreshes its root.
// `last` remembers the label a lane was united with most recently: the neighbours in one window row nearly always
// carry the same (possibly no longer root) label, and re-uniting them would cost a find each.
__device__ __forceinline__ int lds_find_compress(int *L, int a) {
  const int r = lds_find(L, a);
  if (r != a && ld_relaxed(L + (a & 0x7fff)) != r) L[a & 0x7fff] = r;   // benign race: r is an ancestor of a
  return r;
}
__device__ __forceinline__ void wave_unite_lds(int *L, bool need, int &cur, int &last, int other, int lane) {
  if (__ballot(need) == 0) return;
  const int pc = wave_prev_i32(cur), po = wave_prev_i32(other), pn = wave_prev_i32((int)need);
  const bool rep = need && !(lane > 0 && pn && pc == cur && po == other);
  if (rep) lds_unite(L, cur, other);
  if (need) { last = other; cur = lds_find_compress(L, cur); }
}

// Tile-local connected components INCLUDING the edges that leave the tile.
// Workgroup = 4 waves = one 64 x TH tile plus an n-pixel halo above and to the left (masked depth + parents in LDS).
//   A  cooperative load of the (TH+n) x (64+n) grid; rows pre-linked into runs with wave ballots (no atomics)
//   B  the rest of each pixel's up-left window: (n+1) batched LDS reads per window row, unions deduplicated per wave
//   C  publish: interior pixels hook onto their tile root, linked halo pixels are united with it in HBM (atomicMin
//      only), tile roots get an empty statistics record
//   D  partial statistics (size, first_edge_key, bbox) of the tile's components, one set of atomics per (wave, root)
// CCL_TPB consecutive tiles of a tile row share one workgroup, which walks the active ones one after the other: three quarters of
// the tiles of a street scene are empty, and 460 k workgroups that only read a header word and exit cost 0.3 ms per 512 pairs
// of workgroup launches (the kernel's time on a batch without any dynamic pixel).
#ifndef CCL_TPB
#define CCL_TPB 1
#endif
template <int TH, int NMAX, int NW, bool EXACT>
__device__ __forceinline__ void ccl_tile_body(const DevCam &c, const ClArgs &a, const int wi, const int ty, const int f, const int tiles_x, const int tiles_y);

template <int TH, int NMAX, int NW, bool EXACT>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_ccl_tile(DevCam c, ClArgs a, int tiles_x, int tiles_y) {
  const int ty = blockIdx.y, f = blockIdx.z;
#pragma unroll 1
  for (int t = 0; t < CCL_TPB; t++) {
    const int wi = blockIdx.x * CCL_TPB + t;
    if (wi >= tiles_x) break;
    // The tile's header says whether it holds a dynamic pixel at all (written with the mask words: by the scene-flow kernel's
    // epilogue, or by k_tile_flags).  Three quarters of the tiles of a street scene do not: they cost one scalar load.
    const int *hdr = a.tilehdr + ((size_t)f * tiles_y * tiles_x + (size_t)ty * tiles_x + wi) * 2;
    if (__builtin_amdgcn_readfirstlane(hdr[0]) == 0) continue;
    ccl_tile_body<TH, NMAX, NW, EXACT>(c, a, wi, ty, f, tiles_x, tiles_y);
    if (CCL_TPB > 1) __syncthreads();                // the next tile re-uses the LDS arrays
  }
}

template <int TH, int NMAX, int NW, bool EXACT>
__device__ __forceinline__ void ccl_tile_body(const DevCam &c, const ClArgs &a, const int wi, const int ty, const int f, const int tiles_x, const int tiles_y) {
  constexpr int RPW = TH / NW;                      // rows per wave (NW waves per tile)
  constexpr int PW = 64 + NMAX, PH = TH + NMAX, G = PW * PH;
  __shared__ float zt[G];
  __shared__ int Lt[G];
  __shared__ uint64_t m0[PH], mL[PH];
  __shared__ int s_uni[PH];                         // the one label every dynamic cell of the grid row carries after phase A3, or -1
  constexpr int kSlots = 32;
  __shared__ int s_nreq, s_nslots, s_anyhalo;
  __shared__ RootRec srec[kSlots];
  __shared__ int sroot[kSlots];
  // threadIdx.y is the wave index: the same in all 64 lanes, but it arrives in a vector register — as a scalar, every row
  // index derived from it stays on the scalar unit
  const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane((int)threadIdx.y), tid = w * 64 + lane;
  const int x0 = wi * 64, y0 = ty * TH;
  // EXACT: neighbor_distance equals the instance's halo width (the default n = 4 does): the window loops get constant bounds
  const int MW = c.mask_words, n = EXACT ? NMAX : c.n;
  const size_t N = (size_t)c.W * c.H;
  const size_t fN = (size_t)f * N;
  int *hdr = a.tilehdr + ((size_t)f * tiles_y * tiles_x + (size_t)ty * tiles_x + wi) * 2;
  if (tid == 0) { s_nreq = 0; s_nslots = 0; }
  // ---- all HBM reads of the kernel, in ONE round trip: the mask words of the PH grid rows and the depth rows (unconditional,
  // clamped addresses; values of non-dynamic pixels are discarded: predicated loads would compile to one exec-masked branch +
  // wait each) ----
  {
    uint64_t q0 = 0ull, qL = 0ull;
    if (tid < PH) {
      const int gy = y0 - NMAX + tid;
      const bool inrow = gy >= 0 && gy < c.H && tid >= NMAX - n;
      const uint64_t *mr = a.mask + ((size_t)f * c.H + (inrow ? gy : 0)) * MW;
      q0 = inrow ? mr[wi] : 0ull;
      qL = (inrow && wi > 0) ? mr[wi - 1] : 0ull;
      m0[tid] = q0; mL[tid] = qL;
    }
    if (w == 0) {                                    // PH <= 64: all mask words sit in wave 0.  Any dynamic halo cell at all?
      const uint64_t hb = __ballot(tid < PH && (((tid < NMAX) ? q0 : 0ull) | (qL >> (64 - NMAX))) != 0ull);
      if (lane == 0) s_anyhalo = hb != 0ull;
    }
  }
  constexpr int AROWS = (PH + NW - 1) / NW;
  float zl[AROWS], zh[AROWS];
  const int xc = min(x0 + lane, c.W - 1), xhc = max(x0 - 1 - lane, 0);
#pragma unroll
  for (int i = 0; i < AROWS; i++) {
    const int gy = min(max(y0 - NMAX + w + NW * i, 0), c.H - 1);
    const size_t rowp = fN + (size_t)gy * c.W;
    zl[i] = a.z[rowp + xc];
    zh[i] = a.z[rowp + xhc];                          // left halo: column x0 - 1 - lane (lanes < n)
  }
  __syncthreads();
  // rows are dealt to the waves round-robin (wave w owns rows w, w + 4, ...): a blob's rows are contiguous, so contiguous
  // row blocks would leave most of a tile's work to one wave while the others wait at the barriers
  const float th = c.depth_th;
#ifdef MOD_PHASE_COUNTERS   // diagnostic build only (make PHASE_COUNTERS=1): per-phase cycle sums in ClArgs.dbg
  const bool prof = c.debug & 128;
  unsigned long long t0 = prof ? clock64() : 0, t1;
#define STAMP(i) if (prof) { t1 = clock64(); if (lane == 0) { atomicAdd(&a.dbg[i], t1 - t0); atomicMax(&a.dbg[16 + i], t1 - t0); } t0 = t1; }
#define COUNT(i, v) if (prof && lane == 0) atomicAdd(&a.dbg[i], (unsigned long long)(v));
#elif defined(MOD_PHASE_MARKERS)
#define STAMP(i) asm volatile("; PHASE_MARK " #i ::: "memory");
#define COUNT(i, v)
#else
#define STAMP(i)
#define COUNT(i, v)
#endif
  // ---- phase A: masked depth + identity parents; grid row gr = image row y0 - NMAX + gr, 4 rows per step -----------
#pragma unroll
  for (int i = 0; i < AROWS; i++) {
    const int gr = w + NW * i;
    if (gr < PH) {
      const uint64_t q0 = m0[gr], qL = mL[gr];        // zero for rows that are unused or outside the image
      const int cell = gr * PW + NMAX + lane;
      zt[cell] = ((q0 >> lane) & 1ull) ? zl[i] : 0.0f;
      Lt[cell] = (gr >= NMAX) ? cell : (cell | kHaloBit);
      if (lane < n) {
        const int hc = gr * PW + NMAX - 1 - lane;
        zt[hc] = ((qL >> (63 - lane)) & 1ull) ? zh[i] : 0.0f;
        Lt[hc] = hc | kHaloBit;
      }
    }
  }
  lds_barrier();
  STAMP(0)
  if (MOD_ABLATE(c, 1 << 14)) return;                // (ablation builds: kernel truncated after a phase, timing only — tools/ablate.sh)
  // ---- phase A1: horizontal runs of the wave's rows by ballot (no atomics) ---------------------------------------------
  // One row of the grid: runs of the 64 tile columns by ballot; the left-halo cells chained to lane 0 by horizontal links
  // (h0 - lane 0, h1 - h0, ...) join lane 0's run.  Returns "linked to the left neighbour" (an up-left edge).
  auto link_row = [&](int gr) -> bool {
    const bool halo_row = gr < NMAX;
    const int me = gr * PW + NMAX + lane;
    const uint64_t mw = m0[gr], ml = mL[gr];
    const bool dyn = (mw >> lane) & 1ull;
    const float z = zt[me];
    const float zl = wave_prev_f32(z);
    const bool cl = dyn && lane > 0 && ((mw >> (lane - 1)) & 1ull) && !(fabsf(z - zl) > th);   // linked to the left neighbour
    const uint64_t C = __ballot(cl);
    const uint64_t starts = mw & ~C;                // run starts: dynamic and not linked to the left
    // left-halo cells: lane j < n owns the cell in column x0-1-j; bit j of lk: it is linked to its right neighbour
    const int hc = gr * PW + NMAX - 1 - min(lane, NMAX - 1);
    int m = 0;                                      // cells h0 .. h(m-1) hang on lane 0 through an unbroken chain of links
    if ((ml >> 63) & mw & 1ull) {                   // wave-uniform: the chain starts with h0 - lane 0, both dynamic
      const bool hdyn = lane < n && ((ml >> (63 - lane)) & 1ull);
      const bool rdyn = lane == 0 ? (bool)(mw & 1ull) : (bool)((ml >> (64 - lane)) & 1ull);
      const bool hl = hdyn && rdyn && !(fabsf(zt[hc] - zt[hc + 1]) > th);
      const uint32_t lk = (uint32_t)__ballot(hl);
      m = __builtin_ctz(~lk);
    }
    // label of lane 0's run: its own cell — in a halo row the leftmost chained cell (parents must not be larger than children)
    const int id0 = halo_row ? ((gr * PW + NMAX - m) | kHaloBit) : (gr * PW + NMAX);
    if (dyn) {
      const int s = 63 - __clzll((long long)(starts & (~0ull >> (63 - lane))));
      Lt[me] = s == 0 ? id0 : ((me - lane + s) | (halo_row ? kHaloBit : 0));
    }
    if (lane < m) Lt[hc] = id0;
    return cl || (lane == 0 && m > 0);
  };
  bool upr[RPW];
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int rr = w + NW * j;
    upr[j] = link_row(rr + NMAX);
  }
  for (int hr = NMAX - 1 - w; hr >= NMAX - n; hr -= NW)   // the halo rows above the tile, dealt to the waves like the tile rows
    if (m0[hr] | mL[hr]) link_row(hr);                    // wave-uniform
  lds_barrier();
  if (MOD_ABLATE(c, 1 << 15)) return;
  // ---- phase A2: vertical pre-link (the pixel straight above), one union per distinct (run, run-above) pair -----------
  // The unions of all the wave's rows are first collected in a queue in LDS and then run ONE PER LANE: run in place they keep
  // about eight of a row's 64 lanes busy (the first lane of each run pair) — and a wave64 instruction costs its four cycles
  // however few lanes are active, which is what bounds this kernel.
  constexpr int kJobCap = 128;
  __shared__ int s_ja[NW][kJobCap], s_jb[NW][kJobCap];
  int njobs = 0;                                     // wave-uniform
  auto push_jobs = [&](bool rep, int ja, int jb) {
    const uint64_t m = __ballot(rep);
    if (m == 0) return;                              // wave-uniform
    const int cnt = __popcll((unsigned long long)m);
    if (njobs + cnt > kJobCap) { if (rep) lds_unite(Lt, ja, jb); return; }   // queue full (wave-uniform): unite in place
    if (rep) {
      const int slot = njobs + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      s_ja[w][slot] = ja; s_jb[w][slot] = jb;
    }
    njobs += cnt;
  };
  auto link_up = [&](int gr) -> bool {              // grid row gr with grid row gr - 1 (tile rows and halo rows alike)
    const int me = gr * PW + NMAX + lane;
    const uint64_t both = m0[gr] & m0[gr - 1], hboth = (mL[gr] & mL[gr - 1]) >> (64 - NMAX);
    bool link = false;
    if (both != 0 && !MOD_ABLATE(c, 2048)) {                                // wave-uniform
      link = ((both >> lane) & 1ull) && !(fabsf(zt[me] - zt[me - PW]) > th);
      // lanes of one run looking at one run above all hold the same (label, label above) pair: the first lane of each contiguous
      // group of equal pairs represents it.  (Labels are not refreshed afterwards: phase A3 flattens every cell.)
      const int cur = ld_relaxed(&Lt[me]), oth = ld_relaxed(&Lt[me - PW]);
      const int pc = wave_prev_i32(cur), po = wave_prev_i32(oth), pn = wave_prev_i32((int)link);
      push_jobs(link && !(lane > 0 && pn && pc == cur && po == oth), cur, oth);
    }
    if (hboth != 0) {                                                       // wave-uniform: the left-halo cells, column by column
      const int hc = gr * PW + NMAX - 1 - min(lane, NMAX - 1);
      const bool hlink = lane < n && ((hboth >> (NMAX - 1 - lane)) & 1ull) && !(fabsf(zt[hc] - zt[hc - PW]) > th);
      const int cur = ld_relaxed(&Lt[hc]), oth = ld_relaxed(&Lt[hc - PW]);
      const int pc = wave_prev_i32(cur), po = wave_prev_i32(oth), pn = wave_prev_i32((int)hlink);
      push_jobs(hlink && !(lane > 0 && pn && pc == cur && po == oth), cur, oth);
    }
    return link;
  };
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int rr = w + NW * j;
    const bool link = link_up(rr + NMAX);
    upr[j] = upr[j] || link;
  }
  for (int hr = NMAX - 1 - w; hr > NMAX - n; hr -= NW) link_up(hr);
  auto run_jobs = [&]() {
    for (int j0 = 0; j0 < njobs; j0 += 64) {         // wave-uniform; the wave's own queue: its LDS writes are in order, no barrier
      const int j = j0 + lane;
      if (j < njobs) lds_unite(Lt, s_ja[w][j], s_jb[w][j]);
    }
    njobs = 0;
  };
  run_jobs();
  lds_barrier();
  if (MOD_ABLATE(c, 1 << 16)) return;
  // ---- phase A3: flatten, so that phase B can compare labels directly -------------------------------------------------
  auto flatten_row = [&](int gr) {
    const int me = gr * PW + NMAX + lane;
    const bool dyn = (m0[gr] >> lane) & 1ull, hdyn = lane < n && ((mL[gr] >> (63 - lane)) & 1ull);
    int lab = -1, hlab = -1;
    if (dyn) { lab = lds_find(Lt, ld_relaxed(&Lt[me])); if (ld_relaxed(&Lt[me]) != lab) Lt[me] = lab; }
    if (hdyn) {
      const int hc = gr * PW + NMAX - 1 - lane;
      hlab = lds_find(Lt, ld_relaxed(&Lt[hc]));
      if (ld_relaxed(&Lt[hc]) != hlab) Lt[hc] = hlab;
    }
    // row summary for phase B: when this row and a window row both carry one and the same label, no union can come of them
    const uint64_t db = __ballot(dyn), hb = __ballot(hdyn);
    int first = -1;
    if (db) first = __builtin_amdgcn_readlane(lab, __builtin_ctzll(db));
    else if (hb) first = __builtin_amdgcn_readlane(hlab, __builtin_ctzll(hb));
    const bool uni = __ballot((dyn && lab != first) || (hdyn && hlab != first)) == 0;
    if (lane == 0) s_uni[gr] = uni ? first : -1;
  };
#pragma unroll
  for (int j = 0; j < RPW; j++) flatten_row(w + NW * j + NMAX);
  for (int hr = NMAX - 1 - w; hr >= NMAX - n; hr -= NW) flatten_row(hr);
  lds_barrier();
  STAMP(1)
  if (MOD_ABLATE(c, 1 << 17)) return;
  // ---- phase B: the rest of the up-left window --------------------------------------------------------------------------
  const uint32_t kmask = (2u << n) - 1u;              // n + 1 low bits
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int rr = w + NW * j;
    const uint64_t mw = m0[rr + NMAX];
    if (mw == 0 || MOD_ABLATE(c, 256)) continue;                           // wave-uniform
    const bool dyn = (mw >> lane) & 1ull;
    const int me = (rr + NMAX) * PW + NMAX + lane;
    const float zp = zt[me];
    int cur = dyn ? ld_relaxed(&Lt[me]) : -1, last = -1;
    bool up = upr[j];
    const int ul = s_uni[rr + NMAX];                 // wave-uniform
    bool rowup = __ballot(dyn & !up) == 0;           // every dynamic pixel of the row has an up-left edge (wave-uniform)
    COUNT(13, 1)
#pragma unroll
    for (int dv = 0; dv <= (EXACT ? NMAX : n); dv++) {   // unrolled in the EXACT instance
      const int qg = rr + NMAX - dv;                 // grid row of the window row
      const uint64_t q0 = m0[qg], qL = mL[qg];
      if ((q0 | qL) == 0) continue;                  // wave-uniform
      // both rows carry one label, the same (phase A3's summaries): nothing to unite; when every pixel of the row has its
      // up-left edge already, the window row is of no interest at all (wave-uniform, scalar)
      const bool same = ul >= 0 && s_uni[qg] == ul;
      if (same && rowup) continue;
      // bit i of nb = pixel (lane - n + i) of the window row is dynamic  (i = n - k)
      uint32_t nb;
      if (lane >= n) nb = (uint32_t)(q0 >> (lane - n));
      else nb = (uint32_t)((q0 << (n - lane)) | (qL >> (64 - (n - lane))));
      nb = dyn ? (nb & kmask) : 0u;
      if (dv == 0) nb &= ~(1u << n);                 // k == 0 is p itself
      if (__ballot(nb != 0) == 0) continue;          // wave-uniform
      COUNT(9, 1)
      if (MOD_ABLATE(c, 8192)) continue;             // (ablation builds: loop header only)
      const int base = qg * PW + NMAX + lane;
      if (same) {
        if (MOD_ABLATE(c, 4096)) continue;                                    // only pixels that still lack their first up-left edge look for it
        if (__ballot(!up & (nb != 0)) == 0) continue;
        uint32_t cand2 = __brev(nb) >> (31 - n);
        if (dv == 0) cand2 &= ~3u;
        if (dv == 1) cand2 &= ~1u;
        uint32_t g2 = 0;
#pragma unroll
        for (int k = 0; k <= NMAX; k++) g2 |= (fabsf(zp - zt[base - k]) > th) ? 0u : (1u << k);
        up = up || ((cand2 & g2) != 0);
        rowup = __ballot(dyn & !up) == 0;
        continue;
      }
      // pass 1, branch-free: labels first, one bit per window position (k = columns to the left)
      int lq[NMAX + 1];
#pragma unroll
      for (int k = 0; k <= NMAX; k++) lq[k] = ld_relaxed(&Lt[base - k]);
      uint32_t dmask = 0;                            // bit k: label differs from this pixel's
#pragma unroll
      for (int k = 0; k <= NMAX; k++) dmask |= (lq[k] != cur) ? (1u << k) : 0u;
      // nb holds pixel (lane - n + i) at bit i: reverse it so that bit k = pixel (lane - k)
      uint32_t cand = __brev(nb) >> (31 - n);
      // (0,0) is p itself; (0,-1) inside the wave is the run link of phase A1, (-1,0) the vertical link of phase A2
      if (dv == 0) cand &= ~3u;                      // (lane 0: its link to the halo cell left of it was made in phase A1 as well)
      if (dv == 1) cand &= ~1u;
      // Inside a blob that phases A1-A3 already merged, every candidate carries this pixel's label and the pixel has its
      // up-left edge: the depth gate cannot change anything, so its LDS reads and compares are skipped (wave-uniform).
      if (__ballot(((cand & dmask) != 0) | (!up & (cand != 0))) == 0) continue;
      COUNT(14, 1)
      COUNT(15, __popcll(__ballot((cand & dmask) != 0)) ? 1 : 0)
      float zq[NMAX + 1];
#pragma unroll
      for (int k = 0; k <= NMAX; k++) zq[k] = zt[base - k];
      uint32_t gmask = 0;                            // bit k: depth gate passes
#pragma unroll
      for (int k = 0; k <= NMAX; k++) gmask |= (fabsf(zp - zq[k]) > th) ? 0u : (1u << k);   // depthDiff gate (:194); NaN links
      const uint32_t vmask = cand & gmask;
      // halo cells are ordinary nodes of the union-find (their ids carry bit 15, so they never become the root of a set that has
      // a tile pixel); phases A1-A3 have linked them like tile pixels, so they mostly carry this pixel's label already
      const bool need_any = (vmask & dmask) != 0;
      up = up || (vmask != 0);
      // pass 2, rare after A1-A3: the unions go to the wave's job queue (run one per lane after the last row, like phase A2's);
      // `cur` is not refreshed meanwhile, so a pair may be queued again from a later window row — a void union, two finds
      if (!MOD_ABLATE(c, 1) && __ballot(need_any)) {
        COUNT(10, 1)
#pragma unroll
        for (int k = 0; k <= NMAX; k++) {
          if (k > n || (dv == 0 && k == 0)) continue;
          const int lab = lq[k];
          const bool need = ((vmask >> k) & 1u) && lab != cur && lab != last;
          if (__ballot(need)) {                        // wave-uniform
            COUNT(12, 1)
            const int pl = wave_prev_i32(lab), pc = wave_prev_i32(cur), pn = wave_prev_i32((int)need);
            push_jobs(need && !(lane > 0 && pn && pc == cur && pl == lab), cur, lab);
            if (need) last = lab;
          }
        }
      }
    }
    STAMP(3)
    upr[j] = up;
  }
  run_jobs();
  lds_barrier();
  STAMP(4)
  // ---- phase C: publish ----------------------------------------------------------------------------------------------
  // interior pixels point at their tile root (plain stores: nobody else writes these entries in this kernel), tile roots
  // get an empty statistics record and a bit in the root plane
  int rootg[RPW], rootc[RPW];
  uint64_t rootbits[RPW];
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int rr = w + NW * j, gy = y0 + rr;
    const bool dyn = (m0[rr + NMAX] >> lane) & 1ull;
    int rg = -1, rc = -1;
    bool isroot = false;
    if (dyn) {
      const int me = (rr + NMAX) * PW + NMAX + lane;
      const int r = lds_find(Lt, me);                // interior by construction
      const int rgr = r / PW, rgc = r - rgr * PW;
      rg = (y0 + rgr - NMAX) * c.W + x0 + rgc - NMAX;
      rc = r;
      if (MOD_CHECK(a, rg >= 0 && (size_t)rg < N && rgr >= NMAX && rgc >= NMAX, 13)) a.parent[fN + (size_t)gy * c.W + x0 + lane] = rg;
      isroot = (r == me);
    }
    rootg[j] = rg; rootc[j] = rc;
    const uint64_t rb = __ballot(isroot);
    rootbits[j] = rb;
    if (lane == 0 && gy < c.H) a.lroot[((size_t)f * c.H + gy) * MW + wi] = rb;
  }
  STAMP(5)
  if (MOD_ABLATE(c, 1 << 18)) return;
  // halo pixels that ended up in a tile component belong to other tiles, whose roots are not known yet: leave one link
  // request (halo pixel, tile root) per connected group of them for k_ccl_link.  First every halo cell is flattened to its root
  // (no union runs any more), so that the neighbour tests below are plain LDS reads.
  {
    const int topcells = n * (64 + n);               // n rows x (n + 64) columns above the tile
    const int total = topcells + TH * n;             // + TH rows x n columns left of it
    auto cell_of = [&](int i, int &gr, int &gc) {
      if (i < topcells) { gr = NMAX - n + i / (64 + n); gc = NMAX - n + i % (64 + n); }
      else { const int t = i - topcells; gr = NMAX + t / n; gc = NMAX - n + t % n; }
    };
    auto halo_dyn = [&](int gr, int gc) {
      const int gx = x0 - NMAX + gc;
      const uint64_t mw = (gc >= NMAX) ? m0[gr] : mL[gr];
      return gx >= 0 && ((mw >> (gx & 63)) & 1ull);
    };
    const bool any_halo = !MOD_ABLATE(c, 1024) && s_anyhalo;   // workgroup-uniform
    for (int i0 = 0; i0 < total && any_halo; i0 += NW * 64) {
      const int i = i0 + tid;
      if (i < total) {
        int gr, gc;
        cell_of(i, gr, gc);
        if (halo_dyn(gr, gc)) { const int cell = gr * PW + gc; Lt[cell] = lds_find(Lt, cell | kHaloBit); }
      }
    }
    lds_barrier();
    uint2 *req = a.requests + ((size_t)f * tiles_y * tiles_x + (size_t)ty * tiles_x + wi) * a.req_cap;
    for (int i0 = 0; i0 < total && any_halo; i0 += NW * 64) {
      const int i = i0 + tid;
      bool linked = false;
      int hg = 0, rg = 0;
      if (i < total) {
        int gr, gc;
        cell_of(i, gr, gc);
        if (halo_dyn(gr, gc)) {
          const int r = Lt[gr * PW + gc];
          if (!(r & kHaloBit)) {                       // in a component that has a tile pixel (halo cells may also be linked among themselves)
            // A halo cell whose left or upper neighbour is a halo cell of the SAME set with a direct edge to it (dynamic, depth gate
            // passes) leaves the request to that neighbour: the edge between the two is an edge of the image graph that the tile
            // owning this cell sees itself, so one request per connected group of halo cells (its top-left-most cell) is enough.
            const float zme = zt[gr * PW + gc];
            auto covered = [&](int gr2, int gc2) {
              if (gr2 < NMAX - n || gc2 < NMAX - n || !halo_dyn(gr2, gc2)) return false;
              const int cell2 = gr2 * PW + gc2;
              return !(fabsf(zme - zt[cell2]) > th) && Lt[cell2] == r;
            };
            if (!(covered(gr, gc - 1) || covered(gr - 1, gc))) {
              const int rgr = r / PW, rgc = r - rgr * PW;
              rg = (y0 + rgr - NMAX) * c.W + x0 + rgc - NMAX;
              hg = (y0 - NMAX + gr) * c.W + x0 - NMAX + gc;
              linked = true;
            }
          }
        }
      }
      const uint64_t lb = __ballot(linked);
      if (lb) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_nreq, __popcll((unsigned long long)lb));
        base = __builtin_amdgcn_readfirstlane(base);
        if (linked) {
          const int slot = base + __popcll((unsigned long long)(lb & ((1ull << lane) - 1ull)));
          if (MOD_CHECK(a, slot >= 0 && slot < a.req_cap, 12) && MOD_CHECK(a, rg >= 0 && (size_t)rg < N && hg >= 0 && (size_t)hg < N, 13))
            req[slot] = make_uint2((uint32_t)hg, (uint32_t)rg);
        }
      }
    }
  }
  STAMP(6)
  lds_barrier();                                     // every find on Lt is done: root cells can be re-used as slot tags
  // ---- phase D: partial statistics of the tile's components --------------------------------------------------------
  // The tile owns its roots' records, so they are reduced in LDS slots and stored once — no global atomics.  Roots beyond
  // kSlots (very fragmented tiles) fall back to initialise-then-atomics on the global record.
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const bool isroot = (rootbits[j] >> lane) & 1ull;
    if (isroot) {
      const int slot = atomicAdd(&s_nslots, 1);
      RootRec rec;
      rec.size = 0; rec.key = kKeyNone;
      if (slot < kSlots) { srec[slot] = rec; sroot[slot] = rootg[j]; Lt[rootc[j]] = -(slot + 1); }
      else { a.rsize[fN + rootg[j]] = 0; a.rkey[fN + rootg[j]] = kKeyNone; }
    }
  }
  lds_barrier();     // slot tags visible
  if (s_nslots > kSlots) __syncthreads();   // overflow records must have reached L2 before any wave's atomics on them
  STAMP(7)
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int rr = w + NW * j, gy = y0 + rr;
    if (m0[rr + NMAX] == 0 || MOD_ABLATE(c, 520)) continue;                // wave-uniform
    const int rg = rootg[j];
    uint32_t key = (uint32_t)kKeyNone;
    int slot = -1, over = -1;
    if (rg >= 0) {
      const size_t gp = (size_t)gy * c.W + x0 + lane;
      if (upr[j]) key = (uint32_t)gp;
      const int tag = Lt[rootc[j]];
      if (tag < 0) slot = -tag - 1; else over = rg;
    }
    wave_accumulate(&srec[0].size, &srec[0].key, 2, slot, key, lane);
    if (__ballot(over >= 0)) wave_accumulate(a.rsize + fN, a.rkey + fN, 1, over, key, lane);
  }
  lds_barrier();
  {
    const int ns = min(s_nslots, kSlots);
    if (tid < ns) { a.rsize[fN + sroot[tid]] = srec[tid].size; a.rkey[fN + sroot[tid]] = srec[tid].key; }
  }
  STAMP(8)
#undef STAMP
#undef COUNT
  if (tid == 0) hdr[1] = s_nreq;                   // s_nreq is final: the barrier after the request loop has passed (hdr[0] is 1 already)
}

// The same tile body for the tiles of a LIST (k_ccl_bits leaves the tiles it cannot decide there): a fixed number of workgroups,
// each pulling the next tile with one atomic — the list is a fifth of the active tiles, one workgroup per grid tile would spend
// the launch on 460 k header reads.  counters[4] = length of the list, counters[3] = cursor (both zeroed by the caller).
template <int TH, int NMAX, int NW, bool EXACT>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_ccl_tile_list(DevCam c, ClArgs a, int tiles_x, int tiles_y) {
  __shared__ int s_next;
  const int count = a.counters[4];
  // workgroup i starts with list entry i (a short list — the usual case — costs the other workgroups one scalar load), further
  // entries are pulled with the cursor, which starts behind the statically assigned ones
  int i = (int)blockIdx.x;
  const int per_frame = tiles_x * tiles_y;
  while (i < count) {
    const uint32_t t = a.tilelist[i];
    const int f = (int)(t / (uint32_t)per_frame), r = (int)(t - (uint32_t)f * (uint32_t)per_frame), ty = r / tiles_x, wi = r - ty * tiles_x;
    ccl_tile_body<TH, NMAX, NW, EXACT>(c, a, wi, ty, f, tiles_x, tiles_y);
    __syncthreads();                                 // the next tile re-uses the LDS arrays and s_next
    if (threadIdx.x == 0 && threadIdx.y == 0) s_next = (int)gridDim.x + atomicAdd(&a.counters[3], 1);
    __syncthreads();
    i = __builtin_amdgcn_readfirstlane(s_next);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_ccl_bits<n> — the tile stage on BIT PLANES, for the tiles whose depth gates are decided by a few DEPTH CLASSES; one instance per
// neighbor_distance n = 1 .. 10 (Clusterer.cfg:11; the description uses the default n = 4).
//
// comparePoints links two dynamic pixels unless |z_p - z_q| > depth_diff (clusterer_nodelet.cpp:186-219).  When the dynamic cells
// of a tile and its halo, sorted by depth, fall into stretches that are each no wider than depth_diff and clear of each other by
// more than it (a tile inside one object: one stretch; at an object's rim: two — 99 % of the active tiles of the synthetic street
// scene need at most four), every gate inside a stretch passes and every gate between two fires: the tile's components are those of
// each stretch's MASK under the up-left 5 x 5 window.  They are found without touching a pixel: ONE wave per tile, lane = grid row
// (4 halo rows above + 16 tile rows), a row = 68 bits (4 halo columns + 64) in three registers, so that
//   * "has an up-left edge" (first_edge_key), "is somebody's up-left neighbour" and the closing of <= 3-cell gaps are a few
//     shifts / ORs per row, the rows above / below arrive by DPP wave shifts;
//   * the usual tile (every row one closed run, every row linked to a row above) is ONE component by inspection;
//   * anything else is flooded component by component from its first pixel in raster order (which is its root): per sweep the set
//     grows by the window in all rows at once and fills the closed runs it touches with a carry chain (c + s ripples through a
//     run of ones), 5-6 sweeps for a tile-high blob;
//   * publishing turns a component's row bits into lane predicates (v_readlane -> exec): parent[p] = root, one (size,
//     first_edge_key) record, root bits, one link request per connected group of halo cells — k_ccl_tile's contract.
// ~700 wave-instructions per such tile against ~17 000 of the union-find kernel.  Tiles that fail the depth test go to a list for
// k_ccl_tile_list.  Model of the bit algorithm against brute force: tests/models/ccl_bits_model.py.
namespace bits {
constexpr int TH = 16;

struct Row3 { uint32_t h, a, b; };   // the n halo columns x0-n .. x0-1 in the top n bits of h, x0 .. x0+31 in a, x0+32 .. x0+63 in b

__device__ __forceinline__ Row3 operator|(Row3 x, Row3 y) { return {x.h | y.h, x.a | y.a, x.b | y.b}; }
__device__ __forceinline__ Row3 operator&(Row3 x, Row3 y) { return {x.h & y.h, x.a & y.a, x.b & y.b}; }
__device__ __forceinline__ Row3 operator^(Row3 x, Row3 y) { return {x.h ^ y.h, x.a ^ y.a, x.b ^ y.b}; }
__device__ __forceinline__ Row3 andn(Row3 x, Row3 y) { return {x.h & ~y.h, x.a & ~y.a, x.b & ~y.b}; }
__device__ __forceinline__ bool any(Row3 x) { return (x.h | x.a | x.b) != 0u; }
// Shifts by K >= 1 columns.  NH = number of halo columns (= neighbor_distance): they sit in the top NH bits of h.
template <int K> __device__ __forceinline__ Row3 shl(Row3 v) {      // towards larger x
  return {v.h << K, __builtin_amdgcn_alignbit(v.a, v.h, 32 - K), __builtin_amdgcn_alignbit(v.b, v.a, 32 - K)};
}
template <int K, int NH> __device__ __forceinline__ Row3 shr(Row3 v) {      // towards smaller x; cells left of column x0 - NH do not exist
  return {__builtin_amdgcn_alignbit(v.a, v.h, K) & (~0u << (32 - NH)), __builtin_amdgcn_alignbit(v.b, v.a, K), v.b >> K};
}
// the row above / below arrives (zero into the first / last lane).  All 64 lanes must be active.
__device__ __forceinline__ uint32_t dn1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false); }   // wave_shr:1
__device__ __forceinline__ uint32_t up1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false); }   // wave_shl:1
__device__ __forceinline__ Row3 row_dn(Row3 v) { return {dn1(v.h), dn1(v.a), dn1(v.b)}; }
__device__ __forceinline__ Row3 row_up(Row3 v) { return {up1(v.h), up1(v.a), up1(v.b)}; }
template <int K, bool DN> __device__ __forceinline__ Row3 row_shift(Row3 v) {   // K rows
  if constexpr (K == 0) return v;
  else return row_shift<K - 1, DN>(DN ? row_dn(v) : row_up(v));
}
// OR of v shifted by 0 .. N - 1 (columns: DIR 0 right, 1 left; rows: DIR 2 down, 3 up) by doubling: a set that covers shifts
// 0 .. COV is ORed with itself shifted by min(COV + 1, rest).  N = 4: two steps (1, 2); N = 10: four (1, 2, 4, 2).
template <int COV, int N, int DIR> __device__ __forceinline__ Row3 grow(Row3 y) {
  if constexpr (COV >= N - 1) return y;
  else {
    constexpr int S = (COV + 1 < N - 1 - COV) ? COV + 1 : N - 1 - COV;
    Row3 t;
    if constexpr (DIR == 0) t = shl<S>(y);
    else if constexpr (DIR == 1) t = shr<S, N>(y);
    else t = row_shift<S, DIR == 2>(y);
    return grow<COV + S, N, DIR>(y | t);
  }
}
// v | v << 1 | .. | v << N; x: the same without v itself
template <int N> __device__ __forceinline__ void dil_r(Row3 v, Row3 &all, Row3 &x) { x = shl<1>(grow<0, N, 0>(v)); all = v | x; }
template <int N> __device__ __forceinline__ void dil_l(Row3 v, Row3 &all, Row3 &x) { x = shr<1, N>(grow<0, N, 1>(v)); all = v | x; }
// OR over dv = 1 .. N of the rows dv above (DN) / below
template <int N, bool DN> __device__ __forceinline__ Row3 vert(Row3 v) { return row_shift<1, DN>(grow<0, N, DN ? 2 : 3>(v)); }
// M with the gaps of <= N - 1 cells between two dynamic cells closed: a cell is in the result iff a dynamic cell lies i to its left
// and one j to its right with i + j <= N (cells of one run of the result are chained by same-row links)
template <int I, int N> __device__ __forceinline__ Row3 closed_rec(Row3 M, Row3 lprev) {   // term I: (dynamic within I to the left) & (dynamic N - I to the right)
  Row3 l, r;
  if constexpr (I == 0) l = M; else l = lprev | shl<I>(M);
  if constexpr (I == N) r = M; else r = shr<N - I, N>(M);
  const Row3 t = l & r;
  if constexpr (I == N) return t;
  else return t | closed_rec<I + 1, N>(M, l);
}
template <int N> __device__ __forceinline__ Row3 closed(Row3 M) { return closed_rec<0, N>(M, M); }
__device__ __forceinline__ Row3 rev(Row3 v) { return {__builtin_bitreverse32(v.b), __builtin_bitreverse32(v.a), __builtin_bitreverse32(v.h)}; }
// all bits of the runs of c that hold a bit of s (s subset of c), from the lowest such bit upwards: c + s ripples through a run
__device__ __forceinline__ Row3 fill_up(Row3 c, Row3 s) {
  uint32_t c1, c2, c3;
  Row3 t;
  t.h = __builtin_addc(c.h, s.h, 0u, &c1);
  t.a = __builtin_addc(c.a, s.a, c1, &c2);
  t.b = __builtin_addc(c.b, s.b, c2, &c3);
  return ((t ^ c) & c) | s;
}
__device__ __forceinline__ int popc3(Row3 v) { return __popc(v.h) + __popc(v.a) + __popc(v.b); }

// min / max of depths that are never signalling NaNs where it matters: a dynamic cell's depth is a number (checked for a caller's
// cloud, guaranteed by the fused kernel) and everything else has been replaced by a quiet NaN, which v_min / v_max pass over —
// the plain instructions, without the canonicalising v_max x, x that fminf / fmaxf put in front of every loaded value
__device__ __forceinline__ float zmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float zmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// the F32 of lane l (v_readlane moves bit patterns; the builtin is typed int)
__device__ __forceinline__ float lane_f32(float v, int l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); }
__device__ __forceinline__ float wave_fmin(float v) {
  v = fminf(v, __uint_as_float(MOD_DPP(__float_as_uint(v), 0xB1))); v = fminf(v, __uint_as_float(MOD_DPP(__float_as_uint(v), 0x4E)));
  v = fminf(v, __uint_as_float(MOD_DPP(__float_as_uint(v), 0x141))); v = fminf(v, __uint_as_float(MOD_DPP(__float_as_uint(v), 0x140)));
  const float a = lane_f32(v, 0), b = lane_f32(v, 16), c = lane_f32(v, 32), d = lane_f32(v, 48);
  return fminf(fminf(a, b), fminf(c, d));
}
__device__ __forceinline__ float wave_fmax(float v) {
  v = fmaxf(v, __uint_as_float(MOD_DPP(__float_as_uint(v), 0xB1))); v = fmaxf(v, __uint_as_float(MOD_DPP(__float_as_uint(v), 0x4E)));
  v = fmaxf(v, __uint_as_float(MOD_DPP(__float_as_uint(v), 0x141))); v = fmaxf(v, __uint_as_float(MOD_DPP(__float_as_uint(v), 0x140)));
  const float a = lane_f32(v, 0), b = lane_f32(v, 16), c = lane_f32(v, 32), d = lane_f32(v, 48);
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
// sum over lanes 0 .. 31 (the grid rows live in lanes 0 .. 15 + neighbor_distance)
__device__ __forceinline__ int wave_sum_lo32(int v) {
  v += (int)MOD_DPP(v, 0xB1); v += (int)MOD_DPP(v, 0x4E); v += (int)MOD_DPP(v, 0x141); v += (int)MOD_DPP(v, 0x140);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16);
}
__device__ __forceinline__ uint64_t lane_bits(uint32_t lo, uint32_t hi, int l) {   // words of lane l as one scalar mask
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, l) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)lo, l);
}

// one tile (= one wave) per workgroup: 1 / 2 / 4 / 8 tiles per workgroup measured 0.451 / 0.455 / 0.521 / 0.642 ms per 512 pairs
// (in-process A/B): a workgroup's slot and LDS stay occupied until its slowest wave is done, and three quarters of the tiles are empty.
// (One wave walking a STRIP of 2 / 4 / 5 / 10 tiles, to save the empty tiles' launches — 0.17 ms per 512 pairs when no pixel is
// dynamic —: 0.495 / 0.495 / 0.510 / 0.515 against 0.435: active tiles sit next to each other and would queue behind one wave.)
constexpr int kTilesPerBlock = 1, kMaxClasses = 4, kMaxComps = 32;

// Global accesses of this kernel: ONE wave-uniform base per plane (frame's plane, in SGPRs) + a 32-bit byte offset per lane — the
// `global_load / global_store v, v_off, s[base]` form.  (A frame's planes span less than 2^29 bytes: mod_create caps W * H at 2^27.)
// Written as base + zext(offset) with the row's share folded into the OFFSET: left to itself the compiler adds the lane offset to
// the base first and then keeps one 64-bit VGPR address per row alive across the class loop (32 registers for the 16 parent rows).
template <class T> __device__ __forceinline__ T ldo(const void *base, uint32_t byte_off) { return *(const T *)((const char *)base + byte_off); }
template <class T> __device__ __forceinline__ void sto(void *base, uint32_t byte_off, T v) { *(T *)((char *)base + byte_off) = v; }

template <int HL>                                                     // HL = neighbor_distance = halo rows above = halo columns left
__global__ __launch_bounds__(64 * kTilesPerBlock) void k_ccl_bits(DevCam c, ClArgs a, int tiles_x, int tiles_y) {
  constexpr int PH = TH + HL;                                          // grid rows = lanes in use
  const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
  const int wi = blockIdx.x * kTilesPerBlock + wv, ty = blockIdx.y, f = blockIdx.z;
  if (wi >= tiles_x) return;
  const size_t tix = (size_t)f * tiles_y * tiles_x + (size_t)ty * tiles_x + wi;
  int *hdr = a.tilehdr + tix * 2;
  if (__builtin_amdgcn_readfirstlane(hdr[0]) == 0) return;          // no dynamic pixel in the tile
  const int x0 = wi * 64, y0 = ty * TH, MW = c.mask_words, W = c.W, H = c.H;
  const size_t N = (size_t)W * H, fN = (size_t)f * N;
  // ---- the mask words of the 20 grid rows, lane = row -------------------------------------------------------------------------
  Row3 Mrem;                                                           // dynamic cells that no depth class holds yet
  {
    const int gy = y0 - HL + lane;
    const bool inrow = lane < PH && gy >= 0 && gy < H;
    const uint64_t *mr = a.mask + ((size_t)f * H + (inrow ? gy : 0)) * MW;
    const uint64_t q0 = mr[wi], qL = mr[max(wi - 1, 0)];              // unconditional loads, clamped addresses
    Mrem.h = (inrow && wi > 0) ? ((uint32_t)(qL >> 32) & (~0u << (32 - HL))) : 0u;
    Mrem.a = inrow ? (uint32_t)q0 : 0u;
    Mrem.b = inrow ? (uint32_t)(q0 >> 32) : 0u;
  }
  const bool il = lane >= HL && lane < PH;                             // a tile row
  const float th = c.depth_th, qnan = __uint_as_float(0x7fc00000u);
  // (uniform row base + 32-bit lane offset: the `global_load v, v_off, s[base]` form, no 64-bit address arithmetic per lane)
  const uint32_t oc = 4u * (uint32_t)min(x0 + lane, W - 1);
  uint32_t rootA = 0u, rootB = 0u;                                     // root bits of the tile rows (lane = row)
  int nreq = 0, ncomp = 0;
  uint2 *req = a.requests + tix * a.req_cap;
  float hi_prev = 0.0f;
  bool bail = false;
  // ---- depth classes.  comparePoints links two dynamic pixels unless |z_p - z_q| > depth_diff.  Sort the dynamic cells of the grid
  // by depth in thought: a CLASS is a stretch that spans at most depth_diff (every gate inside it passes: |z_p - z_q| <= max - min,
  // and F32 subtraction is monotone) and lies more than depth_diff from the next cell on either side (every gate that leaves it
  // fires).  Then the tile's components are those of each class's MASK, class by class.  Classes are peeled off from the
  // nearest: lo = smallest depth left, members = cells with !(z - lo > th); the tile goes to the union-find kernel when a class
  // is not clear of the next one, when more than kMaxClasses are needed, or when a dynamic cell has a NaN depth (NaN links with
  // everything).  One class — a tile inside one object — is the usual case; object rims have two. --------------------------------
  // The depths of the grid are read from HBM once: the first sweep (smallest / largest depth of the tile) works on them as they
  // arrive and parks the 64 tile columns in LDS (5 KB per wave) for the sweeps of a tile with more than one class.  The four halo
  // columns stay in registers in the ROW's lane (lane = row: four loads serve all 20 rows), next to the row's halo bits.
  const float *zplane = a.z + fN;
  __shared__ float zlds[kTilesPerBlock][PH][64];
  float(*zl)[64] = zlds[wv];
  float zh[HL];
  float lo = qnan, hi = qnan;
  // Round 5: the fused scene-flow kernel leaves, next to every non-zero mask word, the smallest and largest depth of the word's
  // dynamic pixels (ClArgs.zrange).  The union of the ranges of this tile's words (word wi of the grid rows; word wi - 1 of the rows
  // whose halo columns hold a dynamic cell — that word spans 64 columns, the halo HL of them, so the union can only be WIDER than the
  // grid's own range) no wider than depth_diff means ONE class (F32 subtraction is monotone: hi - lo of a subset cannot round
  // above that of its superset) and NO depth row is loaded at all; otherwise, and for a caller's cloud (zrange == null), the
  // depths are read as before.  A sufficient test only: the result is the same either way.
  bool need_z = true;
  if (a.zrange) {                                                      // wave-uniform
    const int gy = y0 - HL + lane;
    const bool inrow = lane < PH && gy >= 0 && gy < H;
    const float2 *zr = a.zrange + ((size_t)f * H + (inrow ? gy : 0)) * MW;
    const float2 r0 = zr[wi], rL = zr[max(wi - 1, 0)];                 // unconditional loads, clamped addresses (beside the mask words')
    const bool u0 = (Mrem.a | Mrem.b) != 0u, uL = Mrem.h != 0u;        // Mrem is zero outside the grid's rows
    lo = zmin(u0 ? r0.x : qnan, uL ? rL.x : qnan); hi = zmax(u0 ? r0.y : qnan, uL ? rL.y : qnan);
    lo = wave_fmin(lo); hi = wave_fmax(hi);
    need_z = hi - lo > th;                                             // (a NaN — impossible here — would take the row path's answer below)
  }
  if (need_z) {
    lo = qnan; hi = qnan;
    {
      const int gy = min(max(y0 - HL + lane, 0), H - 1);               // (lanes >= PH read the clamped last row: never used)
      // wi == 0 has no halo columns (Mrem.h == 0); an image narrower than HL still reads inside its rows (values unused)
      const int hx = max(x0 - HL, 0);
#pragma unroll
      for (int j = 0; j < HL; j++) zh[j] = ldo<float>(zplane, 4u * (uint32_t)(gy * W + min(hx + j, W - 1)));
    }
    uint64_t nanb = 0ull;
    constexpr int HB = (PH + 1) / 2;                                   // two batches of rows (10 + 10 at HL = 4)
#pragma unroll
    for (int g0 = 0; g0 < PH; g0 += HB) {
      float zr[HB];
#pragma unroll
      for (int i = 0; i < HB; i++) {
        const int gy = min(max(y0 - HL + g0 + i, 0), H - 1);
        zr[i] = ldo<float>(zplane, oc + 4u * (uint32_t)(gy * W));
      }
#pragma unroll
      for (int i = 0; i < HB; i++) {
        const int gr = g0 + i;
        if (gr >= PH) break;
        const bool dyn = __builtin_amdgcn_inverse_ballot_w64(lane_bits(Mrem.a, Mrem.b, gr));
        const float z1 = dyn ? zr[i] : qnan;
        lo = zmin(lo, z1); hi = zmax(hi, z1);
        zl[gr][lane] = zr[i];
        // the fused scene-flow kernel never marks a pixel without a finite depth as dynamic; a caller's cloud may
        if (!a.xy_from_z) nanb |= __ballot(dyn & (zr[i] != zr[i]));
      }
    }
#pragma unroll
    for (int j = 0; j < HL; j++) {
      const bool hd = (Mrem.h >> (32 - HL + j)) & 1u;
      const float z2 = hd ? zh[j] : qnan;
      lo = zmin(lo, z2); hi = zmax(hi, z2);
      if (!a.xy_from_z) nanb |= __ballot(hd & (zh[j] != zh[j]));
    }
    if (nanb != 0ull) {                                                // wave-uniform: NaN links with everything
      if (lane == 0) a.tilelist[atomicAdd(&a.counters[4], 1)] = (uint32_t)tix;
      return;
    }
  } else {
#pragma unroll
    for (int j = 0; j < HL; j++) zh[j] = qnan;                         // never read: one class ends the loop below after its first pass
  }
  for (int pass = 0;; pass++) {                                        // wave-uniform
    if (pass == kMaxClasses) { bail = true; break; }
    if (pass > 0) {                                                    // smallest / largest depth of what is left
      lo = qnan; hi = qnan;
#pragma unroll 4
      for (int gr = 0; gr < PH; gr++) {
        const bool dyn = __builtin_amdgcn_inverse_ballot_w64(lane_bits(Mrem.a, Mrem.b, gr));
        const float z1 = dyn ? zl[gr][lane] : qnan;
        lo = zmin(lo, z1); hi = zmax(hi, z1);
      }
#pragma unroll
      for (int j = 0; j < HL; j++) {
        const float z2 = ((Mrem.h >> (32 - HL + j)) & 1u) ? zh[j] : qnan;
        lo = zmin(lo, z2); hi = zmax(hi, z2);
      }
    }
    lo = wave_fmin(lo); hi = wave_fmax(hi);
    Row3 M;                                                            // the class
    if (!(hi - lo > th)) { M = Mrem; Mrem = {0u, 0u, 0u}; }            // everything that is left (wave-uniform)
    else {
      M = {0u, 0u, 0u};
      float chi = qnan;
#pragma unroll 4
      for (int gr = 0; gr < PH; gr++) {
        const bool dyn = __builtin_amdgcn_inverse_ballot_w64(lane_bits(Mrem.a, Mrem.b, gr));
        const float z1 = zl[gr][lane];
        const bool in1 = dyn & !(z1 - lo > th);
        const uint64_t b1 = __ballot(in1);
        const bool me = lane == gr;                                    // into the row's lane (two selects)
        M.a = me ? (uint32_t)b1 : M.a; M.b = me ? (uint32_t)(b1 >> 32) : M.b;
        chi = zmax(chi, in1 ? z1 : qnan);
      }
#pragma unroll
      for (int j = 0; j < HL; j++) {                                   // the halo columns, in the row's own lane
        const bool in2 = ((Mrem.h >> (32 - HL + j)) & 1u) & !(zh[j] - lo > th);
        M.h |= in2 ? (1u << (32 - HL + j)) : 0u;
        chi = zmax(chi, in2 ? zh[j] : qnan);
      }
      hi = wave_fmax(chi);
      Mrem = andn(Mrem, M);
    }
    if (pass > 0 && !(lo - hi_prev > th)) { bail = true; break; }      // the class before this one was not clear of it
    hi_prev = hi;
  // ---- per-row facts ---------------------------------------------------------------------------------------------------------------
  Row3 drA, drX, dlA, dlX;
  dil_r<HL>(M, drA, drX);
  dil_l<HL>(M, dlA, dlX);
  const Row3 above = vert<HL, true>(drA);                              // cells that have a dynamic cell up-left in one of the HL rows above
  const Row3 UL = M & (drX | above);                                   // has an up-left edge
  const Row3 E = UL | (M & (dlX | vert<HL, false>(dlA)));              // has any edge
  const Row3 C = closed<HL>(M);                                        // M with gaps of < HL cells closed: one run = one chain of same-row links
  const Row3 rC = rev(C);
  Row3 R = {0u, il ? (M.a & E.a) : 0u, il ? (M.b & E.b) : 0u};         // tile pixels with an edge that no component holds yet
  const uint32_t Za = il ? (M.a & ~E.a) : 0u, Zb = il ? (M.b & ~E.b) : 0u;   // tile pixels without any edge: roots of their own
  rootA |= Za; rootB |= Zb;
  // one component by inspection?  Every non-empty row is ONE closed run whose cells all have an edge, and every non-empty row
  // but the first has a link to a row above: by induction over the rows all dynamic cells of the grid are connected.
  bool single;
  {
    const uint64_t ne = __ballot(any(M));
    const int first = __builtin_ctzll(ne);                              // ne != 0: the tile has a dynamic pixel
    const Row3 starts = andn(C, shl<1>(C));
    const bool ok = !any(M) || (popc3(starts) == 1 && !any(E ^ M) && (lane == first || any(M & above)));
    single = __ballot(!ok) == 0ull && __ballot(any(R)) != 0ull;
  }
  for (;;) {                                                           // wave-uniform loop over the components with a tile pixel
    const uint64_t rrows = __ballot(any(R));
    if (rrows == 0ull) break;
    Row3 S;
    if (single) S = M;
    else {
      const int sr = __builtin_ctzll(rrows);                           // first pixel in raster order of what is left: the seed
      const uint32_t sa = (uint32_t)__builtin_amdgcn_readlane((int)R.a, sr), sb = (uint32_t)__builtin_amdgcn_readlane((int)R.b, sr);
      const uint32_t ba = sa & (0u - sa), bb = sa ? 0u : (sb & (0u - sb));
      S = {0u, lane == sr ? ba : 0u, lane == sr ? bb : 0u};
      for (;;) {                                                       // flood: S only grows, inside M
        Row3 rA, rX, lA, lX;
        dil_r<HL>(S, rA, rX);
        dil_l<HL>(S, lA, lX);
        const Row3 reach = M & (rA | lA | vert<HL, true>(rA) | vert<HL, false>(lA));
        const Row3 S2 = (fill_up(C, reach) | rev(fill_up(rC, rev(reach)))) & M;
        const bool grew = any(S2 ^ S);
        S = S2;
        if (__ballot(grew) == 0ull) break;
      }
    }
    // ---- publish the component ----
    const uint64_t irows = __ballot(il && (S.a | S.b) != 0u);
    const int rr = __builtin_ctzll(irows);                              // S holds a tile pixel (its seed, or R != 0)
    const uint32_t fa = (uint32_t)__builtin_amdgcn_readlane((int)S.a, rr), fb = (uint32_t)__builtin_amdgcn_readlane((int)S.b, rr);
    const int rcol = fa ? __builtin_ctz(fa) : 32 + __builtin_ctz(fb);
    const int rootg = (y0 + rr - HL) * W + x0 + rcol;                  // the component's first tile pixel in raster order
    if (lane == rr) { if (rcol < 32) rootA |= 1u << rcol; else rootB |= 1u << (rcol - 32); }
    int *pplane = a.parent + fN;
    uint32_t po = 4u * (uint32_t)(y0 * W + x0 + lane);                  // this lane's pixel in the tile's first row
    asm volatile("" : "+v"(po));                                       // (computed here, per component: not 16 row offsets held across the loop)
#pragma unroll
    for (int j = 0; j < TH; j++) {
      const uint64_t bitsj = lane_bits(S.a, S.b, HL + j);
      if (bitsj != 0ull && __builtin_amdgcn_inverse_ballot_w64(bitsj))
        sto<int>(pplane, po + 4u * (uint32_t)(j * W), rootg);
    }
    {
      const int cnt = il ? __popc(S.a) + __popc(S.b) : 0;
      const uint32_t ka = S.a & UL.a, kb = S.b & UL.b;
      uint32_t key = (uint32_t)kKeyNone;
      if (il && (ka | kb)) key = (uint32_t)((y0 + lane - HL) * W + x0 + (ka ? __builtin_ctz(ka) : 32 + __builtin_ctz(kb)));
      const int size = wave_sum_lo32(cnt);
      key = wave_min_u32(key);
      if (lane == 0) { sto<int>(a.rsize + fN, 4u * (uint32_t)rootg, size); sto<int>(a.rkey + fN, 4u * (uint32_t)rootg, (int)key); }
    }
    // halo cells of the component belong to other tiles: one link request (halo cell, root) per group of halo cells that are
    // direct neighbours (a cell whose left or upper neighbour is a halo cell of the set leaves it to that neighbour: the edge
    // between them is seen by the tile that owns the cell) — k_ccl_tile's rule
    {
      const Row3 HS = {S.h, lane < HL ? S.a : 0u, lane < HL ? S.b : 0u};
      const Row3 em = andn(HS, shl<1>(HS) | row_dn(HS));
      uint64_t todo = __ballot(any(em));
      while (todo) {                                                   // wave-uniform: the few grid rows that emit
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint64_t eb = lane_bits(em.a, em.b, l);
        const uint32_t eh = (uint32_t)__builtin_amdgcn_readlane((int)em.h, l) >> (32 - HL);
        const int hgrow = (y0 - HL + l) * W + x0;
        if (__builtin_amdgcn_inverse_ballot_w64((uint64_t)eh)) {       // left-halo columns x0 - HL + lane, lanes 0 .. HL - 1
          const int slot = nreq + __popc(eh & ((1u << lane) - 1u));
          if (MOD_CHECK(a, slot < a.req_cap, 12)) req[slot] = make_uint2((uint32_t)(hgrow - HL + lane), (uint32_t)rootg);
        }
        nreq += __popc(eh);
        if (eb != 0ull && __builtin_amdgcn_inverse_ballot_w64(eb)) {
          const int slot = nreq + __popcll((unsigned long long)(eb & ((1ull << lane) - 1ull)));
          if (MOD_CHECK(a, slot < a.req_cap, 12)) req[slot] = make_uint2((uint32_t)(hgrow + lane), (uint32_t)rootg);
        }
        nreq += __popcll((unsigned long long)eb);
      }
    }
    if (single) break;
    R = andn(R, S);
    // a tile in dozens of pieces (noise at a small window) is flooded piece by piece: beyond kMaxComps the union-find kernel,
    // which takes them all at once, is the cheaper one
    if (++ncomp >= kMaxComps && __ballot(any(R)) != 0ull) { bail = true; break; }
  }
  // ---- tile pixels without any edge: each its own root with an empty key (the reference never labels them) ----
  {
    uint64_t todo = __ballot((Za | Zb) != 0u);
    while (todo) {
      const int l = __builtin_ctzll(todo);
      todo &= todo - 1ull;
      if (__builtin_amdgcn_inverse_ballot_w64(lane_bits(Za, Zb, l))) {
        const int p = (y0 + l - HL) * W + x0 + lane;
        sto<int>(a.parent + fN, 4u * (uint32_t)p, p); sto<int>(a.rsize + fN, 4u * (uint32_t)p, 1); sto<int>(a.rkey + fN, 4u * (uint32_t)p, kKeyNone);
      }
    }
  }
    if (bail || __ballot(any(Mrem)) == 0ull) break;                    // wave-uniform: every dynamic cell is in a class
  }
  if (bail) {                                                          // wave-uniform: leave the tile to the union-find kernel, which
    if (lane == 0) a.tilelist[atomicAdd(&a.counters[4], 1)] = (uint32_t)tix;   // rewrites whatever classes published before the bail
    return;
  }
  if (il && y0 + lane - HL < H) a.lroot[((size_t)f * H + (y0 + lane - HL)) * MW + wi] = ((uint64_t)rootB << 32) | rootA;
  if (lane == 0) { hdr[1] = nreq; hdr[0] = 2; }                       // 2: done here (k_ccl_tile_list never sees the tile)
}
}  // namespace bits


// Tile headers from a mask plane that did not come out of the fused scene-flow kernel (mod_cluster_dev): {1, 0} for tiles with a
// dynamic pixel, {0, 0} for the others.  One thread per tile.
template <int TH>
__global__ __launch_bounds__(256) void k_tile_flags(DevCam c, ClArgs a, int tiles_x, int tiles_per_frame, int total) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int f = t / tiles_per_frame, tt = t - f * tiles_per_frame, ty = tt / tiles_x, wi = tt - ty * tiles_x;
  const uint64_t *m = a.mask + ((size_t)f * c.H + (size_t)ty * TH) * c.mask_words + wi;
  const int rows = min(TH, c.H - ty * TH);
  uint64_t any = 0;
#pragma unroll 4
  for (int r = 0; r < rows; r++) any |= m[(size_t)r * c.mask_words];
  a.tilehdr[(size_t)t * 2] = any != 0 ? 1 : 0;
  a.tilehdr[(size_t)t * 2 + 1] = 0;
}

// Cross-tile links.  One workgroup per kLinkTiles consecutive tiles (most tiles have no dynamic pixel and no request: a
// workgroup each would spend the kernel on dispatch).  The tiles' request lists are walked as ONE index space by all 256
// threads — every request is a chain of dependent global accesses (request, parent of the halo pixel, two root searches, a
// hook), so what matters is how many are in flight, not which wave owns which tile.  A request is (halo pixel h, tile root r):
// h's own tile has published parent[h] = its tile root by now, so the union is between two tile roots — all parent writes
// here are atomicMin hooks on root entries.  Consecutive requests usually name the same pair; only the first lane of a run acts.
constexpr int kLinkTiles = 32;   // 4 / 8 / 16 / 32 / 64 tiles per workgroup: 0.138 / 0.108 / 0.093 / 0.083 / 0.093 ms per 512 pairs (in-process A/B)

__global__ __launch_bounds__(256) void k_ccl_link(DevCam c, ClArgs a, int tiles_per_frame) {
  const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int t0 = blockIdx.x * kLinkTiles;
  int cnt[kLinkTiles], total = 0;
#pragma unroll
  for (int u = 0; u < kLinkTiles; u++) {               // block-uniform scalar loads of the headers
    cnt[u] = 0;
    if (t0 + u < tiles_per_frame) {
      const int *hdr = a.tilehdr + ((size_t)f * tiles_per_frame + t0 + u) * 2;
      cnt[u] = hdr[0] ? hdr[1] : 0;
    }
    total += cnt[u];
  }
  if (total == 0) return;
  const size_t N = (size_t)c.W * c.H;
  int *parent = a.parent + (size_t)f * N;
  for (int i0 = 0; i0 < total; i0 += 256) {            // block-uniform
    int i = i0 + tid, u = 0;
#pragma unroll
    for (int v = 0; v < kLinkTiles - 1; v++) if (u == v && i >= cnt[v]) { i -= cnt[v]; u = v + 1; }
    int ra = -1, rb = -1;
    if (i0 + tid < total) {
      const uint2 q = a.requests[((size_t)f * tiles_per_frame + t0 + u) * a.req_cap + i];
      if (MOD_CHECK(a, (size_t)q.x < N && (size_t)q.y < N, 0)) { ra = parent[q.x]; rb = (int)q.y; }
      if (!MOD_CHECK(a, ra < 0 || ((size_t)ra < N && rb >= 0), 1)) ra = -1;
    }
    const int pa = wave_prev_i32(ra), pb = wave_prev_i32(rb);
    if (ra >= 0 && !(lane > 0 && pa == ra && pb == rb)) uf_unite(parent, ra, rb);
  }
}

// Root-level flatten: every tile root finds its final root, remembers it (path compression, so pixels are two hops from
// their final root), and folds its partial record into the final root's record; final roots list themselves for k_select.
// A wave takes 64 / TH tiles at once: lane = (tile, row of the tile) reads that row's root bits and walks them — the few roots
// of a row one after the other, all rows and tiles of the wave side by side (each root is a chain of dependent accesses).
__device__ void select_frame(const DevCam &c, const ClArgs &a, ClusterInfo *tmp, int f, int tid);

template <int TH, bool FILTER>
__global__ __launch_bounds__(256) void k_ccl_merge(DevCam c, ClArgs a, int tiles_per_frame, ClusterInfo *tmp) {
  static_assert(64 % TH == 0, "a wave covers whole tiles");
  constexpr int TPW = 64 / TH;                         // tiles per wave
  const int f = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int t = (blockIdx.x * 4 + wv) * TPW + lane / TH, j = lane % TH;
  const size_t N = (size_t)c.W * c.H;
  int *parent = a.parent + (size_t)f * N;
  int *rsize = a.rsize + (size_t)f * N, *rkey = a.rkey + (size_t)f * N;
  const int wi = t % c.mask_words, ty = t / c.mask_words, y = ty * TH + j;
  unsigned long long bits = 0ull;
  if (t < tiles_per_frame && y < c.H && a.tilehdr[((size_t)f * tiles_per_frame + t) * 2] != 0)   // (0: nothing dynamic in the tile)
    bits = a.lroot[((size_t)f * c.H + y) * c.mask_words + wi];
  while (bits) {
    const int b = __ffsll(bits) - 1;
    bits &= bits - 1ull;
    const int p = y * c.W + wi * 64 + b;
    const int r = uf_find(parent, p);
    if (!MOD_CHECK(a, r >= 0 && (size_t)r < N, 2)) continue;
    if (r == p) {
      const int slot = atomicAdd(&a.counters[f * 8 + 0], 1);
      if (MOD_CHECK(a, slot >= 0 && (size_t)slot < N, 3)) st_agent(&a.rootlist[(size_t)f * N + slot], p);   // read by the frame's last workgroup
    } else {
      parent[p] = r;   // r is final: no union runs after k_ccl_link
      const int sz = rsize[p], ky = rkey[p];
      // the running sum the add returns is this tile root's place among the component's members: [old, old + sz) of the component's
      // member segment (the final root's own tile pixels hold [0, its count): rsize[r] starts there).  Parked in the tile root's key
      // entry, which nobody reads again as a key — k_final takes it from there instead of reserving slots with an atomic of its own.
      rkey[p] = atomicAdd(&rsize[r], sz);
      if (ky != kKeyNone) atomicMin(&rkey[r], ky);
    }
  }
  // FILTER (small batches): the size filter needs every record of the frame folded — the frame's LAST workgroup to get here runs it
  // (one dependent launch less per call, which is what a small batch is made of; in a large batch the count below — an atomic and
  // two barriers in each of 30 k workgroups, most of which have nothing else to do — costs more than the launch: 0.080 -> 0.101 ms
  // per 512 pairs, so large batches keep the filter as a kernel of its own, k_select).  What it reads of the other
  // workgroups' work went to memory past the XCD's L2 — device-scope atomics (counts, sizes, keys) and device-scope stores (the root
  // list) — and has been acknowledged when the writer's waves pass the barrier below (it drains their memory counters); the last
  // workgroup reads it with device-scope loads.  No fence: a device-scope release is a write-back of the XCD's whole L2, and one per
  // workgroup made this kernel 20x slower (0.017 -> 0.36 ms per 64 pairs, measured).
  if (!FILTER) return;
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(&a.counters[f * 8 + 2], 1) == (int)gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;                                 // block-uniform
  if (threadIdx.x == 0) a.counters[f * 8 + 2] = 0;     // (k_median_ties counts its workgroups in frame 0's slot)
  select_frame(c, a, tmp, f, (int)threadIdx.x);
}

// removeSmallClusters + the reference's numbering as a kernel of its own (large batches): one workgroup per frame
__global__ __launch_bounds__(256) void k_select(DevCam c, ClArgs a, ClusterInfo *tmp) { select_frame(c, a, tmp, (int)blockIdx.x, (int)threadIdx.x); }

// ---------------------------------------------------------------------------------------------------------------
// One workgroup of 256 threads per frame (the frame's last one in k_ccl_merge): size filter, ordering by first_edge_key, new labels,
// member-segment offsets, object shells.  What other workgroups of the same kernel wrote (root list, records) is read past the
// caches of this CU / XCD (ld_agent: device-scope loads).
__device__ void select_frame(const DevCam &c, const ClArgs &a, ClusterInfo *tmp, int f, int tid) {
  __shared__ int s_n;
  if (tid == 0) s_n = 0;
  __syncthreads();
  const size_t N = (size_t)c.W * c.H;
  const int nroots = ld_agent(&a.counters[f * 8 + 0]);
  int *rsize = a.rsize + (size_t)f * N, *rkey = a.rkey + (size_t)f * N;
  const int *roots = a.rootlist + (size_t)f * N;
  ClusterInfo *T = tmp + (size_t)f * a.max_objects;
  ClusterInfo *C = a.clusters + (size_t)f * a.max_objects;
  // removeSmallClusters: `cluster_size.at(i) < cluster_size_th_` drops the component (clusterer_nodelet.cpp:374);
  // a component without any edge never got a label in the reference (key == none)
  for (int i = tid; i < nroots; i += 256) {
    const int r = ld_agent(roots + i);
    if (!MOD_CHECK(a, r >= 0 && (size_t)r < N, 4)) continue;
    const int size = ld_agent(rsize + r), key = ld_agent(rkey + r);
    bool keep = (key != kKeyNone) && (size >= c.cluster_size);
    if (keep) {
      const int slot = atomicAdd(&s_n, 1);
      // capacity: mod_set_params admits a cluster_size only if max_width * max_height / cluster_size clusters fit max_objects,
      // so this cannot overflow (asserted in the checked build, code 15)
      if (MOD_CHECK(a, slot < a.max_objects, 15) && slot < a.max_objects) { T[slot].comp = r; T[slot].size = size; T[slot].offset = key; }
      else keep = false;
    }
    if (!keep) rkey[r] = -1;
  }
  __syncthreads();
  const int K = min(s_n, a.max_objects);
  // rank by key (distinct pixel indices) = the reference's increasing-root-id renumbering (:381)
  for (int s = tid; s < K; s += 256) {
    const int key = T[s].offset;
    int rank = 0;
    for (int t = 0; t < K; t++) rank += (T[t].offset < key) ? 1 : 0;
    ClusterInfo ci;
    ci.comp = T[s].comp; ci.size = T[s].size; ci.offset = 0; ci.med_pix = -1; ci.med_bits = 0; ci.ambiguous = 0;
    ci.pad[0] = ci.pad[1] = 0;
    C[rank] = ci;
    rkey[ci.comp] = rank;                               // k_final looks the new label up here
    ClusterBox bx;
    for (int d = 0; d < 8; d++) bx.w[d] = 0xffffffffu;
    a.cbox[(size_t)f * a.max_objects + rank] = bx;      // k_final folds the members' x, y, z into it
  }
  __syncthreads();
  if (tid == 0) {
    int off = 0;
    // the segment's start also goes to the final root's size entry (its size lives on in C[k].size): k_final's tile roots read
    // (new label, segment start) of their final root in one round trip
    for (int k = 0; k < K; k++) { C[k].offset = off; rsize[C[k].comp] = off; off += C[k].size; }
    a.counters[f * 8 + 1] = K;
    if (a.n_clusters) a.n_clusters[f] = K;
    // the launch's cluster list for k_median (order irrelevant): workgroups are then launched per cluster, not per frame
    const int base = K ? atomicAdd(&a.counters[6], K) : 0;
    for (int k = 0; k < K; k++) a.worklist[base + k] = (uint32_t)f * (uint32_t)a.max_objects + (uint32_t)k;
  }
  // object shells; bounding_box / center / velocity are filled by k_median once k_final has folded the members' coordinates
  ModObject *O = (ModObject *)a.objects + (size_t)f * a.max_objects;
  for (int k = tid; k < K; k += 256) {
    ModObject o;
    o.id = k; o.n_points = C[k].size;
    for (int d = 0; d < 3; d++) { o.bounding_box[d] = 0.0; o.center[d] = 0.0; o.velocity[d] = 0.0; }
    o.orientation[0] = 0.0; o.orientation[1] = 0.0; o.orientation[2] = 0.0; o.orientation[3] = 1.0;
    O[k] = o;
  }
}

// Final labels + member compaction, one workgroup per tile.  labels[p] = new label of p's component or -1 (the whole
// plane is written here, 4 B/px, when the caller asks for it: the reference renders its cluster image only for subscribers,
// clusterer_nodelet.cpp:235-236).  A pixel's parent entry names its tile root, so the few tile roots resolve their final
// root's new label once (into LDS) and every pixel just looks it up; members of surviving clusters are counted per tile
// root in LDS, one cursor atomic per (tile root) reserves their slots, then (||v|| bits, pixel) records are appended.
// XY_FROM_Z: the planes come from the fused scene-flow kernel of the same call (mod_process_dev), where every valid pixel has
// x = F32(ray_x(column) * (double)z), y = F32(ray_y(row) * (double)z) (sceneflow.hip sf_stage1, getPoint3D): the members' x, y are
// then recomputed from z and the ray tables — the same two operations, the same bits — instead of being read back (8 B/px of the
// active tiles less for this HBM-bound kernel).  Caller-supplied clouds (mod_cluster_dev) are read as they are.
#ifndef FINAL_PLAIN_LABELS   // streaming stores for the label plane: nobody on the GPU reads it (k_final 1.03 -> 0.97 ms per 512 pairs)
#define FINAL_ST(p, v) __builtin_nontemporal_store((int)(v), (p))
#else
#define FINAL_ST(p, v) (*(p) = (v))
#endif
constexpr int kFinalEarly = 32;   // dynamic pixels in a wave's rows from which its velocity / depth rows are fetched with the first round trip
template <int TH, int NW, bool XY_FROM_Z>
__global__ __launch_bounds__(NW * 64) void k_final(DevCam c, ClArgs a) {
  constexpr int RPW = TH / NW;
  __shared__ int nlmap[TH * 64];                     // per tile-root cell: new label of its component (or -1)
  __shared__ int lcount[TH * 64];                    // per tile-root cell: member count, then base slot of its members
  const int wi = blockIdx.x, f = blockIdx.z, lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
  const size_t tidx = (size_t)f * gridDim.y * gridDim.x + (size_t)blockIdx.y * gridDim.x + wi;
  const size_t N = (size_t)c.W * c.H;
  const size_t fN = (size_t)f * N;
  const int x0 = wi * 64, y0 = blockIdx.y * TH, r0 = w * RPW, x = x0 + lane;
  if (a.tilehdr[tidx * 2] == 0) {                    // nothing dynamic in the tile
    if (a.labels) {
#pragma unroll
      for (int j = 0; j < RPW; j++) { const int y = y0 + r0 + j; if (y < c.H && x < c.W) FINAL_ST(&a.labels[fN + (size_t)y * c.W + x], -1); }
    }
    return;
  }
  // ---- all row-independent HBM reads up front (clamped addresses, results of non-dynamic lanes are ignored) ----
  const int xc = min(x, c.W - 1);
  uint64_t mw[RPW], rw[RPW];
  int par[RPW];
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int y = y0 + r0 + j, yc = min(y, c.H - 1);
    const size_t wo = ((size_t)f * c.H + yc) * c.mask_words + wi;
    mw[j] = (y < c.H) ? a.mask[wo] : 0ull;
    rw[j] = (y < c.H) ? a.lroot[wo] : 0ull;
    par[j] = a.parent[fN + (size_t)yc * c.W + xc];
  }
  // ---- tile roots: their parent entry names the final root (k_ccl_merge), whose entries hold the new label (rkey, k_select) and the
  // start of the cluster's member segment (rsize, k_select); the tile root's own key entry holds its place inside that segment
  // (k_ccl_merge).  ONE round trip, three independent gathers; the cell's counter starts at the first slot of its members, so that
  // the LDS atomics below hand out absolute slots — no global cursor atomic, no second barrier (round 4: four dependent round
  // trips and three barriers per tile) ----
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    if ((rw[j] >> lane) & 1ull) {
      const int cell = (r0 + j) * 64 + lane;
      const int p = (y0 + r0 + j) * c.W + x;
      int lab = -1, first = 0;
      if (MOD_CHECK(a, par[j] >= 0 && (size_t)par[j] < N, 5)) {
        const int fr = par[j];
        lab = a.rkey[fN + fr];
        const int seg = a.rsize[fN + fr];
        const int own = a.rkey[fN + p];
        first = seg + (fr == p ? 0 : own);
      }
      nlmap[cell] = lab;
      lcount[cell] = first;
    }
  }
  // velocity (for the ||v|| bits) and depth (for the bounding box) of the wave's rows: tiles with this many dynamic pixels nearly
  // always hold members of a surviving cluster — their loads leave with the gathers above instead of after them
  float vx[RPW], vy[RPW], vz[RPW], px[RPW], py[RPW], pz[RPW];
  int ndyn = 0;
#pragma unroll
  for (int j = 0; j < RPW; j++) ndyn += __popcll((unsigned long long)mw[j]);
  const bool early = ndyn >= kFinalEarly;                            // wave-uniform
  auto load_members = [&]() {
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const size_t gp = fN + (size_t)min(y0 + r0 + j, c.H - 1) * c.W + xc;
      vx[j] = a.vx[gp]; vy[j] = a.vy[gp]; vz[j] = a.vz[gp];
      pz[j] = a.z[gp];
      if (!XY_FROM_Z) { px[j] = a.x[gp]; py[j] = a.y[gp]; }
    }
  };
  if (early) load_members();
  lds_barrier();
  // every wave of the tile has read the header: the tile stage's last reader leaves it zero for the next call (the scene-flow
  // epilogue / k_tile_flags only ever SET headers, so nobody has to clear 0.9 M of them per 512 pairs beforehand)
  if (w == 0 && lane == 0) a.tilehdr[tidx * 2] = 0;
  // ---- labels ----
  const float invW = 1.0f / (float)c.W;
  const int tile0 = y0 * c.W + x0;
  int nl[RPW], cell[RPW], slot[RPW];
  bool any_member = false;
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int y = y0 + r0 + j;
    const bool dyn = (mw[j] >> lane) & 1ull;
    // tile-local coordinates of the tile root: d = ly * W + lx with lx < 64, ly < TH, so (d + 0.5) / W truncates to ly exactly
    const int d = par[j] - tile0;
    const int ly = (int)(((float)d + 0.5f) * invW);
    // a tile root's own entry was redirected to the final root (possibly in another tile) by k_ccl_merge: it is its own cell
    const bool isroot = (rw[j] >> lane) & 1ull;
    int cl = !dyn ? 0 : isroot ? ((r0 + j) * 64 + lane) : (ly * 64 + (d - ly * c.W));
    if (!MOD_CHECK(a, cl >= 0 && cl < TH * 64 && (!dyn || isroot || (d >= 0 && d - ly * c.W < 64)), 6)) cl = 0;
    int l = dyn ? nlmap[cl] : -1;
    if (!MOD_CHECK(a, l >= -1 && l < a.max_objects, 7)) l = -1;
    nl[j] = l; cell[j] = cl; slot[j] = 0;
    if (a.labels && y < c.H && x < c.W) FINAL_ST(&a.labels[fN + (size_t)y * c.W + x], l);
    any_member = any_member || (__ballot(l >= 0) != 0);
  }
  if (any_member) {                                  // wave-uniform
    if (!early) load_members();
    // ---- members: slots handed out per tile root in LDS ----
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int l = nl[j], cl = cell[j];
      const uint64_t mb = __ballot(l >= 0);
      if (mb == 0) continue;                         // wave-uniform
      // lanes that share the first member's tile root reserve their slots with one LDS atomic; stragglers use their own
      const int lead = __ffsll((unsigned long long)mb) - 1;
      const int c0 = __builtin_amdgcn_readlane(cl, lead);
      const bool grp = l >= 0 && cl == c0;
      const uint64_t gb = __ballot(grp);
      int base = 0;
      if (lane == lead) base = atomicAdd(&lcount[c0], __popcll((unsigned long long)gb));
      base = __builtin_amdgcn_readlane(base, lead);
      if (grp) slot[j] = base + __popcll((unsigned long long)(gb & ((1ull << lane) - 1ull)));
      else if (l >= 0) slot[j] = atomicAdd(&lcount[cl], 1);
    }
    if (XY_FROM_Z) {
      const double rx = c.rayx[xc];
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const double zd = (double)pz[j];
        px[j] = (float)(rx * zd);
        py[j] = (float)(c.rayy[min(y0 + r0 + j, c.H - 1)] * zd);
      }
    }
    uint32_t nb[RPW];                                // ||v|| bit patterns (norms are >= 0 and never NaN: the patterns order like the values)
#pragma unroll
    for (int j = 0; j < RPW; j++) nb[j] = __float_as_uint(norm3_f32(vx[j], vy[j], vz[j]));
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      if (nl[j] >= 0) {
        if (MOD_CHECK(a, slot[j] >= 0 && (size_t)slot[j] < N, 8)) {
          a.mbits[fN + slot[j]] = nb[j];
          a.mpix[fN + slot[j]] = (uint32_t)((y0 + r0 + j) * c.W + x);
        }
      }
    }
    // bounding box of the cluster (pcl::getMinMax3D in cluster2MovingObject, clusterer_nodelet.cpp:151-152): the wave's members
    // are folded per cluster label with DPP reductions, one set of atomics per (wave, cluster) — nearly always one cluster
    int pend[RPW];
#pragma unroll
    for (int j = 0; j < RPW; j++) pend[j] = nl[j];
    for (;;) {
      int mine = -1;
#pragma unroll
      for (int j = 0; j < RPW; j++) mine = pend[j] >= 0 ? pend[j] : mine;
      const uint64_t b = __ballot(mine >= 0);
      if (b == 0) break;                               // wave-uniform
      const int L = __builtin_amdgcn_readlane(mine, __ffsll((unsigned long long)b) - 1);
      uint32_t mn0 = 0xffffffffu, mn1 = 0xffffffffu, mn2 = 0xffffffffu, mx0 = 0u, mx1 = 0u, mx2 = 0u, mn3 = 0xffffffffu, mx3 = 0u;
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        if (pend[j] == L) {
          const uint32_t ox = f2ord(px[j]), oy = f2ord(py[j]), oz = f2ord(pz[j]);
          mn0 = min(mn0, ox); mx0 = max(mx0, ox); mn1 = min(mn1, oy); mx1 = max(mx1, oy); mn2 = min(mn2, oz); mx2 = max(mx2, oz);
          mn3 = min(mn3, nb[j]); mx3 = max(mx3, nb[j]);
          pend[j] = -1;
        }
      }
      mn0 = wave_min_u32(mn0); mn1 = wave_min_u32(mn1); mn2 = wave_min_u32(mn2); mn3 = wave_min_u32(mn3);
      mx0 = wave_max_u32(mx0); mx1 = wave_max_u32(mx1); mx2 = wave_max_u32(mx2); mx3 = wave_max_u32(mx3);
      // ONE atomic instruction per (wave, cluster): lanes 0..7 each fold one word (an atomic instruction costs the CU's memory
      // pipeline the same whether one lane or eight are active); the maxima are kept complemented so that all eight are minima.
      // Words 6, 7 (round 5): smallest / largest ||v|| of the members — k_median starts its selection from that range instead of
      // scanning the members for it
      uint32_t v = ~mx3;
      v = lane == 0 ? mn0 : v; v = lane == 1 ? mn1 : v; v = lane == 2 ? mn2 : v; v = lane == 3 ? ~mx0 : v; v = lane == 4 ? ~mx1 : v;
      v = lane == 5 ? ~mx2 : v; v = lane == 6 ? mn3 : v;
      if (lane < 8) atomicMin(&a.cbox[(size_t)f * a.max_objects + L].w[lane], v);
    }
  }
}

// Median-velocity member per cluster: the element at position size/2 of the members sorted by ||v|| descending
// (clusterer_nodelet.cpp:168-174).  All norms are >= 0 and never NaN, so their F32 bit patterns order like the values:
// a range-adaptive 2048-bin histogram over (bits - min) >> shift isolates the bin that holds rank size/2, its members
// (up to kMedDirect of them) are ranked exactly in LDS; degenerate distributions narrow the range and repeat.
// Round 5 — a cluster is a chain of barriers and memory round trips, so what counts is how many clusters are in flight; rounds 1-4
// held the norms in registers (32 per thread: 125 VGPRs with spills, ONE workgroup per CU, 25 us per cluster).  Now:
//   * the norms are STREAMED from L2 in every pass (eight loads in flight per thread; a cluster is some 90 KB that its own workgroup
//     read a few microseconds before): under 32 VGPRs, two 1024-thread workgroups per CU;
//   * the members' norm range comes with the cluster's box record (k_final folds it with the same atomic instruction): no pass
//     over the members for it;
//   * the bin that holds the rank is found by all 16 waves (two bins per thread, one wave scan + 16 wave totals), not by wave 0
//     walking 2048 bins; the bins are zeroed again as they are read;
//   * a bin with up to kMedDirect members ends the rounds: they are compacted with their list positions, ranked directly, and the
//     tie-break (smallest column-major index, differing vectors flagged) runs on that list — two passes over the norms for most
//     clusters;
//   * the rare paths left the kernel: NaN coordinates of a caller's cloud are k_box_nan's, the tie replay k_median_ties'.
constexpr int kMedThreads = 1024, kMedBins = 2048, kMedDirect = 256, kMedCand = kMedBins / 2;

// box of a cluster record -> bounding_box / center of the object (cluster2MovingObject, clusterer_nodelet.cpp:151-161): F32 max - min
// and (min + max) / 2, widened to F64 (getMinMax3D starts from +-FLT_MAX: members at +inf leave the minimum there, members at -inf the maximum)
__device__ __forceinline__ void box_to_object(const ClusterBox &rec, ModObject *o) {
  for (int d = 0; d < 3; d++) {
    const float mn = fminf(ord2f(rec.w[d]), 3.402823466e38f), mx = fmaxf(ord2f(~rec.w[3 + d]), -3.402823466e38f);
    o->bounding_box[d] = (double)(mx - mn);
    o->center[d] = (double)((mn + mx) / 2.0f);
  }
}

// f(bits, list position) for every member: kMedFlight coalesced loads in flight per thread — unconditional, at clamped positions
// (a predicated load is an exec-masked region of its own: the loads would leave one by one), 0xffffffff (not a norm) past the end
constexpr int kMedFlight = 10;   // (12: the first spills at 64 VGPRs)
template <class F> __device__ __forceinline__ void med_stream(const uint32_t *sbits, int size, int tid, F f) {
  for (int i0 = tid; i0 < size; i0 += kMedThreads * kMedFlight) {
    uint32_t b[kMedFlight];
#pragma unroll
    for (int u = 0; u < kMedFlight; u++) b[u] = *(const uint32_t *)((const char *)sbits + 4u * (uint32_t)min(i0 + u * kMedThreads, size - 1));
#pragma unroll
    for (int u = 0; u < kMedFlight; u++) { const int i = i0 + u * kMedThreads; f(i < size ? b[u] : 0xffffffffu, i); }
  }
}

__global__ __launch_bounds__(kMedThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_median(DevCam c, ClArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t N = (size_t)c.W * c.H;
  const int nwork = a.counters[6];
  __shared__ uint32_t hist[kMedBins];                // the bins; between the rounds and the tie-break: candidates (bits | list position)
  __shared__ uint32_t s_wsum[kMedThreads / 64];
  __shared__ uint32_t s_bin, s_rem, s_inbin, s_cnt, s_val, s_ties;
  __shared__ unsigned long long s_best;
  __shared__ int s_amb, s_item;
#ifdef MOD_PHASE_COUNTERS
  const unsigned long long bt0 = wall_clock64();
  if (tid == 0) atomicMin(&a.dbg[42], bt0);
  struct BlockEnd { unsigned long long *d; unsigned long long t0; int tid; bool busy;
    __device__ ~BlockEnd() { const unsigned long long t1 = wall_clock64(); if (tid == 0) { atomicMax(&d[43], t1); if (busy) { atomicAdd(&d[40], t1 - t0); atomicAdd(&d[41], 1ull); } } } };
  BlockEnd be{a.dbg, bt0, tid, (int)blockIdx.x < nwork};
#endif
  hist[tid] = 0u; hist[tid + kMedThreads] = 0u;      // every phase leaves the bins zeroed again
  for (;;) {
    // clusters differ 10x in size: workgroups take the next one when they are free (counters[5]) instead of a fixed share
    if (tid == 0) { s_item = atomicAdd(&a.counters[5], 1); s_cnt = 0u; s_best = ~0ull; s_amb = 0; }
    __syncthreads();
    // (the work item is the same in every lane, but arrives through LDS in a vector register: as a scalar, everything derived from
    // it — frame, cluster record, the list's base addresses — stays on the scalar unit and the member loads are `base + lane offset`)
    const int wi = __builtin_amdgcn_readfirstlane(s_item);
    if (wi >= nwork) break;                          // block-uniform; every workgroup gets here
    const uint32_t item = a.worklist[wi];
    const int f = (int)(item / (uint32_t)a.max_objects), k = (int)(item % (uint32_t)a.max_objects);
    ClusterInfo *ci = a.clusters + (size_t)f * a.max_objects + k;
    const int ci_size = __builtin_amdgcn_readfirstlane(ci->size), ci_off = __builtin_amdgcn_readfirstlane(ci->offset);
    const int size = MOD_CHECK(a, ci_size >= 1 && ci_off >= 0 && (size_t)ci_off + (size_t)ci_size <= N, 9) ? ci_size : 0;
    if (size == 0) continue;                                        // (checked build only; block-uniform)
    const uint32_t *sbits = a.mbits + (size_t)f * N + ci_off;       // ||v|| bits of the members ...
    const uint32_t *spix = a.mpix + (size_t)f * N + ci_off;         // ... and their pixel indices
    const ClusterBox *box = a.cbox + (size_t)f * a.max_objects + k; // complete: k_final has finished
#ifdef MOD_PHASE_COUNTERS
    unsigned long long mt0 = wall_clock64(), mt1;
#define MSTAMP(i) { __syncthreads(); mt1 = wall_clock64(); if (tid == 0) atomicAdd(&a.dbg[i], mt1 - mt0); mt0 = mt1; }
#else
#define MSTAMP(i)
#endif
    uint32_t lo = box->w[6], hi = ~box->w[7], rem = (uint32_t)(size / 2);   // the members' norm range (k_final)
    if (!MOD_CHECK(a, lo <= hi, 9)) { lo = 0u; hi = 0x7f800000u; }
    bool exact = false;                      // the live range is a single value: every member of it ties
    uint32_t val = 0, nties = 0;
    MSTAMP(26)
    for (int round = 0; round < 8; round++) {        // one or two rounds in practice; the bound only guards against corrupt input
      const uint32_t range = hi - lo;
      const int shift = range < (uint32_t)kMedBins ? 0 : (32 - __clz((int)range) - 11);   // (range >> shift) < 2048
      // ---- histogram of the live range (the bins are zero: start of the kernel / the scan of the round before) ----
      med_stream(sbits, size, tid, [&](uint32_t b, int) { if (b >= lo && b <= hi) atomicAdd(&hist[(b - lo) >> shift], 1u); });
      __syncthreads();
      MSTAMP(30 + (round > 0 ? 2 : 0))
      // ---- bin that holds descending rank `rem`: thread t owns bins 2047 - 2t and 2046 - 2t (walking down from the top) ----
      const int b0 = kMedBins - 1 - 2 * tid;
      const uint32_t h0 = hist[b0], h1 = hist[b0 - 1];
      hist[b0] = 0u; hist[b0 - 1] = 0u;
      uint32_t incl = h0 + h1;                       // inclusive prefix over the wave's lanes
      for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
      if (lane == 63) s_wsum[wv] = incl;
      __syncthreads();
      uint32_t before = 0;                           // members in the waves above this one
      for (int w2 = 0; w2 < kMedThreads / 64; w2++) before += (w2 < wv) ? s_wsum[w2] : 0u;
      const uint32_t upto = before + incl, excl = upto - (h0 + h1);
      if (rem >= excl && rem < upto) {               // exactly one thread
        const bool first = rem - excl < h0;
        s_bin = (uint32_t)(first ? b0 : b0 - 1); s_rem = first ? rem - excl : rem - excl - h0; s_inbin = first ? h0 : h1;
      }
      __syncthreads();
      const uint32_t bin = s_bin, inbin = s_inbin;
      rem = s_rem;
      const uint32_t nlo = lo + (bin << shift);
      const uint32_t nhi = shift ? (nlo + ((1u << shift) - 1u)) : nlo;
      lo = nlo; hi = nhi < hi ? nhi : hi;
      MSTAMP(31 + (round > 0 ? 2 : 0))
      if (shift == 0) { exact = true; val = lo; nties = inbin; break; }
      if (inbin <= (uint32_t)kMedDirect) break;                     // rank the survivors directly
    }
    MSTAMP(27)
    const float *pvx = a.vx + (size_t)f * N, *pvy = a.vy + (size_t)f * N, *pvz = a.vz + (size_t)f * N;
    uint32_t best, bvx, bvy, bvz;
    // canonical pick among the members of norm `val`: the smallest column-major index (the member list is in tile order)
    auto offer = [&](uint32_t p) {
      const uint32_t px = p % (uint32_t)c.W, py = p / (uint32_t)c.W;
      atomicMin(&s_best, ((unsigned long long)(px * (uint32_t)c.H + py) << 32) | p);
    };
    auto settle = [&]() {                            // after a barrier: the pick and its vector
      best = (uint32_t)(s_best & 0xffffffffull);
      if (s_best == ~0ull) best = spix[0];           // unreachable for consistent input; keeps every access in bounds
      if (!MOD_CHECK(a, s_best != ~0ull && (size_t)best < N, 10)) best = 0;
      bvx = __float_as_uint(pvx[best]); bvy = __float_as_uint(pvy[best]); bvz = __float_as_uint(pvz[best]);
    };
    auto differs = [&](uint32_t p) {
      return p != best && (__float_as_uint(pvx[p]) != bvx || __float_as_uint(pvy[p]) != bvy || __float_as_uint(pvz[p]) != bvz);
    };
    if (!exact) {
      // ---- the (<= kMedDirect) members left in [lo, hi], compacted with their list positions into the (zeroed) bins: exact rank,
      // then the tie-break on the same list ----
      med_stream(sbits, size, tid, [&](uint32_t b, int i) {
        if (b >= lo && b <= hi) { const uint32_t s0 = atomicAdd(&s_cnt, 1u); if (s0 < (uint32_t)kMedCand) { hist[s0] = b; hist[kMedCand + s0] = (uint32_t)i; } }
      });
      __syncthreads();
      MSTAMP(34)
      const int fn = MOD_CHECK(a, s_cnt <= (uint32_t)kMedCand, 9) ? (int)s_cnt : kMedCand;
      uint32_t mine = 0, mypix = 0;
      if (tid < fn) {
        mine = hist[tid];
        uint32_t gt = 0, ge = 0;
        for (int j = 0; j < fn; j++) { const uint32_t o = hist[j]; gt += o > mine; ge += o >= mine; }
        if (gt <= rem && rem < ge) { s_val = mine; s_ties = ge - gt; }  // every member of that value writes the same
      }
      __syncthreads();
      val = s_val; nties = s_ties;
      const bool tied = tid < fn && mine == val;
      if (tied) { mypix = spix[hist[kMedCand + tid]]; offer(mypix); }
      __syncthreads();
      if (tid < fn) { hist[tid] = 0u; hist[kMedCand + tid] = 0u; }   // the bins are a histogram again
      settle();
      if (nties > 1u && tied && differs(mypix)) s_amb = 1;
    } else {
      // ---- a single value fills the bin (quantised inputs: possibly thousands of members): the tie-break streams the list ----
      med_stream(sbits, size, tid, [&](uint32_t b, int i) { if (b == val) offer(spix[i]); });
      __syncthreads();
      settle();
      if (nties > 1u) med_stream(sbits, size, tid, [&](uint32_t b, int i) { if (b == val && differs(spix[i])) s_amb = 1; });
    }
    __syncthreads();
    MSTAMP(28)
    MSTAMP(29)
    if (tid == 0) {
      ci->med_pix = (int)best; ci->med_bits = val; ci->ambiguous = s_amb;
      if (s_amb) a.tielist[atomicAdd(&a.counters[7], 1)] = item;
      ModObject *o = (ModObject *)a.objects + (size_t)f * a.max_objects + k;
      o->velocity[0] = (double)__uint_as_float(bvx); o->velocity[1] = (double)__uint_as_float(bvy); o->velocity[2] = (double)__uint_as_float(bvz);
      box_to_object(*box, o);
    }
    // (the next turn's first barrier orders tid 0's reads of s_amb / s_best with their re-initialisation)
    __syncthreads();
  }
}

// NaN coordinates among a cluster's members — only possible for caller-supplied clouds (mod_cluster_dev / mod_cluster_cloud_host: the
// scene-flow stage never marks a pixel dynamic without finite x, y, z), so the launcher starts this kernel for those calls only.
// pcl::getMinMax3D's dense path folds min_p = min_p.min(pt) in member order (column-major) with SSE semantics "(a < b) ? a : b": a NaN
// replaces the running value and the next point replaces the NaN, i.e. the result is the min / max over the members AFTER the
// last NaN, or NaN when the last member is NaN.  One workgroup walks the launch's clusters; those without a NaN cost three loads.
__global__ __launch_bounds__(kMedThreads) void k_box_nan(DevCam c, ClArgs a) {
  const int tid = threadIdx.x;
  const size_t N = (size_t)c.W * c.H;
  const int nwork = a.counters[6];
  __shared__ uint32_t s_last, s_mn, s_mx, s_n;
  for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint32_t item = a.worklist[wi];
    const int f = (int)(item / (uint32_t)a.max_objects), k = (int)(item % (uint32_t)a.max_objects);
    const ClusterInfo *ci = a.clusters + (size_t)f * a.max_objects + k;
    const ClusterBox rec = a.cbox[(size_t)f * a.max_objects + k];
    bool anynan = false;
    for (int d = 0; d < 3; d++) anynan = anynan || isnan(ord2f(rec.w[d])) || isnan(ord2f(~rec.w[3 + d]));
    if (!anynan) continue;                           // block-uniform
    const int size = ci->size;
    const uint32_t *spix = a.mpix + (size_t)f * N + ci->offset;
    const float *pl[3] = {a.x + (size_t)f * N, a.y + (size_t)f * N, a.z + (size_t)f * N};
    for (int d = 0; d < 3; d++) {
      __syncthreads();
      if (tid == 0) { s_last = 0u; s_mn = 0xffffffffu; s_mx = 0u; s_n = 0u; }   // last NaN key + 1, min, max, survivors
      __syncthreads();
      for (int i = tid; i < size; i += kMedThreads) {
        const uint32_t p = spix[i];
        if (isnan(pl[d][p])) atomicMax(&s_last, (p % (uint32_t)c.W) * (uint32_t)c.H + p / (uint32_t)c.W + 1u);
      }
      __syncthreads();
      const uint32_t lastnan = s_last;              // column-major key + 1 of the last NaN member, 0 if none
      for (int i = tid; i < size; i += kMedThreads) {
        const uint32_t p = spix[i];
        const uint32_t key1 = (p % (uint32_t)c.W) * (uint32_t)c.H + p / (uint32_t)c.W + 1u;
        if (key1 > lastnan) { const uint32_t o = f2ord(pl[d][p]); atomicMin(&s_mn, o); atomicMax(&s_mx, o); atomicAdd(&s_n, 1u); }
      }
      __syncthreads();
      if (tid == 0) {
        ModObject *o = (ModObject *)a.objects + (size_t)f * a.max_objects + k;
        const float nanv = __uint_as_float(0x7fc00000u);
        float mn = s_n ? ord2f(s_mn) : nanv, mx = s_n ? ord2f(s_mx) : nanv;
        if (lastnan == 0u) { mn = fminf(mn, 3.402823466e38f); mx = fmaxf(mx, -3.402823466e38f); }   // no NaN in this coordinate: the +-FLT_MAX start values hold
        o->bounding_box[d] = (double)(mx - mn);
        o->center[d] = (double)((mn + mx) / 2.0f);
      }
    }
  }
}

// Tied medians, exactly.  When k_median flagged a cluster (members tie on ||v|| with different vectors), the member that
// libstdc++'s std::sort leaves at size/2 is found by replaying introsort's moves on the sub-range that holds that position
// (introsort_emul.h): the members are laid out in the reference's initial order (column-major pixel order,
// clusterMap2IndicesCluster, clusterer_nodelet.cpp:97-117), each __unguarded_partition is done by the whole workgroup —
// its k-th swap exchanges the k-th element from the left that is not before the pivot with the k-th from the right that is
// not after it, so ranks from two prefix counts give every swap at once — and the finished (<= 16 element) range gets the
// stable insertion sort.  Rare path: one workgroup per flagged cluster, nothing to do for the others.
// Scratch (all dead by now): keys -> parent plane, pixels -> rsize plane, swap lists -> the member arrays.
constexpr int kTieThreads = 1024, kTieLds = 8192;   // (13 Ki elements = 156 KB, one partition level less in HBM: measured in round 5, no faster, and a
                                                     // workgroup that needs a whole CU's LDS waits for one beside the other chunk's kernels)

struct TieShared {           // control block of one workgroup of k_median_ties
  int cntA[kTieThreads / 64], cntB[kTieThreads / 64];
  int first, last, depth, m, totA, done;
  uint32_t answer, pivot;
#ifdef MOD_PHASE_COUNTERS
  unsigned long long sec[3][8];   // shader cycles per section of a partition step: [HBM arrays / LDS arrays, range > 2048 / LDS, range <= 2048][section]; [..][7] = steps
  int mode;
#endif
};
#ifdef MOD_PHASE_COUNTERS
#define TIE_SEC(i) { if (tid == 0) { const unsigned long long t_ = clock64(); sh.sec[grp][i] += t_ - tsec; tsec = t_; } }
#else
#define TIE_SEC(i)
#endif

// One __unguarded_partition_pivot step on [first, last) of (key, val), by the whole workgroup; updates sh.first / sh.last
// to the side that holds `want`.  Works on HBM or LDS arrays alike (KP is deduced per call site, so each instance keeps its
// address space).  All threads must call it; ends with a barrier.
template <class KP, class PP>
__device__ __forceinline__ void tie_partition_step(KP key, KP val, PP Apos, PP Bpos, int first, int last, int want, TieShared &sh,
                                                   int tid) {
  using namespace introsort_emul;
  const int lane = tid & 63, wv = tid >> 6;
  constexpr int NW = kTieThreads / 64;
  // elements a wave takes per loop trip, in units of 64: eight reads in flight on HBM arrays (a trip is a memory round trip); ONE on
  // LDS arrays — there a step is bound by the CU's vector issue (16 waves x the unrolled body: the position pass alone took 4-7 k
  // cycles whatever the range), and a wave whose slice is 128 elements must not execute the body of 512 (round 5, section clocks)
  constexpr int U = std::is_same<PP, uint16_t *>::value ? 1 : 8;
#ifdef MOD_PHASE_COUNTERS
  const int grp = sh.mode == 0 ? 0 : (last - first > 2048 ? 1 : 2);
  unsigned long long tsec = clock64();
  if (tid == 0) sh.sec[grp][7] += 1;
#endif
  if (tid == 0) {
    sh.depth--;
    // __move_median_to_first(first, first + 1, mid, last - 1): the three keys are fetched together (one memory round trip
    // instead of one per comparison), the comparison tree of introsort_emul::move_median_to_first picks the median's
    // position, and the pivot's key is handed to the workgroup through LDS
    const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
    const uint32_t ka = key[ia], kb = key[ib], kc = key[ic];
    int im;                                              // comp(x, y) = key[x] > key[y]
    if (ka > kb) im = (kb > kc) ? ib : (ka > kc) ? ic : ia;
    else im = (ka > kc) ? ia : (kb > kc) ? ic : ib;
    const uint32_t km = (im == ia) ? ka : (im == ib) ? kb : kc;
    const uint32_t kf = key[first], vf = val[first], vm = val[im];
    key[first] = km; val[first] = vm; key[im] = kf; val[im] = vf;
    sh.pivot = km;
  }
  __syncthreads();
  TIE_SEC(0)
  const uint32_t pk = sh.pivot;
  // A: elements NOT before the pivot (key <= pk), ranked from the left; B: elements NOT after it (key >= pk), ranked from the
  // right.  Wave w owns one contiguous slice and reads it 64 consecutive elements at a time (4 reads in flight).
  const int lo = first + 1, L = last - lo;
  const int slice = (((L + NW - 1) / NW) + 63) & ~63;          // a multiple of 64: a wave's 64-element reads never straddle slices
  const int w0 = min(lo + wv * slice, last), w1 = min(w0 + slice, last);
  int cntA = 0, cntB = 0;
  for (int j0 = w0; j0 < w1; j0 += 64 * U) {
    uint32_t k4[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int j = j0 + u * 64 + lane; k4[u] = (j < w1) ? key[j] : 0u; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool in = j0 + u * 64 + lane < w1;
      cntA += __popcll((unsigned long long)__ballot(in && k4[u] <= pk));
      cntB += __popcll((unsigned long long)__ballot(in && k4[u] >= pk));
    }
  }
  if (lane == 0) { sh.cntA[wv] = cntA; sh.cntB[wv] = cntB; }
  __syncthreads();
  TIE_SEC(1)
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < NW; t++) { const int x = sh.cntA[t]; sh.cntA[t] = run; run += x; }
    sh.totA = run; run = 0;
    for (int t = NW - 1; t >= 0; t--) { const int x = sh.cntB[t]; sh.cntB[t] = run; run += x; }
    sh.m = 0;
  }
  __syncthreads();
  TIE_SEC(2)
  {
    int runA = sh.cntA[wv], seenB = 0, mloc = 0;
    const int sufB = sh.cntB[wv];
    const uint64_t lt = (1ull << lane) - 1ull, le_m = lt | (1ull << lane);
    for (int j0 = w0; j0 < w1; j0 += 64 * U) {
      uint32_t k4[U];
#pragma unroll
      for (int u = 0; u < U; u++) { const int j = j0 + u * 64 + lane; k4[u] = (j < w1) ? key[j] : 0u; }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int j = j0 + u * 64 + lane;
        const bool in = j < w1, le = in && k4[u] <= pk, ge = in && k4[u] >= pk;
        const uint64_t bl = __ballot(le), bg = __ballot(ge);
        const int iA = runA + __popcll((unsigned long long)(bl & lt));
        // elements of B strictly right of j: later waves' + this wave's not yet seen, minus those up to and including this lane
        const int geR = sufB + (cntB - seenB - __popcll((unsigned long long)(bg & le_m)));
        // the iA-th stop of the left pointer swaps iff the iA-th stop of the right pointer lies right of it
        if (le) Apos[iA] = j;
        mloc += __popcll((unsigned long long)__ballot(le && geR >= iA + 1));   // (wave-uniform: one LDS atomic per wave below, not one per lane — 1024 adds on one address serialise)
        if (ge) Bpos[geR] = j;
        runA += __popcll((unsigned long long)bl);
        seenB += __popcll((unsigned long long)bg);
      }
    }
    if (mloc && lane == 0) atomicAdd(&sh.m, mloc);
  }
  __syncthreads();
  TIE_SEC(3)
  const int m = sh.m;
  for (int i0 = tid; i0 < m; i0 += kTieThreads * 2) {           // the m swaps, two per thread in flight
    const int i1 = i0 + kTieThreads;
    const int a0 = (int)Apos[i0], b0 = (int)Bpos[i0];
    const int a1 = (i1 < m) ? (int)Apos[i1] : a0, b1 = (i1 < m) ? (int)Bpos[i1] : b0;
    const uint32_t ka0 = key[a0], va0 = val[a0], kb0 = key[b0], vb0 = val[b0];
    const uint32_t ka1 = key[a1], va1 = val[a1], kb1 = key[b1], vb1 = val[b1];
    key[a0] = kb0; val[a0] = vb0; key[b0] = ka0; val[b0] = va0;
    if (i1 < m) { key[a1] = kb1; val[a1] = vb1; key[b1] = ka1; val[b1] = va1; }
  }
  __syncthreads();
  TIE_SEC(4)
  if (tid == 0) {
    // the left pointer's final stop: the next untouched element of A, unless the right pointer's last swap partner comes first
    const int am = (m < sh.totA) ? (int)Apos[m] : 0x7fffffff, bm = (m > 0) ? (int)Bpos[m - 1] : 0x7fffffff;
    const int cut = am < bm ? am : bm;
    if (want >= cut) sh.first = cut; else sh.last = cut;
  }
  __syncthreads();
  TIE_SEC(5)
}

// Runs partition steps until the live range is at most `stop` elements long (or the depth limit turns it into a heap sort).
template <class KP, class PP>
__device__ __forceinline__ void tie_narrow(KP key, KP val, PP Apos, PP Bpos, int want, int stop, TieShared &sh, int tid) {
  using namespace introsort_emul;
  while (true) {
    const int first = sh.first, last = sh.last;
    if (last - first <= stop || sh.done) break;
    if (sh.depth == 0) {                             // depth limit hit: __partial_sort(first, last, last) = heap sort
      __syncthreads();
      if (tid == 0) { const View v{&key[0], &val[0]}; heap_sort(v, first, last); sh.answer = val[want]; sh.done = 1; }
      __syncthreads();
      break;
    }
    __syncthreads();
    tie_partition_step(key, val, Apos, Bpos, first, last, want, sh, tid);
  }
}

// publishMovingObjects (clusterer_nodelet.cpp:324-343): ids run over ACCEPTED clusters only; a cluster is rejected when
// (double)||median v|| < dynamic_speed (:176) — unreachable for members that are all dynamic, kept for exactness.
__device__ __forceinline__ void finalize_frame(const DevCam &c, const ClArgs &a, int f) {
  const int K = a.counters[f * 8 + 1];                 // (k_ccl_merge's, an earlier kernel)
  ModObject *O = (ModObject *)a.objects + (size_t)f * a.max_objects;
  const ClusterInfo *C = a.clusters + (size_t)f * a.max_objects;
  int n = 0;
  for (int k = 0; k < K; k++) {
    const float nrm = __uint_as_float(C[k].med_bits);
    if ((double)nrm < c.speed_th_d) continue;
    if (n != k) O[n] = O[k];
    O[n].id = n;
    n++;
  }
  a.n_objects[f] = n;
}

__global__ __launch_bounds__(kTieThreads) void k_median_ties(DevCam c, ClArgs a, int frames) {
  using namespace introsort_emul;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t N = (size_t)c.W * c.H;
  const int nwork = a.counters[7];
  // 96 KB of LDS, used twice: while the members are laid out, as kTieLds 64-bit cell masks + kTieLds cell offsets; during the
  // LDS part of the sort, as keys, values and two 16-bit position lists (positions inside the resident range fit 16 bits)
  __shared__ uint64_t lraw[kTieLds + kTieLds / 2];
  uint32_t *lkey = reinterpret_cast<uint32_t *>(&lraw[0]), *lval = lkey + kTieLds;
  uint16_t *lA = reinterpret_cast<uint16_t *>(lval + kTieLds), *lB = lA + kTieLds;
  uint64_t *cmask = &lraw[0];
  uint32_t *cstart = reinterpret_cast<uint32_t *>(&lraw[kTieLds]);
  __shared__ int s_box[4];
  __shared__ TieShared sh;
  for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint32_t item = a.tielist[wi];
    const int f = (int)(item / (uint32_t)a.max_objects), k = (int)(item % (uint32_t)a.max_objects);
    const size_t fN = (size_t)f * N;
    ClusterInfo *ci = a.clusters + (size_t)f * a.max_objects + k;
    if (ci->ambiguous != 1) continue;                // block-uniform (always 1 for a listed cluster)
#ifdef MOD_PHASE_COUNTERS
    unsigned long long tt0 = wall_clock64(), tt1;
#define TSTAMP(i) { __syncthreads(); tt1 = wall_clock64(); if (tid == 0) atomicAdd(&a.dbg[i], tt1 - tt0); tt0 = tt1; }
#else
#define TSTAMP(i)
#endif
    const int size = ci->size, off = ci->offset;
    uint32_t *key = (uint32_t *)(a.parent + fN) + off;
    uint32_t *val = (uint32_t *)(a.rsize + fN) + off;
    uint32_t *Apos = a.mbits + fN + off, *Bpos = a.mpix + fN + off;
    // ---- image-space bounding box of the cluster (from its member list, before that list becomes scratch) ----
    if (tid == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0; s_box[2] = 0x7fffffff; s_box[3] = 0; }
    __syncthreads();
    {
      uint32_t x0 = 0xffffffffu, x1 = 0, y0 = 0xffffffffu, y1 = 0;
      const float invW = 1.0f / (float)c.W;          // p < 2^24: (p + 0.5) / W truncates to the row exactly (else: divide)
      const bool small_image = N < (1u << 24);
      for (int i0 = tid; i0 < size; i0 += kTieThreads * 4) {
        uint32_t p4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p4[u] = Bpos[min(i0 + u * kTieThreads, size - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t y = small_image ? (uint32_t)(((float)p4[u] + 0.5f) * invW) : p4[u] / (uint32_t)c.W, x = p4[u] - y * (uint32_t)c.W;
          x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1; y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
        }
      }
      x0 = wave_min_u32(x0); x1 = wave_max_u32(x1); y0 = wave_min_u32(y0); y1 = wave_max_u32(y1);
      if (lane == 0 && x0 != 0xffffffffu) { atomicMin(&s_box[0], (int)x0); atomicMax(&s_box[1], (int)x1); atomicMin(&s_box[2], (int)y0); atomicMax(&s_box[3], (int)y1); }
    }
    __syncthreads();
    TSTAMP(20)
    if (MOD_ABLATE(c, 1 << 20)) continue;
    const int xmin = s_box[0], ymin = s_box[2], ymax = s_box[3], ncols = s_box[1] - xmin + 1;
    // ---- members in column-major order (clusterMap2IndicesCluster's order, :97-117) ----
    // Fast path, straight from the member list (two coalesced passes, no image reads): the bounding box is cut into cells of
    // one column x 64 rows; pass A sets bit (row & 63) of a member's cell, a prefix over the cells in column-major order gives
    // every cell its first slot, pass B puts member (column, row) at slot = start[cell] + popcount(mask[cell] below its bit).
    const int nseg64 = (ymax - ymin + 64) / 64;
    {
      // Boxes with more than kTieLds cells (one tied cluster across most of a large image) go through the same two passes once
      // per run of kTieLds cells in column-major order; `base` carries the members of the runs before.  (Round 1 scanned the
      // labels plane for such boxes; the member list needs neither that plane nor the velocities again.)
      const int ncell = ncols * nseg64;              // < 2^21 + W: W * H < 2^27
      const float invWf = 1.0f / (float)c.W;
      auto cell_of = [&](uint32_t p, int &bit) {
        const uint32_t y = (N < (1u << 24)) ? (uint32_t)(((float)p + 0.5f) * invWf) : p / (uint32_t)c.W, x = p - y * (uint32_t)c.W;
        const int ry = (int)y - ymin;
        bit = ry & 63;
        return ((int)x - xmin) * nseg64 + (ry >> 6);
      };
      int base = 0;
      for (int g0 = 0; g0 < ncell; g0 += kTieLds) {  // block-uniform
        const int ncl = min(kTieLds, ncell - g0);
        for (int i = tid; i < kTieLds; i += kTieThreads) cmask[i] = 0ull;
        __syncthreads();
        for (int i0 = tid; i0 < size; i0 += kTieThreads * 8) {
          uint32_t p8[8];
#pragma unroll
          for (int u = 0; u < 8; u++) p8[u] = Bpos[min(i0 + u * kTieThreads, size - 1)];
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (i0 + u * kTieThreads < size) {
              int bit;
              const int cl = cell_of(p8[u], bit) - g0;
              if (cl >= 0 && cl < ncl) atomicOr((unsigned long long *)&cmask[cl], 1ull << bit);
            }
        }
        __syncthreads();
        TSTAMP(21)
        {   // exclusive prefix of the cell populations; thread t owns cells CPT t .. CPT t + CPT - 1
          constexpr int CPT = kTieLds / kTieThreads;
          static_assert(CPT * kTieThreads == kTieLds, "every thread owns the same number of cells");
          int cnt[CPT], tot = 0;
#pragma unroll
          for (int u = 0; u < CPT; u++) { const int ci2 = tid * CPT + u; cnt[u] = ci2 < ncl ? __popcll((unsigned long long)cmask[ci2]) : 0; tot += cnt[u]; }
          int incl = tot;
          for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
          if (lane == 63) sh.cntA[wv] = incl;
          __syncthreads();
          if (tid == 0) { int run = base; for (int t = 0; t < kTieThreads / 64; t++) { const int x = sh.cntA[t]; sh.cntA[t] = run; run += x; } sh.cntB[0] = run; }
          __syncthreads();
          int run = sh.cntA[wv] + incl - tot;
#pragma unroll
          for (int u = 0; u < CPT; u++) { cstart[tid * CPT + u] = (uint32_t)run; run += cnt[u]; }
          base = sh.cntB[0];
        }
        __syncthreads();
        for (int i0 = tid; i0 < size; i0 += kTieThreads * 8) {
          uint32_t p8[8], k8[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { const int i = min(i0 + u * kTieThreads, size - 1); p8[u] = Bpos[i]; k8[u] = Apos[i]; }
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (i0 + u * kTieThreads < size) {
              int bit;
              const int cl = cell_of(p8[u], bit) - g0;
              if (cl >= 0 && cl < ncl) {
                const int slot = (int)cstart[cl] + __popcll((unsigned long long)(cmask[cl] & ((1ull << bit) - 1ull)));
                if (MOD_CHECK(a, slot >= 0 && slot < size, 11)) {
                  key[slot] = k8[u];                    // ||v|| bits as k_final computed them (norm3_f32)
                  val[slot] = p8[u];
                }
              }
            }
        }
        __syncthreads();
        TSTAMP(22)
      }
      (void)MOD_CHECK(a, base == size, 14);          // every member lies in exactly one cell
    }
    TSTAMP(23)
    if (MOD_ABLATE(c, 1 << 22)) continue;
    if (tid == 0) { sh.first = 0; sh.last = size; sh.depth = 2 * floor_log2(size); sh.done = 0; }
    __syncthreads();
    // ---- introsort, only along the range that holds position size/2: in HBM while the range is long, then in LDS ----
    const int want = size / 2;
#ifdef MOD_PHASE_COUNTERS
    if (tid == 0) { sh.mode = 0; for (int g = 0; g < 3; g++) for (int i = 0; i < 8; i++) sh.sec[g][i] = 0; }
    __syncthreads();
#endif
    tie_narrow(key, val, Apos, Bpos, want, kTieLds, sh, tid);
    TSTAMP(24)
    if (MOD_ABLATE(c, 1 << 23)) continue;
    __syncthreads();
    if (!sh.done) {
      const int first = sh.first, len = sh.last - first;
      for (int i = tid; i < len; i += kTieThreads) { lkey[i] = key[first + i]; lval[i] = val[first + i]; }
      __syncthreads();
      if (tid == 0) { sh.first = 0; sh.last = len; }
#ifdef MOD_PHASE_COUNTERS
      if (tid == 0) sh.mode = 1;
#endif
      __syncthreads();
      tie_narrow(lkey, lval, lA, lB, want - first, 16, sh, tid);
      __syncthreads();
      if (!sh.done && tid == 0) {
        const View v{lkey, lval};
        insertion_sort(v, sh.first, sh.last);
        sh.answer = lval[want - first];
      }
      __syncthreads();
    }
    TSTAMP(25)
#ifdef MOD_PHASE_COUNTERS
    if (tid == 0) for (int g = 0; g < 3; g++) for (int i = 0; i < 8; i++) atomicAdd(&a.dbg[(g == 0 ? 0 : g == 1 ? 8 : 44) + i], sh.sec[g][i]);   // (slots 0-15 are k_ccl_tile's, unused with the bit-plane tile stage at n = 4)
#endif
    if (tid == 0) {
      const uint32_t best = sh.answer;
      ci->med_pix = (int)best; ci->ambiguous = 2;    // 2 = tie resolved by replaying the reference's sort
      ModObject *o = (ModObject *)a.objects + (size_t)f * a.max_objects + k;
      o->velocity[0] = (double)a.vx[fN + best]; o->velocity[1] = (double)a.vy[fN + best]; o->velocity[2] = (double)a.vz[fN + best];
      __threadfence();                               // device-wide before this workgroup's count below (rare path: once per tied cluster)
    }
    __syncthreads();
  }
  // ---- the launch's LAST workgroup to get here closes the call (round 5; k_finalize was a kernel of its own until then):
  // publishMovingObjects for every frame, and the counters go back to zero — every call finds them as mod_create left them,
  // no memset in front of the tile stage (mod_sf.hip: scratch_clean) ----
  __shared__ int s_last;
  __syncthreads();
  if (tid == 0) s_last = (atomicAdd(&a.counters[2], 1) == (int)gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;                                 // block-uniform
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the velocities other workgroups resolved (one invalidate per launch)
  for (int f = tid; f < frames; f += kTieThreads) finalize_frame(c, a, f);
  __syncthreads();                                     // (frame 0's thread has read its count before the words go)
  for (int i = tid; i < frames * 8; i += kTieThreads) a.counters[i] = 0;
}

constexpr int kFusedFilterFrames = 16;   // batches up to this size run the size filter inside k_ccl_merge (see there)
constexpr int kTileH = 16, kTileWaves = 4;   // tile = 64 x 16 px, 4 rows per wave (measured best of 8x4, 16x4, 16x8, 32x8)

}  // namespace

static dim3 tile_grid(const DevCam &c, int frames) { return dim3(c.mask_words, (c.H + kTileH - 1) / kTileH, frames); }

void launch_ccl_tile(const DevCam &c, const ClArgs &a, int frames, hipStream_t s) {
  const dim3 block(64, kTileWaves, 1), tgrid = tile_grid(c, frames);
  const int tx = (int)tgrid.x, tyn = (int)tgrid.y;
  if (c.n >= 1 && c.n <= 10) {
    // the reference's parameter range (Clusterer.cfg:11): bit-plane kernel first (one wave per tile, one instance per window size),
    // then the union-find kernel over the tiles it listed — resident workgroups (8 per CU) that pull tiles with an atomic cursor
    const dim3 bgrid((tgrid.x + bits::kTilesPerBlock - 1) / bits::kTilesPerBlock, tgrid.y, tgrid.z), bblock(64, bits::kTilesPerBlock, 1);
    switch (c.n) {
#define MOD_BITS_CASE(n) case n: hipLaunchKernelGGL(bits::k_ccl_bits<n>, bgrid, bblock, 0, s, c, a, tx, tyn); break;
      MOD_BITS_CASE(1) MOD_BITS_CASE(2) MOD_BITS_CASE(3) MOD_BITS_CASE(4) MOD_BITS_CASE(5)
      MOD_BITS_CASE(6) MOD_BITS_CASE(7) MOD_BITS_CASE(8) MOD_BITS_CASE(9) MOD_BITS_CASE(10)
#undef MOD_BITS_CASE
    }
    // (2048 workgroups = the 8 per CU that fit: 512 / 1024 / 4096 / 8192 measured slower or equal)
    const unsigned total = tgrid.x * tgrid.y * tgrid.z;
    const dim3 lgrid(std::min(2048u, total));
    if (c.n == 4) hipLaunchKernelGGL((k_ccl_tile_list<kTileH, 4, kTileWaves, true>), lgrid, block, 0, s, c, a, tx, tyn);
    else if (c.n < 4) hipLaunchKernelGGL((k_ccl_tile_list<kTileH, 4, kTileWaves, false>), lgrid, block, 0, s, c, a, tx, tyn);
    else if (c.n <= 8) hipLaunchKernelGGL((k_ccl_tile_list<kTileH, 8, kTileWaves, false>), lgrid, block, 0, s, c, a, tx, tyn);
    else hipLaunchKernelGGL((k_ccl_tile_list<kTileH, 16, kTileWaves, false>), lgrid, block, 0, s, c, a, tx, tyn);
    return;
  }
  const dim3 ggrid((tgrid.x + CCL_TPB - 1) / CCL_TPB, tgrid.y, tgrid.z);
  if (c.n < 4) hipLaunchKernelGGL((k_ccl_tile<kTileH, 4, kTileWaves, false>), ggrid, block, 0, s, c, a, tx, tyn);
  else if (c.n <= 8) hipLaunchKernelGGL((k_ccl_tile<kTileH, 8, kTileWaves, false>), ggrid, block, 0, s, c, a, tx, tyn);
  else hipLaunchKernelGGL((k_ccl_tile<kTileH, 16, kTileWaves, false>), ggrid, block, 0, s, c, a, tx, tyn);
}
void launch_tile_flags(const DevCam &c, const ClArgs &a, int frames, hipStream_t s) {
  const dim3 g = tile_grid(c, frames);
  const int tiles = (int)(g.x * g.y), total = tiles * frames;
  hipLaunchKernelGGL(k_tile_flags<kTileH>, dim3((total + 255) / 256), dim3(256), 0, s, c, a, (int)g.x, tiles, total);
}
void launch_ccl_link(const DevCam &c, const ClArgs &a, int frames, hipStream_t s) {
  const dim3 g = tile_grid(c, frames);
  const int tiles = (int)(g.x * g.y), per_block = kLinkTiles;
  hipLaunchKernelGGL(k_ccl_link, dim3((tiles + per_block - 1) / per_block, frames), dim3(256), 0, s, c, a, tiles);
}
void launch_ccl_merge(const DevCam &c, const ClArgs &a, int frames, ClusterInfo *rank_scratch, hipStream_t s) {
  const dim3 g = tile_grid(c, frames);
  const int tiles = (int)(g.x * g.y), per_block = 4 * (64 / kTileH);
  if (frames <= kFusedFilterFrames) hipLaunchKernelGGL((k_ccl_merge<kTileH, true>), dim3((tiles + per_block - 1) / per_block, frames), dim3(256), 0, s, c, a, tiles, rank_scratch);
  else {
    hipLaunchKernelGGL((k_ccl_merge<kTileH, false>), dim3((tiles + per_block - 1) / per_block, frames), dim3(256), 0, s, c, a, tiles, rank_scratch);
    hipLaunchKernelGGL(k_select, dim3(frames), dim3(256), 0, s, c, a, rank_scratch);
  }
}
void launch_final(const DevCam &c, const ClArgs &a, int frames, hipStream_t s) {
  // two waves per tile (8 rows each): 1 / 2 / 4 / 8 waves measured 1.419 (176 VGPRs: 2 waves per SIMD) / 0.965 / 1.001 / 1.460 ms per 512
  // pairs (in-process A/B)
  constexpr int kFinalWaves = 2;
  if (a.xy_from_z) hipLaunchKernelGGL((k_final<kTileH, kFinalWaves, true>), tile_grid(c, frames), dim3(64, kFinalWaves, 1), 0, s, c, a);
  else hipLaunchKernelGGL((k_final<kTileH, kFinalWaves, false>), tile_grid(c, frames), dim3(64, kFinalWaves, 1), 0, s, c, a);
}
void launch_median(const DevCam &c, const ClArgs &a, int frames, hipStream_t s) {
  // one 1024-thread workgroup fills a CU and costs ~80 ns of wave dispatch whether it finds work or not: launch at most one
  // per CU (fewer for small batches) and let each walk the launch's cluster list (k_select) / tie list (k_median)
  // (round 5: two 64-VGPR workgroups fit a CU — 512 of them)
  hipLaunchKernelGGL(k_median, dim3(std::min(512, frames * 8)), dim3(kMedThreads), 0, s, c, a);
  if (!a.xy_from_z) hipLaunchKernelGGL(k_box_nan, dim3(std::min(64, frames * 2)), dim3(kMedThreads), 0, s, c, a);   // a caller's cloud may hold NaN coordinates
  hipLaunchKernelGGL(k_median_ties, dim3(std::min(64, frames * 2)), dim3(kTieThreads), 0, s, c, a, frames);   // + publishMovingObjects in its last workgroup
}

int ccl_tile_rows() { return kTileH; }
int ccl_request_capacity(int n) { return n * (64 + n) + kTileH * n; }
