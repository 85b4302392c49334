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
// GPU formulation
//   k_ccl_tile    one workgroup per 64 x TH tile: masked depth + parents live in LDS; rows are pre-linked into runs
//                 with wave ballots (no atomics), the remaining window edges are united with LDS atomicMin hooks; the
//                 tile's components leave as parent pointers + one partial statistics record per tile-local root.
//   k_ccl_border  only the edges that cross a tile border, united in HBM (device-scope atomicMin).
//   k_ccl_flatten root per pixel; tile-local records are folded into their global root's record; roots are listed.
//   k_select      size filter + ordering by first_edge_key (the reference's numbering) + bbox/centre.
//   k_relabel     final labels + per-cluster member segments; k_median: radix select of the median-||v|| member.
#include "mod_launch.h"
#include "../../include/mod_sf.h"

#pragma clang fp contract(off)

namespace {

constexpr int kKeyNone = 0x7fffffff;

// ---------------------------------------------------------------------------------------------------------------
// Union-find helpers.  parent[i] <= i always; roots satisfy parent[r] == r; a larger root is hooked under a smaller
// one with atomicMin, whose return value tells whether the node was still a root.  Reads are relaxed atomic loads so
// the compiler re-reads memory; values that are stale in a CU's L1 are harmless: every value ever stored in parent[a]
// is a member of a's set and smaller than a, so a stale chain still ends inside the same set.
__device__ __forceinline__ int ld_relaxed(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

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

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(v, o); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(v, o); v = t > v ? t : v; }
  return v;
}

// Adds the members of one wave (grouped by the record they belong to) into the statistics records with one set of
// atomics per (wave, record).  `rec_idx` < 0 marks a lane without contribution.
__device__ __forceinline__ void wave_accumulate(CompRec *recs, int rec_idx, uint32_t key, uint32_t ox, uint32_t oy, uint32_t oz,
                                                int lane) {
  uint64_t todo = __ballot(rec_idx >= 0);
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int lid = __shfl(rec_idx, leader);
    const bool mine = rec_idx == lid;
    const uint64_t grp = __ballot(mine);
    const uint32_t k = wave_min_u32(mine ? key : (uint32_t)kKeyNone);
    const uint32_t mnx = wave_min_u32(mine ? ox : 0xffffffffu), mxx = wave_max_u32(mine ? ox : 0u);
    const uint32_t mny = wave_min_u32(mine ? oy : 0xffffffffu), mxy = wave_max_u32(mine ? oy : 0u);
    const uint32_t mnz = wave_min_u32(mine ? oz : 0xffffffffu), mxz = wave_max_u32(mine ? oz : 0u);
    if (lane == leader) {
      CompRec *r = recs + lid;
      atomicAdd(&r->size, __popcll((unsigned long long)grp));
      if (k != (uint32_t)kKeyNone) atomicMin(&r->key, (int)k);
      atomicMin(&r->mn[0], mnx); atomicMax(&r->mx[0], mxx);
      atomicMin(&r->mn[1], mny); atomicMax(&r->mx[1], mxy);
      atomicMin(&r->mn[2], mnz); atomicMax(&r->mx[2], mxz);
    }
    todo &= ~grp;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Tile-local connected components.  Workgroup = 4 waves = one 64 x TH tile; wave w owns rows [w*TH/4, (w+1)*TH/4).
template <int TH>
__global__ __launch_bounds__(256) void k_ccl_tile(DevCam c, ClArgs a) {
  constexpr int RPW = TH / 4;                       // rows per wave
  __shared__ float zt[TH * 64];
  __shared__ int Lt[TH * 64];
  __shared__ uint64_t mrow[TH];
  __shared__ int s_any;
  const int lane = threadIdx.x, w = threadIdx.y, tid = w * 64 + lane;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * TH, f = blockIdx.z;
  const int MW = c.mask_words, n = c.n;
  const size_t N = (size_t)c.W * c.H;
  const size_t fN = (size_t)f * N;
  if (tid == 0) s_any = 0;
  __syncthreads();
  if (tid < TH) {
    const int gy = y0 + tid;
    const uint64_t m = (gy < c.H) ? a.mask[((size_t)f * c.H + gy) * MW + blockIdx.x] : 0ull;
    mrow[tid] = m;
    if (m) s_any = 1;
  }
  __syncthreads();
  const int r0 = w * RPW;
  if (!s_any) {                                     // nothing dynamic in the tile: only the root bits need clearing
    if (lane == 0)
      for (int j = 0; j < RPW; j++) if (y0 + r0 + j < c.H) a.lroot[((size_t)f * c.H + y0 + r0 + j) * MW + blockIdx.x] = 0ull;
    return;
  }
  const float th = c.depth_th;
  float zr[RPW];
  bool upr[RPW];
  // ---- phase A: masked depth into LDS, horizontal runs by ballot --------------------------------------------
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int rr = r0 + j;
    const uint64_t mw = mrow[rr];
    const bool dyn = (mw >> lane) & 1ull;
    const float z = dyn ? a.z[fN + (size_t)(y0 + rr) * c.W + x0 + lane] : 0.0f;
    zr[j] = z;
    zt[rr * 64 + lane] = z;
    const float zl = __shfl_up(z, 1);
    const bool cl = dyn && lane > 0 && ((mw >> (lane - 1)) & 1ull) && !(fabsf(z - zl) > th);   // linked to the left neighbour
    const uint64_t C = __ballot(cl);
    const uint64_t starts = mw & ~C;                // run starts: dynamic and not linked to the left
    if (dyn) {
      const int s = 63 - __clzll((long long)(starts & (~0ull >> (63 - lane))));
      Lt[rr * 64 + lane] = rr * 64 + s;
    }
    upr[j] = cl;
  }
  __syncthreads();
  // ---- phase B: the rest of the up-left window, in-tile edges only ------------------------------------------
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int rr = r0 + j;
    const bool dyn = (mrow[rr] >> lane) & 1ull;
    if (!dyn) continue;
    const float zp = zr[j];
    const int me = rr * 64 + lane;
    int cur = Lt[me];
    bool up = upr[j];
    for (int dv = -n; dv <= 0; dv++) {
      const int qr = rr + dv;
      if (qr < 0) continue;
      const uint64_t mq = mrow[qr];
      for (int du = -n; du <= 0; du++) {
        const int qc = lane + du;
        if (qc < 0 || (dv == 0 && du >= -1)) continue;      // (0,0) is p itself, (0,-1) is the run link
        if (!((mq >> qc) & 1ull)) continue;
        const int qi = qr * 64 + qc;
        if (fabsf(zp - zt[qi]) > th) continue;               // depthDiff gate (clusterer_nodelet.cpp:194); NaN links
        up = true;
        const int rq = ld_relaxed(&Lt[qi]);
        if (rq != cur) cur = uf_unite(Lt, cur, rq);
      }
    }
    upr[j] = up;
  }
  __syncthreads();
  // ---- phase C: flatten inside the tile, publish parents, root bits and empty records ------------------------
  int rootg[RPW];
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int rr = r0 + j, gy = y0 + rr;
    const bool dyn = (mrow[rr] >> lane) & 1ull;
    int rg = -1;
    bool isroot = false;
    if (dyn) {
      const int me = rr * 64 + lane;
      const int r = uf_find(Lt, me);
      rg = (y0 + (r >> 6)) * c.W + x0 + (r & 63);
      a.parent[fN + (size_t)gy * c.W + x0 + lane] = rg;
      isroot = (r == me);
      if (isroot) {
        CompRec rec;
        rec.size = 0; rec.key = kKeyNone;
        rec.mn[0] = rec.mn[1] = rec.mn[2] = 0xffffffffu;
        rec.mx[0] = rec.mx[1] = rec.mx[2] = 0u;
        a.comps[fN + rg] = rec;
      }
    }
    rootg[j] = rg;
    const uint64_t rb = __ballot(isroot);
    if (lane == 0 && gy < c.H) a.lroot[((size_t)f * c.H + gy) * MW + blockIdx.x] = rb;
  }
  __syncthreads();   // record initialisation has reached L2 before any wave's atomics on it
  // ---- phase D: partial statistics of the tile's components -----------------------------------------------------
#pragma unroll
  for (int j = 0; j < RPW; j++) {
    const int rr = r0 + j, gy = y0 + rr;
    if (mrow[rr] == 0) continue;                     // wave-uniform
    const int rg = rootg[j];
    uint32_t ox = 0, oy = 0, oz = 0, key = (uint32_t)kKeyNone;
    if (rg >= 0) {
      const size_t gp = (size_t)gy * c.W + x0 + lane;
      ox = f2ord(a.x[fN + gp]); oy = f2ord(a.y[fN + gp]); oz = f2ord(zr[j]);
      if (upr[j]) key = (uint32_t)gp;
    }
    wave_accumulate(a.comps + fN, rg, key, ox, oy, oz, lane);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Edges that leave the tile (to the tile on the left, above, or above-left).  Thread = pixel, wave = one mask word.
template <int TH>
__global__ __launch_bounds__(256) void k_ccl_border(DevCam c, ClArgs a) {
  const int lane = threadIdx.x, wi = blockIdx.x, x = wi * 64 + lane, y = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
  if (y >= c.H) return;
  const int MW = c.mask_words, n = c.n;
  const uint64_t *mf = a.mask + (size_t)f * c.H * MW;
  const uint64_t mword = mf[(size_t)y * MW + wi];
  if (mword == 0) return;
  const int tr = y % TH;                             // row inside the tile band
  const bool dyn = (mword >> lane) & 1ull;
  const bool work = dyn && (tr < n || lane < n);
  if (__ballot(work) == 0) return;
  if (!work) return;
  const size_t N = (size_t)c.W * c.H;
  const float *zf = a.z + (size_t)f * N;
  int *parent = a.parent + (size_t)f * N;
  const int p = y * c.W + x;
  const float zp = zf[p];
  int cur = ld_relaxed(parent + p);
  bool up = false;
  for (int dv = -n; dv <= 0; dv++) {
    const int yy = y + dv;
    if (yy < 0) continue;
    const bool above = dv < -tr;                     // row yy lies in the band above
    const uint64_t w0 = mf[(size_t)yy * MW + wi];
    const uint64_t wm = (wi > 0) ? mf[(size_t)yy * MW + wi - 1] : 0ull;
    for (int du = -n; du <= 0; du++) {
      if (dv == 0 && du == 0) continue;
      const int qc = lane + du;                      // column relative to the tile
      if (!(above || qc < 0)) continue;              // in-tile edge: k_ccl_tile did it
      const int xx = x + du;
      if (xx < 0) continue;
      const uint64_t wq = (qc >= 0) ? w0 : wm;
      if (!((wq >> (xx & 63)) & 1ull)) continue;
      const int q = yy * c.W + xx;
      if (fabsf(zp - zf[q]) > c.depth_th) continue;
      up = true;
      const int rq = ld_relaxed(parent + q);
      if (rq != cur) cur = uf_unite(parent, cur, rq);
    }
  }
  if (up) {
    // p has an up-left edge: candidate for first_edge_key.  parent[p] always names a tile-local root, whose record is
    // folded into the final root's record by k_ccl_flatten.
    atomicMin(&a.comps[(size_t)f * N + ld_relaxed(parent + p)].key, p);
  }
}

// Root per pixel (labels plane holds the root's pixel index, -1 for non-dynamic pixels); tile-local records are
// folded into their root's record; roots append themselves to the frame's root list.
__global__ __launch_bounds__(256) void k_ccl_flatten(DevCam c, ClArgs a) {
  const int lane = threadIdx.x, wi = blockIdx.x, x = wi * 64 + lane, y = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
  if (y >= c.H || x >= c.W) return;
  const size_t N = (size_t)c.W * c.H;
  const size_t wofs = ((size_t)f * c.H + y) * c.mask_words + wi;
  const uint64_t mword = a.mask[wofs];
  const int p = y * c.W + x;
  int out = -1;
  if ((mword >> lane) & 1ull) {
    const int *parent = a.parent + (size_t)f * N;
    const int r = uf_find(parent, p);
    out = r;
    if ((a.lroot[wofs] >> lane) & 1ull) {
      CompRec *recs = a.comps + (size_t)f * N;
      if (r == p) {
        const int slot = atomicAdd(&a.counters[f * 8 + 0], 1);
        a.rootlist[(size_t)f * N + slot] = p;
      } else {
        const CompRec mine = recs[p];
        CompRec *t = recs + r;
        atomicAdd(&t->size, mine.size);
        if (mine.key != kKeyNone) atomicMin(&t->key, mine.key);
        atomicMin(&t->mn[0], mine.mn[0]); atomicMax(&t->mx[0], mine.mx[0]);
        atomicMin(&t->mn[1], mine.mn[1]); atomicMax(&t->mx[1], mine.mx[1]);
        atomicMin(&t->mn[2], mine.mn[2]); atomicMax(&t->mx[2], mine.mx[2]);
      }
    }
  }
  a.labels[(size_t)f * N + p] = out;
}

// ---------------------------------------------------------------------------------------------------------------
// One block per frame: size filter, ordering by first_edge_key, new labels, member-segment offsets, bbox/centre.
__global__ __launch_bounds__(256) void k_select(DevCam c, ClArgs a, ClusterInfo *tmp) {
  const int f = blockIdx.x, tid = threadIdx.x;
  __shared__ int s_n;
  if (tid == 0) s_n = 0;
  __syncthreads();
  const size_t N = (size_t)c.W * c.H;
  const int nroots = a.counters[f * 8 + 0];
  CompRec *recs = a.comps + (size_t)f * N;
  const int *roots = a.rootlist + (size_t)f * N;
  ClusterInfo *T = tmp + (size_t)f * a.max_objects;
  ClusterInfo *C = a.clusters + (size_t)f * a.max_objects;
  // removeSmallClusters: `cluster_size.at(i) < cluster_size_th_` drops the component (clusterer_nodelet.cpp:374);
  // a component without any edge never got a label in the reference (key == none)
  for (int i = tid; i < nroots; i += 256) {
    const int r = roots[i];
    const int size = recs[r].size, key = recs[r].key;
    bool keep = (key != kKeyNone) && (size >= c.cluster_size);
    if (keep) {
      const int slot = atomicAdd(&s_n, 1);
      if (slot < a.max_objects) { T[slot].comp = r; T[slot].size = size; T[slot].offset = key; }
      else { a.counters[f * 8 + 3] = 2; keep = false; }   // more clusters than max_objects: the excess is dropped (flagged)
    }
    if (!keep) recs[r].key = -1;
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
    recs[ci.comp].key = rank;
    a.cursors[(size_t)f * a.max_objects + rank] = 0;
  }
  __syncthreads();
  if (tid == 0) {
    int off = 0;
    for (int k = 0; k < K; k++) { C[k].offset = off; off += C[k].size; }
    a.counters[f * 8 + 1] = K;
    if (a.n_clusters) a.n_clusters[f] = K;
  }
  // bbox / centre (cluster2MovingObject, clusterer_nodelet.cpp:151-161): F32 max-min and (min+max)/2, widened to F64
  ModObject *O = (ModObject *)a.objects + (size_t)f * a.max_objects;
  for (int k = tid; k < K; k += 256) {
    const CompRec r = recs[C[k].comp];
    ModObject o;
    o.id = k; o.n_points = r.size;
    for (int d = 0; d < 3; d++) {
      const float mn = ord2f(r.mn[d]), mx = ord2f(r.mx[d]);
      o.bounding_box[d] = (double)(mx - mn);
      o.center[d] = (double)((mn + mx) / 2.0f);
      o.velocity[d] = 0.0;
    }
    o.orientation[0] = 0.0; o.orientation[1] = 0.0; o.orientation[2] = 0.0; o.orientation[3] = 1.0;
    O[k] = o;
  }
}

// Final labels + member compaction: labels[p] = new label (or -1); members of surviving clusters append
// (||v|| bits, pixel) to their cluster's segment, one cursor atomic per (wave, cluster).
__global__ __launch_bounds__(256) void k_relabel(DevCam c, ClArgs a) {
  const int lane = threadIdx.x, wi = blockIdx.x, x = wi * 64 + lane, y = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
  if (y >= c.H) return;
  const size_t N = (size_t)c.W * c.H;
  const uint64_t mword = a.mask[((size_t)f * c.H + y) * c.mask_words + wi];
  if (mword == 0) return;                                 // labels already hold -1 there (k_ccl_flatten)
  int *lab = a.labels + (size_t)f * N;
  const int p = y * c.W + x;
  const bool dyn = (mword >> lane) & 1ull;
  int nl = -1;
  uint32_t nb = 0;
  if (dyn) {
    const int r = lab[p];
    nl = a.comps[(size_t)f * N + r].key;                  // new label of the root's component, or -1
    lab[p] = nl;
    if (nl >= 0) nb = __float_as_uint(norm3_f32(a.vx[(size_t)f * N + p], a.vy[(size_t)f * N + p], a.vz[(size_t)f * N + p]));
  }
  uint64_t todo = __ballot(nl >= 0);
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int l = __shfl(nl, leader);
    const bool mine = (nl == l);
    const uint64_t grp = __ballot(mine);
    int base = 0;
    if (lane == leader) base = atomicAdd(&a.cursors[(size_t)f * a.max_objects + l], __popcll((unsigned long long)grp));
    base = __shfl(base, leader);
    if (mine) {
      const int rank = __popcll((unsigned long long)(grp & ((1ull << lane) - 1ull)));
      const int off = a.clusters[(size_t)f * a.max_objects + l].offset;
      a.members[(size_t)f * N + off + base + rank] = make_uint2(nb, (uint32_t)p);
    }
    todo &= ~grp;
  }
}

// Median-velocity member per cluster: the element at position size/2 of the members sorted by ||v|| descending
// (clusterer_nodelet.cpp:168-174) == radix select on the F32 bits (all norms are finite and >= 0).
__global__ __launch_bounds__(256) void k_median(DevCam c, ClArgs a) {
  const int f = blockIdx.y, tid = threadIdx.x;
  const size_t N = (size_t)c.W * c.H;
  const int K = a.counters[f * 8 + 1];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_prefix, s_rem;
  __shared__ unsigned long long s_best;
  __shared__ int s_amb;
  for (int k = blockIdx.x; k < K; k += gridDim.x) {
    ClusterInfo *ci = a.clusters + (size_t)f * a.max_objects + k;
    const int size = ci->size;
    const uint2 *seg = a.members + (size_t)f * N + ci->offset;
    uint32_t prefix = 0, pmask = 0, rem = (uint32_t)(size / 2);
    for (int pass = 0; pass < 4; pass++) {
      const int shift = 24 - 8 * pass;
      hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < size; i += 256) {
        const uint32_t b = seg[i].x;
        if ((b & pmask) == prefix) atomicAdd(&hist[(b >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t r = rem;
        int bin = 255;
        for (; bin > 0; bin--) { if (r < hist[bin]) break; r -= hist[bin]; }
        s_prefix = prefix | ((uint32_t)bin << shift);
        s_rem = r;
      }
      __syncthreads();
      prefix = s_prefix; rem = s_rem; pmask |= 255u << shift;
      __syncthreads();
    }
    // candidates: members whose norm equals the selected value; canonical pick = smallest column-major index
    if (tid == 0) { s_best = ~0ull; s_amb = 0; }
    __syncthreads();
    for (int i = tid; i < size; i += 256) {
      const uint2 m = seg[i];
      if (m.x == prefix) {
        const uint32_t px = m.y % (uint32_t)c.W, py = m.y / (uint32_t)c.W;
        atomicMin(&s_best, ((unsigned long long)(px * (uint32_t)c.H + py) << 32) | m.y);
      }
    }
    __syncthreads();
    const uint32_t best = (uint32_t)(s_best & 0xffffffffull);
    const float bvx = a.vx[(size_t)f * N + best], bvy = a.vy[(size_t)f * N + best], bvz = a.vz[(size_t)f * N + best];
    for (int i = tid; i < size; i += 256) {
      const uint2 m = seg[i];
      if (m.x == prefix && m.y != best) {
        const size_t q = (size_t)f * N + m.y;
        if (__float_as_uint(a.vx[q]) != __float_as_uint(bvx) || __float_as_uint(a.vy[q]) != __float_as_uint(bvy) ||
            __float_as_uint(a.vz[q]) != __float_as_uint(bvz)) s_amb = 1;
      }
    }
    __syncthreads();
    if (tid == 0) {
      ci->med_pix = (int)best; ci->med_bits = prefix; ci->ambiguous = s_amb;
      ModObject *o = (ModObject *)a.objects + (size_t)f * a.max_objects + k;
      o->velocity[0] = (double)bvx; o->velocity[1] = (double)bvy; o->velocity[2] = (double)bvz;
    }
    __syncthreads();
  }
}

// publishMovingObjects (clusterer_nodelet.cpp:324-343): ids run over ACCEPTED clusters only; a cluster is rejected when
// (double)||median v|| < dynamic_speed (:176) — unreachable for members that are all dynamic, kept for exactness.
__global__ void k_finalize(DevCam c, ClArgs a, int frames) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= frames) return;
  const int K = a.counters[f * 8 + 1];
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
  a.counters[f * 8 + 2] = n;
  a.n_objects[f] = n;
}

constexpr int kTileH = 16;

}  // namespace

void launch_ccl(const DevCam &c, const ClArgs &a, int frames, hipStream_t s) {
  dim3 block(64, 4, 1);
  dim3 tgrid(c.mask_words, (c.H + kTileH - 1) / kTileH, frames);
  hipLaunchKernelGGL(k_ccl_tile<kTileH>, tgrid, block, 0, s, c, a);
  dim3 grid(c.mask_words, (c.H + 3) / 4, frames);
  hipLaunchKernelGGL(k_ccl_border<kTileH>, grid, block, 0, s, c, a);
  hipLaunchKernelGGL(k_ccl_flatten, grid, block, 0, s, c, a);
}

void launch_objects(const DevCam &c, const ClArgs &a, int frames, hipStream_t s) {
  dim3 block(64, 4, 1), grid(c.mask_words, (c.H + 3) / 4, frames);
  // the second ClusterInfo array (rank scratch) lives right behind the first one
  ClusterInfo *tmp = a.clusters + (size_t)frames * a.max_objects;
  hipLaunchKernelGGL(k_select, dim3(frames), dim3(256), 0, s, c, a, tmp);
  hipLaunchKernelGGL(k_relabel, grid, block, 0, s, c, a);
  hipLaunchKernelGGL(k_median, dim3(16, frames), dim3(256), 0, s, c, a);
  hipLaunchKernelGGL(k_finalize, dim3((frames + 63) / 64), dim3(64), 0, s, c, a, frames);
}
