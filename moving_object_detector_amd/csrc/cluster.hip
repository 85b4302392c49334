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
// So the GPU runs a lock-free union-find (hook larger root under smaller with atomicMin), then rebuilds the
// reference's numbering from per-component statistics.
#include "mod_launch.h"
#include "../../include/mod_sf.h"

#pragma clang fp contract(off)

namespace {

// ---------------------------------------------------------------------------------------------------------------
// union-find on a per-frame parent plane; parent[i] <= i always, roots satisfy parent[r] == r.
// Plain (possibly L1-stale) loads are safe here: every value ever stored in parent[a] is a member of a's set and
// smaller than a, so a stale chain still ends in the same set; the hook itself is a device-scope atomicMin whose
// return value tells whether the node was still a root.
__device__ __forceinline__ int uf_find(int *__restrict__ parent, int a) {
  int p = parent[a];
  while (p != a) { a = p; p = parent[a]; }
  return a;
}

__device__ __forceinline__ void uf_unite(int *__restrict__ parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }      // hook the larger root a under the smaller b
    const int old = atomicMin(&parent[a], b);
    if (old == a) return;                         // a was still a root: done
    a = old;                                      // somebody hooked a meanwhile: keep uniting its new parent with b
  }
}

// bit `xx` of a mask row given the word holding the wave's own 64 columns (w0, index wi) and the word to its left (wm)
__device__ __forceinline__ bool row_bit(uint64_t w0, uint64_t wm, int wi, int xx) {
  if (xx < 0) return false;
  const uint64_t w = ((xx >> 6) == wi) ? w0 : wm;
  return (w >> (xx & 63)) & 1ull;
}

// parent[p] = p for dynamic pixels; other entries are never read.
__global__ __launch_bounds__(256) void k_ccl_init(DevCam c, ClArgs a) {
  const int lane = threadIdx.x, x = blockIdx.x * 64 + lane, y = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
  if (y >= c.H) return;
  const uint64_t w = a.mask[((size_t)f * c.H + y) * c.mask_words + blockIdx.x];
  if ((w >> lane) & 1ull) {
    const int p = y * c.W + x;
    a.parent[(size_t)f * c.W * c.H + p] = p;
  }
}

// Windowed union: thread = pixel, wave = 64 consecutive pixels of a row (= one mask word).
__global__ __launch_bounds__(256) void k_ccl_union(DevCam c, ClArgs a) {
  const int lane = threadIdx.x, wi = blockIdx.x, x = wi * 64 + lane, y = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
  if (y >= c.H) return;
  const int MW = c.mask_words, n = c.n;
  const size_t N = (size_t)c.W * c.H;
  const uint64_t *mf = a.mask + (size_t)f * c.H * MW;
  const size_t wofs = ((size_t)f * c.H + y) * MW + wi;
  const uint64_t mword = mf[(size_t)y * MW + wi];
  // does any row of the wave's neighbourhood hold a dynamic pixel?  If the own word is empty nothing can start here.
  if (mword == 0) {
    if (lane == 0) { a.edge_up[wofs] = 0; a.edge_any[wofs] = 0; }
    return;
  }
  const bool dyn = (mword >> lane) & 1ull;
  const float *zf = a.z + (size_t)f * N;
  int *parent = a.parent + (size_t)f * N;
  const int p = y * c.W + x;
  const float zp = dyn ? zf[p] : 0.0f;
  bool up = false;
  for (int dv = -n; dv <= 0; dv++) {
    const int yy = y + dv;
    if (yy < 0) continue;
    const uint64_t w0 = mf[(size_t)yy * MW + wi];
    const uint64_t wm = (wi > 0) ? mf[(size_t)yy * MW + wi - 1] : 0ull;
    if ((w0 | wm) == 0) continue;                                     // wave-uniform early out for empty rows
    for (int du = -n; du <= 0; du++) {
      if (dv == 0 && du == 0) continue;
      const int xx = x + du;
      if (dyn && row_bit(w0, wm, wi, xx)) {
        const int q = yy * c.W + xx;
        const float zq = zf[q];
        if (!(fabsf(zp - zq) > c.depth_th)) {                          // depthDiff gate; NaN links (clusterer_nodelet.cpp:194)
          up = true;
          uf_unite(parent, p, q);
        }
      }
    }
  }
  // pixels without an up-left edge may still be the up-left end of someone else's edge: look down-right, stop at first hit
  bool any = up;
  if (dyn && !up) {
    for (int dv = 0; dv <= n && !any; dv++) {
      const int yy = y + dv;
      if (yy >= c.H) break;
      const uint64_t w0 = mf[(size_t)yy * MW + wi];
      const uint64_t wp = (wi + 1 < MW) ? mf[(size_t)yy * MW + wi + 1] : 0ull;
      for (int du = 0; du <= n && !any; du++) {
        if (dv == 0 && du == 0) continue;
        const int xx = x + du;
        if (xx >= c.W) break;
        const uint64_t w = ((xx >> 6) == wi) ? w0 : wp;
        if ((w >> (xx & 63)) & 1ull) {
          const float zq = zf[yy * c.W + xx];
          if (!(fabsf(zq - zp) > c.depth_th)) any = true;
        }
      }
    }
  }
  const uint64_t bu = __ballot(up), ba = __ballot(any);
  if (lane == 0) { a.edge_up[wofs] = bu; a.edge_any[wofs] = ba; }
}

// Flatten: labels[p] = root index for members of labelled components, -1 otherwise; every root takes a dense component
// id, initialises its record and stores the code -(id+2) in its own labels entry.
__global__ __launch_bounds__(256) void k_ccl_flatten(DevCam c, ClArgs a) {
  const int lane = threadIdx.x, wi = blockIdx.x, x = wi * 64 + lane, y = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
  if (y >= c.H || x >= c.W) return;
  const size_t N = (size_t)c.W * c.H;
  const uint64_t any = a.edge_any[((size_t)f * c.H + y) * c.mask_words + wi];
  const int p = y * c.W + x;
  int out = -1;
  if ((any >> lane) & 1ull) {
    int *parent = a.parent + (size_t)f * N;
    const int r = uf_find(parent, p);
    out = r;
    if (r == p) {
      const int id = atomicAdd(&a.counters[f * 8 + 0], 1);
      if (id < a.comp_cap) {
        CompRec rec;
        rec.size = 0; rec.key = 0x7fffffff;
        rec.mn[0] = rec.mn[1] = rec.mn[2] = 0xffffffffu;
        rec.mx[0] = rec.mx[1] = rec.mx[2] = 0u;
        a.comps[(size_t)f * a.comp_cap + id] = rec;
        out = -(id + 2);
      } else {
        a.counters[f * 8 + 3] = 1;   // cannot happen: comp_cap >= N/2 >= number of components with an edge
        out = -1;
      }
    }
  }
  a.labels[(size_t)f * N + p] = out;
}

// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(v, o); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(v, o); v = t > v ? t : v; }
  return v;
}

// Per-component statistics with one set of atomics per (wave, component): size, first_edge_key, bbox (pcl::getMinMax3D
// dense path, clusterer_nodelet.cpp:151-152).  Also rewrites labels[p] to the component code -(id+2) for every member.
__global__ __launch_bounds__(256) void k_comp_stats(DevCam c, ClArgs a) {
  const int lane = threadIdx.x, wi = blockIdx.x, x = wi * 64 + lane, y = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
  if (y >= c.H) return;
  const size_t N = (size_t)c.W * c.H;
  const size_t wofs = ((size_t)f * c.H + y) * c.mask_words + wi;
  const uint64_t any = a.edge_any[wofs];
  if (any == 0) return;
  const uint64_t upw = a.edge_up[wofs];
  int *lab = a.labels + (size_t)f * N;
  const int p = y * c.W + x;
  const bool member = (any >> lane) & 1ull;
  int id = -1;
  uint32_t ox = 0, oy = 0, oz = 0;
  if (member) {
    int code = lab[p];
    if (code >= 0) { code = lab[code]; lab[p] = code; }   // the root's own entry already holds the code
    id = -code - 2;
    if (code == -1) id = -1;                                // overflowed root (never happens)
    ox = f2ord(a.x[(size_t)f * N + p]);
    oy = f2ord(a.y[(size_t)f * N + p]);
    oz = f2ord(a.z[(size_t)f * N + p]);
  }
  uint64_t todo = __ballot(member && id >= 0);
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int lid = __shfl(id, leader);
    const bool mine = member && id == lid;
    const uint64_t grp = __ballot(mine);
    const uint32_t key = wave_min_u32((mine && ((upw >> lane) & 1ull)) ? (uint32_t)p : 0x7fffffffu);
    const uint32_t mnx = wave_min_u32(mine ? ox : 0xffffffffu), mxx = wave_max_u32(mine ? ox : 0u);
    const uint32_t mny = wave_min_u32(mine ? oy : 0xffffffffu), mxy = wave_max_u32(mine ? oy : 0u);
    const uint32_t mnz = wave_min_u32(mine ? oz : 0xffffffffu), mxz = wave_max_u32(mine ? oz : 0u);
    if (lane == leader) {
      CompRec *r = a.comps + (size_t)f * a.comp_cap + lid;
      atomicAdd(&r->size, __popcll((unsigned long long)grp));
      if (key != 0x7fffffffu) atomicMin(&r->key, (int)key);
      atomicMin(&r->mn[0], mnx); atomicMax(&r->mx[0], mxx);
      atomicMin(&r->mn[1], mny); atomicMax(&r->mx[1], mxy);
      atomicMin(&r->mn[2], mnz); atomicMax(&r->mx[2], mxz);
    }
    todo &= ~grp;
  }
}

// One block per frame: size filter, ordering by first_edge_key, new labels, member-segment offsets, bbox/centre.
__global__ __launch_bounds__(256) void k_select(DevCam c, ClArgs a, ClusterInfo *tmp) {
  const int f = blockIdx.x, tid = threadIdx.x;
  __shared__ int s_n;
  if (tid == 0) s_n = 0;
  __syncthreads();
  const int nc = min(a.counters[f * 8 + 0], a.comp_cap);
  CompRec *comps = a.comps + (size_t)f * a.comp_cap;
  ClusterInfo *T = tmp + (size_t)f * a.max_objects;
  ClusterInfo *C = a.clusters + (size_t)f * a.max_objects;
  // removeSmallClusters: `cluster_size.at(i) < cluster_size_th_` drops the component (clusterer_nodelet.cpp:374)
  for (int i = tid; i < nc; i += 256) {
    const int size = comps[i].size, key = comps[i].key;
    bool keep = (key != 0x7fffffff) && (size >= c.cluster_size);
    if (keep) {
      const int slot = atomicAdd(&s_n, 1);
      if (slot < a.max_objects) { T[slot].comp = i; T[slot].size = size; T[slot].offset = key; }
      else { a.counters[f * 8 + 3] = 2; keep = false; }   // more clusters than max_objects: extra ones are dropped (flagged)
    }
    if (!keep) comps[i].key = -1;
  }
  __syncthreads();
  const int K = min(s_n, a.max_objects);
  // rank by key (keys are distinct pixel indices) = the reference's increasing-root-id renumbering (:381)
  for (int s = tid; s < K; s += 256) {
    const int key = T[s].offset;
    int rank = 0;
    for (int t = 0; t < K; t++) rank += (T[t].offset < key) ? 1 : 0;
    ClusterInfo ci;
    ci.comp = T[s].comp; ci.size = T[s].size; ci.offset = 0; ci.med_pix = -1; ci.med_bits = 0; ci.ambiguous = 0;
    ci.pad[0] = ci.pad[1] = 0;
    C[rank] = ci;
    comps[ci.comp].key = rank;
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
    const CompRec r = comps[C[k].comp];
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
  const uint64_t any = a.edge_any[((size_t)f * c.H + y) * c.mask_words + wi];
  if (any == 0) return;                                   // labels already hold -1 there (k_ccl_flatten)
  int *lab = a.labels + (size_t)f * N;
  const int p = y * c.W + x;
  const bool member = (any >> lane) & 1ull;
  int nl = -1;
  uint32_t nb = 0;
  if (member) {
    const int code = lab[p];
    if (code <= -2) nl = a.comps[(size_t)f * a.comp_cap + (-code - 2)].key;
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

}  // namespace

void launch_ccl(const DevCam &c, const ClArgs &a, int frames, hipStream_t s) {
  dim3 block(64, 4, 1), grid(c.mask_words, (c.H + 3) / 4, frames);
  hipLaunchKernelGGL(k_ccl_init, grid, block, 0, s, c, a);
  hipLaunchKernelGGL(k_ccl_union, grid, block, 0, s, c, a);
  hipLaunchKernelGGL(k_ccl_flatten, grid, block, 0, s, c, a);
}

void launch_objects(const DevCam &c, const ClArgs &a, int frames, hipStream_t s) {
  dim3 block(64, 4, 1), grid(c.mask_words, (c.H + 3) / 4, frames);
  hipLaunchKernelGGL(k_comp_stats, grid, block, 0, s, c, a);
  // the second ClusterInfo array (rank scratch) lives right behind the first one
  ClusterInfo *tmp = a.clusters + (size_t)frames * a.max_objects;
  hipLaunchKernelGGL(k_select, dim3(frames), dim3(256), 0, s, c, a, tmp);
  hipLaunchKernelGGL(k_relabel, grid, block, 0, s, c, a);
  hipLaunchKernelGGL(k_median, dim3(16, frames), dim3(256), 0, s, c, a);
  hipLaunchKernelGGL(k_finalize, dim3((frames + 63) / 64), dim3(64), 0, s, c, a, frames);
}
