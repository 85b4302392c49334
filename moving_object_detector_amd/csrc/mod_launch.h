// mod_launch.h — host-callable launchers of the gfx950 kernels (internal to libmod_sf.so).
#pragma once
#include "mod_device.h"

struct SfArgs {
  const float *dnow, *dprev, *flow;   // [F][H][W], [F][H][W], [F][H][W][2]
  float *x, *y, *z, *vx, *vy, *vz;    // [F][H][W]
  uint64_t *mask;                     // [F][H][mask_words] or null
  float4 *aos;                        // [F][H][W][2 x float4] or null
  float *depth;                       // [F][H][W] or null
  float *sflow;                       // [F][H][W][2] or null
  const FrameConst *fc;               // [F] device
  int32_t *tilehdr;                   // [F][tiles][2] or null: header word 0 of every cluster tile with a dynamic pixel is set to 1
  int32_t tile_rows, tiles_x, tiles_per_frame;   // (zeroed by the caller beforehand); tile = 64 x tile_rows pixels
  float2 *zrange;                     // [F][H][mask_words] or null: (smallest, largest) depth z of the DYNAMIC pixels of every non-zero mask
                                      // word, written with the word (k_ccl_bits decides one-class tiles from it without loading a depth row)
  unsigned long long *dbg;            // the context's diagnostic counters (checked build: index assertion 16), unused by a product build
};

// Scratch the clustering kernels work in; all device pointers, sized by the context.
struct ClArgs {
  const float *x, *y, *z, *vx, *vy, *vz;  // input planes [F][H][W]
  const uint64_t *mask;       // [F][H][mask_words] dynamic bits
  const float2 *zrange;       // [F][H][mask_words] depth range of the dynamic pixels of every NON-ZERO mask word (the fused scene-flow
                              // kernel's epilogue writes it with the word; entries of zero words are undefined), or null: a caller's cloud
  uint64_t *lroot;            // [F][H][mask_words] pixel is the root of a tile-local component (owns a partial record)
  int32_t *parent;            // [F][N] union-find parents (only dynamic entries are ever touched)
  int32_t *rootlist;          // [F][N] pixel indices of the final roots (aliases `mpix`, dead before k_final)
  int32_t *labels;            // [F][N] output plane; used as the root/code plane in between
  int32_t *rsize;             // [F][N] member count of the component rooted at this pixel (sparse: only root entries are touched); after
                              // k_select, at the final root of a surviving cluster: start of the cluster's member segment
  int32_t *rkey;              // [F][N] its first_edge_key; after k_ccl_merge, at a tile root that is not final: its members' place inside the
                              // component's member segment; after k_select, at a final root: the new label, or -1
  ClusterBox *cbox;           // [F][max_objects] bounding boxes of the surviving clusters (k_select init, k_final atomics)
  int32_t *counters;          // [F][8]: 0 n_comps, 1 n_clusters, 2 workgroups of k_ccl_merge done with the frame (of frame 0, later: of
                              // k_median_ties done); of frame 0 also (counts of the whole launch): 3 cursor into `tilelist`, 4 its
                              // length, 5 k_median's cursor into `worklist`, 6 its length, 7 length of `tielist`.  ALL ZERO between
                              // calls: k_median_ties' last workgroup clears them
  ClusterInfo *clusters;      // [F][max_objects]
  uint32_t *mbits;            // [F][N] ||v|| bit patterns of the members, grouped per cluster (SoA with mpix)
  uint32_t *mpix;             // [F][N] pixel index of each member
  uint32_t *worklist;         // [F * max_objects] frame * max_objects + cluster of every cluster of the launch; count in counters[6]
  uint32_t *tielist;          // [F * max_objects] the clusters k_median flagged ambiguous;                count in counters[7]
  void *objects;              // [F][max_objects] ModObject
  int32_t *n_objects;         // [F]
  int32_t *n_clusters;        // [F] or null
  int32_t max_objects;
  int32_t xy_from_z;          // the planes are the fused scene-flow kernel's of this call: x, y of a valid pixel are functions of z
  uint2 *requests;            // [F][tiles][req_cap] cross-tile link requests (halo pixel, tile root)
  int32_t *tilehdr;           // [F][tiles][2]: 0 no dynamic pixel / 1 has one (set together with the mask words) / 2 done by k_ccl_bits;
                              // number of requests.  Word 0 is ZERO between calls: k_final, its last reader, clears it
  uint32_t *tilelist;         // [F * tiles] tiles that k_ccl_bits left to the union-find kernel (frame * tiles + tile)
  int32_t req_cap;
  unsigned long long *dbg;    // [96] diagnostic counters (mod_device.h): cycle counters when DevCam.debug & 128, index assertions of the checked build
};

// inline_consts: the frames' constants ride in the kernel arguments (frames <= MOD_SF_INLINE_FRAMES; a.fc is not read), or null
#define MOD_SF_INLINE_FRAMES 8
void launch_scene_flow(const DevCam &c, const SfArgs &a, int frames, const FrameConst *inline_consts, hipStream_t s);
void launch_dynamic_mask(const DevCam &c, int frames, const float *vx, const float *vy, const float *vz, uint64_t *mask,
                         hipStream_t s);
void launch_depth(const DevCam &c, int frames, const float *dnow, float *depth, hipStream_t s);
// n 8-byte words from (device-visible, e.g. pinned host) src to dst
void launch_copy_words(const unsigned long long *src, unsigned long long *dst, size_t n, hipStream_t s);
void launch_pack(size_t n, const float *x, const float *y, const float *z, const float *vx, const float *vy, const float *vz,
                 void *aos, hipStream_t s);
void launch_unpack(size_t n, const void *aos, float *x, float *y, float *z, float *vx, float *vy, float *vz, hipStream_t s);

// clustering kernels, one launcher per timed stage (include/mod_sf.h MOD_STAGE_*)
void launch_ccl_tile(const DevCam &c, const ClArgs &a, int frames, hipStream_t s);
void launch_tile_flags(const DevCam &c, const ClArgs &a, int frames, hipStream_t s);   // tile headers from a mask plane
void launch_ccl_link(const DevCam &c, const ClArgs &a, int frames, hipStream_t s);
void launch_ccl_merge(const DevCam &c, const ClArgs &a, int frames, ClusterInfo *rank_scratch, hipStream_t s);   // + size filter
void launch_final(const DevCam &c, const ClArgs &a, int frames, hipStream_t s);
void launch_median(const DevCam &c, const ClArgs &a, int frames, hipStream_t s);
int ccl_tile_rows();                 // tile height of k_ccl_tile
int ccl_request_capacity(int n);     // link requests one tile can emit at neighbor_distance n

// on-GPU disparity, first stages (sgm.hip)
void launch_sgm_census(int W, int H, int frames, const uint8_t *img, uint32_t *out, hipStream_t s);
// right_plane_padded: the 127 words before `cr` are readable (the four-lines-per-wave kernels of D == 128 read their census
// windows unconditionally; words left of a row belong to disparities that do not exist and are never used)
void launch_sgm_path(int W, int H, int frames, int D, int P1, int P2, int direction, const uint32_t *cl, const uint32_t *cr,
                     uint8_t *L, uint8_t *cost, bool right_plane_padded, hipStream_t s);
bool launch_sgm_paths_all(int W, int H, int frames, int D, int P1, int P2, int paths, size_t path_stride, const uint32_t *cl, const uint32_t *cr,
                          uint8_t *L, hipStream_t s);
void launch_sgm_finish(int W, int H, int frames, int D, int paths, size_t path_stride, int median, int lr_check, const uint8_t *Lv,
                       uint8_t *dl, uint8_t *dr, uint8_t *dlm, uint8_t *drm, float *disparity, hipStream_t s);
