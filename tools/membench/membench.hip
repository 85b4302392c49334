// Memory skeletons of the scene-flow kernel's traffic (16 B/px read from 3 planes, 24 B/px written to 6 planes, one gather):
// which thread / block shape streams it fastest on gfx950?  Build + run: see tools/membench/run.sh
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Args { const float *dn, *dp, *fl; float *o[6]; int W, H; };

template <class T> __device__ __forceinline__ T ld(const void *b, uint32_t off) { return *reinterpret_cast<const T *>((const char *)b + off); }
template <class T> __device__ __forceinline__ void st(void *b, uint32_t off, const T &v) { *reinterpret_cast<T *>((char *)b + off) = v; }

__device__ __forceinline__ void remap(uint32_t &bx, uint32_t &by, uint32_t &bz, bool on) {
  const uint32_t total = gridDim.x * gridDim.y * gridDim.z;
  if (on && (total & 7u) == 0u) {
    const uint32_t lin = bx + gridDim.x * (by + gridDim.y * bz);
    const uint32_t m = (lin & 7u) * (total >> 3) + (lin >> 3);
    bx = m % gridDim.x; const uint32_t t = m / gridDim.x; by = t % gridDim.y; bz = t / gridDim.y;
  }
}

// V0: the product's shape: block 64x4, thread = 4 px of a row
template <bool REMAP, bool NT>
__global__ __launch_bounds__(256) void k_v0(Args a) {
  uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  remap(bx, by, bz, REMAP);
  const int x0 = (bx * 64 + threadIdx.x) * 4, y = by * 4 + threadIdx.y;
  if (x0 >= a.W || y >= a.H) return;
  const size_t fN = (size_t)bz * a.W * a.H;
  const uint32_t pix = y * a.W + x0, o4 = pix * 4, o8 = pix * 8;
  const float4 dn = ld<float4>(a.dn + fN, o4), dp = ld<float4>(a.dp + fN, o4), fa = ld<float4>(a.fl + 2 * fN, o8), fb = ld<float4>(a.fl + 2 * fN, o8 + 16);
  const float g = ld<float>(a.dp + fN, ((pix ^ 64u) % (uint32_t)(a.W * a.H)) * 4u);
  const float4 q = make_float4(dn.x + dp.x, fa.x + fb.x, dn.w + fa.w, g + fb.y);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (NT) { typedef float v4 __attribute__((ext_vector_type(4))); v4 t = {q.x, q.y, q.z, q.w}; __builtin_nontemporal_store(t, (v4 *)((char *)(a.o[k] + fN) + o4)); }
    else st(a.o[k] + fN, o4, q);
  }
}

// Partial skeletons: which side of the traffic costs the bandwidth?  MODE bit 0: reads, bit 1: gather, bit 2: six writes (else one)
template <int MODE>
__global__ __launch_bounds__(256) void k_part(Args a) {
  uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  remap(bx, by, bz, true);
  const int x0 = (bx * 64 + threadIdx.x) * 4, y = by * 4 + threadIdx.y;
  if (x0 >= a.W || y >= a.H) return;
  const size_t fN = (size_t)bz * a.W * a.H;
  const uint32_t pix = y * a.W + x0, o4 = pix * 4, o8 = pix * 8;
  float4 q = make_float4((float)x0, (float)y, 1.f, 2.f);
  if (MODE & 1) {
    const float4 dn = ld<float4>(a.dn + fN, o4), dp = ld<float4>(a.dp + fN, o4), fa = ld<float4>(a.fl + 2 * fN, o8), fb = ld<float4>(a.fl + 2 * fN, o8 + 16);
    q = make_float4(dn.x + dp.x, fa.x + fb.x, dn.w + fa.w, dp.y + fb.y);
  }
  if (MODE & 2) q.w += ld<float>(a.dp + fN, ((pix ^ 64u) % (uint32_t)(a.W * a.H)) * 4u);
  if (MODE & 4) {
#pragma unroll
    for (int k = 0; k < 6; k++) st(a.o[k] + fN, o4, q);
  } else if (q.x == 12345.678f) st(a.o[0] + fN, o4, q);   // keeps the loads alive, never true
}

// Row-interleaved output planes: one buffer, row y holds the six planes' row segments back to back (pitch = 6 W), so the six
// stores of a wave land in one 6 x W x 4 B region instead of six regions a plane apart.  MODE as k_part.
template <int MODE>
__global__ __launch_bounds__(256) void k_rowil(Args a) {
  uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  remap(bx, by, bz, true);
  const int x0 = (bx * 64 + threadIdx.x) * 4, y = by * 4 + threadIdx.y;
  if (x0 >= a.W || y >= a.H) return;
  const size_t fN = (size_t)bz * a.W * a.H;
  const uint32_t pix = y * a.W + x0, o4 = pix * 4, o8 = pix * 8;
  float4 q = make_float4((float)x0, (float)y, 1.f, 2.f);
  if (MODE & 1) {
    const float4 dn = ld<float4>(a.dn + fN, o4), dp = ld<float4>(a.dp + fN, o4), fa = ld<float4>(a.fl + 2 * fN, o8), fb = ld<float4>(a.fl + 2 * fN, o8 + 16);
    q = make_float4(dn.x + dp.x, fa.x + fb.x, dn.w + fa.w, dp.y + fb.y);
  }
  if (MODE & 2) q.w += ld<float>(a.dp + fN, ((pix ^ 64u) % (uint32_t)(a.W * a.H)) * 4u);
  float *base = a.o[0] + 6 * fN;                      // o[0] is allocated 6 planes long in this variant
  const uint32_t row = ((uint32_t)y * 6u * (uint32_t)a.W + (uint32_t)x0) * 4u;
#pragma unroll
  for (int k = 0; k < 6; k++) st(base, row + (uint32_t)k * (uint32_t)a.W * 4u, q);
}

// one plane written (float4 per thread, the product's block shape): per-plane write bandwidth
__global__ __launch_bounds__(256) void k_write1(float *o, int W, int H) {
  uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  remap(bx, by, bz, true);
  const int x0 = (bx * 64 + threadIdx.x) * 4, y = by * 4 + threadIdx.y;
  if (x0 >= W || y >= H) return;
  const size_t fN = (size_t)bz * W * H;
  st(o + fN, (uint32_t)(y * W + x0) * 4u, make_float4((float)x0, (float)y, 1.f, 2.f));
}

// V1: thread = 8 px (two float4 per plane), block 64x4 -> a wave covers 512 px of a row
__global__ __launch_bounds__(256) void k_v1(Args a) {
  uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  remap(bx, by, bz, true);
  const int y = by * 4 + threadIdx.y;
  const size_t fN = (size_t)bz * a.W * a.H;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int x0 = (bx * 128 + h * 64 + threadIdx.x) * 4;
    if (x0 >= a.W || y >= a.H) continue;
    const uint32_t pix = y * a.W + x0, o4 = pix * 4, o8 = pix * 8;
    const float4 dn = ld<float4>(a.dn + fN, o4), dp = ld<float4>(a.dp + fN, o4), fa = ld<float4>(a.fl + 2 * fN, o8), fb = ld<float4>(a.fl + 2 * fN, o8 + 16);
    const float g = ld<float>(a.dp + fN, ((pix ^ 64u) % (uint32_t)(a.W * a.H)) * 4u);
    const float4 q = make_float4(dn.x + dp.x, fa.x + fb.x, dn.w + fa.w, g + fb.y);
#pragma unroll
    for (int k = 0; k < 6; k++) st(a.o[k] + fN, o4, q);
  }
}

// V2: block 256x1: four waves side by side in ONE row (1024 px), rows by blockIdx.y
__global__ __launch_bounds__(256) void k_v2(Args a) {
  uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  remap(bx, by, bz, true);
  const int x0 = (bx * 256 + threadIdx.x) * 4, y = by;
  if (x0 >= a.W || y >= a.H) return;
  const size_t fN = (size_t)bz * a.W * a.H;
  const uint32_t pix = y * a.W + x0, o4 = pix * 4, o8 = pix * 8;
  const float4 dn = ld<float4>(a.dn + fN, o4), dp = ld<float4>(a.dp + fN, o4), fa = ld<float4>(a.fl + 2 * fN, o8), fb = ld<float4>(a.fl + 2 * fN, o8 + 16);
  const float g = ld<float>(a.dp + fN, ((pix ^ 64u) % (uint32_t)(a.W * a.H)) * 4u);
  const float4 q = make_float4(dn.x + dp.x, fa.x + fb.x, dn.w + fa.w, g + fb.y);
#pragma unroll
  for (int k = 0; k < 6; k++) st(a.o[k] + fN, o4, q);
}

// V3: frame-linear: ignore rows, thread = 4 consecutive px of the flattened frame (W*H % 4 == 0), 1D blocks of 256
__global__ __launch_bounds__(256) void k_v3(Args a) {
  uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  remap(bx, by, bz, true);
  const uint32_t pix = (bx * 256 + threadIdx.x) * 4;
  if (pix >= (uint32_t)(a.W * a.H)) return;
  const size_t fN = (size_t)bz * a.W * a.H;
  const uint32_t o4 = pix * 4, o8 = pix * 8;
  const float4 dn = ld<float4>(a.dn + fN, o4), dp = ld<float4>(a.dp + fN, o4), fa = ld<float4>(a.fl + 2 * fN, o8), fb = ld<float4>(a.fl + 2 * fN, o8 + 16);
  const float g = ld<float>(a.dp + fN, ((pix ^ 64u) % (uint32_t)(a.W * a.H)) * 4u);
  const float4 q = make_float4(dn.x + dp.x, fa.x + fb.x, dn.w + fa.w, g + fb.y);
#pragma unroll
  for (int k = 0; k < 6; k++) st(a.o[k] + fN, o4, q);
}

// plain copy of the same byte volume for reference: 40 B/px as float4 in -> float4 out (20 B/px each way)
__global__ __launch_bounds__(256) void k_copy(const float4 *in, float4 *out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
  const int W = 1280, H = 720, F = argc > 1 ? atoi(argv[1]) : 128;
  const size_t N = (size_t)W * H, FN = N * F;
  Args a; a.W = W; a.H = H;
  float *dn, *dp, *fl, *o[6];
  CK(hipMalloc(&dn, FN * 4)); CK(hipMalloc(&dp, FN * 4)); CK(hipMalloc(&fl, FN * 8));
  for (int k = 0; k < 6; k++) { CK(hipMalloc(&o[k], FN * 4)); a.o[k] = o[k]; }
  CK(hipMemset(dn, 0, FN * 4)); CK(hipMemset(dp, 0, FN * 4)); CK(hipMemset(fl, 0, FN * 8));
  a.dn = dn; a.dp = dp; a.fl = fl;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 40.0 * FN;
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 2; i++) launch();
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-44s %8.3f ms  %6.2f us/frame  %6.0f GB/s\n", name, ms, 1e3 * ms / F, bytes / ms / 1e6);
  };
  run("V0 64x4, 4 px/thread (product)", [&] { hipLaunchKernelGGL((k_v0<false, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, a); });
  run("V0 + XCD remap", [&] { hipLaunchKernelGGL((k_v0<true, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, a); });
  run("V0 + XCD remap + nontemporal stores", [&] { hipLaunchKernelGGL((k_v0<true, true>), dim3(5, 180, F), dim3(64, 4), 0, 0, a); });
  if (getenv("SPACING")) {   // one allocation, the nine planes laid out with a chosen gap between them: does the spacing set the speed?
    const size_t KB = 1024, MBy = 1024 * 1024;
    const size_t gaps[] = {0, 2 * MBy, 4 * MBy, 6 * MBy, 8 * MBy, 16 * MBy, 32 * MBy, 62 * MBy, 64 * KB, 256 * KB, 1 * MBy, 3 * MBy, 5 * MBy, 7 * MBy, 0, 2 * MBy};
    char *blk; CK(hipMalloc(&blk, FN * 40 + 10 * 64 * MBy + 4 * MBy));
    CK(hipMemset(blk, 0, FN * 40 + 10 * 64 * MBy + 4 * MBy));
    char *blk0 = (char *)(((size_t)blk + 2 * MBy - 1) / (2 * MBy) * (2 * MBy));
    for (size_t gap : gaps) {
      Args b = a; char *p = blk0;
      auto take = [&](size_t bytes) { char *r = p; p += bytes + gap; return (float *)r; };
      b.dn = take(FN * 4); b.dp = take(FN * 4); b.fl = take(FN * 8); for (int k = 0; k < 6; k++) b.o[k] = take(FN * 4);
      char name[96]; snprintf(name, sizeof name, "one block, gap %zu KiB between planes", gap / KB);
      run(name, [&] { hipLaunchKernelGGL((k_v0<true, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, b); });
    }
    CK(hipFree(blk));
    return 0;
  }
  if (getenv("PLANE_SIZES")) {   // single planes allocated with different (padded) sizes: does the allocation size decide the plane's write speed?
    const size_t MBy = 1024 * 1024;
    const size_t sizes[] = {FN * 4, FN * 4 + 2 * MBy, 512 * MBy, 1024 * MBy, 2048 * MBy};
    std::vector<float *> keep;
    for (size_t sz : sizes) {
      printf("  planes of %4zu MiB allocations, written alone (GB/s):", sz / MBy);
      for (int k = 0; k < 10; k++) {
        float *o; CK(hipMalloc(&o, sz)); keep.push_back(o);
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_write1, dim3(5, 180, F), dim3(64, 4), 0, 0, o, W, H);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_write1, dim3(5, 180, F), dim3(64, 4), 0, 0, o, W, H);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf(" %5.0f", 4.0 * FN / ms / 1e6);
      }
      printf("\n");
    }
    for (float *q : keep) CK(hipFree(q));
    return 0;
  }
  if (getenv("BLOCK_TRIALS")) {   // like PLACEMENT_TRIALS, but every set of nine planes is carved out of ONE allocation
    const int trials = atoi(getenv("BLOCK_TRIALS"));
    const size_t MB2 = (size_t)2 << 20;
    std::vector<char *> keep;
    for (int tr = 0; tr < trials; tr++) {
      Args b = a; char *blk; CK(hipMalloc(&blk, FN * 40 + 16 * MB2)); keep.push_back(blk);
      CK(hipMemset(blk, 0, FN * 40 + 16 * MB2));
      char *p = (char *)(((size_t)blk + MB2 - 1) / MB2 * MB2);
      auto take = [&](size_t bytes) { char *r = p; p += (bytes + MB2 - 1) / MB2 * MB2; return (float *)r; };
      b.dn = take(FN * 4); b.dp = take(FN * 4); b.fl = take(FN * 8); for (int k = 0; k < 6; k++) b.o[k] = take(FN * 4);
      char name[96]; snprintf(name, sizeof name, "block trial %d (va %p)", tr, (void *)blk);
      run(name, [&] { hipLaunchKernelGGL((k_v0<true, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, b); });
    }
    for (char *q : keep) CK(hipFree(q));
    return 0;
  }
  if (getenv("PLACEMENT_TRIALS")) {   // same kernel on fresh sets of nine planes, earlier sets kept alive: how much does placement matter?
    const int trials = atoi(getenv("PLACEMENT_TRIALS"));
    std::vector<float *> keep;
    for (int tr = 0; tr < trials; tr++) {
      Args b = a; float *base[9];
      for (int k = 0; k < 9; k++) { CK(hipMalloc(&base[k], FN * (k == 2 ? 8 : 4))); keep.push_back(base[k]); }
      b.dn = base[0]; b.dp = base[1]; b.fl = base[2]; for (int k = 0; k < 6; k++) b.o[k] = base[3 + k];
      CK(hipMemset(base[0], 0, FN * 4)); CK(hipMemset(base[1], 0, FN * 4)); CK(hipMemset(base[2], 0, FN * 8));
      char name[96]; snprintf(name, sizeof name, "placement trial %d (va %p)", tr, (void *)base[0]);
      run(name, [&] { hipLaunchKernelGGL((k_v0<true, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, b); });
    }
    for (int round = 0; round < 1; round++)            // the same sets again: is the speed a property of the set or of the moment?
      for (int tr = 0; tr < trials; tr++) {
        Args b = a; float **base = &keep[9 * tr];
        b.dn = base[0]; b.dp = base[1]; b.fl = base[2]; for (int k = 0; k < 6; k++) b.o[k] = base[3 + k];
        char name[96]; snprintf(name, sizeof name, "  again: set %d", tr);
        run(name, [&] { hipLaunchKernelGGL((k_v0<true, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, b); });
      }
    {   // every output plane of every set written alone
      for (int tr = 0; tr < trials; tr++) {
        printf("  set %d, planes written alone (GB/s):", tr);
        for (int k = 3; k < 9; k++) {
          float *o = keep[9 * tr + k];
          for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_write1, dim3(5, 180, F), dim3(64, 4), 0, 0, o, W, H);
          CK(hipEventRecord(e0));
          for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_write1, dim3(5, 180, F), dim3(64, 4), 0, 0, o, W, H);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
          printf(" %5.0f", 4.0 * FN / ms / 1e6);
        }
        printf("\n");
      }
    }
    {   // inputs of one set with outputs of another: which side carries the effect?
      for (int i = 0; i < std::min(trials, 4); i++)
        for (int j = 0; j < std::min(trials, 4); j++) {
          Args b = a; float **bi = &keep[9 * i], **bo = &keep[9 * j];
          b.dn = bi[0]; b.dp = bi[1]; b.fl = bi[2]; for (int k = 0; k < 6; k++) b.o[k] = bo[3 + k];
          char name[96]; snprintf(name, sizeof name, "  inputs of set %d, outputs of set %d", i, j);
          run(name, [&] { hipLaunchKernelGGL((k_v0<true, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, b); });
        }
    }
    for (float *q : keep) CK(hipFree(q));
    return 0;
  }
  {   // placement: nine separate allocations vs one contiguous block vs 2 MiB-aligned, 2 MiB-padded planes
    const size_t MB2 = (size_t)2 << 20;
    for (int rep = 0; rep < 3; rep++) {
      {
        Args b = a; float *base[9];
        for (int k = 0; k < 9; k++) CK(hipMalloc(&base[k], FN * (k == 2 ? 8 : 4)));
        b.dn = base[0]; b.dp = base[1]; b.fl = base[2]; for (int k = 0; k < 6; k++) b.o[k] = base[3 + k];
        char name[96]; snprintf(name, sizeof name, "separate allocations (base0 %% 2MiB = %zu KiB)", ((size_t)base[0] % MB2) >> 10);
        run(name, [&] { hipLaunchKernelGGL((k_v0<true, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, b); });
        for (int k = 0; k < 9; k++) CK(hipFree(base[k]));
      }
      {
        Args b = a; char *blk; CK(hipMalloc(&blk, FN * 40 + 16 * MB2));
        char *p = (char *)(((size_t)blk + MB2 - 1) / MB2 * MB2);
        auto take = [&](size_t bytes) { char *r = p; p += (bytes + MB2 - 1) / MB2 * MB2; return (float *)r; };
        b.dn = take(FN * 4); b.dp = take(FN * 4); b.fl = take(FN * 8); for (int k = 0; k < 6; k++) b.o[k] = take(FN * 4);
        run("one block, planes 2 MiB aligned", [&] { hipLaunchKernelGGL((k_v0<true, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, b); });
        CK(hipFree(blk));
      }
    }
  }
  {   // same kernel, plane bases skewed against each other (do the 9 streams collide in the channel / bank hash?)
    for (size_t skew : {(size_t)0, (size_t)4096, (size_t)65536 + 4096 + 256, (size_t)0}) {
      Args b = a;
      float *base[9];
      for (int k = 0; k < 9; k++) CK(hipMalloc(&base[k], FN * (k == 2 ? 8 : 4) + 16 * skew));
      b.dn = (float *)((char *)base[0] + 1 * skew); b.dp = (float *)((char *)base[1] + 2 * skew); b.fl = (float *)((char *)base[2] + 3 * skew);
      for (int k = 0; k < 6; k++) b.o[k] = (float *)((char *)base[3 + k] + (4 + k) * skew);
      char name[96]; snprintf(name, sizeof name, "V0 + remap, plane bases skewed by k x %zu B", skew);
      run(name, [&] { hipLaunchKernelGGL((k_v0<true, false>), dim3(5, 180, F), dim3(64, 4), 0, 0, b); });
      for (int k = 0; k < 9; k++) CK(hipFree(base[k]));
    }
  }
  {
    const double all = bytes;
    auto runp = [&](const char *name, double b, auto launch) {
      for (int i = 0; i < 2; i++) launch();
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; i++) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      printf("%-44s %8.3f ms  %6.2f us/frame  %6.0f GB/s of its own bytes\n", name, ms, 1e3 * ms / F, b / ms / 1e6);
    };
    (void)all;
    runp("reads only (16 B/px)", 16.0 * FN, [&] { hipLaunchKernelGGL((k_part<1>), dim3(5, 180, F), dim3(64, 4), 0, 0, a); });
    runp("reads + gather (16 B/px)", 16.0 * FN, [&] { hipLaunchKernelGGL((k_part<3>), dim3(5, 180, F), dim3(64, 4), 0, 0, a); });
    runp("six plane writes only (24 B/px)", 24.0 * FN, [&] { hipLaunchKernelGGL((k_part<4>), dim3(5, 180, F), dim3(64, 4), 0, 0, a); });
    runp("reads + six writes, no gather (40 B/px)", 40.0 * FN, [&] { hipLaunchKernelGGL((k_part<5>), dim3(5, 180, F), dim3(64, 4), 0, 0, a); });
  }
  {
    Args b = a;
    float *big; CK(hipMalloc(&big, FN * 4 * 6));
    b.o[0] = big;
    run("row-interleaved outputs, full traffic", [&] { hipLaunchKernelGGL((k_rowil<3>), dim3(5, 180, F), dim3(64, 4), 0, 0, b); });
    CK(hipFree(big));
  }
  run("V1 64x4, 8 px/thread", [&] { hipLaunchKernelGGL(k_v1, dim3(3, 180, F), dim3(64, 4), 0, 0, a); });
  run("V2 256x1 (one row per block)", [&] { hipLaunchKernelGGL(k_v2, dim3(2, 720, F), dim3(256), 0, 0, a); });
  run("V3 frame-linear 1D", [&] { hipLaunchKernelGGL(k_v3, dim3(900, 1, F), dim3(256), 0, 0, a); });
  {
    const size_t n4 = FN * 20 / 16;   // 20 B/px in, 20 B/px out
    float4 *in = (float4 *)fl, *out = (float4 *)o[0];
    (void)in; (void)out;
    float4 *ci, *co; CK(hipMalloc(&ci, n4 * 16)); CK(hipMalloc(&co, n4 * 16));
    run("float4 copy, same byte volume", [&] { hipLaunchKernelGGL(k_copy, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, ci, co, n4); });
  }
  return 0;
}
