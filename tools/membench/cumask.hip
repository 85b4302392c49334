// Do two HIP streams with disjoint CU masks run kernels side by side on gfx950?  A bandwidth kernel (stand-in for the
// scene-flow / tile kernels) on 240 CUs and a latency-bound kernel of a few big workgroups (stand-in for k_median /
// k_median_ties) on the remaining 16, alone and together.  Build + run: hipcc -O3 --offload-arch=gfx950 cumask.hip && ./a.out
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_stream(const float4 *in, float4 *out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(1024) void k_chain(int *p, int steps) {       // dependent loads: latency-bound, one big workgroup each
  __shared__ int s[1024];
  int v = p[blockIdx.x * 1024 + threadIdx.x];
  for (int i = 0; i < steps; i++) { v = p[(v + i) & 0xfffff]; s[threadIdx.x] = v; __syncthreads(); v += s[(threadIdx.x + 1) & 1023]; }
  p[blockIdx.x * 1024 + threadIdx.x] = v;
}

int main() {
  const size_t n = (size_t)1 << 27;            // 2 GiB in, 2 GiB out
  float4 *in, *out; int *chain;
  CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&chain, (1 << 20) * 4 + 64 * 1024 * 4));
  CK(hipMemset(in, 0, n * 16)); CK(hipMemset(chain, 0, (1 << 20) * 4 + 64 * 1024 * 4));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%d CUs\n", cus);
  // masks: bit i = CU i; the last 16 CUs for the side stream
  const int words = (cus + 31) / 32;
  std::vector<uint32_t> mmain(words, 0), mside(words, 0);
  for (int i = 0; i < cus; i++) (i < cus - 16 ? mmain : mside)[i / 32] |= 1u << (i % 32);
  hipStream_t sm, ss, plain;
  CK(hipExtStreamCreateWithCUMask(&sm, words, mmain.data()));
  CK(hipExtStreamCreateWithCUMask(&ss, words, mside.data()));
  CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto fn) {
    fn(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); CK(hipStreamSynchronize(0));
    const auto t0 = std::chrono::steady_clock::now();
    fn(); CK(hipDeviceSynchronize());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("%-52s %8.3f ms\n", name, ms);
  };
  const int steps = 3000;
  timeit("copy alone, unmasked stream", [&] { hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, plain, in, out, n); });
  timeit("copy alone, 240-CU stream", [&] { hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, sm, in, out, n); });
  timeit("chain alone (16 workgroups), unmasked stream", [&] { hipLaunchKernelGGL(k_chain, dim3(16), dim3(1024), 0, plain, chain, steps); });
  timeit("chain alone, 16-CU stream", [&] { hipLaunchKernelGGL(k_chain, dim3(16), dim3(1024), 0, ss, chain, steps); });
  timeit("copy + chain, two unmasked streams", [&] {
    hipStream_t p2; CK(hipStreamCreateWithFlags(&p2, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, plain, in, out, n);
    hipLaunchKernelGGL(k_chain, dim3(16), dim3(1024), 0, p2, chain, steps);
    CK(hipDeviceSynchronize()); CK(hipStreamDestroy(p2)); });
  timeit("copy (240 CUs) + chain (16 CUs), masked streams", [&] {
    hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, sm, in, out, n);
    hipLaunchKernelGGL(k_chain, dim3(16), dim3(1024), 0, ss, chain, steps); });
  return 0;
}
