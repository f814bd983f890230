// Per-kernel floor of a chain of dependent tiny kernels (tuning aid): plain stream launches from a tight host
// loop against the same chain replayed from a HIP graph.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_tiny(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

int main() {
  unsigned* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 2000;
  for (int grid : {1, 64, 512}) {
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_tiny, dim3(grid), dim3(256), 0, s, d);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_tiny, dim3(grid), dim3(256), 0, s, d);
    CK(hipEventRecord(e1, s));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
    printf("grid %3d  stream launches: %.2f us per kernel on the device, %.2f us of host time per launch\n", grid, ms * 1e3 / n, host_us);

    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k_tiny, dim3(grid), dim3(256), 0, s, d);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("grid %3d  graph replay   : %.2f us per kernel node\n", grid, ms * 1e3 / (5 * 400));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
