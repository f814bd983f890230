// Floor probe: how fast can ONE short kernel stream ~9.6 MB (and 33 MB) from HBM on this GPU?
// Rotates through > 256 MiB of source buffers so every launch misses the Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int UNROLL>
__global__ void k_read(const uint4* __restrict__ src, size_t n16, unsigned* __restrict__ out) {
  // each workgroup owns a contiguous span; every load of a thread is issued before the first use
  const size_t per_wg = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t base = (size_t)blockIdx.x * per_wg;
  const size_t end = base + per_wg < n16 ? base + per_wg : n16;
  unsigned acc = 0;
  for (size_t i = base + threadIdx.x; i < end; i += (size_t)blockDim.x * UNROLL) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      size_t j = i + (size_t)u * blockDim.x;
      v[u] = j < end ? src[j] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;   // practically never: keeps the loads live
}

int main() {
  const size_t sizes[] = {9584640, 22063104, 40108032, 83066880, 134184960};
  for (size_t bytes : sizes) {
    int rot = (int)(340000000 / bytes) + 1; if (rot < 2) rot = 2;
    std::vector<uint4*> bufs(rot);
    for (int i = 0; i < rot; ++i) { CK(hipMalloc(&bufs[i], bytes)); CK(hipMemset(bufs[i], i + 1, bytes)); }
    unsigned* out; CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int cfgs[][2] = {{256, 512}, {256, 1024}, {512, 256}, {1024, 256}, {2048, 256}, {4096, 256}};
    for (auto& c : cfgs) {
      const int iters = 20;
      for (int w = 0; w < 2; ++w) for (int i = 0; i < rot; ++i) hipLaunchKernelGGL(k_read<4>, dim3(c[0]), dim3(c[1]), 0, 0, bufs[i], bytes / 16, out);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int it = 0; it < iters; ++it) for (int i = 0; i < rot; ++i) hipLaunchKernelGGL(k_read<4>, dim3(c[0]), dim3(c[1]), 0, 0, bufs[i], bytes / 16, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double us = ms * 1e3 / (iters * rot);
      printf("read %6.1f MB grid=%4d x %4d thr: %7.2f us/launch (back-to-back) -> %6.0f GB/s\n", bytes / 1e6, c[0], c[1], us, bytes / us / 1e3);
    }
    for (int i = 0; i < rot; ++i) CK(hipFree(bufs[i]));
    CK(hipFree(out));
  }
  return 0;
}
