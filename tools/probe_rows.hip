// Achievable read bandwidth of the decode GEMM's ACCESS PATTERN with no compute (tuning aid):
// a persistent workgroup walks tiles of R weight rows; per item every thread loads one 16-byte unit, the
// workgroup covering R rows x SEG contiguous bytes; D items are in flight per thread.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/probe_rows tools/probe_rows.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int D>
__global__ void k_rows(const unsigned char* __restrict__ src, int nrows, int row_bytes, int R, int seg, unsigned* __restrict__ out) {
  const int tid = threadIdx.x;
  const int units_per_row = seg / 16;              // threads per row
  const int r = tid / units_per_row, c = tid % units_per_row;
  const int tiles = nrows / R;
  const int items = row_bytes / seg;
  unsigned acc = 0;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const unsigned char* base = src + (size_t)(t * R + r) * row_bytes + c * 16;
    for (int i = 0; i < items; i += D) {
      uint4 v[D];
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int it = i + u < items ? i + u : items - 1;
        v[u] = *reinterpret_cast<const uint4*>(base + (size_t)it * seg);
      }
#pragma unroll
      for (int u = 0; u < D; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int D>
static int run(const char* tag, int nrows, int row_bytes, int R, int threads, int grid) {
  const size_t bytes = (size_t)nrows * row_bytes;
  int rot = (int)(340000000 / bytes) + 1; if (rot < 2) rot = 2;
  std::vector<unsigned char*> bufs(rot);
  for (int i = 0; i < rot; ++i) { CK(hipMalloc(&bufs[i], bytes)); CK(hipMemset(bufs[i], i + 1, bytes)); }
  unsigned* out; CK(hipMalloc(&out, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int seg = threads / R * 16;
  const int iters = 10;
  for (int i = 0; i < rot; ++i) hipLaunchKernelGGL(k_rows<D>, dim3(grid), dim3(threads), 0, 0, bufs[i], nrows, row_bytes, R, seg, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) for (int i = 0; i < rot; ++i) hipLaunchKernelGGL(k_rows<D>, dim3(grid), dim3(threads), 0, 0, bufs[i], nrows, row_bytes, R, seg, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / (iters * rot);
  printf("%s rows=%6d x %5d B  R=%3d seg=%5d thr=%4d grid=%4d depth=%d: %7.2f us -> %6.0f GB/s\n", tag, nrows, row_bytes, R, seg, threads, grid, D, us, bytes / us / 1e3);
  for (int i = 0; i < rot; ++i) CK(hipFree(bufs[i]));
  CK(hipFree(out));
  return 0;
}

int main() {
  // gate_up of Qwen2.5-7B: 37888 rows x 1824 B (+ scales, ignored); down: 3584 rows x 9504 B
  struct Shape { const char* tag; int nrows, row_bytes; } shapes[] = {{"gateup", 37888, 2048}, {"down  ", 4096, 8192}, {"4096sq", 4096, 2048}};
  for (auto& s : shapes) {
    for (int threads : {512, 1024}) {
      for (int R : {threads / 4, threads / 8, 16}) {   // threads/4: a wave loads 16 rows x 64 B (MFMA operand layout)
        if ((s.row_bytes % (threads / R * 16)) != 0) continue;
        const int grid = threads == 512 ? 512 : 256;
        if (run<1>(s.tag, s.nrows, s.row_bytes, R, threads, grid)) return 1;
        if (run<3>(s.tag, s.nrows, s.row_bytes, R, threads, grid)) return 1;
        if (run<6>(s.tag, s.nrows, s.row_bytes, R, threads, grid)) return 1;
      }
    }
  }
  return 0;
}
