// Issue rate of the instructions the NVFP4 dequantisation is made of (gfx950): cycles per wave-instruction for one wave per SIMD
// and for two (s_memtime around an unrolled loop of independent instructions).  build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int kOp>
__global__ __launch_bounds__(512) void k(unsigned long long* out, uint32_t seed, int iters) {
  uint32_t x[8];
  f16x2 h[8];
  for (int i = 0; i < 8; ++i) { x[i] = seed * (threadIdx.x + 1 + i); uint32_t b = 0x3c003c00u + i; __builtin_memcpy(&h[i], &b, 4); }
  unsigned long long t0, t1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      const int i = r & 7;
      if (kOp == 0) h[i] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(x[i], 256.0f, 1);
      if (kOp == 1) h[i] = h[i] * h[(i + 1) & 7];
      if (kOp == 2) x[i] = __builtin_amdgcn_perm(x[i], x[(i + 1) & 7], 0x0c040c00u);
      if (kOp == 3) x[i] = (x[i] & 0x07070707u) | x[(i + 3) & 7];
      if (kOp == 4) { h[i] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(x[i], 256.0f, 1) * h[(i + 1) & 7]; }
      asm volatile("" : "+v"(x[i]), "+v"(h[i]));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  uint32_t acc = 0;
  for (int i = 0; i < 8; ++i) { uint32_t b; __builtin_memcpy(&b, &h[i], 4); acc ^= b ^ x[i]; }
  if (acc == 0x12345u) out[1023] = acc;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int kOp>
static void run(const char* name, int threads) {
  unsigned long long* d;
  hipMalloc(&d, 8192);
  const int iters = 200;
  k<kOp><<<1, threads>>>(d, 12345u, iters);
  k<kOp><<<1, threads>>>(d, 12345u, iters);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * REP * (kOp == 4 ? 2 : 1);
  printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_instr\": %.3f}\n", name, threads / 256, (double)h[0] / n);
  hipFree(d);
}
int main() {
  for (int threads : {256, 512}) {
    run<0>("v_cvt_scalef32_pk_f16_fp4", threads);
    run<1>("v_pk_mul_f16", threads);
    run<2>("v_perm_b32", threads);
    run<3>("v_and_or_b32", threads);
    run<4>("cvt + pk_mul pair", threads);
  }
  return 0;
}
