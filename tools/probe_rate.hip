// Issue rate of the dequantisation instructions on gfx950 (tuning aid): cycles per wave-instruction for
// v_cvt_scalef32_pk_f16_fp4, v_pk_mul_f16, v_perm_b32 and an interleaved cvt+mul stream, at 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/probe_rate tools/probe_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 2000, kUnroll = 16;

template <int kKind>
__global__ void k_rate(unsigned* out, unsigned long long* cyc, unsigned seed) {
  unsigned c0 = seed + threadIdx.x, c1 = c0 * 3, c2 = c0 * 5, c3 = c0 * 7;
  f16x2 a0 = {1, 2}, a1 = {3, 4}, a2 = {5, 6}, a3 = {7, 8};
  const f16x2 s = {(_Float16)1.0009765625f, (_Float16)0.99951171875f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < kUnroll / 4; ++u) {
      if (kKind == 0) {        // 4 independent cvt
        a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(c0, 256.0f, 0); a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(c1, 256.0f, 1);
        a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(c2, 256.0f, 2); a3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(c3, 256.0f, 3);
        unsigned t; __builtin_memcpy(&t, &a0, 4); c0 ^= t; __builtin_memcpy(&t, &a1, 4); c1 ^= t;
        __builtin_memcpy(&t, &a2, 4); c2 ^= t; __builtin_memcpy(&t, &a3, 4); c3 ^= t;      // + 4 xor (full rate)
      } else if (kKind == 1) { // 4 independent pk_mul (+ nothing)
        a0 = a0 * s; a1 = a1 * s; a2 = a2 * s; a3 = a3 * s;
      } else if (kKind == 2) { // 4 xor only (baseline for kind 0)
        c0 ^= c1 + u; c1 ^= c2; c2 ^= c3; c3 ^= c0;
      } else if (kKind == 3) { // v_perm_b32 x4
        c0 = __builtin_amdgcn_perm(c0, c1, 0x07050301u); c1 = __builtin_amdgcn_perm(c1, c2, 0x06040200u);
        c2 = __builtin_amdgcn_perm(c2, c3, 0x07050301u); c3 = __builtin_amdgcn_perm(c3, c0, 0x06040200u);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned t; __builtin_memcpy(&t, &a0, 4); unsigned r = c0 ^ c1 ^ c2 ^ c3 ^ t;
  __builtin_memcpy(&t, &a1, 4); r ^= t; __builtin_memcpy(&t, &a2, 4); r ^= t; __builtin_memcpy(&t, &a3, 4); r ^= t;
  if (r == 0x12345u) out[0] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int kKind>
static int run(const char* name, int threads) {
  unsigned* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 8 * 256));
  hipLaunchKernelGGL(k_rate<kKind>, dim3(256), dim3(threads), 0, 0, out, cyc, 12345u);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rate<kKind>, dim3(256), dim3(threads), 0, 0, out, cyc, 12345u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double ns_per_group = ms * 1e6 / (kIters * (kUnroll / 4));   // wall time per 4-instruction group of EVERY wave on a SIMD
  unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
  const double per_group = avg / (kIters * (kUnroll / 4));     // cycles per group of 4 (or 8) instructions per wave
  printf("%-28s waves/SIMD=%d: %8.1f cycles per 4-instruction group per wave (s_memtime units; x%d waves on the SIMD)\n", name, threads / 256,
         per_group, threads / 256);
  printf("%-28s   wall: %.2f ns per group-round = %.2f ns per wave-instruction slot (%d waves x 4 instr)\n", "", ns_per_group, ns_per_group / (4.0 * (threads / 256)), threads / 256);
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main() {
  for (int threads : {256, 512, 1024}) {
    if (run<2>("4 x v_xor/add (baseline)", threads)) return 1;
    if (run<0>("4 x cvt_scalef32_pk + 4 xor", threads)) return 1;
    if (run<1>("4 x v_pk_mul_f16", threads)) return 1;
    if (run<3>("4 x v_perm_b32", threads)) return 1;
  }
  return 0;
}
