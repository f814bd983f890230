// Is  bf16( a * r corrected by two FMAs )  ==  bf16( a / b )  for every bf16 a and every bf16 scale b in range?
// (tuning aid for the dynamic-scale quantiser: an IEEE fp32 division costs ~10 instructions per element, the
// reciprocal + Markstein correction 3 + a sign fix-up.)   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float bf16f(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f2bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__global__ void k_check(int e_lo, int e_hi, unsigned long long* mism, unsigned* first) {
  // blockIdx.x enumerates b: exponent field e in [e_lo, e_hi], 7-bit mantissa; threads enumerate a
  const int eb = e_lo + blockIdx.x / 128, mb = blockIdx.x % 128;
  if (eb > e_hi) return;
  const float b = bf16f((uint32_t)(eb << 7) | mb);
  const float r = 1.0f / b;
  unsigned long long bad = 0;
  for (uint32_t ab = threadIdx.x; ab < 65536u; ab += blockDim.x) {
    const uint32_t ea = (ab >> 7) & 0xffu;
    if (ea == 0xffu) continue;                                   // inf / nan inputs are out of contract
    const float a = bf16f(ab);
    if (fabsf(a) > 4096.0f * b) continue;                        // b = amax/2688 of the same tensor: |a/b| <= 2688
    const float want = a / b;
    float q = a * r;
    const float e = __builtin_fmaf(-q, b, a);
    q = __builtin_fmaf(e, r, q);
    q = __builtin_copysignf(q, a);
    if (f2bf16(q) != f2bf16(want)) {
      ++bad;
      if (atomicCAS(first, 0u, 1u) == 0u) { first[1] = ab; first[2] = (uint32_t)(eb << 7) | mb; first[3] = f2bf16(q); first[4] = f2bf16(want); }
    }
  }
  if (bad) atomicAdd(mism, bad);
}

int main() {
  unsigned long long* mism; unsigned* first;
  CK(hipMalloc(&mism, 8)); CK(hipMalloc(&first, 32));
  CK(hipMemset(mism, 0, 8)); CK(hipMemset(first, 0, 32));
  const int e_lo = 127 - 100, e_hi = 127 + 100;                  // scales 2^-100 .. 2^100
  hipLaunchKernelGGL(k_check, dim3((e_hi - e_lo + 1) * 128), dim3(256), 0, 0, e_lo, e_hi, mism, first);
  CK(hipDeviceSynchronize());
  unsigned long long h; unsigned f[8];
  CK(hipMemcpy(&h, mism, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 32, hipMemcpyDeviceToHost));
  printf("scales: bf16 with exponent fields %d..%d; a: every finite bf16 with |a| <= 4096 b.  mismatching (a, b) pairs: %llu\n", e_lo, e_hi, h);
  if (h) printf("first: a=0x%04x b=0x%04x fast=0x%04x ieee=0x%04x\n", f[1], f[2], f[3], f[4]);
  return 0;
}
