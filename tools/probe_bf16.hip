// Is the gfx950 instruction v_cvt_pk_bf16_f32 (what hipcc emits for a float -> __bf16 cast) the same function as the
// integer round-to-nearest-even the kernels used, for EVERY fp32 bit pattern?  (NaNs: both must give a NaN.)
// hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t soft(float f) {
  uint32_t u = __float_as_uint(f);
  uint32_t r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  return ((u & 0x7fffffffu) > 0x7f800000u) ? ((u >> 16) | 0x40u) : r;
}
__device__ __forceinline__ uint32_t hard(float f) {
  const __bf16 h = (__bf16)f;
  return (uint32_t)__builtin_bit_cast(unsigned short, h);
}

__global__ void k_check(unsigned long long* mism, unsigned* first) {
  const uint32_t hi = blockIdx.x;                       // upper 16 bits of the pattern
  unsigned long long bad = 0;
  for (uint32_t lo = threadIdx.x; lo < 65536u; lo += blockDim.x) {
    const uint32_t u = (hi << 16) | lo;
    const float f = __uint_as_float(u);
    const uint32_t a = soft(f), b = hard(f);
    const bool nan = (u & 0x7fffffffu) > 0x7f800000u;
    const bool ok = nan ? ((b & 0x7f80u) == 0x7f80u && (b & 0x7fu) != 0) : (a == b);
    if (!ok) {
      ++bad;
      if (atomicCAS(first, 0u, 1u) == 0u) { first[1] = u; first[2] = a; first[3] = b; }
    }
  }
  if (bad) atomicAdd(mism, bad);
}

int main() {
  unsigned long long* mism; unsigned* first;
  CK(hipMalloc(&mism, 8)); CK(hipMalloc(&first, 32));
  CK(hipMemset(mism, 0, 8)); CK(hipMemset(first, 0, 32));
  hipLaunchKernelGGL(k_check, dim3(65536), dim3(256), 0, 0, mism, first);
  CK(hipDeviceSynchronize());
  unsigned long long h; unsigned f[8];
  CK(hipMemcpy(&h, mism, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 32, hipMemcpyDeviceToHost));
  printf("all 2^32 fp32 patterns: v_cvt_pk_bf16_f32 vs integer RNE: %llu mismatches\n", h);
  if (h) printf("first: f32=0x%08x soft=0x%04x hard=0x%04x\n", f[1], f[2], f[3]);
  return 0;
}
