// Hardware-semantics probe for gfx950 conversion / MFMA instructions used by the ARC-NVFP4 kernels.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_gfx950.hip -o gpurun_out/probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_cvt(const float* scales, int ns, uint32_t* out_bf16, uint32_t* out_f16, float* out_f32) {
  int b = threadIdx.x;          // byte value 0..255 : two fp4 codes
  for (int s = 0; s < ns; ++s) {
    float sc = scales[s];
    uint32_t src = (uint32_t)b; // byte 0
    bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(src, sc, 0);
    f16x2 c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(src, sc, 0);
    f32x2 d = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(src, sc, 0);
    out_bf16[s * 256 + b] = *(uint32_t*)&a;
    out_f16[s * 256 + b] = *(uint32_t*)&c;
    out_f32[(s * 256 + b) * 2 + 0] = d.x;
    out_f32[(s * 256 + b) * 2 + 1] = d.y;
  }
}

// float -> fp4 (scale 1.0) and float -> fp8 rounding probes
__global__ void k_cvt_to(const float* xs, int n, uint32_t* out_fp4, uint32_t* out_fp8) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = xs[i];
  uint32_t r4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, x, x, 1.0f, 0);
  out_fp4[i] = r4;
  uint32_t r8 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false);
  out_fp8[i] = r8;
}

// MFMA f16 16x16x32: C = A(16xK32) * B(K32x16); check lane maps with integer data and denormal handling.
__global__ void k_mfma(const _Float16* A /*16x32 row-major*/, const _Float16* Bt /*16(n)x32(k) row-major*/, float* C /*16x16*/) {
  int l = threadIdx.x;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = Bt[(l & 15) * 32 + 8 * (l >> 4) + j];
  }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

static float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float f16_to_f(uint16_t h) {
  int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
  float v = e == 0 ? ldexpf((float)m, -24) : (e == 31 ? INFINITY : ldexpf(1.0f + m / 1024.0f, e - 15));
  return s ? -v : v;
}
static const float MAG[8] = {0, .5f, 1, 1.5f, 2, 3, 4, 6};
static float e2m1(int c) { float m = MAG[c & 7]; return (c & 8) ? -m : m; }

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, gcn %s, CUs %d, clock %d kHz, LDS/blk %zu\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);

  // ---------------- cvt_scalef32_pk_*_fp4 : is the scale a full multiply or exponent-only?
  std::vector<float> scales = {1.0f, 2.0f, 0.5f, 1.5f, 1.75f, 0.001953125f, 448.0f, 288.0f, 0.013671875f /*7*2^-9*/, 3.0f};
  int ns = (int)scales.size();
  float* dsc; uint32_t *db, *dh; float* df;
  CK(hipMalloc(&dsc, ns * 4)); CK(hipMalloc(&db, ns * 256 * 4)); CK(hipMalloc(&dh, ns * 256 * 4)); CK(hipMalloc(&df, ns * 256 * 8));
  CK(hipMemcpy(dsc, scales.data(), ns * 4, hipMemcpyHostToDevice));
  k_cvt<<<1, 256>>>(dsc, ns, db, dh, df);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> hb(ns * 256), hh(ns * 256); std::vector<float> hf(ns * 512);
  CK(hipMemcpy(hb.data(), db, ns * 256 * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hh.data(), dh, ns * 256 * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hf.data(), df, ns * 256 * 8, hipMemcpyDeviceToHost));
  for (int s = 0; s < ns; ++s) {
    int full_bf = 0, exp_bf = 0, full_h = 0, exp_h = 0, full_f = 0, exp_f = 0;
    float sc = scales[s]; int ex; frexpf(sc, &ex); float pow2 = ldexpf(1.0f, ex - 1);
    for (int b = 0; b < 256; ++b) {
      float lo = e2m1(b & 15), hi = e2m1(b >> 4);
      float g0 = bf16_to_f(hb[s * 256 + b] & 0xffff), g1 = bf16_to_f(hb[s * 256 + b] >> 16);
      full_bf += (g0 == lo * sc && g1 == hi * sc); exp_bf += (g0 == lo * pow2 && g1 == hi * pow2);
      float h0 = f16_to_f(hh[s * 256 + b] & 0xffff), h1 = f16_to_f(hh[s * 256 + b] >> 16);
      full_h += (h0 == lo * sc && h1 == hi * sc); exp_h += (h0 == lo * pow2 && h1 == hi * pow2);
      float f0 = hf[(s * 256 + b) * 2], f1 = hf[(s * 256 + b) * 2 + 1];
      full_f += (f0 == lo * sc && f1 == hi * sc); exp_f += (f0 == lo * pow2 && f1 == hi * pow2);
    }
    printf("cvt_scalef32 fp4 scale=%-12g : bf16 full=%3d exp=%3d | f16 full=%3d exp=%3d | f32 full=%3d exp=%3d   (of 256) sample b=0x75: bf16 %g,%g f32 %g,%g\n",
           sc, full_bf, exp_bf, full_h, exp_h, full_f, exp_f, bf16_to_f(hb[s * 256 + 0x75] & 0xffff), bf16_to_f(hb[s * 256 + 0x75] >> 16), hf[(s * 256 + 0x75) * 2], hf[(s * 256 + 0x75) * 2 + 1]);
  }

  // ---------------- float -> fp4 / fp8 rounding
  std::vector<float> xs = {0.f, -0.f, 0.1f, -0.1f, 0.25f, 0.2500001f, 0.75f, 1.25f, 1.75f, 2.5f, 3.5f, 5.0f, 5.0000005f, 6.f, 7.f, 100.f, -0.25f, -0.75f, -5.f, -100.f,
                           0.001953125f, 0.0029296875f /*1.5*2^-9*/, 0.0048828125f /*2.5*2^-9*/, 0.015625f, 1.0625f, 1.1875f, 447.f, 448.f, 460.f, 464.f, 480.f, 500.f, 1e-4f};
  int nx = (int)xs.size(); float* dx; uint32_t *d4, *d8;
  CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&d4, nx * 4)); CK(hipMalloc(&d8, nx * 4));
  CK(hipMemcpy(dx, xs.data(), nx * 4, hipMemcpyHostToDevice));
  k_cvt_to<<<1, 64>>>(dx, nx, d4, d8); CK(hipDeviceSynchronize());
  std::vector<uint32_t> h4(nx), h8(nx);
  CK(hipMemcpy(h4.data(), d4, nx * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h8.data(), d8, nx * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < nx; ++i) printf("cvt_to x=%-14.9g fp4 code=0x%x (%g)  fp8 byte=0x%02x\n", xs[i], h4[i] & 15, e2m1(h4[i] & 15), h8[i] & 255);

  // ---------------- MFMA f16 layout + denormal inputs
  std::vector<_Float16> A(16 * 32), Bt(16 * 32); std::vector<float> C(256), R(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) { A[i * 32 + k] = (_Float16)((i * 3 + k) % 7 - 3); Bt[i * 32 + k] = (_Float16)((i * 5 + 2 * k) % 5 - 2); }
  A[0] = (_Float16)ldexpf(1.0f, -20);  // f16 denormal
  Bt[0] = (_Float16)4096.0f;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 32; ++k) s += (double)(float)A[i * 32 + k] * (double)(float)Bt[j * 32 + k]; R[i * 16 + j] = (float)s; }
  _Float16 *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
  CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
  k_mfma<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]);
  printf("mfma_f32_16x16x32_f16 layout check: %d mismatches of 256 (C[0][0]=%.9g expect %.9g -> f16 denormal input %s)\n", bad, C[0], R[0], C[0] == R[0] ? "PRESERVED" : "FLUSHED?");
  return 0;
}
