// Device-side format helpers shared by the gfx950 kernels (HIP, wave64).
//
// Formats (reference kernels/src/reorder.cu:17-31,98-104):
//   e2m1   4-bit code  s e1 e0 m        magnitudes {0,.5,1,1.5,2,3,4,6}, RNE ties-to-even-code, saturating
//   ue4m3  8-bit       e4m3 magnitude    bias 7, subnormals k*2^-9, max 448, RNE
//   bf16   RNE from fp32
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace arcq {

constexpr float kFp4Max = 6.0f;             // reorder.cu:17
constexpr float kFp8Max = 448.0f;           // reorder.cu:18
constexpr float kScaleEps = 0.001953125f;   // reorder.cu:19 (2^-9)

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

// fp32 -> bf16 bits, round-to-nearest-even (inputs are finite on this path; NaN is kept quiet).
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  uint32_t r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  return ((u & 0x7fffffffu) > 0x7f800000u) ? ((u >> 16) | 0x40u) : r;
}
// x rounded to bf16 and widened again, through gfx950's v_cvt_pk_bf16_f32 (what the float -> __bf16 cast compiles to):
// two instructions where the integer formulation above plus the shift back take eight.  The same function for all 2^32 bit
// patterns, NaN stays NaN (tools/probe_bf16.hip, checked exhaustively on MI355X).  Used by the quantisers, which round
// EVERY element (RMSNorm, dynamic scale, residual); the GEMM epilogues keep the integer form (a few conversions per
// thread, and their register allocation is tuned around it).
__device__ __forceinline__ float round_to_bf16(float f) { return (float)(__bf16)f; }

// ue4m3 byte -> fp32.  (bits << 20) is the value scaled by 2^-120 (subnormals included); one multiply undoes it.
__device__ __forceinline__ float ue4m3_to_f32(uint32_t b) { return __uint_as_float((b & 0x7fu) << 20) * 0x1p120f; }

// fp32 in [2^-9, 448] -> ue4m3 byte, RNE (Float2Ue4m3, reorder.cu:100).
__device__ __forceinline__ uint32_t f32_to_ue4m3(float s) {
  uint32_t u = __float_as_uint(s);
  uint32_t n = ((u + 0x7ffffu + ((u >> 20) & 1u)) >> 20) - ((127u - 7u) << 3);  // normal grid
  uint32_t d = (uint32_t)__builtin_rintf(s * 512.0f);                              // subnormal grid, step 2^-9
  return s < 0.015625f ? d : (n > 0x7eu ? 0x7eu : n);
}

// fp32 (already clamped to [-6,6]) -> e2m1 code, RNE, sign of zero kept (Float2E2m1, reorder.cu:98).
__device__ __forceinline__ uint32_t f32_to_e2m1(float r) {
  uint32_t u = __float_as_uint(r);
  uint32_t sign = (u >> 28) & 8u;
  uint32_t a = u & 0x7fffffffu;
  float af = __uint_as_float(a);
  // >= 1: keep one mantissa bit (RNE on the fp32 bits); < 1: two thresholds on the 0.5 grid
  uint32_t hi = ((a + 0x1fffffu + ((a >> 22) & 1u)) >> 22) - 252u;
  uint32_t lo = (af > 0.25f ? 1u : 0u) + (af >= 0.75f ? 1u : 0u);
  uint32_t c = af < 1.0f ? lo : hi;
  return sign | (c > 7u ? 7u : c);
}

__device__ __forceinline__ float e2m1_to_f32(uint32_t c) {
  // magnitude m = c & 7: (m << 22) read as fp32 is the value * 2^-126 (m = 1 is the subnormal 0.5)
  float v = __uint_as_float((c & 7u) << 22) * 0x1p126f;
  return (c & 8u) ? -v : v;
}

// Scale-factor byte offset (CUTLASS Sm1xx block-scaled atom, reorder.cuh:118-123 / reorder.cu:139-143).
__host__ __device__ __forceinline__ int64_t sf_offset(int64_t r, int64_t p, int64_t K) {
  return ((r >> 7) * (K >> 6) + (p >> 2)) * 512 + (r & 31) * 16 + ((r >> 5) & 3) * 4 + (p & 3);
}

// act = silu(gate) * up on bf16 bits, rounded where torch's two elementwise kernels round (model/qLlamaLayer.py:
// `self.act_fn(gate) * up`): silu in fp32 as x / (1 + exp(-x)) -> bf16 (ActivationSiluKernel), then the fp32 product
// -> bf16 (MulFunctor).  expf / the division are the same ocml / IEEE operations torch's HIP build uses.
__device__ __forceinline__ uint32_t silu_mul_bf16(uint32_t g_bits, uint32_t u_bits) {
  const float g = bf16_bits_to_f32(g_bits);
  const float s = g / (1.0f + expf(-g));
  const float sb = bf16_bits_to_f32(f32_to_bf16_bits(s));
  return f32_to_bf16_bits(sb * bf16_bits_to_f32(u_bits));
}

// ---- cross-lane helpers on DPP (no LDS crossbar: __shfl_down is a ds_bpermute, ~100 cycles of latency on a pipe the gathers use)
// lane i <- lane i + N inside its row of 16 lanes, 0 beyond the row
template <int N>
__device__ __forceinline__ uint32_t dpp_row_shl_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}
// max over the 64 lanes of a wave, returned in every lane (a scalar): four DPP levels inside the rows, then the four row heads
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, dpp_row_shl_u32<8>(v));
  v = max(v, dpp_row_shl_u32<4>(v));
  v = max(v, dpp_row_shl_u32<2>(v));
  v = max(v, dpp_row_shl_u32<1>(v));
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}

}  // namespace arcq
