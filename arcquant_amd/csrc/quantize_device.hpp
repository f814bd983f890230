// Per-group NVFP4 quantisation of the quantiser kernels (quantize.hip), kept apart from the row/launch logic: ONE
// statement of the arithmetic for every kernel that has to produce the same bytes (a GEMM prologue that quantised its
// own activations with it was bit-identical but slower).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "arcq_device.hpp"
#include "arcq_internal.hpp"

namespace arcq {

struct GroupQ {
  uint2 packed;    // 16 e2m1 codes, low nibble = even element (reorder.cu:28-31)
  uint32_t s8;     // ue4m3 scale byte
  float s_round;   // decoded ue4m3 scale
  float s_raw;     // clamp(amax/6, 2^-9, 448) before rounding
};

// amax -> scale -> codes for one 16-element group held in registers (reorder.cu:119-164).
// When kResid, v[] is overwritten with bf16(v - q*S), S = rounded scale (G16) or raw scale (G32).
template <bool kResid, int kVariant>
__device__ __forceinline__ GroupQ quantize_group(float (&v)[16]) {
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
  float s = amax / kFp4Max;                      // IEEE division (hipcc default: correctly rounded)
  s = fminf(fmaxf(s, kScaleEps), kFp8Max);
  GroupQ g;
  g.s_raw = s;
  // gfx950 conversion instructions, probed (tools/probe_gfx950.hip) and byte-checked against the oracle by the GPU
  // tests: v_cvt_pk_fp8_f32 is OCP e4m3 RNE incl. subnormals (s is already clamped to [2^-9, 448]);
  // v_cvt_scalef32_pk_fp4_f32 is RNE, ties to the even code, saturating at +-6 (== clamp then convert,
  // reorder.cu:153), sign of zero kept; v_cvt_scalef32_pk_f32_fp4 decodes two codes.  Scale operands are 1.0.
  g.s8 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(s, s, 0, false) & 0xffu;
  g.s_round = ue4m3_to_f32(g.s8);
  const float r = 1.0f / g.s_round;              // == (float)(1.0/(double)s8), tests/test_oracle_formats.py
  const float S = (kVariant == ARCQ_VARIANT_G16) ? g.s_round : g.s_raw;   // reorder.cu:157 vs :474
  uint32_t lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(lo, v[0] * r, v[1] * r, 1.0f, 0);
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(lo, v[2] * r, v[3] * r, 1.0f, 1);
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(lo, v[4] * r, v[5] * r, 1.0f, 2);
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(lo, v[6] * r, v[7] * r, 1.0f, 3);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(hi, v[8] * r, v[9] * r, 1.0f, 0);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(hi, v[10] * r, v[11] * r, 1.0f, 1);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(hi, v[12] * r, v[13] * r, 1.0f, 2);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(hi, v[14] * r, v[15] * r, 1.0f, 3);
  if (kResid) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const uint32_t w = b < 4 ? lo : hi;
      f32x2 q2;
      switch (b & 3) {
        case 0: q2 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 0); break;
        case 1: q2 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 1); break;
        case 2: q2 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 2); break;
        default: q2 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 3); break;
      }
      const float d0 = __builtin_fmaf(-q2.x, S, v[2 * b]);       // fused, oracle assumption A2
      const float d1 = __builtin_fmaf(-q2.y, S, v[2 * b + 1]);
      v[2 * b] = round_to_bf16(d0);
      v[2 * b + 1] = round_to_bf16(d1);
    }
  }
  g.packed = make_uint2(lo, hi);
  return g;
}

// max |x| over eight bf16 values (as bit patterns: |bf16| ordering == ordering of the low 15 bits)
__device__ __forceinline__ uint32_t absmax_bits_chunk(const uint4 d, uint32_t m) {
  const uint32_t w4[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    m = max(m, w4[j] & 0x7fffu);
    m = max(m, (w4[j] >> 16) & 0x7fffu);
  }
  return m;
}


// LDS copy of a row (and of the norm weight): ONE PAD DWORD per 16-element group.  Unpadded, lane t of a wave gathers
// element idx[16 t + j]; with reorder_index = identity (what the reference's own latency benchmark uses) that is byte
// 32 t + 2 j -- the same bank for every 8th lane, an 8-way conflict on each of the 16 reads (measured: identity was SLOWER
// than a random permutation, 47.2 vs 35.8 us at 8192^2).  With 36-byte groups the 64 lanes hit 64 different banks; a random
// permutation is unaffected.  The pad is applied to a PAIR of int16 indices at once: e + 2 (e >> 4) <= 36861 fits 16 bits.
__device__ __forceinline__ uint32_t lds_pad_pair(uint32_t w) { return w + (((w >> 4) & 0x0fff0fffu) << 1); }
__device__ __forceinline__ void lds_store_chunk(uint16_t* row, int c, uint4 d) {       // chunk = 8 elements = half a group
  uint32_t* p = reinterpret_cast<uint32_t*>(row) + 4 * c + (c >> 1);
  p[0] = d.x; p[1] = d.y; p[2] = d.z; p[3] = d.w;
}
__host__ __device__ constexpr size_t lds_row_bytes(size_t KQ) { return KQ * 2 + KQ / 4; }


// x / scale for the per-tensor dynamic scale, without ten instructions of IEEE division per element: q = x * r corrected by
// two FMAs (Markstein) and the sign of x restored (-0 / s = -0).  With r = RN(1 / scale) this IS the correctly rounded
// quotient; checked EXHAUSTIVELY on MI355X for every finite bf16 x with |x| <= 4096 scale and every bf16 scale in
// 2^-100 .. 2^100 (tools/probe_div.hip: 0 mismatches after the bf16 rounding; |x / scale| <= 2688 holds by construction).
// Scales outside that range take the IEEE division.  `scale` is the bf16-ROUNDED scale torch divides by.
struct DynDiv {
  float scale, rcp;
  bool fast;
  __device__ __forceinline__ DynDiv(float s, bool enabled) : scale(s), rcp(1.0f / s), fast(enabled && s >= 0x1p-100f && s <= 0x1p100f) {}
  // `fast` is uniform over the launch: callers test it ONCE and instantiate their element loops for either value (as a branch per
  // element it costs a scalar branch and a full wait on outstanding loads for each of the 16 values of a group)
  template <bool kFast>
  __device__ __forceinline__ float div(float x) const {
    if constexpr (kFast) {
      float q = x * rcp;
      const float e = __builtin_fmaf(-q, scale, x);
      q = __builtin_fmaf(e, rcp, q);
      return __builtin_copysignf(q, x);
    } else {
      return x / scale;
    }
  }
};

}  // namespace arcq
