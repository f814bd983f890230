// Shared device pieces of the two NVFP4 GEMM kernels (gfx950).
//
// Why fp16 MFMA: an NVFP4 operand element is e2m1 (2 significant bits) times a ue4m3 block scale
// (4 significant bits): the product needs up to 6 significant bits and spans [2^-10, 2688], which
// fp16 holds EXACTLY and in its normal range.  gfx950's block-scaled fp4 MFMA
// (v_mfma_scale_f32_16x16x128_f8f6f4) applies one power-of-two (E8M0) scale per 32 elements, so it
// cannot express a 3-mantissa-bit scale per 16; the exact contraction therefore runs on
// v_mfma_f32_16x16x32_f16 after an in-register dequantisation:
//     v_cvt_scalef32_pk_f16_fp4   (2 codes -> 2 fp16, probed on MI355X: the scale operand only
//                                  contributes its exponent; it is fed 2^8, see sf_pair)
//     v_pk_mul_f16                (x the ue4m3 scale re-read as fp16 bits; exact)
// Products of two such fp16 values are exact in fp32, accumulation is the MFMA's fp32 chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "arcq_device.hpp"

// Weight streams of the decode kernels: every byte is read ONCE per launch by ONE CU, so it is loaded with the non-temporal
// policy (global_load ... nt).  Measured on MI355X against default-policy loads (tools/decode_stream_bench.py, HBM-cold): gate|up
// 20.5 -> 19.2 us, N=14336 9.9 -> 9.3, down 13.3 -> 12.7, the 28-layer decode step 1758 -> 1842 tok/s.  -DARCQ_PLAIN_WEIGHT_LOADS
// (make plain) builds the A-B library.  Never used for operands other workgroups re-read (the tile GEMM's panels).
#ifdef ARCQ_PLAIN_WEIGHT_LOADS
#define ARCQ_WLOAD(ptr) (*(ptr))
#else
#define ARCQ_WLOAD(ptr) __builtin_nontemporal_load(ptr)
#endif

namespace arcq {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

union Frag8 {        // 8 fp16 = one MFMA 16x16x32 operand fragment = 16 bytes
  f16x8 v;
  f16x2 p[4];
  uint4 u;
};

// ue4m3 byte -> fp16 pair (s', s') with s' = scale * 2^-8.  The e4m3 bit pattern shifted into the fp16 fields
// (b << 7: exponent field = E, top mantissa bits = m) reads as (1 + m/8) * 2^(E-15) for E >= 1 and as the fp16
// SUBNORMAL m * 2^-17 for E == 0 -- in both cases exactly scale * 2^-8 -- so the conversion is one shift and one
// pack, no float math.  The missing 2^8 is supplied by the conversion below (its scale operand contributes
// exactly its exponent).  fp16 denormals are preserved by v_pk_mul_f16 in the default kernel mode
// (amdhsa_float_denorm_mode_16_64 = 3) and every product lands in the normal range again.
__device__ __forceinline__ f16x2 sf_pair(uint32_t byte) {
  const uint32_t h = (byte & 0x7fu) << 7;
  const uint32_t w = h | (h << 16);
  f16x2 r;
  __builtin_memcpy(&r, &w, 4);
  return r;
}

// The same for the scale byte at bit offset `off` of a packed word, in two instructions: v_bfe_u32 takes the seven
// magnitude bits, one 24-bit multiply by (2^7 + 2^23) shifts them into both fp16 halves (x < 128, so the copies
// cannot overlap).
__device__ __forceinline__ f16x2 sf_pair_at(uint32_t word, uint32_t off) {
  const uint32_t w = __builtin_amdgcn_ubfe(word, off, 7u) * 0x00800080u;   // both factors < 2^24: compiles to v_mul_u32_u24
  f16x2 r;
  __builtin_memcpy(&r, &w, 4);
  return r;
}

// 8 e2m1 codes (one dword, low nibble first) x scale -> 8 fp16:  (code * 2^8) * (scale * 2^-8), both factors and
// the product exact in fp16.
__device__ __forceinline__ Frag8 dequant8(uint32_t codes, f16x2 s2) {
  Frag8 f;
#ifdef ARCQ_EXPERIMENT_NO_SCALE_MUL
  // TIMING EXPERIMENT ONLY (tools/scripts/build_variant_lib.sh; results are WRONG): the block-scale multiply dropped.  +12 % on the
  // tile GEMM -- but NOT because of the instruction: the operands then carry one significant mantissa bit, the matrix pipe draws less
  // power and the chip clocks higher.  The instruction-count experiment is ARCQ_EXPERIMENT_A_RAW (gemm_tile_common.hpp): half of
  // all dequantisation instructions removed at unchanged operand density = +3 % (profiles/r03_tile_dequant_upper_bound_ab.jsonl)
  f.p[0] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes, 256.0f, 0);
  f.p[1] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes, 256.0f, 1);
  f.p[2] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes, 256.0f, 2);
  f.p[3] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes, 256.0f, 3);
  (void)s2;
  return f;
#endif
  f.p[0] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes, 256.0f, 0) * s2;
  f.p[1] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes, 256.0f, 1) * s2;
  f.p[2] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes, 256.0f, 2) * s2;
  f.p[3] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes, 256.0f, 3) * s2;
  return f;
}

// Byte offset of the 4 scale bytes (one K-atom of 64 elements) of row r: they are contiguous and
// 4-byte aligned in the swizzled layout (arcq.h).
__device__ __forceinline__ int64_t sf_atom_offset(int r, int atom, int atoms_k) {
  return ((int64_t)(r >> 7) * atoms_k + atom) * 512 + (r & 31) * 16 + ((r >> 5) & 3) * 4;
}

// alpha*acc (+bias) -> bf16 / fp32 store helpers
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) { return f32_to_bf16_bits(a) | (f32_to_bf16_bits(b) << 16); }

// Epilogue helpers shared by the GEMM kernels; P is any parameter block with D, N, bias, residual, out_dtype.
// Idx = uint32_t inside the decode kernel (M <= 16: element offsets fit 32 bits, so the address is an SGPR base
// plus one VGPR offset instead of a 64-bit VGPR pair per lane -- the kernel is register bound), size_t elsewhere.
template <typename Idx, typename P>
__device__ __forceinline__ void store_out4(const P& p, int m, int n, const float (&d)[4]) {
  // d[r] is the finished value of D[m, n + r]
  if (p.out_dtype == ARCQ_OUT_F32) {
    float* o = reinterpret_cast<float*>(p.D) + ((Idx)m * (Idx)p.N + (Idx)n);
    if (n + 3 < p.N && (p.N & 3) == 0) {
      *reinterpret_cast<float4*>(o) = make_float4(d[0], d[1], d[2], d[3]);
    } else {
      for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = d[r];
    }
  } else {
    uint16_t* o = reinterpret_cast<uint16_t*>(p.D) + ((Idx)m * (Idx)p.N + (Idx)n);
    if (n + 3 < p.N && (p.N & 3) == 0) {
      *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]));
    } else {
      for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = (uint16_t)f32_to_bf16_bits(d[r]);
    }
  }
}

// finish4 with the four bias / residual values of (m, n .. n + 3) already in registers (bf16 bits, two per dword; N % 4 == 0):
// the decode kernels fetch them while their last weights stream instead of after the final barrier
template <typename Idx, typename P>
__device__ __forceinline__ void finish4_pre(const P& p, float alpha, int m, int n, const float (&acc)[4], uint2 bias_bits, uint2 res_bits) {
  const uint32_t bw[2] = {bias_bits.x, bias_bits.y}, rw[2] = {res_bits.x, res_bits.y};
  float d[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    d[r] = alpha * acc[r];
    if (p.bias) {
      const float b = bf16_bits_to_f32((bw[r >> 1] >> (16 * (r & 1))) & 0xffffu);
      d[r] = (p.out_dtype == ARCQ_OUT_F32 ? d[r] : bf16_bits_to_f32(f32_to_bf16_bits(d[r]))) + b;
    }
    if (p.residual) {
      const float res = bf16_bits_to_f32((rw[r >> 1] >> (16 * (r & 1))) & 0xffffu);
      d[r] = (p.out_dtype == ARCQ_OUT_F32 ? d[r] : bf16_bits_to_f32(f32_to_bf16_bits(d[r]))) + res;
    }
  }
  store_out4<Idx>(p, m, n, d);
}

template <typename Idx, typename P>
__device__ __forceinline__ void finish4(const P& p, float alpha, int m, int n, const float (&acc)[4]) {
  // the common case -- N % 4 == 0, operands 8-byte aligned: ONE 8-byte load per operand instead of four element loads that hipcc awaits
  // one by one (on the tile GEMM's prefill shapes that cost + 19 % with a bias and + 32 % with a residual)
  if ((p.N & 3) == 0 && (p.bias || p.residual) && ((reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.residual)) & 7) == 0) {
    uint2 b2 = make_uint2(0, 0), r2 = make_uint2(0, 0);
    if (p.bias) b2 = *reinterpret_cast<const uint2*>(p.bias + n);
    if (p.residual) r2 = *reinterpret_cast<const uint2*>(p.residual + ((Idx)m * (Idx)p.N + (Idx)n));
    finish4_pre<Idx>(p, alpha, m, n, acc, b2, r2);
    return;
  }
  float d[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    d[r] = alpha * acc[r];                                          // epilogue in fp32 (nvfp4.cu:117-121)
    // bf16 output: every fused operand is added the way the reference's separate torch op adds it -- to the ROUNDED bf16
    // result, rounding again: `y = matmul(...); y = y + bias` (qLinearLayer.py:74-76), then `x + y` in the decoder layer.
    // fp32 output (row-parallel partials, tests) adds everything in fp32, one result.
    if (p.bias && n + r < p.N) {
      const float b = bf16_bits_to_f32(p.bias[n + r]);
      d[r] = (p.out_dtype == ARCQ_OUT_F32 ? d[r] : bf16_bits_to_f32(f32_to_bf16_bits(d[r]))) + b;
    }
    if (p.residual && n + r < p.N) {
      const float res = bf16_bits_to_f32(p.residual[(Idx)m * (Idx)p.N + (Idx)(n + r)]);
      d[r] = (p.out_dtype == ARCQ_OUT_F32 ? d[r] : bf16_bits_to_f32(f32_to_bf16_bits(d[r]))) + res;
    }
  }
  store_out4<Idx>(p, m, n, d);
}

}  // namespace arcq
