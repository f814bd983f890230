// ARC-NVFP4 GEMM for decode BATCHES (16 < M <= 64) over the REPACKED weight: the weight is still read ONCE.
//
// The M <= 16 kernels (gemm_rowblock.hip, gemm_stream.hip) keep an fp16 image of the activations in LDS (M x K x 2 bytes): at
// M = 32, K = 4160 that would be 279 KB.  Before this kernel, M = 17 fell through to the LDS-tiled GEMM with split-K
// (gemm_tile.hip, 32 x 256 / 64 x 256 tiles + a finish pass): 16.9 us at M = 32, N = K = 4096 against 7.9 us at M = 16 -- a cliff
// in the reference's own M sweep (kernels/bench.py:8-49 times agemm.matmul at N = K = 4096 for M = 8 ... 4096).  Here the
// activations stay PACKED in LDS (M x K x 9/16 bytes: 64 x 4160 fits), in their natural layout -- a lane's four MFMA operands of
// a 128-element tile are the 16 contiguous bytes [16 q, 16 q + 16) of the token's row, its two scale bytes the natural bytes
// 2 q, 2 q + 1 of the tile -- and are dequantised per MFMA next to the weights (exact: gemm_common.hpp).  A weight unit
// (16 rows x 256 K, 2 KB) is fetched once and contracted against ceil(M / 16) token tiles; the extra work per unit is
// ceil(M / 16) x (2 LDS reads + 2 x 20 conversion instructions + 8 MFMA), which two 8-wave workgroups per CU still hide behind the
// weight stream at M = 32.  Work split, register ring and slice reduction are those of gemm_rowblock.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "arcq_internal.hpp"
#include "gemm_common.hpp"
#include "rowblock_split.hpp"

namespace arcq {

struct RowmidParams {
  const uint8_t* A;       // activations, reference layout [M, K/2]
  const uint8_t* SFA;     // ... swizzled scales
  const uint8_t* RW;      // repacked weight tiles
  const uint8_t* RSF;     // repacked weight scales
  void* D;
  const float* alpha_dev;
  const uint16_t* bias;
  const uint16_t* residual;
  int M, N, K;
  float alpha_host;
  int out_dtype;
  int pairs;              // K_padded / 256: tile pairs per row block
  int row_blocks;         // ceil(N / 16)
  int slices;             // waves that share one row block (1, 2, 4 or 8)
  int a_stride;           // bytes per token row of the packed codes in LDS
  int s_stride;           // bytes per token row of the scale bytes in LDS
  int sf_off;             // LDS byte offset of the scale rows
};

typedef uint32_t rm_u32x4 __attribute__((ext_vector_type(4)));
struct RowmidRegs {       // one tile pair of this lane: 2 x 16 bytes of codes, 4 scale bytes
  rm_u32x4 b0, b1;
  uint32_t s;
};

constexpr int kRmWaves = 8, kRmThreads = kRmWaves * 64;
#ifndef ARCQ_ROWMID_WPS2
#define ARCQ_ROWMID_WPS2 4      // waves per SIMD asked of the two-token-tile kernels (4 = two workgroups per CU, 128 registers; A-B: 2)
#endif

template <int kTok>       // token tiles of 16: 2, 3 or 4
// (second launch bound = waves per SIMD: 4 = two 8-wave workgroups per CU, which the LDS of two token tiles allows; above, one)
__global__ __launch_bounds__(kRmThreads, kTok <= 2 ? ARCQ_ROWMID_WPS2 : 2) void gemm_rowmid_kernel(RowmidParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const a_lds = smem;                                         // [M][a_stride] packed codes, K padded with zeros
  unsigned char* const s_lds = smem + p.sf_off;                              // [M][s_stride] ue4m3 bytes, natural order
  float* const red = reinterpret_cast<float*>(smem);    // [8 waves][kTok][64][4] when slices > 1: REUSES the codes after the K loop

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, rl = lane & 15;
  const int bpw = kRmWaves / p.slices;                    // row blocks per workgroup
  const int rb = blockIdx.x * bpw + wave / p.slices;
  const int slice = wave % p.slices;
  const bool active = rb < p.row_blocks;
  int pr_begin, pr_count;
  rowblock_slice_range(p.pairs, p.slices, slice, &pr_begin, &pr_count);
  const int npairs = active ? pr_count : 0;
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);

  // ---- the weight stream first this time: the activations (M x K x 9/16 bytes, the same for every workgroup: L2 hits after the
  //      first) are many loads per thread and are consumed as they arrive; three tile pairs per lane stay in flight behind them
  const int rbc = active ? rb : 0;
  const int pr_load = rowblock_load_base(pr_begin, npairs);   // never past the row block's own pairs
  const uint8_t* wp = p.RW + ((size_t)rbc * p.pairs + pr_load) * 2048 + lane * 16;
  const uint8_t* sp = p.RSF + ((size_t)rbc * p.pairs + pr_load) * 256 + lane * 4;
  const int last = npairs > 0 ? npairs - 1 : 0;
  int issued = 0;
  auto issue = [&](RowmidRegs& r) __attribute__((always_inline)) {       // unpredicated; the cursor stops at the last pair
    const int i = min(issued, last);
    r.b0 = ARCQ_WLOAD(reinterpret_cast<const rm_u32x4*>(wp + (size_t)i * 2048));
    r.b1 = ARCQ_WLOAD(reinterpret_cast<const rm_u32x4*>(wp + (size_t)i * 2048 + 1024));
    r.s = ARCQ_WLOAD(reinterpret_cast<const uint32_t*>(sp + (size_t)i * 256));
    ++issued;
  };

  // ---- activations -> LDS, natural layout.  16-byte chunks of codes (K padded to 256 with zero codes), one dword of scales per atom
  {
    const int cpr = p.pairs * 8, real = p.K >> 5;            // 16-byte chunks per row: padded / real
    const int total = p.M * cpr;
#pragma unroll 4
    for (int u = tid; u < total; u += kRmThreads) {
      const int m = u / cpr, c = u - m * cpr;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c < real) v = *reinterpret_cast<const uint4*>(p.A + (size_t)m * (p.K >> 1) + (size_t)c * 16);
      *reinterpret_cast<uint4*>(a_lds + (size_t)m * p.a_stride + (size_t)c * 16) = v;
    }
    const int apr = p.pairs * 4, areal = p.K >> 6;           // scale atoms (4 bytes, 64 K) per row: padded / real
    const int stotal = p.M * apr;
#pragma unroll 4
    for (int u = tid; u < stotal; u += kRmThreads) {
      const int m = u / apr, a = u - m * apr;
      uint32_t v = 0;
      if (a < areal) v = *reinterpret_cast<const uint32_t*>(p.SFA + sf_atom_offset(m, a, areal));
      *reinterpret_cast<uint32_t*>(s_lds + (size_t)m * p.s_stride + (size_t)a * 4) = v;
    }
  }
  RowmidRegs r0, r1, r2;
  issue(r0);
  issue(r1);
  issue(r2);
  __syncthreads();

  // ---- K loop: no barrier; per tile one weight dequantisation, kTok x (LDS read, activation dequantisation, 4 MFMA)
  f32x4 acc[kTok];
#pragma unroll
  for (int t = 0; t < kTok; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* a_tok[kTok];
  const unsigned char* s_tok[kTok];
#pragma unroll
  for (int t = 0; t < kTok; ++t) {
    const int m = min(t * 16 + rl, p.M - 1);                 // tokens >= M: any row (never stored)
    a_tok[t] = a_lds + (size_t)m * p.a_stride + (size_t)pr_begin * 128 + q * 16;
    s_tok[t] = s_lds + (size_t)m * p.s_stride + (size_t)pr_begin * 16 + q * 2;
  }
  int done = 0;
  auto tile = [&](rm_u32x4 b, uint32_t s16, int off_a, int off_s) __attribute__((always_inline)) {
    const f16x2 s0 = sf_pair_at(s16, 0), s1 = sf_pair_at(s16, 8);
    const Frag8 b0 = dequant8(b.x, s0), b1 = dequant8(b.y, s0), b2 = dequant8(b.z, s1), b3 = dequant8(b.w, s1);
#pragma unroll
    for (int t = 0; t < kTok; ++t) {
      const uint4 ac = *reinterpret_cast<const uint4*>(a_tok[t] + off_a);
      const uint32_t as16 = *reinterpret_cast<const uint16_t*>(s_tok[t] + off_s);
      const f16x2 t0 = sf_pair_at(as16, 0), t1 = sf_pair_at(as16, 8);
      const Frag8 a0 = dequant8(ac.x, t0), a1 = dequant8(ac.y, t0), a2 = dequant8(ac.z, t1), a3 = dequant8(ac.w, t1);
      // weights are the MFMA A operand (rows = weight rows), activations the B operand (columns = tokens)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc[t], 0, 0, 0);
    }
  };
  auto step = [&](RowmidRegs& r) __attribute__((always_inline)) {
    tile(r.b0, r.s, done * 128, done * 16);
    tile(r.b1, r.s >> 16, done * 128 + 64, done * 16 + 8);
    issue(r);                                               // refill: three pairs ahead
    __builtin_amdgcn_sched_barrier(0);                      // hipcc otherwise sinks all refills to the end of the unrolled body
    ++done;
  };
#pragma unroll 1
  while (done + 3 <= npairs) {
    step(r0);
    step(r1);
    step(r2);
  }
  if (done < npairs) step(r0);
  if (done < npairs) step(r1);

  // ---- lane holds C[token = 16 t + rl][row = 16 rb + 4 q + e]; add the K slices of a row block through LDS
  if (p.slices > 1) {
    __syncthreads();                                        // every wave is done reading the activations
#pragma unroll
    for (int t = 0; t < kTok; ++t)
      *reinterpret_cast<float4*>(red + ((wave * kTok + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
    if (slice != 0) return;
    for (int s2 = 1; s2 < p.slices; ++s2) {
#pragma unroll
      for (int t = 0; t < kTok; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(red + (((wave + s2) * kTok + t) * 64 + lane) * 4);
        acc[t][0] += v.x; acc[t][1] += v.y; acc[t][2] += v.z; acc[t][3] += v.w;
      }
    }
  }
  const int n0 = rb * 16 + 4 * q;
  if (active && n0 < p.N) {
#pragma unroll
    for (int t = 0; t < kTok; ++t) {
      const int m = t * 16 + rl;
      if (m < p.M) {
        const float sum[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
        finish4<uint32_t>(p, alpha, m, n0, sum);
      }
    }
  }
}

// ---- the same contraction with NO activations in LDS ("rowtok"): every lane fetches the MFMA B operands of its token tiles itself,
// per tile pair, from the packed activations in global memory (a few hundred KB that every workgroup reads: L2 / L1 hits) -- the
// no-image kernel of gemm_rowblock.hip generalised to kTok token tiles.  No LDS fill (the 74 KB every workgroup copies at M = 32),
// no barrier before the K loop, and no LDS capacity limit: it reaches M = 128.  Costs kTok x 4 loads per tile pair beside the 3
// weight loads; the weight ring stays three pairs deep, the activation loads of a pair are issued when its step starts (their
// latency is hidden by the other waves of the CU, not by a ring: kTok x 10 registers per pair).
template <int kTok>
__global__ __launch_bounds__(kRmThreads, kTok <= 2 ? 4 : 2) void gemm_rowtok_kernel(RowmidParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const red = reinterpret_cast<float*>(smem);    // [8 waves][kTok][64][4] when slices > 1

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, rl = lane & 15;
  const int bpw = kRmWaves / p.slices;
  const int rb = blockIdx.x * bpw + wave / p.slices;
  const int slice = wave % p.slices;
  const bool active = rb < p.row_blocks;
  int pr_begin, pr_count;
  rowblock_slice_range(p.pairs, p.slices, slice, &pr_begin, &pr_count);
  const int npairs = active ? pr_count : 0;
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);

  const int rbc = active ? rb : 0;
  const int pr_load = rowblock_load_base(pr_begin, npairs);
  const uint8_t* wp = p.RW + ((size_t)rbc * p.pairs + pr_load) * 2048 + lane * 16;
  const uint8_t* sp = p.RSF + ((size_t)rbc * p.pairs + pr_load) * 256 + lane * 4;
  const int last = npairs > 0 ? npairs - 1 : 0;
  int issued = 0;
  auto issue = [&](RowmidRegs& r) __attribute__((always_inline)) {
    const int i = min(issued, last);
    r.b0 = ARCQ_WLOAD(reinterpret_cast<const rm_u32x4*>(wp + (size_t)i * 2048));
    r.b1 = ARCQ_WLOAD(reinterpret_cast<const rm_u32x4*>(wp + (size_t)i * 2048 + 1024));
    r.s = ARCQ_WLOAD(reinterpret_cast<const uint32_t*>(sp + (size_t)i * 256));
    ++issued;
  };
  // activation rows of this lane (tokens >= M: the last row, never stored) and the clamps for tiles / atoms past K (they meet zero
  // weight scales; gemm_rowblock.hip, direct kernel)
  const int atoms_k = p.K >> 6, tiles_k = p.K >> 7, halfk = (p.K & 64) ? 1 : 0;
  const uint8_t* arow[kTok];
  const uint8_t* srow[kTok];
#pragma unroll
  for (int t = 0; t < kTok; ++t) {
    const int tok = min(t * 16 + rl, p.M - 1);
    arow[t] = p.A + (size_t)tok * (p.K >> 1) + q * 16;
    srow[t] = p.SFA + sf_atom_offset(tok, 0, atoms_k) + (q & 1) * 2;
  }
  auto a_off = [&](int t) __attribute__((always_inline)) -> size_t {
    const int tc = min(t, tiles_k - 1 + halfk);
    size_t off = (size_t)tc * 64;
    if (halfk && tc == tiles_k && q >= 2) off -= 32;         // second half of a trailing half tile: quarters q - 2 (zero weight scales there)
    return off;
  };
  auto s_off = [&](int t) __attribute__((always_inline)) -> size_t { return (size_t)min(2 * t + (q >> 1), atoms_k - 1) * 512; };

  RowmidRegs r0, r1, r2;
  issue(r0);
  issue(r1);
  issue(r2);

  f32x4 acc[kTok];
#pragma unroll
  for (int t = 0; t < kTok; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int done = 0;
  auto step = [&](RowmidRegs& r) __attribute__((always_inline)) {
    const int t0 = 2 * (pr_begin + done), t1 = t0 + 1;
    const size_t oa0 = a_off(t0), oa1 = a_off(t1), os0 = s_off(t0), os1 = s_off(t1);
    rm_u32x4 xa0[kTok], xa1[kTok];
    uint32_t xs0[kTok], xs1[kTok];
#pragma unroll
    for (int t = 0; t < kTok; ++t) {                         // every activation load of the pair first ...
      xa0[t] = *reinterpret_cast<const rm_u32x4*>(arow[t] + oa0);
      xa1[t] = *reinterpret_cast<const rm_u32x4*>(arow[t] + oa1);
      xs0[t] = *reinterpret_cast<const uint16_t*>(srow[t] + os0);
      xs1[t] = *reinterpret_cast<const uint16_t*>(srow[t] + os1);
    }
    {                                                        // ... then the pair's two tiles against every token tile
      const f16x2 s0 = sf_pair_at(r.s, 0), s1 = sf_pair_at(r.s, 8);
      const Frag8 b0 = dequant8(r.b0.x, s0), b1 = dequant8(r.b0.y, s0), b2 = dequant8(r.b0.z, s1), b3 = dequant8(r.b0.w, s1);
#pragma unroll
      for (int t = 0; t < kTok; ++t) {
        const f16x2 u0 = sf_pair_at(xs0[t], 0), u1 = sf_pair_at(xs0[t], 8);
        const Frag8 a0 = dequant8(xa0[t].x, u0), a1 = dequant8(xa0[t].y, u0), a2 = dequant8(xa0[t].z, u1), a3 = dequant8(xa0[t].w, u1);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc[t], 0, 0, 0);
      }
    }
    {
      const f16x2 s0 = sf_pair_at(r.s >> 16, 0), s1 = sf_pair_at(r.s >> 16, 8);
      const Frag8 b0 = dequant8(r.b1.x, s0), b1 = dequant8(r.b1.y, s0), b2 = dequant8(r.b1.z, s1), b3 = dequant8(r.b1.w, s1);
#pragma unroll
      for (int t = 0; t < kTok; ++t) {
        const f16x2 u0 = sf_pair_at(xs1[t], 0), u1 = sf_pair_at(xs1[t], 8);
        const Frag8 a0 = dequant8(xa1[t].x, u0), a1 = dequant8(xa1[t].y, u0), a2 = dequant8(xa1[t].z, u1), a3 = dequant8(xa1[t].w, u1);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc[t], 0, 0, 0);
      }
    }
    issue(r);                                               // refill: three pairs ahead
    __builtin_amdgcn_sched_barrier(0);
    ++done;
  };
#pragma unroll 1
  while (done + 3 <= npairs) {
    step(r0);
    step(r1);
    step(r2);
  }
  if (done < npairs) step(r0);
  if (done < npairs) step(r1);

  if (p.slices > 1) {
#pragma unroll
    for (int t = 0; t < kTok; ++t)
      *reinterpret_cast<float4*>(red + ((wave * kTok + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
    if (slice != 0) return;
    for (int s2 = 1; s2 < p.slices; ++s2) {
#pragma unroll
      for (int t = 0; t < kTok; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(red + (((wave + s2) * kTok + t) * 64 + lane) * 4);
        acc[t][0] += v.x; acc[t][1] += v.y; acc[t][2] += v.z; acc[t][3] += v.w;
      }
    }
  }
  const int n0 = rb * 16 + 4 * q;
  if (active && n0 < p.N) {
#pragma unroll
    for (int t = 0; t < kTok; ++t) {
      const int m = t * 16 + rl;
      if (m < p.M) {
        const float sum[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
        finish4<uint32_t>(p, alpha, m, n0, sum);
      }
    }
  }
}

static int64_t rowmid_pairs(int64_t K) { return (K + 255) / 256; }

static int rowmid_lds_bytes(int M, int64_t K, int slices, int tok, RowmidParams* p) {
  const int pairs = (int)rowmid_pairs(K);
  const int a_stride = pairs * 128 + 16;                    // + 16: token rows start in different banks
  const int s_stride = pairs * 16 + 4;
  const int sf_off = (M * a_stride + 15) & ~15;
  int total = sf_off + M * s_stride;
  const int red = slices > 1 ? kRmWaves * tok * 64 * 4 * (int)sizeof(float) : 0;      // aliased onto the codes (after a barrier)
  if (red > total) total = red;
  if (p) { p->a_stride = a_stride; p->s_stride = s_stride; p->sf_off = sf_off; }
  return (total + 15) & ~15;
}

// 1 = this shape runs on the mid-M repacked path: 16 < M <= 64 and the packed activations fit one workgroup's LDS twice per CU
// is NOT required -- one workgroup per CU is enough to stream (the launcher asks for what fits)
// ---- which kernel for a decode batch (16 < M <= 128)?  Measured on MI355X (tools/rowtok_ab.py, profiles/r03_decode_batch_kernels_ab.jsonl;
// HBM-cold, us: LDS-resident packed activations | no LDS | tiled GEMM + finish pass):
//   4096 x 4096    M = 32  8.9 |  7.5 | 16.7    M = 64 13.5 | 10.9 | 21.3    M = 128    - | 19.2 | 23.2
//   3584 x 3648    M = 32  8.7 |  6.5 | 19.7    M = 64 13.2 |  8.8 | 24.5    M = 128    - | 14.8 | 21.6
//   10752 x 3648   M = 32 15.1 | 16.4 | 21.4    M = 64 23.9 | 24.7 | 26.6    M = 128    - | 41.7 | 30.4
//   37888 x 3648   M = 32 34.7 | 40.1 | 49.4    M = 64 67.0 | 68.7 | 58.0
//   3584 x 19008   M = 32    - | 21.1 | 31.3    (the packed activations of 32 tokens x 19008 do not fit LDS)
// Small weights are latency chains: no LDS fill, no barrier wins at every M.  Large weights are compute-bound per token tile: LDS operands
// (one ds_read_b128 instead of a global load per tile) win while they fit, then the tiled GEMM.
enum : int { kMidNone = 0, kMidLds = 1, kMidTok = 2 };
static int rowtok_env() {       // ARCQ_ROWTOK (tuning / A-B): 0 = never the no-LDS kernel, 1 = always (16 < M <= 128), unset = by shape
  static const int v = getenv("ARCQ_ROWTOK") ? atoi(getenv("ARCQ_ROWTOK")) : -1;
  return v;
}
static int mid_kind(int64_t M, int64_t N, int64_t K) {
  if (M <= 16 || M > 128 || N < 1 || K < 64 || (K % 64)) return kMidNone;
  const bool lds_fits = M <= 64 && rowmid_lds_bytes((int)M, K, 8, (int)((M + 15) / 16), nullptr) <= 160 * 1024;
  const int e = rowtok_env();
  if (e == 1) return kMidTok;
  const int64_t w = N * K;
  if (e != 0 && w <= ((int64_t)32 << 20)) return kMidTok;                 // small weights: every M up to 128
  if (M <= 32) return lds_fits ? kMidLds : (e != 0 ? kMidTok : kMidNone);
  if (M <= 64 && w <= ((int64_t)64 << 20) && lds_fits) return kMidLds;
  return kMidNone;                                                          // the tiled GEMM (arcq_gemm_nvfp4) is faster
}

int gemm_repacked_mid_supported(int64_t M, int64_t N, int64_t K) { return mid_kind(M, N, K) != kMidNone ? 1 : 0; }

int gemm_repacked_mid(const GemmArgs& a, const uint8_t* RW, const uint8_t* RSF, hipStream_t stream) {
  if (a.epilogue != kEpiPlain) return fail(ARCQ_ERR_UNSUPPORTED, "arcq_gemm_nvfp4_repacked: M=%d > 16 has no SiLU epilogue", a.M);
  const int kind = mid_kind(a.M, a.N, a.K);
  if (kind == kMidNone)
    return fail(ARCQ_ERR_UNSUPPORTED, "arcq_gemm_nvfp4_repacked: M=%d N=%d K=%d outside the repacked path (see arcq_gemm_repacked_supported)", a.M, a.N, a.K);
  const bool tokk = kind == kMidTok;
  RowmidParams p;
  p.A = a.A; p.SFA = a.SFA; p.RW = RW; p.RSF = RSF; p.D = a.D;
  p.alpha_dev = a.alpha_dev; p.bias = a.bias; p.residual = a.residual;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  p.pairs = (int)rowmid_pairs(a.K);
  p.row_blocks = (a.N + 15) / 16;
  static const int forced = getenv("ARCQ_ROWMID_SLICES") ? atoi(getenv("ARCQ_ROWMID_SLICES")) : 0;   // tuning only
  int s = rowblock_choose_slices(p.row_blocks, p.pairs);
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8) s = forced;
  if (s > p.pairs) s = 1;
  p.slices = s;
  const int tok = (a.M + 15) / 16;
  if (tokk) {                                               // no activations in LDS: only the slice reduction's scratch
    const int ktok = tok <= 4 ? tok : (tok <= 6 ? 6 : 8);
    const int tlds = s > 1 ? kRmWaves * ktok * 64 * 4 * (int)sizeof(float) : 0;
    const int tbpw = kRmWaves / s;
    const int tgrid = (p.row_blocks + tbpw - 1) / tbpw;
    p.a_stride = p.s_stride = p.sf_off = 0;
    static LdsOptIn tok_lds[5];
    auto tlaunch = [&](auto kernel, LdsOptIn* opt) -> int {
      if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), *opt, tlds, "arcq_gemm_nvfp4_repacked")) return rc;
      hipLaunchKernelGGL(kernel, dim3((unsigned)tgrid), dim3(kRmThreads), tlds, stream, p);
      return ARCQ_OK;
    };
    int rc;
    switch (ktok) {
      case 2: rc = tlaunch(gemm_rowtok_kernel<2>, &tok_lds[0]); break;
      case 3: rc = tlaunch(gemm_rowtok_kernel<3>, &tok_lds[1]); break;
      case 4: rc = tlaunch(gemm_rowtok_kernel<4>, &tok_lds[2]); break;
      case 6: rc = tlaunch(gemm_rowtok_kernel<6>, &tok_lds[3]); break;
      default: rc = tlaunch(gemm_rowtok_kernel<8>, &tok_lds[4]); break;
    }
    if (rc != ARCQ_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4_repacked: launch failed: %s", hipGetErrorString(e));
    return ARCQ_OK;
  }
  const int lds = rowmid_lds_bytes(a.M, a.K, s, tok, &p);
  const int bpw = kRmWaves / s;
  const int grid = (p.row_blocks + bpw - 1) / bpw;
  auto launch = [&](auto kernel, LdsOptIn* opt) -> int {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), *opt, lds, "arcq_gemm_nvfp4_repacked")) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kRmThreads), lds, stream, p);
    return ARCQ_OK;
  };
  static LdsOptIn lds_set[3];           // one per kernel instantiation, each per device
  int rc;
  if (tok == 2) rc = launch(gemm_rowmid_kernel<2>, &lds_set[0]);
  else if (tok == 3) rc = launch(gemm_rowmid_kernel<3>, &lds_set[1]);
  else rc = launch(gemm_rowmid_kernel<4>, &lds_set[2]);
  if (rc != ARCQ_OK) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4_repacked: launch failed: %s", hipGetErrorString(e));
  return ARCQ_OK;
}

}  // namespace arcq
