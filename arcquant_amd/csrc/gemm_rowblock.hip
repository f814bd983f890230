// ARC-NVFP4 GEMM for decode shapes over a REPACKED weight: no LDS transpose, no barrier in the K loop.
//
// The reference layout (row-major packed codes, CUTLASS-swizzled scales; kernels/src/nvfp4.cu:35-132 consumes it through
// TMA) forces the decode kernels of gemm_skinny.hip / gemm_decode.hip to fetch full 512-byte row segments and
// transpose them to the MFMA operand layout through LDS: one barrier per item and ~92 instructions per 16 bytes of
// weights, which is what bounds them (profiles/r01d_pmc_decode_gemm.json; the same weights served from the Infinity
// Cache only gain 0-20 %).  Loading the operand layout directly from the row-major weight means 64-byte row segments:
// 0.6-3.8 TB/s (tools/probe_rows.hip).  A weight is static, so it can be laid out for the hardware ONCE:
//
//   RW : tiles of 16 weight rows x 128 K elements = 1 KB, stored in MFMA operand order -- lane l = 16*q + r holds the
//        16 bytes (32 codes) of row r, K elements [32q, 32q + 32) of the tile; tiles of one row block are consecutive,
//        so a wave streams ONE contiguous span with fully coalesced global_load_dwordx4
//   RSF: per PAIR of tiles 256 bytes: lane l holds the four ue4m3 bytes of its two 16-element groups in tile 2j and in
//        tile 2j + 1 (one dword load per lane per two tiles)
//   K is padded to a multiple of 256 with zero codes and zero scales, N to a multiple of 16 with zero scales.
//   (arcquant_amd/agemm.py: repack_w builds both with torch ops from the reference layout; nothing is re-quantised.)
//
// Kernel: a workgroup of 8 waves first dequantises the M x K activations once into an fp16 LDS image (one barrier);
// after that every wave is on its own: it owns (row block, K slice), streams its tiles through a 3-deep register ring
// addressed by name (exactly counted vmcnt), dequantises in registers (exact, gemm_common.hpp), multiplies on
// v_mfma_f32_16x16x32_f16 with the activation fragments read from the LDS image, and -- when a row block is split over
// S waves -- adds the S partial tiles through LDS at the very end.  ~50 instructions per 16 bytes of weights.
// Shapes: M <= 16 and an activation image that fits LDS (M * K_padded * 2 B <= 160 KB); others use the other kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "arcq_internal.hpp"
#include "gemm_common.hpp"
#include "rowblock_split.hpp"

namespace arcq {

struct RowblockParams {
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
  int a_stride;           // bytes per token row of the LDS image
  unsigned int* silu_slots;   // kSiluAbsmax: one word per row block, max |silu(gate) * up| of its outputs (bf16 bits)
};

typedef uint32_t rb_u32x4 __attribute__((ext_vector_type(4)));
struct RowblockRegs {     // one tile pair of this lane: 2 x 16 bytes of codes, 4 scale bytes
  rb_u32x4 b0, b1;
  uint32_t s;
};

constexpr int kRbWaves = 8, kRbThreads = kRbWaves * 64;

// kUnits: image units (32 activations) per thread whose loads are issued up front (1, 2, 4 or 8)
// (a ring of six pairs instead of three was measured on eight decode shapes, also those that leave a CU with a single
//  workgroup: 0-15 % SLOWER everywhere -- the stream is not latency bound, the extra loads only delay the first tile)
// kSiluAbsmax: the weight rows interleave gate and up (g0, u0, g1, u1, ...) and the caller quantises silu(gate) * up next
//              with a per-tensor dynamic scale: besides D, every row block leaves max |silu(g) * u| of its outputs (computed
//              from the very bf16 values it stores, with the quantiser's own silu_mul_bf16) in silu_slots[row block], which
//              saves the quantiser its abs-max launch (~4 us + a graph-node gap per decoder layer).  Two exp per lane, once.
template <int kUnits, bool kSiluAbsmax>
__global__ __launch_bounds__(kRbThreads, 4) void gemm_rowblock_kernel(RowblockParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const a_img = smem;                                        // [M][a_stride] fp16
  float* const red = reinterpret_cast<float*>(smem);    // [8 waves][64][4] when slices > 1: REUSES the image after the K loop

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, rl = lane & 15;
  const int bpw = kRbWaves / p.slices;                    // row blocks per workgroup
  const int rb = blockIdx.x * bpw + wave / p.slices;
  const int slice = wave % p.slices;
  const bool active = rb < p.row_blocks;
  // balanced K split: slice s owns base (+1 for the first `extra` slices) consecutive tile pairs -- with the launcher's
  // slices <= pairs no slice is empty (an empty slice used to address one pair past its row block: out of bounds behind
  // the last one).  A forced slices > pairs leaves trailing slices with npairs == 0; their loads are clamped to pair 0.
  int pr_begin, pr_count;
  rowblock_slice_range(p.pairs, p.slices, slice, &pr_begin, &pr_count);
  const int npairs = active ? pr_count : 0;
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);

  // ---- this thread's units of the activation image are requested FIRST: a wave's loads return in order, so behind
  //      the nine weight loads below they would only arrive with them (HBM latency) instead of at L2 latency -- and a
  //      thread with several units (large M * K) would pay that latency once per unit
  const int upr = p.pairs * 8, real = p.K >> 5;            // image units (32 elements) per token row: padded / real
  const int atoms_k = p.K >> 6;
  const int units = p.M * upr;
  uint4 qv_pre[kUnits];
  uint32_t sf_pre[kUnits];
#pragma unroll
  for (int j = 0; j < kUnits; ++j) {
    const int u = min(tid + j * kRbThreads, units - 1);
    const int m = u / upr, c = min(u - m * upr, real - 1);
    qv_pre[j] = *reinterpret_cast<const uint4*>(p.A + (size_t)m * (p.K >> 1) + c * 16);
    sf_pre[j] = *reinterpret_cast<const uint32_t*>(p.SFA + sf_atom_offset(m, c >> 1, atoms_k));   // the atom's 4 bytes
  }

  // ---- then the weight stream: three tile pairs per lane in flight
  const int rbc = active ? rb : 0;
  const int pr_load = rowblock_load_base(pr_begin, npairs);   // never past the row block's own pairs
  const uint8_t* wp = p.RW + ((size_t)rbc * p.pairs + pr_load) * 2048 + lane * 16;
  const uint8_t* sp = p.RSF + ((size_t)rbc * p.pairs + pr_load) * 256 + lane * 4;
  const int last = npairs > 0 ? npairs - 1 : 0;
  int issued = 0;
  auto issue = [&](RowblockRegs& r) __attribute__((always_inline)) {       // unpredicated; the cursor stops at the last pair
    const int i = min(issued, last);
    // a weight byte is read once per launch by one CU: non-temporal loads (gemm_common.hpp)
    r.b0 = ARCQ_WLOAD(reinterpret_cast<const rb_u32x4*>(wp + (size_t)i * 2048));
    r.b1 = ARCQ_WLOAD(reinterpret_cast<const rb_u32x4*>(wp + (size_t)i * 2048 + 1024));
    r.s = ARCQ_WLOAD(reinterpret_cast<const uint32_t*>(sp + (size_t)i * 256));
    ++issued;
  };
  RowblockRegs r0, r1, r2;
  issue(r0);
  issue(r1);
  issue(r2);

  // ---- activation image: unit = 32 elements (16 packed bytes, two scale bytes) -> 64 bytes of fp16
  auto put_unit = [&](int u, uint4 qv, uint32_t sf) __attribute__((always_inline)) {
    const int m = u / upr, c = u - m * upr;
    uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0, f2 = f0, f3 = f0;
    if (c < real) {
      sf >>= (c & 1) * 16;
      const f16x2 s0 = sf_pair_at(sf, 0), s1 = sf_pair_at(sf, 8);
      f0 = dequant8(qv.x, s0).u; f1 = dequant8(qv.y, s0).u; f2 = dequant8(qv.z, s1).u; f3 = dequant8(qv.w, s1).u;
    }
    uint4* dst = reinterpret_cast<uint4*>(a_img + (size_t)m * p.a_stride + c * 64);
    dst[0] = f0; dst[1] = f1; dst[2] = f2; dst[3] = f3;
  };
#pragma unroll
  for (int j = 0; j < kUnits; ++j) {
    const int u = tid + j * kRbThreads;
    if (u < units) put_unit(u, qv_pre[j], sf_pre[j]);
  }
  for (int u = tid + kUnits * kRbThreads; u < units; u += kRbThreads) {       // beyond the prefetch (M * K > 128 K elements)
    const int m = u / upr, c = min(u - m * upr, real - 1);
    put_unit(u, *reinterpret_cast<const uint4*>(p.A + (size_t)m * (p.K >> 1) + c * 16),
             *reinterpret_cast<const uint32_t*>(p.SFA + sf_atom_offset(m, c >> 1, atoms_k)));
  }
  __syncthreads();

  // ---- K loop: no barrier, no LDS traffic for the weights
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned char* a_lane = a_img + (size_t)min(rl, p.M - 1) * p.a_stride + (size_t)pr_begin * 512 + q * 64;   // tokens >= M: any row
  int done = 0;
  auto tile = [&](rb_u32x4 b, uint32_t s16, const unsigned char* ap) __attribute__((always_inline)) {
    Frag8 a0, a1, a2, a3;
    a0.u = *reinterpret_cast<const uint4*>(ap);
    a1.u = *reinterpret_cast<const uint4*>(ap + 16);
    a2.u = *reinterpret_cast<const uint4*>(ap + 32);
    a3.u = *reinterpret_cast<const uint4*>(ap + 48);
    const f16x2 s0 = sf_pair_at(s16, 0), s1 = sf_pair_at(s16, 8);
    const Frag8 b0 = dequant8(b.x, s0), b1 = dequant8(b.y, s0), b2 = dequant8(b.z, s1), b3 = dequant8(b.w, s1);
    // weights are the MFMA A operand (rows = weight rows), activations the B operand (columns = tokens)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc, 0, 0, 0);
  };
  auto step = [&](RowblockRegs& r) __attribute__((always_inline)) {
    const unsigned char* ap = a_lane + (size_t)done * 512;
    tile(r.b0, r.s, ap);
    tile(r.b1, r.s >> 16, ap + 256);
    issue(r);                                               // refill: three pairs ahead
    __builtin_amdgcn_sched_barrier(0);                      // hipcc otherwise sinks all refills to the end of the unrolled body
    ++done;
  };
  // bias / residual of this wave's output tile (the slice-0 wave finishes the row block): fetched now, behind the ring's first
  // loads, instead of as a global round trip after the last barrier (N % 4 == 0: 8-byte loads; by-name registers)
  typedef uint32_t rb_u32x2 __attribute__((ext_vector_type(2)));
  rb_u32x2 ep_bias = {0, 0}, ep_res = {0, 0};
  const bool ep_pre = !kSiluAbsmax && (p.N & 3) == 0 && (p.bias || p.residual);
  if (ep_pre && slice == 0 && active && rl < p.M && rb * 16 + 4 * q < p.N) {
    const int n0p = rb * 16 + 4 * q;
    if (p.bias) ep_bias = *reinterpret_cast<const rb_u32x2*>(p.bias + n0p);
    if (p.residual) ep_res = *reinterpret_cast<const rb_u32x2*>(p.residual + (size_t)rl * p.N + n0p);
  }
#pragma unroll 1
  while (done + 3 <= npairs) {
    step(r0);
    step(r1);
    step(r2);
  }
  if (done < npairs) step(r0);
  if (done < npairs) step(r1);

  // ---- lane holds C[token = rl][row = 16 rb + 4q + e]; add the K slices of a row block through LDS
  float sum[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (p.slices > 1) {
    __syncthreads();                                        // every wave is done reading the activation image
    *reinterpret_cast<float4*>(red + (wave * 64 + lane) * 4) = make_float4(sum[0], sum[1], sum[2], sum[3]);
    __syncthreads();
    if (slice != 0) return;
    for (int s2 = 1; s2 < p.slices; ++s2) {
      const float4 v = *reinterpret_cast<const float4*>(red + ((wave + s2) * 64 + lane) * 4);
      sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
    }
  }
  const int n0 = rb * 16 + 4 * q;
  if (active && rl < p.M && n0 < p.N) {
    if (ep_pre) finish4_pre<uint32_t>(p, alpha, rl, n0, sum, make_uint2(ep_bias.x, ep_bias.y), make_uint2(ep_res.x, ep_res.y));
    else finish4<uint32_t>(p, alpha, rl, n0, sum);
  }
  if constexpr (kSiluAbsmax) {                              // N % 4 == 0, bf16 out, no bias / residual (checked by the launcher)
    uint32_t m = 0;
    if (active && rl < p.M && n0 < p.N) {
      const uint32_t b0 = f32_to_bf16_bits(alpha * sum[0]), b1 = f32_to_bf16_bits(alpha * sum[1]);
      const uint32_t b2 = f32_to_bf16_bits(alpha * sum[2]), b3 = f32_to_bf16_bits(alpha * sum[3]);
      m = max(silu_mul_bf16(b0, b1) & 0x7fffu, silu_mul_bf16(b2, b3) & 0x7fffu);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, sh, 64));
    if (lane == 0 && active) p.silu_slots[rb] = m;
  }
}

// ---- the same contraction WITHOUT the LDS image (kernel "direct") ------------------------------------------------------------------
// For small M the image is a detour on the critical path of a latency-bound kernel: activations -> dequantise -> LDS -> barrier ->
// first MFMA (config[1]: ~2 000 of 9 500 cycles, and every wave waits for the slowest one's activation loads), and for large K it
// costs the kernel its occupancy (Qwen2.5-7B down projection at bs = 4: a 154 KB image = ONE 8-wave workgroup per CU).  Here every
// lane fetches its MFMA B operand itself: lane (token rl, quarter q) of a wave loads the 16 packed bytes [16 q, 16 q + 16) of token
// min(rl, M - 1)'s row for each 128-element tile of its K slice, and the two scale bytes of its groups, next to the weight tile
// (requested BEFORE it: a wave's loads return in order), and dequantises them in registers exactly as the weights.  No LDS, no
// barrier before the K loop; the packed activations are a few KB per token and L2-resident (every workgroup reads the same bytes).
// Costs per tile pair: 4 more loads per lane (7 instead of 3) and 80 more conversion instructions -- which a latency-bound kernel has.
struct RowblockActRegs {  // activations of one tile pair for this lane: 2 x 16 bytes of codes, 2 x 2 scale bytes
  rb_u32x4 a0, a1;
  uint32_t s0, s1;
};

template <bool kSiluAbsmax>
__global__ __launch_bounds__(kRbThreads, 2) void gemm_rowblock_direct_kernel(RowblockParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const red = reinterpret_cast<float*>(smem);    // [8 waves][64][4] when slices > 1

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, rl = lane & 15;
  const int bpw = kRbWaves / p.slices;                    // row blocks per workgroup
  const int rb = blockIdx.x * bpw + wave / p.slices;
  const int slice = wave % p.slices;
  const bool active = rb < p.row_blocks;
  int pr_begin, pr_count;
  rowblock_slice_range(p.pairs, p.slices, slice, &pr_begin, &pr_count);
  const int npairs = active ? pr_count : 0;
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);

  const int rbc = active ? rb : 0;
  const int pr_load = rowblock_load_base(pr_begin, npairs);   // never past the row block's own pairs
  const uint8_t* wp = p.RW + ((size_t)rbc * p.pairs + pr_load) * 2048 + lane * 16;
  const uint8_t* sp = p.RSF + ((size_t)rbc * p.pairs + pr_load) * 256 + lane * 4;
  // activations of this lane: token min(rl, M - 1) (tokens >= M: any row, never stored); K padded to 256 in the weight only, so
  // tiles and atoms beyond K are clamped to the last real one and meet zero weight scales
  const int tok = min(rl, p.M - 1);
  const int atoms_k = p.K >> 6, tiles_k = p.K >> 7;          // K % 64 == 0: the last tile may be half (one atom)
  const uint8_t* arow = p.A + (size_t)tok * (p.K >> 1) + q * 16;
  const uint8_t* srow = p.SFA + sf_atom_offset(tok, 0, atoms_k) + (q & 1) * 2;
  const int last = npairs > 0 ? npairs - 1 : 0;
  int issued = 0;
  auto issue = [&](RowblockRegs& r, RowblockActRegs& x) __attribute__((always_inline)) {   // unpredicated; the cursor stops at the last pair
    const int i = min(issued, last);
    const int t0 = 2 * (pr_load + i), t1 = t0 + 1;                      // the pair's two 128-element tiles
    // a tile past K (padding): bytes of the last real tile (any finite codes), its weight scales are zero.  The second half of a
    // trailing half tile (K % 128 == 64) lies past the row as well: quarters q >= 2 re-read quarters q - 2.
    const int halfk = (p.K & 64) ? 1 : 0;
    auto a_off = [&](int t) __attribute__((always_inline)) -> size_t {
      const int tc = min(t, tiles_k - 1 + halfk);
      size_t off = (size_t)tc * 64;
      if (halfk && tc == tiles_k && q >= 2) off -= 32;                  // (only the half tile; its groups 4 .. 7 have zero weight scales)
      return off;
    };
    auto s_off = [&](int t) __attribute__((always_inline)) -> size_t {
      const int atom = min(2 * t + (q >> 1), atoms_k - 1);
      return (size_t)atom * 512;
    };
    x.a0 = *reinterpret_cast<const rb_u32x4*>(arow + a_off(t0));
    x.a1 = *reinterpret_cast<const rb_u32x4*>(arow + a_off(t1));
    x.s0 = *reinterpret_cast<const uint16_t*>(srow + s_off(t0));
    x.s1 = *reinterpret_cast<const uint16_t*>(srow + s_off(t1));
    // a weight byte is read once per launch by one CU: non-temporal loads (gemm_common.hpp)
    r.b0 = ARCQ_WLOAD(reinterpret_cast<const rb_u32x4*>(wp + (size_t)i * 2048));
    r.b1 = ARCQ_WLOAD(reinterpret_cast<const rb_u32x4*>(wp + (size_t)i * 2048 + 1024));
    r.s = ARCQ_WLOAD(reinterpret_cast<const uint32_t*>(sp + (size_t)i * 256));
    ++issued;
  };
  RowblockRegs r0, r1, r2;
  RowblockActRegs x0, x1, x2;
  issue(r0, x0);
  issue(r1, x1);
  issue(r2, x2);

  // bias / residual of this wave's output tile (the slice-0 wave finishes the row block), behind the ring's first loads
  typedef uint32_t rb_u32x2 __attribute__((ext_vector_type(2)));
  rb_u32x2 ep_bias = {0, 0}, ep_res = {0, 0};
  const bool ep_pre = !kSiluAbsmax && (p.N & 3) == 0 && (p.bias || p.residual);
  if (ep_pre && slice == 0 && active && rl < p.M && rb * 16 + 4 * q < p.N) {
    const int n0p = rb * 16 + 4 * q;
    if (p.bias) ep_bias = *reinterpret_cast<const rb_u32x2*>(p.bias + n0p);
    if (p.residual) ep_res = *reinterpret_cast<const rb_u32x2*>(p.residual + (size_t)rl * p.N + n0p);
  }

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int done = 0;
  auto tile = [&](rb_u32x4 b, uint32_t s16, rb_u32x4 a, uint32_t as16) __attribute__((always_inline)) {
    const f16x2 t0 = sf_pair_at(as16, 0), t1 = sf_pair_at(as16, 8);
    const Frag8 a0 = dequant8(a.x, t0), a1 = dequant8(a.y, t0), a2 = dequant8(a.z, t1), a3 = dequant8(a.w, t1);
    const f16x2 s0 = sf_pair_at(s16, 0), s1 = sf_pair_at(s16, 8);
    const Frag8 b0 = dequant8(b.x, s0), b1 = dequant8(b.y, s0), b2 = dequant8(b.z, s1), b3 = dequant8(b.w, s1);
    // weights are the MFMA A operand (rows = weight rows), activations the B operand (columns = tokens)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc, 0, 0, 0);
  };
  auto step = [&](RowblockRegs& r, RowblockActRegs& x) __attribute__((always_inline)) {
    tile(r.b0, r.s, x.a0, x.s0);
    tile(r.b1, r.s >> 16, x.a1, x.s1);
    issue(r, x);                                            // refill: three pairs ahead
    __builtin_amdgcn_sched_barrier(0);
    ++done;
  };
#pragma unroll 1
  while (done + 3 <= npairs) {
    step(r0, x0);
    step(r1, x1);
    step(r2, x2);
  }
  if (done < npairs) step(r0, x0);
  if (done < npairs) step(r1, x1);

  // ---- lane holds C[token = rl][row = 16 rb + 4q + e]; add the K slices of a row block through LDS
  float sum[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (p.slices > 1) {
    *reinterpret_cast<float4*>(red + (wave * 64 + lane) * 4) = make_float4(sum[0], sum[1], sum[2], sum[3]);
    __syncthreads();
    if (slice != 0) return;
    for (int s2 = 1; s2 < p.slices; ++s2) {
      const float4 v = *reinterpret_cast<const float4*>(red + ((wave + s2) * 64 + lane) * 4);
      sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
    }
  }
  const int n0 = rb * 16 + 4 * q;
  if (active && rl < p.M && n0 < p.N) {
    if (ep_pre) finish4_pre<uint32_t>(p, alpha, rl, n0, sum, make_uint2(ep_bias.x, ep_bias.y), make_uint2(ep_res.x, ep_res.y));
    else finish4<uint32_t>(p, alpha, rl, n0, sum);
  }
  if constexpr (kSiluAbsmax) {                              // N % 4 == 0, bf16 out, no bias / residual (checked by the launcher)
    uint32_t m = 0;
    if (active && rl < p.M && n0 < p.N) {
      const uint32_t b0 = f32_to_bf16_bits(alpha * sum[0]), b1 = f32_to_bf16_bits(alpha * sum[1]);
      const uint32_t b2 = f32_to_bf16_bits(alpha * sum[2]), b3 = f32_to_bf16_bits(alpha * sum[3]);
      m = max(silu_mul_bf16(b0, b1) & 0x7fffu, silu_mul_bf16(b2, b3) & 0x7fffu);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, sh, 64));
    if (lane == 0 && active) p.silu_slots[rb] = m;
  }
}

// Repacked sizes: rows padded to 16, K to 256.
static int64_t rowblock_pairs(int64_t K) { return (K + 255) / 256; }
int64_t gemm_repacked_w_bytes(int64_t N, int64_t K) { return ((N + 15) / 16) * rowblock_pairs(K) * 2048; }
int64_t gemm_repacked_sf_bytes(int64_t N, int64_t K) { return ((N + 15) / 16) * rowblock_pairs(K) * 256; }

static int rowblock_lds_bytes(int M, int64_t K, int slices, int* a_stride) {
  const int stride = (int)(rowblock_pairs(K) * 512 + 16);          // + 16: token rows start in different banks
  *a_stride = stride;
  const int red = slices > 1 ? kRbWaves * 64 * 4 * (int)sizeof(float) : 0;      // aliased onto the image
  return M * stride > red ? M * stride : red;
}

// 1 = this shape can run on the repacked path: M <= 16 and the fp16 activation image fits LDS (this file), or 16 < M <= 64 and
// the PACKED activations fit LDS (gemm_rowmid.hip)
int gemm_repacked_supported(int64_t M, int64_t N, int64_t K) {
  if (M > 16) return gemm_repacked_mid_supported(M, N, K);
  if (M < 1 || N < 1 || K < 64 || (K % 64)) return 0;
  int stride;
  return rowblock_lds_bytes((int)M, K, 8, &stride) <= 160 * 1024 ? 1 : 0;
}

// Which plain repacked shapes take the no-image kernel (tools/direct_ab.py on MI355X, us per launch image / direct, HBM-cold):
// M=1 N=K=4096 5.65 / 5.53, M=4 6.10 / 5.74, M=16 7.77 / 6.37, 3584 x 3648 M=4 5.53 / 5.35 -- but 1024 x 4160 4.74 / 4.92 (too few
// workgroups to hide seven loads per pair) and every large weight loses (14336 x 4160 9.6 / 11.8, 10752 x 3648 8.3 / 10.3,
// 37888 x 3648 19.6 / 23.9: bandwidth-bound kernels pay for the four extra loads per pair), 3584 x 19008 ties.
static bool rowblock_use_direct(int M, int N, int64_t K) {
  const int64_t w = (int64_t)N * K;
  return w >= ((int64_t)8 << 20) && w <= ((int64_t)24 << 20);
}

int gemm_repacked(const GemmArgs& a, const uint8_t* RW, const uint8_t* RSF, hipStream_t stream) {
  static const int use_stream = getenv("ARCQ_REPACKED_STREAM") ? atoi(getenv("ARCQ_REPACKED_STREAM")) : 0;   // tuning / A-B only
  if (a.M > 16) return gemm_repacked_mid(a, RW, RSF, stream);
  if (use_stream) return gemm_repacked_stream(a, RW, RSF, stream);
  const bool silu = a.epilogue == kEpiSiluMul;              // here: D stays gate|up, absmax_slots gets max |silu(g) * u| per row block
  if (silu && (!a.absmax_slots || (a.N % 4) || a.bias || a.residual || a.out_dtype != ARCQ_OUT_BF16))
    return fail(ARCQ_ERR_SHAPE, "arcq_gemm_nvfp4_repacked_silu_absmax: needs absmax_slots, N %% 4 == 0, bf16 output, no bias / residual");
  if (!gemm_repacked_supported(a.M, a.N, a.K))
    return fail(ARCQ_ERR_UNSUPPORTED, "arcq_gemm_nvfp4_repacked: M=%d K=%d outside the repacked path (M <= 16, activation image <= 160 KB)", a.M, a.K);
  RowblockParams p;
  p.A = a.A; p.SFA = a.SFA; p.RW = RW; p.RSF = RSF; p.D = a.D;
  p.alpha_dev = a.alpha_dev; p.bias = a.bias; p.residual = a.residual;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  p.silu_slots = a.absmax_slots;
  p.pairs = (int)rowblock_pairs(a.K);
  p.row_blocks = (a.N + 15) / 16;
  static const int forced = getenv("ARCQ_ROWBLOCK_SLICES") ? atoi(getenv("ARCQ_ROWBLOCK_SLICES")) : 0;   // tuning only
  int s = rowblock_choose_slices(p.row_blocks, p.pairs);
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8) s = forced;
  p.slices = s;
  const int lds = rowblock_lds_bytes(a.M, a.K, s, &p.a_stride);
  const int units = a.M * p.pairs * 8;
  const int per_thread = (units + kRbThreads - 1) / kRbThreads;
  const int bpw = kRbWaves / s;
  const int grid = (p.row_blocks + bpw - 1) / bpw;
  auto launch = [&](auto kernel, LdsOptIn* opt) -> int {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), *opt, lds, "arcq_gemm_nvfp4_repacked")) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kRbThreads), lds, stream, p);
    return ARCQ_OK;
  };
  static LdsOptIn lds_set[8];           // one per kernel instantiation, each per device
  // the no-image kernel: ARCQ_ROWBLOCK_DIRECT=0/1 forces (A-B), default by shape (rowblock_use_direct)
  static const int direct_env = getenv("ARCQ_ROWBLOCK_DIRECT") ? atoi(getenv("ARCQ_ROWBLOCK_DIRECT")) : -1;
  const bool direct = direct_env >= 0 ? direct_env != 0 : rowblock_use_direct(a.M, a.N, a.K);
  if (direct) {
    const int dlds = s > 1 ? kRbWaves * 64 * 4 * (int)sizeof(float) : 0;
    if (silu) hipLaunchKernelGGL(gemm_rowblock_direct_kernel<true>, dim3((unsigned)grid), dim3(kRbThreads), dlds, stream, p);
    else hipLaunchKernelGGL(gemm_rowblock_direct_kernel<false>, dim3((unsigned)grid), dim3(kRbThreads), dlds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4_repacked: launch failed: %s", hipGetErrorString(e));
    return ARCQ_OK;
  }
  int rc;
  if (silu) {
    if (per_thread <= 1) rc = launch(gemm_rowblock_kernel<1, true>, &lds_set[4]);
    else if (per_thread <= 2) rc = launch(gemm_rowblock_kernel<2, true>, &lds_set[5]);
    else if (per_thread <= 4) rc = launch(gemm_rowblock_kernel<4, true>, &lds_set[6]);
    else rc = launch(gemm_rowblock_kernel<8, true>, &lds_set[7]);
  } else {
    if (per_thread <= 1) rc = launch(gemm_rowblock_kernel<1, false>, &lds_set[0]);
    else if (per_thread <= 2) rc = launch(gemm_rowblock_kernel<2, false>, &lds_set[1]);
    else if (per_thread <= 4) rc = launch(gemm_rowblock_kernel<4, false>, &lds_set[2]);
    else rc = launch(gemm_rowblock_kernel<8, false>, &lds_set[3]);
  }
  if (rc != ARCQ_OK) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4_repacked: launch failed: %s", hipGetErrorString(e));
  return ARCQ_OK;
}

}  // namespace arcq
