// ARC-NVFP4 GEMM for decode shapes (M <= 16), second generation: a weight-streaming kernel for gfx950
// designed from its measured limiter.
//
// Replaces the CUTLASS 128x128x128 block-scaled GEMM of the reference (kernels/src/nvfp4.cu:35-132)
// for the shapes where that kernel leaves 127/128 of its M tile empty ("decode").
//
// What bounds it (profiles/r01d_pmc_decode_gemm.json, first-generation kernel on N=37888 K=3648 M=4):
// HBM can deliver this access pattern at 5.1-5.7 TB/s (tools/probe_rows.hip), but the kernel ran at 2.7 TB/s with the
// VALU pipe 57 % busy and waves lock-stepped by one barrier per item -- it is INSTRUCTION bound: 123 instructions
// per wave per 16 bytes of weights, of which only 32 (fp4 -> fp16: v_cvt_scalef32_pk_f16_fp4 at ~7 cycles,
// v_pk_mul_f16 at 4, tools/probe_rate.hip) are irreducible.  Loading weights straight into the MFMA operand
// layout (16 rows x 64 B per wave instruction) is not an option either: that pattern reaches 0.6-3.8 TB/s.
// Hence this design: do more bytes per instruction.
//   * an ITEM is 32 weight rows x 1024 K elements (16 KB of packed weights): every thread of the 8-wave
//     workgroup fetches TWO 16-byte units (full 512-byte row segments per wave instruction), and every
//     wave multiplies a 128-element K chunk against both 16-row blocks -- the activation fragments, the
//     barrier, the address arithmetic and the loop are paid once per 32 bytes per thread instead of per 16
//   * the M x 1024 activations of an item are dequantised cooperatively, one dword (8 codes) per thread
//     and 4 token rows (M <= 4: a single pass), instead of by two loader waves the others wait for
//   * a 3-deep register ring ADDRESSED BY NAME (item loop unrolled x3) keeps three items in flight; every
//     thread issues the same unpredicated loads per item, so hipcc can wait with an exact vmcnt(N) -- a
//     predicated load, a 16-bit load widened in the ring, or a ring rotated by moves each degrade to
//     vmcnt(0), i.e. to one item in flight
//   * the 32 rows of a tile are {128T + 32j + 8t + i}: their scale bytes fill whole 128-byte lines of the
//     swizzled scale layout, and a lane ends up with 8 consecutive output columns
//   * packed bytes are transposed into the MFMA operand layout through a double-buffered LDS image
//     (padded rows: conflict-free), one barrier per item; cross-wave reduction through LDS per tile;
//     fused epilogue; split-K over slabs (second pass) only when the tiles alone cannot fill the chip.
//   * scale pairs are prepared with two instructions (v_bfe_u32 + one 24-bit multiply), row-bound masks once per tile:
//     184 instructions per item-step (32 B of weights per thread) against 246 per 32 B in the first generation.
// Used from N = 5120 up (c_api.hip); below that a workgroup lives for 2-5 items and gemm_skinny.hip is faster.
// Measured (tools/decode_bench.py, M=4, HIP-graph replay): N=37888 K=3648 28.7 -> 22.6 us (3.46 TB/s), N=14336 K=4160
// 16.0 -> 13.5 us.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "arcq_internal.hpp"
#include "gemm_common.hpp"

namespace arcq {

struct DecodeParams {
  const uint8_t* A;
  const uint8_t* B;
  const uint8_t* SFA;
  const uint8_t* SFB;
  void* D;
  float* partial;         // [splitk, M, N] fp32 when splitk > 1
  const float* alpha_dev;
  const uint16_t* bias;
  const uint16_t* residual;
  int M, N, K;
  float alpha_host;
  int out_dtype;
  int tiles;              // 32-row tiles: ceil(N / 128) * 4
  int slabs_per_split;
  int epi;                // kEpiPlain | kEpiSiluMul (interleaved gate/up rows -> D = bf16 [M, N/2], slots[tile] = max |D|)
  unsigned int* slots;
};

constexpr int kDecWaves = 8, kDecThreads = kDecWaves * 64;
constexpr int kDecSlabK = 1024;                      // K elements per item
constexpr int kDecSlabBytes = kDecSlabK / 2;         // packed bytes per row per item
constexpr int kDecSlabAtoms = kDecSlabK / 64;        // scale-factor atoms per item
constexpr int kDecBStride = kDecSlabBytes + 16;      // packed B image: [32 rows][512 + 16]
constexpr int kDecBImg = 32 * kDecBStride;
constexpr int kDecAStride = kDecSlabK * 2 + 16;      // fp16 A image: [M + 1 tokens][2048 + 16]; the extra token stays zero
constexpr int kDecRedBytes = kDecWaves * 64 * 8 * (int)sizeof(float);
static int decode_lds_bytes(int M) { return 2 * (kDecBImg + (M + 1) * kDecAStride) + kDecRedBytes; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int kADw>
struct DecodeRegs {
  u32x4 b0, b1;          // loader role: this thread's 16-byte units of row blocks 0 and 1
  uint32_t sb0, sb1;     // compute role: this lane's 4 scale bytes (one atom) of its row in block 0 / 1
  // loader role: 8 activation codes of token 4i + (tid >> 7), i < kADw, and the 4 scale bytes of their atom.
  // Plain scalar members accessed by name: an array member (or an accessor returning a reference) sends the whole
  // ring to scratch memory.
  uint32_t a0, a1, a2, a3;
  uint32_t sa0, sa1, sa2, sa3;
};

// kADw = activation dwords per thread and item = ceil(M / 4); kEpi = kEpiPlain | kEpiSiluMul (a template parameter: as
// a run-time branch the SiLU epilogue's exp code cost the plain kernel 26 spilled registers and 3.5 us on qkv)
template <int kADw, int kEpi>
__global__ __launch_bounds__(kDecThreads, kADw == 1 ? 4 : 2) void gemm_decode_kernel(DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lds_buf = kDecBImg + (p.M + 1) * kDecAStride;
  float* red = reinterpret_cast<float*>(smem + 2 * lds_buf);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int q = lane >> 4;           // K quarter of the wave's 128-element chunk
  const int rl = lane & 15;          // MFMA row index rho (weight row within a block) / token index
  const int ld_rho = tid >> 5;       // B loader role: row within a block ...
  const int ld_u = tid & 31;         // ... and 16-byte unit (32 elements) of the item's row segment
  const int la_m = tid >> 7;         // A loader role: token (mod 4) ...
  const int la_c = tid & 127;        // ... and dword (8 elements) of the item

  const int atoms_k = p.K >> 6;
  const uint32_t half_k = (uint32_t)p.K >> 1;
  const int slab_begin = blockIdx.y * p.slabs_per_split;
  const int nslabs = min((p.K + kDecSlabK - 1) / kDecSlabK, slab_begin + p.slabs_per_split) - slab_begin;
  const int G = gridDim.x;
  const int my_tiles = (p.tiles - (int)blockIdx.x + G - 1) / G;
  const int nitems = my_tiles * nslabs;
  const bool m_ok = rl < p.M;
  // read before the ring starts: a load consumed inside the item loop would have to wait for every ring load
  // issued before it (vmcnt is an in-order counter)
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);

  // A tile's 32 rows are {128*T + 32*j + 8*t + 4*b + i}: block b, rho = 4j + i
  const int ld_rowpart = (ld_rho >> 2) * 32 + (ld_rho & 3);
  const int cm_rowpart = (rl >> 2) * 32 + (rl & 3);
  const uint32_t k_first = (uint32_t)slab_begin * kDecSlabBytes + ld_u * 16;   // byte offset of this thread's unit in slab 0
  const uint32_t k_last = half_k - 16u;                                        // clamp for the partial tail slab
  const uint32_t sfb_small = (rl & 3) * 16 + (rl >> 2) * 4;
  const uint32_t sfb_lane = sfb_small + (2 * wave + (q >> 1)) * 512 + (uint32_t)slab_begin * kDecSlabAtoms * 512u;
  const uint32_t ak_first = (uint32_t)slab_begin * kDecSlabBytes + la_c * 4;
  const uint32_t ak_last = half_k - 4u;
  uint32_t a_row[kADw], sfa_row[kADw];
#pragma unroll
  for (int i = 0; i < kADw; ++i) {
    const int m = min(4 * i + la_m, p.M - 1);             // tokens >= M duplicate the last one (never stored)
    a_row[i] = (uint32_t)m * half_k;
    sfa_row[i] = (uint32_t)m * 16;                        // sf_atom_offset(m < 32, atom) = atom * 512 + m * 16
  }
  const uint32_t sfa_first = ((uint32_t)slab_begin * kDecSlabAtoms + (la_c >> 3)) * 512u;
  const uint32_t sfa_last = (uint32_t)(atoms_k - 1) * 512u;

  // ---- issue side state (runs 3 items ahead of the compute side); all offsets are carried incrementally
  int iss_tile = blockIdx.x, iss_slab = 0, issued = 0;
  uint32_t b_row0 = 0, b_row1 = 0, sfb_row = 0, sfb_rowmax = 0;       // per-tile parts
  auto issue_tile_setup = [&]() __attribute__((always_inline)) {
    const int tile_part = (iss_tile >> 2) * 128 + (iss_tile & 3) * 8;
    b_row0 = (uint32_t)min(tile_part + ld_rowpart, p.N - 1) * half_k;
    b_row1 = (uint32_t)min(tile_part + 4 + ld_rowpart, p.N - 1) * half_k;
    sfb_row = (uint32_t)(iss_tile >> 2) * atoms_k * 512u + (iss_tile & 3) * 128u;
    sfb_rowmax = sfb_row + (uint32_t)(atoms_k - 1) * 512u + sfb_small;
  };
  issue_tile_setup();
  uint32_t k_cur = k_first, sfb_cur = sfb_lane, ak_cur = ak_first, sfa_cur = sfa_first;
  // Every thread issues the SAME loads for every item (rows clamped, nothing predicated), and the cursor stops
  // at the last item, so the ring's final refills re-read valid memory.
  auto issue_next = [&](DecodeRegs<kADw>& r) __attribute__((always_inline)) {
    const uint32_t koff = min(k_cur, k_last);
    r.b0 = ARCQ_WLOAD(reinterpret_cast<const u32x4*>(p.B + (size_t)(b_row0 + koff)));
    r.b1 = ARCQ_WLOAD(reinterpret_cast<const u32x4*>(p.B + (size_t)(b_row1 + koff)));
    const uint32_t so = min(sfb_row + sfb_cur, sfb_rowmax);
    r.sb0 = ARCQ_WLOAD(reinterpret_cast<const uint32_t*>(p.SFB + (size_t)so));
    r.sb1 = ARCQ_WLOAD(reinterpret_cast<const uint32_t*>(p.SFB + (size_t)so + 64));
    const uint32_t akoff = min(ak_cur, ak_last), sao = min(sfa_cur, sfa_last);
    auto ld_a = [&](int i) { return *reinterpret_cast<const uint32_t*>(p.A + (size_t)(a_row[i] + akoff)); };
    auto ld_sa = [&](int i) { return *reinterpret_cast<const uint32_t*>(p.SFA + (size_t)(sfa_row[i] + sao)); };
    r.a0 = ld_a(0); r.sa0 = ld_sa(0);
    if constexpr (kADw >= 2) { r.a1 = ld_a(1); r.sa1 = ld_sa(1); }
    if constexpr (kADw >= 4) { r.a2 = ld_a(2); r.sa2 = ld_sa(2); r.a3 = ld_a(3); r.sa3 = ld_sa(3); }
    if (issued + 1 < nitems) {                                 // wave-uniform, address arithmetic only
      k_cur += kDecSlabBytes; ak_cur += kDecSlabBytes; sfb_cur += kDecSlabAtoms * 512u; sfa_cur += kDecSlabAtoms * 512u;
      if (++iss_slab == nslabs) {
        iss_slab = 0; iss_tile += G;
        k_cur = k_first; ak_cur = ak_first; sfb_cur = sfb_lane; sfa_cur = sfa_first;
        issue_tile_setup();
      }
    }
    ++issued;
  };

  DecodeRegs<kADw> r0, r1, r2;
  issue_next(r0);
  issue_next(r1);
  issue_next(r2);

  // the spare token row of both A images stays zero; MFMA columns >= M read it
  if (tid < kDecAStride / 16) {
    *reinterpret_cast<uint4*>(smem + kDecBImg + p.M * kDecAStride + tid * 16) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(smem + lds_buf + kDecBImg + p.M * kDecAStride + tid * 16) = make_uint4(0, 0, 0, 0);
  }

  const int wrb_off = ld_rho * kDecBStride + ld_u * 16;                  // loader role: packed B unit of block 0
  const int wra_off = kDecBImg + la_m * kDecAStride + la_c * 16;         // loader role: 8 dequantised activations of token la_m
  const int rdb_off = rl * kDecBStride + (4 * wave + q) * 16;            // compute role: this lane's packed B fragment of block 0
  const int rda_off = kDecBImg + min(rl, p.M) * kDecAStride + (4 * wave + q) * 64;   // ... and its four fp16 A fragments
  const int sh = (q & 1) * 16;                                           // which two of the atom's four B scale bytes
  const int sa_sh = ((la_c >> 1) & 3) * 8;                               // which of the atom's four A scale bytes
  const int lane_atom = slab_begin * kDecSlabAtoms + 2 * wave + (q >> 1);   // compute role: atom of slab 0
  const int ld_atom = slab_begin * kDecSlabAtoms + (la_c >> 3);             // A loader role: atom of slab 0
  int parity = 0;
  int cur_tile = blockIdx.x, cur_slab = 0;
  // rows of this lane beyond N contribute nothing: their scale bytes are masked to zero (per tile, not per item)
  uint32_t nmask0 = 0, nmask1 = 0;
  auto tile_masks = [&]() __attribute__((always_inline)) {
    const int row0 = (cur_tile >> 2) * 128 + (cur_tile & 3) * 8 + cm_rowpart;
    nmask0 = row0 < p.N ? 0xffffffffu : 0u;
    nmask1 = row0 + 4 < p.N ? 0xffffffffu : 0u;
  };
  tile_masks();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

  // One item: stage `r` into LDS, refill `r` with the item three ahead, barrier, multiply; close the tile after
  // its last slab (cross-wave reduction through LDS, fused epilogue).
  auto step = [&](DecodeRegs<kADw>& r) __attribute__((always_inline)) {
    unsigned char* buf = smem + parity * lds_buf;
    parity ^= 1;
    *reinterpret_cast<u32x4*>(buf + wrb_off) = r.b0;
    *reinterpret_cast<u32x4*>(buf + wrb_off + 16 * kDecBStride) = r.b1;
    {
      const bool a_live = ld_atom + cur_slab * kDecSlabAtoms < atoms_k;   // false only in the partial tail slab
      auto stage_a = [&](int i, uint32_t codes, uint32_t sf4) {
        const Frag8 f = dequant8(codes, sf_pair_at(a_live ? sf4 : 0u, sa_sh));
        if (4 * i + la_m < p.M) *reinterpret_cast<uint4*>(buf + wra_off + 4 * i * kDecAStride) = f.u;
      };
      stage_a(0, r.a0, r.sa0);
      if constexpr (kADw >= 2) stage_a(1, r.a1, r.sa1);
      if constexpr (kADw >= 4) { stage_a(2, r.a2, r.sa2); stage_a(3, r.a3, r.sa3); }
    }
    uint32_t bs0 = r.sb0, bs1 = r.sb1;
    issue_next(r);                                             // refill: 3 items ahead
    __syncthreads();
    const bool live = lane_atom + cur_slab * kDecSlabAtoms < atoms_k;     // false only in the partial tail slab
    bs0 = live ? bs0 & nmask0 : 0u;
    bs1 = live ? bs1 & nmask1 : 0u;
    const uint4 bq0 = *reinterpret_cast<const uint4*>(buf + rdb_off);
    const uint4 bq1 = *reinterpret_cast<const uint4*>(buf + rdb_off + 16 * kDecBStride);
    Frag8 a0, a1, a2, a3;
    a0.u = *reinterpret_cast<const uint4*>(buf + rda_off);
    a1.u = *reinterpret_cast<const uint4*>(buf + rda_off + 16);
    a2.u = *reinterpret_cast<const uint4*>(buf + rda_off + 32);
    a3.u = *reinterpret_cast<const uint4*>(buf + rda_off + 48);
    // weights are the MFMA A operand (rows i = rho), activations the B operand (cols j = token)
    {
      const f16x2 s0 = sf_pair_at(bs0, sh), s1 = sf_pair_at(bs0, sh + 8);
      const Frag8 b0 = dequant8(bq0.x, s0), b1 = dequant8(bq0.y, s0), b2 = dequant8(bq0.z, s1), b3 = dequant8(bq0.w, s1);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc0, 0, 0, 0);
    }
    {
      const f16x2 s0 = sf_pair_at(bs1, sh), s1 = sf_pair_at(bs1, sh + 8);
      const Frag8 b0 = dequant8(bq1.x, s0), b1 = dequant8(bq1.y, s0), b2 = dequant8(bq1.z, s1), b3 = dequant8(bq1.w, s1);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc1, 0, 0, 0);
    }
    if (++cur_slab == nslabs) {                                // wave-uniform
      // ---- tile done: cross-wave reduction; lane holds C[rho = 4q + e][token = rl] of both row blocks
      float* mine = red + (wave * 64 + lane) * 8;
      *reinterpret_cast<float4*>(mine) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
      *reinterpret_cast<float4*>(mine + 4) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
      __syncthreads();
      if (wave == 0) {
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < kDecWaves; ++w) {
          const float4 u = *reinterpret_cast<const float4*>(red + (w * 64 + lane) * 8);
          const float4 v = *reinterpret_cast<const float4*>(red + (w * 64 + lane) * 8 + 4);
          s0[0] += u.x; s0[1] += u.y; s0[2] += u.z; s0[3] += u.w;
          s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
        }
        // rho = 4q + e of block b is row 32q + 8t + 4b + e of the super-tile: 8 consecutive columns per lane
        const int nn = (cur_tile >> 2) * 128 + (cur_tile & 3) * 8 + q * 32;
        if (kEpi == kEpiSiluMul) {
          // columns nn..nn+7 = four (gate, up) pairs -> four activations of token rl; never split (launcher); N % 8 == 0
          uint32_t mx = 0;
          if (m_ok && nn < p.N) {
            const uint32_t a0 = silu_mul_bf16(f32_to_bf16_bits(alpha * s0[0]), f32_to_bf16_bits(alpha * s0[1]));
            const uint32_t a1 = silu_mul_bf16(f32_to_bf16_bits(alpha * s0[2]), f32_to_bf16_bits(alpha * s0[3]));
            const uint32_t a2 = silu_mul_bf16(f32_to_bf16_bits(alpha * s1[0]), f32_to_bf16_bits(alpha * s1[1]));
            const uint32_t a3 = silu_mul_bf16(f32_to_bf16_bits(alpha * s1[2]), f32_to_bf16_bits(alpha * s1[3]));
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.D) + ((uint32_t)rl * (uint32_t)(p.N >> 1) + (uint32_t)(nn >> 1))) =
                make_uint2(a0 | (a1 << 16), a2 | (a3 << 16));
            mx = max(max(a0 & 0x7fffu, a1 & 0x7fffu), max(a2 & 0x7fffu, a3 & 0x7fffu));
          }
#pragma unroll
          for (int sh2 = 32; sh2 > 0; sh2 >>= 1) mx = max(mx, (uint32_t)__shfl_down((int)mx, sh2, 64));
          if (lane == 0) p.slots[cur_tile] = mx;
        } else if (m_ok) {
          if (gridDim.y == 1) {
            if (nn < p.N) finish4<uint32_t>(p, alpha, rl, nn, s0);
            if (nn + 4 < p.N) finish4<uint32_t>(p, alpha, rl, nn + 4, s1);
          } else {
            float* o = p.partial + ((uint32_t)(blockIdx.y * p.M + rl) * (uint32_t)p.N + (uint32_t)nn);
            for (int e = 0; e < 4; ++e) if (nn + e < p.N) o[e] = s0[e];
            for (int e = 0; e < 4; ++e) if (nn + 4 + e < p.N) o[4 + e] = s1[e];
          }
        }
      }
      // `red` is next written after at least one more __syncthreads (the next tile's first item), so no barrier here
      cur_slab = 0;
      cur_tile += G;
      tile_masks();
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  int it = 0;
#pragma unroll 1
  for (; it + 3 <= nitems; it += 3) {
    step(r0);
    step(r1);
    step(r2);
  }
  if (it < nitems) step(r0);
  if (it + 1 < nitems) step(r1);
}

// split-K (over whole slabs) only when the tiles alone leave most CUs idle
static void decode_split(int64_t N, int64_t K, int* splitk, int* slabs_per_split) {
  const int64_t tiles = ((N + 127) / 128) * 4;
  const int nslabs = (int)((K + kDecSlabK - 1) / kDecSlabK);
  int s = 1;
  if ((N % 4) == 0) {
    while (tiles * s < 192 && nslabs / (s * 2) >= 8 && s < 16) s *= 2;
  }
  static const int forced = getenv("ARCQ_DECODE_SPLIT") ? atoi(getenv("ARCQ_DECODE_SPLIT")) : 0;   // tuning only
  if (forced > 0 && (N % 4) == 0) s = forced < nslabs ? forced : nslabs;
  const int per = (nslabs + s - 1) / s;
  *splitk = (nslabs + per - 1) / per;          // drop empty splits
  *slabs_per_split = per;
}

int64_t gemm_decode_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int s, per;
  decode_split(N, K, &s, &per);
  return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

template <int kADw, int kEpi>
static int launch_decode(const DecodeParams& p, int splitk, hipStream_t stream) {
  static LdsOptIn lds_opt;             // per kernel instantiation, per device
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_decode_kernel<kADw, kEpi>), lds_opt, decode_lds_bytes(4 * kADw),
                                  "arcq_gemm_nvfp4 (decode)"))
    return rc;
  // persistent workgroups, each walks tiles blockIdx.x, +grid, ...: two 8-wave groups per CU while their LDS fits
  static const int forced_grid = getenv("ARCQ_DECODE_GRID") ? atoi(getenv("ARCQ_DECODE_GRID")) : 0;
  const int per_cu = 2 * decode_lds_bytes(p.M) <= 160 * 1024 ? 2 : 1;
  const int max_wg = forced_grid > 0 ? forced_grid : 256 * per_cu;
  const int per_split = max_wg / splitk > 0 ? max_wg / splitk : 1;
  const int gx = p.tiles < per_split ? p.tiles : per_split;
  hipLaunchKernelGGL((gemm_decode_kernel<kADw, kEpi>), dim3((unsigned)gx, (unsigned)splitk), dim3(kDecThreads), decode_lds_bytes(p.M), stream, p);
  return ARCQ_OK;
}

int64_t gemm_decode_silu_slots(int64_t M, int64_t N, int64_t K) {
  (void)M; (void)K;
  return ((N + 127) / 128) * 4;              // one per 32-row tile
}

int gemm_decode(const GemmArgs& a, hipStream_t stream) {
  int splitk, per;
  decode_split(a.N, a.K, &splitk, &per);
  if (a.epilogue == kEpiSiluMul) {             // the fused epilogue needs whole sums: one split, all slabs
    splitk = 1;
    per = (int)((a.K + kDecSlabK - 1) / kDecSlabK);
  }
  DecodeParams p;
  p.A = a.A; p.B = a.B; p.SFA = a.SFA; p.SFB = a.SFB; p.D = a.D;
  p.partial = reinterpret_cast<float*>(a.workspace);
  p.alpha_dev = a.alpha_dev; p.bias = a.bias; p.residual = a.residual;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  p.tiles = ((a.N + 127) / 128) * 4;
  p.slabs_per_split = per;
  p.epi = a.epilogue; p.slots = a.absmax_slots;
  if (splitk > 1) {
    const int64_t need = (int64_t)splitk * a.M * a.N * (int64_t)sizeof(float);
    if (!a.workspace || a.workspace_bytes < need)
      return fail(ARCQ_ERR_WORKSPACE, "arcq_gemm_nvfp4: split-K needs %lld B of workspace, got %lld", (long long)need,
                  (long long)a.workspace_bytes);
  }
  if ((int64_t)a.N * (a.K / 2) >= ((int64_t)1 << 32) || (int64_t)((a.N + 127) / 128) * 128 * (a.K / 16) >= ((int64_t)1 << 32))
    return fail(ARCQ_ERR_UNSUPPORTED, "arcq_gemm_nvfp4 (decode): operand larger than 4 GiB");
  int rc;
  if (a.epilogue == kEpiSiluMul)
    rc = a.M <= 4 ? launch_decode<1, kEpiSiluMul>(p, splitk, stream)
                  : a.M <= 8 ? launch_decode<2, kEpiSiluMul>(p, splitk, stream) : launch_decode<4, kEpiSiluMul>(p, splitk, stream);
  else
    rc = a.M <= 4 ? launch_decode<1, kEpiPlain>(p, splitk, stream)
                  : a.M <= 8 ? launch_decode<2, kEpiPlain>(p, splitk, stream) : launch_decode<4, kEpiPlain>(p, splitk, stream);
  if (rc != ARCQ_OK) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (decode): launch failed: %s", hipGetErrorString(e));
  if (splitk > 1) return gemm_splitk_finish(a, splitk, stream);
  return ARCQ_OK;
}

}  // namespace arcq
