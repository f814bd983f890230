// ARC-NVFP4 GEMM for decode on the reference layout (M <= 16) and the shapes between decode and prefill (16 < M <~ 1024): a register-tiled
// fp16-MFMA kernel WITHOUT an LDS operand stage, K split over the waves of a workgroup.
//
// Why not the LDS-tiled kernel (gemm_tile.hip) here: with few tokens the grid of 128 x 128 / 256 x 256 tiles does not fill 256 CUs, so
// that kernel splits K over workgroups (fp32 partial planes + a finish launch: 16.8 MB written and re-read at M = 256, N = 4096) and
// runs a K loop of one barrier per 64 K elements on tiles too small to cover it (measured, profiles/r03_midm_tile_sweep.jsonl:
// M = 256, N = K = 4096 takes 26-34 us under every one of 17 tile / split configurations; the arithmetic is 3.4 us).  And the
// LDS-transposing decode kernels (gemm_skinny.hip / gemm_decode.hip) spend ~92 instructions per 16 bytes of weights on their transpose.
// Here (reference: the one CUTLASS instantiation of kernels/src/nvfp4.cu:48-74 serves every M):
//   * a wave owns a (16 TM) token x (16 TN) row output tile -- 64 x 64 (4 x 4 MFMA 16x16x32 tiles) down to 16 x 16 for decode -- and
//     reads its operand fragments STRAIGHT from the reference layout into the MFMA operand registers: lane (r, c) of fragment i takes
//     the 32 packed bytes of row 16 i + r that hold scale-factor atom 4 s + c of quad-step s (64 K elements = 4 groups, their 4 scale
//     bytes are ONE aligned dword of the swizzled layout), i.e. the four lane groups of a fragment cover one full 128-byte line per
//     row and quad-step.  The K index a lane feeds into MFMA slot (c, j) is a fixed permutation of the true one, the SAME for both
//     operands, so every product meets its partner (the contraction is a sum over K: order inside the fp32 chain differs from the
//     tiled kernel, the set of products does not);
//   * no LDS and no barrier in the K loop; operands are BUFFER loads (descriptor + 32-bit lane offset + scalar step offset) requested one
//     step ahead (64 x 64 tiles, in halves rotating through three register sets) or up to three steps ahead (smaller tiles), every wait
//     counted (see the loop);
//   * the eight waves of a workgroup split K (KSPLIT = 8: one tile per workgroup -- 256 workgroups at M = 256, N = 4096 with 64 x 64,
//     at M <= 16 with 16 x 16), or tile a larger block with less K splitting (2 x 4 x 1 ... 1 x 1 x 8); the partial tiles meet in LDS
//     once, at the end, summed in a fixed order (deterministic), and the epilogue (alpha, bias, residual, bf16 rounding:
//     gemm_common.hpp finish4) is spread over the waves;
//   * a K tail that is not a multiple of 256 (KE = 64: one atom) runs as single-atom steps of a quarter of the MFMAs.
// Bound: with K split inside the workgroup every operand byte enters a CU once per tile row / column: (BM + BN) x K x 9/16 bytes per
// workgroup, delivered at 28-57 GB/s per CU for this access shape although 86-91 % of the requests hit L2 (profiles/r03_pmc_regtile.json),
// and the dequantisation is per wave (64 + 64 rows per 64 x 64 tile: 2 x the tiled kernel's share per MFMA): the kernel is for grids
// the tiled kernel cannot fill and for decode, not for prefill.  gemm_regtile_cfg below holds the measured crossovers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "arcq_internal.hpp"
#include "gemm_common.hpp"

namespace arcq {

struct RegTileParams {
  const uint8_t* A;
  const uint8_t* B;
  const uint8_t* SFA;
  const uint8_t* SFB;
  void* D;
  const float* alpha_dev;
  const uint16_t* bias;
  const uint16_t* residual;
  int M, N, K;
  float alpha_host;
  int out_dtype;
  int tiles_m, tiles_n;
#ifdef ARCQ_STREAM_STAMPS
  unsigned long long* stamps;
#endif
};

#ifdef ARCQ_STREAM_STAMPS
// DIAGNOSTIC build only (make diag, tools/regtile_stamps.py): s_memrealtime (100 MHz) of every wave's lane 0 at five points of the kernel
static unsigned long long* g_regtile_stamps = nullptr;
extern "C" void arcq_debug_set_regtile_stamps(void* p) { g_regtile_stamps = reinterpret_cast<unsigned long long*>(p); }
#define ARCQ_RT_STAMP(k)                                                                                          \
  do {                                                                                                            \
    if (p.stamps && lane == 0) {                                                                                  \
      unsigned long long t_;                                                                                      \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
      p.stamps[((size_t)blockIdx.x * 16 + wave) * 8 + (k)] = t_;                                                  \
    }                                                                                                             \
  } while (0)
#else
#define ARCQ_RT_STAMP(k) do { } while (0)
#endif

typedef uint32_t rt_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t rt_u32x2 __attribute__((ext_vector_type(2)));

template <int TM, int TN, int WAVES_M, int WAVES_N, int KSPLIT>
__global__ __launch_bounds__(WAVES_M* WAVES_N* KSPLIT * 64) void gemm_regtile_kernel(RegTileParams p) {
  constexpr int kWaves = WAVES_M * WAVES_N * KSPLIT;
  constexpr int BM = 16 * TM * WAVES_M, BN = 16 * TN * WAVES_N;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // XCD-aware tile order (as gemm_tile.hip): the blocks of one XCD get a contiguous range of tiles, M fastest
  const int ntiles = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q8 = ntiles >> 3, r8 = ntiles & 7, x = bid & 7, j = bid >> 3;
    bid = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + j;
  }
  const int tm = bid % p.tiles_m, tn = bid / p.tiles_m;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: K ranges and loops are wave-uniform
  ARCQ_RT_STAMP(0);
  const int ks = wave % KSPLIT, wn = (wave / KSPLIT) % WAVES_N, wm = wave / (KSPLIT * WAVES_N);
  const int r = lane & 15, c = lane >> 4;
  const int m0 = tm * BM + wm * 16 * TM, n0 = tn * BN + wn * 16 * TN;
  const int half_k = p.K >> 1, atoms_k = p.K >> 6;

  // this wave's share of K: quad-steps [qb, qe) and, of the K % 256 tail atoms, those with index % KSPLIT == KSPLIT - 1 - ks
  const int Q = atoms_k >> 2, R = atoms_k & 3;
  const int qb = (ks * Q) / KSPLIT, qe = ((ks + 1) * Q) / KSPLIT;

  // per-fragment 32-bit byte offsets (the launcher keeps every operand below 2 GiB): rows beyond the matrix are clamped and simply
  // computed -- output element (m, n) depends on row m of A and row n of B only, and the epilogue stores nothing outside the matrix
  uint32_t a_off[TM], a_sfo[TM], b_off[TN], b_sfo[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rc = min(m0 + 16 * i + r, p.M - 1);
    a_off[i] = (uint32_t)rc * (uint32_t)half_k + c * 32;
    a_sfo[i] = (uint32_t)sf_atom_offset(rc, c, atoms_k);
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int rc = min(n0 + 16 * t + r, p.N - 1);
    b_off[t] = (uint32_t)rc * (uint32_t)half_k + c * 32;
    b_sfo[t] = (uint32_t)sf_atom_offset(rc, c, atoms_k);
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one quad-step of one operand fragment = two 16-byte halves (slices 0-3 / 4-7) and the atom's four scale bytes.  A step requests the
  // next step's first halves + scales at its start and the second halves at its middle, when its own first halves are dead: every
  // load has a full step of lead and at most 3.5 halves per fragment are live (two full stages do not fit 256 registers beside the
  // 64 accumulators: 412 bytes of scratch, measured in the ISA)
  constexpr bool kSliceFence = TM * TN >= 16;     // keep hipcc from dequantising a slice ahead (registers)
  struct Half { rt_u32x4 a[TM], b[TN]; };
  struct Scales { uint32_t a[TM], b[TN]; };
  // buffer loads: descriptor (scalar) + the fragment's 32-bit lane offset + the step's scalar offset -- no 64-bit address is ever formed in
  // vector registers.  (With global loads hipcc built the addresses in registers that were still the targets of loads in flight and had to
  // wait for ALL of them before it could request the next step: M = 256 19.6 us, loads and multiplies one after the other.)
  const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A), 0, p.M * half_k, 0x00020000);
  const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.B), 0, p.N * half_k, 0x00020000);
  const auto rs_sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.SFA), 0, ((p.M + 127) >> 7) * atoms_k * 512, 0x00020000);
  const auto rs_sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.SFB), 0, ((p.N + 127) >> 7) * atoms_k * 512, 0x00020000);
  auto load_lo = [&](Half& h, Scales& sc, int q, uint32_t drop = 0u) __attribute__((always_inline)) {     // + the step's scale bytes: needed from its first slice
#ifdef ARCQ_EXPERIMENT_RT_NOLOAD
    // TIMING EXPERIMENT ONLY (results are WRONG): no operand is loaded, the multiply runs on register contents
#pragma unroll
    for (int i = 0; i < TM; ++i) { h.a[i] = rt_u32x4{(uint32_t)q, a_off[i], 0x12345678u, 0x9abcdef0u}; sc.a[i] = 0x38383838u; }
#pragma unroll
    for (int t = 0; t < TN; ++t) { h.b[t] = rt_u32x4{(uint32_t)q, b_off[t], 0x12345678u, 0x9abcdef0u}; sc.b[t] = 0x38383838u; }
    return;
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      h.a[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_off[i] | drop, q * 128, 0);
      sc.a[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_sa, a_sfo[i] | drop, q * 2048, 0);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      h.b[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_off[t] | drop, q * 128, 0);
      sc.b[t] = __builtin_amdgcn_raw_buffer_load_b32(rs_sb, b_sfo[t] | drop, q * 2048, 0);
    }
  };
  auto load_hi = [&](Half& h, int q, uint32_t drop = 0u) __attribute__((always_inline)) {
#ifdef ARCQ_EXPERIMENT_RT_NOLOAD
#pragma unroll
    for (int i = 0; i < TM; ++i) h.a[i] = rt_u32x4{(uint32_t)q, a_off[i], 0x12345678u, 0x9abcdef0u};
#pragma unroll
    for (int t = 0; t < TN; ++t) h.b[t] = rt_u32x4{(uint32_t)q, b_off[t], 0x12345678u, 0x9abcdef0u};
    return;
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i) h.a[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (a_off[i] | drop) + 16, q * 128, 0);
#pragma unroll
    for (int t = 0; t < TN; ++t) h.b[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (b_off[t] | drop) + 16, q * 128, 0);
  };
  // four MFMA K slices: slice j = dword j of the half, scale byte 2 * half + j / 2.  Weights are the MFMA A operand (rows = weight
  // rows), activations the B operand (columns = tokens): a lane ends up with four consecutive output columns n of one token
  auto mma_half = [&](const Half& h, const Scales& sc, int half) __attribute__((always_inline)) {
#ifdef ARCQ_EXPERIMENT_RT_NOCOMPUTE
    // TIMING EXPERIMENT ONLY (results are WRONG): the operands are awaited and folded into one accumulator, nothing is multiplied
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[0][0][0] += __builtin_bit_cast(float, h.a[i][0] ^ h.a[i][1] ^ h.a[i][2] ^ h.a[i][3] ^ (sc.a[i] & 0x7f7f7f7f));
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[0][0][1] += __builtin_bit_cast(float, h.b[t][0] ^ h.b[t][1] ^ h.b[t][2] ^ h.b[t][3] ^ (sc.b[t] & 0x7f7f7f7f));
    (void)half;
    return;
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Frag8 xa[TM], xb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) xa[i] = dequant8(h.a[i][j], sf_pair_at(sc.a[i], 16 * half + 8 * (j >> 1)));
#pragma unroll
      for (int t = 0; t < TN; ++t) xb[t] = dequant8(h.b[t][j], sf_pair_at(sc.b[t], 16 * half + 8 * (j >> 1)));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[t].v, xa[i].v, acc[i][t], 0, 0, 0);
      if (kSliceFence) __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (TM * TN >= 16) {
    // three sets of halves rotate: step s reads lo(s) = S[2s % 3] and hi(s) = S[(2s + 1) % 3], requests lo(s + 1) into the third set at its
    // start and hi(s + 1) into lo(s)'s set at its middle -- a period of three steps, unrolled, so that no set is ever moved.  Every step
    // ISSUES its requests, also the wave's last one: there with bit 31 set in every lane offset, which the descriptor's range check drops
    // without touching memory.  Exits leave the loop and never rejoin it, so on the one straight path through it the number of loads in
    // flight at every wait is a constant and hipcc waits for exactly the operands of the slice it is about to multiply (with `if (more) load`
    // inside the step it drained the queue at every join; a peeled copy of the last step per phase instead cost 1.2 KB of scratch)
    constexpr uint32_t kDrop = 0x80000000u;
    Half s0, s1, s2;
    Scales c0, c1, c2;
    auto step = [&](const Half& lo, const Half& hi, const Scales& sc, Half& lon, Half& hin, Scales& scn, int q) __attribute__((always_inline)) {
      const uint32_t drop = q + 1 < qe ? 0u : kDrop;                 // wave-uniform (scalar)
      load_lo(lon, scn, q + 1, drop);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(lo, sc, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_hi(hin, q + 1, drop);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(hi, sc, 1);
      __builtin_amdgcn_sched_barrier(0);
    };
    if (qb < qe) {
      load_lo(s0, c0, qb, 0u);
      load_hi(s1, qb, 0u);
#ifdef ARCQ_STREAM_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (diagnostic build: the first step's operands have arrived)
      ARCQ_RT_STAMP(1);
#endif
#pragma unroll 1
      for (int q = qb;; q += 3) {
        step(s0, s1, c0, s2, s0, c1, q);
        if (q + 1 >= qe) break;
        step(s2, s0, c1, s1, s2, c2, q + 1);
        if (q + 2 >= qe) break;
        step(s1, s2, c2, s0, s1, c0, q + 2);
        if (q + 3 >= qe) break;
      }
    }
  } else {
    // smaller wave tiles: a ring of whole steps, kStages - 1 of them in flight while one is multiplied (a step of a 2 x 2 tile is 32 MFMAs:
    // one step of lead does not cover a loaded memory latency; the registers the 64 accumulators would take hold the ring instead)
    constexpr int kStages = TM * TN <= 4 ? 4 : 3, kLead = kStages - 1;
    Half rlo[kStages], rhi[kStages];
    Scales rsc[kStages];
#pragma unroll
    for (int d = 0; d < kLead; ++d)
      if (qb + d < qe) {
        load_lo(rlo[d], rsc[d], qb + d);
        load_hi(rhi[d], qb + d);
      }
#ifdef ARCQ_STREAM_STAMPS
    ARCQ_RT_STAMP(1);
#endif
#pragma unroll 1
    for (int q = qb; q < qe; q += kStages) {
#pragma unroll
      for (int u = 0; u < kStages; ++u) {
        if (q + u < qe) {
          if (q + u + kLead < qe) {
            load_lo(rlo[(u + kLead) % kStages], rsc[(u + kLead) % kStages], q + u + kLead);
            load_hi(rhi[(u + kLead) % kStages], q + u + kLead);
          }
          __builtin_amdgcn_sched_barrier(0);
          mma_half(rlo[u], rsc[u], 0);
          mma_half(rhi[u], rsc[u], 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  ARCQ_RT_STAMP(2);
  // ---- tail atoms (K % 256 != 0): lane (r, c) takes group c of the atom (8 bytes), its scale is byte c of the atom's dword
#pragma unroll 1
  for (int e = 0; e < R; ++e) {
    if ((e % KSPLIT) != KSPLIT - 1 - ks) continue;                 // wave-uniform
    const int atom = 4 * Q + e;
    Frag8 xa[2][TM], xb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const rt_u32x2 w = *reinterpret_cast<const rt_u32x2*>((p.A + (size_t)atom * 32) + (a_off[i] - c * 24));
      const uint32_t sf = *reinterpret_cast<const uint32_t*>((p.SFA + (size_t)atom * 512) + (a_sfo[i] - c * 512));
      const f16x2 s2 = sf_pair_at(sf, 8 * c);
      xa[0][i] = dequant8(w.x, s2);
      xa[1][i] = dequant8(w.y, s2);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const rt_u32x2 w = *reinterpret_cast<const rt_u32x2*>((p.B + (size_t)atom * 32) + (b_off[t] - c * 24));
      const uint32_t sf = *reinterpret_cast<const uint32_t*>((p.SFB + (size_t)atom * 512) + (b_sfo[t] - c * 512));
      const f16x2 s2 = sf_pair_at(sf, 8 * c);
      xb[0][t] = dequant8(w.x, s2);
      xb[1][t] = dequant8(w.y, s2);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[j][t].v, xa[j][i].v, acc[i][t], 0, 0, 0);
  }

  ARCQ_RT_STAMP(3);
  // ---- the K splits meet in LDS: [wave][tile][lane] float4; split ks then finishes tiles ks, ks + KSPLIT, ... of its (wm, wn) block
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);
  auto finish_tile = [&](int i, int t, const f32x4& s) __attribute__((always_inline)) {
    const int m = m0 + 16 * i + r, n = n0 + 16 * t + 4 * c;
    if (m < p.M && n < p.N) {
      const float d[4] = {s[0], s[1], s[2], s[3]};
      finish4<size_t>(p, alpha, m, n, d);                   // (8-byte operand loads when N % 4 == 0: gemm_common.hpp)
    }
  };
  if constexpr (KSPLIT == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t) finish_tile(i, t, acc[i][t]);
  } else {
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t) red[(wave * TM * TN + i * TN + t) * 64 + lane] = acc[i][t];
    __syncthreads();
    const int wbase = wave - ks;                                 // first wave of this (wm, wn) block
#pragma unroll
    for (int x = 0; x < (TM * TN + KSPLIT - 1) / KSPLIT; ++x) {
      const int tile = ks + x * KSPLIT;                          // wave-uniform
      if (tile < TM * TN) {
        f32x4 s = red[(wbase * TM * TN + tile) * 64 + lane];
#pragma unroll
        for (int k2 = 1; k2 < KSPLIT; ++k2) {
          const f32x4 v = red[((wbase + k2) * TM * TN + tile) * 64 + lane];
          s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        finish_tile(tile / TN, tile % TN, s);
      }
    }
  }
  ARCQ_RT_STAMP(4);
  (void)kWaves;
}

// ---- launcher ---------------------------------------------------------------------------------------------------------------
// configurations (ARCQ_REGTILE_CFG forces one, tuning only): token tiles x row tiles per wave, waves along M / N / K
//   1 = 4x4 1x1x8 (64 x 64 per workgroup)   2 = 4x4 1x2x4 (64 x 128)   3 = 4x4 2x2x2 (128 x 128)   4 = 4x4 2x4x1 (128 x 256)
//   5 = 2x4 1x1x8 (32 x 64)                 6 = 4x2 1x1x8 (64 x 32)     7 = 2x2 1x1x8 (32 x 32)     8 = 4x4 2x1x4 (128 x 64)
//   9 = 1x1 1x1x8 (16 x 16)                 10 = 1x2 1x1x8 (16 x 32)    11 = 1x4 1x1x8 (16 x 64)    (decode batches on the reference layout)
//   12 = 2x1 1x1x8 (32 x 16)                13 = 4x1 1x1x8 (64 x 16)
struct RegCfg { int id, bm, bn, ksplit, tm, tn; };
static constexpr RegCfg kRegCfgs[] = {{1, 64, 64, 8, 4, 4}, {2, 64, 128, 4, 4, 4}, {3, 128, 128, 2, 4, 4}, {4, 128, 256, 1, 4, 4},
                                      {5, 32, 64, 8, 2, 4}, {6, 64, 32, 8, 4, 2},  {7, 32, 32, 8, 2, 2},   {8, 128, 64, 4, 4, 4},
                                      {9, 16, 16, 8, 1, 1}, {10, 16, 32, 8, 1, 2}, {11, 16, 64, 8, 1, 4},
                                      {12, 32, 16, 8, 2, 1}, {13, 64, 16, 8, 4, 1}};

static int regtile_override() {              // ARCQ_REGTILE_CFG: 0 = by shape, -1 = never (the tiled kernel), n = forced configuration
  static const int v = getenv("ARCQ_REGTILE_CFG") ? atoi(getenv("ARCQ_REGTILE_CFG")) : 0;
  return v;
}

template <int TM, int TN, int WAVES_M, int WAVES_N, int KSPLIT>
static int launch_regtile(const GemmArgs& a, hipStream_t stream) {
  RegTileParams p;
  p.A = a.A; p.B = a.B; p.SFA = a.SFA; p.SFB = a.SFB; p.D = a.D;
  p.alpha_dev = a.alpha_dev; p.bias = a.bias; p.residual = a.residual;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  constexpr int BM = 16 * TM * WAVES_M, BN = 16 * TN * WAVES_N, kWaves = WAVES_M * WAVES_N * KSPLIT;
  p.tiles_m = (a.M + BM - 1) / BM;
  p.tiles_n = (a.N + BN - 1) / BN;
#ifdef ARCQ_STREAM_STAMPS
  p.stamps = g_regtile_stamps;
#endif
  const int lds = KSPLIT > 1 ? kWaves * TM * TN * 64 * 16 : 0;
  auto kern = gemm_regtile_kernel<TM, TN, WAVES_M, WAVES_N, KSPLIT>;
  static LdsOptIn lds_opt;
  if (lds > 0)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_opt, lds, "arcq_gemm_nvfp4 (regtile)")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(kWaves * 64), lds, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (regtile): launch failed: %s", hipGetErrorString(e));
  return ARCQ_OK;
}

// 0 = the tiled kernel serves this shape, else the configuration: the SMALLEST tile whose grid is one round of the chip (<= 256 workgroups;
// >= 96, or the tiled kernel's split-K fills the chip better), up to ~1.6 * 10^10 of M N K, where the tiled kernel's one dequantisation per
// workgroup starts to win.  Measured (tools/midm_tile_sweep.py, profiles/r03_midm_regtile_sweep.jsonl; graph replay, us, this kernel / the tiled
// kernel): N = K = 4096: M = 32 7.7 (32 x 16 tiles; 32 x 32: 9.5) / 15.7, 64 9.1 / 20.6, 128 11.3 / 25.8, 256 16.8 / 27.2, 512 27.5 / 32.6, 1024 46.8 / 46.5;
// N = 10752, K = 3584: M = 32 11.4 / 20.1, 64 15.4 / 25.9, 128 25.9 / 28.7, 256 40.8 / 43.9, 512 75.6 / 65.0; N = 3584, K = 18944: M = 64 25.2 /
// 34.1, 128 35.8 / 46.2, 256 54.3 / 59.6, 512 94.6 / 92.4, 1024 172 / 156.
int gemm_regtile_cfg(int64_t M, int64_t N, int64_t K, int epilogue) {
  if (epilogue != kEpiPlain || (K & 63) != 0) return 0;
  if ((int64_t)max(M, N) * (K / 2) >= ((int64_t)1 << 31) || ((max(M, N) + 127) / 128) * (K / 64) * 512 >= ((int64_t)1 << 31)) return 0;   // 32-bit offsets, buffer descriptors
  const int ov = regtile_override();
  if (ov < 0) return 0;
  if (ov > 0) {
    for (const RegCfg& c : kRegCfgs)
      if (c.id == ov) return ov;
    return 0;
  }
  if (M <= 16) {
    // decode on the REFERENCE layout: 16 x 16 tiles, K over the 8 waves (N / 16 workgroups).  HBM-cold, graph replay, us, this kernel / the
    // LDS-transposing decode kernels (gemm_skinny.hip, gemm_decode.hip) -- profiles/r03_decode_reference_layout_regtile.txt: N = K = 4096
    // (BASELINE config[1]) M = 1 5.55 / 6.74, M = 16 6.75 / 8.30; 10752 x 3648 9.6 / 11.1; 1024 x 4160 5.1 / 6.0; 14336 x 4160 12.3 / 12.5; it
    // loses where a wave has many short rounds or a very long K at M <= 8: 37888 x 3648 M = 1 25.7 / 20.7, 3584 x 19008 M = 1 14.6 / 13.5
    if (M > 8 || (N <= 16384 && K <= 8448)) return 9;
    return 0;
  }
  if ((double)M * (double)N * (double)K > 1.6e10) return 0;
  static const int order[] = {12, 7, 6, 5, 1, 8, 2, 3};       // by tile area, token-narrow before row-narrow
  for (int id : order)
    for (const RegCfg& c : kRegCfgs) {
      if (c.id != id || (c.bm > 32 && c.bm >= 2 * M)) continue;       // a tile twice as tall as the batch multiplies padding
      const int64_t wgs = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
      if (wgs <= 256) return wgs >= 96 ? id : 0;
    }
  // no tile covers the problem in one round (very wide weights): up to three rounds of 32 x 64 / 64 x 64 tiles still beat the tiled kernel's
  // split-K for decode batches (37888 x 3648, graph replay: M = 32 34.5 against 39.6 us, M = 64 48.0 against 52.3; from M = 128 it loses)
  if (M <= 64 && ((N + 63) / 64) <= 768) return M <= 32 ? 5 : 1;
  return 0;
}

int gemm_regtile(const GemmArgs& a, int cfg, hipStream_t stream) {
  switch (cfg) {
    case 1: return launch_regtile<4, 4, 1, 1, 8>(a, stream);
    case 2: return launch_regtile<4, 4, 1, 2, 4>(a, stream);
    case 3: return launch_regtile<4, 4, 2, 2, 2>(a, stream);
    case 4: return launch_regtile<4, 4, 2, 4, 1>(a, stream);
    case 5: return launch_regtile<2, 4, 1, 1, 8>(a, stream);
    case 6: return launch_regtile<4, 2, 1, 1, 8>(a, stream);
    case 7: return launch_regtile<2, 2, 1, 1, 8>(a, stream);
    case 8: return launch_regtile<4, 4, 2, 1, 4>(a, stream);
    case 9: return launch_regtile<1, 1, 1, 1, 8>(a, stream);
    case 10: return launch_regtile<1, 2, 1, 1, 8>(a, stream);
    case 11: return launch_regtile<1, 4, 1, 1, 8>(a, stream);
    case 12: return launch_regtile<2, 1, 1, 1, 8>(a, stream);
    case 13: return launch_regtile<4, 1, 1, 1, 8>(a, stream);
    default: return fail(ARCQ_ERR_UNSUPPORTED, "arcq_gemm_nvfp4 (regtile): unknown configuration %d", cfg);
  }
}

}  // namespace arcq
