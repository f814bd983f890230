// ARC-NVFP4 GEMM for prefill shapes (M > 16): LDS-tiled fp16-MFMA kernel for gfx950.
//
// Replaces the CUTLASS sm120 block-scaled GEMM instantiation of the reference
// (kernels/src/nvfp4.cu:10-33,48-74: 128x128x128 tiles, mma.sync block_scale) with a CDNA4 design:
//
//   HBM/L2 --(packed e2m1 + ue4m3 bytes, 16 B per lane)--> VGPR --dequantise once per block-->
//   fp16 tiles in LDS (XOR-swizzled 128-byte rows) --ds_read_b128--> v_mfma_f32_16x16x32_f16
//
// Each operand byte crosses HBM/L2 in its 4.5-bit form (3.6x less traffic than an fp16 GEMM) and is
// expanded exactly once per workgroup; the MFMA mainloop is then an ordinary fp16 contraction whose
// products are exact (gemm_common.hpp), fp32 accumulate, alpha / bias / bf16 rounding fused in the
// epilogue.  The bound is the dense fp16/bf16 MFMA rate (~2.5 PFLOP/s), see DESIGN.md.
//
// Tile: BM x BN x 64, kThreads = 256 (2x2 wave64, each 64x64 = 4x4 MFMA tiles) or 512 (2x4 waves on
// 128x256).  Double-buffered LDS, one barrier per K step: the global loads of step k+1 are issued
// before the MFMAs of step k and written to the other buffer after them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "arcq_internal.hpp"
#include "gemm_common.hpp"
#include "gemm_tile_common.hpp"

namespace arcq {

#ifdef ARCQ_STREAM_STAMPS
// DIAGNOSTIC build only: the in-kernel clock of the K loop = delta s_memtime / delta s_memrealtime x 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back" item 6), stamped once around the loop by wave 0 of every workgroup into a buffer
// nothing else reads (tools/tile_clock.py).  The product library has no stamps.
static unsigned long long* g_tile_stamps = nullptr;
extern "C" void arcq_debug_set_tile_stamps(void* p) { g_tile_stamps = reinterpret_cast<unsigned long long*>(p); }
#define ARCQ_TILE_STAMP(k)                                                                                     \
  do {                                                                                                         \
    if (p.stamps && tid == 0) {                                                                                \
      unsigned long long t0_, t1_;                                                                             \
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_), "=s"(t1_)::"memory"); \
      p.stamps[(size_t)blockIdx.x * 4 + 2 * (k)] = t0_;                                                        \
      p.stamps[(size_t)blockIdx.x * 4 + 2 * (k) + 1] = t1_;                                                    \
    }                                                                                                          \
  } while (0)
#else
#define ARCQ_TILE_STAMP(k) do { } while (0)
#endif

// kStagger (8-wave tiles): the two waves of a SIMD (w and w + 4) run half a step apart -- waves 0-3 multiply step k and
// then stage step k+1, waves 4-7 stage step k+2 FIRST (into the buffer step k was just read from, after the barrier)
// and then multiply step k+1 -- so that one wave's dequantise/ds_write phase overlaps the other's MFMA phase instead
// of both waves of a SIMD leaving the matrix pipe idle together.  ONE loop body with a barrier on either side of the
// staging block, each taken by one group (every wave still meets one barrier per step): no code is duplicated.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool kMfma32, int kEpi, bool kStagger, bool kPipe>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_tile_kernel(TileParams p) {
  constexpr int kThreads = WAVES_M * WAVES_N * 64;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;     // wave tile
  constexpr int kT = kMfma32 ? 32 : 16;                   // MFMA tile edge
  constexpr int TM = WM / kT, TN = WN / kT;               // MFMA tiles per wave
  using acc_t = typename std::conditional<kMfma32, f32x16, f32x4>::type;
  // 16-byte staging units per thread; a short tile edge (BM*2 or BN*2 < kThreads) leaves the upper threads without a unit
  constexpr int A_UNITS = (BM * 2 + kThreads - 1) / kThreads, B_UNITS = (BN * 2 + kThreads - 1) / kThreads;
  constexpr bool A_PARTIAL = (BM * 2) % kThreads != 0, B_PARTIAL = (BN * 2) % kThreads != 0;
  static_assert(!A_PARTIAL || A_UNITS == 1, "a partial A pass is a single pass");
  static_assert(!B_PARTIAL || B_UNITS == 1, "a partial B pass is a single pass");
  constexpr int A_TILE = BM * kRowBytes, B_TILE = BN * kRowBytes;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const lds_a0 = smem;                        // [buf][A | B]
  unsigned char* const lds_b0 = smem + A_TILE;
  unsigned char* const lds_a1 = smem + A_TILE + B_TILE;
  unsigned char* const lds_b1 = smem + 2 * A_TILE + B_TILE;

  // ---- XCD-aware tile order: blocks that land on one XCD (id % 8) get a contiguous range of tiles, and
  //      within it tiles walk M fastest in groups of 8 so concurrently resident blocks share B panels.
  const int ntiles = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x, split = 0;
  if (p.splits > 1) {
    split = bid / ntiles;
    bid -= split * ntiles;
  }
  {
    const int q8 = ntiles >> 3, r8 = ntiles & 7, x = bid & 7, j = bid >> 3;
    bid = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + j;     // bijective for any ntiles
  }
  constexpr int kGroupM = 8;
  const int per_group = kGroupM * p.tiles_n;
  const int grp = bid / per_group;
  const int first_m = grp * kGroupM;
  const int gsz = min(p.tiles_m - first_m, kGroupM);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WAVES_N, wc = wave % WAVES_N;
  const int half_k = p.K >> 1, atoms_k = p.K >> 6;
  const int a_begin = split * p.atoms_per_split, a_end = min(atoms_k, a_begin + p.atoms_per_split);   // never empty

  // ---- per-thread staging geometry (loop invariant)
  const uint8_t* a_q[A_UNITS];
  const uint8_t* a_sf[A_UNITS];
  uint32_t a_live[A_UNITS];
  int a_slot[A_UNITS][4];
#pragma unroll
  for (int u = 0; u < A_UNITS; ++u) {
    const int unit = tid + u * kThreads, r = A_PARTIAL ? min(unit >> 1, BM - 1) : unit >> 1, h = unit & 1;
    const int row = m0 + r, rc = row < p.M ? row : p.M - 1;
    a_q[u] = p.A + (size_t)rc * half_k + h * 16;
    a_sf[u] = p.SFA + sf_atom_offset(rc, 0, atoms_k) + h * 2;
    a_live[u] = row < p.M ? 0xffffu : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) a_slot[u][j] = kMfma32 ? lds_slot32(r, h * 4 + j) : lds_slot(r, h * 4 + j);
  }
  const uint8_t* b_q[B_UNITS];
  const uint8_t* b_sf[B_UNITS];
  uint32_t b_live[B_UNITS];
  int b_slot[B_UNITS][4];
#pragma unroll
  for (int u = 0; u < B_UNITS; ++u) {
    const int unit = tid + u * kThreads, r = B_PARTIAL ? min(unit >> 1, BN - 1) : unit >> 1, h = unit & 1;
    const int row = n0 + r, rc = row < p.N ? row : p.N - 1;
    b_q[u] = p.B + (size_t)rc * half_k + h * 16;
    b_sf[u] = p.SFB + sf_atom_offset(rc, 0, atoms_k) + h * 2;
    b_live[u] = row < p.N ? 0xffffu : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) b_slot[u][j] = kMfma32 ? lds_slot32(r, h * 4 + j) : lds_slot(r, h * 4 + j);
  }
  // fragment read offsets: one base per operand; tile i adds i*16 rows (the swizzle term is invariant under
  // +16 rows) and the second half of K flips bit 2 of the slot index, i.e. byte-offset bit 6
  const int fa_base = kMfma32 ? lds_slot32(wr * WM + (lane & 31), lane >> 5) : lds_slot(wr * WM + (lane & 15), lane >> 4);
  const int fb_base = kMfma32 ? lds_slot32(wc * WN + (lane & 31), lane >> 5) : lds_slot(wc * WN + (lane & 15), lane >> 4);
  acc_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < (kMfma32 ? 16 : 4); ++r) acc[i][j][r] = 0.f;

  Staged sa[A_UNITS], sb[B_UNITS];     // registers holding step kt+1 while step kt is multiplied
  auto load_step = [&](int atom) {
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) sa[u] = stage_load(a_q[u], a_sf[u], atom);
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u) sb[u] = stage_load(b_q[u], b_sf[u], atom);
  };
  auto store_step = [&](unsigned char* la, unsigned char* lb) {
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u)
      if (!A_PARTIAL || tid < BM * 2) stage_store(la, a_slot[u], sa[u], a_live[u]);
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u)
      if (!B_PARTIAL || tid < BN * 2) stage_store(lb, b_slot[u], sb[u], b_live[u]);
  };
  auto mma_step = [&](const unsigned char* la, const unsigned char* lb) {
    // weights (B of the GEMM) are the MFMA A operand, so a lane ends up with runs of 4 consecutive n
    if constexpr (kMfma32) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {          // 16 K elements per 32x32x16 MFMA: slot = 2*ks + (lane >> 5)
        Frag8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i].u = *reinterpret_cast<const uint4*>(la + (fa_base ^ (ks * 32)) + i * 32 * kRowBytes);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j].u = *reinterpret_cast<const uint4*>(lb + (fb_base ^ (ks * 32)) + j * 32 * kRowBytes);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j].v, fa[i].v, acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {          // 32 K elements per 16x16x32 MFMA: slot = 4*ks + (lane >> 4)
        Frag8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i].u = *reinterpret_cast<const uint4*>(la + (fa_base ^ (ks * 64)) + i * 16 * kRowBytes);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j].u = *reinterpret_cast<const uint4*>(lb + (fb_base ^ (ks * 64)) + j * 16 * kRowBytes);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j].v, fa[i].v, acc[i][j], 0, 0, 0);
      }
    }
  };
  // One K step, branch-free: multiply the tile in (ca, cb) while the registers of the next step are
  // dequantised into (na, nb) and the loads of the step after that are issued.  The K index is clamped,
  // so the last steps re-stage the final atom into a buffer nobody reads.
  const bool late = kStagger && wave >= 4;       // wave-uniform: this wave stages one step further ahead, after the barrier
  auto k_step = [&](int kt, unsigned char* ca, unsigned char* cb, unsigned char* na, unsigned char* nb) {
    Staged ta[A_UNITS], tb[B_UNITS];
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) ta[u] = sa[u];
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u) tb[u] = sb[u];
    load_step(min(kt + 2 + (late ? 1 : 0), a_end - 1));   // in flight during this whole step
    __builtin_amdgcn_sched_barrier(0);            // keep the loads at the top: hipcc otherwise sinks them to the barrier
    mma_step(ca, cb);
    if (kStagger && late) __syncthreads();        // every wave has multiplied step kt: its buffer may be overwritten
    unsigned char* const wa = late ? ca : na;
    unsigned char* const wb = late ? cb : nb;
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u)
      if (!A_PARTIAL || tid < BM * 2) stage_store(wa, a_slot[u], ta[u], a_live[u]);
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u)
      if (!B_PARTIAL || tid < BN * 2) stage_store(wb, b_slot[u], tb[u], b_live[u]);
    if (!kStagger || !late) __syncthreads();
  };

  // Hand-pipelined K step (16x16x32 MFMA): the compiler's own schedule waits for every pair of fragment reads right
  // after issuing it (LDS latency exposed every 8 MFMAs) and clusters the dequantisation.  Here a step is cut into
  // TM blocks of [2 fragment reads for the NEXT block | 2 x TN MFMAs of this block | 1/TM of the staging work], fenced
  // with sched_barrier so that hipcc keeps the order; waits become counted lgkmcnt(N) on reads issued a block earlier.
  auto k_step_pipe = [&](int kt, unsigned char* ca, unsigned char* cb, unsigned char* na, unsigned char* nb) {
    static_assert(kMfma32 || TM % 2 == 0, "pairs of A fragments");
    Staged ta[A_UNITS], tb[B_UNITS];
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) ta[u] = sa[u];
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u) tb[u] = sb[u];
    load_step(min(kt + 2, a_end - 1));            // in flight during this whole step
    __builtin_amdgcn_sched_barrier(0);
    constexpr int kPieces = (A_UNITS + B_UNITS) * 4, kBlocks = kMfma32 ? 4 : TM;   // 32x32x16: one block per 16-wide K slice
    auto pieces = [&](int blk) {                  // this block's share of the staging of step kt + 1
#pragma unroll
      for (int c = blk; c < kPieces; c += kBlocks) {
        const int u = c >> 2, j = c & 3;
        if (u < A_UNITS) {
#ifdef ARCQ_EXPERIMENT_A_RAW
          if (!A_PARTIAL || tid < BM * 2) stage_piece_raw(na, a_slot[u < A_UNITS ? u : 0][j], ta[u < A_UNITS ? u : 0], a_live[u < A_UNITS ? u : 0], j);
#else
          if (!A_PARTIAL || tid < BM * 2) stage_piece(na, a_slot[u < A_UNITS ? u : 0][j], ta[u < A_UNITS ? u : 0], a_live[u < A_UNITS ? u : 0], j);
#endif
        } else {
          const int v = u - A_UNITS < B_UNITS ? u - A_UNITS : 0;
#ifdef ARCQ_EXPERIMENT_B_RAW
          if (!B_PARTIAL || tid < BN * 2) stage_piece_raw(nb, b_slot[v][j], tb[v], b_live[v], j);
#else
          if (!B_PARTIAL || tid < BN * 2) stage_piece(nb, b_slot[v][j], tb[v], b_live[v], j);
#endif
        }
      }
    };
    if constexpr (kMfma32) {
      // 32x32x16 MFMAs last 32 cycles: ~3 VALU instructions fit into each one's shadow.  Blocks = the four 16-wide K
      // slices; all TM + TN fragments of slice ks + 1 are requested before the TM x TN MFMAs of slice ks.
      Frag8 f32a[2][TM], f32b[2][TN];
      auto rd_set = [&](int buf, int ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) f32a[buf][i].u = *reinterpret_cast<const uint4*>(ca + (fa_base ^ (ks * 32)) + i * 32 * kRowBytes);
#pragma unroll
        for (int j = 0; j < TN; ++j) f32b[buf][j].u = *reinterpret_cast<const uint4*>(cb + (fb_base ^ (ks * 32)) + j * 32 * kRowBytes);
      };
      rd_set(0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) rd_set((ks + 1) & 1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if constexpr (kMfma32)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f32b[ks & 1][j].v, f32a[ks & 1][i].v, acc[i][j], 0, 0, 0);
        pieces(ks);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      return;
    }
    auto rd_a = [&](int ks, int i) { Frag8 f; f.u = *reinterpret_cast<const uint4*>(ca + (fa_base ^ (ks * 64)) + i * 16 * kRowBytes); return f; };
    auto rd_b = [&](int ks, int j) { Frag8 f; f.u = *reinterpret_cast<const uint4*>(cb + (fb_base ^ (ks * 64)) + j * 16 * kRowBytes); return f; };
    // fragment queue: A fragments are requested kAhead pairs before their MFMAs, the B set of the second K half during the
    // last pairs of the first
    constexpr int kPairs = TM / 2, kSeq = 2 * kPairs, kAhead = 1;   // 2 measured the same (1188-1202 vs 1193-1200) with 8 more registers
    Frag8 fb[2][TN], fq[kSeq][2];
    auto rd_pair = [&](int q) {                   // q = ks * kPairs + pr (compile-time after unrolling)
      fq[q][0] = rd_a(q / kPairs, 2 * (q % kPairs));
      fq[q][1] = rd_a(q / kPairs, 2 * (q % kPairs) + 1);
    };
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = rd_b(0, j);
#pragma unroll
    for (int q = 0; q < kAhead; ++q) rd_pair(q);
#pragma unroll
    for (int q = 0; q < kSeq; ++q) {
      const int ks = q / kPairs, pr = q % kPairs;
      // (1) reads for later blocks
      if (q + kAhead < kSeq) rd_pair(q + kAhead);
      if (ks == 0 && pr == kPairs - 1 - (kAhead - 1 < kPairs - 1 ? kAhead - 1 : kPairs - 1)) {
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[1][j] = rd_b(1, j);
      }
      __builtin_amdgcn_sched_barrier(0);
      // (2) this block's MFMAs; weights are the MFMA A operand
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if constexpr (!kMfma32)
            acc[2 * pr + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[ks][j].v, fq[q][ii].v, acc[2 * pr + ii][j], 0, 0, 0);
      pieces(q);                                  // (3) its share of the staging of step kt + 1
      // spacing inside the block: two vector instructions in the shadow of every MFMA (measured on 4096^2, TFLOP/s:
      // compiler's own placement 1211-1219, 1 MFMA + 1 VALU 1195-1201, 1+2 1234-1236, 1+3 1217-1220, 2+4 1227, 4+5 1180)
      // The smaller tiles measure the same or slightly worse with it (M=1024: 612 vs 624), so only the 8-fragment tile.
#pragma unroll
      for (int m8 = 0; m8 < (TM == 8 ? 2 * TN : 0); ++m8) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };

  // ---- prologue: tile 0 into buffer 0, registers <- step 1
  load_step(a_begin);
  store_step(lds_a0, lds_b0);
  load_step(min(a_begin + 1, a_end - 1));
  if (kStagger && late) {                       // the late group enters the loop with step 1 staged and step 2 in registers
    store_step(lds_a1, lds_b1);
    load_step(min(a_begin + 2, a_end - 1));
  }
  __syncthreads();
  int kt = a_begin;
  // the second-dispatched half of an 8-wave workgroup loses every issue arbitration against its SIMD partner at equal
  // priority; a static priority for it (never flipped) measured +0.8 % (1192-1203 -> 1209-1212 TFLOP/s)
  if (WAVES_M * WAVES_N == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  ARCQ_TILE_STAMP(0);
  if constexpr (kPipe) {
    for (; kt + 1 < a_end; kt += 2) {
      k_step_pipe(kt, lds_a0, lds_b0, lds_a1, lds_b1);
      k_step_pipe(kt + 1, lds_a1, lds_b1, lds_a0, lds_b0);
    }
    if (kt < a_end) k_step_pipe(kt, lds_a0, lds_b0, lds_a1, lds_b1);
  } else {
    for (; kt + 1 < a_end; kt += 2) {
      k_step(kt, lds_a0, lds_b0, lds_a1, lds_b1);
      k_step(kt + 1, lds_a1, lds_b1, lds_a0, lds_b0);
    }
    if (kt < a_end) k_step(kt, lds_a0, lds_b0, lds_a1, lds_b1);
  }
  ARCQ_TILE_STAMP(1);

  // ---- epilogue.  16x16 tiles: lane holds D[m = +(lane & 15)][n = +4*(lane >> 4) + r], r = 0..3.
  //      32x32 tiles: lane holds D[m = +(lane & 31)][n = +8*g + 4*(lane >> 5) + r], g = 0..3, r = 0..3.
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);
  const bool vec_ok = (p.N & 3) == 0;
  uint32_t act_max = 0;                       // kEpiSiluMul: max |act| of this thread, as bf16 magnitude bits
  // `pre`: the four bias / residual values of (m, n .. n + 3) already fetched with ONE 8-byte load each (N % 4 == 0), two bf16 per dword;
  // otherwise they are read element by element (ragged N).  Same operations in the same order either way.
  auto store4 = [&](int m, int n, float d0, float d1, float d2, float d3, bool pre, uint2 bias2, uint2 res2) __attribute__((always_inline)) {
    if (m >= p.M || n >= p.N) return;
    const uint32_t bw[2] = {bias2.x, bias2.y}, rw[2] = {res2.x, res2.y};
    auto bias_at = [&](int r) { return bf16_bits_to_f32(pre ? (bw[r >> 1] >> (16 * (r & 1))) & 0xffffu : (uint32_t)p.bias[n + r]); };
    if (kEpi == kEpiSiluMul) {                // columns n..n+3 = (gate_j, up_j, gate_j+1, up_j+1), j = n / 2; N % 4 == 0
      uint32_t y[4] = {f32_to_bf16_bits(alpha * d0), f32_to_bf16_bits(alpha * d1), f32_to_bf16_bits(alpha * d2), f32_to_bf16_bits(alpha * d3)};
      if (p.bias) {                             // `y = matmul(...); y = y + bias` of the separate GEMM, in bf16 (qLinearLayer.py:74-76)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = f32_to_bf16_bits(bf16_bits_to_f32(y[e]) + bias_at(e));
      }
      const uint32_t a0 = silu_mul_bf16(y[0], y[1]);
      const uint32_t a1 = silu_mul_bf16(y[2], y[3]);
      *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(p.D) + (size_t)m * (p.N >> 1) + (n >> 1)) = a0 | (a1 << 16);
      act_max = max(act_max, max(a0 & 0x7fffu, a1 & 0x7fffu));
      return;
    }
    if (p.splits > 1) {                       // raw partial sums; the launcher only splits when N % 4 == 0
      *reinterpret_cast<float4*>(p.partial + ((size_t)split * p.M + m) * p.N + n) = make_float4(d0, d1, d2, d3);
      return;
    }
    float d[4] = {alpha * d0, alpha * d1, alpha * d2, alpha * d3};
    if (p.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (pre || n + r < p.N) d[r] = (p.out_dtype == ARCQ_OUT_F32 ? d[r] : bf16_bits_to_f32(f32_to_bf16_bits(d[r]))) + bias_at(r);
    }
    if (p.residual) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (pre || n + r < p.N) {
          const float res = bf16_bits_to_f32(pre ? (rw[r >> 1] >> (16 * (r & 1))) & 0xffffu : (uint32_t)p.residual[(size_t)m * p.N + n + r]);
          d[r] = (p.out_dtype == ARCQ_OUT_F32 ? d[r] : bf16_bits_to_f32(f32_to_bf16_bits(d[r]))) + res;
        }
    }
    if (p.out_dtype == ARCQ_OUT_F32) {
      float* o = reinterpret_cast<float*>(p.D) + (size_t)m * p.N + n;
      if (vec_ok) *reinterpret_cast<float4*>(o) = make_float4(d[0], d[1], d[2], d[3]);
      else for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = d[r];
    } else {
      uint16_t* o = reinterpret_cast<uint16_t*>(p.D) + (size_t)m * p.N + n;
      if (vec_ok) *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]));
      else for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = (uint16_t)f32_to_bf16_bits(d[r]);
    }
  };
  if constexpr (kMfma32) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int m = m0 + wr * WM + i * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wc * WN + j * 32 + 8 * g + 4 * (lane >> 5);
          store4(m, n, acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], false, make_uint2(0, 0), make_uint2(0, 0));
        }
      }
  } else {
    // Epilogue operands, fetched AHEAD of their use: a lane's TN bias quads once (they do not depend on the row), the residual quads of
    // row tile i + 1 while row tile i is finished and stored.  (Element-wise loads inside store4, each awaited before the next, cost the
    // model's prefill GEMMs + 19 % with a bias, + 32 % with a residual: tools/tile_epilogue_cost.py.)
    const bool pre = vec_ok && p.splits <= 1 && ((reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.residual)) & 7) == 0;   // (a sliced view may be misaligned)
    const int nb = n0 + wc * WN + 4 * (lane >> 4), mb = m0 + wr * WM + (lane & 15);
    if (pre) {                                 // (two copies of the store loops: `pre` as a constant inside each keeps store4 free of branches on it)
      uint2 bias_v[TN], res_cur[TN], res_nxt[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bias_v[j] = res_cur[j] = res_nxt[j] = make_uint2(0, 0);
        if (p.bias && nb + j * 16 < p.N) bias_v[j] = *reinterpret_cast<const uint2*>(p.bias + nb + j * 16);
      }
      auto res_load = [&](int i, uint2 (&r)[TN]) {
        if (!p.residual) return;
        const int m = mb + i * 16;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (m < p.M && nb + j * 16 < p.N) r[j] = *reinterpret_cast<const uint2*>(p.residual + (size_t)m * p.N + nb + j * 16);
      };
      res_load(0, res_cur);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i + 1 < TM) res_load(i + 1, res_nxt);
#pragma unroll
        for (int j = 0; j < TN; ++j) store4(mb + i * 16, nb + j * 16, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], true, bias_v[j], res_cur[j]);
#pragma unroll
        for (int j = 0; j < TN; ++j) res_cur[j] = res_nxt[j];
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          store4(mb + i * 16, nb + j * 16, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], false, make_uint2(0, 0), make_uint2(0, 0));
    }
  }
  if (kEpi == kEpiSiluMul) {
    __shared__ uint32_t wave_max[WAVES_M * WAVES_N];
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) act_max = max(act_max, (uint32_t)__shfl_down((int)act_max, sh, 64));
    if (lane == 0) wave_max[wave] = act_max;
    __syncthreads();
    if (tid == 0) {
      uint32_t m = 0;
#pragma unroll
      for (int w = 0; w < WAVES_M * WAVES_N; ++w) m = max(m, wave_max[w]);
      p.slots[blockIdx.x] = m;
    }
  }
}

// ---- tile configurations --------------------------------------------------------------------------------------------------
// id = the value of ARCQ_TILE_CFG that forces it (debug / tuning only; 0 = heuristic).
//   1 = 128x128 (4 waves), 2 = 256x256 (4 waves), 3 = 256x256 (8 waves), 4 = 128x256 (4 waves), 5 / 6 = 32x32x16 MFMA variants of
//   3 / 1, 7 = 64x256 (4 waves), 8 = 32x256 (4 waves), 9 = 1 without split-K; 8-wave tiles for the shapes between decode and
//   prefill: 10 = 128x256, 11 = 256x128, 12 = 128x128, 13 = 64x256, 14 = 64x128; 4-wave: 15 = 64x64, 16 = 64x128, 17 = 128x64
struct TileCfg { int id, bm, bn, waves; };
static constexpr TileCfg kTileCfgs[] = {{1, 128, 128, 4}, {2, 256, 256, 4}, {3, 256, 256, 8}, {4, 128, 256, 4}, {5, 256, 256, 8}, {6, 128, 128, 4},
                                        {7, 64, 256, 4}, {8, 32, 256, 4}, {9, 128, 128, 4}, {10, 128, 256, 8}, {11, 256, 128, 8}, {12, 128, 128, 8},
                                        {13, 64, 256, 8}, {14, 64, 128, 8}, {15, 64, 64, 4}, {16, 64, 128, 4}, {17, 128, 64, 4}};
static const TileCfg& tile_cfg(int id) {
  for (const TileCfg& c : kTileCfgs)
    if (c.id == id) return c;
  return kTileCfgs[2];
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static int tile_cfg_override() {                // ARCQ_TILE_CFG (tuning)
  static const int v = env_int("ARCQ_TILE_CFG", 0);
  return v;
}
static int tile_split_override() {              // ARCQ_TILE_SPLIT = forced split-K factor (tuning; 0 = by shape)
  static const int v = env_int("ARCQ_TILE_SPLIT", 0);
  return v;
}
static bool tile_stagger() {                    // ARCQ_TILE_STAGGER=0|1 (tuning)
  static const int v = env_int("ARCQ_TILE_STAGGER", 0);
  return v != 0;
}
static bool tile_pipe() {                       // ARCQ_TILE_PIPE=0|1 (tuning)
  static const int v = env_int("ARCQ_TILE_PIPE", 1);
  return v != 0;
}

// Split-K factor of a tile shape: split while the tiles alone leave CUs idle, keeping >= 8 atoms (512 K elements)
// per split so that the prologue/epilogue stay amortised; needs N % 4 == 0 (16-byte partial stores).
static void tile_split(int64_t M, int64_t N, int64_t K, int BM, int BN, int* splits, int* atoms_per_split) {
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int atoms = (int)(K / 64);
  int s = 1;
  if ((N % 4) == 0) {
    if (tile_split_override() > 0) s = min(tile_split_override(), max(atoms / 2, 1));
    else
      while (tiles * s < 256 && atoms / (s * 2) >= 8 && s < 32) s *= 2;
  }
  const int per = (atoms + s - 1) / s;
  *splits = (atoms + per - 1) / per;      // drop empty splits
  *atoms_per_split = per;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool kMfma32 = false, int kEpi = kEpiPlain>
static int launch_tile(const GemmArgs& a, hipStream_t stream, bool allow_split = false) {
  TileParams p;
  p.A = a.A; p.B = a.B; p.SFA = a.SFA; p.SFB = a.SFB; p.D = a.D;
  p.alpha_dev = a.alpha_dev; p.bias = a.bias; p.residual = a.residual;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  p.tiles_m = (a.M + BM - 1) / BM;
  p.tiles_n = (a.N + BN - 1) / BN;
  p.splits = 1; p.atoms_per_split = a.K / 64; p.partial = nullptr;
  p.epi = a.epilogue; p.slots = a.absmax_slots;
#ifdef ARCQ_STREAM_STAMPS
  p.stamps = g_tile_stamps;
#endif
  if (allow_split && a.epilogue == kEpiPlain) {
    tile_split(a.M, a.N, a.K, BM, BN, &p.splits, &p.atoms_per_split);
    if (p.splits > 1) {
      const int64_t need = (int64_t)p.splits * a.M * a.N * (int64_t)sizeof(float);
      if (!a.workspace || a.workspace_bytes < need)
        return fail(ARCQ_ERR_WORKSPACE, "arcq_gemm_nvfp4: split-K needs %lld B of workspace, got %lld", (long long)need,
                    (long long)a.workspace_bytes);
      p.partial = reinterpret_cast<float*>(a.workspace);
    }
  }
  const size_t lds = 2 * (size_t)(BM + BN) * kRowBytes;
  constexpr bool kCanStagger = WAVES_M * WAVES_N == 8 && BM == 256 && BN == 256, kCanPipe = true;
  auto kern = gemm_tile_kernel<BM, BN, WAVES_M, WAVES_N, kMfma32, kEpi, false, false>;
  if (kCanPipe && tile_pipe()) kern = gemm_tile_kernel<BM, BN, WAVES_M, WAVES_N, kMfma32, kEpi, false, kCanPipe>;
  else if (kCanStagger && tile_stagger()) kern = gemm_tile_kernel<BM, BN, WAVES_M, WAVES_N, kMfma32, kEpi, kCanStagger, false>;
  // per instantiation and per (pipe, stagger) variant of it, per device: the driver call costs host time on every launch otherwise
  static LdsOptIn lds_opt[3];
  const int which = (kCanPipe && tile_pipe()) ? 1 : (kCanStagger && tile_stagger()) ? 2 : 0;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_opt[which], (int)lds, "arcq_gemm_nvfp4 (tile)")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n * p.splits)), dim3(WAVES_M * WAVES_N * 64), lds, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (tile): launch failed: %s", hipGetErrorString(e));
  if (p.splits > 1) return gemm_splitk_finish(a, p.splits, stream);
  return ARCQ_OK;
}

// Shape -> tile: 256x256 with 8 waves (one workgroup per CU, two waves per SIMD) once it yields enough tiles to
// fill the 256 CUs (measured, tools/gemm_sweep.py: 4096^2 1066 vs 762 TFLOP/s, 8192^2 1347 vs 984); otherwise
// 128x128 (two workgroups per CU), or a 64- / 32-row tile over 256 weight rows for M <= 64 / 32, each with split-K.
static int tile_choice(int64_t M, int64_t N, int64_t K) {
  const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
  if (t256 >= 192) return 3;
  if (M <= 32) return 8;
  if (M <= 64) return 7;
  // 128 x 256 with 8 waves where it measured faster than 128 x 128 (profiles/r03_midm_tile_sweep.jsonl: M = 2048, N = 4096 65.7 against
  // 75.7 us, M = 512 31.8 / 37.0, M = 1024 46.2 / 50.3; K = 18944, N = 3584: M = 1024 155 / 164, M = 2048 282 / 307): one exact round of the
  // chip, half a round, few tiles, or a long K -- elsewhere 128 x 128 (two workgroups per CU) fills the ragged rounds better
  const int64_t t10 = ((M + 127) / 128) * ((N + 255) / 256);
  if (t10 <= 64 || t10 == 128 || (t10 > 224 && t10 <= 256) || (K >= 8192 && t10 >= 96 && t10 <= 256)) return 10;
  return 1;
}

// the configuration a launch uses, and whether it may split K (the silu-mul epilogue never does; 2 - 6 and 9 are the unsplit tuning arms)
static int effective_cfg(int64_t M, int64_t N, int64_t K, bool* may_split) {
  int id = tile_cfg_override();
  if (id == 0) id = tile_choice(M, N, K);
  id = tile_cfg(id).id;
  *may_split = !(id >= 2 && id <= 6) && id != 9;
  return id;
}

// workgroups (= abs-max slots) of the silu-mul epilogue, which never splits K
int64_t gemm_tile_silu_slots(int64_t M, int64_t N, int64_t K) {
  bool sp;
  const TileCfg& c = tile_cfg(effective_cfg(M, N, K, &sp));
  return ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
}

int64_t gemm_tile_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int s = 1, per = 0;
  bool sp;
  const TileCfg& c = tile_cfg(effective_cfg(M, N, K, &sp));
  if (sp) tile_split(M, N, K, c.bm, c.bn, &s, &per);
  return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

template <int kEpi>
static int gemm_tile_epi(const GemmArgs& a, hipStream_t stream) {
  bool sp;
  switch (effective_cfg(a.M, a.N, a.K, &sp)) {
    case 1: case 9: return launch_tile<128, 128, 2, 2, false, kEpi>(a, stream, sp);
    case 2: return launch_tile<256, 256, 2, 2, false, kEpi>(a, stream);
    case 4: return launch_tile<128, 256, 2, 2, false, kEpi>(a, stream);
    case 5: return launch_tile<256, 256, 2, 4, true, kEpi>(a, stream);
    case 6: return launch_tile<128, 128, 2, 2, true, kEpi>(a, stream);
    case 7: return launch_tile<64, 256, 1, 4, false, kEpi>(a, stream, sp);
    case 8: return launch_tile<32, 256, 1, 4, false, kEpi>(a, stream, sp);
    case 10: return launch_tile<128, 256, 2, 4, false, kEpi>(a, stream, sp);
    case 11: return launch_tile<256, 128, 4, 2, false, kEpi>(a, stream, sp);
    case 12: return launch_tile<128, 128, 2, 4, false, kEpi>(a, stream, sp);
    case 13: return launch_tile<64, 256, 1, 8, false, kEpi>(a, stream, sp);
    case 14: return launch_tile<64, 128, 2, 4, false, kEpi>(a, stream, sp);
    case 15: return launch_tile<64, 64, 2, 2, false, kEpi>(a, stream, sp);
    case 16: return launch_tile<64, 128, 2, 2, false, kEpi>(a, stream, sp);
    case 17: return launch_tile<128, 64, 2, 2, false, kEpi>(a, stream, sp);
    default: return launch_tile<256, 256, 2, 4, false, kEpi>(a, stream);
  }
}

int gemm_tile(const GemmArgs& a, hipStream_t stream) {
  // the epilogue is a template parameter: the plain kernels do not carry the exp code of silu-mul
  return a.epilogue == kEpiSiluMul ? gemm_tile_epi<kEpiSiluMul>(a, stream) : gemm_tile_epi<kEpiPlain>(a, stream);
}

}  // namespace arcq
