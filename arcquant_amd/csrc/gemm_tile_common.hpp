// Pieces shared by the tile GEMM kernels (gemm_tile.hip, gemm_tile_ws.hip): parameter block, LDS tile
// addressing, the register staging unit and its dequantising store.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "arcq_internal.hpp"
#include "gemm_common.hpp"

namespace arcq {

struct TileParams {
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
  // split-K (shapes whose tiles alone leave CUs idle): workgroup blockIdx.x / (tiles_m*tiles_n) owns the scale-factor
  // atoms [split*atoms_per_split, +atoms_per_split) and writes raw fp32 sums to partial[split][M][N]
  int splits, atoms_per_split;
  float* partial;
  // epilogue ARCQ_EPI_SILU_MUL: weight rows interleave gate and up (g0,u0,g1,u1,...), D is the bf16 [M, N/2] tensor
  // silu(gate)*up and slots[blockIdx.x] receives this workgroup's max |D| (bit pattern) for the dynamic quantiser
  int epi;
  unsigned int* slots;
#ifdef ARCQ_STREAM_STAMPS      // DIAGNOSTIC build only (make diag): [workgroup][4] = s_memtime / s_memrealtime before and after the K loop
  unsigned long long* stamps;
#endif
};

constexpr int kBK = 64;                 // K elements per step = one scale-factor atom column (4 groups)
constexpr int kRowBytes = kBK * 2;      // fp16 row of a tile in LDS

// Byte offset of 16-byte slot `ks` (0..7) of tile row `r`.  XOR with (r & 7):
//   * ds_read_b128 of an MFMA fragment (16 rows x one slot per 16-lane group) touches all 16 slots of the
//     256-byte bank row exactly once  -> conflict-free;
//   * ds_write_b128 of the staging pass (8-lane groups = 4 rows x 2 halves) touches all 8 slots of the
//     128-byte write bank period exactly once -> conflict-free.
// The term is invariant under r += 16, so fragment tile i is a constant byte offset from tile 0.
__device__ __forceinline__ int lds_slot(int r, int ks) { return r * kRowBytes + ((ks ^ (r & 7)) << 4); }
// Variant for 32-row MFMA fragments (v_mfma_f32_32x32x16_f16): a 16-lane ds_read_b128 group then spans rows
// {0-3,12-15,20-27} of ONE slot, which needs ((r >> 1) & 7) to stay conflict-free (invariant under r += 32).
__device__ __forceinline__ int lds_slot32(int r, int ks) { return r * kRowBytes + ((ks ^ ((r >> 1) & 7)) << 4); }

// One staging unit = 16 packed bytes (32 elements, two scale groups) of one tile row.
struct Staged {
  uint4 q;
  uint32_t sf;   // the two scale bytes in bits [15:0] (0 for rows outside the matrix)
};

// Unconditional loads (row clamped into the matrix, dead rows neutralised through their scale bytes at
// dequantisation time): nothing here waits on a load, so the prefetch stays in flight across the MFMAs.
__device__ __forceinline__ Staged stage_load(const uint8_t* __restrict__ qrow, const uint8_t* __restrict__ sfrow, int atom) {
  Staged s;
  s.q = *reinterpret_cast<const uint4*>(qrow + (size_t)atom * 32);
  s.sf = *reinterpret_cast<const uint16_t*>(sfrow + (size_t)atom * 512);
  return s;
}

// One quarter (8 elements, one 16-byte slot) of a staging unit: lets the K loop spread the dequantisation between its
// MFMA groups.
__device__ __forceinline__ void stage_piece(unsigned char* tile, int slot, const Staged& s, uint32_t live_mask, int j) {
  const uint32_t w = j == 0 ? s.q.x : j == 1 ? s.q.y : j == 2 ? s.q.z : s.q.w;
  const Frag8 f = dequant8(w, sf_pair_at(s.sf & live_mask, j < 2 ? 0 : 8));
  *reinterpret_cast<uint4*>(tile + slot) = f.u;
}

#if defined(ARCQ_EXPERIMENT_A_RAW) || defined(ARCQ_EXPERIMENT_B_RAW)
// TIMING EXPERIMENT ONLY (tools/scripts/build_variant_lib.sh; results are WRONG): the A panel (activations) staged WITHOUT its
// dequantisation -- what a tile GEMM reading pre-dequantised fp16 activations (emitted by the quantiser) could gain at most
__device__ __forceinline__ void stage_piece_raw(unsigned char* tile, int slot, const Staged& s, uint32_t live_mask, int j) {
  const uint32_t w = (j == 0 ? s.q.x : j == 1 ? s.q.y : j == 2 ? s.q.z : s.q.w) & (live_mask | (live_mask << 16)) & 0x3bff3bffu;   // finite fp16 patterns
  *reinterpret_cast<uint4*>(tile + slot) = make_uint4(w, w, w, w);
}
#endif

__device__ __forceinline__ void stage_store(unsigned char* tile, const int (&slot)[4], const Staged& s, uint32_t live_mask) {
  const uint32_t sf = s.sf & live_mask;
  const f16x2 s0 = sf_pair_at(sf, 0), s1 = sf_pair_at(sf, 8);
  Frag8 f0 = dequant8(s.q.x, s0), f1 = dequant8(s.q.y, s0), f2 = dequant8(s.q.z, s1), f3 = dequant8(s.q.w, s1);
  *reinterpret_cast<uint4*>(tile + slot[0]) = f0.u;
  *reinterpret_cast<uint4*>(tile + slot[1]) = f1.u;
  *reinterpret_cast<uint4*>(tile + slot[2]) = f2.u;
  *reinterpret_cast<uint4*>(tile + slot[3]) = f3.u;
}


}  // namespace arcq
