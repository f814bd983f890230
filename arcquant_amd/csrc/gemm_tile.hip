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
  int M, N, K;
  float alpha_host;
  int out_dtype;
  int tiles_m, tiles_n;
};

constexpr int kBK = 64;                 // K elements per step = one scale-factor atom column (4 groups)
constexpr int kRowBytes = kBK * 2;      // fp16 row of a tile in LDS

// byte offset of 16-byte slot `ks` (0..7) of tile row `r`; the XOR spreads the 16 lanes of a
// ds_read_b128 group over all 16 slots of the 256-byte bank row (conflict-free for MFMA fragments)
__device__ __forceinline__ int lds_slot(int r, int ks) { return r * kRowBytes + ((ks ^ ((r >> 1) & 7)) << 4); }

// One staging unit = 16 packed bytes (32 elements, two scale groups) of one tile row.
struct Staged {
  uint4 q;
  uint32_t sf;   // the two scale bytes in bits [15:0]
};

__device__ __forceinline__ Staged stage_load(const uint8_t* Q, const uint8_t* SF, int row, int rows, int half_k, int atoms_k,
                                             int atom, int half) {
  Staged s;
  const int rc = row < rows ? row : rows - 1;          // clamp, neutralise through the scale bytes
  s.q = *reinterpret_cast<const uint4*>(Q + (size_t)rc * half_k + atom * 32 + half * 16);
  const uint32_t v = *reinterpret_cast<const uint16_t*>(SF + sf_atom_offset(rc, atom, atoms_k) + half * 2);
  s.sf = row < rows ? v : 0u;
  return s;
}

__device__ __forceinline__ void stage_store(unsigned char* tile, int r, int half, const Staged& s) {
  const f16x2 s0 = sf_pair(s.sf & 0xffu), s1 = sf_pair((s.sf >> 8) & 0xffu);
  Frag8 f0 = dequant8(s.q.x, s0), f1 = dequant8(s.q.y, s0), f2 = dequant8(s.q.z, s1), f3 = dequant8(s.q.w, s1);
  *reinterpret_cast<uint4*>(tile + lds_slot(r, half * 4 + 0)) = f0.u;
  *reinterpret_cast<uint4*>(tile + lds_slot(r, half * 4 + 1)) = f1.u;
  *reinterpret_cast<uint4*>(tile + lds_slot(r, half * 4 + 2)) = f2.u;
  *reinterpret_cast<uint4*>(tile + lds_slot(r, half * 4 + 3)) = f3.u;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_tile_kernel(TileParams p) {
  constexpr int kThreads = WAVES_M * WAVES_N * 64;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;     // wave tile
  constexpr int TM = WM / 16, TN = WN / 16;               // MFMA tiles per wave
  // 16-byte staging units (2 per tile row) per thread; when a tile has fewer units than threads the
  // surplus waves skip it (wave-uniform guard)
  constexpr int A_UNITS = (BM * 2 + kThreads - 1) / kThreads, B_UNITS = (BN * 2 + kThreads - 1) / kThreads;
  constexpr int A_TILE = BM * kRowBytes, B_TILE = BN * kRowBytes;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [buf][A | B]
  unsigned char* const lds_a0 = smem;
  unsigned char* const lds_b0 = smem + A_TILE;
  unsigned char* const lds_a1 = smem + A_TILE + B_TILE;
  unsigned char* const lds_b1 = smem + 2 * A_TILE + B_TILE;

  // ---- XCD-aware tile order: blocks that land on one XCD (id % 8) get a contiguous range of tiles, and
  //      within it tiles walk M fastest in groups of 8 so concurrently resident blocks share B panels.
  const int ntiles = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
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

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Staged sa[A_UNITS], sb[B_UNITS];
  auto load_step = [&](int atom) {
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) {
      const int unit = tid + u * kThreads;
      if (unit < BM * 2) sa[u] = stage_load(p.A, p.SFA, m0 + (unit >> 1), p.M, half_k, atoms_k, atom, unit & 1);
    }
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u) {
      const int unit = tid + u * kThreads;
      if (unit < BN * 2) sb[u] = stage_load(p.B, p.SFB, n0 + (unit >> 1), p.N, half_k, atoms_k, atom, unit & 1);
    }
  };
  auto store_step = [&](unsigned char* la, unsigned char* lb) {
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) {
      const int unit = tid + u * kThreads;
      if (unit < BM * 2) stage_store(la, unit >> 1, unit & 1, sa[u]);
    }
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u) {
      const int unit = tid + u * kThreads;
      if (unit < BN * 2) stage_store(lb, unit >> 1, unit & 1, sb[u]);
    }
  };
  auto mma_step = [&](const unsigned char* la, const unsigned char* lb) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag8 fa[TM], fb[TN];
      const int slot = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i].u = *reinterpret_cast<const uint4*>(la + lds_slot(wr * WM + i * 16 + (lane & 15), slot));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j].u = *reinterpret_cast<const uint4*>(lb + lds_slot(wc * WN + j * 16 + (lane & 15), slot));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)   // weights (B of the GEMM) are the MFMA A operand: lane gets 4 consecutive n
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j].v, fa[i].v, acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop: one barrier per K step, loads of step k+1 in flight during the MFMAs of step k
  load_step(0);
  store_step(lds_a0, lds_b0);
  __syncthreads();
  for (int kt = 0; kt < atoms_k; ++kt) {
    unsigned char* const ca = (kt & 1) ? lds_a1 : lds_a0;
    unsigned char* const cb = (kt & 1) ? lds_b1 : lds_b0;
    unsigned char* const na = (kt & 1) ? lds_a0 : lds_a1;
    unsigned char* const nb = (kt & 1) ? lds_b0 : lds_b1;
    const bool more = kt + 1 < atoms_k;
    if (more) load_step(kt + 1);
    mma_step(ca, cb);
    if (more) store_step(na, nb);
    __syncthreads();
  }

  // ---- epilogue: lane holds D[m = .. + (lane & 15)][n = .. + 4*(lane >> 4) + r]
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);
  const bool vec_ok = (p.N & 3) == 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wr * WM + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wc * WN + j * 16 + 4 * (lane >> 4);
      if (n >= p.N) continue;
      float d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[r] = alpha * acc[i][j][r];
        if (p.bias && n + r < p.N) d[r] += bf16_bits_to_f32(p.bias[n + r]);
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
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_tile(const GemmArgs& a, hipStream_t stream) {
  TileParams p;
  p.A = a.A; p.B = a.B; p.SFA = a.SFA; p.SFB = a.SFB; p.D = a.D;
  p.alpha_dev = a.alpha_dev; p.bias = a.bias;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  p.tiles_m = (a.M + BM - 1) / BM;
  p.tiles_n = (a.N + BN - 1) / BN;
  const size_t lds = 2 * (size_t)(BM + BN) * kRowBytes;
  auto kern = gemm_tile_kernel<BM, BN, WAVES_M, WAVES_N>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (tile): cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(WAVES_M * WAVES_N * 64), lds, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (tile): launch failed: %s", hipGetErrorString(e));
  return ARCQ_OK;
}

int gemm_tile(const GemmArgs& a, hipStream_t stream) {
  // small M: 64-row tiles keep more workgroups in flight; otherwise 128x128
  if (a.M <= 64) return launch_tile<64, 128, 1, 4>(a, stream);
  return launch_tile<128, 128, 2, 2>(a, stream);
}

}  // namespace arcq
