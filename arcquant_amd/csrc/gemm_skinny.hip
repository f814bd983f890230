// ARC-NVFP4 GEMM for decode shapes (M <= 16): a weight-streaming, HBM-bound kernel for gfx950.
//
// Replaces the CUTLASS 128x128x128 block-scaled GEMM of the reference (kernels/src/nvfp4.cu:35-132)
// for the shapes where that kernel leaves 127/128 of its M tile empty ("decode").
//
// Roofline: bytes per launch = N*K*9/16 (packed B + scale bytes) + M*K*9/16 + M*N*2; everything else is
// on-chip.  Design for that bound:
//   * one workgroup (8 wave64) per 16 output columns = 16 rows of B; K is walked in slabs of 2048
//     elements (1 KiB of packed codes per row)
//   * every byte enters the CU through LDS-DMA (global_load_lds_dwordx4 / _dword): a wave-instruction
//     moves one whole 1 KiB row slab, i.e. full 128-byte lines.  (The first version loaded B straight
//     into the MFMA operand layout, 16 rows x 64 B per instruction; loads alone then took as long as
//     the whole kernel.)  A 3-deep LDS ring keeps two slabs in flight behind the one being multiplied;
//     waits are counted `s_waitcnt vmcnt(N)` + raw `s_barrier`, never vmcnt(0) inside the loop
//   * LDS-DMA writes lane-linearly, so the bank-conflict swizzle is applied to the per-lane SOURCE
//     address (16-byte unit ^ (row & 15)) and undone by the reader (cdna_hip_programming.md rule 21)
//   * scale bytes are staged with 4-byte LDS-DMA pieces that pick exactly this tile's bytes out of the
//     CUTLASS-swizzled layout (16 rows x 4 atoms = 256 useful bytes per instruction)
//   * operands are dequantised in registers to fp16 (exact, gemm_common.hpp) and contracted on
//     v_mfma_f32_16x16x32_f16 with the weights as the MFMA "A" operand, so a lane ends with 4
//     consecutive output columns of one token (one 8-byte store)
//   * cross-wave reduction of the 16x16 fp32 tile through LDS; epilogue fused (alpha, optional bias,
//     bf16 rounding); split-K over slabs (second pass) only when N/16 alone cannot fill the chip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "arcq_internal.hpp"
#include "gemm_common.hpp"

namespace arcq {

struct SkinnyParams {
  const uint8_t* A;
  const uint8_t* B;
  const uint8_t* SFA;
  const uint8_t* SFB;
  void* D;
  float* partial;         // [splitk, M, N] fp32 when splitk > 1
  const float* alpha_dev;
  const uint16_t* bias;
  int M, N, K;
  float alpha_host;
  int out_dtype;
  int tiles;              // ceil(N / 16)
  int slabs_per_split;
};

__device__ __forceinline__ void store_out4(const SkinnyParams& p, int m, int n, const float (&d)[4]) {
  // d[r] is the finished value of D[m, n + r]
  if (p.out_dtype == ARCQ_OUT_F32) {
    float* o = reinterpret_cast<float*>(p.D) + (size_t)m * p.N + n;
    if (n + 3 < p.N && (p.N & 3) == 0) {
      *reinterpret_cast<float4*>(o) = make_float4(d[0], d[1], d[2], d[3]);
    } else {
      for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = d[r];
    }
  } else {
    uint16_t* o = reinterpret_cast<uint16_t*>(p.D) + (size_t)m * p.N + n;
    if (n + 3 < p.N && (p.N & 3) == 0) {
      *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]));
    } else {
      for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = (uint16_t)f32_to_bf16_bits(d[r]);
    }
  }
}

__device__ __forceinline__ void finish4(const SkinnyParams& p, int m, int n, const float (&acc)[4]) {
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);
  float d[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    d[r] = alpha * acc[r];                                          // epilogue in fp32 (nvfp4.cu:117-121)
    if (p.bias && n + r < p.N) d[r] += bf16_bits_to_f32(p.bias[n + r]);
  }
  store_out4(p, m, n, d);
}

constexpr int kSlabK = 2048;                 // K elements per slab
constexpr int kSlabBytes = kSlabK / 2;       // packed bytes per row per slab (= one LDS-DMA wave-instruction)
constexpr int kSlabAtoms = kSlabK / 64;      // scale-factor atoms per slab
constexpr int kSkWaves = 8;
constexpr int kChunksPerWave = kSlabK / 128 / kSkWaves;   // 128-element MFMA chunks per wave per slab (= 2)

// LDS stage: [B rows 16 KiB][A rows 16 KiB][SFB 2 KiB][SFA 2 KiB]
constexpr int kStageB = 0;
constexpr int kStageA = 16 * kSlabBytes;
constexpr int kStageSFB = 2 * 16 * kSlabBytes;
constexpr int kStageSFA = kStageSFB + kSlabAtoms * 16 * 4;
constexpr int kStageBytes = kStageSFA + kSlabAtoms * 16 * 4;
// LDS-DMA instructions per wave per slab: 2 B rows + 2 A rows + 1 SFB + 1 SFA.  FIXED: the waits are counted.
constexpr int kGldsPerWavePerSlab = 6;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ void glds16(const void* src, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)lds_dst_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* src, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)lds_dst_wave_base, 4, 0, 0);
}

template <int kStages>
__global__ __launch_bounds__(kSkWaves * 64) void gemm_skinny_kernel(SkinnyParams p) {
  static_assert(kStages >= 2 && kStages <= 4, "ring depth");
  
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [kStages][kStageBytes] + reduction
  float* red = reinterpret_cast<float*>(smem + kStages * kStageBytes);   // [kSkWaves][64][4]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4;           // K quarter of a 128-element chunk
  const int rl = lane & 15;          // weight row within the tile / token index

  // XCD-aware tile order: the 8 tiles of a 128-row scale-factor super-tile share 64-byte lines of SFB, keep
  // them on one XCD (blocks are dealt round-robin over the 8 XCDs); bijective for any tile count.
  int tile = blockIdx.x;
  {
    const int q8 = p.tiles >> 3, r8 = p.tiles & 7, x = tile & 7, j = tile >> 3;
    tile = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + j;
  }
  const int n0 = tile * 16;
  const int atoms_k = p.K >> 6;
  const int half_k = p.K >> 1;
  const int nslabs_total = (p.K + kSlabK - 1) / kSlabK;
  const int slab_begin = blockIdx.y * p.slabs_per_split;
  const int slab_end = min(nslabs_total, slab_begin + p.slabs_per_split);
  const int nslabs = slab_end - slab_begin;

  const int n = n0 + rl, m = rl;
  const bool n_ok = n < p.N, m_ok = m < p.M;

  // ---- LDS-DMA sources of this wave (loop invariant parts).  Addresses are clamped into the buffers;
  //      whatever lands beyond the valid K range / row range is never read or is masked via its scale.
  const int brow0 = min(n0 + wave, p.N - 1), brow1 = min(n0 + wave + 8, p.N - 1);
  // A rows beyond M are not fetched at all (with M = 1 they would be 15 redundant copies of row 0 from one
  // hot L2 line); the number of LDS-DMA ops this wave issues per slab is therefore wave-dependent (4..6)
  const bool a0_on = wave < p.M, a1_on = wave + 8 < p.M;
  const int per_slab = 4 + (a0_on ? 1 : 0) + (a1_on ? 1 : 0);
  const int arow0 = min(wave, p.M - 1), arow1 = min(wave + 8, p.M - 1);
  const uint8_t* bsrc0 = p.B + (size_t)brow0 * half_k;
  const uint8_t* bsrc1 = p.B + (size_t)brow1 * half_k;
  const uint8_t* asrc0 = p.A + (size_t)arow0 * half_k;
  const uint8_t* asrc1 = p.A + (size_t)arow1 * half_k;
  // source 16-byte unit of this lane within a slab row: lane ^ (tile row & 15)   (involution, undone by the reader)
  const int unit0 = (lane ^ (wave & 15)) * 16, unit1 = (lane ^ ((wave + 8) & 15)) * 16;
  // scale bytes: lane (row rl, atom wave*4 + q of the slab) fetches its 4-byte group
  const uint8_t* sfb_src = p.SFB + sf_atom_offset(min(n, p.N - 1), 0, atoms_k);
  const uint8_t* sfa_src = p.SFA + sf_atom_offset(min(m, p.M - 1), 0, atoms_k);

  auto issue_slab = [&](int slab, int stage) {
    unsigned char* st = smem + stage * kStageBytes;
    const int kbyte = slab * kSlabBytes;
    const int lim = half_k - 16 - kbyte;                      // last valid 16-byte unit of the row in this slab
    const int o0 = kbyte + min(unit0, lim), o1 = kbyte + min(unit1, lim);
    glds16(bsrc0 + o0, st + kStageB + wave * kSlabBytes);
    glds16(bsrc1 + o1, st + kStageB + (wave + 8) * kSlabBytes);
    if (a0_on) glds16(asrc0 + o0, st + kStageA + wave * kSlabBytes);
    if (a1_on) glds16(asrc1 + o1, st + kStageA + (wave + 8) * kSlabBytes);
    const int atom = min(slab * kSlabAtoms + wave * 4 + q, atoms_k - 1);
    glds4(sfb_src + (size_t)atom * 512, st + kStageSFB + wave * 256);    // LDS dword index = (wave*4 + q)*16 + rl
    glds4(sfa_src + (size_t)atom * 512, st + kStageSFA + wave * 256);
  };

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void_t*)smem;   // LDS byte address of the dynamic region

  // ---- prologue: fill the ring
#pragma unroll
  for (int s = 0; s < kStages - 1; ++s)
    if (s < nslabs) issue_slab(slab_begin + s, s);

  for (int s = 0; s < nslabs; ++s) {
    // wait for this wave's pieces of slab s: everything it issued later may stay in flight
    const int later = min(kStages - 2, nslabs - 1 - s);
    switch (later * per_slab) {        // wave-uniform; the immediate must be a literal
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();      // all pieces of slab s have landed; every wave is done reading slab s-1
    if (s + kStages - 1 < nslabs) issue_slab(slab_begin + s + kStages - 1, (s + kStages - 1) % kStages);

    const unsigned char* st = smem + (s % kStages) * kStageBytes;
    const int slab = slab_begin + s;
#pragma unroll
    for (int cc = 0; cc < kChunksPerWave; ++cc) {
      const int c = wave * kChunksPerWave + cc;                 // chunk within the slab: bytes [64c, 64c+64) of each row
      const int atom_l = 2 * c + (q >> 1);                      // this lane's atom within the slab
      const bool live = slab * kSlabAtoms + atom_l < atoms_k;
      const int unit = (4 * c + q) ^ rl;                        // undo the source swizzle
      // The LDS reads are inline asm: hipcc's waitcnt pass cannot tell that a ds_read of stage s does not alias
      // the LDS-DMA writes in flight to the other stages and would drain them with vmcnt(0) (seen in the .s),
      // serialising the ring.  Ordering is ours: the counted vmcnt + barrier above (cdna_hip_programming.md 5.7).
      u32x4 bq, aq;
      uint32_t bs, as;
      const uint32_t frag_addr = (uint32_t)(uintptr_t)(st - smem) + kStageB + rl * kSlabBytes + unit * 16;
      const uint32_t sf_addr = (uint32_t)(uintptr_t)(st - smem) + kStageSFB + (atom_l * 16 + rl) * 4;
      asm volatile(
          "ds_read_b128 %0, %4\n\t"
          "ds_read_b128 %1, %4 offset:%c6\n\t"
          "ds_read_b32 %2, %5\n\t"
          "ds_read_b32 %3, %5 offset:%c7\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(bq), "=&v"(aq), "=&v"(bs), "=&v"(as)
          : "v"(frag_addr + lds_base), "v"(sf_addr + lds_base), "i"(kStageA - kStageB), "i"(kStageSFA - kStageSFB));
      bs = (live && n_ok) ? bs : 0u;
      as = (live && m_ok) ? as : 0u;
      const int sh = (q & 1) * 16;                              // which two of the atom's four scale bytes
      const f16x2 sb0 = sf_pair((bs >> sh) & 0xffu), sb1 = sf_pair((bs >> (sh + 8)) & 0xffu);
      const f16x2 sa0 = sf_pair((as >> sh) & 0xffu), sa1 = sf_pair((as >> (sh + 8)) & 0xffu);
      Frag8 b0 = dequant8(bq.x, sb0), b1 = dequant8(bq.y, sb0), b2 = dequant8(bq.z, sb1), b3 = dequant8(bq.w, sb1);
      Frag8 a0 = dequant8(aq.x, sa0), a1 = dequant8(aq.y, sa0), a2 = dequant8(aq.z, sa1), a3 = dequant8(aq.w, sa1);
      // weights are the MFMA A operand (rows i = n), activations the B operand (cols j = m)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc, 0, 0, 0);
    }
  }

  // ---- cross-wave reduction; lane holds C[n = n0 + 4q + r][m = rl]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 64 + lane) * 4 + r] = acc[r];
  __syncthreads();
  if (wave == 0) {
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < kSkWaves; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(red + (w * 64 + lane) * 4);
      sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
    }
    if (m_ok) {
      const int nn = n0 + 4 * q;
      if (gridDim.y == 1) {
        finish4(p, m, nn, sum);
      } else {
        float* o = p.partial + ((size_t)blockIdx.y * p.M + m) * p.N + nn;
        for (int r = 0; r < 4; ++r) if (nn + r < p.N) o[r] = sum[r];
      }
    }
  }
}

// second pass of split-K: D[m,n] = epilogue(sum_s partial[s,m,n]) in a fixed order (deterministic)
__global__ __launch_bounds__(256) void splitk_finish_kernel(SkinnyParams p, int splitk) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t total = (int64_t)p.M * p.N;
  if (i4 >= total) return;
  const int m = (int)(i4 / p.N), n = (int)(i4 % p.N);   // N % 4 == 0 is required by the launcher for this path
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < splitk; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(p.partial + (size_t)k * total + i4);
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
  }
  finish4(p, m, n, s);
}

// split-K (over 2048-element slabs) only when the N tiles alone leave most CUs idle
static void choose_split(int64_t N, int64_t K, int* splitk, int* slabs_per_split) {
  const int64_t tiles = (N + 15) / 16;
  const int nslabs = (int)((K + kSlabK - 1) / kSlabK);
  int s = 1;
  if ((N % 4) == 0) {
    while (tiles * s < 192 && s * 2 <= nslabs && s < 16) s *= 2;
  }
  int per = (nslabs + s - 1) / s;
  s = (nslabs + per - 1) / per;          // drop empty splits
  *splitk = s;
  *slabs_per_split = per;
}

int64_t gemm_skinny_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int s, per;
  choose_split(N, K, &s, &per);
  return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

int gemm_skinny(const GemmArgs& a, hipStream_t stream) {
  int splitk, per;
  choose_split(a.N, a.K, &splitk, &per);
  SkinnyParams p;
  p.A = a.A; p.B = a.B; p.SFA = a.SFA; p.SFB = a.SFB; p.D = a.D;
  p.partial = reinterpret_cast<float*>(a.workspace);
  p.alpha_dev = a.alpha_dev; p.bias = a.bias;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  p.tiles = (a.N + 15) / 16;
  p.slabs_per_split = per;
  if (splitk > 1) {
    const int64_t need = (int64_t)splitk * a.M * a.N * (int64_t)sizeof(float);
    if (!a.workspace || a.workspace_bytes < need)
      return fail(ARCQ_ERR_WORKSPACE, "arcq_gemm_nvfp4: split-K needs %lld B of workspace, got %lld", (long long)need,
                  (long long)a.workspace_bytes);
  }
  const dim3 grid((unsigned)p.tiles, (unsigned)splitk);
  // one workgroup per CU fits a 4-deep ring (152 KiB: K <= 6144 is then entirely in flight at once); with more tiles than CUs use a 2-deep ring so that two
  // workgroups share a CU and cover each other's start-up latency
  const bool deep = (int64_t)p.tiles * splitk <= 320;
  const size_t lds = (size_t)(deep ? 4 : 2) * kStageBytes + kSkWaves * 64 * 4 * sizeof(float);
  auto kern = deep ? gemm_skinny_kernel<4> : gemm_skinny_kernel<2>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[deep ? 1 : 0]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (skinny): cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    attr_set[deep ? 1 : 0] = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(kSkWaves * 64), lds, stream, p);
  if (splitk > 1) {
    const int64_t quads = ((int64_t)a.M * a.N + 3) / 4;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, p, splitk);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (skinny): launch failed: %s", hipGetErrorString(e));
  return ARCQ_OK;
}

}  // namespace arcq
