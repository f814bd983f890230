// ARC-NVFP4 GEMM for decode shapes (M <= 16): a weight-streaming, HBM-bound kernel for gfx950.
//
// Replaces the CUTLASS 128x128x128 block-scaled GEMM of the reference (kernels/src/nvfp4.cu:35-132)
// for the shapes where that kernel leaves 127/128 of its M tile empty ("decode").
//
// Roofline: bytes per launch = N*K*9/16 (packed B + scale bytes) + M*K*9/16 + M*N*2; everything else
// is on-chip.  Design for that bound:
//   * one workgroup per 16 output columns (16 rows of B), its kWaves wave64s split K in interleaved
//     128-element chunks; with split-K over blockIdx.y when N/16 alone cannot fill 256 CUs
//   * B and its scale bytes go HBM -> VGPR directly (16 B / lane, each B byte read exactly once, no LDS
//     round trip); several chunks are in flight per wave before the first use
//   * dequantise in registers to fp16 (exact, gemm_common.hpp) and contract on
//     v_mfma_f32_16x16x32_f16 with the weights as the MFMA "A" operand, so that a lane ends up with
//     4 consecutive output columns of one token (one 8-byte store)
//   * cross-wave reduction of the 16x16 fp32 tile through 1 KB of LDS per wave; epilogue fused
//     (alpha, optional bias, bf16 rounding) unless split-K needs the second pass.
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

template <int kWaves, int kUnroll, int kDbg = 0>
__global__ __launch_bounds__(kWaves * 64) void gemm_skinny_kernel(SkinnyParams p) {
  __shared__ float red[kWaves][64][4];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = lane >> 4;           // K quarter of a 128-element chunk
  const int rl = lane & 15;          // weight row within the tile / token index
  const int n0 = blockIdx.x * 16;
  const int n = n0 + rl;
  const int m = rl;
  const int atoms_k = p.K >> 6;
  const int nchunks = (p.K + 127) >> 7;
  const int slot = blockIdx.y * kWaves + wave;
  const int nslots = gridDim.y * kWaves;
  const int half_k = p.K >> 1;

  const bool n_ok = n < p.N;
  const bool m_ok = m < p.M;
  // Loads are unconditional (addresses clamped into the buffers) and dead lanes are neutralised by
  // zeroing their scale bytes afterwards: a branch around each load would serialise the unrolled
  // prefetch (cdna_hip_programming.md, "Three .s-level traps" (c)).
  const uint8_t* brow = p.B + (size_t)(n_ok ? n : p.N - 1) * half_k + (q & 1) * 16;
  const uint8_t* arow = p.A + (size_t)(m_ok ? m : p.M - 1) * half_k + (q & 1) * 16;
  const uint8_t* bsf = p.SFB + sf_atom_offset(n_ok ? n : p.N - 1, 0, atoms_k);
  const uint8_t* asf = p.SFA + sf_atom_offset(m_ok ? m : p.M - 1, 0, atoms_k);
  const int sh = (q & 1) * 16;       // which two of the atom's four scale bytes this lane uses

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};

  for (int c0 = slot; c0 < nchunks; c0 += nslots * kUnroll) {
    uint4 bq[kUnroll], aq[kUnroll];
    uint32_t bs[kUnroll], as[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int c = c0 + u * nslots;
      const int atom = 2 * c + (q >> 1);           // this lane's 64-element K atom
      const bool live = atom < atoms_k;            // false past the end of K (also when c >= nchunks)
      const int at = live ? atom : atoms_k - 1;
      bq[u] = *reinterpret_cast<const uint4*>(brow + (size_t)at * 32);
      aq[u] = *reinterpret_cast<const uint4*>(arow + (size_t)at * 32);
      const uint32_t sbv = *reinterpret_cast<const uint32_t*>(bsf + (size_t)at * 512);
      const uint32_t sav = *reinterpret_cast<const uint32_t*>(asf + (size_t)at * 512);
      bs[u] = (live && n_ok) ? sbv : 0u;
      as[u] = (live && m_ok) ? sav : 0u;
    }
    // Every load of this pass is issued before the first use: hipcc otherwise sinks each chunk's loads next
    // to its MFMAs and the wave pays one HBM round trip per chunk instead of one per pass.
    __builtin_amdgcn_sched_barrier(0);
    if (kDbg == 1) {   // tuning aid: loads only (keep them live), no dequantisation / MFMA
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
        acc[0] += __uint_as_float((bq[u].x ^ bq[u].y ^ bq[u].z ^ bq[u].w ^ aq[u].x ^ aq[u].w ^ bs[u] ^ as[u]) & 0x3fffffffu);
      continue;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const f16x2 sb0 = sf_pair((bs[u] >> sh) & 0xffu), sb1 = sf_pair((bs[u] >> (sh + 8)) & 0xffu);
      const f16x2 sa0 = sf_pair((as[u] >> sh) & 0xffu), sa1 = sf_pair((as[u] >> (sh + 8)) & 0xffu);
      Frag8 b0 = dequant8(bq[u].x, sb0), b1 = dequant8(bq[u].y, sb0), b2 = dequant8(bq[u].z, sb1), b3 = dequant8(bq[u].w, sb1);
      Frag8 a0 = dequant8(aq[u].x, sa0), a1 = dequant8(aq[u].y, sa0), a2 = dequant8(aq[u].z, sa1), a3 = dequant8(aq[u].w, sa1);
      // weights are the MFMA A operand (rows i = n), activations the B operand (cols j = m)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc, 0, 0, 0);
    }
  }

  // cross-wave reduction; lane holds C[n = n0 + 4q + r][m = rl]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
  __syncthreads();
  if (wave == 0) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += red[w][lane][r];
    if (m_ok) {
      const int nn = n0 + 4 * q;
      if (gridDim.y == 1) {
        finish4(p, m, nn, s);
      } else {
        float* o = p.partial + ((size_t)blockIdx.y * p.M + m) * p.N + nn;
        for (int r = 0; r < 4; ++r) if (nn + r < p.N) o[r] = s[r];
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

// ARCQ_SKINNY_WAVES (tuning only): waves per workgroup, 8 or 16
static int skinny_waves_override() {
  static const int v = [] {
    const char* e = getenv("ARCQ_SKINNY_WAVES");
    return e ? atoi(e) : 0;
  }();
  return v;
}

static void choose_split(int64_t M, int64_t N, int64_t K, int* waves, int* splitk) {
  (void)M;
  const int64_t tiles = (N + 15) / 16;
  const int64_t nchunks = (K + 127) / 128;
  int w = skinny_waves_override();
  if (w != 8 && w != 16) w = (tiles <= 512) ? 16 : 8;   // few tiles: more waves per tile keep more loads in flight
  *waves = w;
  int s = 1;
  // fill ~256 CUs with at least one workgroup each, but keep >= 1 chunk per wave; split-K needs N % 4 == 0
  if ((N % 4) == 0) {
    while (tiles * s < 256 && nchunks / (w * (s * 2)) >= 1 && s < 16) s *= 2;
  }
  *splitk = s;
}

int64_t gemm_skinny_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int w, s;
  choose_split(M, N, K, &w, &s);
  return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

int gemm_skinny(const GemmArgs& a, hipStream_t stream) {
  int waves, splitk;
  choose_split(a.M, a.N, a.K, &waves, &splitk);
  SkinnyParams p;
  p.A = a.A; p.B = a.B; p.SFA = a.SFA; p.SFB = a.SFB; p.D = a.D;
  p.partial = reinterpret_cast<float*>(a.workspace);
  p.alpha_dev = a.alpha_dev; p.bias = a.bias;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  if (splitk > 1) {
    const int64_t need = (int64_t)splitk * a.M * a.N * (int64_t)sizeof(float);
    if (!a.workspace || a.workspace_bytes < need)
      return fail(ARCQ_ERR_WORKSPACE, "arcq_gemm_nvfp4: split-K needs %lld B of workspace, got %lld", (long long)need,
                  (long long)a.workspace_bytes);
  }
  const dim3 grid((unsigned)((a.N + 15) / 16), (unsigned)splitk);
  static const int dbg = getenv("ARCQ_SKINNY_DBG") ? atoi(getenv("ARCQ_SKINNY_DBG")) : 0;
  if (dbg == 1 && waves == 16) hipLaunchKernelGGL((gemm_skinny_kernel<16, 3, 1>), grid, dim3(16 * 64), 0, stream, p);
  else if (dbg == 1) hipLaunchKernelGGL((gemm_skinny_kernel<8, 4, 1>), grid, dim3(8 * 64), 0, stream, p);
  else if (waves == 16) hipLaunchKernelGGL((gemm_skinny_kernel<16, 3>), grid, dim3(16 * 64), 0, stream, p);
  else hipLaunchKernelGGL((gemm_skinny_kernel<8, 4>), grid, dim3(8 * 64), 0, stream, p);
  if (splitk > 1) {
    const int64_t quads = ((int64_t)a.M * a.N + 3) / 4;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, p, splitk);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (skinny): launch failed: %s", hipGetErrorString(e));
  return ARCQ_OK;
}

}  // namespace arcq
