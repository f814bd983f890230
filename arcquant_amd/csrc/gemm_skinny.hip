// ARC-NVFP4 GEMM for decode shapes (M <= 16) with FEW weight tiles (N < 5120): the latency-bound member of the
// decode pair (gemm_decode.hip is the throughput-bound one; c_api.hip picks).
//
// Replaces the CUTLASS 128x128x128 block-scaled GEMM of the reference (kernels/src/nvfp4.cu:35-132)
// for the shapes where that kernel leaves 127/128 of its M tile empty ("decode").
//
// Roofline: bytes per launch = N*K*9/16 (packed B + scale bytes) + M*K*9/16 + M*N*2; everything else is
// on-chip.  Measured floor for ONE short kernel that only reads that many bytes (tools/probe_stream.hip):
// 9.6 MB -> 3.3 us, 33.5 MB -> 6.5 us.  At these sizes a workgroup sees 2-5 items in its whole life, so what
// counts is the start-up path (cold instruction cache, first HBM round trip), not steady-state issue rate:
//   * persistent workgroups (8 or 16 wave64); a workgroup walks "items" = (tile of 16 weight rows, slab of
//     kWaves*128 K elements).  Per item every thread fetches exactly ONE 16-byte unit of packed B with a plain
//     global_load_dwordx4: a wave covers two 512-byte row segments (full lines; loading B straight into the
//     MFMA operand layout, 16 rows x 64 B per instruction, reaches only 0.6-3.8 TB/s: tools/probe_rows.hip).
//     Three items are requested before the first is used; the item loop is ROLLED (ring rotated by moves) to
//     keep the code small -- the rotation makes hipcc wait for the youngest load, which costs steady-state
//     depth this kernel does not live long enough to use (the unrolled, exactly-counted ring of gemm_decode.hip
//     measured 5.9 -> 6.7 us here on N=K=3584 and wins only from N = 5120 up)
//   * the packed bytes are transposed into the MFMA operand layout through a small double-buffered LDS
//     image (padded rows: conflict-free), one barrier per item
//   * a tile's 16 rows are {32j + 4t + i} of a 128-row super-tile: their scale bytes then fill whole
//     64-byte lines of the CUTLASS-swizzled scale layout (4x fewer scale-line fetches than 16 consecutive
//     rows), and a lane still ends up with 4 consecutive output columns
//   * operands are dequantised in registers to fp16 (exact, gemm_common.hpp) and contracted on
//     v_mfma_f32_16x16x32_f16 with the weights as the MFMA "A" operand
//   * cross-wave reduction of the 16x16 fp32 tile through LDS; epilogue fused (alpha, optional bias,
//     bf16 rounding); split-K over slabs (second pass) only when the tiles alone cannot fill the chip.
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
  const uint16_t* residual;
  int M, N, K;
  float alpha_host;
  int out_dtype;
  int tiles;              // ceil(N / 16)
  int slabs_per_split;
};

// A workgroup of kWaves wave64 walks items of kWaves*128 K elements: one 16-byte unit per thread (loader role), one
// 128-element MFMA chunk per wave (compute role).  kWaves = 16 (1024 threads, 2048-element items) halves the number
// of barriers per byte and doubles the waves that hide each other's latencies; its fp16 A image only fits for M <= 8.
template <int kWaves>
struct SkinnyCfg {
  static constexpr int kSlabK = kWaves * 128;            // K elements per item
  static constexpr int kSlabBytes = kSlabK / 2;          // packed bytes per row per item
  static constexpr int kUnits = kSlabBytes / 16;         // 16-byte units per row per item (= threads per row)
  static constexpr int kSlabAtoms = kSlabK / 64;         // scale-factor atoms per item
  // LDS per buffer: packed B image [16 rows][kSlabBytes + 16] and DEQUANTISED (fp16) A image [M + 1 tokens]
  // [2*kSlabK + 16] (the extra token row stays zero and serves every MFMA column >= M)
  static constexpr int kBStride = kSlabBytes + 16;
  static constexpr int kBImg = 16 * kBStride;
  static constexpr int kAStride = kSlabK * 2 + 16;
  static constexpr int kRedBytes = kWaves * 64 * 4 * (int)sizeof(float);
  static int lds_bytes(int M) { return 2 * (kBImg + (M + 1) * kAStride) + kRedBytes; }
};

struct ItemRegs {
  uint4 b;             // loader role: this thread's 16-byte unit of packed B
  uint4 a;             // loader role (token rows only): 16-byte unit of packed A
  uint32_t sa;         // ... and the two scale bytes of that unit
  uint32_t sb;         // compute role: this lane's 4 scale bytes (one atom) of B for its wave's chunk
};

// At this size the kernel is INSTRUCTION-ISSUE bound, not only HBM bound (a 9.6 MB launch lasts ~3 us = ~7000
// cycles per wave): every address below is carried incrementally and every mask is hoisted out of the item loop.
template <int kWaves>
__global__ __launch_bounds__(kWaves * 64, 4) void gemm_skinny_kernel(SkinnyParams p) {
  using C = SkinnyCfg<kWaves>;
  constexpr int kSlabK = C::kSlabK, kSlabBytes = C::kSlabBytes, kSlabAtoms = C::kSlabAtoms, kUnits = C::kUnits;
  constexpr int kBStride = C::kBStride, kBImg = C::kBImg, kAStride = C::kAStride, kSkWaves = kWaves;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lds_buf = kBImg + (p.M + 1) * kAStride;
  float* red = reinterpret_cast<float*>(smem + 2 * lds_buf);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int q = lane >> 4;           // K quarter of the wave's 128-element chunk
  const int rl = lane & 15;          // MFMA row index (weight row rho) / token index
  const int ld_rho = tid / kUnits;   // loader role: row of the tile (B) / token (A) ...
  const int ld_u = tid % kUnits;     // ... and 16-byte unit (32 elements) within the row segment of this item

  const int atoms_k = p.K >> 6;
  const uint32_t half_k = (uint32_t)p.K >> 1;
  const int slab_begin = blockIdx.y * p.slabs_per_split;
  const int nslabs = min((p.K + kSlabK - 1) / kSlabK, slab_begin + p.slabs_per_split) - slab_begin;
  const int G = gridDim.x;
  const int my_tiles = (p.tiles - (int)blockIdx.x + G - 1) / G;
  const int nitems = my_tiles * nslabs;
  const bool m_ok = rl < p.M;
  const bool a_loader = ld_rho < p.M;

  // read before the ring starts: a load consumed inside the item loop would wait for every ring load before it
  const float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);

  // A tile's 16 rows are {128*T + 32*j + 4*t + i}: rho = 4j + i
  const int ld_rowpart = (ld_rho >> 2) * 32 + (ld_rho & 3);
  const int cm_rowpart = (rl >> 2) * 32 + (rl & 3);
  const uint32_t k_first = (uint32_t)slab_begin * kSlabBytes + ld_u * 16;      // byte offset of this thread's unit in slab 0
  const uint32_t k_last = half_k - 16u;                                        // clamp for the partial tail slab
  const uint32_t sfb_lane = (rl & 3) * 16 + (rl >> 2) * 4 + (2 * wave + (q >> 1)) * 512 + (uint32_t)slab_begin * kSlabAtoms * 512u;
  const uint32_t sfb_small = (rl & 3) * 16 + (rl >> 2) * 4;
  const uint32_t sfa_first = (uint32_t)ld_rho * 16 + ((uint32_t)slab_begin * kSlabAtoms + (ld_u >> 1)) * 512u + (ld_u & 1) * 2;
  const uint32_t sfa_last = (uint32_t)ld_rho * 16 + (uint32_t)(atoms_k - 1) * 512u + (ld_u & 1) * 2;

  // ---- issue side state (runs kRing items ahead of the compute side); all offsets are carried incrementally
  int iss_tile = blockIdx.x, iss_slab = 0, issued = 0;
  uint32_t b_row = 0, sfb_row = 0, sfb_rowmax = 0;       // per-tile parts
  auto issue_tile_setup = [&]() {
    const int tile_part = (iss_tile >> 3) * 128 + (iss_tile & 7) * 4;
    b_row = (uint32_t)min(tile_part + ld_rowpart, p.N - 1) * half_k;
    sfb_row = (uint32_t)(iss_tile >> 3) * atoms_k * 512u + (iss_tile & 7) * 64u;
    sfb_rowmax = sfb_row + (uint32_t)(atoms_k - 1) * 512u + sfb_small;
  };
  issue_tile_setup();
  uint32_t k_cur = k_first, sfb_cur = sfb_lane, sfa_cur = sfa_first;
  const uint32_t a_row = (uint32_t)ld_rho * half_k;
  auto issue_next = [&](ItemRegs& r) {
    if (issued < nitems) {                                     // wave-uniform
      const uint32_t koff = min(k_cur, k_last);
      {
        typedef uint32_t sk_u32x4 __attribute__((ext_vector_type(4)));
        const sk_u32x4 wv = ARCQ_WLOAD(reinterpret_cast<const sk_u32x4*>(p.B + (size_t)(b_row + koff)));
        r.b = make_uint4(wv.x, wv.y, wv.z, wv.w);
      }
      r.sb = ARCQ_WLOAD(reinterpret_cast<const uint32_t*>(p.SFB + (size_t)min(sfb_row + sfb_cur, sfb_rowmax)));
      if (a_loader) {
        r.a = *reinterpret_cast<const uint4*>(p.A + (size_t)(a_row + koff));
        r.sa = *reinterpret_cast<const uint16_t*>(p.SFA + (size_t)min(sfa_cur, sfa_last));
      }
      ++issued;
      k_cur += kSlabBytes; sfb_cur += kSlabAtoms * 512u; sfa_cur += kSlabAtoms * 512u;
      if (++iss_slab == nslabs) {
        iss_slab = 0; iss_tile += G;
        k_cur = k_first; sfb_cur = sfb_lane; sfa_cur = sfa_first;
        issue_tile_setup();
      }
    }
  };

  // Code size matters more than steady-state depth here: every launch starts with a cold instruction cache and
  // most of this kernel runs exactly once per workgroup, so the item loop is ROLLED (the register ring advances
  // by moves instead of by unrolling) and the per-tile epilogue exists once.
  ItemRegs r0, r1, r2;
  r0.b = r0.a = r1.b = r1.a = r2.b = r2.a = make_uint4(0, 0, 0, 0);
  r0.sb = r0.sa = r1.sb = r1.sa = r2.sb = r2.sa = 0;
  issue_next(r0);
  issue_next(r1);
  issue_next(r2);

  // the spare token row of both A images stays zero; MFMA columns >= M read it
  if (ld_rho == 0) {
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<uint4*>(smem + b2 * lds_buf + kBImg + p.M * kAStride + ld_u * 64 + c * 16) = make_uint4(0, 0, 0, 0);
  }

  const int wrb_off = ld_rho * kBStride + ld_u * 16;                  // loader role: packed B unit
  const int wra_off = kBImg + ld_rho * kAStride + ld_u * 64;          // loader role: 32 dequantised A values (64 B)
  const int wra_swz = (ld_u >> 1) & 3;                                // 16-byte chunk c is stored at position c ^ swz
  const int rdb_off = rl * kBStride + (4 * wave + q) * 16;            // compute role: this lane's packed B fragment
  const int rda_off = kBImg + min(rl, p.M) * kAStride + (4 * wave + q) * 64;   // ... and its four fp16 A fragments
  const int rda_swz = (2 * wave + (q >> 1)) & 3;
  const int sh = (q & 1) * 16;                                        // which two of the atom's four B scale bytes
  const int lane_atom = slab_begin * kSlabAtoms + 2 * wave + (q >> 1);   // compute role: atom of slab 0
  const int ld_atom = slab_begin * kSlabAtoms + (ld_u >> 1);             // loader role (A): atom of slab 0
  int parity = 0;

#pragma unroll 1
  for (int cur_tile = blockIdx.x; cur_tile < p.tiles; cur_tile += G) {
    const bool n_ok = (cur_tile >> 3) * 128 + (cur_tile & 7) * 4 + cm_rowpart < p.N;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int cur_slab = 0; cur_slab < nslabs; ++cur_slab) {
      unsigned char* buf = smem + parity * lds_buf;
      parity ^= 1;
      *reinterpret_cast<uint4*>(buf + wrb_off) = r0.b;
      if (a_loader) {
        // dequantise this thread's 32 activations once for the whole workgroup (cost scales with M, not 16)
        const uint32_t sa = ld_atom + cur_slab * kSlabAtoms < atoms_k ? r0.sa : 0u;
        const f16x2 s0 = sf_pair_at(sa, 0), s1 = sf_pair_at(sa, 8);
        Frag8 f0 = dequant8(r0.a.x, s0), f1 = dequant8(r0.a.y, s0), f2 = dequant8(r0.a.z, s1), f3 = dequant8(r0.a.w, s1);
        *reinterpret_cast<uint4*>(buf + wra_off + ((0 ^ wra_swz) << 4)) = f0.u;
        *reinterpret_cast<uint4*>(buf + wra_off + ((1 ^ wra_swz) << 4)) = f1.u;
        *reinterpret_cast<uint4*>(buf + wra_off + ((2 ^ wra_swz) << 4)) = f2.u;
        *reinterpret_cast<uint4*>(buf + wra_off + ((3 ^ wra_swz) << 4)) = f3.u;
      }
      uint32_t bs = r0.sb;
      r0 = r1;                                                 // advance the ring ...
      r1 = r2;
      issue_next(r2);                                          // ... and refill its tail: kRing items ahead
      __syncthreads();
      const bool live = lane_atom + cur_slab * kSlabAtoms < atoms_k;   // false only in the partial tail slab
      bs = (live && n_ok) ? bs : 0u;
      const uint4 bq = *reinterpret_cast<const uint4*>(buf + rdb_off);
      Frag8 a0, a1, a2, a3;
      a0.u = *reinterpret_cast<const uint4*>(buf + rda_off + ((0 ^ rda_swz) << 4));
      a1.u = *reinterpret_cast<const uint4*>(buf + rda_off + ((1 ^ rda_swz) << 4));
      a2.u = *reinterpret_cast<const uint4*>(buf + rda_off + ((2 ^ rda_swz) << 4));
      a3.u = *reinterpret_cast<const uint4*>(buf + rda_off + ((3 ^ rda_swz) << 4));
      const f16x2 sb0 = sf_pair_at(bs, sh), sb1 = sf_pair_at(bs, sh + 8);
      Frag8 b0 = dequant8(bq.x, sb0), b1 = dequant8(bq.y, sb0), b2 = dequant8(bq.z, sb1), b3 = dequant8(bq.w, sb1);
      // weights are the MFMA A operand (rows i = rho), activations the B operand (cols j = token)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc, 0, 0, 0);
    }

    // ---- tile done: cross-wave reduction; lane holds C[rho = 4q + r][token = rl]
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
      const int nn = (cur_tile >> 3) * 128 + (cur_tile & 7) * 4 + q * 32;   // rho = 4q + r -> 4 consecutive columns
      if (m_ok && nn < p.N) {
        if (gridDim.y == 1) {
          finish4<uint32_t>(p, alpha, rl, nn, sum);
        } else {
          float* o = p.partial + ((uint32_t)(blockIdx.y * p.M + rl) * (uint32_t)p.N + (uint32_t)nn);
          for (int r = 0; r < 4; ++r) if (nn + r < p.N) o[r] = sum[r];
        }
      }
    }
    // `red` is next written after at least one more __syncthreads (the next tile's first item), so no barrier here
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
  finish4<size_t>(p, p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f), m, n, s);
}

int gemm_splitk_finish(const GemmArgs& a, int splitk, hipStream_t stream) {
  SkinnyParams p{};
  p.D = a.D; p.partial = reinterpret_cast<float*>(a.workspace);
  p.alpha_dev = a.alpha_dev; p.bias = a.bias; p.residual = a.residual;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  const int64_t quads = ((int64_t)a.M * a.N + 3) / 4;
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, p, splitk);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (split-K finish): launch failed: %s", hipGetErrorString(e));
  return ARCQ_OK;
}

// split-K (over whole items) only when the tiles alone leave most CUs idle AND every split keeps >= 8 items: the
// second pass costs a launch (~5 us in a graph), more than a short serial K loop (N=1024 K=4160: 14.3 us split,
// measured, against ~7 us for the same per-workgroup work unsplit)
static void choose_split(int64_t N, int64_t K, int slab_k, int* splitk, int* slabs_per_split) {
  const int64_t tiles = ((N + 127) / 128) * 8;
  const int nslabs = (int)((K + slab_k - 1) / slab_k);
  int s = 1;
  if ((N % 4) == 0) {
    while (tiles * s < 192 && nslabs / (s * 2) >= 8 && s < 16) s *= 2;
  }
  int per = (nslabs + s - 1) / s;
  s = (nslabs + per - 1) / per;          // drop empty splits
  *splitk = s;
  *slabs_per_split = per;
}

// 16 waves (1024 threads, one workgroup per CU) while one workgroup per CU covers all tiles and the fp16 A image
// fits (M <= 8); 8 waves (two workgroups per CU) otherwise.  Measured on MI355X (tools/decode_bench.py, M = 1..4):
// N=4096 KQ=4096: 16 waves 7.1 us vs 8 waves 7.2 us;  N=14336 KQ=4096: 8 waves 15.3 us vs 16 waves 19.2 us.
static int skinny_waves(int64_t M, int64_t N) {
  static const int forced = getenv("ARCQ_SKINNY_WAVES") ? atoi(getenv("ARCQ_SKINNY_WAVES")) : 0;
  if (forced == 8 || (forced == 16 && M <= 8)) return forced;
  const int64_t tiles = ((N + 127) / 128) * 8;
  return (M <= 8 && tiles <= 256) ? 16 : 8;
}

int64_t gemm_skinny_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int s, per;
  choose_split(N, K, skinny_waves(M, N) * 128, &s, &per);
  return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

template <int kWaves>
static int launch_skinny(const SkinnyParams& p, int splitk, hipStream_t stream) {
  using C = SkinnyCfg<kWaves>;
  const int lds = C::lds_bytes(p.M);
  static LdsOptIn lds_opt;             // per kernel instantiation, per device
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_skinny_kernel<kWaves>), lds_opt, C::lds_bytes(kWaves == 16 ? 8 : 16),
                                  "arcq_gemm_nvfp4 (skinny)"))
    return rc;
  // persistent workgroups, each walks tiles blockIdx.x, +grid, ...; 8-wave groups fit two per CU
  static const int forced_grid = getenv("ARCQ_SKINNY_GRID") ? atoi(getenv("ARCQ_SKINNY_GRID")) : 0;
  const int max_wg = forced_grid > 0 ? forced_grid : (kWaves == 16 ? 256 : 512);
  const int per_split = max_wg / splitk > 0 ? max_wg / splitk : 1;
  const int gx = p.tiles < per_split ? p.tiles : per_split;
  hipLaunchKernelGGL(gemm_skinny_kernel<kWaves>, dim3((unsigned)gx, (unsigned)splitk), dim3(kWaves * 64), lds, stream, p);
  return ARCQ_OK;
}

int gemm_skinny(const GemmArgs& a, hipStream_t stream) {
  const int waves = skinny_waves(a.M, a.N);
  int splitk, per;
  choose_split(a.N, a.K, waves * 128, &splitk, &per);
  SkinnyParams p;
  p.A = a.A; p.B = a.B; p.SFA = a.SFA; p.SFB = a.SFB; p.D = a.D;
  p.partial = reinterpret_cast<float*>(a.workspace);
  p.alpha_dev = a.alpha_dev; p.bias = a.bias; p.residual = a.residual;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  p.tiles = ((a.N + 127) / 128) * 8;
  p.slabs_per_split = per;
  if (splitk > 1) {
    const int64_t need = (int64_t)splitk * a.M * a.N * (int64_t)sizeof(float);
    if (!a.workspace || a.workspace_bytes < need)
      return fail(ARCQ_ERR_WORKSPACE, "arcq_gemm_nvfp4: split-K needs %lld B of workspace, got %lld", (long long)need,
                  (long long)a.workspace_bytes);
  }
  if ((int64_t)a.N * (a.K / 2) >= ((int64_t)1 << 32) || (int64_t)((a.N + 127) / 128) * 128 * (a.K / 16) >= ((int64_t)1 << 32))
    return fail(ARCQ_ERR_UNSUPPORTED, "arcq_gemm_nvfp4 (skinny): operand larger than 4 GiB");
  const int rc = waves == 16 ? launch_skinny<16>(p, splitk, stream) : launch_skinny<8>(p, splitk, stream);
  if (rc != ARCQ_OK) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_gemm_nvfp4 (skinny): launch failed: %s", hipGetErrorString(e));
  if (splitk > 1) return gemm_splitk_finish(a, splitk, stream);
  return ARCQ_OK;
}

}  // namespace arcq
