// Reorder + NVFP4 quantise kernels for gfx950 (activation / weight / fused RMSNorm).
//
// Replaces kernels/src/reorder.cu (reorder_{x,w}_kernel, reorder32_{x,w}_kernel), kernels/src/down.cu
// (down32_{x,w}_kernel) and kernels/src/rmsnorm.cu (rmsnorm_x_kernel) of the reference with ONE
// KQ-generic kernel family: the reference instantiates a template per hidden size and needs a host
// index_select pre-pass for rows that do not fit 48 KB of static shared memory (bindings.cpp:29-38);
// a CDNA4 CU has 160 KB of LDS, so every supported row (<= 28672 bf16 = 56 KB) is staged and
// gathered in LDS directly.
//
// Work decomposition (HBM-bound, integer-exact output):
//   grid  = min(rows, kMaxBlocks) workgroups of 256 threads (4 wave64), each looping over rows
//   row   -> LDS with 16-byte coalesced loads; thread t owns the 16-element groups t, t+256, ...
//   group -> gather 16 bf16 from LDS by reorder_index, |max| -> ue4m3 scale -> e2m1 codes -> 8 packed
//            bytes (one 8-byte store) + 1 scale byte at its swizzled offset; groups in the outlier
//            tail additionally emit the quantised residual (x) or a duplicate (w).
//
// Numerics follow the kernel text of the reference exactly (see oracle/arcq_oracle.c for the CPU
// restatement these kernels are tested against, byte for byte).
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "arcq_device.hpp"
#include "arcq_internal.hpp"
#include "quantize_device.hpp"

namespace arcq {

enum : int { kModeX = 0, kModeW = 1, kModeRms = 2 };

constexpr int kQuantThreads = 256;

__device__ __forceinline__ uint4 silu_mul_chunk(const uint4 g, const uint4 u) {
  const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, uw[4] = {u.x, u.y, u.z, u.w};
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    o[j] = silu_mul_bf16(gw[j] & 0xffffu, uw[j] & 0xffffu) | (silu_mul_bf16(gw[j] >> 16, uw[j] >> 16) << 16);
  return make_uint4(o[0], o[1], o[2], o[3]);
}
// eight activations from sixteen interleaved values (g0, u0, g1, u1, ...): every dword is one (gate, up) pair
__device__ __forceinline__ uint4 silu_mul_pairs(const uint4 lo, const uint4 hi) {
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    o[j] = silu_mul_bf16(w[2 * j] & 0xffffu, w[2 * j] >> 16) | (silu_mul_bf16(w[2 * j + 1] & 0xffffu, w[2 * j + 1] >> 16) << 16);
  return make_uint4(o[0], o[1], o[2], o[3]);
}
// chunk c (8 activations) of a row: kSiluHalves: gate at g[8c..], up at u[8c..]; kSiluPairs: pairs at g[16c..]
enum : int { kSiluNone = 0, kSiluHalves = 1, kSiluPairs = 2 };
template <int kSilu>
__device__ __forceinline__ uint4 silu_act_chunk(const uint16_t* g, const uint16_t* u, int64_t c) {
  if (kSilu == kSiluPairs)
    return silu_mul_pairs(*reinterpret_cast<const uint4*>(g + c * 16), *reinterpret_cast<const uint4*>(g + c * 16 + 8));
  return silu_mul_chunk(*reinterpret_cast<const uint4*>(g + c * 8), *reinterpret_cast<const uint4*>(u + c * 8));
}
template <int kSilu>
__device__ __forceinline__ uint32_t silu_act_elem(const uint16_t* g, const uint16_t* u, uint32_t i) {
  if (kSilu == kSiluPairs) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(g + 2 * i);
    return silu_mul_bf16(w & 0xffffu, w >> 16);
  }
  return silu_mul_bf16(g[i], u[i]);
}

// Sum of squares of one row in the reference's association order (rmsnorm.cu:113-154), so that the
// fp32 result is bit-identical to the oracle's: virtual thread v in [0, bdx = KQ/16) owns the 16-byte
// chunks v and bdx + v and accumulates their 16 squares sequentially (the caller passes the partial sums of
// v = tid and v = tid + 256); then the fixed tree s[v] += s[v + stride], stride = 256 ... 32, and a 32-lane shuffle.
// The same tree with TWO barriers instead of seven: the stride-256 step adds two values this thread computed itself;
// strides 128 and 64 only ever feed s[0..63], so wave 0 evaluates them for its 64 columns from four LDS reads;
// stride 32 and below are shuffles.  `s` is an LDS array of >= 512 floats.  Returns rstd = 1 / sqrt(total / KQ + eps) in every
// thread: thread 0 forms it (the correctly rounded double-precision evaluation is ~35 half-rate instructions -- done by all 256
// threads of all resident workgroups it was ~2 us of the 18.5 us launch at 4096 x 4096) and publishes it with the second barrier.
__device__ __forceinline__ float rms_rstd_tree(float* s, int bdx, float p_lo, float p_hi, int KQ, float eps) {
  const int tid = threadIdx.x;
  const float s256 = (tid + 256 < bdx) ? p_lo + p_hi : p_lo;           // stride 256
  if (tid < bdx) s[tid] = s256;
  __syncthreads();                                                     // (also publishes the staged row)
  if (tid < 64) {
    const float x0 = s[tid], x1 = s[tid + 64], x2 = s[tid + 128], x3 = s[tid + 192];
    const float y0 = (tid + 128 < bdx) ? x0 + x2 : x0;                 // stride 128: columns tid and tid + 64
    const float y1 = (tid + 192 < bdx) ? x1 + x3 : x1;
    float z = (tid + 64 < bdx) ? y0 + y1 : y0;                         // stride 64
    const float up = __shfl_down(z, 32, 64);
    if (tid < 32 && tid + 32 < bdx) z = z + up;                        // stride 32
    float val = tid < 32 ? z : 0.0f;
    val += __shfl_down(val, 16, 64);                                       // lane 0's cone = reference's
    val += dpp_row_shl<8>(val);                                            // strides 8 .. 1 stay inside lane 0's row of 16 (DPP)
    val += dpp_row_shl<4>(val);
    val += dpp_row_shl<2>(val);
    val += dpp_row_shl<1>(val);
    if (tid == 0) {
      const float var = val / (float)KQ + eps;                             // rmsnorm.cu:157
      s[256] = (float)(1.0 / sqrt((double)var));                           // oracle assumption A4
    }
  }
  __syncthreads();
  return s[256];
}

// Per-tensor dynamic scale (kModeX only, arcq_quantize_x_dyn / arcq_silu_mul_quantize_x_dyn), selected by kDyn:
//   kDynState: `dyn` = one abs-max word per workgroup of a preceding abs-max kernel (`nslots` of them, plain stores:
//              4096 same-address atomics, or a completion ticket taken by 2048 workgroups, serialise on the fabric and
//              cost 50-100 us -- measured -- where the data pass itself takes 15);
//   kDynLocal: every workgroup computes max|X| of the WHOLE (small) tensor itself -- decode-sized inputs then need a
//              single launch (see kDynLocalMaxBytes for where that pays).
// Every element is first divided by scale = amax * (1/2688) and rounded to bf16 -- exactly what torch's GPU `x / scale`
// with a 0-dim fp32 scale computes (model/qLlamaLayer.py:74-76) -- so the separate abs/max/div passes vanish.
// kSilu: the row is silu(gate) * up computed on the fly from X (gate) and Xup (up), both with row stride ldx
// (kSiluHalves), or from (gate, up) pairs interleaved in X (kSiluPairs).
// kDyn / kSilu are template parameters: as run-time branches inside the 16-element gather they cost the static
// quantiser 18 % (15.7 -> 18.5 us at 4096^2) and the dynamic one most of its time.
enum : int { kDynNone = 0, kDynState = 1, kDynLocal = 2 };
constexpr int kGatherCache = 2;   // groups per thread whose gather offsets live in registers across rows

template <int kVariant, int kMode, int kDyn, int kSilu>
__global__ __launch_bounds__(kQuantThreads) void quantize_rows_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Xup, int64_t ldx, const uint16_t* __restrict__ Wn, float eps,
    const int16_t* __restrict__ idx, uint8_t* __restrict__ Q, uint8_t* __restrict__ SF, int rows, int KQ, int KE,
    const unsigned int* __restrict__ dyn, int nslots, float* scale_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* row_lds = reinterpret_cast<uint16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + lds_row_bytes(KQ));   // kModeRms only: 512 floats of reduction scratch ...
  uint16_t* wn_lds = reinterpret_cast<uint16_t*>(smem + lds_row_bytes(KQ) + 512 * sizeof(float));   // ... and the norm weights

  const int tid = threadIdx.x;
  const int K = KQ + KE;
  const int G = KQ >> 4;
  const int P = (KQ - KE) >> 4;
  const int chunks = KQ >> 3;           // 16-byte chunks per row
  const int bdx = KQ >> 4;              // the reference's block size (rmsnorm.cu:269-270)
  // decode-sized inputs (a handful of rows) are split along the row as well: blockIdx.y owns a 4-aligned range of
  // groups; every workgroup still stages the whole row (the gather may touch any channel), which is an L2 hit
  const int g_per = (((G + (int)gridDim.y - 1) / (int)gridDim.y) + 3) & ~3;
  const int g_begin = (int)blockIdx.y * g_per;
  const int g_end = min(G, g_begin + g_per);
  float dyn_scale = 1.0f;
  if (kDyn != kDynNone) {
    __shared__ unsigned int wave_max[kQuantThreads / 64];
    uint32_t m = 0;
    if (kDyn == kDynLocal) {
      for (int r = 0; r < rows; ++r)
        for (int c = tid; c < chunks; c += kQuantThreads) m = absmax_bits_chunk(*reinterpret_cast<const uint4*>(X + (size_t)r * ldx + (size_t)c * 8), m);
    } else {
      for (int i = tid; i < nslots; i += kQuantThreads) m = max(m, dyn[i]);
    }
    m = wave_max_u32(m);
    if ((tid & 63) == 0) wave_max[tid >> 6] = m;
    __syncthreads();
    const unsigned int amax_bits = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    dyn_scale = bf16_bits_to_f32(amax_bits) * (1.0f / (448.0f * 6.0f));
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) scale_out[0] = dyn_scale;      // the caller's fp32 per-tensor scale
    // torch on the GPU divides a bf16 tensor by a 0-dim fp32 tensor in the COMMON dtype bf16: the scale operand is
    // rounded to bf16 at load (BinaryFunctor<BFloat16, BFloat16, BFloat16, DivFunctor>), the quotient is formed in fp32
    dyn_scale = round_to_bf16(dyn_scale);
  }
  const DynDiv dyn_div(dyn_scale, kDyn != kDynNone);       // x / scale, bit-exact (quantize_device.hpp)

  if (kMode == kModeRms) {
    // the gather reads the norm weight of every channel: 16 scattered 2-byte global loads per group cost 2.6x
    // the whole static quantiser (41 vs 16 us at 4096^2), one LDS copy per workgroup does not
    for (int c = threadIdx.x; c < chunks; c += kQuantThreads)
      lds_store_chunk(wn_lds, c, *reinterpret_cast<const uint4*>(Wn + (size_t)c * 8));
    // visible after the barriers of the first row's reduction
  }
  // Every row of this workgroup gathers through the same reorder_index: the LDS byte offsets of a thread's first kGatherCache
  // groups (all of them up to KQ = 8192) are computed ONCE -- per row that leaves one ds_read_u16 + one shift per element instead
  // of two index loads and ~5 integer instructions per element (the kernel is issue-bound).
  const bool cache_on = !(kSilu && gridDim.y > 1);           // that path gathers from global memory by element index
  uint32_t gofs[kGatherCache][16];
#pragma unroll
  for (int c = 0; c < kGatherCache; ++c) {
    const int g = min(g_begin + tid + c * kQuantThreads, G - 1);
    const uint4 i0 = *reinterpret_cast<const uint4*>(idx + (size_t)g * 16);
    const uint4 i1 = *reinterpret_cast<const uint4*>(idx + (size_t)g * 16 + 8);
    const uint32_t iw[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t pw = lds_pad_pair(iw[j]);
      gofs[c][2 * j] = (pw & 0xffffu) * 2;
      gofs[c][2 * j + 1] = (pw >> 16) * 2;
    }
  }
  auto run_rows = [&](auto fast_tag) __attribute__((always_inline)) {
  constexpr bool kFast = decltype(fast_tag)::value;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const uint16_t* xrow = X + (size_t)row * ldx;
    float rstd = 1.0f;
    if (kMode == kModeRms) {
      float part[2] = {0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < 2; ++k) {                              // bdx <= 512: virtual threads tid and tid + 256
        const int v = tid + k * kQuantThreads;
        if (v < bdx) {
          float acc = 0.0f;
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int c = it * bdx + v;
            uint4 d = *reinterpret_cast<const uint4*>(xrow + (size_t)c * 8);
            lds_store_chunk(row_lds, c, d);
            const uint32_t w4[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float a = bf16_bits_to_f32(w4[j] & 0xffffu), b = bf16_bits_to_f32(w4[j] >> 16);
              acc = acc + a * a;
              acc = acc + b * b;
            }
          }
          part[k] = acc;
        }
      }
      rstd = rms_rstd_tree(red, bdx, part[0], part[1], KQ, eps);
    } else {
      if (kSilu && gridDim.y > 1) {
        // decode-sized silu*up: a staged row would be recomputed (one exp per element) by each of the gridDim.y
        // workgroups that share it; gather straight from global instead, every element is then computed once
      } else if (kSilu) {
        const uint16_t* urow = Xup + (size_t)row * ldx;
        for (int c = tid; c < chunks; c += kQuantThreads)
          lds_store_chunk(row_lds, c, silu_act_chunk<kSilu>(xrow, urow, c));
      } else {
        for (int c = tid; c < chunks; c += kQuantThreads)
          lds_store_chunk(row_lds, c, *reinterpret_cast<const uint4*>(xrow + (size_t)c * 8));
      }
      __syncthreads();
    }

    uint8_t* qrow = Q + (size_t)row * (K >> 1);
    const uint16_t* urow_g = kSilu ? Xup + (size_t)row * ldx : nullptr;
    // one group: 16 gathered values -> codes + scale byte(s) at the group's augmented-K position
    auto finish = [&](int g, float (&v)[16]) __attribute__((always_inline)) {
      const bool tail = g >= P;
      // augmented-K position of this group (reorder.cu:139 / :451-452)
      int p;
      if (kVariant == ARCQ_VARIANT_G16) {
        p = g + (g > P ? g - P : 0);
      } else {
        const int g1 = g & ~1;
        p = g1 + (g1 > P ? g1 - P : 0) + (g & 1);
      }
      const int pr = p + (kVariant == ARCQ_VARIANT_G16 ? 1 : 2);

      if (kMode == kModeW) {
        GroupQ q = quantize_group<false, kVariant>(v);
        *reinterpret_cast<uint2*>(qrow + (size_t)p * 8) = q.packed;
        SF[sf_offset(row, p, K)] = (uint8_t)q.s8;
        if (tail) {                                              // duplicate: reorder.cu:306-316, 671-683
          *reinterpret_cast<uint2*>(qrow + (size_t)pr * 8) = q.packed;
          SF[sf_offset(row, pr, K)] = (uint8_t)q.s8;
        }
      } else if (!tail) {
        GroupQ q = quantize_group<false, kVariant>(v);
        *reinterpret_cast<uint2*>(qrow + (size_t)p * 8) = q.packed;
        // outside the outlier tail p == g, so the four lanes of a quad own the four bytes of ONE aligned dword of
        // the swizzled scale layout (P and G are multiples of 4): gather them and store once
        uint32_t w = q.s8;
        w |= dpp_row_shl_u32<1>(q.s8) << 8;                 // the quad's lanes i + 1 .. i + 3 (DPP, not three ds_bpermutes)
        w |= dpp_row_shl_u32<2>(q.s8) << 16;
        w |= dpp_row_shl_u32<3>(q.s8) << 24;
        if ((tid & 3) == 0) *reinterpret_cast<uint32_t*>(SF + sf_offset(row, p, K)) = w;
      } else {                                                   // residual: reorder.cu:166-198, 499-550
        GroupQ q = quantize_group<true, kVariant>(v);
        *reinterpret_cast<uint2*>(qrow + (size_t)p * 8) = q.packed;
        SF[sf_offset(row, p, K)] = (uint8_t)q.s8;
        GroupQ r = quantize_group<false, kVariant>(v);
        *reinterpret_cast<uint2*>(qrow + (size_t)pr * 8) = r.packed;
        SF[sf_offset(row, pr, K)] = (uint8_t)r.s8;
      }
    };
    auto scale_pair = [&](float& a, float& b, uint32_t off_a, uint32_t off_b) __attribute__((always_inline)) {
      if (kDyn != kDynNone) {                                   // torch: bf16(float(x) / scale)
        a = round_to_bf16(dyn_div.template div<kFast>(a));
        b = round_to_bf16(dyn_div.template div<kFast>(b));
      }
      if (kMode == kModeRms) {                                  // rmsnorm.cu:165-171
        a = round_to_bf16(a * bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(reinterpret_cast<const unsigned char*>(wn_lds) + off_a)) * rstd);
        b = round_to_bf16(b * bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(reinterpret_cast<const unsigned char*>(wn_lds) + off_b)) * rstd);
      }
    };
    // (a) the thread's first kGatherCache groups: LDS byte offsets computed once per workgroup (gofs)
#pragma unroll
    for (int c = 0; c < kGatherCache; ++c) {
      const int g = g_begin + tid + c * kQuantThreads;
      if (cache_on && g < g_end) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t oa = gofs[c][2 * j], ob = gofs[c][2 * j + 1];
          float a = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(reinterpret_cast<const unsigned char*>(row_lds) + oa));
          float b = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(reinterpret_cast<const unsigned char*>(row_lds) + ob));
          scale_pair(a, b, oa, ob);
          v[2 * j] = a;
          v[2 * j + 1] = b;
        }
        finish(g, v);
      }
    }
    // (b) further groups of long rows, and the decode-sized SiLU*up path that gathers straight from global memory
    for (int g = g_begin + tid + (cache_on ? kGatherCache * kQuantThreads : 0); g < g_end; g += kQuantThreads) {
      // reorder_index for this group: 16 x int16 = two 16-byte loads
      const uint4 i0 = *reinterpret_cast<const uint4*>(idx + (size_t)g * 16);
      const uint4 i1 = *reinterpret_cast<const uint4*>(idx + (size_t)g * 16 + 8);
      const uint32_t iw[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
      float v[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t ia = iw[j] & 0xffffu, ib = iw[j] >> 16;
        const uint32_t pw = lds_pad_pair(iw[j]), pa = pw & 0xffffu, pb = pw >> 16;      // the same elements in the padded LDS rows
        float a, b;
        if (kSilu && gridDim.y > 1) {
          a = bf16_bits_to_f32(silu_act_elem<kSilu>(xrow, urow_g, ia));
          b = bf16_bits_to_f32(silu_act_elem<kSilu>(xrow, urow_g, ib));
        } else {
          a = bf16_bits_to_f32(row_lds[pa]);
          b = bf16_bits_to_f32(row_lds[pb]);
        }
        scale_pair(a, b, pa * 2, pb * 2);
        v[2 * j] = a;
        v[2 * j + 1] = b;
      }
      finish(g, v);
    }
    __syncthreads();   // row_lds is rewritten by the next row
  }
  };
  if constexpr (kDyn == kDynNone) {
    run_rows(std::true_type{});
  } else {
    if (dyn_div.fast) run_rows(std::true_type{});
    else run_rows(std::false_type{});
  }
}

// The dynamic-scale activation quantiser for an input that is ALREADY in reordered channel order (reorder_index == NULL at the
// C-ABI = identity): the producing kernel applied the permutation when it stored (the fused gate|up GEMM writes
// act[m][inverse_index[j]]), so a group is 32 contiguous bytes -- no row staging in LDS, no index loads, no barrier per row,
// one group per thread over the flattened (row, group) space.  Same arithmetic, same bytes as quantize_rows_kernel on the
// un-permuted tensor with the permutation as reorder_index (tests).  Decode: 4 x 18944 in ~3 us instead of ~7.
template <int kVariant>
__global__ __launch_bounds__(kQuantThreads) void quantize_contig_dyn_kernel(const uint16_t* __restrict__ X, uint8_t* __restrict__ Q,
                                                                             uint8_t* __restrict__ SF, int rows, int KQ, int KE,
                                                                             const unsigned int* __restrict__ dyn, int nslots,
                                                                             float* scale_out) {
  __shared__ unsigned int wave_max[kQuantThreads / 64];
  const int tid = threadIdx.x;
  const int K = KQ + KE, G = KQ >> 4, P = (KQ - KE) >> 4;
  uint32_t m = 0;
  for (int i = tid; i < nslots; i += kQuantThreads) m = max(m, dyn[i]);
  m = wave_max_u32(m);
  if ((tid & 63) == 0) wave_max[tid >> 6] = m;
  __syncthreads();
  const unsigned int amax_bits = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
  float dyn_scale = bf16_bits_to_f32(amax_bits) * (1.0f / (448.0f * 6.0f));
  if (blockIdx.x == 0 && tid == 0) scale_out[0] = dyn_scale;
  dyn_scale = round_to_bf16(dyn_scale);                   // torch divides by the scale rounded to bf16 (see quantize_rows_kernel)
  const DynDiv dyn_div(dyn_scale, true);
  const int64_t total = (int64_t)rows * G;
  auto run = [&](auto fast_tag) __attribute__((always_inline)) {
  constexpr bool kFast = decltype(fast_tag)::value;
  for (int64_t t = (int64_t)blockIdx.x * kQuantThreads + tid; t < ((total + 3) & ~(int64_t)3); t += (int64_t)gridDim.x * kQuantThreads) {
    const bool live = t < total;                          // (total is a multiple of 4: whole quads are live or dead together)
    const int64_t tc = live ? t : total - 1;
    const int row = (int)(tc / G), g = (int)(tc - (int64_t)row * G);
    const uint16_t* src = X + (size_t)row * KQ + (size_t)g * 16;
    const uint4 d0 = *reinterpret_cast<const uint4*>(src), d1 = *reinterpret_cast<const uint4*>(src + 8);
    const uint32_t w[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    float v[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[2 * j] = round_to_bf16(dyn_div.template div<kFast>(bf16_bits_to_f32(w[j] & 0xffffu)));
      v[2 * j + 1] = round_to_bf16(dyn_div.template div<kFast>(bf16_bits_to_f32(w[j] >> 16)));
    }
    const bool tail = g >= P;
    int p;
    if (kVariant == ARCQ_VARIANT_G16) {
      p = g + (g > P ? g - P : 0);
    } else {
      const int g1 = g & ~1;
      p = g1 + (g1 > P ? g1 - P : 0) + (g & 1);
    }
    const int pr = p + (kVariant == ARCQ_VARIANT_G16 ? 1 : 2);
    uint8_t* qrow = Q + (size_t)row * (K >> 1);
    if (!tail) {
      GroupQ q = quantize_group<false, kVariant>(v);
      if (live) *reinterpret_cast<uint2*>(qrow + (size_t)p * 8) = q.packed;
      uint32_t sw = q.s8;                                 // the quad's four scale bytes are one aligned dword (P, G multiples of 4)
      sw |= dpp_row_shl_u32<1>(q.s8) << 8;                   // the quad's lanes i + 1 .. i + 3 (DPP, not three ds_bpermutes)
      sw |= dpp_row_shl_u32<2>(q.s8) << 16;
      sw |= dpp_row_shl_u32<3>(q.s8) << 24;
      if (live && (tid & 3) == 0) *reinterpret_cast<uint32_t*>(SF + sf_offset(row, p, K)) = sw;
    } else if (live) {                                    // residual channels: reorder.cu:166-198, 499-550
      GroupQ q = quantize_group<true, kVariant>(v);
      *reinterpret_cast<uint2*>(qrow + (size_t)p * 8) = q.packed;
      SF[sf_offset(row, p, K)] = (uint8_t)q.s8;
      GroupQ r = quantize_group<false, kVariant>(v);
      *reinterpret_cast<uint2*>(qrow + (size_t)pr * 8) = r.packed;
      SF[sf_offset(row, pr, K)] = (uint8_t)r.s8;
    }
  }
  };
  if (dyn_div.fast) run(std::true_type{});
  else run(std::false_type{});
}

// max|x| helpers.  |bf16| ordering == ordering of the low 15 bits, so integer max is exact.
constexpr int kAbsmaxThreads = 1024;     // one 16-wave workgroup per CU streams well and keeps the slot count small
constexpr int kAbsmaxMaxBlocks = 256;    // == ARCQ_DYN_STATE_BYTES / 4

// workgroup maximum of `m`; valid in thread 0
__device__ __forceinline__ uint32_t block_max_bits(uint32_t m) {
  __shared__ uint32_t wmax[kAbsmaxThreads / 64];
  m = wave_max_u32(m);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 64) {
    m = threadIdx.x < (blockDim.x >> 6) ? wmax[threadIdx.x] : 0u;
#pragma unroll
    for (int sh = 8; sh > 0; sh >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, sh, 64));
  }
  return m;
}

// slots[blockIdx.x] = max |X| over this workgroup's share (kAtomic: atomicMax into slots[0] instead, <= 256 of them)
template <bool kAtomic>
__global__ __launch_bounds__(kAbsmaxThreads) void absmax_bits_kernel(const uint16_t* __restrict__ X, int64_t n8, int64_t n,
                                                                      unsigned int* __restrict__ slots) {
  uint32_t m = 0;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (int64_t)gridDim.x * blockDim.x)
    m = absmax_bits_chunk(*reinterpret_cast<const uint4*>(X + c * 8), m);
  if (blockIdx.x == 0)
    for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) m = max(m, (uint32_t)X[i] & 0x7fffu);
  m = block_max_bits(m);
  if (threadIdx.x == 0) {
    if (kAtomic) atomicMax(slots, m);
    else slots[blockIdx.x] = m;
  }
}

// slots[blockIdx.x] = max |silu(G) * U| over a [rows, KQ] view of two strided operands (KQ % 8 == 0)
template <int kSilu>
__global__ __launch_bounds__(kAbsmaxThreads) void silu_mul_absmax_kernel(const uint16_t* __restrict__ G, const uint16_t* __restrict__ U,
                                                                          int64_t ldx, int rows, int chunks, unsigned int* __restrict__ slots) {
  uint32_t m = 0;
  const int64_t total = (int64_t)rows * chunks;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / chunks, c = i - r * chunks;
    m = absmax_bits_chunk(silu_act_chunk<kSilu>(G + r * ldx, U + r * ldx, c), m);
  }
  m = block_max_bits(m);
  if (threadIdx.x == 0) slots[blockIdx.x] = m;
}

__global__ void absmax_finish_kernel(unsigned int* slot, float* scale_out) {
  // torch's GPU true-divide by a host scalar multiplies by the fp32 reciprocal (BinaryDivTrueKernel):
  // `torch.max(x.abs()).float() / (448.0*6.0)` == amax * (1.0f / 2688.0f), bit for bit.
  float amax = bf16_bits_to_f32(*slot);
  scale_out[0] = amax * (1.0f / (448.0f * 6.0f));
}

// ----------------------------------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------------------------------
constexpr int kMaxQuantBlocks = 2048;   // 256 CUs x 8 resident workgroups, rows are grid-strided beyond

// grid.x = one workgroup per row (grid-strided beyond kMaxQuantBlocks); few rows (decode): each row is also split
// over up to 16 workgroups of >= 32 groups so that the chip is not idle
static void quant_grid(int64_t rows, int64_t KQ, int* grid, int* gsplit) {
  const int g = (int)(rows < kMaxQuantBlocks ? rows : kMaxQuantBlocks);
  int s = 1;
  while (g * s < 256 && s < 16 && (KQ / 16) / (s * 2) >= 32) s *= 2;
  *grid = g;
  *gsplit = s;
}

template <int kMode, int kDyn = kDynNone, int kSilu = kSiluNone>
static int launch_quantize(const void* X, const void* Wn, float eps, const int16_t* idx, uint8_t* Q, uint8_t* SF,
                           int64_t rows, int64_t KQ, int64_t KE, int variant, hipStream_t stream, const char* who,
                           const unsigned int* dyn = nullptr, int nslots = 0, float* scale_out = nullptr, const void* Xup = nullptr,
                           int64_t ldx = 0) {
  // KQ and KE in whole scale-factor atoms (64 elements = 4 groups): every size the reference dispatches (bindings.cpp:141-160),
  // every select_num it produces (multiples of 64) and every 64-aligned TP shard.  The kernel relies on it: the four
  // lanes of a quad store the four scale bytes of one aligned dword of the swizzled layout.
  if (rows < 0 || KQ <= 0 || (KQ % 64) || (KE % 64) || KE < 0 || KE > KQ)
    return fail(ARCQ_ERR_SHAPE, "%s: need KQ%%64==0, KE%%64==0, 0<=KE<=KQ (rows=%lld KQ=%lld KE=%lld)", who,
                (long long)rows, (long long)KQ, (long long)KE);
  if (variant != ARCQ_VARIANT_G16 && variant != ARCQ_VARIANT_G32)
    return fail(ARCQ_ERR_SHAPE, "%s: unknown variant %d", who, variant);
  if (KQ > 32767)   // int16 reorder_index (bindings.cpp:137)
    return fail(ARCQ_ERR_UNSUPPORTED, "%s: KQ=%lld does not fit an int16 reorder_index", who, (long long)KQ);
  if (kMode == kModeRms && (KQ < 2048 || KQ > 8192))   // reduction tree of rmsnorm.cu:130-146 is defined for 128..512 threads
    return fail(ARCQ_ERR_UNSUPPORTED, "%s: KQ=%lld outside the reference's RMSNorm range [2048, 8192]", who, (long long)KQ);
  if (rows == 0) return ARCQ_OK;
  if (!X || !idx || !Q || !SF || (kMode == kModeRms && !Wn)) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
  if (rows > INT32_MAX) return fail(ARCQ_ERR_UNSUPPORTED, "%s: too many rows", who);
  // 16-byte row / index loads, 8-byte code stores, 4-byte scale stores
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(Xup)) & 15)
    return fail(ARCQ_ERR_SHAPE, "%s: X and reorder_index must be 16-byte aligned", who);
  if ((reinterpret_cast<uintptr_t>(Q) & 7) || (reinterpret_cast<uintptr_t>(SF) & 3))
    return fail(ARCQ_ERR_SHAPE, "%s: the packed output must be 8-byte and the scale buffer 4-byte aligned", who);

  size_t lds = lds_row_bytes((size_t)KQ) + (kMode == kModeRms ? 512 * sizeof(float) + lds_row_bytes((size_t)KQ) : 0);
  int grid, gsplit;
  quant_grid(rows, KQ, &grid, &gsplit);
  auto go = [&](auto kern, LdsOptIn& opt) -> int {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), opt, (int)lds, who)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid, gsplit), dim3(kQuantThreads), lds, stream, (const uint16_t*)X, (const uint16_t*)Xup,
                       ldx ? ldx : KQ, (const uint16_t*)Wn, eps, idx, Q, SF, (int)rows, (int)KQ, (int)KE, dyn, nslots, scale_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "%s: launch failed: %s", who, hipGetErrorString(e));
    return ARCQ_OK;
  };
  static LdsOptIn lds_opt[2];           // per (kMode, kDyn, kSilu) instantiation and variant, each per device
  if (variant == ARCQ_VARIANT_G16) return go(quantize_rows_kernel<ARCQ_VARIANT_G16, kMode, kDyn, kSilu>, lds_opt[0]);
  return go(quantize_rows_kernel<ARCQ_VARIANT_G32, kMode, kDyn, kSilu>, lds_opt[1]);
}

int quantize_x(const void* X, const int16_t* idx, uint8_t* QX, uint8_t* SFX, int64_t M, int64_t KQ, int64_t KE, int variant,
               hipStream_t stream) {
  return launch_quantize<kModeX>(X, nullptr, 0.f, idx, QX, SFX, M, KQ, KE, variant, stream, "arcq_quantize_x");
}
int quantize_w(const void* W, const int16_t* idx, uint8_t* QW, uint8_t* SFW, int64_t N, int64_t KQ, int64_t KE, int variant,
               hipStream_t stream) {
  return launch_quantize<kModeW>(W, nullptr, 0.f, idx, QW, SFW, N, KQ, KE, variant, stream, "arcq_quantize_w");
}
int rmsnorm_quantize_x(const void* X, const void* Wn, float eps, const int16_t* idx, uint8_t* QX, uint8_t* SFX, int64_t M,
                       int64_t KQ, int64_t KE, int variant, hipStream_t stream) {
  if (M > 0 && (reinterpret_cast<uintptr_t>(Wn) & 15) != 0)
    return fail(ARCQ_ERR_SHAPE, "arcq_rmsnorm_quantize_x: the norm weight must be 16-byte aligned");
  return launch_quantize<kModeRms>(X, Wn, eps, idx, QX, SFX, M, KQ, KE, variant, stream, "arcq_rmsnorm_quantize_x");
}

// workgroups of the abs-max pass: ~16 chunks (256 B) per thread, at most one slot per CU
static int absmax_grid(int64_t chunks) {
  const int64_t want = (chunks + (int64_t)kAbsmaxThreads * 16 - 1) / ((int64_t)kAbsmaxThreads * 16);
  return (int)(want < 1 ? 1 : (want > kAbsmaxMaxBlocks ? kAbsmaxMaxBlocks : want));
}

// Every workgroup recomputes max|X| itself (one launch instead of two) while that costs less than the launch it
// saves.  Measured in a replayed graph (tools/decode_quant_bench.py): a dependent tiny kernel costs ~1.7 us plus its own
// latency chain; the in-kernel pass over 28 KB (4 x 3584) adds 2.2 us (5.5 against 3.3 us static), over 151 KB
// (4 x 18944) 7.5 us -- more than an abs-max launch (~3 us).  Also bounded by the total redundant L2 traffic.
constexpr int64_t kDynLocalMaxBytes = 48 * 1024;
constexpr int64_t kDynLocalMaxTotalBytes = 16 * 1024 * 1024;

int quantize_x_dyn(const void* X, const int16_t* idx, uint8_t* QX, uint8_t* SFX, float* scale_out, void* state, int64_t M,
                   int64_t KQ, int64_t KE, int variant, hipStream_t stream) {
  if (!scale_out || !state) return fail(ARCQ_ERR_NULL, "arcq_quantize_x_dyn: NULL scale_out / state");
  // validate the shape first (rows = 0 runs every check and launches nothing): a failure after the abs-max
  // launch would leave a stale maximum in `state`
  const int rc = launch_quantize<kModeX>(X, nullptr, 0.f, idx, QX, SFX, 0, KQ, KE, variant, stream, "arcq_quantize_x_dyn");
  if (rc != ARCQ_OK || M <= 0) return rc;
  if (!X || !idx || !QX || !SFX) return fail(ARCQ_ERR_NULL, "arcq_quantize_x_dyn: NULL pointer");
  if ((reinterpret_cast<uintptr_t>(X) & 15) != 0) return fail(ARCQ_ERR_SHAPE, "arcq_quantize_x_dyn: X must be 16-byte aligned");
  unsigned int* st = reinterpret_cast<unsigned int*>(state);     // ARCQ_DYN_STATE_BYTES of scratch, fully rewritten here
  const int64_t n = M * KQ, n8 = n / 8;
  int qg, qs;
  quant_grid(M, KQ, &qg, &qs);
  if (n * 2 <= kDynLocalMaxBytes && n * 2 * qg * qs <= kDynLocalMaxTotalBytes)     // decode-sized: one launch, `state` untouched
    return launch_quantize<kModeX, kDynLocal>(X, nullptr, 0.f, idx, QX, SFX, M, KQ, KE, variant, stream, "arcq_quantize_x_dyn", nullptr, 0, scale_out);
  const int grid = absmax_grid(n8);
  hipLaunchKernelGGL(absmax_bits_kernel<false>, dim3(grid), dim3(kAbsmaxThreads), 0, stream, (const uint16_t*)X, n8, n, st);
  return launch_quantize<kModeX, kDynState>(X, nullptr, 0.f, idx, QX, SFX, M, KQ, KE, variant, stream, "arcq_quantize_x_dyn", st, grid, scale_out);
}

// The abs-max words were produced elsewhere (the silu-mul GEMM epilogue): one quantiser launch.
int quantize_x_dyn_slots(const void* X, const int16_t* idx, uint8_t* QX, uint8_t* SFX, float* scale_out, const uint32_t* slots,
                         int64_t nslots, int64_t M, int64_t KQ, int64_t KE, int variant, hipStream_t stream) {
  const char* who = "arcq_quantize_x_dyn_slots";
  if (!scale_out || !slots || nslots <= 0 || nslots > INT32_MAX) return fail(ARCQ_ERR_NULL, "%s: NULL scale_out / absmax_slots, or no slots", who);
  if (!idx) {                                               // identity: X is already in reordered channel order
    if (M < 0 || KQ <= 0 || (KQ % 64) || (KE % 64) || KE < 0 || KE > KQ || KQ > 32767)
      return fail(ARCQ_ERR_SHAPE, "%s: need KQ%%64==0, KE%%64==0, 0<=KE<=KQ<=32767 (M=%lld KQ=%lld KE=%lld)", who, (long long)M, (long long)KQ, (long long)KE);
    if (variant != ARCQ_VARIANT_G16 && variant != ARCQ_VARIANT_G32) return fail(ARCQ_ERR_SHAPE, "%s: unknown variant %d", who, variant);
    if (M == 0) return ARCQ_OK;
    if (!X || !QX || !SFX) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(QX) & 7) || (reinterpret_cast<uintptr_t>(SFX) & 3))
      return fail(ARCQ_ERR_SHAPE, "%s: X must be 16-byte, QX 8-byte and SFX 4-byte aligned", who);
    const int64_t total = M * (KQ / 16);
    const int grid = (int)((total + kQuantThreads - 1) / kQuantThreads < kMaxQuantBlocks ? (total + kQuantThreads - 1) / kQuantThreads : kMaxQuantBlocks);
    if (variant == ARCQ_VARIANT_G16)
      hipLaunchKernelGGL(quantize_contig_dyn_kernel<ARCQ_VARIANT_G16>, dim3(grid), dim3(kQuantThreads), 0, stream, (const uint16_t*)X, QX, SFX, (int)M,
                         (int)KQ, (int)KE, slots, (int)nslots, scale_out);
    else
      hipLaunchKernelGGL(quantize_contig_dyn_kernel<ARCQ_VARIANT_G32>, dim3(grid), dim3(kQuantThreads), 0, stream, (const uint16_t*)X, QX, SFX, (int)M,
                         (int)KQ, (int)KE, slots, (int)nslots, scale_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "%s: launch failed: %s", who, hipGetErrorString(e));
    return ARCQ_OK;
  }
  return launch_quantize<kModeX, kDynState>(X, nullptr, 0.f, idx, QX, SFX, M, KQ, KE, variant, stream, who, slots, (int)nslots, scale_out);
}

// GU = [M, 2*KQ] bf16 (gate | up, the fused gate_up projection's output): quantise silu(gate) * up with its
// per-tensor dynamic scale.  Two launches (abs-max of the product, quantise), the product is never materialised.
// layout: ARCQ_GU_HALVES (gate in columns [0, KQ), up in [KQ, 2KQ)) or ARCQ_GU_PAIRS (g0, u0, g1, u1, ...).
int silu_mul_quantize_x_dyn(const void* GU, const int16_t* idx, uint8_t* QX, uint8_t* SFX, float* scale_out, void* state, int64_t M,
                            int64_t KQ, int64_t KE, int variant, int layout, hipStream_t stream) {
  const char* who = "arcq_silu_mul_quantize_x_dyn";
  if (layout != ARCQ_GU_HALVES && layout != ARCQ_GU_PAIRS) return fail(ARCQ_ERR_SHAPE, "%s: unknown layout %d", who, layout);
  if (!scale_out || !state) return fail(ARCQ_ERR_NULL, "%s: NULL scale_out / state", who);
  const int rc = launch_quantize<kModeX>(GU, nullptr, 0.f, idx, QX, SFX, 0, KQ, KE, variant, stream, who);
  if (rc != ARCQ_OK || M <= 0) return rc;
  if (!GU || !idx || !QX || !SFX) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
  if ((reinterpret_cast<uintptr_t>(GU) & 15) != 0) return fail(ARCQ_ERR_SHAPE, "%s: GU must be 16-byte aligned", who);
  unsigned int* st = reinterpret_cast<unsigned int*>(state);
  const uint16_t* G = reinterpret_cast<const uint16_t*>(GU);
  const uint16_t* U = G + KQ;
  const int64_t chunks = KQ / 8, total = M * chunks;
  // a chunk of the product costs eight exp: spread it as widely as the slot count allows (decode: one chunk per thread)
  const int threads = total >= (int64_t)kAbsmaxMaxBlocks * kAbsmaxThreads ? kAbsmaxThreads : 256;
  const int64_t want = (total + threads - 1) / threads;
  const int grid = (int)(want > kAbsmaxMaxBlocks ? kAbsmaxMaxBlocks : want);
  if (layout == ARCQ_GU_PAIRS) {
    hipLaunchKernelGGL(silu_mul_absmax_kernel<kSiluPairs>, dim3(grid), dim3(threads), 0, stream, G, G, 2 * KQ, (int)M, (int)chunks, st);
    return launch_quantize<kModeX, kDynState, kSiluPairs>(G, nullptr, 0.f, idx, QX, SFX, M, KQ, KE, variant, stream, who, st, grid, scale_out, G, 2 * KQ);
  }
  hipLaunchKernelGGL(silu_mul_absmax_kernel<kSiluHalves>, dim3(grid), dim3(threads), 0, stream, G, U, 2 * KQ, (int)M, (int)chunks, st);
  return launch_quantize<kModeX, kDynState, kSiluHalves>(G, nullptr, 0.f, idx, QX, SFX, M, KQ, KE, variant, stream, who, st, grid, scale_out, U, 2 * KQ);
}

// The abs-max words of silu(gate) * up were produced elsewhere (the repacked gate|up GEMM's epilogue): one launch.
int silu_mul_quantize_x_dyn_slots(const void* GU, const int16_t* idx, uint8_t* QX, uint8_t* SFX, float* scale_out, const uint32_t* slots,
                                  int64_t nslots, int64_t M, int64_t KQ, int64_t KE, int variant, int layout, hipStream_t stream) {
  const char* who = "arcq_silu_mul_quantize_x_dyn_slots";
  if (layout != ARCQ_GU_HALVES && layout != ARCQ_GU_PAIRS) return fail(ARCQ_ERR_SHAPE, "%s: unknown layout %d", who, layout);
  if (!scale_out || !slots || nslots <= 0 || nslots > INT32_MAX) return fail(ARCQ_ERR_NULL, "%s: NULL scale_out / absmax_slots, or no slots", who);
  const int rc = launch_quantize<kModeX>(GU, nullptr, 0.f, idx, QX, SFX, 0, KQ, KE, variant, stream, who);
  if (rc != ARCQ_OK || M <= 0) return rc;
  if (!GU || !idx || !QX || !SFX) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
  if ((reinterpret_cast<uintptr_t>(GU) & 15) != 0) return fail(ARCQ_ERR_SHAPE, "%s: GU must be 16-byte aligned", who);
  const uint16_t* G = reinterpret_cast<const uint16_t*>(GU);
  if (layout == ARCQ_GU_PAIRS)
    return launch_quantize<kModeX, kDynState, kSiluPairs>(G, nullptr, 0.f, idx, QX, SFX, M, KQ, KE, variant, stream, who, slots, (int)nslots, scale_out, G, 2 * KQ);
  return launch_quantize<kModeX, kDynState, kSiluHalves>(G, nullptr, 0.f, idx, QX, SFX, M, KQ, KE, variant, stream, who, slots, (int)nslots, scale_out, G + KQ, 2 * KQ);
}

int absmax_scale(const void* X, int64_t n, float* scale_out, hipStream_t stream) {
  if (n < 0) return fail(ARCQ_ERR_SHAPE, "arcq_absmax_scale: n < 0");
  if (!scale_out || (n > 0 && !X)) return fail(ARCQ_ERR_NULL, "arcq_absmax_scale: NULL pointer");
  if ((reinterpret_cast<uintptr_t>(X) & 15) != 0) return fail(ARCQ_ERR_SHAPE, "arcq_absmax_scale: X must be 16-byte aligned");
  // the fp32 output slot doubles as the integer max accumulator
  unsigned int* slot = reinterpret_cast<unsigned int*>(scale_out);
  hipError_t e = hipMemsetAsync(slot, 0, sizeof(unsigned int), stream);
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_absmax_scale: memset failed: %s", hipGetErrorString(e));
  const int64_t n8 = n / 8;
  hipLaunchKernelGGL(absmax_bits_kernel<true>, dim3(absmax_grid(n8)), dim3(kAbsmaxThreads), 0, stream, (const uint16_t*)X, n8, n, slot);
  hipLaunchKernelGGL(absmax_finish_kernel, dim3(1), dim3(1), 0, stream, slot, scale_out);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "arcq_absmax_scale: launch failed: %s", hipGetErrorString(e));
  return ARCQ_OK;
}

}  // namespace arcq
