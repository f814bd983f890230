// Decode attention of the END-TO-END HARNESS (arcquant_amd/e2e.py; SURVEY.md 8-f2: "attention stub"), NOT part of the drop-in
// boundary: the reference's attention is flashinfer over an int4 paged KV cache (kernels/src/flashinfer.cu, model/kv_cache.py),
// which stays out of scope (agemm.batch_decode_* raise NotImplementedError).  The harness keeps a dense bf16 cache and, for a
// decode step (one query token per sequence), torch's SDPA / bmm need 45-48 us per layer for the 60 MB of K and V they read
// (1.2 TB/s; tools/sdpa_decode_probe.py) -- 41 % of the measured Qwen2.5-7B full-cache decode step.  This is the streaming
// form of that one call (flash-decoding): HBM-bound, bytes = 2 * B * H * T * 128 * 2.
//
//   attn_decode_fused (default): ONE launch, grid (B * H) x 512 threads: every wave streams a contiguous range of positions with an
//       online softmax, K and V rows of a 16-position block requested together and double-buffered; the waves' records merge in LDS;
//       the new token's k / v come from the fused q|k|v projection output and are appended to the cache on the way (replaces the
//       harness's strided copy launch).  fp32 throughout.
//   attn_decode_partial + attn_decode_combine (ARCQ_HARNESS_ATTN_SLICED=1, the A-B alternative it was measured against: full-cache
//       1707 vs 1714 tok/s, current-token 2070 vs 2092): grid (B * H, S) slices with a two-pass softmax through LDS, then a merge launch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "arcq_device.hpp"
#include "arcq_internal.hpp"

namespace arcq {

constexpr int kAttnD = 128;            // head dimension (every model of benchmarks/benchmark_e2e_arc.py:26-77)
constexpr int kAttnThreads = 256;
constexpr int kAttnMaxChunk = 1024;    // positions per workgroup (LDS score buffer)

struct AttnParams {
  const uint16_t* qkv;    // bf16 [B, 3 * H * 128]: q | k | v of the current token (the fused projection's output)
  uint16_t* kcache;       // bf16 [B, H, Tmax, 128]
  uint16_t* vcache;
  float* ws;              // [B * H, S, 130]: max, sum, 128 accumulators
  uint16_t* out;          // bf16 [B, H * 128]
  int B, H, Tmax, pos, first, S, chunk;     // positions attended: [first, pos]
  float scale;
};

__global__ __launch_bounds__(kAttnThreads) void attn_decode_partial(AttnParams p) {
  __shared__ float sc[kAttnMaxChunk];
  __shared__ float red[kAttnThreads / 64][kAttnD];
  __shared__ float wred[8];
  const int bh = blockIdx.x, s = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane >> 4, c = lane & 15;                  // row within a 4-row wave load, 16-byte column chunk
  const int T = p.pos + 1;                                 // positions attended: the cache [first, pos) and the new token
  const int t0 = p.first + s * p.chunk, t1 = min(T, t0 + p.chunk);
  const size_t hidden = (size_t)p.H * kAttnD;
  const uint16_t* qrow = p.qkv + (size_t)b * 3 * hidden + (size_t)h * kAttnD;
  uint16_t* K = p.kcache + ((size_t)bh * p.Tmax) * kAttnD;
  uint16_t* V = p.vcache + ((size_t)bh * p.Tmax) * kAttnD;

  // the new token's k / v: append to the cache (one slice does it) -- the scores below read them from the projection output
  if (t0 <= p.pos && p.pos < t1 && tid < 2 * kAttnD / 8) {
    const int which = tid >> 4, cc = tid & 15;             // 0: k, 1: v
    const uint4 d = *reinterpret_cast<const uint4*>(qrow + (1 + which) * hidden + cc * 8);
    *reinterpret_cast<uint4*>((which ? V : K) + (size_t)p.pos * kAttnD + cc * 8) = d;
  }
  float q8[8];
  {
    const uint4 d = *reinterpret_cast<const uint4*>(qrow + c * 8);
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q8[2 * j] = bf16_bits_to_f32(w[j] & 0xffffu) * p.scale;
      q8[2 * j + 1] = bf16_bits_to_f32(w[j] >> 16) * p.scale;
    }
  }
  auto row_ptr = [&](const uint16_t* base, int which, int t) -> const uint16_t* {      // position t of K (which = 1) / V (2)
    return t == p.pos ? qrow + which * hidden : base + (size_t)t * kAttnD;
  };
  // ---- pass 1: scores of this slice -> LDS; four rows per wave instruction, four instructions in flight
  float mloc = -3.0e38f;
  for (int tb = t0 + wave * 16; tb < t1; tb += (kAttnThreads / 64) * 16) {
    uint4 kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = min(tb + u * 4 + r, t1 - 1);
      kv[u] = *reinterpret_cast<const uint4*>(row_ptr(K, 1, t) + c * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t w[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w};
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) d += q8[2 * j] * bf16_bits_to_f32(w[j] & 0xffffu) + q8[2 * j + 1] * bf16_bits_to_f32(w[j] >> 16);
#pragma unroll
      for (int sh = 8; sh > 0; sh >>= 1) d += __shfl_xor(d, sh, 64);
      const int t = tb + u * 4 + r;
      if (t < t1) {
        if (c == 0) sc[t - t0] = d;
        mloc = fmaxf(mloc, d);
      }
    }
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, sh, 64));
  if (lane == 0) wred[wave] = mloc;
  __syncthreads();
  const float m = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  float lsum = 0.f;
  for (int i = tid; i < t1 - t0; i += kAttnThreads) {
    const float e = __expf(sc[i] - m);
    sc[i] = e;
    lsum += e;
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) lsum += __shfl_xor(lsum, sh, 64);
  if (lane == 0) wred[4 + wave] = lsum;
  __syncthreads();
  const float l = wred[4] + wred[5] + wred[6] + wred[7];
  // ---- pass 2: acc[d] = sum_t p[t] * V[t][d]
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int tb = t0 + wave * 16; tb < t1; tb += (kAttnThreads / 64) * 16) {
    uint4 vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = min(tb + u * 4 + r, t1 - 1);
      vv[u] = *reinterpret_cast<const uint4*>(row_ptr(V, 2, t) + c * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = tb + u * 4 + r;
      const float pt = t < t1 ? sc[t - t0] : 0.f;
      const uint32_t w[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += pt * bf16_bits_to_f32(w[j] & 0xffffu);
        acc[2 * j + 1] += pt * bf16_bits_to_f32(w[j] >> 16);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {                            // the four row groups of a wave hold the same columns
    acc[j] += __shfl_xor(acc[j], 16, 64);
    acc[j] += __shfl_xor(acc[j], 32, 64);
  }
  if (r == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave][c * 8 + j] = acc[j];
  }
  __syncthreads();
  if (gridDim.y == 1) {                                    // one slice: nothing to combine, no second launch
    if (tid < kAttnD) p.out[(size_t)b * hidden + (size_t)h * kAttnD + tid] = (uint16_t)f32_to_bf16_bits((red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) / l);
    return;
  }
  float* o = p.ws + ((size_t)bh * p.S + s) * (kAttnD + 2);
  if (tid < kAttnD) o[2 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  if (tid == 0) {
    o[0] = m;
    o[1] = l;
  }
}

__global__ __launch_bounds__(kAttnD) void attn_decode_combine(AttnParams p) {
  const int bh = blockIdx.x, d = threadIdx.x;
  const float* w = p.ws + (size_t)bh * p.S * (kAttnD + 2);
  const int T = p.pos + 1 - p.first;
  const int live = min(p.S, (T + p.chunk - 1) / p.chunk);   // slices that hold positions
  // every slice's (max, sum, accumulator[d]) is requested before anything is combined: ONE global round trip for the launch
  // (a max pass followed by an accumulate pass was two); 8 slices per batch, more only beyond 2048 cached positions
  float M = -3.0e38f, L = 0.f, a = 0.f;
  for (int s0 = 0; s0 < live; s0 += 8) {
    float ms[8], ls[8], as[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = min(s0 + j, live - 1);
      ms[j] = w[s * (kAttnD + 2)];
      ls[j] = w[s * (kAttnD + 2) + 1];
      as[j] = w[s * (kAttnD + 2) + 2 + d];
    }
    float Mb = M;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (s0 + j < live) Mb = fmaxf(Mb, ms[j]);
    const float r = __expf(M - Mb);                          // rescale what earlier batches left (first batch: exp(-inf) = 0 of 0)
    L *= r;
    a *= r;
    M = Mb;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (s0 + j < live) {
        const float f = __expf(ms[j] - M);
        L += ls[j] * f;
        a += as[j] * f;
      }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  p.out[(size_t)b * p.H * kAttnD + (size_t)h * kAttnD + d] = (uint16_t)f32_to_bf16_bits(a / L);
}

// ONE launch, no scratch: a 512-thread workgroup per (batch, head); its 8 waves each stream a contiguous range of positions with an
// online softmax (running max / sum / accumulator, K and V of a 16-position block requested together, the next block's rows in
// flight while this one is reduced), then merge their 8 (max, sum, accumulator) records through LDS.  112 workgroups do not fill the
// 256 CUs, but 8 waves x 16 outstanding 1-KB loads per CU keep ~60 MB in ~12 us within reach, and the slice kernel's second launch,
// its scratch round trip and its 37 % imbalance (560 slices on 256 CUs) disappear.
constexpr int kAttnFusedWaves = 8;
__global__ __launch_bounds__(kAttnFusedWaves * 64) void attn_decode_fused(AttnParams p) {
  __shared__ float rec[kAttnFusedWaves][kAttnD + 2];
  const int bh = blockIdx.x;
  const int b = bh / p.H, h = bh - b * p.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane >> 4, c = lane & 15;                  // row within a 4-row wave load, 16-byte column chunk
  const int T = p.pos + 1;                                 // positions attended: [first, pos]
  const int per = ((T - p.first + kAttnFusedWaves - 1) / kAttnFusedWaves + 3) & ~3;      // positions per wave, whole 4-row loads
  const int t0 = p.first + wave * per, t1 = min(T, t0 + per);
  const size_t hidden = (size_t)p.H * kAttnD;
  const uint16_t* qrow = p.qkv + (size_t)b * 3 * hidden + (size_t)h * kAttnD;
  uint16_t* K = p.kcache + ((size_t)bh * p.Tmax) * kAttnD;
  uint16_t* V = p.vcache + ((size_t)bh * p.Tmax) * kAttnD;
  if (tid < 2 * kAttnD / 8) {                              // the new token's k / v: append to the cache (read below from the projection output)
    const int which = tid >> 4, cc = tid & 15;
    const uint4 d = *reinterpret_cast<const uint4*>(qrow + (1 + which) * hidden + cc * 8);
    *reinterpret_cast<uint4*>((which ? V : K) + (size_t)p.pos * kAttnD + cc * 8) = d;
  }
  float q8[8];
  {
    const uint4 d = *reinterpret_cast<const uint4*>(qrow + c * 8);
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q8[2 * j] = bf16_bits_to_f32(w[j] & 0xffffu) * p.scale;
      q8[2 * j + 1] = bf16_bits_to_f32(w[j] >> 16) * p.scale;
    }
  }
  auto row_ptr = [&](const uint16_t* base, int which, int t) -> const uint16_t* {
    return t == p.pos ? qrow + which * hidden : base + (size_t)t * kAttnD;
  };
  auto load_block = [&](int tb, uint4 (&kk)[4], uint4 (&vv)[4]) __attribute__((always_inline)) {
    const int last = max(t1 - 1, p.first);                   // (an idle wave clamps to a valid row and masks everything)
#pragma unroll
    for (int u = 0; u < 4; ++u) kk[u] = *reinterpret_cast<const uint4*>(row_ptr(K, 1, min(tb + u * 4 + r, last)) + c * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = *reinterpret_cast<const uint4*>(row_ptr(V, 2, min(tb + u * 4 + r, last)) + c * 8);
  };
  float m = -3.0e38f, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint4 k0[4], v0[4], k1[4], v1[4];
  auto consume = [&](int tb, const uint4 (&kk)[4], const uint4 (&vv)[4]) __attribute__((always_inline)) {
    float d[4], mb = -3.0e38f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t w[4] = {kk[u].x, kk[u].y, kk[u].z, kk[u].w};
      float x = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) x += q8[2 * j] * bf16_bits_to_f32(w[j] & 0xffffu) + q8[2 * j + 1] * bf16_bits_to_f32(w[j] >> 16);
#pragma unroll
      for (int sh = 8; sh > 0; sh >>= 1) x += __shfl_xor(x, sh, 64);
      d[u] = tb + u * 4 + r < t1 ? x : -3.0e38f;
      mb = fmaxf(mb, d[u]);
    }
    mb = fmaxf(mb, __shfl_xor(mb, 16, 64));                  // over the four row groups: the block's maximum, wave-uniform
    mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
    const float mn = fmaxf(m, mb), a = __expf(m - mn);
    l *= a;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= a;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float pt = tb + u * 4 + r < t1 ? __expf(d[u] - mn) : 0.f;
      l += pt;                                               // (this lane's row group; the groups are added at the end)
      const uint32_t w[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += pt * bf16_bits_to_f32(w[j] & 0xffffu);
        acc[2 * j + 1] += pt * bf16_bits_to_f32(w[j] >> 16);
      }
    }
    m = mn;
  };
  if (t0 < t1) {
    load_block(t0, k0, v0);
    for (int tb = t0; tb < t1; tb += 32) {                   // two 16-position blocks per trip: the other buffer's loads stay in flight
      if (tb + 16 < t1) load_block(tb + 16, k1, v1);         // (a third buffer measured slower: 1700 vs 1714 tok/s full-cache)
      consume(tb, k0, v0);
      if (tb + 16 < t1) {
        if (tb + 32 < t1) load_block(tb + 32, k0, v0);
        consume(tb + 16, k1, v1);
      }
    }
  }
  // the four row groups of a wave hold the same columns: add them; then the wave's record
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[j] += __shfl_xor(acc[j], 16, 64);
    acc[j] += __shfl_xor(acc[j], 32, 64);
  }
  if (r == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) rec[wave][2 + c * 8 + j] = acc[j];
    if (c == 0) {
      rec[wave][0] = m;
      rec[wave][1] = l;
    }
  }
  __syncthreads();
  if (tid < kAttnD) {                                        // merge of the waves (idle ones carry m = -3e38, l = 0)
    float M = -3.0e38f;
#pragma unroll
    for (int w = 0; w < kAttnFusedWaves; ++w) M = fmaxf(M, rec[w][0]);
    float L = 0.f, a = 0.f;
#pragma unroll
    for (int w = 0; w < kAttnFusedWaves; ++w) {
      const float f = __expf(rec[w][0] - M);
      L += rec[w][1] * f;
      a += rec[w][2 + tid] * f;
    }
    p.out[(size_t)b * hidden + (size_t)h * kAttnD + tid] = (uint16_t)f32_to_bf16_bits(a / L);
  }
}

// The model's final RMSNorm (not an ARC operator: the reference uses the stock module there) as ONE launch for the harness: torch's
// F.rms_norm is ~15 small kernels on this stack, ~40 us of a 2 ms decode step with its index arithmetic.  One workgroup per row,
// fp32 throughout: out = bf16(float(x) * rsqrt(mean(x^2) + eps) * float(w)).
__global__ __launch_bounds__(256) void harness_rmsnorm_kernel(const uint16_t* __restrict__ X, int64_t ldx, const uint16_t* __restrict__ W,
                                                                 uint16_t* __restrict__ out, int H, float eps) {
  __shared__ float wsum[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const uint16_t* x = X + (size_t)row * ldx;
  float acc = 0.f;
  for (int c = tid; c < (H >> 3); c += 256) {
    const uint4 d = *reinterpret_cast<const uint4*>(x + (size_t)c * 8);
    const uint32_t w4[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf16_bits_to_f32(w4[j] & 0xffffu), b = bf16_bits_to_f32(w4[j] >> 16);
      acc += a * a + b * b;
    }
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) acc += __shfl_xor(acc, sh, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = acc;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (float)H + eps);
  for (int c = tid; c < (H >> 3); c += 256) {
    const uint4 d = *reinterpret_cast<const uint4*>(x + (size_t)c * 8);
    const uint4 g = *reinterpret_cast<const uint4*>(W + (size_t)c * 8);
    const uint32_t w4[4] = {d.x, d.y, d.z, d.w}, g4[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf16_bits_to_f32(w4[j] & 0xffffu) * rstd * bf16_bits_to_f32(g4[j] & 0xffffu);
      const float b = bf16_bits_to_f32(w4[j] >> 16) * rstd * bf16_bits_to_f32(g4[j] >> 16);
      o[j] = f32_to_bf16_bits(a) | (f32_to_bf16_bits(b) << 16);
    }
    *reinterpret_cast<uint4*>(out + (size_t)row * H + (size_t)c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace arcq

using namespace arcq;

extern "C" int64_t arcq_harness_attn_workspace_bytes(int64_t B, int64_t H, int64_t Tmax) {
  if (B <= 0 || H <= 0 || Tmax <= 0) return 0;
  const int64_t S = (Tmax + 255) / 256;
  return B * H * S * (kAttnD + 2) * (int64_t)sizeof(float);
}

// HARNESS ONLY (see the header of this file).  qkv bf16 [B, 3 * H * 128] (q | k | v of ONE new token per sequence), caches bf16
// [B, H, Tmax, 128]; appends k / v at position `pos` and writes softmax(q k^T / sqrt(128)) v over positions [0, pos] to `out`
// (bf16 [B, H * 128]).  workspace >= arcq_harness_attn_workspace_bytes(B, H, Tmax).
extern "C" int arcq_harness_attn_decode_window(const void* qkv, void* kcache, void* vcache, void* out, void* workspace, int64_t B, int64_t H,
                                               int64_t Tmax, int64_t pos, int64_t first, void* stream);
extern "C" int arcq_harness_attn_decode(const void* qkv, void* kcache, void* vcache, void* out, void* workspace, int64_t B, int64_t H, int64_t Tmax,
                                        int64_t pos, void* stream) {
  return arcq_harness_attn_decode_window(qkv, kcache, vcache, out, workspace, B, H, Tmax, pos, 0, stream);
}

// the same over positions [first, pos] only (first == pos: what benchmarks/modeling_arc.py:169-198 attends over in a decode step)
extern "C" int arcq_harness_attn_decode_window(const void* qkv, void* kcache, void* vcache, void* out, void* workspace, int64_t B, int64_t H,
                                               int64_t Tmax, int64_t pos, int64_t first, void* stream) {
  const char* who = "arcq_harness_attn_decode";
  if (B <= 0 || H <= 0 || Tmax <= 0 || pos < 0 || pos >= Tmax || first < 0 || first > pos) return fail(ARCQ_ERR_SHAPE, "%s: bad B / H / Tmax / pos / first", who);
  if (!qkv || !kcache || !vcache || !out || !workspace) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
  if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(kcache) | reinterpret_cast<uintptr_t>(vcache)) & 15)
    return fail(ARCQ_ERR_SHAPE, "%s: qkv and the caches must be 16-byte aligned", who);
  AttnParams p;
  p.qkv = (const uint16_t*)qkv; p.kcache = (uint16_t*)kcache; p.vcache = (uint16_t*)vcache; p.ws = (float*)workspace; p.out = (uint16_t*)out;
  p.B = (int)B; p.H = (int)H; p.Tmax = (int)Tmax; p.pos = (int)pos; p.first = (int)first;
  // slices of >= 128 positions, enough of them to give every CU a workgroup, at most kAttnMaxChunk positions each; the slice
  // count is fixed per Tmax (workspace layout), the slices beyond pos stay empty
  const int T = (int)(pos + 1 - first);
  int S = (int)((Tmax + 255) / 256);
  int chunk = (T + S - 1) / S;
  chunk = (chunk + 15) & ~15;
  if (chunk > kAttnMaxChunk) return fail(ARCQ_ERR_UNSUPPORTED, "%s: more than %d positions per slice", who, kAttnMaxChunk);
  p.S = S; p.chunk = chunk;
  p.scale = 0.08838834764831845f;                           // 128^-0.5
  const int live = (T + chunk - 1) / chunk;
  static const int sliced = getenv("ARCQ_HARNESS_ATTN_SLICED") ? atoi(getenv("ARCQ_HARNESS_ATTN_SLICED")) : 0;   // A-B: the two-launch slice kernels
  if (!sliced) {
    hipLaunchKernelGGL(attn_decode_fused, dim3((unsigned)(B * H)), dim3(kAttnFusedWaves * 64), 0, (hipStream_t)stream, p);
    hipError_t e1 = hipGetLastError();
    if (e1 != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "%s: launch failed: %s", who, hipGetErrorString(e1));
    return ARCQ_OK;
  }
  hipLaunchKernelGGL(attn_decode_partial, dim3((unsigned)(B * H), (unsigned)live), dim3(kAttnThreads), 0, (hipStream_t)stream, p);
  if (live > 1) hipLaunchKernelGGL(attn_decode_combine, dim3((unsigned)(B * H)), dim3(kAttnD), 0, (hipStream_t)stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "%s: launch failed: %s", who, hipGetErrorString(e));
  return ARCQ_OK;
}

// HARNESS ONLY: out[r, :] = rmsnorm(X[r, :]) * W for `rows` rows of H bf16 values (row stride ldx elements), one launch.
extern "C" int arcq_harness_rmsnorm(const void* X, int64_t ldx, const void* W, void* out, int64_t rows, int64_t H, float eps, void* stream) {
  const char* who = "arcq_harness_rmsnorm";
  if (rows < 0 || H <= 0 || (H % 8) || ldx < H || (ldx % 8)) return fail(ARCQ_ERR_SHAPE, "%s: need H %% 8 == 0, ldx >= H, ldx %% 8 == 0", who);
  if (rows == 0) return ARCQ_OK;
  if (!X || !W || !out) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(out)) & 15)
    return fail(ARCQ_ERR_SHAPE, "%s: pointers must be 16-byte aligned", who);
  hipLaunchKernelGGL(harness_rmsnorm_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X, ldx, (const uint16_t*)W,
                     (uint16_t*)out, (int)H, eps);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "%s: launch failed: %s", who, hipGetErrorString(e));
  return ARCQ_OK;
}
