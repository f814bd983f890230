// ARC-NVFP4 linear for decode shapes (M <= 16) over a REPACKED weight: ONE persistent, stream-K kernel whose prologue
// can also BE the activation quantiser.
//
// Replaces, for decode, the reference's per-linear sequence (model/qLlamaLayer.py:73-77 + qLinearLayer.py:62-78;
// benchmarks/modeling_arc.py:211-228,279-310)
//       [rmsnorm_quantize_x | max|x| -> x/scale -> reorder_quantize_x]  ->  agemm.matmul (+ bias, + residual)
// -- two to five launches of a few microseconds each for a GEMM that streams 10-80 MB -- by one launch:
//
//   prologue (once per workgroup = once per CU; the first weight loads are already in flight)
//     kSrcPacked : packed e2m1 + ue4m3 activations (the reference layout)           -> fp16 image in LDS
//     kSrcRms    : bf16 X, norm weight, eps, reorder_index  (agemm.rmsnorm_quantize_x, rmsnorm.cu:68-255)
//     kSrcDyn    : bf16 X, reorder_index, per-tensor scale max|X|/2688 from abs-max words of the producing kernel or from
//                  X itself (NVFP4_reorder_quantize_x, qLlamaLayer.py:73-77 -> reorder.cu:68-203 / 380-555)
//                  both: the quantiser's OWN group arithmetic (quantize_device.hpp) -> codes + scale byte -> the same
//                  exact fp16 values the packed path decodes: results are BIT-IDENTICAL to the separate launches
//   K loop   (no barrier, no LDS traffic for the weights)
//     the weight is a sequence of 2 KB units (tile pair: 16 rows x 256 K, MFMA operand order; arcq.h "REPACKED").  A
//     workgroup owns consecutive row blocks = one contiguous span; its 16 waves cut the span into 16 balanced contiguous
//     ranges (stream_split.hpp), each streamed through a 4-deep register ring of fully coalesced global_load_dwordx4
//     (named registers, exactly counted vmcnt), dequantised in registers (gemm_common.hpp) and contracted on
//     v_mfma_f32_16x16x32_f16 against activation fragments read from the LDS image.  No tail: a row block is split
//     over as many waves as it takes, whatever N and K are (the one-row-block-per-wave kernel this replaces left the chip
//     half idle whenever its workgroup count was not a multiple of the resident set: 592 workgroups on 512 slots).
//   end      partial 16x16 tiles of split row blocks meet in LDS (fixed order: deterministic), then the epilogue:
//     kOutPlain      alpha (host float x optional device scalar x the prologue's dynamic scale), bias, residual, bf16 / fp32
//     kOutSiluAbsmax D as kOutPlain for interleaved gate|up rows + max |silu(g) * u| per row block (r1 entry point)
//     kOutSiluAct    ACT = bf16 silu(gate) * up [M, N/2] with torch's roundings (qLlamaLayer.py:417) + its abs-max per row
//                    block: the down projection's kSrcDyn prologue then needs no abs-max pass and no SiLU
//
// Roofline: HBM.  Bytes per launch = N*K*9/16 (+ padding of K to 256) + activations + output.
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "arcq_internal.hpp"
#include "gemm_common.hpp"
#include "quantize_device.hpp"
#include "stream_split.hpp"

namespace arcq {

// DIAGNOSTIC BUILD ONLY (make diag: -DARCQ_STREAM_STAMPS, a separate libarcq_hip_diag.so; in the product library no stamp
// executes): s_memtime at the phase boundaries of every wave, stored to a buffer of its own that nothing else reads
// (tools/stream_stamps.py prints the shares).  Stamps cost cycles and forbid overlaps: read the SHARES, not the total.
#ifdef ARCQ_STREAM_STAMPS
static unsigned long long* g_stream_stamps = nullptr;     // [workgroup][wave][16]
extern "C" void arcq_debug_set_stream_stamps(void* p) { g_stream_stamps = reinterpret_cast<unsigned long long*>(p); }
#define ARCQ_STAMP(i)                                                                                                   \
  do {                                                                                                                  \
    if (p.stamps) {                                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                                \
      unsigned long long t_;                                                                                            \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                \
      if (lane == 0) p.stamps[((size_t)blockIdx.x * kStreamWaves + wave) * 16 + (i)] = t_;                               \
    }                                                                                                                   \
  } while (0)
#define ARCQ_STAMP_RT(i)                                                                                                \
  do {                                                                                                                  \
    if (p.stamps) {                                                                                                     \
      unsigned long long t_;                                                                                            \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                    \
      if (lane == 0) p.stamps[((size_t)blockIdx.x * kStreamWaves + wave) * 16 + (i)] = t_;                              \
    }                                                                                                                   \
  } while (0)
#else
#define ARCQ_STAMP(i) do { } while (0)
#define ARCQ_STAMP_RT(i) do { } while (0)
#endif

enum : int { kSrcPacked = 0, kSrcRms = 1, kSrcDyn = 2 };
enum : int { kOutPlain = 0, kOutSiluAbsmax = 1, kOutSiluAct = 2 };

struct StreamParams {
  const uint8_t* RW;          // repacked weight tiles
  const uint8_t* RSF;         // repacked weight scales
  // activation source
  const uint8_t* A;           // kSrcPacked: [M, K/2] packed codes, reference layout
  const uint8_t* SFA;         //             swizzled ue4m3
  const uint16_t* X;          // kSrcRms / kSrcDyn: bf16 [M, KQ], row stride KQ
  const uint16_t* Wn;         // kSrcRms: bf16 [KQ]
  const int16_t* idx;         // reorder_index [KQ]
  const uint32_t* in_slots;   // kSrcDyn: abs-max words of X (bf16 magnitude bits), or NULL: computed here
  float* scale_out;           // kSrcDyn: max|X| / 2688 (written by workgroup 0), may be NULL
  float eps;
  int KQ, KE, n_in_slots;
  // output
  void* D;
  const float* alpha_dev;
  const uint16_t* bias;
  const uint16_t* residual;
  uint32_t* out_slots;        // kOutSilu*: one word per row block
  const int16_t* act_scatter; // kOutSiluAct: output column of activation j (NULL: j)
  int M, N, K;
  float alpha_host;
  int out_dtype;
  // geometry (host: stream_geometry)
  int pairs;                  // K padded to 256, in units
  int row_blocks;             // ceil(N / 16)
  int a_stride;               // bytes per token row of the fp16 image
  int img_bytes;              // image region (also: reduction-tree scratch before, partial-tile slots after the K loop)
  int stage;                  // kSrcRms / kSrcDyn: X rows are staged in LDS (1) or gathered from global memory (0)
  int xrow_bytes;             // bytes per staged row (padded, quantize_device.hpp)
  int slot_off;               // LDS byte offset of the partial-tile slots; 0: they alias the image (one more barrier)
  int stage_off, misc_off;    // LDS byte offsets of the staged rows and of the scalars
#ifdef ARCQ_STREAM_STAMPS
  unsigned long long* stamps;
#endif
};

typedef uint32_t st_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t st_u32x2 __attribute__((ext_vector_type(2)));
struct StreamRegs {           // one unit of this lane: 2 x 16 bytes of codes, 4 scale bytes
  st_u32x4 b0, b1;
  uint32_t s;
};

constexpr int kStThreads = kStreamWaves * 64;
constexpr int kStSlotBytes = kStreamWaves * 2 * 64 * 16;     // partial tiles: [wave][segment][lane] float4
constexpr int kStMiscBytes = 1024;                           // wave maxima, rstd per token, the dynamic scale

// ---- prologue, kSrcPacked: unit = 32 codes (16 bytes + two scale bytes) -> 64 bytes of fp16 --------------------------------
__device__ __forceinline__ void image_put_unit(unsigned char* a_img, const StreamParams& p, int u, int upr, int real, uint4 qv, uint32_t sf) {
  const int m = u / upr, c = u - m * upr;
  uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0, f2 = f0, f3 = f0;
  if (c < real) {
    sf >>= (c & 1) * 16;
    const f16x2 s0 = sf_pair_at(sf, 0), s1 = sf_pair_at(sf, 8);
    f0 = dequant8(qv.x, s0).u; f1 = dequant8(qv.y, s0).u; f2 = dequant8(qv.z, s1).u; f3 = dequant8(qv.w, s1).u;
  }
  uint4* dst = reinterpret_cast<uint4*>(a_img + (size_t)m * p.a_stride + c * 64);
  dst[0] = f0; dst[1] = f1; dst[2] = f2; dst[3] = f3;
}

// one quantised group -> its 16 exact fp16 values at group position `pos` of token row m
__device__ __forceinline__ void image_put_group(unsigned char* a_img, const StreamParams& p, int m, int pos, const GroupQ& g) {
  const f16x2 s2 = sf_pair(g.s8);
  uint4* dst = reinterpret_cast<uint4*>(a_img + (size_t)m * p.a_stride + (size_t)pos * 32);
  dst[0] = dequant8(g.packed.x, s2).u;
  dst[1] = dequant8(g.packed.y, s2).u;
}

// Weight units of a wave are fetched in TASKS of up to kTask units: every load of a task is issued, then the task is consumed.
// No register ring, no refill: the 16 waves of the workgroup are each other's latency hiding, every load is a real one and
// belongs to the range (a ring's clamped refills past the end of a short range cost the first version 2-3x the L1 requests
// on shapes where a wave owns 1-3 units -- most decode shapes; buffer loads dropped by the range check cost the same L1
// issue time as real ones, measured).  A range is cut into ceil(n / kTask) balanced tasks.  The loads of a task sit under
// wave-uniform branches, so hipcc waits for the whole task before its first unit (they were issued together: no loss), but
// that conservatism must not reach the PROLOGUE's own loads: those are inline-asm loads with a hand-counted wait.
constexpr int kTask = 6;

// ---- prologue loads hipcc must not wait for conservatively: issued by asm, retired by asm_wait_loads<N>() + asm_tie() -----------
__device__ __forceinline__ uint4 asm_load_b128(const void* ptr) {
  st_u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t asm_load_b32(const void* ptr) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return v;
}
// wait until at most 3 * units loads are outstanding: the asm loads above were issued BEFORE `units` weight units (3 loads each)
__device__ __forceinline__ void asm_wait_behind_units(int units) {
  switch (units) {                                           // wave-uniform
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
  }
}
__device__ __forceinline__ void asm_tie(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void asm_tie(uint32_t& v) { asm volatile("" : "+v"(v)); }

template <int kSrc, int kOut, int kVariant>
__global__ __launch_bounds__(kStThreads) void gemm_stream_kernel(StreamParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const a_img = smem;
  float* const slots = reinterpret_cast<float*>(smem + p.slot_off);                 // partial tiles; slot_off == 0: aliases the image
  unsigned char* const xstage = smem + p.stage_off;                                 // [M (+1: norm weight)][xrow_bytes]
  float* const misc = reinterpret_cast<float*>(smem + p.misc_off);
  uint32_t* const misc_u = reinterpret_cast<uint32_t*>(misc);                       // [0..15] wave maxima, [32..47] rstd per token

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, rl = lane & 15;

  // ---- this workgroup's row blocks, this wave's contiguous unit range
  int rb0, nrb;
  stream_wg_range(p.row_blocks, (int)gridDim.x, (int)blockIdx.x, &rb0, &nrb);
  const int P = p.pairs;
  const int U = nrb * P;
  int u0, n;
  stream_wave_range(U, wave, &u0, &n);
  const int rbl0 = u0 / P;                                  // row block (inside the workgroup) of the first unit
  const int pa0 = u0 - rbl0 * P;                            // first pair of the first segment inside its row block
  const int len1 = min(n, P - pa0);                         // units of the first segment; the rest starts a new row block

  float alpha = p.alpha_host * (p.alpha_dev ? *p.alpha_dev : 1.0f);
  ARCQ_STAMP(0);
  ARCQ_STAMP_RT(10);

  // ---- prologue loads.  kSrcPacked: the packed activations are requested first, the weights right behind them (inline-asm loads with
  //      a hand-counted wait: hipcc would otherwise wait for the whole first task).  kSrcRms / kSrcDyn ("activations first"): X, the norm
  //      weight, the reorder_index and the abs-max words are requested AND AWAITED before the first weight load is issued.  A CU's
  //      vector memory pipeline serves its requests in order: behind the first task of 16 waves (192 KB per CU, 49 MB over the chip)
  //      the 28 KB of activations came back after 2 400 (o_proj) ... 7 000 cycles (gate|up), and the quantiser prologue -- the
  //      kernel's critical path on every decode shape -- could not start; alone they are back at L2 / Infinity-Cache latency, and the
  //      weight stream loses those few hundred cycles at its start, once.
  const int G = p.KQ >> 4, Ptail = (p.KQ - p.KE) >> 4, Rg = G - Ptail;   // groups per row; first group with a residual twin; their number
  constexpr int kPre = 2;
  uint4 pre_q[kPre];
  uint32_t pre_s[kPre];
  uint4 pre_wn = make_uint4(0, 0, 0, 0), pre_i0 = make_uint4(0, 0, 0, 0), pre_i1 = make_uint4(0, 0, 0, 0);
  uint32_t pre_slot[4] = {0, 0, 0, 0};
  const int upr = P * 8, real = p.K >> 5, atoms_k = p.K >> 6;       // kSrcPacked: image units (32 elements) per token row: padded / real
  const int chunks = p.KQ >> 3;                                     // kSrcRms / kSrcDyn: 16-byte chunks per row
  // quantiser tasks of the fused sources: [0, M G) = (token, group) primaries, then [M G, M G + M Rg) = the residual twins of the last
  // Rg groups of every token -- tasks of their own, so that no wave quantises three groups where the others quantise one (as lanes of
  // the primary's wave they made that wave the workgroup's critical path: 2 500 cycles at the barrier behind the quantiser)
  const int ntask_q = p.M * (G + Rg);
  auto qtask = [&](int t, int& m, int& g) __attribute__((always_inline)) -> bool {
    if (t < p.M * G) { m = t / G; g = t - m * G; return false; }
    const int r = t - p.M * G;                               // only reached with Rg > 0
    m = r / Rg; g = Ptail + (r - m * Rg);
    return true;
  };
  // kSrcRms: wave w < M stages token w and forms its sum of squares in the reference's association order without leaving the wave:
  // lane l plays the reference's threads l + 64 j (rmsnorm.cu:113-131: thread v adds the 16 squares of chunks v and KQ/16 + v)
  // (64 registers of X per lane: consumed -- staged in LDS, squares summed -- BEFORE the weight ring claims its 54)
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int first_m = 0, first_g = 0;                             // this thread's first quantiser task (its reorder_index is prefetched)
  bool first_twin = false;
  if constexpr (kSrc == kSrcPacked) {
    const int units = p.M * upr;
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const int u = min(tid + j * kStThreads, units - 1);
      const int m = u / upr, c = min(u - m * upr, real - 1);
      pre_q[j] = asm_load_b128(p.A + (size_t)m * (p.K >> 1) + c * 16);
      pre_s[j] = asm_load_b32(p.SFA + sf_atom_offset(m, c >> 1, atoms_k));   // the atom's 4 bytes
    }
  } else {
    const int bdx = chunks >> 1;
    if constexpr (kSrc == kSrcRms) {
      pre_wn = *reinterpret_cast<const uint4*>(p.Wn + (size_t)min(tid, chunks - 1) * 8);
    } else {
      // kSrcDyn: thread t < M * KQ/16 owns chunks v and KQ/16 + v of row t / (KQ/16) (no order to keep: abs-max)
      const int vt = min(tid, p.M * bdx - 1), m = vt / bdx, v = vt - m * bdx;
      pre_q[0] = *reinterpret_cast<const uint4*>(p.X + ((size_t)m * chunks + v) * 8);
      pre_q[1] = *reinterpret_cast<const uint4*>(p.X + ((size_t)m * chunks + bdx + v) * 8);
      const uint32_t* sl = p.in_slots ? p.in_slots : reinterpret_cast<const uint32_t*>(p.X);
      const int ns = p.in_slots ? p.n_in_slots : 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) pre_slot[j] = sl[min(tid + j * kStThreads, ns - 1)];
    }
    {                                                        // reorder_index of this thread's first quantiser task
      first_twin = qtask(min(tid, ntask_q - 1), first_m, first_g);
      pre_i0 = *reinterpret_cast<const uint4*>(p.idx + (size_t)first_g * 16);
      pre_i1 = *reinterpret_cast<const uint4*>(p.idx + (size_t)first_g * 16 + 8);
    }
    if constexpr (kSrc == kSrcRms) {
      // ---- (a) token `wave` (M <= 16 = waves): stage the row in LDS and add the squares, thread by thread of the reference
      if (wave < p.M) {
        const uint16_t* xrow = p.X + (size_t)wave * p.KQ;
        uint16_t* row = reinterpret_cast<uint16_t*>(xstage + (size_t)wave * p.xrow_bytes);
        uint4 xa[8], xb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (64 * j < bdx) {                                // wave-uniform
            const int v = min(lane + 64 * j, bdx - 1);
            xa[j] = *reinterpret_cast<const uint4*>(xrow + (size_t)v * 8);
            xb[j] = *reinterpret_cast<const uint4*>(xrow + (size_t)(bdx + v) * 8);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int v = lane + 64 * j;
          if (v < bdx) {                                     // (implies 64 j < bdx: xa[j] / xb[j] were loaded)
            lds_store_chunk(row, v, xa[j]);
            lds_store_chunk(row, bdx + v, xb[j]);
            const uint32_t w8[8] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w, xb[j].x, xb[j].y, xb[j].z, xb[j].w};
            float acc = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float a = bf16_bits_to_f32(w8[e] & 0xffffu), b = bf16_bits_to_f32(w8[e] >> 16);
              acc = acc + a * a;
              acc = acc + b * b;
            }
            s8[j] = acc;                                     // absent partners of the tree stay 0.0f: x + 0.0f is x
          }
        }
      }
    }
    // a use of every loaded register: hipcc waits for all of them HERE (nothing else is in flight), not behind the weights below
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kSrc == kSrcRms) {
      asm_tie(pre_wn);
    } else {
      asm_tie(pre_q[0]); asm_tie(pre_q[1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) asm_tie(pre_slot[j]);
    }
    asm_tie(pre_i0); asm_tie(pre_i1);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- the first task of the weight stream
  const uint8_t* const wp = p.RW + ((size_t)rb0 * P + u0) * 2048 + lane * 16;
  const uint8_t* const sp = p.RSF + ((size_t)rb0 * P + u0) * 256 + lane * 4;
  auto load_unit = [&](StreamRegs& r, int i) __attribute__((always_inline)) {       // unit i of this wave's range
    // a weight byte is read once per launch by one CU: non-temporal loads (gemm_common.hpp)
    r.b0 = ARCQ_WLOAD(reinterpret_cast<const st_u32x4*>(wp + (size_t)i * 2048));
    r.b1 = ARCQ_WLOAD(reinterpret_cast<const st_u32x4*>(wp + (size_t)i * 2048 + 1024));
    r.s = ARCQ_WLOAD(reinterpret_cast<const uint32_t*>(sp + (size_t)i * 256));
  };
  // bias / residual of the (<= 2) row blocks this wave owns, fetched behind its LAST task's loads: they arrive with the last
  // weights instead of costing a global round trip after the final barrier (plain epilogue, N % 4 == 0: 8-byte loads)
  st_u32x2 ep_bias0 = {0, 0}, ep_bias1 = {0, 0}, ep_res0 = {0, 0}, ep_res1 = {0, 0};   // by name: a runtime-indexed array lands in scratch
  const bool ep_pre = (kOut == kOutPlain || kOut == kOutSiluAct) && (p.N & 3) == 0 && (p.bias || p.residual);
  auto prefetch_epilogue = [&]() __attribute__((always_inline)) {
    if (!ep_pre) return;
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
      const int seg_n = seg == 0 ? len1 : n - len1;
      if (seg_n <= 0 || (seg == 0 && pa0 != 0)) continue;    // as in the epilogue: not this wave's row block
      const int n0 = (rb0 + rbl0 + seg) * 16 + 4 * q;
      if (rl < p.M && n0 < p.N) {
        if (p.bias) (seg == 0 ? ep_bias0 : ep_bias1) = *reinterpret_cast<const st_u32x2*>(p.bias + n0);
        if (p.residual) (seg == 0 ? ep_res0 : ep_res1) = *reinterpret_cast<const st_u32x2*>(p.residual + (size_t)rl * p.N + n0);
      }
    }
  };
  const int ntasks = (n + kTask - 1) / kTask;
  int task_left = ntasks, done = 0;
  int c = ntasks > 0 ? (n + ntasks - 1) / ntasks : 0;       // units of the current task (wave-uniform)
  StreamRegs r0 = {}, r1 = {}, r2 = {}, r3 = {}, r4 = {}, r5 = {};
  auto load_task_part = [&](int lo, int hi) __attribute__((always_inline)) {      // units [lo, hi) of the current task
    if (c > 0 && lo <= 0 && 0 < hi) load_unit(r0, done);
    if (c > 1 && lo <= 1 && 1 < hi) load_unit(r1, done + 1);
    if (c > 2 && lo <= 2 && 2 < hi) load_unit(r2, done + 2);
    if (c > 3 && lo <= 3 && 3 < hi) load_unit(r3, done + 3);
    if (c > 4 && lo <= 4 && 4 < hi) load_unit(r4, done + 4);
    if (c > 5 && lo <= 5 && 5 < hi) load_unit(r5, done + 5);
  };
  auto load_task = [&]() __attribute__((always_inline)) { load_task_part(0, kTask); };
  // The fused sources issue their first task in three parts, between the phases of the prologue: a 16-byte wave load occupies the
  // CU's address pipeline for ~16 cycles, so 16 waves x 6 units x 3 loads held every wave in its issue sequence for 3 800 cycles
  // (gate|up) before the quantiser could start.  Two units per wave keep HBM busy through a phase (64 KB per CU ~ 6 400 cycles).
  constexpr int kPart = kSrc == kSrcPacked ? kTask : 2;
  // kSrcRms: the waves that hold a token (wave < M) are the ones the workgroup's first barrier waits for (reduction tree, rstd): they
  // issue NOTHING before it -- behind the other waves' loads their own issue sequence delayed the barrier by 2 000 - 4 000 cycles
  const bool token_wave = kSrc == kSrcRms && wave < p.M;
  if (!token_wave) load_task_part(0, kPart);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (kSrc == kSrcPacked) {
    // the prologue's own loads are older than the c units just issued: retire exactly them
    asm_wait_behind_units(c);
#pragma unroll
    for (int j = 0; j < kPre; ++j) { asm_tie(pre_q[j]); asm_tie(pre_s[j]); }
    __builtin_amdgcn_sched_barrier(0);
  }
  ARCQ_STAMP(1);
  if (ntasks <= 1) prefetch_epilogue();                       // (after the counted wait above: it assumes 3 loads per unit behind it)

  // ---- activation image --------------------------------------------------------------------------------------------------
  if constexpr (kSrc == kSrcPacked) {
    const int units = p.M * upr;
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const int u = tid + j * kStThreads;
      if (u < units) image_put_unit(a_img, p, u, upr, real, pre_q[j], pre_s[j]);
    }
    for (int u = tid + kPre * kStThreads; u < units; u += kStThreads) {       // beyond the prefetch (M * K > 64 K elements)
      const int m = u / upr, cc = min(u - m * upr, real - 1);
      image_put_unit(a_img, p, u, upr, real, *reinterpret_cast<const uint4*>(p.A + (size_t)m * (p.K >> 1) + cc * 16),
                     *reinterpret_cast<const uint32_t*>(p.SFA + sf_atom_offset(m, cc >> 1, atoms_k)));
    }
  } else {
    // ---- (a) stage X (and the norm weight) in LDS; on the way: abs-max of X (kSrcDyn without abs-max words) or, kSrcRms, the sums
    //      of squares in the reference's association order (thread v adds the 16 squares of chunks v and bdx + v sequentially,
    //      rmsnorm.cu:113-131; oracle rms_sumsq), formed from the very registers that loaded X
    uint32_t amax = 0;
    const int bdx = chunks >> 1;                             // = G: the reference's block size
    if constexpr (kSrc == kSrcRms) {
      if (wave < p.M) {
        // ---- (b) the reference's fixed tree s[v] += s[v + stride], stride = 256 ... 1 (rmsnorm.cu:133-154), inside the wave: lane l
        //      holds s[l + 64 j]; strides 256 / 128 / 64 combine its own registers, 32 and 16 are shuffles, 8 .. 1 DPP row shifts
        //      (the same scheme as quantize.hip's rms_sumsq_tree, byte-checked against the oracle)
#pragma unroll
        for (int j = 0; j < 4; ++j) s8[j] = s8[j] + s8[j + 4];                          // stride 256
        s8[0] = s8[0] + s8[2];                                                          // stride 128
        s8[1] = s8[1] + s8[3];
        float z = s8[0] + s8[1];                                                        // stride 64
        const float up = __shfl_down(z, 32, 64);
        if (lane < 32) z = z + up;                                                      // stride 32
        float val = lane < 32 ? z : 0.0f;
        val += __shfl_down(val, 16, 64);                                                // lane 0's cone = the reference's
        val += dpp_row_shl<8>(val);                                                     // strides 8 .. 1 stay inside lane 0's row of 16:
        val += dpp_row_shl<4>(val);                                                     // DPP (lane i <- lane i + n, 0 beyond the row)
        val += dpp_row_shl<2>(val);
        val += dpp_row_shl<1>(val);
        if (lane == 0) {
          const float var = val / (float)p.KQ + p.eps;                                  // rmsnorm.cu:157
          misc[32 + wave] = (float)(1.0 / sqrt((double)var));                           // oracle assumption A4
        }
      }
      if (tid < chunks) lds_store_chunk(reinterpret_cast<uint16_t*>(xstage + (size_t)p.M * p.xrow_bytes), tid, pre_wn);
      for (int cidx = tid + kStThreads; cidx < chunks; cidx += kStThreads)              // KQ > 8192 never reaches the RMSNorm path
        lds_store_chunk(reinterpret_cast<uint16_t*>(xstage + (size_t)p.M * p.xrow_bytes), cidx, *reinterpret_cast<const uint4*>(p.Wn + (size_t)cidx * 8));
      ARCQ_STAMP(8);
      __syncthreads();                                       // staged rows, norm weight and rstd visible
    } else {
      const int vthreads = p.M * bdx;
      const bool need = p.stage || !p.in_slots;
      if (need) {
        for (int t = tid; t < vthreads; t += kStThreads) {
          const int m = t / bdx, v = t - m * bdx;
          uint4 d0 = pre_q[0], d1 = pre_q[1];
          if (t != tid) {                                      // beyond the prefetch (M * KQ > 16 K elements)
            d0 = *reinterpret_cast<const uint4*>(p.X + ((size_t)m * chunks + v) * 8);
            d1 = *reinterpret_cast<const uint4*>(p.X + ((size_t)m * chunks + bdx + v) * 8);
          }
          if (p.stage) {
            uint16_t* row = reinterpret_cast<uint16_t*>(xstage + (size_t)m * p.xrow_bytes);
            lds_store_chunk(row, v, d0);
            lds_store_chunk(row, bdx + v, d1);
          }
          amax = absmax_bits_chunk(d1, absmax_bits_chunk(d0, amax));
        }
      }
    }
    float dyn_scale = 1.0f;
    if constexpr (kSrc == kSrcDyn) {
      // ---- (b) per-tensor scale = max|X| / 2688 (qLlamaLayer.py:74), exactly as quantize.hip computes it
      if (p.in_slots) {
        amax = max(max(pre_slot[0], pre_slot[1]), max(pre_slot[2], pre_slot[3]));
        for (int i = tid + 4 * kStThreads; i < p.n_in_slots; i += kStThreads) amax = max(amax, p.in_slots[i]);
      }
      amax = wave_max_u32(amax);
      if (lane == 0) misc_u[wave] = amax;
      ARCQ_STAMP(8);
      __syncthreads();                                       // (also publishes the staged rows)
      uint32_t mbits = 0;
#pragma unroll
      for (int w = 0; w < kStreamWaves; ++w) mbits = max(mbits, misc_u[w]);
      dyn_scale = bf16_bits_to_f32(mbits) * (1.0f / (448.0f * 6.0f));
      if (blockIdx.x == 0 && tid == 0 && p.scale_out) p.scale_out[0] = dyn_scale;
      alpha *= dyn_scale;                                    // the caller's scale_x * scale_w (qLinearLayer.py:69)
      dyn_scale = round_to_bf16(dyn_scale);                  // torch divides a bf16 tensor by the scale rounded to bf16
    }
    const DynDiv dyn_div(dyn_scale, kSrc == kSrcDyn);
    load_task_part(token_wave ? 0 : kPart, 2 * kPart);
    __builtin_amdgcn_sched_barrier(0);
    ARCQ_STAMP(9);
    // ---- (c) quantise task by task straight into the image; the K padding is zero
    const uint16_t* wn_lds = reinterpret_cast<const uint16_t*>(xstage + (size_t)p.M * p.xrow_bytes);
    // `p.stage` and the divider's fast path are uniform over the launch: the task loop is instantiated per combination (as
    // branches inside the 16-element gather they cost a scalar branch and a FULL wait per element -- sixteen serialised LDS
    // round trips per group)
    const st_u32x4 pi0 = {pre_i0.x, pre_i0.y, pre_i0.z, pre_i0.w}, pi1 = {pre_i1.x, pre_i1.y, pre_i1.z, pre_i1.w};
    auto run_groups = [&](auto stage_tag, auto fast_tag) __attribute__((always_inline)) {
    constexpr bool kStage = decltype(stage_tag)::value, kFast = decltype(fast_tag)::value;
    for (int t = tid; t < ntask_q; t += kStThreads) {
      int m = first_m, g = first_g;
      bool twin = first_twin;                                // the residual twin of (m, g): uniform over a wave except at one boundary
      st_u32x4 i0 = pi0, i1 = pi1;                           // (native vectors: a HIP uint4 captured by the lambda lands in scratch)
      if (t != tid) {                                        // later tasks of this thread (M * KQ > 16 K elements)
        twin = qtask(t, m, g);
        i0 = *reinterpret_cast<const st_u32x4*>(p.idx + (size_t)g * 16);
        i1 = *reinterpret_cast<const st_u32x4*>(p.idx + (size_t)g * 16 + 8);
      }
      const uint32_t iw[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
      const uint16_t* xrow_lds = reinterpret_cast<const uint16_t*>(xstage + (size_t)m * p.xrow_bytes);
      const uint16_t* xrow_g = p.X + (size_t)m * p.KQ;
      const float rstd = kSrc == kSrcRms ? misc[32 + m] : 1.0f;
      float v[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t ia = iw[j] & 0xffffu, ib = iw[j] >> 16;
        const uint32_t pw = lds_pad_pair(iw[j]), pa = pw & 0xffffu, pb = pw >> 16;
        float a, b;
        if (kStage) {
          a = bf16_bits_to_f32(xrow_lds[pa]);
          b = bf16_bits_to_f32(xrow_lds[pb]);
        } else {
          a = bf16_bits_to_f32(xrow_g[ia]);
          b = bf16_bits_to_f32(xrow_g[ib]);
        }
        if (kSrc == kSrcDyn) {                                 // torch: bf16(float(x) / scale)
          a = round_to_bf16(dyn_div.template div<kFast>(a));
          b = round_to_bf16(dyn_div.template div<kFast>(b));
        }
        if (kSrc == kSrcRms) {                                 // rmsnorm.cu:165-171
          a = round_to_bf16(a * bf16_bits_to_f32(wn_lds[pa]) * rstd);
          b = round_to_bf16(b * bf16_bits_to_f32(wn_lds[pb]) * rstd);
        }
        v[2 * j] = a;
        v[2 * j + 1] = b;
      }
      int pos;                                                 // augmented-K position (reorder.cu:139 / :451-452)
      if (kVariant == ARCQ_VARIANT_G16) {
        pos = g + (g > Ptail ? g - Ptail : 0);
      } else {
        const int g1 = g & ~1;
        pos = g1 + (g1 > Ptail ? g1 - Ptail : 0) + (g & 1);
      }
      if (!twin) {                                             // the primary's codes do not depend on kResid (quantize_group)
        image_put_group(a_img, p, m, pos, quantize_group<false, kVariant>(v));
      } else {                                                 // residual channels: reorder.cu:166-198, 499-550
        quantize_group<true, kVariant>(v);                     // v <- bf16(v - q * S)
        image_put_group(a_img, p, m, pos + (kVariant == ARCQ_VARIANT_G16 ? 1 : 2), quantize_group<false, kVariant>(v));
      }
    }
    };
    if constexpr (kSrc == kSrcRms) {
      run_groups(std::true_type{}, std::true_type{});          // always staged (stream_geometry), no division
    } else if (p.stage) {
      if (dyn_div.fast) run_groups(std::true_type{}, std::true_type{});
      else run_groups(std::true_type{}, std::false_type{});
    } else {
      if (dyn_div.fast) run_groups(std::false_type{}, std::true_type{});
      else run_groups(std::false_type{}, std::false_type{});
    }
    load_task_part(2 * kPart, kTask);
    __builtin_amdgcn_sched_barrier(0);
    const int pad_groups = P * 16 - (p.K >> 4);               // zero scale bytes of the repacked weight meet zeros here
    for (int t = tid; t < p.M * pad_groups; t += kStThreads) {
      const int m = t / pad_groups, g = (p.K >> 4) + (t - m * pad_groups);
      uint4* dst = reinterpret_cast<uint4*>(a_img + (size_t)m * p.a_stride + (size_t)g * 32);
      dst[0] = make_uint4(0, 0, 0, 0);
      dst[1] = make_uint4(0, 0, 0, 0);
    }
  }
  ARCQ_STAMP(2);
  __syncthreads();
  ARCQ_STAMP(3);

  // ---- K loop: no barrier, no LDS traffic for the weights ------------------------------------------------------------------
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const unsigned char* const a_tok = a_img + (size_t)min(rl, p.M - 1) * p.a_stride + q * 64;   // tokens >= M: any row (never stored)
  int pr = pa0;
  auto tile = [&](st_u32x4 b, uint32_t s16, const unsigned char* ap) __attribute__((always_inline)) {
    Frag8 a0, a1, a2, a3;
    a0.u = *reinterpret_cast<const uint4*>(ap);
    a1.u = *reinterpret_cast<const uint4*>(ap + 16);
    a2.u = *reinterpret_cast<const uint4*>(ap + 32);
    a3.u = *reinterpret_cast<const uint4*>(ap + 48);
    const f16x2 s0 = sf_pair_at(s16, 0), s1 = sf_pair_at(s16, 8);
    const Frag8 b0 = dequant8(b.x, s0), b1 = dequant8(b.y, s0), b2 = dequant8(b.z, s1), b3 = dequant8(b.w, s1);
    // weights are the MFMA A operand (rows = weight rows), activations the B operand (columns = tokens)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0.v, a0.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1.v, a1.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2.v, a2.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b3.v, a3.v, acc, 0, 0, 0);
  };
  auto step = [&](const StreamRegs& r) __attribute__((always_inline)) {
    const unsigned char* ap = a_tok + (size_t)pr * 512;
    tile(r.b0, r.s, ap);
    tile(r.b1, r.s >> 16, ap + 256);
    ++done;
    ++pr;
    if (done == len1) {                                     // wave-uniform: the first segment ends (row block boundary or range end)
      acc1 = acc;
      acc = f32x4{0.f, 0.f, 0.f, 0.f};
      pr = 0;
    }
  };
#pragma unroll 1
  while (task_left > 0) {
    const int cc = c;
    if (cc > 0) step(r0);
    if (cc > 1) step(r1);
    if (cc > 2) step(r2);
    if (cc > 3) step(r3);
    if (cc > 4) step(r4);
    if (cc > 5) step(r5);
    if (--task_left > 0) {
      c = (n - done + task_left - 1) / task_left;
      load_task();
      if (task_left == 1) prefetch_epilogue();
    }
  }
  ARCQ_STAMP(4);

  // ---- partial tiles meet in LDS.  Segment 1 = units [u0, u0 + len1) of row block rbl0, starting at pair pa0; segment 2 = the
  //      rest, from pair 0 of the next row block.  A lane holds C[token rl][row 16 rb + 4 q + e].
  if (p.slot_off == 0) __syncthreads();                     // the slots alias the image: every wave must be done reading it
  {
    float* mine = slots + ((wave * 2) * 64 + lane) * 4;
    *reinterpret_cast<float4*>(mine) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
    *reinterpret_cast<float4*>(mine + 64 * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  __syncthreads();
  ARCQ_STAMP(5);
  // the wave whose segment STARTS a row block owns it: it adds the head segments of the following waves (they continue the
  // row block) in wave order, then finishes the tile
#pragma unroll 1
  for (int seg = 0; seg < 2; ++seg) {
    const int seg_n = seg == 0 ? len1 : n - len1;
    if (seg_n <= 0 || (seg == 0 && pa0 != 0)) continue;     // nothing, or a head segment (someone else's row block)
    const int rbl = seg == 0 ? rbl0 : rbl0 + 1;             // row block inside the workgroup
    float4 sum = *reinterpret_cast<const float4*>(slots + ((wave * 2 + seg) * 64 + lane) * 4);
    const int rb_end = (rbl + 1) * P;
    for (int w2 = wave + 1; w2 < kStreamWaves; ++w2) {
      const int s2 = stream_wave_start(U, w2), e2 = stream_wave_start(U, w2 + 1);
      if (s2 >= rb_end) break;
      if (e2 > s2) {                                        // its first segment continues this row block (s2 > rbl * P)
        const float4 v = *reinterpret_cast<const float4*>(slots + ((w2 * 2) * 64 + lane) * 4);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
    }
    const int rb = rb0 + rbl;
    const int n0 = rb * 16 + 4 * q;
    const bool live = rl < p.M && n0 < p.N;
    const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
    if constexpr (kOut == kOutSiluAct) {                    // rows interleave gate and up: (g, u, g, u) -> two activations
      uint32_t mx = 0;
      if (live) {
        uint32_t y[4];                                      // bf16 (gate, up, gate, up) as the separate GEMM (+ bias) would leave them
        const st_u32x2 eb = seg == 0 ? ep_bias0 : ep_bias1;   // this tile's four bias values, prefetched (N % 4 == 0 here)
        const uint32_t bw[2] = {eb.x, eb.y};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = f32_to_bf16_bits(alpha * sv[e]);
          if (p.bias) y[e] = f32_to_bf16_bits(bf16_bits_to_f32(y[e]) + bf16_bits_to_f32((bw[e >> 1] >> (16 * (e & 1))) & 0xffffu));
        }
        const uint32_t a0 = silu_mul_bf16(y[0], y[1]);
        const uint32_t a1 = silu_mul_bf16(y[2], y[3]);
        uint16_t* arow = reinterpret_cast<uint16_t*>(p.D) + (size_t)rl * (p.N >> 1);
        if (p.act_scatter) {                                // the consumer's reorder_index applied here: it then reads contiguous groups
          const uint32_t jj = *reinterpret_cast<const uint32_t*>(p.act_scatter + (n0 >> 1));
          arow[jj & 0xffffu] = (uint16_t)a0;
          arow[jj >> 16] = (uint16_t)a1;
        } else {
          *reinterpret_cast<uint32_t*>(arow + (n0 >> 1)) = a0 | (a1 << 16);
        }
        mx = max(a0 & 0x7fffu, a1 & 0x7fffu);
      }
      mx = wave_max_u32(mx);
      if (lane == 0) p.out_slots[rb] = mx;
    } else {
      if (live) {
        if (ep_pre) {
          const st_u32x2 eb = seg == 0 ? ep_bias0 : ep_bias1, er = seg == 0 ? ep_res0 : ep_res1;
          finish4_pre<uint32_t>(p, alpha, rl, n0, sv, make_uint2(eb.x, eb.y), make_uint2(er.x, er.y));
        }
        else finish4<uint32_t>(p, alpha, rl, n0, sv);
      }
      if constexpr (kOut == kOutSiluAbsmax) {               // N % 4 == 0, bf16 out, no bias / residual (checked by the launcher)
        uint32_t mx = 0;
        if (live) {
          const uint32_t b0 = f32_to_bf16_bits(alpha * sv[0]), b1 = f32_to_bf16_bits(alpha * sv[1]);
          const uint32_t b2 = f32_to_bf16_bits(alpha * sv[2]), b3 = f32_to_bf16_bits(alpha * sv[3]);
          mx = max(silu_mul_bf16(b0, b1) & 0x7fffu, silu_mul_bf16(b2, b3) & 0x7fffu);
        }
        mx = wave_max_u32(mx);
        if (lane == 0) p.out_slots[rb] = mx;
      }
    }
  }
  ARCQ_STAMP(6);
  ARCQ_STAMP_RT(11);
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static int64_t stream_pairs(int64_t K) { return (K + 255) / 256; }

struct StreamGeom {
  int pairs, row_blocks, grid, a_stride, img_bytes, stage, xrow_bytes, slot_off, stage_off, misc_off, lds;
};

// 1 = the shape fits this kernel (M <= 16, the fp16 image and -- for kSrcRms -- the staged rows fit the 160 KB of LDS)
static int stream_geometry(int src, int64_t M, int64_t N, int64_t K, int64_t KQ, StreamGeom* g) {
  if (M < 1 || M > 16 || N < 1 || K < 64 || (K % 64)) return 0;
  const int64_t cap = 160 * 1024;
  g->pairs = (int)stream_pairs(K);
  g->row_blocks = (int)((N + 15) / 16);
  g->grid = stream_grid(g->row_blocks);
  g->a_stride = g->pairs * 512 + 16;                        // + 16: token rows start in different banks
  int64_t img = M * (int64_t)g->a_stride;
  img = (img + 15) & ~(int64_t)15;
  g->xrow_bytes = src == kSrcPacked ? 0 : (int)((lds_row_bytes((size_t)KQ) + 15) & ~(size_t)15);
  const int64_t rows = src == kSrcPacked ? 0 : M + (src == kSrcRms ? 1 : 0);
  const int64_t stage_bytes = rows * g->xrow_bytes;
  // preference order: separate partial-tile slots (one barrier less at the end) and staged rows; give up the separate slots
  // first, then the staging (kSrcDyn only: it can gather from global memory)
  for (int attempt = 0; attempt < 3; ++attempt) {
    const bool sep = attempt == 0, stage = attempt < 2 && src != kSrcPacked;
    if (attempt == 2 && src == kSrcRms) return 0;           // the reference-order sum of squares reads the staged rows
    int64_t image = img;
    if (!sep && image < kStSlotBytes) image = kStSlotBytes;
    const int64_t total = image + (sep ? kStSlotBytes : 0) + (stage ? stage_bytes : 0) + kStMiscBytes;
    if (total > cap) continue;
    g->img_bytes = (int)image;
    g->slot_off = sep ? (int)image : 0;
    g->stage = stage ? 1 : 0;
    g->stage_off = (int)(image + (sep ? kStSlotBytes : 0));
    g->misc_off = g->stage_off + (int)(stage ? stage_bytes : 0);
    g->lds = (int)total;
    return 1;
  }
  return 0;
}

int gemm_fused_supported(int kind, int64_t M, int64_t N, int64_t KQ, int64_t KE) {
  StreamGeom g;
  if (kind != kSrcRms && kind != kSrcDyn) return 0;
  if (KQ <= 0 || (KQ % 64) || (KE % 64) || KE < 0 || KE > KQ || KQ > 32767) return 0;
  if (kind == kSrcRms && (KQ < 2048 || KQ > 8192)) return 0;
  return stream_geometry(kind, M, N, KQ + KE, KQ, &g);
}

template <int kSrc, int kOut, int kVariant>
static int launch_stream(const StreamParams& p, const StreamGeom& g, hipStream_t stream, const char* who) {
  static LdsOptIn lds_opt;               // per kernel instantiation, per device
  auto kern = gemm_stream_kernel<kSrc, kOut, kVariant>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_opt, g.lds, who)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)g.grid), dim3(kStThreads), g.lds, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "%s: launch failed: %s", who, hipGetErrorString(e));
  return ARCQ_OK;
}

template <int kSrc, int kOut>
static int launch_stream_variant(const StreamParams& p, const StreamGeom& g, int variant, hipStream_t stream, const char* who) {
  if (kSrc == kSrcPacked || variant == ARCQ_VARIANT_G16) return launch_stream<kSrc, kOut, ARCQ_VARIANT_G16>(p, g, stream, who);
  return launch_stream<kSrc, kOut, ARCQ_VARIANT_G32>(p, g, stream, who);
}

static void fill_common(StreamParams& p, const StreamGeom& g, const GemmArgs& a, const uint8_t* RW, const uint8_t* RSF) {
  p = StreamParams{};
  p.RW = RW; p.RSF = RSF; p.A = a.A; p.SFA = a.SFA; p.D = a.D;
  p.alpha_dev = a.alpha_dev; p.bias = a.bias; p.residual = a.residual; p.out_slots = a.absmax_slots;
  p.M = a.M; p.N = a.N; p.K = a.K; p.alpha_host = a.alpha_host; p.out_dtype = a.out_dtype;
  p.pairs = g.pairs; p.row_blocks = g.row_blocks; p.a_stride = g.a_stride; p.img_bytes = g.img_bytes;
  p.stage = g.stage; p.xrow_bytes = g.xrow_bytes; p.slot_off = g.slot_off; p.stage_off = g.stage_off; p.misc_off = g.misc_off;
#ifdef ARCQ_STREAM_STAMPS
  p.stamps = g_stream_stamps;
#endif
}

// The packed-activation path through THIS kernel (ARCQ_REPACKED_STREAM=1, tuning / A-B only: gemm_rowblock.hip, two 8-wave
// workgroups per CU with a three-deep ring, measured faster on every plain shape and stays the default there)
int gemm_repacked_stream(const GemmArgs& a, const uint8_t* RW, const uint8_t* RSF, hipStream_t stream) {
  const bool silu = a.epilogue == kEpiSiluMul;              // here: D stays gate|up, absmax_slots gets max |silu(g) * u| per row block
  if (silu && (!a.absmax_slots || (a.N % 4) || a.bias || a.residual || a.out_dtype != ARCQ_OUT_BF16))
    return fail(ARCQ_ERR_SHAPE, "arcq_gemm_nvfp4_repacked_silu_absmax: needs absmax_slots, N %% 4 == 0, bf16 output, no bias / residual");
  StreamGeom g;
  if (!stream_geometry(kSrcPacked, a.M, a.N, a.K, 0, &g))
    return fail(ARCQ_ERR_UNSUPPORTED, "arcq_gemm_nvfp4_repacked: M=%d K=%d outside the repacked path (M <= 16, activation image <= 160 KB)", a.M, a.K);
  StreamParams p;
  fill_common(p, g, a, RW, RSF);
  if (silu) return launch_stream<kSrcPacked, kOutSiluAbsmax, ARCQ_VARIANT_G16>(p, g, stream, "arcq_gemm_nvfp4_repacked_silu_absmax");
  return launch_stream<kSrcPacked, kOutPlain, ARCQ_VARIANT_G16>(p, g, stream, "arcq_gemm_nvfp4_repacked");
}

// arcq_linear_*: the activation quantiser runs as the GEMM's prologue
int gemm_fused(const FusedArgs& f, hipStream_t stream) {
  const char* who = f.kind == kSrcRms ? (f.silu_act ? "arcq_linear_rmsnorm_silu_repacked" : "arcq_linear_rmsnorm_repacked") : "arcq_linear_dynamic_repacked";
  StreamGeom g;
  const int64_t K = (int64_t)f.KQ + f.KE;
  if (f.kind != kSrcRms && f.kind != kSrcDyn) return fail(ARCQ_ERR_SHAPE, "%s: unknown source kind %d", who, f.kind);
  if (!gemm_fused_supported(f.kind, f.M, f.N, f.KQ, f.KE) || !stream_geometry(f.kind, f.M, f.N, K, f.KQ, &g))
    return fail(ARCQ_ERR_UNSUPPORTED, "%s: M=%d N=%d KQ=%d KE=%d outside the fused decode path (see arcq_linear_fused_supported)", who, f.M, f.N, f.KQ, f.KE);
  GemmArgs a{};
  a.D = f.D; a.M = f.M; a.N = f.N; a.K = (int)K; a.alpha_host = f.alpha_host; a.alpha_dev = f.alpha_dev;
  a.bias = f.bias; a.residual = f.residual; a.out_dtype = f.out_dtype; a.absmax_slots = f.out_slots;
  StreamParams p;
  fill_common(p, g, a, f.RW, f.RSF);
  p.X = f.X; p.Wn = f.Wn; p.idx = f.idx; p.in_slots = f.in_slots; p.n_in_slots = f.n_in_slots; p.scale_out = f.scale_out;
  p.eps = f.eps; p.KQ = f.KQ; p.KE = f.KE; p.act_scatter = f.act_scatter;
  if (f.kind == kSrcRms) {
    if (f.silu_act) return launch_stream_variant<kSrcRms, kOutSiluAct>(p, g, f.variant, stream, who);
    return launch_stream_variant<kSrcRms, kOutPlain>(p, g, f.variant, stream, who);
  }
  return launch_stream_variant<kSrcDyn, kOutPlain>(p, g, f.variant, stream, who);
}

}  // namespace arcq
