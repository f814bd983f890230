/*
 * arcq_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, never shipped, never on the product path)
 * of ARCQuant's NVFP4 + Augmented-Residual-Channel hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY STATUS
 *   The reference implementation of this path is CUDA (sm_120a) + CUTLASS (un-vendored submodule):
 *   it cannot be compiled or run here, and the reference holds no golden vectors for it.
 *   => the byte-level quantiser oracle below is a restatement of the kernel TEXT and is
 *      "parity unpinned" against the CUDA binary.  It is pinned as far as possible by
 *        (1) the e2m1 / e4m3 tables cross-checked against torch's float8_e4m3fn CPU conversion,
 *        (2) the reference's importable Python fake-quant path (kernels/fake.py,
 *            model/quantize.py), whose outputs are committed under tests/golden/: with the
 *            "fake semantics" switches below (ARCQ_SEM_*) this SAME group/row code reproduces those
 *            outputs BIT FOR BIT (block partition, amax, /6, e2m1 grid, residual-channel selection,
 *            dequantisation are thereby pinned to the reference's own outputs), and every element on
 *            which the kernel-text semantics (flags = 0) differ is attributed by tests/test_oracle_pin.py
 *            to exactly the switched rule that causes it: tie rule, x/s vs x*(1/s), scale grid, bf16
 *            rounding of the residual.  What stays unpinned is only what the fake path does not have:
 *            those four kernel-text rules themselves (A1-A3) and the rsqrt of the RMSNorm variant (A4).
 *
 * Every function cites the reference file:line it follows (paths under /root/reference).
 *
 * Numeric assumptions (documented in DESIGN.md "oracle assumptions"):
 *   A1  CUTLASS NumericConverter<..., round_to_nearest> == IEEE RNE, saturating to the finite
 *       maximum, sign of zero preserved (cvt.rn.satfinite semantics).
 *   A2  nvcc's default -fmad=true contracts `x - q*S` into one fused multiply-add.  For the
 *       16-per-thread kernels q*S is exact (2 x 4 significant bits) so this is immaterial; for
 *       the 32-per-thread kernels S is the UN-rounded fp32 scale and the fused form is used.
 *   A3  float division (`maxv / 6`, `sum / KQ`) is IEEE correctly rounded (nvcc -prec-div=true).
 *   A4  rsqrt() is taken as the correctly rounded (float)(1/sqrt((double)v)); the CUDA intrinsic is a
 *       2-ulp approximation, so rmsnorm parity with the CUDA binary can never be bit-exact.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ARCQ_VARIANT_G16 0 /* reorder.cu:68-330, rmsnorm.cu:68-255 : one 16-group per thread   */
#define ARCQ_VARIANT_G32 1 /* reorder.cu:380-696, down.cu:71-361  : two 16-groups per thread   */

#define FP4_MAX 6.0f            /* reorder.cu:17 */
#define FP8_MAX 448.0f          /* reorder.cu:18 */
#define SCALE_EPS 0.001953125f  /* reorder.cu:19  (2^-9, the smallest e4m3 subnormal) */

/* ------------------------------------------------------------------------------------------ */
/* scalar formats                                                                              */
/* ------------------------------------------------------------------------------------------ */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* bf16 <-> f32.  Float2Bfloat16 = NumericConverter<bfloat16_t,float,RNE> (reorder.cu:103). */
float arcq_o_bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }

uint16_t arcq_o_f32_to_bf16(float f) {
  uint32_t u = f2u(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* quiet NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

/* e2m1: 1 sign, 2 exponent, 1 mantissa bit; magnitudes {0,.5,1,1.5,2,3,4,6}.
 * Float2E2m1 (reorder.cu:98) RNE, ties to the even code, saturating; sign kept (A1). */
static const float E2M1_MAG[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};

uint8_t arcq_o_e2m1_encode(float x) {
  uint8_t sign = (uint8_t)((f2u(x) >> 31) << 3);
  float a = fabsf(x);
  uint8_t c;
  if (a != a) c = 7;               /* NaN: satfinite -> max magnitude */
  else c = (uint8_t)((a > 0.25f) + (a >= 0.75f) + (a > 1.25f) + (a >= 1.75f) + (a > 2.5f) +
                     (a >= 3.5f) + (a > 5.0f));
  return (uint8_t)(sign | c);
}

float arcq_o_e2m1_decode(uint8_t code) { /* E2m12Float, reorder.cu:99 */
  float m = E2M1_MAG[code & 7];
  return (code & 8) ? -m : m;
}

/* ue4m3: e4m3 magnitude (bias 7, subnormals k*2^-9, max 448), sign bit always 0.
 * Float2Ue4m3 (reorder.cu:100).  Input is always pre-clamped to [2^-9, 448] (reorder.cu:138). */
uint8_t arcq_o_ue4m3_encode(float s) {
  if (!(s > 0.0f)) return 0;
  if (s >= FP8_MAX) return 0x7e;
  if (s < 0.015625f) {                   /* below 2^-6: subnormal grid, step 2^-9, RNE */
    float k = nearbyintf(s * 512.0f);    /* exact product; default rounding mode = RNE */
    return (uint8_t)k;                   /* k == 8 is the encoding of 2^-6 itself */
  }
  uint32_t u = f2u(s);
  u += 0x7ffffu + ((u >> 20) & 1u);      /* RNE to 3 mantissa bits */
  u >>= 20;                              /* (exp_f32 << 3) | m3 */
  int code = (int)u - ((127 - 7) << 3);
  if (code > 0x7e) code = 0x7e;
  return (uint8_t)code;
}

float arcq_o_ue4m3_decode(uint8_t b) { /* Ue4m32Float, reorder.cu:101 */
  int e = (b >> 3) & 0xf, m = b & 7;
  if (e == 0) return (float)m * 0.001953125f;
  return ldexpf(1.0f + (float)m * 0.125f, e - 7);
}

/* ------------------------------------------------------------------------------------------ */
/* layouts                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* Scale-factor byte offset of (row r, group position p) for K = KQ+KE columns.
 * CUTLASS Sm1xxBlkScaledConfig::tile_atom_to_shape_SF{A,B} as used at reorder.cuh:118-123 and
 * addressed at reorder.cu:139-143: coords ((r%32,(r/32)%4), r/128), ((0,p%4), p/4);
 * atom = 128 rows x 4 groups = 512 B with strides (16, 4 | 1), atoms K-major. */
int64_t arcq_o_sf_offset(int64_t r, int64_t p, int64_t K) {
  int64_t atoms_k = K / 64;
  return (r / 128) * atoms_k * 512 + (p / 4) * 512 + (r % 32) * 16 + ((r / 32) % 4) * 4 + (p % 4);
}

/* bindings.cpp:83-95: allocated size (one spare 128-row tile when rows % 128 == 0). */
int64_t arcq_o_sf_alloc_bytes(int64_t rows, int64_t K) { return (rows / 128 + 1) * 128 * K / 16; }
int64_t arcq_o_sf_used_bytes(int64_t rows, int64_t K) { return ((rows + 127) / 128) * 128 * K / 16; }

/* Augmented-K map: position (in units of 16-element groups) of reordered group g.
 * G16: pos = tid + max(0, tid - (KQ-KE)/16), residual at pos+1          (reorder.cu:139,175)
 * G32: thread t owns groups 2t,2t+1; pos1 = 2t + max(0, 2t - (KQ-KE)/16); primaries at
 *      pos1, pos1+1, residuals at pos1+2, pos1+3                         (reorder.cu:451-452,510,515) */
int64_t arcq_o_primary_pos(int64_t g, int64_t KQ, int64_t KE, int variant) {
  int64_t P = (KQ - KE) / 16;
  if (variant == ARCQ_VARIANT_G16) return g + (g > P ? g - P : 0);
  int64_t t = g / 2, g1 = 2 * t;
  int64_t pos1 = g1 + (g1 > P ? g1 - P : 0);
  return pos1 + (g & 1);
}
/* -1 when the group has no residual / duplicate slot */
int64_t arcq_o_residual_pos(int64_t g, int64_t KQ, int64_t KE, int variant) {
  int64_t P = (KQ - KE) / 16;
  if (g < P) return -1;
  if (variant == ARCQ_VARIANT_G16) return arcq_o_primary_pos(g, KQ, KE, variant) + 1;
  return arcq_o_primary_pos(g, KQ, KE, variant) + 2;
}

/* ------------------------------------------------------------------------------------------ */
/* semantics switches (TEST-ONLY: pin the shared code below to the reference's Python fake path) */
/* ------------------------------------------------------------------------------------------ */
#define ARCQ_SEM_TIE_FIRSTMIN 1  /* e2m1 by 15-way argmin, first minimum wins (kernels/fake.py:6-16) instead of RNE        */
#define ARCQ_SEM_DIV_TRUE 2      /* x / s (kernels/fake.py:51) instead of x * (float)(1.0 / s) (reorder.cu:146,153)          */
#define ARCQ_SEM_SCALE_FAKE 4    /* floor(log2)-rebuilt 3-bit-mantissa scale with a floor, no subnormals (fake.py:20-30)     */
#define ARCQ_SEM_RESID_F32 8     /* residual x - q*s kept in fp32 (model/quantize.py:265 on an fp32 tensor), no bf16 rounding */

typedef struct {
  int flags;
  float scale_floor; /* 1/512 in kernels/fake.py:21, 2e-3 in model/quantize.py:41 */
  float log_eps;     /* 0 in kernels/fake.py:23, 1e-9 in model/quantize.py:43     */
} Sem;
static const Sem SEM_KERNEL = {0, 0.0f, 0.0f};

/* quantize_e2m1 of the fake path (kernels/fake.py:6-16 == model/quantize.py:14-22): |t - g| in fp32 for the 15
 * values in ascending order, first minimum.  Returns an e2m1 code; zero has no sign (the grid holds one 0.0). */
static uint8_t fake_e2m1_encode(float t) {
  static const float grid[15] = {-6.0f, -4.0f, -3.0f, -2.0f, -1.5f, -1.0f, -0.5f, 0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
  static const uint8_t code[15] = {15, 14, 13, 12, 11, 10, 9, 0, 1, 2, 3, 4, 5, 6, 7};
  int best = 0;
  float bd = fabsf(t - grid[0]);
  for (int i = 1; i < 15; ++i) {
    float d = fabsf(t - grid[i]);
    if (d < bd) { bd = d; best = i; }
  }
  return code[best];
}

/* quantize_ue4m3 of the fake path (kernels/fake.py:20-30 / model/quantize.py:40-49), fp32 */
static float fake_ue4m3(float s, const Sem *sem) {
  s = s < sem->scale_floor ? sem->scale_floor : s;
  s = s > FP8_MAX ? FP8_MAX : s;
  float e = floorf(log2f(sem->log_eps != 0.0f ? s + sem->log_eps : s));
  float p2 = exp2f(e);                     /* 2**exponent: exact */
  float mant = s / p2 - 1.0f;
  float qm = nearbyintf(mant * 8.0f) / 8.0f; /* torch.round: half to even */
  return (1.0f + qm) * p2;
}

/* ------------------------------------------------------------------------------------------ */
/* one 16-element group                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* amax -> (fp32 scale, ue4m3 byte, reciprocal) : reorder.cu:138,143,146 */
/* `sdec` = the scale the codes are divided by and dequantised with: the decoded ue4m3 byte, or -- under
 * ARCQ_SEM_SCALE_FAKE -- the fake path's rebuilt scale, which need not be an e4m3 value (then *s8 = 0xff). */
static void group_scale(const float *v, const Sem *sem, float *scale, uint8_t *s8, float *sdec, float *rscale) {
  float maxv = 0.0f;
  for (int i = 0; i < 16; ++i) {
    float a = fabsf(v[i]);
    maxv = maxv > a ? maxv : a;          /* mymax, reorder.cu:53-63 == torch.max(torch.abs(.)), fake.py:45 */
  }
  float s = maxv / FP4_MAX;              /* A3; fake.py:46 */
  if (sem->flags & ARCQ_SEM_SCALE_FAKE) {
    if (s == 0.0f) s = 1e-9f;            /* fake.py:47 */
    *scale = s;
    *sdec = fake_ue4m3(s, sem);
    uint8_t b = arcq_o_ue4m3_encode(*sdec);
    *s8 = arcq_o_ue4m3_decode(b) == *sdec ? b : 0xff;
  } else {
    s = s < SCALE_EPS ? SCALE_EPS : s;   /* clamp(.., SCALE_EPS, FP8_MAX) reorder.cu:37,138 */
    s = s > FP8_MAX ? FP8_MAX : s;
    *scale = s;
    *s8 = arcq_o_ue4m3_encode(s);
    *sdec = arcq_o_ue4m3_decode(*s8);
  }
  *rscale = (float)(1.0 / (double)*sdec); /* `1.0 / float` is a double division */
}

static inline float clamp6(float x) { /* clamp(x,-6,6) via fpmax(a,fpmin(b,x)) reorder.cu:33-37 */
  float t = 6.0f < x ? 6.0f : x;
  return -6.0f > t ? -6.0f : t;
}

static void pack8(const uint8_t *codes, uint8_t *out) { /* PackFp4{low,high} reorder.cu:28-31,161-164 */
  for (int j = 0; j < 8; ++j) out[j] = (uint8_t)((codes[2 * j] & 0xf) | (codes[2 * j + 1] << 4));
}

/* Quantise 16 values (fp32 copies of bf16) -> 8 packed bytes + scale byte.  If resid != NULL also
 * emit the bf16-rounded residuals (as fp32) with S = rounded (G16) or un-rounded (G32) scale. */
static void quant_group(const float *v, const Sem *sem, uint8_t *packed, uint8_t *s8_out, float *sdec_out, float *resid,
                        int variant) {
  float scale, rscale, sdec;
  uint8_t s8;
  group_scale(v, sem, &scale, &s8, &sdec, &rscale);
  float S = variant == ARCQ_VARIANT_G16 ? sdec : scale; /* reorder.cu:157 vs :474 */
  uint8_t codes[16];
  for (int i = 0; i < 16; ++i) {
    float t = (sem->flags & ARCQ_SEM_DIV_TRUE) ? v[i] / sdec : v[i] * rscale;   /* fake.py:51 | reorder.cu:153 */
    codes[i] = (sem->flags & ARCQ_SEM_TIE_FIRSTMIN) ? fake_e2m1_encode(t) : arcq_o_e2m1_encode(clamp6(t));
    if (resid) {
      float q = arcq_o_e2m1_decode(codes[i]);
      if (sem->flags & ARCQ_SEM_RESID_F32) {
        resid[i] = v[i] - q * S;                                       /* x - q_x in fp32, model/quantize.py:265 */
      } else {
        float d = fmaf(-q, S, v[i]);                                   /* reorder.cu:157, A2 */
        resid[i] = arcq_o_bf16_to_f32(arcq_o_f32_to_bf16(d));
      }
    }
  }
  pack8(codes, packed);
  *s8_out = s8;
  if (sdec_out) *sdec_out = sdec;
}

/* ------------------------------------------------------------------------------------------ */
/* quantisers                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* One row that has already been gathered / normalised into fp32 copies of bf16 values. */
/* SF (swizzled ue4m3 bytes, the product format) and/or sf_f32 (row-major [K/16] float scales of THIS row, by position:
 * the only form that can hold a fake-path scale) receive the scales; either may be NULL. */
static void quant_row(const float *xr, int64_t row, int64_t KQ, int64_t KE, int variant, int is_weight, const Sem *sem,
                      uint8_t *q_row, uint8_t *SF, float *sf_f32) {
  int64_t K = KQ + KE, G = KQ / 16, P = (KQ - KE) / 16;
  for (int64_t g = 0; g < G; ++g) {
    float resid[16], sdec;
    uint8_t packed[8], s8;
    int has_res = g >= P;
    quant_group(xr + 16 * g, sem, packed, &s8, &sdec, (has_res && !is_weight) ? resid : NULL, variant);
    int64_t p = arcq_o_primary_pos(g, KQ, KE, variant);
    memcpy(q_row + 8 * p, packed, 8);
    if (SF) SF[arcq_o_sf_offset(row, p, K)] = s8;
    if (sf_f32) sf_f32[p] = sdec;
    if (has_res) {
      int64_t pr = arcq_o_residual_pos(g, KQ, KE, variant);
      if (is_weight) {                       /* duplicate codes + scale: reorder.cu:306-316, 671-683 */
        memcpy(q_row + 8 * pr, packed, 8);
        if (SF) SF[arcq_o_sf_offset(row, pr, K)] = s8;
        if (sf_f32) sf_f32[pr] = sdec;
      } else {                               /* quantise the residual: reorder.cu:168-190, 502-541 */
        uint8_t rp[8], rs8;
        float rdec;
        quant_group(resid, sem, rp, &rs8, &rdec, NULL, variant);
        memcpy(q_row + 8 * pr, rp, 8);
        if (SF) SF[arcq_o_sf_offset(row, pr, K)] = rs8;
        if (sf_f32) sf_f32[pr] = rdec;
      }
    }
  }
}

/* agemm.reorder_quantize_x : bindings.cpp:122-163 -> reorder.cu:68-203 / 380-555 / down.cu:71-233.
 * X [M,KQ] bf16 bits, idx [KQ] int16, QX [M,(KQ+KE)/2], SFX >= sf_used_bytes (bytes not written by
 * the kernel are left untouched, as with torch::empty). Returns 0, or -1 on a bad shape. */
int arcq_o_quantize_x(const uint16_t *X, const int16_t *idx, int64_t M, int64_t KQ, int64_t KE, int variant,
                      uint8_t *QX, uint8_t *SFX) {
  if (KQ % 16 || KE % 16 || KE > KQ || KE < 0) return -1;
  if (variant == ARCQ_VARIANT_G32 && (KQ % 32 || KE % 32)) return -1;
  float *xr = (float *)malloc(sizeof(float) * (size_t)KQ);
  for (int64_t m = 0; m < M; ++m) {
    for (int64_t c = 0; c < KQ; ++c) xr[c] = arcq_o_bf16_to_f32(X[m * KQ + idx[c]]); /* reorder.cu:114-118 */
    quant_row(xr, m, KQ, KE, variant, 0, &SEM_KERNEL, QX + m * (KQ + KE) / 2, SFX, NULL);
  }
  free(xr);
  return 0;
}

/* agemm.reorder_quantize_w : bindings.cpp:170-210 -> reorder.cu:210-330 / 562-696 / down.cu:240-361 */
int arcq_o_quantize_w(const uint16_t *W, const int16_t *idx, int64_t N, int64_t KQ, int64_t KE, int variant,
                      uint8_t *QW, uint8_t *SFW) {
  if (KQ % 16 || KE % 16 || KE > KQ || KE < 0) return -1;
  if (variant == ARCQ_VARIANT_G32 && (KQ % 32 || KE % 32)) return -1;
  float *xr = (float *)malloc(sizeof(float) * (size_t)KQ);
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t c = 0; c < KQ; ++c) xr[c] = arcq_o_bf16_to_f32(W[n * KQ + idx[c]]);
    quant_row(xr, n, KQ, KE, variant, 1, &SEM_KERNEL, QW + n * (KQ + KE) / 2, SFW, NULL);
  }
  free(xr);
  return 0;
}

/* TEST-ONLY twin of arcq_o_quantize_{x,w} with selectable semantics (ARCQ_SEM_*), for pinning the shared code above to
 * the reference's Python fake path.  X is fp32 [rows, KQ] ALREADY in reordered channel order (the fake path
 * quantises blocks of original channels, so only the identity permutation is element-comparable); outputs:
 *   Q   [rows, K/2]  packed codes in the augmented-K layout (as the product format),
 *   SFf [rows, K/16] float scale per group POSITION (row-major, not swizzled),
 *   DQ  [rows, K]    decode(code) * scale by position -- what the fake path returns, for the G16 layout with the
 *                    residual/duplicate groups gathered behind the primaries by the caller.
 * flags = 0 reproduces arcq_o_quantize_{x,w} on bf16-representable input (tests assert this). */
int arcq_o_quantize_sem(const float *X, int64_t rows, int64_t KQ, int64_t KE, int variant, int is_weight, int flags,
                        float scale_floor, float log_eps, uint8_t *Q, float *SFf, float *DQ) {
  if (KQ % 16 || KE % 16 || KE > KQ || KE < 0) return -1;
  if (variant == ARCQ_VARIANT_G32 && (KQ % 32 || KE % 32)) return -1;
  Sem sem = {flags, scale_floor, log_eps};
  int64_t K = KQ + KE;
  for (int64_t r = 0; r < rows; ++r) {
    uint8_t *q = Q + r * (K / 2);
    float *sf = SFf + r * (K / 16);
    quant_row(X + r * KQ, r, KQ, KE, variant, is_weight, &sem, q, NULL, sf);
    if (DQ)
      for (int64_t p = 0; p < K / 16; ++p)
        for (int i = 0; i < 16; ++i) {
          uint8_t b = q[8 * p + i / 2];
          uint8_t c = (i & 1) ? (uint8_t)(b >> 4) : (uint8_t)(b & 0xf);
          DQ[r * K + 16 * p + i] = arcq_o_e2m1_decode(c) * sf[p];
        }
  }
  return 0;
}

/* Block sum of squares in the reference's order (rmsnorm.cu:113-154), bdx = KQ/16 threads.
 * Thread t loads 16-byte chunks t and bdx+t of the row (8 bf16 each), sums squares sequentially in
 * fp32, then: smem tree 256,128 (guarded by < bdx), 64, 32, then a 32-lane shuffle tree 16..1. */
static float rms_sumsq(const uint16_t *x, int64_t KQ) {
  int bdx = (int)(KQ / 16);
  float *s = (float *)calloc((size_t)(bdx < 512 ? 512 : bdx), sizeof(float));
  for (int t = 0; t < bdx; ++t) {
    float acc = 0.0f;
    for (int it = 0; it < 2; ++it) {
      const uint16_t *p = x + (int64_t)it * bdx * 8 + (int64_t)t * 8;
      for (int j = 0; j < 8; ++j) {
        float v = arcq_o_bf16_to_f32(p[j]);
        acc = acc + v * v; /* v*v exact in fp32 */
      }
    }
    s[t] = acc;
  }
  /* Each stage reads partner values written by the previous stage; the per-thread register `sumv`
   * equals s[t] for every thread that is still active, so the array form below is equivalent. */
  for (int t = 0; t < 256 && t < bdx; ++t) s[t] = s[t] + ((t + 256) < bdx ? s[t + 256] : 0.0f);
  for (int t = 0; t < 128 && t < bdx; ++t) s[t] = s[t] + ((t + 128) < bdx ? s[t + 128] : 0.0f);
  for (int t = 0; t < 64 && t < bdx; ++t) s[t] = s[t] + s[t + 64];  /* unguarded in the reference: */
  for (int t = 0; t < 32 && t < bdx; ++t) s[t] = s[t] + s[t + 32];  /* bdx >= 128 always (KQ>=2048) */
  for (int sh = 16; sh > 0; sh >>= 1)
    for (int t = 0; t < sh; ++t) s[t] = s[t] + s[t + sh];            /* lane 0's dependency cone */
  float r = s[0];
  free(s);
  return r;
}

/* agemm.rmsnorm_quantize_x : bindings.cpp:216-254 -> rmsnorm.cu:68-255.
 * The reference always uses the G16 layout here; `variant` lets the caller keep x and w consistent
 * for KQ in the G32 set (deviation D1). */
int arcq_o_rmsnorm_quantize_x(const uint16_t *X, const uint16_t *Wn, float eps, const int16_t *idx, int64_t M,
                              int64_t KQ, int64_t KE, int variant, uint8_t *QX, uint8_t *SFX) {
  if (KQ % 16 || KE % 16 || KE > KQ || KE < 0 || KQ < 2048 || KQ > 8192) return -1;
  if (variant == ARCQ_VARIANT_G32 && (KQ % 32 || KE % 32)) return -1;
  float *xr = (float *)malloc(sizeof(float) * (size_t)KQ);
  for (int64_t m = 0; m < M; ++m) {
    const uint16_t *x = X + m * KQ;
    float sum = rms_sumsq(x, KQ);
    float var = sum / (float)KQ + eps;                    /* rmsnorm.cu:157, A3 */
    float rstd = (float)(1.0 / sqrt((double)var));        /* A4 */
    for (int64_t c = 0; c < KQ; ++c) {
      int i = idx[c];
      float v = arcq_o_bf16_to_f32(x[i]) * arcq_o_bf16_to_f32(Wn[i]) * rstd; /* rmsnorm.cu:170 */
      xr[c] = arcq_o_bf16_to_f32(arcq_o_f32_to_bf16(v));
    }
    quant_row(xr, m, KQ, KE, variant, 0, &SEM_KERNEL, QX + m * (KQ + KE) / 2, SFX, NULL);
  }
  free(xr);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* dequantisation and GEMM                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* Format-spec dequantisation of a packed operand: out[r, 16p+i] = e2m1(code) * ue4m3(SF[r,p]).
 * (nvfp4.cu:10-17: row-major e2m1 with one ue4m3 scale per 16 along K.) */
void arcq_o_dequant(const uint8_t *Q, const uint8_t *SF, int64_t rows, int64_t K, float *out) {
  for (int64_t r = 0; r < rows; ++r)
    for (int64_t p = 0; p < K / 16; ++p) {
      float s = arcq_o_ue4m3_decode(SF[arcq_o_sf_offset(r, p, K)]);
      for (int i = 0; i < 16; ++i) {
        uint8_t b = Q[r * (K / 2) + 8 * p + i / 2];
        uint8_t c = (i & 1) ? (b >> 4) : (b & 0xf);
        out[r * K + 16 * p + i] = arcq_o_e2m1_decode(c) * s; /* exact: 2 x 4 significant bits */
      }
    }
}

/* agemm.matmul : bindings.cpp:99-120 -> nvfp4.cu:35-132.  D = bf16(alpha * acc), beta = 0.
 * acc is accumulated in fp64 here (the products are exact in fp32/fp64; the reference accumulates in
 * fp32 in a hardware-defined order).  Dexact (optional) receives alpha*acc in fp64; Dabs (optional)
 * receives alpha * sum |a*b| for error-bound checks; Dbf16 (optional) the rounded output bits. */
int arcq_o_gemm(const uint8_t *A, const uint8_t *B, const uint8_t *SFA, const uint8_t *SFB, int64_t M, int64_t N,
                int64_t K, float alpha, uint16_t *Dbf16, double *Dexact, double *Dabs) {
  if (K % 64) return -1;
  float *a = (float *)malloc(sizeof(float) * (size_t)(M * K));
  float *b = (float *)malloc(sizeof(float) * (size_t)(N * K));
  arcq_o_dequant(A, SFA, M, K, a);
  arcq_o_dequant(B, SFB, N, K, b);
  for (int64_t m = 0; m < M; ++m)
    for (int64_t n = 0; n < N; ++n) {
      double acc = 0.0, aabs = 0.0;
      const float *ar = a + m * K, *br = b + n * K;
      for (int64_t k = 0; k < K; ++k) {
        double p = (double)ar[k] * (double)br[k];
        acc += p;
        aabs += fabs(p);
      }
      if (Dexact) Dexact[m * N + n] = (double)alpha * acc;
      if (Dabs) Dabs[m * N + n] = fabs((double)alpha) * aabs;
      if (Dbf16) Dbf16[m * N + n] = arcq_o_f32_to_bf16(alpha * (float)acc); /* epilogue in fp32 */
    }
  free(a);
  free(b);
  return 0;
}
