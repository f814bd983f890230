/*
 * arcq.h -- C-ABI of the MI355X-native (gfx950) ARC-NVFP4 hot path.
 *
 * This is the drop-in boundary: every entry point takes plain device pointers, sizes and a HIP
 * stream, allocates nothing and returns an int status.  The library holds no caller-visible state: scratch
 * is caller-owned; internally it only remembers, per device, which kernels already have their large-LDS
 * opt-in (thread-safe, any number of devices per process) and the calling thread's last error text.  It replaces what the
 * reference's pybind11 module `agemm` (kernels/src/bindings.cpp:551-575) reaches through
 * libtorch + CUTLASS.  Allocation of outputs stays in the host shim (arcquant_amd/agemm.py), with
 * the reference's sizes (bindings.cpp:83-95,110,133-134,181-182).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host
 *   - bf16 tensors are passed as `const void*` (16-bit patterns), row-major, contiguous
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, which is what
 *     the reference launches on: reorder.cu:344, rmsnorm.cu:272)
 *   - every launch is asynchronous; nothing here synchronises the device
 *   - status: 0 = ok, negative = error (see ARCQ_ERR_*); arcq_last_error() gives the text of the
 *     calling thread's last failure
 *
 * Layout contract (bit-exact with the reference):
 *   packed operand   [rows, K/2] u8, byte j = code[2j] | code[2j+1] << 4   (reorder.cu:28-31,161-164)
 *   scale factors    ue4m3 bytes in the CUTLASS Sm1xx block-scaled layout:
 *                    off(r,p) = (r/128)*(K/64)*512 + (p/4)*512 + (r%32)*16 + ((r/32)%4)*4 + p%4
 *                    (reorder.cuh:118-123, reorder.cu:139-143)
 *   augmented K      K = KQ + KE; group g of the reordered row goes to position
 *                    G16: g + max(0, g-P), residual/duplicate at +1          (reorder.cu:139,175)
 *                    G32: pos1 = 2t + max(0, 2t-P) (+g&1), residual at +2    (reorder.cu:451-452,510)
 *                    with P = (KQ-KE)/16, t = g/2.
 */
#ifndef ARCQ_H_
#define ARCQ_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARCQ_ABI_VERSION 1

#define ARCQ_OK 0
#define ARCQ_ERR_SHAPE (-1)       /* a dimension violates the contract (K % 64, KE % 16, KE > KQ, ...) */
#define ARCQ_ERR_UNSUPPORTED (-2) /* valid but not implemented (e.g. rows that do not fit in LDS)        */
#define ARCQ_ERR_LAUNCH (-3)      /* hipLaunchKernel / hipGetLastError reported a failure               */
#define ARCQ_ERR_NULL (-4)        /* a required pointer is NULL                                          */
#define ARCQ_ERR_WORKSPACE (-5)   /* workspace missing or too small (see arcq_gemm_workspace_bytes)     */

/* Augmented-K layout variants.  The reference picks one by KQ in a closed switch
 * (bindings.cpp:141-160): G32 for {3584, 18944, 27648, 28672}, G16 for its other sizes. */
#define ARCQ_VARIANT_G16 0 /* reorder.cu:68-330, rmsnorm.cu:68-255 */
#define ARCQ_VARIANT_G32 1 /* reorder.cu:380-696, down.cu:71-361   */

#define ARCQ_OUT_BF16 0 /* D is bf16 [M,N]  (the reference's only output type, nvfp4.cu:20-23) */
#define ARCQ_OUT_F32 1  /* D is fp32 [M,N]  (un-rounded alpha*acc: row-parallel partials, tests) */

int arcq_abi_version(void);
const char *arcq_last_error(void);

/* ---- host-side layout helpers (pure integer arithmetic, no GPU) ------------------------------------ */

/* replaces the KQ switch of bindings.cpp:141-160,189-207: KQ-generic rule with the same answers on the
 * reference's closed list. */
int arcq_variant_for_kq(int64_t KQ);
/* bindings.cpp:83-95 get_sf{a,b}_buffer_size_in_bytes: (rows/128 + 1) * 128 * K / 16 */
int64_t arcq_sf_alloc_bytes(int64_t rows, int64_t K);
/* bytes of that buffer a kernel may touch: ceil(rows/128) * 128 * K / 16 */
int64_t arcq_sf_used_bytes(int64_t rows, int64_t K);
int64_t arcq_sf_offset(int64_t row, int64_t pos, int64_t K);
int64_t arcq_primary_pos(int64_t g, int64_t KQ, int64_t KE, int variant);
int64_t arcq_residual_pos(int64_t g, int64_t KQ, int64_t KE, int variant); /* -1 if none */

/* ---- quantisers ------------------------------------------------------------------------------------ */

/* agemm.reorder_quantize_x (bindings.cpp:122-163; kernels reorder.cu:68-203, 380-555, down.cu:71-233).
 *   X [M,KQ] bf16, reorder_index [KQ] int16 (a permutation of 0..KQ-1), QX [M,(KQ+KE)/2] u8,
 *   SFX >= arcq_sf_alloc_bytes(M, KQ+KE) bytes (only the offsets of rows < M are written).
 *   KQ % 64 == 0, KE % 64 == 0, 0 <= KE <= KQ <= 32767 (whole 64-element scale-factor atoms: every size the
 *   reference dispatches and every select_num it produces).  X / reorder_index 16-byte, QX 8-byte, SFX 4-byte aligned. */
int arcq_quantize_x(const void *X, const int16_t *reorder_index, uint8_t *QX, uint8_t *SFX, int64_t M, int64_t KQ,
                    int64_t KE, int variant, void *stream);

/* agemm.reorder_quantize_w (bindings.cpp:170-210; kernels reorder.cu:210-330, 562-696, down.cu:240-361):
 * as above, but the residual slots are copies of the primary codes and scale. */
int arcq_quantize_w(const void *W, const int16_t *reorder_index, uint8_t *QW, uint8_t *SFW, int64_t N, int64_t KQ,
                    int64_t KE, int variant, void *stream);

/* agemm.rmsnorm_quantize_x (bindings.cpp:216-254; kernel rmsnorm.cu:68-255):
 *   xn = bf16(float(x) * float(w) * rsqrt(sum(x^2)/KQ + eps)) gathered by reorder_index, then as
 *   arcq_quantize_x.  2048 <= KQ <= 8192 (the reference's range).  PARITY UNPINNED against the CUDA binary: the
 *   reference's rsqrtf is a <= 2-ulp approximation, this kernel (and the oracle) use the correctly rounded
 *   1/sqrt; the sum-of-squares association order and the fused residual multiply-add are restated from the kernel
 *   text.  Byte-exact to that restatement (oracle/arcq_oracle.c), not provably to the reference's output.  The reference always uses the G16
 *   layout here; pass the variant the weights were quantised with (deviation D1). */
int arcq_rmsnorm_quantize_x(const void *X, const void *W, float eps, const int16_t *reorder_index, uint8_t *QX,
                            uint8_t *SFX, int64_t M, int64_t KQ, int64_t KE, int variant, void *stream);

/* ---- GEMM ------------------------------------------------------------------------------------------ */

/* Bytes of scratch arcq_gemm_nvfp4 needs for this shape (0 when none). */
int64_t arcq_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);

/* agemm.matmul (bindings.cpp:99-120 -> matmul_host_nvfp4_bf16, nvfp4.cu:35-132):
 *   D[m,n] = alpha * sum_g sfa[m,g]*sfb[n,g] * sum_{k in g} a[m,k]*b[n,k]      (fp32 accumulate, beta = 0)
 *   A [M,K/2], B [N,K/2] packed e2m1; SFA/SFB swizzled ue4m3; K % 64 == 0.
 *   alpha = alpha_host * (alpha_dev ? *alpha_dev : 1): the reference takes a host float
 *   (bindings.cpp:104); alpha_dev lets callers keep the per-tensor scale on the device (no sync).
 *   bias (optional, bf16 [N]) and residual (optional, bf16 [M,N], may alias D) are each added to the ROUNDED bf16 result,
 *   rounding again -- exactly the reference's separate torch ops `y = matmul(...); y = y + bias` (qLinearLayer.py:74-76)
 *   and the caller's `x + linear(...)`.  out_dtype: ARCQ_OUT_* (fp32 output adds both in fp32, one result).
 *   workspace / workspace_bytes: scratch of at least arcq_gemm_workspace_bytes(M,N,K) (may be NULL if 0). */
int arcq_gemm_nvfp4(const uint8_t *A, const uint8_t *B, const uint8_t *SFA, const uint8_t *SFB, void *D, int64_t M,
                    int64_t N, int64_t K, float alpha_host, const float *alpha_dev, const void *bias, const void *residual,
                    int out_dtype,
                    void *workspace, int64_t workspace_bytes, void *stream);

/* ---- f1 extension: per-tensor scale on the device (replaces torch.max(x.abs())/2688 + x/scale,
 *      model/qLlamaLayer.py:73-77, with no host sync) ------------------------------------------------- */

/* scale_out[0] = max|x| / (448*6) as fp32; x is bf16 [n]. */
int arcq_absmax_scale(const void *X, int64_t n, float *scale_out, void *stream);

/* NVFP4_reorder_quantize_x (model/qLlamaLayer.py:73-77) in two launches instead of five:
 *   scale = max|X| / 2688 (fp32, written to scale_out[0]);  (QX, SFX) = arcq_quantize_x(bf16(X / scale), ...).
 * `state` is ARCQ_DYN_STATE_BYTES of device scratch owned by the caller (one abs-max word per workgroup of the
 * abs-max pass; plain stores, fully rewritten by every call, so it needs no initialisation and no reset; it must not
 * be shared by calls that may run concurrently on different streams).  Inputs <= 256 KB take a single launch and do
 * not touch it. */
#define ARCQ_DYN_STATE_BYTES 1024
int arcq_quantize_x_dyn(const void *X, const int16_t *reorder_index, uint8_t *QX, uint8_t *SFX, float *scale_out,
                        void *state, int64_t M, int64_t KQ, int64_t KE, int variant, void *stream);

/* Extension: the MLP's `act_fn(gate) * up` (model/qLlamaLayer.py:417, SiLU) folded into the dynamic quantiser.
 * GU = [M, 2*KQ] bf16, the output of a fused gate_up projection, in one of two layouts: ARCQ_GU_HALVES = gate in
 * columns [0, KQ) and up in [KQ, 2*KQ); ARCQ_GU_PAIRS = (g0, u0, g1, u1, ...), the layout of a weight whose gate and
 * up rows interleave (what arcq_gemm_nvfp4_silu_mul consumes).  Equals arcq_quantize_x_dyn on the bf16 tensor torch
 * computes as silu(gate) * up, which is never materialised.  `state` as above. */
#define ARCQ_GU_HALVES 0
#define ARCQ_GU_PAIRS 1
int arcq_silu_mul_quantize_x_dyn(const void *GU, const int16_t *reorder_index, uint8_t *QX, uint8_t *SFX, float *scale_out,
                                 void *state, int64_t M, int64_t KQ, int64_t KE, int variant, int layout, void *stream);

/* ---- f3 extension: the MLP's `act_fn(gate) * up` (SiLU, model/qLlamaLayer.py:417) in the GEMM epilogue -------------
 * B holds the gate and up projections with their ROWS INTERLEAVED (g0, u0, g1, u1, ...: quantise the interleaved
 * weight with arcq_quantize_w as usual), N = 2 * intermediate, N % 8 == 0.  ACT = bf16 [M, N/2] receives
 * silu(bf16(alpha * gate)) * bf16(alpha * up) with torch's roundings (the value `act_fn(gate) * up` has after two
 * separate GEMMs), and absmax_slots[0 .. arcq_gemm_silu_mul_slots(M,N,K)) one max|ACT| word each (bf16 magnitude
 * bits; plain stores, no initialisation needed).  arcq_quantize_x_dyn_slots then quantises ACT with its per-tensor
 * dynamic scale in ONE launch: together two launches replace GEMM + silu + mul + abs-max + quantise. */
int64_t arcq_gemm_silu_mul_slots(int64_t M, int64_t N, int64_t K);
int arcq_gemm_nvfp4_silu_mul(const uint8_t *A, const uint8_t *B, const uint8_t *SFA, const uint8_t *SFB, void *ACT,
                             uint32_t *absmax_slots, int64_t M, int64_t N, int64_t K, float alpha_host,
                             const float *alpha_dev, const void *bias, void *stream);   /* bias: optional bf16 [N], rows interleaved as B's */
/* arcq_quantize_x_dyn with max|X| taken from `nslots` precomputed words instead of an abs-max pass.
 * reorder_index == NULL: X is already in reordered channel order (identity permutation; see act_scatter_index of
 * arcq_linear_rmsnorm_silu_repacked) -- no gather, same bytes. */
int arcq_quantize_x_dyn_slots(const void *X, const int16_t *reorder_index, uint8_t *QX, uint8_t *SFX, float *scale_out,
                              const uint32_t *absmax_slots, int64_t nslots, int64_t M, int64_t KQ, int64_t KE,
                              int variant, void *stream);

/* ---- decode GEMM over a REPACKED weight (optional fast path for M <= 128) ---------------------------------------
 * A static weight can be laid out for the hardware once: RW = tiles of 16 rows x 128 K elements (1 KB) in MFMA operand
 * order (lane 16*q + r of a wave holds the 16 bytes of row r, K elements [32q, 32q+32) of the tile; the tiles of a row
 * block are consecutive), RSF = per pair of tiles 256 bytes, lane l holding the ue4m3 bytes of its two groups in the
 * first and in the second tile; K padded to a multiple of 256 and N to a multiple of 16 with zero scales.  Sizes from
 * arcq_repacked_{w,sf}_bytes; arcquant_amd/agemm.py:repack_w builds both from the reference layout (pure data movement,
 * nothing is re-quantised).  A / SFA stay in the reference layout.  arcq_gemm_repacked_supported tells whether a shape
 * can take this path -- M <= 16 and the fp16 image of the activations fits LDS, or a decode batch 16 < M <= 128 on a
 * weight where reading it once beats the tiled GEMM (activations fetched per tile pair or kept packed in LDS; gemm_rowmid.hip) --;
 * results equal arcq_gemm_nvfp4's up to fp32 accumulation order. */
int64_t arcq_repacked_w_bytes(int64_t N, int64_t K);
int64_t arcq_repacked_sf_bytes(int64_t N, int64_t K);
int arcq_gemm_repacked_supported(int64_t M, int64_t N, int64_t K);
int arcq_gemm_nvfp4_repacked(const uint8_t *A, const uint8_t *RW, const uint8_t *SFA, const uint8_t *RSF, void *D,
                             int64_t M, int64_t N, int64_t K, float alpha_host, const float *alpha_dev,
                             const void *bias, const void *residual, int out_dtype, void *stream);

/* The same contraction through the kernel of the fused decode linears below (arcq_linear_*): identical arguments and
 * result up to fp32 accumulation order.  arcq_gemm_nvfp4_repacked picks the faster kernel for plain packed activations; this
 * entry point exists so that a fused linear can be checked BIT FOR BIT against its two-call equivalent
 * (quantiser + this GEMM): same kernel body, same summation order. */
int arcq_gemm_nvfp4_repacked_stream(const uint8_t *A, const uint8_t *RW, const uint8_t *SFA, const uint8_t *RSF, void *D,
                                    int64_t M, int64_t N, int64_t K, float alpha_host, const float *alpha_dev,
                                    const void *bias, const void *residual, int out_dtype, void *stream);

/* The gate|up projection of a decode step on the repacked path, for weights whose ROWS INTERLEAVE gate and up (g0, u0, g1,
 * u1, ...): D = bf16 [M, N] as arcq_gemm_nvfp4_repacked writes it, and absmax_slots[i], i < ceil(N / 16), = max |silu(g) * u|
 * (bf16 bits, computed from the stored values with the quantiser's own rounding) over the outputs of row block i.
 * arcq_silu_mul_quantize_x_dyn_slots then quantises act = silu(gate) * up (model/qLlamaLayer.py:417 -> :73-77) from D in ONE
 * launch: the abs-max pass of arcq_silu_mul_quantize_x_dyn disappears, the bytes are identical.  N % 4 == 0. */
int arcq_gemm_nvfp4_repacked_silu_absmax(const uint8_t *A, const uint8_t *RW, const uint8_t *SFA, const uint8_t *RSF, void *D,
                                         uint32_t *absmax_slots, int64_t M, int64_t N, int64_t K, float alpha_host,
                                         const float *alpha_dev, void *stream);
int arcq_silu_mul_quantize_x_dyn_slots(const void *GU, const int16_t *reorder_index, uint8_t *QX, uint8_t *SFX,
                                       float *scale_out, const uint32_t *absmax_slots, int64_t nslots, int64_t M,
                                       int64_t KQ, int64_t KE, int variant, int layout, void *stream);

/* ---- fused decode linear: the activation quantiser runs as the prologue of the repacked GEMM ------------------------------
 * One launch replaces  [rmsnorm_quantize_x | abs-max + x/scale + reorder_quantize_x]  ->  matmul (+ bias) (+ residual)
 * of a decode step (model/qLlamaLayer.py:73-77 + qLinearLayer.py:62-78; benchmarks/modeling_arc.py:211-228,279-310).  Every
 * workgroup quantises the M <= 16 token rows itself with the quantisers' own group arithmetic, so the result is BIT-IDENTICAL
 * to arcq_rmsnorm_quantize_x / arcq_quantize_x_dyn followed by arcq_gemm_nvfp4_repacked on the same repacked weight.
 * arcq_linear_fused_supported(kind, M, N, KQ, KE) tells whether a shape fits (M <= 16, LDS); callers fall back to the two
 * calls otherwise.  It is a capability, not a recommendation: every CU repeats the quantisation of all M * KQ elements, which beats
 * the separate quantiser launch (~3 us of launch + cold start) while M * KQ is about 16 K elements or less (measured on MI355X:
 * M = 4, KQ = 3584 / 4096: 1-2 us faster per linear; M = 16, KQ = 4096 or M = 4, KQ = 18944: 2-3x slower). */
#define ARCQ_SRC_RMSNORM 1
#define ARCQ_SRC_DYNAMIC 2
int arcq_linear_fused_supported(int kind, int64_t M, int64_t N, int64_t KQ, int64_t KE);

/* D = alpha * matmul(rmsnorm_quantize_x(X, Wn, eps, reorder_index, KE), W) (+ bias) (+ residual); arguments as in
 * arcq_rmsnorm_quantize_x and arcq_gemm_nvfp4_repacked (K = KQ + KE), alpha = alpha_host * (alpha_dev ? *alpha_dev : 1). */
int arcq_linear_rmsnorm_repacked(const void *X, const void *Wn, float eps, const int16_t *reorder_index, const uint8_t *RW,
                                 const uint8_t *RSF, void *D, int64_t M, int64_t N, int64_t KQ, int64_t KE, int variant,
                                 float alpha_host, const float *alpha_dev, const void *bias, const void *residual, int out_dtype,
                                 void *stream);
/* The gate|up projection of the MLP: as above for a weight whose ROWS INTERLEAVE gate and up (g0, u0, g1, u1, ...); ACT = bf16
 * [M, N/2] receives silu(bf16(alpha*gate)) * bf16(alpha*up) with torch's roundings (what `act_fn(gate) * up`,
 * model/qLlamaLayer.py:417, has after two GEMMs) and absmax_slots[i], i < ceil(N/16), max |ACT| over row block i (bf16 bits):
 * arcq_linear_dynamic_repacked then quantises ACT without an abs-max pass.  N % 4 == 0. */
int arcq_linear_rmsnorm_silu_repacked(const void *X, const void *Wn, float eps, const int16_t *reorder_index, const uint8_t *RW,
                                      const uint8_t *RSF, void *ACT, uint32_t *absmax_slots, int64_t M, int64_t N, int64_t KQ,
                                      int64_t KE, int variant, float alpha_host, const float *alpha_dev, const void *bias,
                                      const int16_t *act_scatter_index, void *stream);
/* bias: optional bf16 [N], interleaved (g0, u0, g1, u1, ...) like the rows.
 * act_scatter_index: optional int16 [N/2], a permutation (not checked): activation j is stored at ACT[m][act_scatter_index[j]].
 * With act_scatter_index = the inverse of the consumer's reorder_index, ACT is already in reordered channel order and
 * arcq_quantize_x_dyn_slots(ACT, reorder_index = NULL, ...) yields the same bytes as quantising the natural-order
 * activation with that reorder_index. */
/* scale = max|X| / 2688 (from absmax_slots[0..nslots) when given, else computed from X; written to scale_out[0] if not NULL);
 * D = alpha_host * scale * matmul(quantize_x(bf16(X / scale), reorder_index, KE), W) (+ bias) (+ residual): the reference's
 * NVFP4_reorder_quantize_x + QLinearLayer.forward with alpha_host = the weight's per-tensor scale. */
int arcq_linear_dynamic_repacked(const void *X, const int16_t *reorder_index, const uint8_t *RW, const uint8_t *RSF, void *D,
                                 float *scale_out, const uint32_t *absmax_slots, int64_t nslots, int64_t M, int64_t N, int64_t KQ,
                                 int64_t KE, int variant, float alpha_host, const void *bias, const void *residual, int out_dtype,
                                 void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ARCQ_H_ */
