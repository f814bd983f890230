/*
 * arcq_harness.h -- entry points used ONLY by this repository's end-to-end harness (arcquant_amd/e2e.py).
 *
 * NOT part of the drop-in boundary (that is include/arcq.h): nothing in the reference binds these.  The reference's
 * attention is flashinfer over an int4 paged KV cache (kernels/src/flashinfer.cu, bindings.cpp:576-581, model/kv_cache.py),
 * which is out of scope here -- `agemm.batch_decode_*` / `init_kv_*` / `append_kv_*` stay NotImplementedError stubs.  The
 * harness needs SOME attention between the ARC-NVFP4 linears to time a decoder step; it uses torch SDPA for prefill and, for
 * the one-token decode step over its dense bf16 cache, the streaming kernel below (torch's SDPA / bmm take 45-48 us per layer
 * for the 60 MB they read; this takes what the bytes take).
 */
#ifndef ARCQ_HARNESS_H_
#define ARCQ_HARNESS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bytes of fp32 scratch for arcq_harness_attn_decode at this cache size */
int64_t arcq_harness_attn_workspace_bytes(int64_t B, int64_t H, int64_t Tmax);

/* One decode step of attention, head dimension 128: qkv = bf16 [B, 3*H*128] (q | k | v of the ONE new token per sequence, the
 * fused projection's output), kcache / vcache = bf16 [B, H, Tmax, 128].  Appends k / v at position `pos`, then
 * out[b, h*128 + d] = softmax(q k^T / sqrt(128)) v over positions [0, pos] (fp32 math, bf16 result).  One launch on `stream`; `workspace` is
 * only used by the sliced two-launch variant (environment ARCQ_HARNESS_ATTN_SLICED=1). */
int arcq_harness_attn_decode(const void *qkv, void *kcache, void *vcache, void *out, void *workspace, int64_t B, int64_t H,
                             int64_t Tmax, int64_t pos, void *stream);
/* The same over positions [first, pos] only (0 <= first <= pos); first == pos is the attention benchmarks/modeling_arc.py:169-198
 * times in a decode step (each sequence attends over the tokens of the current call). */
int arcq_harness_attn_decode_window(const void *qkv, void *kcache, void *vcache, void *out, void *workspace, int64_t B, int64_t H,
                                    int64_t Tmax, int64_t pos, int64_t first, void *stream);

/* The model's final RMSNorm (a stock module in the reference, not an ARC operator) in one launch: out[r, :] = bf16(float(X[r, :]) *
 * rsqrt(mean(X[r, :]^2) + eps) * float(W)), rows of H bf16 values with row stride ldx elements (H % 8 == 0). */
int arcq_harness_rmsnorm(const void *X, int64_t ldx, const void *W, void *out, int64_t rows, int64_t H, float eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ARCQ_HARNESS_H_ */
