// extern "C" entry points of libarcq_hip.so (declared in include/arcq.h): argument validation,
// dispatch between the GEMM kernels, error text.  No allocation, no synchronisation; the only state is the
// per-device record of LDS opt-ins already made (ensure_dynamic_lds) and the thread-local error text.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "arcq_device.hpp"
#include "arcq_internal.hpp"

namespace arcq {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int ensure_dynamic_lds(const void* kernel, LdsOptIn& cache, int bytes, const char* who) {
  if (bytes <= 48 * 1024) return ARCQ_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "%s: hipGetDevice failed", who);
  const bool cached = dev >= 0 && dev < kMaxDevices;
  if (cached && __atomic_load_n(&cache.granted[dev], __ATOMIC_RELAXED) >= bytes) return ARCQ_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return fail(ARCQ_ERR_LAUNCH, "%s: cannot reserve %d B of LDS: %s", who, bytes, hipGetErrorString(e));
  if (cached) {     // keep the maximum: a smaller request after a larger one must not shrink the record
    int seen = __atomic_load_n(&cache.granted[dev], __ATOMIC_RELAXED);
    while (seen < bytes && !__atomic_compare_exchange_n(&cache.granted[dev], &seen, bytes, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  }
  return ARCQ_OK;
}

}  // namespace arcq

using namespace arcq;

extern "C" {

int arcq_abi_version(void) { return ARCQ_ABI_VERSION; }
const char* arcq_last_error(void) { return g_err; }

// KQ-generic replacement of the closed switch in bindings.cpp:141-160: the reference routes
// {3584, 18944} to the 32-per-thread kernels and {27648, 28672} to the "down" kernels (same layout);
// every other size it supports uses the 16-per-thread layout, which is also our default elsewhere.
int arcq_variant_for_kq(int64_t KQ) {
  return (KQ == 3584 || KQ == 18944 || KQ == 27648 || KQ == 28672) ? ARCQ_VARIANT_G32 : ARCQ_VARIANT_G16;
}

int64_t arcq_sf_alloc_bytes(int64_t rows, int64_t K) { return (rows / 128 + 1) * 128 * K / 16; }   // bindings.cpp:83-95
int64_t arcq_sf_used_bytes(int64_t rows, int64_t K) { return ((rows + 127) / 128) * 128 * K / 16; }
int64_t arcq_sf_offset(int64_t row, int64_t pos, int64_t K) { return sf_offset(row, pos, K); }

int64_t arcq_primary_pos(int64_t g, int64_t KQ, int64_t KE, int variant) {
  const int64_t P = (KQ - KE) / 16;
  if (variant == ARCQ_VARIANT_G16) return g + (g > P ? g - P : 0);
  const int64_t g1 = g & ~(int64_t)1;
  return g1 + (g1 > P ? g1 - P : 0) + (g & 1);
}
int64_t arcq_residual_pos(int64_t g, int64_t KQ, int64_t KE, int variant) {
  const int64_t P = (KQ - KE) / 16;
  if (g < P) return -1;
  return arcq_primary_pos(g, KQ, KE, variant) + (variant == ARCQ_VARIANT_G16 ? 1 : 2);
}

int arcq_quantize_x(const void* X, const int16_t* reorder_index, uint8_t* QX, uint8_t* SFX, int64_t M, int64_t KQ, int64_t KE,
                    int variant, void* stream) {
  return quantize_x(X, reorder_index, QX, SFX, M, KQ, KE, variant, (hipStream_t)stream);
}

int arcq_quantize_w(const void* W, const int16_t* reorder_index, uint8_t* QW, uint8_t* SFW, int64_t N, int64_t KQ, int64_t KE,
                    int variant, void* stream) {
  return quantize_w(W, reorder_index, QW, SFW, N, KQ, KE, variant, (hipStream_t)stream);
}

int arcq_rmsnorm_quantize_x(const void* X, const void* W, float eps, const int16_t* reorder_index, uint8_t* QX, uint8_t* SFX,
                            int64_t M, int64_t KQ, int64_t KE, int variant, void* stream) {
  return rmsnorm_quantize_x(X, W, eps, reorder_index, QX, SFX, M, KQ, KE, variant, (hipStream_t)stream);
}

int arcq_quantize_x_dyn(const void* X, const int16_t* reorder_index, uint8_t* QX, uint8_t* SFX, float* scale_out, void* state,
                        int64_t M, int64_t KQ, int64_t KE, int variant, void* stream) {
  return quantize_x_dyn(X, reorder_index, QX, SFX, scale_out, state, M, KQ, KE, variant, (hipStream_t)stream);
}

int arcq_silu_mul_quantize_x_dyn(const void* GU, const int16_t* reorder_index, uint8_t* QX, uint8_t* SFX, float* scale_out, void* state,
                                 int64_t M, int64_t KQ, int64_t KE, int variant, int layout, void* stream) {
  return silu_mul_quantize_x_dyn(GU, reorder_index, QX, SFX, scale_out, state, M, KQ, KE, variant, layout, (hipStream_t)stream);
}

int arcq_absmax_scale(const void* X, int64_t n, float* scale_out, void* stream) {
  return absmax_scale(X, n, scale_out, (hipStream_t)stream);
}


static const int64_t kSkinnyMaxM = 16;

// M <= 16: the 32-row-tile kernel needs enough tiles to occupy the chip without split-K (its tiles are twice as
// tall); below that the 16-row-tile kernel wins.  Measured crossover (tools/decode_bench.py, M=4, K=4160, us):
// N=4096 7.3 vs 9.3, N=5120 9.2 vs 8.5, N=8192 10.2 vs 9.5, N=14336 16.0 vs 14.6, N=37888 28.7 vs 23.9.
// ARCQ_DECODE=1|2 forces the 16-row / 32-row kernel (tuning).
static bool use_decode_v2(int64_t N) {
  static const int forced = getenv("ARCQ_DECODE") ? atoi(getenv("ARCQ_DECODE")) : 0;
  if (forced == 1) return false;
  if (forced == 2) return true;
  return ((N + 127) / 128) * 4 >= 160;
}

int64_t arcq_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (gemm_regtile_cfg(M, N, K, kEpiPlain)) return 0;
  if (M <= kSkinnyMaxM) return use_decode_v2(N) ? gemm_decode_workspace_bytes(M, N, K) : gemm_skinny_workspace_bytes(M, N, K);
  return gemm_tile_workspace_bytes(M, N, K);
}

int arcq_gemm_nvfp4(const uint8_t* A, const uint8_t* B, const uint8_t* SFA, const uint8_t* SFB, void* D, int64_t M, int64_t N,
                    int64_t K, float alpha_host, const float* alpha_dev, const void* bias, const void* residual, int out_dtype, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  if (M < 0 || N < 0 || K <= 0 || (K % 64))
    return fail(ARCQ_ERR_SHAPE, "arcq_gemm_nvfp4: need M,N >= 0 and K %% 64 == 0 (M=%lld N=%lld K=%lld)", (long long)M,
                (long long)N, (long long)K);
  if (out_dtype != ARCQ_OUT_BF16 && out_dtype != ARCQ_OUT_F32) return fail(ARCQ_ERR_SHAPE, "arcq_gemm_nvfp4: bad out_dtype %d", out_dtype);
  if (M == 0 || N == 0) return ARCQ_OK;
  if (!A || !B || !SFA || !SFB || !D) return fail(ARCQ_ERR_NULL, "arcq_gemm_nvfp4: NULL pointer");
  if (M > INT32_MAX / 2 || N > INT32_MAX / 2 || K > INT32_MAX / 2 || M * N > ((int64_t)1 << 40))
    return fail(ARCQ_ERR_UNSUPPORTED, "arcq_gemm_nvfp4: shape too large");
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(D)) & 15)
    return fail(ARCQ_ERR_SHAPE, "arcq_gemm_nvfp4: A, B and D must be 16-byte aligned");
  if ((reinterpret_cast<uintptr_t>(SFA) | reinterpret_cast<uintptr_t>(SFB)) & 3)
    return fail(ARCQ_ERR_SHAPE, "arcq_gemm_nvfp4: SFA and SFB must be 4-byte aligned");
  GemmArgs a;
  a.A = A; a.B = B; a.SFA = SFA; a.SFB = SFB; a.D = D;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.alpha_host = alpha_host; a.alpha_dev = alpha_dev; a.bias = (const uint16_t*)bias; a.residual = (const uint16_t*)residual; a.out_dtype = out_dtype;
  a.workspace = workspace; a.workspace_bytes = workspace_bytes;
  if (const int cfg = gemm_regtile_cfg(M, N, K, kEpiPlain)) return gemm_regtile(a, cfg, (hipStream_t)stream);
  if (M <= kSkinnyMaxM) return use_decode_v2(N) ? gemm_decode(a, (hipStream_t)stream) : gemm_skinny(a, (hipStream_t)stream);
  return gemm_tile(a, (hipStream_t)stream);
}

int64_t arcq_gemm_silu_mul_slots(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return M <= kSkinnyMaxM ? gemm_decode_silu_slots(M, N, K) : gemm_tile_silu_slots(M, N, K);
}

int arcq_gemm_nvfp4_silu_mul(const uint8_t* A, const uint8_t* B, const uint8_t* SFA, const uint8_t* SFB, void* ACT, uint32_t* absmax_slots,
                             int64_t M, int64_t N, int64_t K, float alpha_host, const float* alpha_dev, const void* bias, void* stream) {
  const char* who = "arcq_gemm_nvfp4_silu_mul";
  if (M < 0 || N < 0 || K <= 0 || (K % 64) || (N % 8))
    return fail(ARCQ_ERR_SHAPE, "%s: need M,N >= 0, K %% 64 == 0 and N %% 8 == 0 (M=%lld N=%lld K=%lld)", who, (long long)M, (long long)N,
                (long long)K);
  if (M == 0 || N == 0) return ARCQ_OK;
  if (!A || !B || !SFA || !SFB || !ACT || !absmax_slots) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
  if (M > INT32_MAX / 2 || N > INT32_MAX / 2 || K > INT32_MAX / 2 || M * N > ((int64_t)1 << 40))
    return fail(ARCQ_ERR_UNSUPPORTED, "%s: shape too large", who);
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(ACT)) & 15)
    return fail(ARCQ_ERR_SHAPE, "%s: A, B and ACT must be 16-byte aligned", who);
  if ((reinterpret_cast<uintptr_t>(SFA) | reinterpret_cast<uintptr_t>(SFB) | reinterpret_cast<uintptr_t>(absmax_slots)) & 3)
    return fail(ARCQ_ERR_SHAPE, "%s: SFA, SFB and absmax_slots must be 4-byte aligned", who);
  GemmArgs a;
  a.A = A; a.B = B; a.SFA = SFA; a.SFB = SFB; a.D = ACT;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.alpha_host = alpha_host; a.alpha_dev = alpha_dev; a.bias = (const uint16_t*)bias; a.residual = nullptr; a.out_dtype = ARCQ_OUT_BF16;
  a.workspace = nullptr; a.workspace_bytes = 0;
  a.epilogue = kEpiSiluMul; a.absmax_slots = absmax_slots;
  if (bias && M <= kSkinnyMaxM) return fail(ARCQ_ERR_UNSUPPORTED, "%s: bias is supported by the tile kernel only (M > 16); decode uses arcq_linear_rmsnorm_silu_repacked", who);
  // the 16-row decode kernel has no fused epilogue: every M <= 16 shape takes the 32-row kernel here
  if (M <= kSkinnyMaxM) return gemm_decode(a, (hipStream_t)stream);
  return gemm_tile(a, (hipStream_t)stream);
}

int arcq_quantize_x_dyn_slots(const void* X, const int16_t* reorder_index, uint8_t* QX, uint8_t* SFX, float* scale_out,
                              const uint32_t* absmax_slots, int64_t nslots, int64_t M, int64_t KQ, int64_t KE, int variant, void* stream) {
  return quantize_x_dyn_slots(X, reorder_index, QX, SFX, scale_out, absmax_slots, nslots, M, KQ, KE, variant, (hipStream_t)stream);
}

int64_t arcq_repacked_w_bytes(int64_t N, int64_t K) { return (N > 0 && K > 0) ? gemm_repacked_w_bytes(N, K) : 0; }
int64_t arcq_repacked_sf_bytes(int64_t N, int64_t K) { return (N > 0 && K > 0) ? gemm_repacked_sf_bytes(N, K) : 0; }
int arcq_gemm_repacked_supported(int64_t M, int64_t N, int64_t K) { return gemm_repacked_supported(M, N, K); }

static int gemm_repacked_entry(bool via_stream, const char* who, const uint8_t* A, const uint8_t* RW, const uint8_t* SFA, const uint8_t* RSF, void* D, int64_t M, int64_t N,
                             int64_t K, float alpha_host, const float* alpha_dev, const void* bias, const void* residual, int out_dtype,
                             void* stream) {
  if (M < 0 || N < 0 || K <= 0 || (K % 64))
    return fail(ARCQ_ERR_SHAPE, "%s: need M,N >= 0 and K %% 64 == 0 (M=%lld N=%lld K=%lld)", who, (long long)M, (long long)N, (long long)K);
  if (out_dtype != ARCQ_OUT_BF16 && out_dtype != ARCQ_OUT_F32) return fail(ARCQ_ERR_SHAPE, "%s: bad out_dtype %d", who, out_dtype);
  if (M == 0 || N == 0) return ARCQ_OK;
  if (!A || !RW || !SFA || !RSF || !D) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
  if (N > INT32_MAX / 2 || K > INT32_MAX / 2) return fail(ARCQ_ERR_UNSUPPORTED, "%s: shape too large", who);
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(RW) | reinterpret_cast<uintptr_t>(D)) & 15)
    return fail(ARCQ_ERR_SHAPE, "%s: A, RW and D must be 16-byte aligned", who);
  if ((reinterpret_cast<uintptr_t>(SFA) | reinterpret_cast<uintptr_t>(RSF)) & 3) return fail(ARCQ_ERR_SHAPE, "%s: SFA and RSF must be 4-byte aligned", who);
  // the decode epilogues fetch the four bias / residual values of an output quad with ONE 8-byte load when N % 4 == 0
  if ((N % 4) == 0 && ((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) & 7))
    return fail(ARCQ_ERR_SHAPE, "%s: bias and residual must be 8-byte aligned", who);
  GemmArgs a;
  a.A = A; a.B = nullptr; a.SFA = SFA; a.SFB = nullptr; a.D = D;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.alpha_host = alpha_host; a.alpha_dev = alpha_dev; a.bias = (const uint16_t*)bias; a.residual = (const uint16_t*)residual; a.out_dtype = out_dtype;
  a.workspace = nullptr; a.workspace_bytes = 0;
  return via_stream ? gemm_repacked_stream(a, RW, RSF, (hipStream_t)stream) : gemm_repacked(a, RW, RSF, (hipStream_t)stream);
}

int arcq_gemm_nvfp4_repacked(const uint8_t* A, const uint8_t* RW, const uint8_t* SFA, const uint8_t* RSF, void* D, int64_t M, int64_t N,
                             int64_t K, float alpha_host, const float* alpha_dev, const void* bias, const void* residual, int out_dtype,
                             void* stream) {
  return gemm_repacked_entry(false, "arcq_gemm_nvfp4_repacked", A, RW, SFA, RSF, D, M, N, K, alpha_host, alpha_dev, bias, residual, out_dtype, stream);
}
int arcq_gemm_nvfp4_repacked_stream(const uint8_t* A, const uint8_t* RW, const uint8_t* SFA, const uint8_t* RSF, void* D, int64_t M, int64_t N,
                                    int64_t K, float alpha_host, const float* alpha_dev, const void* bias, const void* residual, int out_dtype,
                                    void* stream) {
  return gemm_repacked_entry(true, "arcq_gemm_nvfp4_repacked_stream", A, RW, SFA, RSF, D, M, N, K, alpha_host, alpha_dev, bias, residual, out_dtype, stream);
}

int arcq_gemm_nvfp4_repacked_silu_absmax(const uint8_t* A, const uint8_t* RW, const uint8_t* SFA, const uint8_t* RSF, void* D,
                                         uint32_t* absmax_slots, int64_t M, int64_t N, int64_t K, float alpha_host, const float* alpha_dev,
                                         void* stream) {
  const char* who = "arcq_gemm_nvfp4_repacked_silu_absmax";
  if (M < 0 || N < 0 || K <= 0 || (K % 64) || (N % 4))
    return fail(ARCQ_ERR_SHAPE, "%s: need M,N >= 0, K %% 64 == 0 and N %% 4 == 0 (M=%lld N=%lld K=%lld)", who, (long long)M, (long long)N, (long long)K);
  if (M == 0 || N == 0) return ARCQ_OK;
  if (!A || !RW || !SFA || !RSF || !D || !absmax_slots) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
  if (N > INT32_MAX / 2 || K > INT32_MAX / 2) return fail(ARCQ_ERR_UNSUPPORTED, "%s: shape too large", who);
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(RW) | reinterpret_cast<uintptr_t>(D)) & 15)
    return fail(ARCQ_ERR_SHAPE, "%s: A, RW and D must be 16-byte aligned", who);
  if ((reinterpret_cast<uintptr_t>(SFA) | reinterpret_cast<uintptr_t>(RSF) | reinterpret_cast<uintptr_t>(absmax_slots)) & 3)
    return fail(ARCQ_ERR_SHAPE, "%s: SFA, RSF and absmax_slots must be 4-byte aligned", who);
  GemmArgs a;
  a.A = A; a.B = nullptr; a.SFA = SFA; a.SFB = nullptr; a.D = D;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.alpha_host = alpha_host; a.alpha_dev = alpha_dev; a.bias = nullptr; a.residual = nullptr; a.out_dtype = ARCQ_OUT_BF16;
  a.workspace = nullptr; a.workspace_bytes = 0;
  a.epilogue = kEpiSiluMul; a.absmax_slots = absmax_slots;
  return gemm_repacked(a, RW, RSF, (hipStream_t)stream);
}

int arcq_linear_fused_supported(int kind, int64_t M, int64_t N, int64_t KQ, int64_t KE) { return gemm_fused_supported(kind, M, N, KQ, KE); }

static int fused_common_checks(const char* who, const void* X, const int16_t* idx, const uint8_t* RW, const uint8_t* RSF, void* D, int64_t M,
                               int64_t N, int64_t KQ, int64_t KE, int variant, int out_dtype) {
  if (M < 0 || N < 0 || KQ <= 0 || (KQ % 64) || (KE % 64) || KE < 0 || KE > KQ)
    return fail(ARCQ_ERR_SHAPE, "%s: need M,N >= 0, KQ%%64==0, KE%%64==0, 0<=KE<=KQ (M=%lld N=%lld KQ=%lld KE=%lld)", who, (long long)M,
                (long long)N, (long long)KQ, (long long)KE);
  if (variant != ARCQ_VARIANT_G16 && variant != ARCQ_VARIANT_G32) return fail(ARCQ_ERR_SHAPE, "%s: unknown variant %d", who, variant);
  if (out_dtype != ARCQ_OUT_BF16 && out_dtype != ARCQ_OUT_F32) return fail(ARCQ_ERR_SHAPE, "%s: bad out_dtype %d", who, out_dtype);
  if (M == 0 || N == 0) return 1;                              // nothing to do
  if (!X || !idx || !RW || !RSF || !D) return fail(ARCQ_ERR_NULL, "%s: NULL pointer", who);
  if (N > INT32_MAX / 2 || KQ > 32767) return fail(ARCQ_ERR_UNSUPPORTED, "%s: shape too large", who);
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(RW) | reinterpret_cast<uintptr_t>(D)) & 15)
    return fail(ARCQ_ERR_SHAPE, "%s: X, reorder_index, RW and D must be 16-byte aligned", who);
  if (reinterpret_cast<uintptr_t>(RSF) & 3) return fail(ARCQ_ERR_SHAPE, "%s: RSF must be 4-byte aligned", who);
  return ARCQ_OK;
}

int arcq_linear_rmsnorm_repacked(const void* X, const void* Wn, float eps, const int16_t* reorder_index, const uint8_t* RW, const uint8_t* RSF,
                                 void* D, int64_t M, int64_t N, int64_t KQ, int64_t KE, int variant, float alpha_host, const float* alpha_dev,
                                 const void* bias, const void* residual, int out_dtype, void* stream) {
  const char* who = "arcq_linear_rmsnorm_repacked";
  const int rc = fused_common_checks(who, X, reorder_index, RW, RSF, D, M, N, KQ, KE, variant, out_dtype);
  if (rc != ARCQ_OK) return rc < 0 ? rc : ARCQ_OK;
  if (!Wn || (reinterpret_cast<uintptr_t>(Wn) & 15)) return fail(Wn ? ARCQ_ERR_SHAPE : ARCQ_ERR_NULL, "%s: the norm weight must be a 16-byte aligned pointer", who);
  if ((N % 4) == 0 && ((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) & 7))
    return fail(ARCQ_ERR_SHAPE, "%s: bias and residual must be 8-byte aligned", who);
  FusedArgs f{};
  f.kind = ARCQ_SRC_RMSNORM; f.X = (const uint16_t*)X; f.Wn = (const uint16_t*)Wn; f.eps = eps; f.idx = reorder_index;
  f.RW = RW; f.RSF = RSF; f.D = D; f.M = (int)M; f.N = (int)N; f.KQ = (int)KQ; f.KE = (int)KE; f.variant = variant;
  f.alpha_host = alpha_host; f.alpha_dev = alpha_dev; f.bias = (const uint16_t*)bias; f.residual = (const uint16_t*)residual; f.out_dtype = out_dtype;
  return gemm_fused(f, (hipStream_t)stream);
}

int arcq_linear_rmsnorm_silu_repacked(const void* X, const void* Wn, float eps, const int16_t* reorder_index, const uint8_t* RW,
                                      const uint8_t* RSF, void* ACT, uint32_t* absmax_slots, int64_t M, int64_t N, int64_t KQ, int64_t KE,
                                      int variant, float alpha_host, const float* alpha_dev, const void* bias, const int16_t* act_scatter_index,
                                      void* stream) {
  const char* who = "arcq_linear_rmsnorm_silu_repacked";
  const int rc = fused_common_checks(who, X, reorder_index, RW, RSF, ACT, M, N, KQ, KE, variant, ARCQ_OUT_BF16);
  if (rc != ARCQ_OK) return rc < 0 ? rc : ARCQ_OK;
  if (N % 4) return fail(ARCQ_ERR_SHAPE, "%s: N %% 4 != 0 (interleaved gate|up rows)", who);
  if (!Wn || !absmax_slots) return fail(ARCQ_ERR_NULL, "%s: NULL norm weight / absmax_slots", who);
  if ((reinterpret_cast<uintptr_t>(Wn) & 15) || (reinterpret_cast<uintptr_t>(absmax_slots) & 3)) return fail(ARCQ_ERR_SHAPE, "%s: misaligned norm weight / absmax_slots", who);
  if (reinterpret_cast<uintptr_t>(act_scatter_index) & 3) return fail(ARCQ_ERR_SHAPE, "%s: misaligned act_scatter_index", who);
  if (reinterpret_cast<uintptr_t>(bias) & 7) return fail(ARCQ_ERR_SHAPE, "%s: bias must be 8-byte aligned", who);
  FusedArgs f{};
  f.act_scatter = act_scatter_index;
  f.kind = ARCQ_SRC_RMSNORM; f.silu_act = 1; f.X = (const uint16_t*)X; f.Wn = (const uint16_t*)Wn; f.eps = eps; f.idx = reorder_index;
  f.RW = RW; f.RSF = RSF; f.D = ACT; f.out_slots = absmax_slots; f.M = (int)M; f.N = (int)N; f.KQ = (int)KQ; f.KE = (int)KE; f.variant = variant;
  f.alpha_host = alpha_host; f.alpha_dev = alpha_dev; f.bias = (const uint16_t*)bias; f.out_dtype = ARCQ_OUT_BF16;
  return gemm_fused(f, (hipStream_t)stream);
}

int arcq_linear_dynamic_repacked(const void* X, const int16_t* reorder_index, const uint8_t* RW, const uint8_t* RSF, void* D, float* scale_out,
                                 const uint32_t* absmax_slots, int64_t nslots, int64_t M, int64_t N, int64_t KQ, int64_t KE, int variant,
                                 float alpha_host, const void* bias, const void* residual, int out_dtype, void* stream) {
  const char* who = "arcq_linear_dynamic_repacked";
  const int rc = fused_common_checks(who, X, reorder_index, RW, RSF, D, M, N, KQ, KE, variant, out_dtype);
  if (rc != ARCQ_OK) return rc < 0 ? rc : ARCQ_OK;
  if (absmax_slots && (nslots <= 0 || nslots > INT32_MAX || (reinterpret_cast<uintptr_t>(absmax_slots) & 3)))
    return fail(ARCQ_ERR_SHAPE, "%s: absmax_slots given but nslots = %lld, or misaligned", who, (long long)nslots);
  if ((N % 4) == 0 && ((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) & 7))
    return fail(ARCQ_ERR_SHAPE, "%s: bias and residual must be 8-byte aligned", who);
  FusedArgs f{};
  f.kind = ARCQ_SRC_DYNAMIC; f.X = (const uint16_t*)X; f.idx = reorder_index; f.in_slots = absmax_slots; f.n_in_slots = absmax_slots ? (int)nslots : 0;
  f.scale_out = scale_out; f.RW = RW; f.RSF = RSF; f.D = D; f.M = (int)M; f.N = (int)N; f.KQ = (int)KQ; f.KE = (int)KE; f.variant = variant;
  f.alpha_host = alpha_host; f.bias = (const uint16_t*)bias; f.residual = (const uint16_t*)residual; f.out_dtype = out_dtype;
  return gemm_fused(f, (hipStream_t)stream);
}

int arcq_silu_mul_quantize_x_dyn_slots(const void* GU, const int16_t* reorder_index, uint8_t* QX, uint8_t* SFX, float* scale_out,
                                       const uint32_t* absmax_slots, int64_t nslots, int64_t M, int64_t KQ, int64_t KE, int variant,
                                       int layout, void* stream) {
  return silu_mul_quantize_x_dyn_slots(GU, reorder_index, QX, SFX, scale_out, absmax_slots, nslots, M, KQ, KE, variant, layout, (hipStream_t)stream);
}

}  // extern "C"
