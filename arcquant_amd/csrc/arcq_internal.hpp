// Internal declarations shared by the translation units of libarcq_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/arcq.h"

namespace arcq {

// Records a formatted message for arcq_last_error() (thread-local) and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Opt-in to more than 48 KB of dynamic LDS for `kernel` on the CURRENT device.  The attribute is per device and per
// kernel, and the driver call costs host time, so each launcher keeps one LdsOptIn per kernel instantiation: a small
// per-device table of the largest size already granted (relaxed atomics: two threads racing both make the same,
// idempotent driver call).  Returns ARCQ_OK or ARCQ_ERR_LAUNCH (message recorded).
constexpr int kMaxDevices = 64;
struct LdsOptIn {
  int granted[kMaxDevices];     // zero-initialised (static storage)
};
int ensure_dynamic_lds(const void* kernel, LdsOptIn& cache, int bytes, const char* who);

// quantize.hip
int quantize_x(const void* X, const int16_t* idx, uint8_t* QX, uint8_t* SFX, int64_t M, int64_t KQ, int64_t KE, int variant,
               hipStream_t stream);
int quantize_w(const void* W, const int16_t* idx, uint8_t* QW, uint8_t* SFW, int64_t N, int64_t KQ, int64_t KE, int variant,
               hipStream_t stream);
int rmsnorm_quantize_x(const void* X, const void* Wn, float eps, const int16_t* idx, uint8_t* QX, uint8_t* SFX, int64_t M,
                       int64_t KQ, int64_t KE, int variant, hipStream_t stream);
int absmax_scale(const void* X, int64_t n, float* scale_out, hipStream_t stream);
int quantize_x_dyn(const void* X, const int16_t* idx, uint8_t* QX, uint8_t* SFX, float* scale_out, void* state, int64_t M,
                   int64_t KQ, int64_t KE, int variant, hipStream_t stream);
int quantize_x_dyn_slots(const void* X, const int16_t* idx, uint8_t* QX, uint8_t* SFX, float* scale_out, const uint32_t* slots,
                         int64_t nslots, int64_t M, int64_t KQ, int64_t KE, int variant, hipStream_t stream);
int silu_mul_quantize_x_dyn_slots(const void* GU, const int16_t* idx, uint8_t* QX, uint8_t* SFX, float* scale_out, const uint32_t* slots,
                                  int64_t nslots, int64_t M, int64_t KQ, int64_t KE, int variant, int layout, hipStream_t stream);
int silu_mul_quantize_x_dyn(const void* GU, const int16_t* idx, uint8_t* QX, uint8_t* SFX, float* scale_out, void* state, int64_t M,
                            int64_t KQ, int64_t KE, int variant, int layout, hipStream_t stream);

// gemm_skinny.hip / gemm_tile.hip
struct GemmArgs {
  const uint8_t* A;     // [M, K/2]
  const uint8_t* B;     // [N, K/2]
  const uint8_t* SFA;   // swizzled ue4m3
  const uint8_t* SFB;
  void* D;              // [M, N] bf16 or fp32
  int M, N, K;
  float alpha_host;
  const float* alpha_dev;   // optional device scalar multiplied into alpha
  const uint16_t* bias;     // optional bf16 [N]
  const uint16_t* residual; // optional bf16 [M, N], added after the bf16 rounding of alpha*acc (+bias)
  int out_dtype;
  void* workspace;
  int64_t workspace_bytes;
  int epilogue = 0;                  // kEpiPlain, or kEpiSiluMul: D = bf16 [M, N/2] silu(gate)*up of interleaved weight rows
  unsigned int* absmax_slots = nullptr;   // kEpiSiluMul: one max|D| word per workgroup / tile (see gemm_silu_slots)
};
enum : int { kEpiPlain = 0, kEpiSiluMul = 1 };
int64_t gemm_silu_slots(int64_t M, int64_t N, int64_t K);   // slots the silu-mul epilogue of this shape writes
// gemm_stream.hip: persistent decode GEMM over a weight repacked into MFMA-operand-order tiles (see arcq.h), optionally with
// the activation quantiser as its prologue
int64_t gemm_repacked_w_bytes(int64_t N, int64_t K);
int64_t gemm_repacked_sf_bytes(int64_t N, int64_t K);
int gemm_repacked_supported(int64_t M, int64_t N, int64_t K);
int gemm_repacked(const GemmArgs& a, const uint8_t* RW, const uint8_t* RSF, hipStream_t stream);          // gemm_rowblock.hip (M <= 16), gemm_rowmid.hip above
int gemm_repacked_mid_supported(int64_t M, int64_t N, int64_t K);                                          // gemm_rowmid.hip: 16 < M <= 64
int gemm_repacked_mid(const GemmArgs& a, const uint8_t* RW, const uint8_t* RSF, hipStream_t stream);
int gemm_repacked_stream(const GemmArgs& a, const uint8_t* RW, const uint8_t* RSF, hipStream_t stream);   // gemm_stream.hip (A-B)
struct FusedArgs {
  int kind;                   // ARCQ_SRC_RMSNORM | ARCQ_SRC_DYNAMIC
  int silu_act;               // ARCQ_SRC_RMSNORM only: D = bf16 silu(gate) * up [M, N/2] of interleaved gate|up rows + out_slots
  const uint16_t* X;          // bf16 [M, KQ]
  const uint16_t* Wn;         // rmsnorm weight bf16 [KQ]
  float eps;
  const int16_t* idx;
  const uint32_t* in_slots;   // ARCQ_SRC_DYNAMIC: abs-max words of X or NULL
  int n_in_slots;
  float* scale_out;           // ARCQ_SRC_DYNAMIC: max|X| / 2688
  const uint8_t* RW;
  const uint8_t* RSF;
  void* D;
  uint32_t* out_slots;
  const int16_t* act_scatter; // silu_act: ACT[m][act_scatter[j]] = activation j (NULL: ACT[m][j]); a permutation of 0 .. N/2-1
  int M, N, KQ, KE, variant;
  float alpha_host;
  const float* alpha_dev;
  const uint16_t* bias;
  const uint16_t* residual;
  int out_dtype;
};
int gemm_fused_supported(int kind, int64_t M, int64_t N, int64_t KQ, int64_t KE);
int gemm_fused(const FusedArgs& f, hipStream_t stream);
int64_t gemm_skinny_workspace_bytes(int64_t M, int64_t N, int64_t K);
int64_t gemm_tile_workspace_bytes(int64_t M, int64_t N, int64_t K);
int64_t gemm_tile_silu_slots(int64_t M, int64_t N, int64_t K);
int64_t gemm_decode_silu_slots(int64_t M, int64_t N, int64_t K);
// D = epilogue(sum_s partial[s]) over the `splitk` fp32 planes [M, N] at a.workspace, summed in a fixed order
int gemm_splitk_finish(const GemmArgs& a, int splitk, hipStream_t stream);
int gemm_skinny(const GemmArgs& a, hipStream_t stream);   // M <= 16, few tiles: weight-streaming MFMA GEMV, 16-row tiles
int64_t gemm_decode_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gemm_decode(const GemmArgs& a, hipStream_t stream);   // M <= 16, many tiles: 32-row tiles, two units per thread and item
int gemm_tile(const GemmArgs& a, hipStream_t stream);     // general M: LDS-tiled MFMA GEMM
// gemm_regtile.hip: 16 < M <~ 1024, grids the tiled kernel cannot fill: operand fragments straight into registers, K split over waves
int gemm_regtile_cfg(int64_t M, int64_t N, int64_t K, int epilogue);   // 0 = not this kernel, else its configuration
int gemm_regtile(const GemmArgs& a, int cfg, hipStream_t stream);

}  // namespace arcq
