// The reference's boundary in the reference's own form: a CPython extension module named `agemm` (pybind11 + libtorch) with the
// four hot-path functions of kernels/src/bindings.cpp:551-575 -- same names, keyword names, argument meaning, return shapes and
// dtypes, RuntimeError on an unsupported shape -- and the six KV-cache names as stubs (bindings.cpp:576-581, out of scope).  Every
// function body is argument checking + one call into the C-ABI of include/arcq.h (libarcq_hip.so, hand-written gfx950 kernels);
// torch is used for device memory and the current HIP stream only.  Drop-in:
//     sys.path.append("<repo>/arcquant_amd/lib"); import agemm          # instead of kernels/build/ (model/qLinearLayer.py:7-8)
// The ctypes mirror arcquant_amd/agemm.py has the same surface plus the extensions; this module exists because a ctypes call
// costs 9.5-13 us of host time (18 marshalled arguments) and an eager decode step makes ~170 of them: here a call is ~3 us.
//
// Differences from the reference a caller can observe (deviations D2-D4): launches go to torch's CURRENT stream; `scale`
// may be a 0-dim CUDA fp32 tensor and is then read on the device (the reference's `const float scale` forces `.item()`); KQ is not
// limited to the reference's template list.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // torch-ROCm tensors say "cuda": the guard / stream types that accept it
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <stdexcept>
#include <string>
#include <tuple>

#include "../../include/arcq.h"

namespace {

void need(const torch::Tensor& t, c10::ScalarType dt, const char* name, int64_t ndim = -1) {
  if (t.scalar_type() != dt) throw std::runtime_error(std::string("agemm: ") + name + " has the wrong dtype");   // reference: data_ptr<T>() throws
  if (!t.is_cuda()) throw std::runtime_error(std::string("agemm: ") + name + " must live on the GPU (there is no CPU path)");
  if (ndim >= 0 && t.dim() != ndim) throw std::runtime_error(std::string("agemm: ") + name + " has the wrong rank");
  if (!t.is_contiguous()) throw std::runtime_error(std::string("agemm: ") + name + " must be contiguous");
}

void check(int status, const char* what) {
  if (status != ARCQ_OK) throw std::runtime_error(std::string(what) + ": " + arcq_last_error());
}

void* stream_of(const torch::Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

// agemm.matmul(A, B, SFA, SFB, scale) -> bf16 [M, N]   (bindings.cpp:99-120)
torch::Tensor matmul(const torch::Tensor& A, const torch::Tensor& B, const torch::Tensor& SFA, const torch::Tensor& SFB, const py::object& scale) {
  need(A, torch::kUInt8, "A", 2);
  need(B, torch::kUInt8, "B", 2);
  need(SFA, torch::kUInt8, "SFA");
  need(SFB, torch::kUInt8, "SFB");
  const int64_t M = A.size(0), N = B.size(0), K = A.size(1) * 2;      // bindings.cpp:107-109
  if (B.size(1) * 2 != K) throw std::runtime_error("agemm.matmul: A and B disagree on K");
  if (SFA.numel() < arcq_sf_used_bytes(M, K) || SFB.numel() < arcq_sf_used_bytes(N, K))
    throw std::runtime_error("agemm.matmul: scale-factor buffer smaller than the swizzled layout of its operand");
  float alpha_host = 1.0f;
  const float* alpha_dev = nullptr;
  torch::Tensor keep;
  if (THPVariable_Check(scale.ptr())) {
    const torch::Tensor& s = THPVariable_Unpack(scale.ptr());
    if (s.is_cuda() && s.scalar_type() == torch::kFloat32 && s.numel() == 1) {
      keep = s;
      alpha_dev = s.data_ptr<float>();                    // consumed on the device: no .item() sync
    } else {
      alpha_host = s.item<float>();                       // the reference's implicit __float__
    }
  } else {
    alpha_host = scale.cast<float>();
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(A.device());
  auto D = torch::empty({M, N}, A.options().dtype(torch::kBFloat16));
  const int64_t ws_bytes = arcq_gemm_workspace_bytes(M, N, K);
  torch::Tensor ws;
  if (ws_bytes) ws = torch::empty({ws_bytes}, A.options());
  check(arcq_gemm_nvfp4(A.data_ptr<uint8_t>(), B.data_ptr<uint8_t>(), SFA.data_ptr<uint8_t>(), SFB.data_ptr<uint8_t>(), D.data_ptr(), M, N, K, alpha_host,
                        alpha_dev, nullptr, nullptr, ARCQ_OUT_BF16, ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream_of(A)),
        "matmul");
  return D;
}

std::tuple<torch::Tensor, torch::Tensor> quantize(bool is_x, const torch::Tensor& X, const torch::Tensor& reorder_index, int64_t KE) {
  const char* who = is_x ? "reorder_quantize_x" : "reorder_quantize_w";
  need(X, torch::kBFloat16, is_x ? "X" : "W", 2);
  need(reorder_index, torch::kInt16, "reorder_index", 1);
  const int64_t rows = X.size(0), KQ = X.size(1), K = KQ + KE;
  if (reorder_index.numel() != KQ || KQ % 64 || KE % 64 || KE < 0 || KE > KQ)
    throw std::runtime_error(std::string("Value error in ") + who + ": KQ / KE / reorder_index are not valid");       // bindings.cpp:157-160
  c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
  auto Q = torch::empty({rows, K / 2}, X.options().dtype(torch::kUInt8));
  auto SF = torch::empty({arcq_sf_alloc_bytes(rows, K)}, X.options().dtype(torch::kUInt8));      // bindings.cpp:83-95
  const int variant = arcq_variant_for_kq(KQ);
  auto fn = is_x ? arcq_quantize_x : arcq_quantize_w;
  check(fn(X.data_ptr(), reorder_index.data_ptr<int16_t>(), Q.data_ptr<uint8_t>(), SF.data_ptr<uint8_t>(), rows, KQ, KE, variant, stream_of(X)), who);
  return {Q, SF};
}

std::tuple<torch::Tensor, torch::Tensor> reorder_quantize_x(const torch::Tensor& X, const torch::Tensor& reorder_index, int64_t KE) {
  return quantize(true, X, reorder_index, KE);           // bindings.cpp:122-163
}
std::tuple<torch::Tensor, torch::Tensor> reorder_quantize_w(const torch::Tensor& W, const torch::Tensor& reorder_index, int64_t KE) {
  return quantize(false, W, reorder_index, KE);          // bindings.cpp:170-210
}

// agemm.rmsnorm_quantize_x(X, W, eps, reorder_index, KE)   (bindings.cpp:216-254)
std::tuple<torch::Tensor, torch::Tensor> rmsnorm_quantize_x(const torch::Tensor& X, const torch::Tensor& W, double eps, const torch::Tensor& reorder_index,
                                                            int64_t KE) {
  need(X, torch::kBFloat16, "X", 2);
  need(W, torch::kBFloat16, "W", 1);
  need(reorder_index, torch::kInt16, "reorder_index", 1);
  const int64_t M = X.size(0), KQ = X.size(1), K = KQ + KE;
  if (W.numel() != KQ || reorder_index.numel() != KQ || KQ % 64 || KE % 64 || KE < 0 || KE > KQ || KQ < 2048 || KQ > 8192)
    throw std::runtime_error("Value error in run_rmsnorm_x_bf16_nvfp4: K value is not valid: " + std::to_string(KQ));   // bindings.cpp:248-251
  c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
  auto Q = torch::empty({M, K / 2}, X.options().dtype(torch::kUInt8));
  auto SF = torch::empty({arcq_sf_alloc_bytes(M, K)}, X.options().dtype(torch::kUInt8));
  check(arcq_rmsnorm_quantize_x(X.data_ptr(), W.data_ptr(), (float)eps, reorder_index.data_ptr<int16_t>(), Q.data_ptr<uint8_t>(), SF.data_ptr<uint8_t>(), M, KQ,
                                KE, arcq_variant_for_kq(KQ), stream_of(X)),
        "rmsnorm_quantize_x");
  return {Q, SF};
}

py::object kv_stub(const char* name) {
  return py::cpp_function([name](py::args, py::kwargs) -> py::object {
    PyErr_SetString(PyExc_NotImplementedError, (std::string("agemm.") + name + ": the int4 paged-KV attention is outside the ARC-NVFP4 GEMM hot path").c_str());
    throw py::error_already_set();
  });
}

}  // namespace

PYBIND11_MODULE(agemm, m) {
  m.doc() = "ARC-NVFP4 hot path on MI355X (gfx950): drop-in for the reference's pybind11 module (kernels/src/bindings.cpp:551-575)";
  m.def("matmul", &matmul, py::arg("A"), py::arg("B"), py::arg("SFA"), py::arg("SFB"), py::arg("scale"));
  m.def("reorder_quantize_x", &reorder_quantize_x, py::arg("X"), py::arg("reorder_index"), py::arg("KE"));
  m.def("reorder_quantize_w", &reorder_quantize_w, py::arg("W"), py::arg("reorder_index"), py::arg("KE"));
  m.def("rmsnorm_quantize_x", &rmsnorm_quantize_x, py::arg("X"), py::arg("W"), py::arg("eps"), py::arg("reorder_index"), py::arg("KE"));
  for (const char* n : {"batch_decode_i4", "init_kv_i4", "append_kv_i4", "batch_decode_f16", "init_kv_f16", "append_kv_f16"}) m.attr(n) = kv_stub(n);
  m.attr("abi_version") = arcq_abi_version();
}
