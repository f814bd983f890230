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

// ---- the decode extensions of arcquant_amd/agemm.py under the same names and keywords: an eager decode step of
//      the model harness makes ~170 calls, at the ctypes mirror's 8-12 us each it is host-paced
struct Alpha {
  float host = 1.0f;
  const float* dev = nullptr;
  torch::Tensor keep;
};
Alpha alpha_of(const py::object& scale, double scale_host) {
  Alpha a;
  a.host = (float)scale_host;
  if (THPVariable_Check(scale.ptr())) {
    const torch::Tensor& t = THPVariable_Unpack(scale.ptr());
    if (t.is_cuda() && t.scalar_type() == torch::kFloat32 && t.numel() == 1) {
      a.keep = t;
      a.dev = t.data_ptr<float>();
    } else {
      a.host *= t.item<float>();
    }
  } else {
    a.host *= scale.cast<float>();
  }
  return a;
}
const void* opt_ptr(const c10::optional<torch::Tensor>& t, c10::ScalarType dt, const char* name, std::initializer_list<int64_t> shape) {
  if (!t.has_value()) return nullptr;
  need(*t, dt, name, (int64_t)shape.size());
  int i = 0;
  for (int64_t d : shape)
    if (t->size(i++) != d) throw std::runtime_error(std::string("agemm: ") + name + " has the wrong shape");
  return t->data_ptr();
}
torch::Tensor out_of(const c10::optional<torch::Tensor>& out, int64_t M, int64_t N, c10::ScalarType dt, const torch::Tensor& like, const char* who) {
  if (!out.has_value()) return torch::empty({M, N}, like.options().dtype(dt));
  if (out->dim() != 2 || out->size(0) != M || out->size(1) != N || out->scalar_type() != dt || !out->is_contiguous())
    throw std::runtime_error(std::string("agemm.") + who + ": out has the wrong shape / dtype");
  return *out;
}
int out_code(c10::ScalarType dt, const char* who) {
  if (dt == torch::kBFloat16) return ARCQ_OUT_BF16;
  if (dt == torch::kFloat32) return ARCQ_OUT_F32;
  throw std::runtime_error(std::string("agemm.") + who + ": out_dtype must be bfloat16 or float32");
}
void need_repacked(const torch::Tensor& RW, const torch::Tensor& RSF, int64_t N, int64_t K, const char* who) {
  need(RW, torch::kUInt8, "RW", 1);
  need(RSF, torch::kUInt8, "RSF", 1);
  if (K % 64 || RW.numel() != arcq_repacked_w_bytes(N, K) || RSF.numel() != arcq_repacked_sf_bytes(N, K))
    throw std::runtime_error(std::string("Value error in ") + who + ": RW / RSF do not belong to a [N, K] weight of this shape");
}

torch::Tensor matmul_repacked(const torch::Tensor& A, const torch::Tensor& RW, const torch::Tensor& SFA, const torch::Tensor& RSF, const py::object& scale,
                              int64_t N, const c10::optional<torch::Tensor>& bias, const c10::optional<torch::Tensor>& residual, py::object out_dtype,
                              const c10::optional<torch::Tensor>& out, double scale_host) {
  need(A, torch::kUInt8, "A", 2);
  need(SFA, torch::kUInt8, "SFA");
  const int64_t M = A.size(0), K = A.size(1) * 2;
  need_repacked(RW, RSF, N, K, "matmul_repacked");
  if (SFA.numel() < arcq_sf_used_bytes(M, K)) throw std::runtime_error("Value error in matmul_repacked: SFA smaller than the swizzled layout of A");
  if (!arcq_gemm_repacked_supported(M, N, K)) throw std::runtime_error("matmul_repacked: this shape is outside the repacked path (see repacked_supported)");
  const auto dt = out_dtype.is_none() ? torch::kBFloat16 : torch::python::detail::py_object_to_dtype(out_dtype);
  const int oc = out_code(dt, "matmul_repacked");
  const Alpha al = alpha_of(scale, scale_host);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(A.device());
  torch::Tensor D = out_of(out, M, N, dt, A, "matmul_repacked");
  check(arcq_gemm_nvfp4_repacked(A.data_ptr<uint8_t>(), RW.data_ptr<uint8_t>(), SFA.data_ptr<uint8_t>(), RSF.data_ptr<uint8_t>(), D.data_ptr(), M, N, K, al.host,
                                 al.dev, opt_ptr(bias, torch::kBFloat16, "bias", {N}), opt_ptr(residual, torch::kBFloat16, "residual", {M, N}), oc, stream_of(A)),
        "matmul_repacked");
  return D;
}

struct FusedShape {
  int64_t M, KQ, KE, K;
  int variant;
};
FusedShape fused_common(const char* who, const torch::Tensor& X, const torch::Tensor& reorder_index, const torch::Tensor& RW, const torch::Tensor& RSF, int64_t N,
                        int64_t KE, const py::object& variant) {
  need(X, torch::kBFloat16, "X", 2);
  need(reorder_index, torch::kInt16, "reorder_index", 1);
  FusedShape f{X.size(0), X.size(1), KE, X.size(1) + KE, 0};
  if (f.KQ % 64 || KE % 64 || KE < 0 || KE > f.KQ || reorder_index.numel() != f.KQ) throw std::runtime_error(std::string("Value error in ") + who + ": KQ / KE are not valid");
  need_repacked(RW, RSF, N, f.K, who);
  f.variant = variant.is_none() ? arcq_variant_for_kq(f.KQ) : variant.cast<int>();
  return f;
}

torch::Tensor rmsnorm_matmul_repacked(const torch::Tensor& X, const torch::Tensor& W, double eps, const torch::Tensor& reorder_index, int64_t KE,
                                      const torch::Tensor& RW, const torch::Tensor& RSF, const py::object& scale, int64_t N,
                                      const c10::optional<torch::Tensor>& bias, const c10::optional<torch::Tensor>& residual, py::object out_dtype,
                                      const c10::optional<torch::Tensor>& out, double scale_host, const py::object& variant) {
  const char* who = "rmsnorm_matmul_repacked";
  const FusedShape f = fused_common(who, X, reorder_index, RW, RSF, N, KE, variant);
  need(W, torch::kBFloat16, "W", 1);
  if (W.numel() != f.KQ || !arcq_linear_fused_supported(ARCQ_SRC_RMSNORM, f.M, N, f.KQ, KE)) throw std::runtime_error("rmsnorm_matmul_repacked: outside the fused path (see fused_supported)");
  const auto dt = out_dtype.is_none() ? torch::kBFloat16 : torch::python::detail::py_object_to_dtype(out_dtype);
  const int oc = out_code(dt, who);
  const Alpha al = alpha_of(scale, scale_host);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
  torch::Tensor D = out_of(out, f.M, N, dt, X, who);
  check(arcq_linear_rmsnorm_repacked(X.data_ptr(), W.data_ptr(), (float)eps, reorder_index.data_ptr<int16_t>(), RW.data_ptr<uint8_t>(), RSF.data_ptr<uint8_t>(), D.data_ptr(),
                                     f.M, N, f.KQ, KE, f.variant, al.host, al.dev, opt_ptr(bias, torch::kBFloat16, "bias", {N}),
                                     opt_ptr(residual, torch::kBFloat16, "residual", {f.M, N}), oc, stream_of(X)),
        who);
  return D;
}

std::tuple<torch::Tensor, torch::Tensor> rmsnorm_matmul_repacked_silu(const torch::Tensor& X, const torch::Tensor& W, double eps, const torch::Tensor& reorder_index,
                                                                      int64_t KE, const torch::Tensor& RW, const torch::Tensor& RSF, const py::object& scale, int64_t N,
                                                                      double scale_host, const py::object& variant, const c10::optional<torch::Tensor>& bias,
                                                                      const c10::optional<torch::Tensor>& act_scatter_index) {
  const char* who = "rmsnorm_matmul_repacked_silu";
  const FusedShape f = fused_common(who, X, reorder_index, RW, RSF, N, KE, variant);
  need(W, torch::kBFloat16, "W", 1);
  if (W.numel() != f.KQ || N % 4 || !arcq_linear_fused_supported(ARCQ_SRC_RMSNORM, f.M, N, f.KQ, KE))
    throw std::runtime_error("rmsnorm_matmul_repacked_silu: outside the fused path (see fused_supported; N % 4 == 0)");
  const Alpha al = alpha_of(scale, scale_host);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
  auto act = torch::empty({f.M, N / 2}, X.options());
  auto slots = torch::empty({(N + 15) / 16}, X.options().dtype(torch::kInt32));
  // (act_scatter_index must be a permutation of 0 .. N/2-1: the ctypes mirror checks it once per index tensor; callers of this binding
  //  pass an index that went through that check or through their own)
  check(arcq_linear_rmsnorm_silu_repacked(X.data_ptr(), W.data_ptr(), (float)eps, reorder_index.data_ptr<int16_t>(), RW.data_ptr<uint8_t>(), RSF.data_ptr<uint8_t>(),
                                          act.data_ptr(), (uint32_t*)slots.data_ptr<int32_t>(), f.M, N, f.KQ, KE, f.variant, al.host, al.dev,
                                          opt_ptr(bias, torch::kBFloat16, "bias", {N}),
                                          (const int16_t*)opt_ptr(act_scatter_index, torch::kInt16, "act_scatter_index", {N / 2}), stream_of(X)),
        who);
  return {act, slots};
}

std::tuple<torch::Tensor, torch::Tensor> dynamic_matmul_repacked(const torch::Tensor& X, const torch::Tensor& reorder_index, int64_t KE, const torch::Tensor& RW,
                                                                 const torch::Tensor& RSF, double scale_w, int64_t N, const c10::optional<torch::Tensor>& absmax_slots,
                                                                 const c10::optional<torch::Tensor>& bias, const c10::optional<torch::Tensor>& residual,
                                                                 py::object out_dtype, const c10::optional<torch::Tensor>& out, const py::object& variant) {
  const char* who = "dynamic_matmul_repacked";
  const FusedShape f = fused_common(who, X, reorder_index, RW, RSF, N, KE, variant);
  if (!arcq_linear_fused_supported(ARCQ_SRC_DYNAMIC, f.M, N, f.KQ, KE)) throw std::runtime_error("dynamic_matmul_repacked: outside the fused path (see fused_supported)");
  const auto dt = out_dtype.is_none() ? torch::kBFloat16 : torch::python::detail::py_object_to_dtype(out_dtype);
  const int oc = out_code(dt, who);
  const uint32_t* sl = nullptr;
  int64_t nsl = 0;
  if (absmax_slots.has_value()) {
    need(*absmax_slots, torch::kInt32, "absmax_slots", 1);
    if (absmax_slots->numel() == 0) throw std::runtime_error("agemm.dynamic_matmul_repacked: absmax_slots must not be empty");
    sl = (const uint32_t*)absmax_slots->data_ptr<int32_t>();
    nsl = absmax_slots->numel();
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
  torch::Tensor D = out_of(out, f.M, N, dt, X, who);
  auto scale = torch::empty({1}, X.options().dtype(torch::kFloat32));
  check(arcq_linear_dynamic_repacked(X.data_ptr(), reorder_index.data_ptr<int16_t>(), RW.data_ptr<uint8_t>(), RSF.data_ptr<uint8_t>(), D.data_ptr(), scale.data_ptr<float>(), sl,
                                     nsl, f.M, N, f.KQ, KE, f.variant, (float)scale_w, opt_ptr(bias, torch::kBFloat16, "bias", {N}),
                                     opt_ptr(residual, torch::kBFloat16, "residual", {f.M, N}), oc, stream_of(X)),
        who);
  return {D, scale.reshape({})};
}

// reorder_quantize_x_dynamic with the abs-max words of the producing kernel (one launch); reorder_index = None: X is already in reordered order
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> reorder_quantize_x_dynamic(const torch::Tensor& X, const c10::optional<torch::Tensor>& reorder_index, int64_t KE,
                                                                                 const py::object& variant, const torch::Tensor& absmax_slots) {
  need(X, torch::kBFloat16, "X", 2);
  need(absmax_slots, torch::kInt32, "absmax_slots", 1);
  const int64_t M = X.size(0), KQ = X.size(1), K = KQ + KE;
  if (reorder_index.has_value()) need(*reorder_index, torch::kInt16, "reorder_index", 1);
  if (KQ % 64 || KE % 64 || KE < 0 || KE > KQ || (reorder_index.has_value() && reorder_index->numel() != KQ) || absmax_slots.numel() == 0)
    throw std::runtime_error("Value error in reorder_quantize_x_dynamic: KQ / KE / reorder_index / absmax_slots are not valid");
  const int var = variant.is_none() ? arcq_variant_for_kq(KQ) : variant.cast<int>();
  c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
  auto Q = torch::empty({M, K / 2}, X.options().dtype(torch::kUInt8));
  auto SF = torch::empty({arcq_sf_alloc_bytes(M, K)}, X.options().dtype(torch::kUInt8));
  auto scale = torch::empty({1}, X.options().dtype(torch::kFloat32));
  check(arcq_quantize_x_dyn_slots(X.data_ptr(), reorder_index.has_value() ? reorder_index->data_ptr<int16_t>() : nullptr, Q.data_ptr<uint8_t>(), SF.data_ptr<uint8_t>(),
                                  scale.data_ptr<float>(), (const uint32_t*)absmax_slots.data_ptr<int32_t>(), absmax_slots.numel(), M, KQ, KE, var, stream_of(X)),
        "reorder_quantize_x_dynamic");
  return {Q, SF, scale.reshape({})};
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
  // decode extensions (arcquant_amd/agemm.py has the same functions through ctypes, and more)
  m.def("repacked_supported", [](int64_t M, int64_t N, int64_t K) { return arcq_gemm_repacked_supported(M, N, K) != 0; });
  m.def("fused_supported", [](int kind, int64_t M, int64_t N, int64_t KQ, int64_t KE) { return arcq_linear_fused_supported(kind, M, N, KQ, KE) != 0; });
  m.attr("SRC_RMSNORM") = ARCQ_SRC_RMSNORM;
  m.attr("SRC_DYNAMIC") = ARCQ_SRC_DYNAMIC;
  m.def("matmul_repacked", &matmul_repacked, py::arg("A"), py::arg("RW"), py::arg("SFA"), py::arg("RSF"), py::arg("scale"), py::arg("N"), py::kw_only(),
        py::arg("bias") = py::none(), py::arg("residual") = py::none(), py::arg("out_dtype") = py::none(), py::arg("out") = py::none(), py::arg("scale_host") = 1.0);
  m.def("rmsnorm_matmul_repacked", &rmsnorm_matmul_repacked, py::arg("X"), py::arg("W"), py::arg("eps"), py::arg("reorder_index"), py::arg("KE"), py::arg("RW"),
        py::arg("RSF"), py::arg("scale"), py::arg("N"), py::kw_only(), py::arg("bias") = py::none(), py::arg("residual") = py::none(),
        py::arg("out_dtype") = py::none(), py::arg("out") = py::none(), py::arg("scale_host") = 1.0, py::arg("variant") = py::none());
  m.def("rmsnorm_matmul_repacked_silu", &rmsnorm_matmul_repacked_silu, py::arg("X"), py::arg("W"), py::arg("eps"), py::arg("reorder_index"), py::arg("KE"),
        py::arg("RW"), py::arg("RSF"), py::arg("scale"), py::arg("N"), py::kw_only(), py::arg("scale_host") = 1.0, py::arg("variant") = py::none(),
        py::arg("bias") = py::none(), py::arg("act_scatter_index") = py::none());
  m.def("dynamic_matmul_repacked", &dynamic_matmul_repacked, py::arg("X"), py::arg("reorder_index"), py::arg("KE"), py::arg("RW"), py::arg("RSF"), py::arg("scale_w"),
        py::arg("N"), py::kw_only(), py::arg("absmax_slots") = py::none(), py::arg("bias") = py::none(), py::arg("residual") = py::none(),
        py::arg("out_dtype") = py::none(), py::arg("out") = py::none(), py::arg("variant") = py::none());
  m.def("reorder_quantize_x_dynamic", &reorder_quantize_x_dynamic, py::arg("X"), py::arg("reorder_index"), py::arg("KE"), py::arg("variant") = py::none(),
        py::kw_only(), py::arg("absmax_slots"));
  for (const char* n : {"batch_decode_i4", "init_kv_i4", "append_kv_i4", "batch_decode_f16", "init_kv_f16", "append_kv_f16"}) m.attr(n) = kv_stub(n);
  m.attr("abi_version") = arcq_abi_version();
}
