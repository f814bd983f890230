"""Drop-in mirror of the reference's pybind11 module ``agemm`` (kernels/src/bindings.cpp:551-575).

Same function names, argument names, argument meaning, return shapes/dtypes and error behaviour
(``RuntimeError`` on an unsupported shape) as the reference, but every call goes through the C-ABI of
``libarcq_hip.so`` (hand-written gfx950 kernels).  ``torch`` is used only for device memory and the
current HIP stream.

Differences a caller can observe, all deliberate (DESIGN.md "deviations"):
  * launches go to torch's CURRENT stream (the reference uses the legacy default stream), so the ops are
    capturable in HIP graphs and ordered with surrounding torch work without device-wide syncs;
  * KQ is not limited to the reference's closed template list (bindings.cpp:141-160);
  * ``matmul(..., scale)`` also accepts a 0-dim device tensor WITHOUT a device->host sync;
  * ``rmsnorm_quantize_x`` uses the same augmented-K layout as the weights for KQ=3584 (the reference
    mixes the two layouts there); ``variant=VARIANT_G16`` reproduces the reference's layout.  This op is
    "parity unpinned" against the CUDA binary (CUDA ``rsqrtf`` is a 2-ulp approximation; kernel and oracle use
    the correctly rounded value): it is byte-exact to the restated kernel text, not provably to the reference.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import VARIANT_G16, VARIANT_G32, OUT_BF16, OUT_F32, ArcqError  # noqa: F401  (re-exported)


import functools

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor) -> int:
    # the current HIP stream of the tensor's device as an integer handle.  The private accessor (what torch's own extensions use)
    # saves ~1.5 us per call over building a torch.cuda.Stream object; an eager decode step makes ~170 of these calls and is
    # host-paced (tools/host_overhead.py)
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


class _on:
    """``with _on(device):`` -- torch.cuda.device(device), but free when the device is already current (the usual case)."""
    __slots__ = ("ctx",)

    def __init__(self, device):
        idx = device.index
        self.ctx = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _need(t: torch.Tensor, dtype, name: str, ndim=None):
    # the reference's data_ptr<T>() throws c10::Error on a dtype mismatch; we raise RuntimeError
    if not isinstance(t, torch.Tensor) or t.dtype != dtype:
        raise RuntimeError(f"agemm: {name} must be a {dtype} tensor, got {getattr(t, 'dtype', type(t))}")
    if not t.is_cuda:
        raise RuntimeError(f"agemm: {name} must live on the GPU (there is no CPU path)")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"agemm: {name} must be {ndim}-D, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise RuntimeError(f"agemm: {name} must be contiguous")


@functools.lru_cache(maxsize=None)
def variant_for_kq(KQ: int) -> int:
    """Layout variant the reference's dispatch picks for this in_features (bindings.cpp:141-160)."""
    return int(_lib.lib().arcq_variant_for_kq(int(KQ)))


@functools.lru_cache(maxsize=None)
def sf_buffer_bytes(rows: int, K: int) -> int:
    """get_sf{a,b}_buffer_size_in_bytes (bindings.cpp:83-95)."""
    return int(_lib.lib().arcq_sf_alloc_bytes(int(rows), int(K)))


@functools.lru_cache(maxsize=None)
def _sf_used(rows: int, K: int) -> int:
    return int(_lib.lib().arcq_sf_used_bytes(int(rows), int(K)))


@functools.lru_cache(maxsize=None)
def _repacked_bytes(N: int, K: int):
    L = _lib.lib()
    return int(L.arcq_repacked_w_bytes(N, K)), int(L.arcq_repacked_sf_bytes(N, K))


def _quantize(fn_name: str, X: torch.Tensor, reorder_index: torch.Tensor, KE: int, variant):
    _need(X, torch.bfloat16, "X" if fn_name.endswith("_x") else "W", 2)
    _need(reorder_index, torch.int16, "reorder_index", 1)
    rows, KQ = X.shape
    KE = int(KE)
    if reorder_index.numel() != KQ:
        raise RuntimeError(f"agemm: reorder_index has {reorder_index.numel()} entries, expected {KQ}")
    K = KQ + KE
    if variant is None:
        variant = variant_for_kq(KQ)
    if KQ % 64 or KE % 64 or KE < 0 or KE > KQ:
        raise RuntimeError(f"Value error in {fn_name}: KQ={KQ}, KE={KE} is not valid")
    Q = torch.empty((rows, K // 2), dtype=torch.uint8, device=X.device)
    SF = torch.empty((sf_buffer_bytes(rows, K),), dtype=torch.uint8, device=X.device)
    L = _lib.lib()
    fn = L.arcq_quantize_x if fn_name.endswith("_x") else L.arcq_quantize_w
    with _on(X.device):
        st = fn(X.data_ptr(), reorder_index.data_ptr(), Q.data_ptr(), SF.data_ptr(), rows, KQ, KE, int(variant), _stream(X))
    _lib.check(st, fn_name)
    return Q, SF


def reorder_quantize_x(X: torch.Tensor, reorder_index: torch.Tensor, KE: int, variant=None):
    """agemm.reorder_quantize_x(X, reorder_index, KE) -> (QX u8 [M,(KQ+KE)/2], SFX u8 [(M/128+1)*128*(KQ+KE)/16])
    (bindings.cpp:122-163)."""
    return _quantize("reorder_quantize_x", X, reorder_index, KE, variant)


def reorder_quantize_w(W: torch.Tensor, reorder_index: torch.Tensor, KE: int, variant=None):
    """agemm.reorder_quantize_w(W, reorder_index, KE) -> (QW, SFW) (bindings.cpp:170-210)."""
    return _quantize("reorder_quantize_w", W, reorder_index, KE, variant)


def rmsnorm_quantize_x(X: torch.Tensor, W: torch.Tensor, eps: float, reorder_index: torch.Tensor, KE: int, variant=None):
    """agemm.rmsnorm_quantize_x(X, W, eps, reorder_index, KE) -> (QX, SFX) (bindings.cpp:216-254)."""
    _need(X, torch.bfloat16, "X", 2)
    _need(W, torch.bfloat16, "W", 1)
    _need(reorder_index, torch.int16, "reorder_index", 1)
    M, KQ = X.shape
    KE = int(KE)
    K = KQ + KE
    if W.numel() != KQ or reorder_index.numel() != KQ:
        raise RuntimeError("agemm: rmsnorm weight / reorder_index length must equal X.shape[1]")
    if variant is None:
        variant = variant_for_kq(KQ)
    if KQ % 64 or KE % 64 or KE < 0 or KE > KQ or not (2048 <= KQ <= 8192):
        raise RuntimeError(f"Value error in run_rmsnorm_x_bf16_nvfp4: K value is not valid: {KQ}")
    QX = torch.empty((M, K // 2), dtype=torch.uint8, device=X.device)
    SFX = torch.empty((sf_buffer_bytes(M, K),), dtype=torch.uint8, device=X.device)
    with _on(X.device):
        st = _lib.lib().arcq_rmsnorm_quantize_x(X.data_ptr(), W.data_ptr(), float(eps), reorder_index.data_ptr(), QX.data_ptr(),
                                                SFX.data_ptr(), M, KQ, KE, int(variant), _stream(X))
    _lib.check(st, "rmsnorm_quantize_x")
    return QX, SFX


def matmul(A: torch.Tensor, B: torch.Tensor, SFA: torch.Tensor, SFB: torch.Tensor, scale, *, bias=None, residual=None,
           out_dtype=torch.bfloat16, out=None, scale_host: float = 1.0):
    """agemm.matmul(A, B, SFA, SFB, scale) -> bf16 [M, N]  (bindings.cpp:99-120).

    ``scale`` may be a Python float (the reference's ``const float``) or a 0-dim / 1-element fp32 device
    tensor; the latter is consumed on the device (the reference converts it with an implicit ``.item()``
    sync).  Extensions used by the host mirror / e2e harness: ``bias`` (bf16 [N]), ``residual`` (bf16 [M,N], added
    after the bf16 rounding, like ``x + linear(...)``), ``out_dtype=torch.float32``, ``out``, and ``scale_host`` (a host
    float multiplied into a device ``scale``: weight scale x activation scale without a tiny multiply kernel).
    """
    _need(A, torch.uint8, "A", 2)
    _need(B, torch.uint8, "B", 2)
    _need(SFA, torch.uint8, "SFA")
    _need(SFB, torch.uint8, "SFB")
    M, N, K = A.shape[0], B.shape[0], A.shape[1] * 2       # bindings.cpp:107-109
    if B.shape[1] * 2 != K:
        raise RuntimeError(f"agemm.matmul: A has K={K}, B has K={B.shape[1] * 2}")
    L = _lib.lib()
    if SFA.numel() < _sf_used(M, K) or SFB.numel() < _sf_used(N, K):
        raise RuntimeError("agemm.matmul: scale-factor buffer smaller than the swizzled layout of its operand")
    alpha_host, alpha_dev = float(scale_host), None
    if isinstance(scale, torch.Tensor):
        if scale.is_cuda and scale.dtype == torch.float32 and scale.numel() == 1:
            alpha_dev = scale
        else:
            alpha_host *= float(scale)         # CPU tensor or other dtype: same as the reference's __float__
    else:
        alpha_host *= float(scale)
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise RuntimeError("agemm.matmul: out_dtype must be bfloat16 or float32")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    else:
        _need(out, out_dtype, "out", 2)
        if tuple(out.shape) != (M, N):
            raise RuntimeError("agemm.matmul: out has the wrong shape")
    if bias is not None:
        _need(bias, torch.bfloat16, "bias", 1)
        if bias.numel() != N:
            raise RuntimeError("agemm.matmul: bias must have N entries")
    if residual is not None:
        _need(residual, torch.bfloat16, "residual", 2)
        if tuple(residual.shape) != (M, N):
            raise RuntimeError("agemm.matmul: residual must be [M, N]")
    ws_bytes = int(L.arcq_gemm_workspace_bytes(M, N, K))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=A.device) if ws_bytes else None
    with _on(A.device):
        st = L.arcq_gemm_nvfp4(A.data_ptr(), B.data_ptr(), SFA.data_ptr(), SFB.data_ptr(), out.data_ptr(), M, N, K,
                               alpha_host, alpha_dev.data_ptr() if alpha_dev is not None else None,
                               bias.data_ptr() if bias is not None else None,
                               residual.data_ptr() if residual is not None else None,
                               OUT_BF16 if out_dtype == torch.bfloat16 else OUT_F32,
                               ws.data_ptr() if ws is not None else None, ws_bytes, _stream(A))
    _lib.check(st, "matmul")
    return out


def absmax_scale(X: torch.Tensor) -> torch.Tensor:
    """Extension: ``max|X| / (448*6)`` as a 0-dim fp32 device tensor, no host sync.
    Equals ``torch.max(x.abs()).float() / (448.0*6.0)`` of model/qLlamaLayer.py:74 bit for bit (0-dim, so that
    ``x / scale`` keeps x's dtype exactly as with the reference's scalar tensor)."""
    _need(X, torch.bfloat16, "X")
    out = torch.empty((1,), dtype=torch.float32, device=X.device)
    with _on(X.device):
        st = _lib.lib().arcq_absmax_scale(X.data_ptr(), X.numel(), out.data_ptr(), _stream(X))
    _lib.check(st, "absmax_scale")
    return out.reshape(())


_dyn_state = {}


class _NoIndex:
    # reorder_index=None (arcq_quantize_x_dyn_slots only): X is already in reordered channel order, NULL at the C-ABI
    @staticmethod
    def data_ptr():
        return None


def _quantize_dynamic(entry: str, who: str, X: torch.Tensor, KQ: int, reorder_index, KE: int, variant, slots=None, layout=None):
    M = X.shape[0]
    KE = int(KE)
    K = KQ + KE
    if variant is None:
        variant = variant_for_kq(KQ)
    if reorder_index is None:
        if entry != "arcq_quantize_x_dyn_slots":
            raise RuntimeError(f"Value error in {who}: reorder_index=None needs absmax_slots")
        if not X.is_contiguous():
            raise RuntimeError(f"Value error in {who}: X must be contiguous")
        reorder_index = _NoIndex
    else:
        _need(reorder_index, torch.int16, "reorder_index", 1)
    if KQ % 64 or KE % 64 or KE < 0 or KE > KQ or (reorder_index is not _NoIndex and reorder_index.numel() != KQ):
        raise RuntimeError(f"Value error in {who}: KQ={KQ}, KE={KE} is not valid")
    dev = X.device
    key = (dev, _stream(X))                # scratch of the abs-max pass: one per device and stream (include/arcq.h)
    state = _dyn_state.get(key)
    if state is None:
        state = _dyn_state[key] = torch.empty(256, dtype=torch.int32, device=dev)
    QX = torch.empty((M, K // 2), dtype=torch.uint8, device=dev)
    SFX = torch.empty((sf_buffer_bytes(M, K),), dtype=torch.uint8, device=dev)
    scale = torch.empty((1,), dtype=torch.float32, device=dev)
    with _on(dev):
        if slots is not None and layout is not None:
            st = getattr(_lib.lib(), entry)(X.data_ptr(), reorder_index.data_ptr(), QX.data_ptr(), SFX.data_ptr(), scale.data_ptr(),
                                            slots.data_ptr(), slots.numel(), M, KQ, KE, int(variant), int(layout), _stream(X))
        elif slots is not None:
            st = getattr(_lib.lib(), entry)(X.data_ptr(), reorder_index.data_ptr(), QX.data_ptr(), SFX.data_ptr(), scale.data_ptr(),
                                            slots.data_ptr(), slots.numel(), M, KQ, KE, int(variant), _stream(X))
        elif layout is not None:
            st = getattr(_lib.lib(), entry)(X.data_ptr(), reorder_index.data_ptr(), QX.data_ptr(), SFX.data_ptr(), scale.data_ptr(),
                                            state.data_ptr(), M, KQ, KE, int(variant), int(layout), _stream(X))
        else:
            st = getattr(_lib.lib(), entry)(X.data_ptr(), reorder_index.data_ptr(), QX.data_ptr(), SFX.data_ptr(), scale.data_ptr(),
                                            state.data_ptr(), M, KQ, KE, int(variant), _stream(X))
    _lib.check(st, who)
    return QX, SFX, scale.reshape(())


def repack_w(QW: torch.Tensor, SFW: torch.Tensor):
    """One-time re-layout of a quantised weight for the decode fast path (include/arcq.h, "REPACKED weight"): returns
    ``(RW, RSF)``.  Pure data movement with torch ops -- codes and scale bytes are those of ``reorder_quantize_w``:
    RW  = [row blocks of 16][tiles of 128 K][lane = 16*q + r][16 bytes], K padded to a multiple of 256, N to 16;
    RSF = [row blocks][tile pairs][lane][4 bytes: the lane's two scale bytes in each tile of the pair]."""
    _need(QW, torch.uint8, "QW", 2)
    _need(SFW, torch.uint8, "SFW", 1)
    N, K = QW.shape[0], QW.shape[1] * 2
    if K % 64 or SFW.numel() < _sf_used(N, K):
        raise RuntimeError("Value error in repack_w: K % 64 != 0 or the scale buffer is too small")
    dev = QW.device
    Np, Kp = (N + 15) // 16 * 16, (K + 255) // 256 * 256
    q = torch.zeros((Np, Kp // 2), dtype=torch.uint8, device=dev)
    q[:N, : K // 2] = QW
    # natural [N, K/16] scale matrix out of the swizzled buffer
    r = torch.arange(N, device=dev).unsqueeze(1)
    g = torch.arange(K // 16, device=dev).unsqueeze(0)
    off = ((r // 128) * (K // 64) + g // 4) * 512 + (r % 32) * 16 + ((r // 32) % 4) * 4 + g % 4
    sf = torch.zeros((Np, Kp // 16), dtype=torch.uint8, device=dev)
    sf[:N, : K // 16] = SFW[off]
    RB, T = Np // 16, Kp // 128
    # codes: [RB, r, T, q, 16 B] -> [RB, T, q, r, 16 B]
    RW = q.view(RB, 16, T, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    # scales: a tile has 8 groups per row, lane (r, q) owns groups 2q and 2q + 1: [RB, r, T/2, 2 tiles, q, 2] -> [RB, T/2, q, r, tile, 2]
    RSF = sf.view(RB, 16, T // 2, 2, 4, 2).permute(0, 2, 4, 1, 3, 5).contiguous().view(-1)
    L = _lib.lib()
    assert RW.numel() == L.arcq_repacked_w_bytes(N, K) and RSF.numel() == L.arcq_repacked_sf_bytes(N, K)
    return RW, RSF


@functools.lru_cache(maxsize=None)
def repacked_supported(M: int, N: int, K: int) -> bool:
    """Whether ``matmul_repacked`` can run this shape: M <= 16 and the fp16 image of the activations fits LDS, or 16 < M <= 64 and
    the packed activations fit LDS (decode batches: the weight is still read once)."""
    return bool(_lib.lib().arcq_gemm_repacked_supported(int(M), int(N), int(K)))


def matmul_repacked(A: torch.Tensor, RW: torch.Tensor, SFA: torch.Tensor, RSF: torch.Tensor, scale, N: int, *, bias=None, residual=None,
                    out_dtype=torch.bfloat16, out=None, scale_host: float = 1.0, kernel: str = "auto"):
    """``matmul`` for decode shapes over a weight prepared by ``repack_w`` (same arguments otherwise, plus the row count
    ``N`` of the weight): the kernel streams the weight in MFMA operand order with no LDS transpose and no barrier in its
    K loop.  Equals ``matmul`` on the un-repacked operands up to fp32 accumulation order."""
    _need(A, torch.uint8, "A", 2)
    _need(RW, torch.uint8, "RW", 1)
    _need(SFA, torch.uint8, "SFA")
    _need(RSF, torch.uint8, "RSF", 1)
    M, K, N = A.shape[0], A.shape[1] * 2, int(N)
    L = _lib.lib()
    if K % 64 or (RW.numel(), RSF.numel()) != _repacked_bytes(N, K):
        raise RuntimeError(f"Value error in matmul_repacked: RW / RSF do not belong to a [{N}, {K}] weight")
    if SFA.numel() < _sf_used(M, K):
        raise RuntimeError("Value error in matmul_repacked: SFA smaller than the swizzled layout of A")
    if not repacked_supported(M, N, K):
        raise RuntimeError(f"matmul_repacked: M={M}, K={K} is outside the repacked path (see repacked_supported)")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise RuntimeError("agemm.matmul_repacked: out_dtype must be bfloat16 or float32")
    alpha_host, alpha_dev = float(scale_host), None
    if isinstance(scale, torch.Tensor) and scale.is_cuda and scale.dtype == torch.float32 and scale.numel() == 1:
        alpha_dev = scale
    else:
        alpha_host *= float(scale)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    elif tuple(out.shape) != (M, N) or out.dtype != out_dtype or not out.is_contiguous():
        raise RuntimeError("agemm.matmul_repacked: out has the wrong shape / dtype")
    if bias is not None:
        _need(bias, torch.bfloat16, "bias", 1)
        if bias.numel() != N:
            raise RuntimeError("agemm.matmul_repacked: bias must have N entries")
    if residual is not None:
        _need(residual, torch.bfloat16, "residual", 2)
        if tuple(residual.shape) != (M, N):
            raise RuntimeError("agemm.matmul_repacked: residual must be [M, N]")
    # kernel="stream": the kernel body of the fused decode linears (rmsnorm_matmul_repacked, dynamic_matmul_repacked), so that
    # quantiser + this call is their bit-exact two-launch equivalent; "auto" = the fastest kernel for plain packed activations
    fn = L.arcq_gemm_nvfp4_repacked_stream if kernel == "stream" else L.arcq_gemm_nvfp4_repacked
    with _on(A.device):
        st = fn(A.data_ptr(), RW.data_ptr(), SFA.data_ptr(), RSF.data_ptr(), out.data_ptr(), M, N, K, alpha_host,
                                        alpha_dev.data_ptr() if alpha_dev is not None else None,
                                        bias.data_ptr() if bias is not None else None,
                                        residual.data_ptr() if residual is not None else None,
                                        OUT_BF16 if out_dtype == torch.bfloat16 else OUT_F32, _stream(A))
    _lib.check(st, "matmul_repacked")
    return out


def matmul_silu_mul(A: torch.Tensor, B: torch.Tensor, SFA: torch.Tensor, SFB: torch.Tensor, scale, *, scale_host: float = 1.0, bias=None):
    """Extension: the gate|up GEMM with the MLP's ``act_fn(gate) * up`` (SiLU, model/qLlamaLayer.py:417) in
    its epilogue.  ``B`` is the quantised weight whose ROWS INTERLEAVE gate and up (g0, u0, g1, u1, ...).  Returns
    ``(act, absmax_slots)``: ``act`` bf16 [M, N/2] equals ``F.silu(y[:, 0::2]) * y[:, 1::2]`` of ``y = matmul(A, B, ...)``
    bit for bit; ``absmax_slots`` feeds ``reorder_quantize_x_dynamic(act, ..., absmax_slots=...)`` (one launch)."""
    _need(A, torch.uint8, "A", 2)
    _need(B, torch.uint8, "B", 2)
    _need(SFA, torch.uint8, "SFA")
    _need(SFB, torch.uint8, "SFB")
    M, N, K = A.shape[0], B.shape[0], A.shape[1] * 2
    if B.shape[1] != A.shape[1] or K % 64 or N % 8:
        raise RuntimeError(f"Value error in matmul_silu_mul: A {tuple(A.shape)} / B {tuple(B.shape)} need equal K, K % 64 == 0, N % 8 == 0")
    if SFA.numel() < _sf_used(M, K) or SFB.numel() < _sf_used(N, K):
        raise RuntimeError("Value error in matmul_silu_mul: scale buffer too small")
    alpha_host, alpha_dev = float(scale_host), None
    if isinstance(scale, torch.Tensor) and scale.is_cuda:
        alpha_dev = scale.reshape(-1)[:1].to(torch.float32)
    else:
        alpha_host *= float(scale)
    L = _lib.lib()
    act = torch.empty((M, N // 2), dtype=torch.bfloat16, device=A.device)
    slots = torch.empty((max(1, int(L.arcq_gemm_silu_mul_slots(M, N, K))),), dtype=torch.int32, device=A.device)
    with _on(A.device):
        st = L.arcq_gemm_nvfp4_silu_mul(A.data_ptr(), B.data_ptr(), SFA.data_ptr(), SFB.data_ptr(), act.data_ptr(), slots.data_ptr(), M, N, K,
                                        alpha_host, alpha_dev.data_ptr() if alpha_dev is not None else None,
                                        _opt(bias, torch.bfloat16, "bias", (N,)), _stream(A))
    _lib.check(st, "matmul_silu_mul")
    return act, slots


def reorder_quantize_x_dynamic(X: torch.Tensor, reorder_index: torch.Tensor, KE: int, variant=None, absmax_slots=None):
    """Extension: ``NVFP4_reorder_quantize_x`` (model/qLlamaLayer.py:73-77) without a host sync, in ONE launch
    for decode-sized inputs (<= 256 KB) and two otherwise: returns (QX, SFX, scale) with ``scale = max|X|/2688`` a 0-dim
    fp32 device tensor and (QX, SFX) byte-identical to ``reorder_quantize_x(X / scale, reorder_index, KE)``.
    ``reorder_index=None`` (with ``absmax_slots``): X is ALREADY in reordered channel order (``rmsnorm_matmul_repacked_silu(...,
    act_scatter_index=)`` stored it so) -- same bytes as the natural-order tensor with the index, without the gather."""
    _need(X, torch.bfloat16, "X", 2)
    if absmax_slots is not None:           # max|X| already known per workgroup (matmul_silu_mul): one launch for any size
        _need(absmax_slots, torch.int32, "absmax_slots", 1)
        return _quantize_dynamic("arcq_quantize_x_dyn_slots", "reorder_quantize_x_dynamic", X, X.shape[1], reorder_index, KE, variant,
                                 slots=absmax_slots)
    return _quantize_dynamic("arcq_quantize_x_dyn", "reorder_quantize_x_dynamic", X, X.shape[1], reorder_index, KE, variant)


GU_HALVES, GU_PAIRS = 0, 1


def silu_mul_quantize_x_dynamic(GU: torch.Tensor, reorder_index: torch.Tensor, KE: int, variant=None, layout: int = GU_HALVES, absmax_slots=None):
    """Extension: the MLP's ``act_fn(gate) * up`` (model/qLlamaLayer.py:417, SiLU) folded into the dynamic quantiser.
    ``GU`` is [M, 2*KQ] bf16, the output of a fused gate_up projection: ``layout=GU_HALVES`` (gate | up) or ``GU_PAIRS``
    (g0, u0, g1, u1, ...: a weight with interleaved gate/up rows, as ``matmul_silu_mul`` takes).  Returns what
    ``reorder_quantize_x_dynamic(F.silu(gate) * up, ...)`` returns, byte for byte, in two launches instead of four and
    without materialising the product."""
    _need(GU, torch.bfloat16, "GU", 2)
    if GU.shape[1] % 2:
        raise RuntimeError("Value error in silu_mul_quantize_x_dynamic: GU must hold gate and up halves of equal width")
    if layout not in (GU_HALVES, GU_PAIRS):
        raise RuntimeError("Value error in silu_mul_quantize_x_dynamic: layout must be GU_HALVES or GU_PAIRS")
    if absmax_slots is not None:           # max |silu(gate) * up| words left by matmul_repacked_silu_absmax: ONE launch
        if absmax_slots.dtype != torch.int32 or not absmax_slots.is_cuda or not absmax_slots.is_contiguous() or absmax_slots.numel() == 0:
            raise RuntimeError("agemm.silu_mul_quantize_x_dynamic: absmax_slots must be a non-empty contiguous int32 GPU tensor")
        return _quantize_dynamic("arcq_silu_mul_quantize_x_dyn_slots", "silu_mul_quantize_x_dynamic", GU, GU.shape[1] // 2, reorder_index, KE,
                                 variant, slots=absmax_slots, layout=layout)
    return _quantize_dynamic("arcq_silu_mul_quantize_x_dyn", "silu_mul_quantize_x_dynamic", GU, GU.shape[1] // 2, reorder_index, KE,
                             variant, layout=layout)


def matmul_repacked_silu_absmax(A: torch.Tensor, RW: torch.Tensor, SFA: torch.Tensor, RSF: torch.Tensor, scale, N: int, *, out=None,
                                scale_host: float = 1.0):
    """Extension for decode: ``matmul_repacked`` for a gate|up weight whose ROWS INTERLEAVE gate and up (g0, u0, g1, u1, ...).
    Returns ``(y, absmax_slots)``: ``y`` bf16 [M, N] exactly as ``matmul_repacked`` returns it, and one int32 word per block of
    16 weight rows holding max |silu(g) * u| of that block's outputs -- what ``silu_mul_quantize_x_dynamic(y, ...,
    layout=GU_PAIRS, absmax_slots=...)`` needs to quantise ``act_fn(gate) * up`` (model/qLlamaLayer.py:417) in ONE launch."""
    _need(A, torch.uint8, "A", 2)
    _need(RW, torch.uint8, "RW", 1)
    _need(SFA, torch.uint8, "SFA")
    _need(RSF, torch.uint8, "RSF", 1)
    M, K, N = A.shape[0], A.shape[1] * 2, int(N)
    L = _lib.lib()
    if K % 64 or N % 4 or (RW.numel(), RSF.numel()) != _repacked_bytes(N, K):
        raise RuntimeError(f"Value error in matmul_repacked_silu_absmax: RW / RSF do not belong to a [{N}, {K}] weight, or N % 4 != 0")
    if SFA.numel() < _sf_used(M, K):
        raise RuntimeError("Value error in matmul_repacked_silu_absmax: SFA smaller than the swizzled layout of A")
    if not repacked_supported(M, N, K):
        raise RuntimeError(f"matmul_repacked_silu_absmax: M={M}, K={K} is outside the repacked path (see repacked_supported)")
    alpha_host, alpha_dev = float(scale_host), None
    if isinstance(scale, torch.Tensor) and scale.is_cuda and scale.dtype == torch.float32 and scale.numel() == 1:
        alpha_dev = scale
    else:
        alpha_host *= float(scale)
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    elif tuple(out.shape) != (M, N) or out.dtype != torch.bfloat16 or not out.is_contiguous():
        raise RuntimeError("agemm.matmul_repacked_silu_absmax: out has the wrong shape / dtype")
    slots = torch.empty(((N + 15) // 16,), dtype=torch.int32, device=A.device)
    with _on(A.device):
        st = L.arcq_gemm_nvfp4_repacked_silu_absmax(A.data_ptr(), RW.data_ptr(), SFA.data_ptr(), RSF.data_ptr(), out.data_ptr(),
                                                    slots.data_ptr(), M, N, K, alpha_host,
                                                    alpha_dev.data_ptr() if alpha_dev is not None else None, _stream(A))
    _lib.check(st, "matmul_repacked_silu_absmax")
    return out, slots


SRC_RMSNORM, SRC_DYNAMIC = 1, 2


@functools.lru_cache(maxsize=None)
def fused_supported(kind: int, M: int, N: int, KQ: int, KE: int) -> bool:
    """Whether the fused decode linear (activation quantiser as the GEMM prologue) can run this shape: M <= 16 and the LDS
    budget of one CU.  ``kind``: SRC_RMSNORM or SRC_DYNAMIC.  Callers fall back to the separate calls otherwise."""
    return bool(_lib.lib().arcq_linear_fused_supported(int(kind), int(M), int(N), int(KQ), int(KE)))


def _fused_common(who, X, reorder_index, RW, RSF, N, KE, variant):
    _need(X, torch.bfloat16, "X", 2)
    _need(reorder_index, torch.int16, "reorder_index", 1)
    _need(RW, torch.uint8, "RW", 1)
    _need(RSF, torch.uint8, "RSF", 1)
    M, KQ = X.shape
    KE, N = int(KE), int(N)
    K = KQ + KE
    L = _lib.lib()
    if KQ % 64 or KE % 64 or KE < 0 or KE > KQ or reorder_index.numel() != KQ:
        raise RuntimeError(f"Value error in {who}: KQ={KQ}, KE={KE} is not valid")
    if (RW.numel(), RSF.numel()) != _repacked_bytes(N, K):
        raise RuntimeError(f"Value error in {who}: RW / RSF do not belong to a [{N}, {K}] weight")
    if variant is None:
        variant = variant_for_kq(KQ)
    return M, KQ, KE, N, int(variant)


def _opt(t, dtype, name, shape):
    if t is None:
        return None
    _need(t, dtype, name, len(shape))
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"agemm: {name} must have shape {tuple(shape)}")
    return t.data_ptr()


def rmsnorm_matmul_repacked(X: torch.Tensor, W: torch.Tensor, eps: float, reorder_index: torch.Tensor, KE: int, RW: torch.Tensor,
                            RSF: torch.Tensor, scale, N: int, *, bias=None, residual=None, out_dtype=torch.bfloat16, out=None,
                            scale_host: float = 1.0, variant=None):
    """Extension for decode: ``matmul_repacked(*rmsnorm_quantize_x(X, W, eps, reorder_index, KE), ...)`` in ONE launch -- the
    RMSNorm + quantiser (benchmarks/modeling_arc.py:211-228) runs as the GEMM's prologue, once per CU.  Bit-identical to the
    two calls.  ``W`` is the norm weight, ``scale`` the per-tensor weight scale (float or 0-dim device tensor)."""
    M, KQ, KE, N, variant = _fused_common("rmsnorm_matmul_repacked", X, reorder_index, RW, RSF, N, KE, variant)
    _need(W, torch.bfloat16, "W", 1)
    if W.numel() != KQ or not fused_supported(SRC_RMSNORM, M, N, KQ, KE):
        raise RuntimeError(f"rmsnorm_matmul_repacked: M={M}, KQ={KQ} outside the fused path (see fused_supported)")
    alpha_host, alpha_dev = float(scale_host), None
    if isinstance(scale, torch.Tensor) and scale.is_cuda and scale.dtype == torch.float32 and scale.numel() == 1:
        alpha_dev = scale
    else:
        alpha_host *= float(scale)
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise RuntimeError("agemm.rmsnorm_matmul_repacked: out_dtype must be bfloat16 or float32")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=X.device)
    elif tuple(out.shape) != (M, N) or out.dtype != out_dtype or not out.is_contiguous():
        raise RuntimeError("agemm.rmsnorm_matmul_repacked: out has the wrong shape / dtype")
    with _on(X.device):
        st = _lib.lib().arcq_linear_rmsnorm_repacked(X.data_ptr(), W.data_ptr(), float(eps), reorder_index.data_ptr(), RW.data_ptr(), RSF.data_ptr(),
                                                     out.data_ptr(), M, N, KQ, KE, variant, alpha_host,
                                                     alpha_dev.data_ptr() if alpha_dev is not None else None,
                                                     _opt(bias, torch.bfloat16, "bias", (N,)), _opt(residual, torch.bfloat16, "residual", (M, N)),
                                                     OUT_BF16 if out_dtype == torch.bfloat16 else OUT_F32, _stream(X))
    _lib.check(st, "rmsnorm_matmul_repacked")
    return out


_scatter_ok = {}


def _check_scatter_index(idx: torch.Tensor, n: int):
    """``act_scatter_index`` values are store columns of the kernel's epilogue: anything but a permutation of 0 .. n-1 writes out of
    bounds or leaves columns unwritten.  Checked ONCE per index tensor (one device sync, at registration time in effect), cached by
    storage address and version counter."""
    key = (idx.data_ptr(), idx.numel(), idx._version, idx.device.index)
    if _scatter_ok.get(key):
        return
    ok = idx.numel() == n and bool(torch.equal(torch.sort(idx.long()).values, torch.arange(n, device=idx.device)))
    if not ok:
        raise RuntimeError(f"agemm: act_scatter_index must be a permutation of 0 .. {n - 1}")
    if len(_scatter_ok) > 4096:
        _scatter_ok.clear()
    _scatter_ok[key] = True


def rmsnorm_matmul_repacked_silu(X: torch.Tensor, W: torch.Tensor, eps: float, reorder_index: torch.Tensor, KE: int, RW: torch.Tensor,
                                 RSF: torch.Tensor, scale, N: int, *, scale_host: float = 1.0, variant=None, bias=None,
                                 act_scatter_index=None):
    """Extension for decode, the MLP's first half in ONE launch: RMSNorm + quantise (prologue), gate|up GEMM over a repacked weight
    whose ROWS INTERLEAVE gate and up, ``act_fn(gate) * up`` (SiLU, model/qLlamaLayer.py:417) in the epilogue.  Returns
    ``(act bf16 [M, N/2], absmax_slots int32 [ceil(N/16)])``; ``act`` equals ``F.silu(y[:, 0::2]) * y[:, 1::2]`` of
    ``y = rmsnorm_matmul_repacked(...)`` bit for bit and the slots let ``dynamic_matmul_repacked`` skip its abs-max pass.
    ``act_scatter_index`` (int16 [N/2], a permutation): activation j goes to column ``act_scatter_index[j]``; with the inverse of
    the down projection's reorder_index the result is ``act[:, reorder_index]`` and the consumer quantises it with
    ``reorder_quantize_x_dynamic(act, None, KE, absmax_slots=slots)``."""
    M, KQ, KE, N, variant = _fused_common("rmsnorm_matmul_repacked_silu", X, reorder_index, RW, RSF, N, KE, variant)
    _need(W, torch.bfloat16, "W", 1)
    if W.numel() != KQ or N % 4 or not fused_supported(SRC_RMSNORM, M, N, KQ, KE):
        raise RuntimeError(f"rmsnorm_matmul_repacked_silu: M={M}, N={N}, KQ={KQ} outside the fused path (see fused_supported; N % 4 == 0)")
    alpha_host, alpha_dev = float(scale_host), None
    if isinstance(scale, torch.Tensor) and scale.is_cuda and scale.dtype == torch.float32 and scale.numel() == 1:
        alpha_dev = scale
    else:
        alpha_host *= float(scale)
    if act_scatter_index is not None:
        _need(act_scatter_index, torch.int16, "act_scatter_index", 1)
        _check_scatter_index(act_scatter_index, N // 2)
    act = torch.empty((M, N // 2), dtype=torch.bfloat16, device=X.device)
    slots = torch.empty(((N + 15) // 16,), dtype=torch.int32, device=X.device)
    with _on(X.device):
        st = _lib.lib().arcq_linear_rmsnorm_silu_repacked(X.data_ptr(), W.data_ptr(), float(eps), reorder_index.data_ptr(), RW.data_ptr(),
                                                          RSF.data_ptr(), act.data_ptr(), slots.data_ptr(), M, N, KQ, KE, variant, alpha_host,
                                                          alpha_dev.data_ptr() if alpha_dev is not None else None,
                                                          _opt(bias, torch.bfloat16, "bias", (N,)),
                                                          _opt(act_scatter_index, torch.int16, "act_scatter_index", (N // 2,)), _stream(X))
    _lib.check(st, "rmsnorm_matmul_repacked_silu")
    return act, slots


def dynamic_matmul_repacked(X: torch.Tensor, reorder_index: torch.Tensor, KE: int, RW: torch.Tensor, RSF: torch.Tensor, scale_w: float, N: int, *,
                            absmax_slots=None, bias=None, residual=None, out_dtype=torch.bfloat16, out=None, variant=None):
    """Extension for decode: ``NVFP4_reorder_quantize_x`` (model/qLlamaLayer.py:73-77) + ``QLinearLayer.forward``
    (qLinearLayer.py:62-78) in ONE launch: scale = max|X|/2688 (from ``absmax_slots`` when the producing kernel left them, else
    computed from X by every workgroup), X/scale quantised in the GEMM's prologue, alpha = scale * scale_w (a host float: the
    weight's per-tensor scale).  Returns ``(y, scale)``; bit-identical to ``reorder_quantize_x_dynamic`` + ``matmul_repacked``."""
    M, KQ, KE, N, variant = _fused_common("dynamic_matmul_repacked", X, reorder_index, RW, RSF, N, KE, variant)
    if not fused_supported(SRC_DYNAMIC, M, N, KQ, KE):
        raise RuntimeError(f"dynamic_matmul_repacked: M={M}, KQ={KQ} outside the fused path (see fused_supported)")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise RuntimeError("agemm.dynamic_matmul_repacked: out_dtype must be bfloat16 or float32")
    if absmax_slots is not None and (absmax_slots.dtype != torch.int32 or not absmax_slots.is_cuda or not absmax_slots.is_contiguous()
                                     or absmax_slots.numel() == 0):
        raise RuntimeError("agemm.dynamic_matmul_repacked: absmax_slots must be a non-empty contiguous int32 GPU tensor")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=X.device)
    elif tuple(out.shape) != (M, N) or out.dtype != out_dtype or not out.is_contiguous():
        raise RuntimeError("agemm.dynamic_matmul_repacked: out has the wrong shape / dtype")
    scale = torch.empty((1,), dtype=torch.float32, device=X.device)
    with _on(X.device):
        st = _lib.lib().arcq_linear_dynamic_repacked(X.data_ptr(), reorder_index.data_ptr(), RW.data_ptr(), RSF.data_ptr(), out.data_ptr(),
                                                     scale.data_ptr(), absmax_slots.data_ptr() if absmax_slots is not None else None,
                                                     absmax_slots.numel() if absmax_slots is not None else 0, M, N, KQ, KE, variant,
                                                     float(scale_w), _opt(bias, torch.bfloat16, "bias", (N,)),
                                                     _opt(residual, torch.bfloat16, "residual", (M, N)),
                                                     OUT_BF16 if out_dtype == torch.bfloat16 else OUT_F32, _stream(X))
    _lib.check(st, "dynamic_matmul_repacked")
    return out, scale.reshape(())


# --- KV-cache functions of the reference module (bindings.cpp:576-581): OUT OF SCOPE.
# Present so that `from model.kv_cache import *` still imports against this module.
def _kv_stub(name):
    def f(*args, **kwargs):
        raise NotImplementedError(f"agemm.{name}: the int4 paged-KV attention is outside the ARC-NVFP4 GEMM hot path")
    f.__name__ = name
    return f


batch_decode_i4 = _kv_stub("batch_decode_i4")
init_kv_i4 = _kv_stub("init_kv_i4")
append_kv_i4 = _kv_stub("append_kv_i4")
batch_decode_f16 = _kv_stub("batch_decode_f16")
init_kv_f16 = _kv_stub("init_kv_f16")
append_kv_f16 = _kv_stub("append_kv_f16")
