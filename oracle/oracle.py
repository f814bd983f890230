"""ctypes front-end of the CPU oracle (oracle/arcq_oracle.c).

TEST INFRASTRUCTURE ONLY.  Nothing under ``arcquant_amd/`` may import this module; it is used by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` as the checker.

Parity status: "parity unpinned" against the reference's CUDA binary (not runnable here, no golden
vectors in the reference); pinned against the reference's importable Python fake-quant path through
``tests/golden`` (see the header of arcq_oracle.c and DESIGN.md).

All arrays are numpy; bf16 tensors travel as ``uint16`` bit patterns.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

G16 = 0  # one 16-group per thread: reorder.cu:68-330, rmsnorm.cu:68-255
G32 = 1  # two 16-groups per thread: reorder.cu:380-696, down.cu:71-361

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ARCQ_ORACLE_LIB") or os.path.join(_HERE, "_build", "libarcq_oracle.so")   # override: the sanitizer build (make asan)
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "arcq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        p, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.arcq_o_bf16_to_f32.restype = f32
        L.arcq_o_bf16_to_f32.argtypes = [ctypes.c_uint16]
        L.arcq_o_f32_to_bf16.restype = ctypes.c_uint16
        L.arcq_o_f32_to_bf16.argtypes = [f32]
        L.arcq_o_e2m1_encode.restype = ctypes.c_uint8
        L.arcq_o_e2m1_encode.argtypes = [f32]
        L.arcq_o_e2m1_decode.restype = f32
        L.arcq_o_e2m1_decode.argtypes = [ctypes.c_uint8]
        L.arcq_o_ue4m3_encode.restype = ctypes.c_uint8
        L.arcq_o_ue4m3_encode.argtypes = [f32]
        L.arcq_o_ue4m3_decode.restype = f32
        L.arcq_o_ue4m3_decode.argtypes = [ctypes.c_uint8]
        L.arcq_o_sf_offset.restype = i64
        L.arcq_o_sf_offset.argtypes = [i64, i64, i64]
        L.arcq_o_sf_alloc_bytes.restype = i64
        L.arcq_o_sf_alloc_bytes.argtypes = [i64, i64]
        L.arcq_o_sf_used_bytes.restype = i64
        L.arcq_o_sf_used_bytes.argtypes = [i64, i64]
        L.arcq_o_primary_pos.restype = i64
        L.arcq_o_primary_pos.argtypes = [i64, i64, i64, i32]
        L.arcq_o_residual_pos.restype = i64
        L.arcq_o_residual_pos.argtypes = [i64, i64, i64, i32]
        L.arcq_o_quantize_x.restype = i32
        L.arcq_o_quantize_x.argtypes = [p, p, i64, i64, i64, i32, p, p]
        L.arcq_o_quantize_w.restype = i32
        L.arcq_o_quantize_w.argtypes = [p, p, i64, i64, i64, i32, p, p]
        L.arcq_o_rmsnorm_quantize_x.restype = i32
        L.arcq_o_rmsnorm_quantize_x.argtypes = [p, p, f32, p, i64, i64, i64, i32, p, p]
        L.arcq_o_quantize_sem.restype = i32
        L.arcq_o_quantize_sem.argtypes = [p, i64, i64, i64, i32, i32, i32, f32, f32, p, p, p]
        L.arcq_o_dequant.restype = None
        L.arcq_o_dequant.argtypes = [p, p, i64, i64, p]
        L.arcq_o_gemm.restype = i32
        L.arcq_o_gemm.argtypes = [p, p, p, p, i64, i64, i64, f32, p, p, p]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    a = np.ascontiguousarray(a)
    assert a.dtype == dtype, (a.dtype, dtype)
    return a


# --------------------------------------------------------------------------------------------------
# scalar formats / layouts
# --------------------------------------------------------------------------------------------------
def e2m1_encode(x: float) -> int:
    return int(lib().arcq_o_e2m1_encode(float(x)))


def e2m1_decode(code: int) -> float:
    return float(lib().arcq_o_e2m1_decode(int(code)))


def ue4m3_encode(x: float) -> int:
    return int(lib().arcq_o_ue4m3_encode(float(x)))


def ue4m3_decode(code: int) -> float:
    return float(lib().arcq_o_ue4m3_decode(int(code)))


def sf_offset(r: int, p: int, K: int) -> int:
    return int(lib().arcq_o_sf_offset(r, p, K))


def sf_alloc_bytes(rows: int, K: int) -> int:
    return int(lib().arcq_o_sf_alloc_bytes(rows, K))


def sf_used_bytes(rows: int, K: int) -> int:
    return int(lib().arcq_o_sf_used_bytes(rows, K))


def primary_pos(g: int, KQ: int, KE: int, variant: int) -> int:
    return int(lib().arcq_o_primary_pos(g, KQ, KE, variant))


def residual_pos(g: int, KQ: int, KE: int, variant: int) -> int:
    return int(lib().arcq_o_residual_pos(g, KQ, KE, variant))


def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """RNE fp32 -> bf16 bit patterns (uint16), vectorised numpy restatement of arcq_o_f32_to_bf16."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = (u + (0x7FFF + ((u >> 16) & 1))) >> 16
    return r.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


# --------------------------------------------------------------------------------------------------
# quantisers / GEMM
# --------------------------------------------------------------------------------------------------
SF_FILL = 0xEE  # poison for bytes the kernels never write (torch::empty in the reference)


def quantize_x(X_bits, idx, KE, variant, sf_fill=SF_FILL):
    """reorder_quantize_x (bindings.cpp:122-163).  X_bits [M,KQ] uint16, idx [KQ] int16."""
    X_bits = _c(X_bits, np.uint16)
    idx = _c(idx, np.int16)
    M, KQ = X_bits.shape
    K = KQ + KE
    QX = np.zeros((M, K // 2), np.uint8)
    SFX = np.full(sf_alloc_bytes(M, K), sf_fill, np.uint8)
    rc = lib().arcq_o_quantize_x(_ptr(X_bits), _ptr(idx), M, KQ, KE, variant, _ptr(QX), _ptr(SFX))
    if rc:
        raise ValueError(f"oracle quantize_x: bad shape M={M} KQ={KQ} KE={KE} variant={variant}")
    return QX, SFX


def quantize_w(W_bits, idx, KE, variant, sf_fill=SF_FILL):
    """reorder_quantize_w (bindings.cpp:170-210)."""
    W_bits = _c(W_bits, np.uint16)
    idx = _c(idx, np.int16)
    N, KQ = W_bits.shape
    K = KQ + KE
    QW = np.zeros((N, K // 2), np.uint8)
    SFW = np.full(sf_alloc_bytes(N, K), sf_fill, np.uint8)
    rc = lib().arcq_o_quantize_w(_ptr(W_bits), _ptr(idx), N, KQ, KE, variant, _ptr(QW), _ptr(SFW))
    if rc:
        raise ValueError(f"oracle quantize_w: bad shape N={N} KQ={KQ} KE={KE} variant={variant}")
    return QW, SFW


def rmsnorm_quantize_x(X_bits, Wn_bits, eps, idx, KE, variant=G16, sf_fill=SF_FILL):
    """rmsnorm_quantize_x (bindings.cpp:216-254)."""
    X_bits = _c(X_bits, np.uint16)
    Wn_bits = _c(Wn_bits, np.uint16)
    idx = _c(idx, np.int16)
    M, KQ = X_bits.shape
    K = KQ + KE
    QX = np.zeros((M, K // 2), np.uint8)
    SFX = np.full(sf_alloc_bytes(M, K), sf_fill, np.uint8)
    rc = lib().arcq_o_rmsnorm_quantize_x(
        _ptr(X_bits), _ptr(Wn_bits), float(eps), _ptr(idx), M, KQ, KE, variant, _ptr(QX), _ptr(SFX)
    )
    if rc:
        raise ValueError(f"oracle rmsnorm_quantize_x: bad shape M={M} KQ={KQ} KE={KE}")
    return QX, SFX


# semantics switches of arcq_o_quantize_sem (TEST-ONLY; see the header of arcq_oracle.c)
SEM_TIE_FIRSTMIN, SEM_DIV_TRUE, SEM_SCALE_FAKE, SEM_RESID_F32 = 1, 2, 4, 8
SEM_KERNEL = 0                    # the kernel text: what arcq_o_quantize_{x,w} compute
SEM_FAKE = 15                     # every rule as the reference's Python fake path has it
FLOOR_KERNELS_FAKE = (1.0 / 512, 0.0)     # kernels/fake.py:21,23   (scale floor, log2 epsilon)
FLOOR_MODEL_QUANTIZE = (2e-3, 1e-9)       # model/quantize.py:41,43


def quantize_sem(X_f32, KE, variant=G16, is_weight=False, flags=SEM_KERNEL, floor=FLOOR_KERNELS_FAKE):
    """arcq_o_quantize_sem: X fp32 [rows, KQ] already in reordered order -> (Q packed [rows,K/2], SFf fp32 [rows,K/16]
    by group position, DQ fp32 [rows,K] by position).  flags=SEM_KERNEL equals quantize_{x,w} on bf16-valued input."""
    X = _c(X_f32, np.float32)
    rows, KQ = X.shape
    K = KQ + KE
    Q = np.zeros((rows, K // 2), np.uint8)
    SFf = np.zeros((rows, K // 16), np.float32)
    DQ = np.zeros((rows, K), np.float32)
    rc = lib().arcq_o_quantize_sem(_ptr(X), rows, KQ, KE, variant, int(bool(is_weight)), int(flags), float(floor[0]), float(floor[1]),
                                   _ptr(Q), _ptr(SFf), _ptr(DQ))
    if rc:
        raise ValueError(f"oracle quantize_sem: bad shape rows={rows} KQ={KQ} KE={KE} variant={variant}")
    return Q, SFf, DQ


def fake_layout(DQ, KQ, KE, variant=G16):
    """Position-ordered [rows, KQ+KE] -> the fake path's [primaries in channel order | residual/duplicate groups in order]
    (model/quantize.py:241,268) for an identity permutation."""
    DQ = np.asarray(DQ)
    prim = np.concatenate([DQ[:, 16 * primary_pos(g, KQ, KE, variant):][:, :16] for g in range(KQ // 16)], axis=1)
    if not KE:
        return prim
    res = np.concatenate([DQ[:, 16 * residual_pos(g, KQ, KE, variant):][:, :16] for g in range((KQ - KE) // 16, KQ // 16)], axis=1)
    return np.concatenate([prim, res], axis=1)


def dequant(Q, SF, K=None):
    """Format-spec dequantisation -> fp32 [rows, K]."""
    Q = _c(Q, np.uint8)
    SF = _c(SF, np.uint8)
    rows = Q.shape[0]
    K = Q.shape[1] * 2 if K is None else K
    out = np.empty((rows, K), np.float32)
    lib().arcq_o_dequant(_ptr(Q), _ptr(SF), rows, K, _ptr(out))
    return out


def gemm(A, B, SFA, SFB, alpha, want_abs=False):
    """matmul (bindings.cpp:99-120): returns (D bf16 bits [M,N], D exact fp64 [M,N][, sum|a*b|])."""
    A = _c(A, np.uint8)
    B = _c(B, np.uint8)
    SFA = _c(SFA, np.uint8)
    SFB = _c(SFB, np.uint8)
    M, N, K = A.shape[0], B.shape[0], A.shape[1] * 2
    assert B.shape[1] * 2 == K
    Db = np.empty((M, N), np.uint16)
    De = np.empty((M, N), np.float64)
    Da = np.empty((M, N), np.float64) if want_abs else None
    rc = lib().arcq_o_gemm(_ptr(A), _ptr(B), _ptr(SFA), _ptr(SFB), M, N, K, float(alpha), _ptr(Db), _ptr(De), _ptr(Da))
    if rc:
        raise ValueError(f"oracle gemm: bad K={K}")
    return (Db, De, Da) if want_abs else (Db, De)
