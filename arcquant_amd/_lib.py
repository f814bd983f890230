"""ctypes binding of libarcq_hip.so (the C-ABI declared in include/arcq.h).

The product path has NO fallback: if the HIP library is missing or fails to load, every operator
raises.  (The CPU oracle under ``oracle/`` is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARCQ_HIP_LIB", os.path.join(_HERE, "lib", "libarcq_hip.so"))

OK = 0
VARIANT_G16 = 0
VARIANT_G32 = 1
OUT_BF16 = 0
OUT_F32 = 1

# every symbol include/arcq.h declares: (name, restype, argtypes)
_p, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
SYMBOLS = {
    "arcq_abi_version": (_i32, []),
    "arcq_last_error": (ctypes.c_char_p, []),
    "arcq_variant_for_kq": (_i32, [_i64]),
    "arcq_sf_alloc_bytes": (_i64, [_i64, _i64]),
    "arcq_sf_used_bytes": (_i64, [_i64, _i64]),
    "arcq_sf_offset": (_i64, [_i64, _i64, _i64]),
    "arcq_primary_pos": (_i64, [_i64, _i64, _i64, _i32]),
    "arcq_residual_pos": (_i64, [_i64, _i64, _i64, _i32]),
    "arcq_quantize_x": (_i32, [_p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "arcq_quantize_w": (_i32, [_p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "arcq_rmsnorm_quantize_x": (_i32, [_p, _p, _f32, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "arcq_gemm_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "arcq_gemm_nvfp4": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _f32, _p, _p, _p, _i32, _p, _i64, _p]),
    "arcq_absmax_scale": (_i32, [_p, _i64, _p, _p]),
    "arcq_quantize_x_dyn": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "arcq_silu_mul_quantize_x_dyn": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _p]),
    "arcq_gemm_silu_mul_slots": (_i64, [_i64, _i64, _i64]),
    "arcq_gemm_nvfp4_silu_mul": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f32, _p, _p, _p]),
    "arcq_quantize_x_dyn_slots": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _p]),
    "arcq_repacked_w_bytes": (_i64, [_i64, _i64]),
    "arcq_repacked_sf_bytes": (_i64, [_i64, _i64]),
    "arcq_gemm_repacked_supported": (_i32, [_i64, _i64, _i64]),
    "arcq_gemm_nvfp4_repacked": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _f32, _p, _p, _p, _i32, _p]),
    "arcq_gemm_nvfp4_repacked_stream": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _f32, _p, _p, _p, _i32, _p]),
    "arcq_gemm_nvfp4_repacked_silu_absmax": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f32, _p, _p]),
    "arcq_linear_fused_supported": (_i32, [_i32, _i64, _i64, _i64, _i64]),
    "arcq_linear_rmsnorm_repacked": (_i32, [_p, _p, _f32, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _f32, _p, _p, _p, _i32, _p]),
    "arcq_linear_rmsnorm_silu_repacked": (_i32, [_p, _p, _f32, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _f32, _p, _p, _p, _p]),
    "arcq_linear_dynamic_repacked": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i32, _f32, _p, _p, _i32, _p]),
    "arcq_silu_mul_quantize_x_dyn_slots": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _i32, _p]),
}

# include/arcq_harness.h: e2e-harness-only entry points (NOT the drop-in boundary)
HARNESS_SYMBOLS = {
    "arcq_harness_attn_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "arcq_harness_attn_decode": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "arcq_harness_attn_decode_window": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "arcq_harness_rmsnorm": (_i32, [_p, _i64, _p, _p, _i64, _i64, _f32, _p]),
}

_lib = None


class ArcqError(RuntimeError):
    """A C-ABI call returned a negative status (the reference raises RuntimeError too: bindings.cpp:157-160)."""


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"arcquant_amd: HIP library not found at {LIB_PATH}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C arcquant_amd/csrc`. "
                "There is no CPU fallback."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(HARNESS_SYMBOLS.items()):
            fn = getattr(L, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.arcq_abi_version() != 1:
            raise ImportError(f"arcquant_amd: ABI version mismatch ({L.arcq_abi_version()} != 1)")
        _lib = L
    return _lib


def check(status: int, what: str) -> None:
    if status != OK:
        msg = lib().arcq_last_error().decode("utf-8", "replace")
        raise ArcqError(f"{what} failed with status {status}: {msg}")
