"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/arcq.h
declares, and its pure-integer layout helpers agree with the oracle.  No kernel is launched."""
import os
import re

import pytest

from arcquant_amd import _lib
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "arcq.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(arcq_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    assert _declared_symbols() == sorted(_lib.SYMBOLS)
    # the harness-only header (not the drop-in boundary) and its binding table
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "arcq_harness.h")).read(), flags=re.S)
    assert sorted(set(re.findall(r"\b(arcq_[a-z0-9_]+)\s*\(", text))) == sorted(_lib.HARNESS_SYMBOLS)
    assert not set(_lib.HARNESS_SYMBOLS) & set(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    for name in _declared_symbols():
        assert hasattr(L, name), name
    assert L.arcq_abi_version() == 1


def test_layout_helpers_match_oracle():
    L = _lib.lib()
    for rows, K in [(0, 64), (1, 4160), (127, 128), (128, 128), (129, 4160), (4096, 4160), (300, 19008)]:
        assert L.arcq_sf_alloc_bytes(rows, K) == O.sf_alloc_bytes(rows, K)
        assert L.arcq_sf_used_bytes(rows, K) == O.sf_used_bytes(rows, K)
    for K in (64, 4160, 3648):
        for r in (0, 1, 31, 32, 127, 128, 255, 300):
            for p in range(0, K // 16, 3):
                assert L.arcq_sf_offset(r, p, K) == O.sf_offset(r, p, K)
    for variant in (0, 1):
        for KQ, KE in [(256, 0), (256, 64), (256, 256), (3584, 128), (4096, 64)]:
            for g in range(KQ // 16):
                assert L.arcq_primary_pos(g, KQ, KE, variant) == O.primary_pos(g, KQ, KE, variant)
                assert L.arcq_residual_pos(g, KQ, KE, variant) == O.residual_pos(g, KQ, KE, variant)


def test_variant_rule_reproduces_the_reference_dispatch():
    # bindings.cpp:141-160
    g16 = [2048, 3072, 4096, 5120, 8192, 11008, 13824, 14336]
    g32 = [3584, 18944, 27648, 28672]
    L = _lib.lib()
    assert all(L.arcq_variant_for_kq(k) == 0 for k in g16)
    assert all(L.arcq_variant_for_kq(k) == 1 for k in g32)
    assert L.arcq_variant_for_kq(1024) == 0          # TP shard sizes outside the reference's list


def test_bad_arguments_are_rejected_without_a_gpu():
    L = _lib.lib()
    # shape validation happens before any HIP call
    assert L.arcq_quantize_x(None, None, None, None, 4, 40, 0, 0, None) == -1          # KQ % 16
    assert b"KQ" in L.arcq_last_error()
    assert L.arcq_quantize_x(None, None, None, None, 4, 4096, 64, 7, None) == -1       # bad variant
    assert L.arcq_quantize_x(None, None, None, None, 4, 4096, 64, 0, None) == -4       # NULL pointers
    assert L.arcq_quantize_x(None, None, None, None, 0, 4096, 64, 0, None) == 0        # empty input is fine
    assert L.arcq_gemm_nvfp4(None, None, None, None, None, 4, 4, 100, 1.0, None, None, None, 0, None, 0, None) == -1
    assert L.arcq_gemm_nvfp4(None, None, None, None, None, 0, 4, 128, 1.0, None, None, None, 0, None, 0, None) == 0
    assert L.arcq_rmsnorm_quantize_x(None, None, 1e-6, None, None, None, 4, 1024, 0, 0, None) == -2   # outside [2048, 8192]
    with pytest.raises(_lib.ArcqError):
        _lib.check(-1, "demo")


def test_round2_entry_points_reject_bad_arguments_without_a_gpu():
    """The fused decode linears, the contiguous dynamic quantiser, the stream variant of the repacked GEMM and the harness entry
    points validate before any HIP call (status codes of include/arcq.h: -1 shape, -2 unsupported, -4 NULL)."""
    L = _lib.lib()
    P = 4096                                                   # a fake, 16-byte aligned device pointer: never dereferenced on these paths
    fs = L.arcq_linear_fused_supported
    assert fs(1, 4, 3584, 3584, 64) == 1 and fs(2, 4, 3584, 512, 64) == 1 and fs(2, 4, 3584, 18944, 64) == 1
    assert fs(1, 17, 3584, 3584, 64) == 0                      # M > 16
    assert fs(1, 4, 3584, 1024, 64) == 0                       # the RMSNorm source covers the reference's 2048 .. 8192
    assert fs(2, 16, 4096, 18944, 64) == 0                     # the fp16 image of 16 x 19008 does not fit LDS
    assert fs(3, 4, 3584, 3584, 64) == 0                       # unknown source kind
    # arcq_quantize_x_dyn_slots with reorder_index == NULL (identity order)
    q = L.arcq_quantize_x_dyn_slots
    assert q(None, None, None, None, None, None, 0, 4, 3584, 64, 1, None) == -4
    assert q(P, None, None, None, P, P, 8, 4, 3600, 64, 1, None) == -1 and b"KQ%64" in L.arcq_last_error()
    assert q(P, None, None, None, P, P, 8, 0, 3584, 64, 1, None) == 0                    # empty input
    assert q(P, None, None, None, P, P, 8, 4, 3584, 64, 1, None) == -4                    # NULL outputs
    assert q(P + 2, None, P, P, P, P, 8, 4, 3584, 64, 1, None) == -1 and b"aligned" in L.arcq_last_error()
    # fused linears
    lin = L.arcq_linear_rmsnorm_repacked
    assert lin(P, P, 1e-6, P, P, P, P, 17, 3584, 3584, 64, 1, 1.0, None, None, None, 0, None) == -2
    assert lin(P, P, 1e-6, P, P, P, P, 4, 3584, 3584, 64, 1, 1.0, None, None, None, 7, None) == -1      # out_dtype
    assert lin(P, None, 1e-6, P, P, P, P, 4, 3584, 3584, 64, 1, 1.0, None, None, None, 0, None) == -4   # norm weight
    silu = L.arcq_linear_rmsnorm_silu_repacked
    assert silu(P, P, 1e-6, P, P, P, P, P, 4, 3586, 3584, 64, 1, 1.0, None, None, None, None) == -1 and b"N % 4" in L.arcq_last_error()
    assert silu(P, P, 1e-6, P, P, P, P, P, 4, 3584, 3584, 64, 1, 1.0, None, None, P + 2, None) == -1    # act_scatter_index alignment
    assert L.arcq_linear_dynamic_repacked(P, P, P, P, P, None, None, 0, 4, 3584, 3584, 100, 1, 1.0, None, None, 0, None) == -1
    for fn, m_unsupported in ((L.arcq_gemm_nvfp4_repacked, 129), (L.arcq_gemm_nvfp4_repacked_stream, 17)):
        assert fn(P, P, P, P, P, m_unsupported, 4096, 4160, 1.0, None, None, None, 0, None) == -2
        # bias / residual are fetched 8 bytes at a time when N % 4 == 0
        assert fn(P, P, P, P, P, 4, 4096, 4160, 1.0, None, P + 2, None, 0, None) == -1 and b"8-byte" in L.arcq_last_error()
        assert fn(P, P, P, P, P, 4, 4096, 4160, 1.0, None, None, P + 4, 0, None) == -1 and b"8-byte" in L.arcq_last_error()
    # decode batches (16 < M <= 128) take the repacked path where it beats the tiled GEMM (gemm_rowmid.hip, mid_kind): small weights at
    # every M, large ones up to M = 32 (64 while the packed activations fit LDS and the weight is < 64 M elements)
    assert L.arcq_gemm_repacked_supported(17, 4096, 4160) == 1 and L.arcq_gemm_repacked_supported(128, 4096, 4160) == 1
    assert L.arcq_gemm_repacked_supported(64, 37888, 3648) == 0 and L.arcq_gemm_repacked_supported(32, 37888, 3648) == 1
    assert L.arcq_gemm_repacked_supported(32, 3584, 19008) == 1 and L.arcq_gemm_repacked_supported(64, 3584, 19008) == 0
    assert L.arcq_gemm_repacked_supported(64, 10752, 3648) == 1 and L.arcq_gemm_repacked_supported(128, 10752, 3648) == 0
    assert L.arcq_gemm_repacked_supported(129, 256, 256) == 0
    assert lin(P, P, 1e-6, P, P, P, P, 4, 3584, 3584, 64, 1, 1.0, None, P + 2, None, 0, None) == -1 and b"8-byte" in L.arcq_last_error()
    # harness (include/arcq_harness.h)
    assert L.arcq_harness_attn_decode_window(P, P, P, P, P, 4, 28, 1152, 10, 11, None) == -1             # first > pos
    assert L.arcq_harness_attn_decode(P, P, P, P, P, 4, 28, 1152, 1152, None) == -1                       # pos >= Tmax
    assert L.arcq_harness_rmsnorm(P, 3584, P, P, 4, 3587, 1e-6, None) == -1
    assert L.arcq_harness_rmsnorm(P, 3584, P, P, 0, 3584, 1e-6, None) == 0
