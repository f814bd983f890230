"""INTEGRATION.md section 2 shows the file a maintainer drops into the reference as ``kernels/build/agemm.py``.
These tests EXECUTE that text verbatim (only the placeholder library path is substituted), so the document
cannot drift from the C-ABI: on CPU the stub must load the library, bind every entry point it names with an
argument list of the right length, and answer the host-only helpers; on the GPU its four functions must
return the same bytes as ``arcquant_amd.agemm``.
"""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from arcquant_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_module():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# kernels/build/agemm\.py.*?)```", text, flags=re.S)
    assert m, "INTEGRATION.md no longer holds the kernels/build/agemm.py stub"
    src = m.group(1)
    assert "/path/to/arcquant_amd/lib/libarcq_hip.so" in src
    src = src.replace("/path/to/arcquant_amd/lib/libarcq_hip.so", _lib.LIB_PATH)
    mod = types.ModuleType("agemm_stub")
    exec(compile(src, "INTEGRATION.md:agemm.py", "exec"), mod.__dict__)
    return mod


def test_stub_executes_and_binds_the_declared_abi():
    stub = _stub_module()
    for name in ("matmul", "reorder_quantize_x", "reorder_quantize_w", "rmsnorm_quantize_x"):     # bindings.cpp:551-575
        assert callable(getattr(stub, name))
    # host-only helpers answer without a GPU and agree with the package's own binding
    L = _lib.lib()
    for rows, K in [(1, 4160), (128, 128), (4096, 4160), (300, 19008)]:
        assert stub._L.arcq_sf_alloc_bytes(rows, K) == L.arcq_sf_alloc_bytes(rows, K)
    for kq in (1024, 3584, 4096, 18944, 28672):
        assert stub._L.arcq_variant_for_kq(kq) == L.arcq_variant_for_kq(kq)
    assert stub._L.arcq_gemm_workspace_bytes(4096, 4096, 4160) == 0
    # every entry point the stub declares argtypes for takes exactly as many arguments as include/arcq.h declares
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "arcq.h")).read(), flags=re.S)
    for name in ("arcq_quantize_x", "arcq_quantize_w", "arcq_rmsnorm_quantize_x", "arcq_gemm_nvfp4", "arcq_sf_alloc_bytes",
                 "arcq_variant_for_kq", "arcq_gemm_workspace_bytes"):
        decl = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, hdr, flags=re.S)
        assert decl, name
        nargs = len([a for a in decl.group(1).split(",") if a.strip() and a.strip() != "void"])
        assert len(getattr(stub._L, name).argtypes) == nargs, name
    # a shape error surfaces as RuntimeError with the library's message (the reference throws std::runtime_error)
    with pytest.raises(RuntimeError, match="KQ"):
        stub._ck(stub._L.arcq_quantize_x(None, None, None, None, 4, 40, 0, 0, None), "reorder_quantize")


@pytest.mark.gpu
def test_stub_results_equal_the_package_binding():
    from arcquant_amd import agemm
    stub = _stub_module()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    M, N, KQ, KE = 5, 192, 2048, 64
    x = (torch.randn(M, KQ, generator=g) * 40).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, KQ, generator=g) * 40).to(torch.bfloat16).to(dev)
    wn = (torch.rand(KQ, generator=g) + 0.5).to(torch.bfloat16).to(dev)
    idx = torch.randperm(KQ, generator=g).to(torch.int16).to(dev)
    for a, b in ((stub.reorder_quantize_x(x, idx, KE), agemm.reorder_quantize_x(x, idx, KE)),
                 (stub.reorder_quantize_w(w, idx, KE), agemm.reorder_quantize_w(w, idx, KE)),
                 (stub.rmsnorm_quantize_x(x, wn, 1e-6, idx, KE), agemm.rmsnorm_quantize_x(x, wn, 1e-6, idx, KE))):
        assert torch.equal(a[0], b[0])
        used = _lib.lib().arcq_sf_used_bytes(a[0].shape[0], KQ + KE)
        # rows beyond the tensor inside the last 128-row tile stay unwritten (torch.empty): compare written rows through a GEMM below
        assert a[1].numel() == b[1].numel() >= used
    A, SFA = stub.reorder_quantize_x(x, idx, KE)
    B, SFB = stub.reorder_quantize_w(w, idx, KE)
    scale = torch.tensor(0.0125, dtype=torch.float32, device=dev)
    d_stub_dev = stub.matmul(A, B, SFA, SFB, scale)
    d_stub_host = stub.matmul(A, B, SFA, SFB, 0.0125)
    A2, SFA2 = agemm.reorder_quantize_x(x, idx, KE)
    B2, SFB2 = agemm.reorder_quantize_w(w, idx, KE)
    d_pkg = agemm.matmul(A2, B2, SFA2, SFB2, 0.0125)
    assert torch.equal(d_stub_dev, d_pkg) and torch.equal(d_stub_host, d_pkg)
    assert np.isfinite(d_pkg.float().cpu().numpy()).all()
