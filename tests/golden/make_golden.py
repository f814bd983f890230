#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Run ONLY in the build container (it needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

Two families of fixtures:

* ``fake_*.npz``  -- inputs and outputs of the REFERENCE's importable Python fake-quant functions
  (kernels/fake.py::quantize_nvfp4_tensor, model/quantize.py::fake_reorder_quantize_{x,w}), executed
  here on CPU.  They pin oracle/fake_quant.py bit-for-bit and bound the disagreement between the
  fake path and the kernel-text oracle.  model/quantize.py does ``import agemm`` at import time; an
  empty stand-in module object is registered for that name (nothing of it is ever called).
* ``oracle_*.npz`` -- outputs of our kernel-text restatement (oracle/arcq_oracle.c) on seeded inputs.
  The reference holds no golden vectors for its CUDA kernels and they cannot run here, so these are
  regression pins of the restatement ("parity unpinned" w.r.t. the CUDA binary), not reference outputs.

Only data (inputs / expected outputs) is written; no reference source text is stored.
"""
from __future__ import annotations

import importlib.util
import sys as _sys
_sys.dont_write_bytecode = True
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def bits(t: torch.Tensor) -> np.ndarray:
    """torch tensor -> raw bit patterns as an unsigned numpy array (bf16/fp16 -> uint16, fp32 -> uint32)."""
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    return t.contiguous().view(torch.int32).numpy().view(np.uint32).copy()


def outlier_activations(M, K, seed, dtype=torch.bfloat16):
    """The structured-outlier recipe of kernels/main.py:13-19 (scaled to K), on CPU."""
    g = torch.Generator().manual_seed(seed)
    ks, ko = max(16, K * 384 // 4096), max(16, K * 128 // 4096)
    signs = torch.randint(0, 2, (M, K), generator=g).to(dtype) * 2 - 1
    x = torch.rand(M, K, generator=g).to(dtype) * 3
    x[:, -ks:] = torch.rand(M, ks, generator=g).to(dtype) * 3 + 3
    x[:, -ko:] = torch.rand(M, ko, generator=g).to(dtype) * 8 + 8
    x[:, -16:] = torch.rand(M, 16, generator=g).to(dtype) * 32 + 32
    return x * signs


def main():
    assert os.path.isdir(REF), "run in the build container: /root/reference is required"
    fake = _load("ref_fake", os.path.join(REF, "kernels/fake.py"))
    sys.modules.setdefault("agemm", types.ModuleType("agemm"))
    quant = _load("ref_quantize", os.path.join(REF, "model/quantize.py"))

    # ---------------------------------------------------------------- fake path, single tensor
    out = {}
    cases = [("bf16", torch.bfloat16), ("fp16", torch.float16), ("fp32", torch.float32)]
    for name, dt in cases:
        g = torch.Generator().manual_seed(45510)
        t = (torch.randn(48, 256, generator=g) * 2.5).to(dt)
        t[0, :16] = 0                      # an all-zero block (scale := 1e-9 branch)
        t[1, :16] = 1e-4                   # below the scale floor
        t[2, 5] = 300.0                    # one outlier
        t[3, :8] = torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 6.0]).to(dt)  # e2m1 ties at scale 1
        t[3, 8:16] = -t[3, :8]
        out[f"in_{name}"] = bits(t)
        out[f"fake_{name}"] = bits(fake.quantize_nvfp4_tensor(t.clone()))
        out[f"model_{name}"] = bits(quant.quantize_nvfp4_tensor(t.clone()))
    np.savez_compressed(os.path.join(HERE, "fake_nvfp4_tensor.npz"), **out)

    # ---------------------------------------------------------------- fake ARC x / w
    out = {}
    for KE in (0, 64):
        M, N, K = 24, 40, 256
        x = outlier_activations(M, K, 7)
        g = torch.Generator().manual_seed(11)
        w = (torch.rand(N, K, generator=g) * 3).to(torch.bfloat16)
        perm = torch.randperm(K, generator=g)
        qx, ax, sx = quant.fake_reorder_quantize_x(x.clone(), perm, KE)
        qw, aw, sw = quant.fake_reorder_quantize_w(w.clone(), perm, KE)
        out[f"x_KE{KE}"] = bits(x)
        out[f"w_KE{KE}"] = bits(w)
        out[f"perm_KE{KE}"] = perm.numpy().astype(np.int64)
        out[f"qx_KE{KE}"] = bits(qx)
        out[f"qw_KE{KE}"] = bits(qw)
        out[f"sx_KE{KE}"] = np.float32(sx.item())
        out[f"sw_KE{KE}"] = np.float32(sw.item())
        out[f"ax_KE{KE}"] = bits(ax)
        out[f"aw_KE{KE}"] = bits(aw)
    np.savez_compressed(os.path.join(HERE, "fake_arc_xw.npz"), **out)

    # ---------------------------------------------------------------- fake ARC x / w at the headline size, IDENTITY permutation
    # The exact pin of the kernel-text oracle (tests/test_oracle_pin.py).  With reorder_index = identity the fake path's
    # 16-channel blocks are the kernel's groups, so outputs are comparable element by element.  The inputs are bf16 values
    # already carrying the callers' per-tensor pre-scale with max(x) == 2688 exactly, so the scale the reference functions
    # compute themselves (signed max / 2688) is exactly 1.0 and they quantise precisely these values.  They run in fp32
    # (the kernels compute in fp32 on bf16 inputs) and, for the port, in bf16 (the dtype the reference's callers use).
    out = {}
    KQ, KE = 4096, 64
    ident = torch.arange(KQ)

    def prescaled(t):                       # t / (max|t| / 2688) in bf16, largest magnitude made positive
        t = t.clone()
        i = torch.argmax(t.abs())
        t.view(-1)[i] = t.view(-1)[i].abs()
        t = (t / (t.abs().max().float() / 2688.0)).to(torch.bfloat16)
        assert float(t.max()) == 2688.0 and float(t.abs().max()) == 2688.0
        return t

    ties = torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 6.0])

    def engineer(t, g):                     # rows that force each rule the kernel text and the fake path disagree on
        r0 = t.shape[0] - 4
        # (1) exact e2m1 ties, both signs: block amax = 6 * 2^k so the block scale is exactly 2^k
        for b, k in enumerate((-3, 0, 2, 5)):
            for blk in (3 + 7 * b, KQ // 16 - 1 - b):          # a plain block and one inside the residual tail
                v = torch.cat([ties, -ties]) * (2.0 ** k)
                t[r0, 16 * blk:16 * blk + 16] = v[torch.randperm(16, generator=g)].to(torch.bfloat16)
        # (2) block scales below 2^-6 (e4m3 subnormals in the kernel, finer grid in the fake path), below both floors, zero
        t[r0 + 1] = (torch.rand(KQ, generator=g) * 0.09 * (torch.randint(0, 2, (KQ,), generator=g) * 2 - 1)).to(torch.bfloat16)
        t[r0 + 1, 64:128] = (torch.rand(64, generator=g) * 0.011).to(torch.bfloat16)      # amax/6 < 2e-3
        t[r0 + 1, 128:192] = (torch.rand(64, generator=g) * 0.004).to(torch.bfloat16)     # amax/6 < 1/512
        t[r0 + 1, 192:208] = 0
        t[r0 + 1, -32:-16] = (torch.rand(16, generator=g) * 0.011).to(torch.bfloat16)
        t[r0 + 1, -16:] = 0
        # (3) wide dynamic range inside blocks (many quotients near rounding boundaries)
        t[r0 + 2] = (torch.randn(KQ, generator=g) * 300).clamp(-2600, 2600).to(torch.bfloat16)
        t[r0 + 3] = (torch.randn(KQ, generator=g).exp() * 20 * (torch.randint(0, 2, (KQ,), generator=g) * 2 - 1)).clamp(-2600, 2600).to(torch.bfloat16)
        t[0, 7] = 2688.0                    # the engineered rows may have replaced the tensor's maximum
        return t

    g = torch.Generator().manual_seed(4096)
    x = engineer(prescaled(outlier_activations(12, KQ, 4096)), g)
    w = engineer(prescaled((torch.rand(10, KQ, generator=g) * 3).to(torch.bfloat16)), g)
    assert float(x.max()) == 2688.0 and float(w.max()) == 2688.0
    for name, t, fn in (("x", x, quant.fake_reorder_quantize_x), ("w", w, quant.fake_reorder_quantize_w)):
        q32, a32, s32 = fn(t.float().clone(), ident, KE)
        q16, a16, s16 = fn(t.clone(), ident, KE)
        assert float(s32) == 1.0 and float(s16) == 1.0
        out[f"{name}_in"] = bits(t)
        out[f"{name}_q_fp32"] = bits(q32)
        out[f"{name}_q_bf16"] = bits(q16)
    out["meta"] = np.array([KQ, KE], np.int64)
    np.savez_compressed(os.path.join(HERE, "fake_arc_identity_4096.npz"), **out)

    # ---------------------------------------------------------------- oracle regression pins
    out = {}
    for tag, (M, KQ, KE, variant) in {
        "g16_a": (3, 256, 64, O.G16),
        "g16_b": (130, 256, 0, O.G16),
        "g16_c": (1, 4096, 64, O.G16),
        "g32_a": (3, 256, 64, O.G32),
        "g32_b": (5, 3584, 64, O.G32),
    }.items():
        x = outlier_activations(M, KQ, 100 + M)
        x = x / (x.abs().max().float() / 2688.0)           # per-tensor pre-scale as the callers do
        g = torch.Generator().manual_seed(KQ + KE)
        perm = torch.randperm(KQ, generator=g).to(torch.int16)
        xb, pb = bits(x.to(torch.bfloat16)), perm.numpy()
        qx, sfx = O.quantize_x(xb, pb, KE, variant, sf_fill=0)
        qw, sfw = O.quantize_w(xb, pb, KE, variant, sf_fill=0)
        out[f"{tag}_meta"] = np.array([M, KQ, KE, variant], np.int64)
        out[f"{tag}_x"] = xb
        out[f"{tag}_idx"] = pb
        out[f"{tag}_qx"], out[f"{tag}_sfx"] = qx, sfx
        out[f"{tag}_qw"], out[f"{tag}_sfw"] = qw, sfw
    # rmsnorm + GEMM on one small case
    M, N, KQ, KE = 4, 24, 2048, 64
    x = outlier_activations(M, KQ, 5)
    g = torch.Generator().manual_seed(99)
    wn = (torch.rand(KQ, generator=g) + 0.5).to(torch.bfloat16)
    perm = torch.randperm(KQ, generator=g).to(torch.int16).numpy()
    qx, sfx = O.rmsnorm_quantize_x(bits(x), bits(wn), 1e-6, perm, KE, O.G16, sf_fill=0)
    w = (torch.rand(N, KQ, generator=g) * 3).to(torch.bfloat16)
    qw, sfw = O.quantize_w(bits(w), perm, KE, O.G16, sf_fill=0)
    db, de = O.gemm(qx, qw, sfx, sfw, 0.0123)
    out.update(rms_x=bits(x), rms_wn=bits(wn), rms_idx=perm, rms_qx=qx, rms_sfx=sfx, rms_w=bits(w), rms_qw=qw,
               rms_sfw=sfw, rms_d_bf16=db, rms_d_exact=de, rms_meta=np.array([M, N, KQ, KE], np.int64))
    np.savez_compressed(os.path.join(HERE, "oracle_pins.npz"), **out)

    # ---------------------------------------------------------------- layout tables (tiny, human-checkable)
    out = {}
    for K in (64, 128, 4160):
        out[f"sf_off_K{K}"] = np.array([[O.sf_offset(r, p, K) for p in range(K // 16)] for r in range(256)], np.int64)
    for variant, vn in ((O.G16, "g16"), (O.G32, "g32")):
        for KE in (0, 64, 256):
            KQ = 256
            out[f"pos_{vn}_KE{KE}"] = np.array(
                [[O.primary_pos(g_, KQ, KE, variant), O.residual_pos(g_, KQ, KE, variant)] for g_ in range(KQ // 16)],
                np.int64)
    np.savez_compressed(os.path.join(HERE, "layout_tables.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
