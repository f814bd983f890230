"""The exact pin of the kernel-text oracle against the reference's own outputs.

The reference's CUDA kernels cannot run here and hold no fixtures; its Python fake-quant path can, and its outputs on
seeded inputs are committed under tests/golden/ (made by tests/golden/make_golden.py from the imported reference
functions).  The C oracle's group/row code has four switches (ARCQ_SEM_*, test-only) that replace exactly the four rules
on which the kernel TEXT and the fake path differ.  This file shows

  1. with all four switched to the fake path's rule the oracle reproduces the reference outputs BIT FOR BIT -- at the
     headline size KQ=4096, KE=64 with identity reorder (x: primaries + quantised residuals, w: primaries + duplicates) and
     on the small single-tensor fixture, for both scale-floor flavours (kernels/fake.py, model/quantize.py).  That pins
     everything the two share: the 16-channel block partition, amax, the /6, the e2m1 grid, which channels get a
     residual / duplicate and where they go, and the dequantisation;
  2. with no switch set the same entry point produces the very bytes of arcq_o_quantize_{x,w} (the byte oracle of the
     GPU tests), so (1) is a statement about the code the kernels are checked against;
  3. every element on which kernel-text mode differs from the reference output is explained by exactly the rule that
     causes it, each recognised by a predicate evaluated HERE in numpy, independently of the oracle:
        S  block scale below 2^-6: e4m3 subnormal grid (reorder.cu:100,138) vs floor + finer 3-bit-mantissa grid (fake.py:20-30)
        D  x * (float)(1.0 / s) (reorder.cu:146,153) and x / s (fake.py:51) round to different fp32 quotients
        T  the quotient sits exactly on a tie whose RNE-to-even-code result (reorder.cu:98) differs from the first-minimum
           argmin (fake.py:13-15)
        R  (residual groups) the kernel rounds x - q*S to bf16 (reorder.cu:157-160), the fp32 fake path does not; or the
           group's primary codes already differ for a reason above (cascade)
     Zero unexplained mismatches; the old statistical bound (< 1.2 % of elements) is gone.
"""
import numpy as np
import pytest
import torch

from oracle import fake_quant as FQ
from oracle import oracle as O
from tests.util import from_bits

KERNEL, FAKE = O.SEM_KERNEL, O.SEM_FAKE


def _f32(bits_u32):
    return np.ascontiguousarray(bits_u32).view(np.float32)


def _bf16_as_f32(bits_u16):
    return O.bf16_bits_to_f32(bits_u16)


def _same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


# ------------------------------------------------------------------------------------------------ 1. bit-for-bit
@pytest.mark.parametrize("name,is_w", [("x", False), ("w", True)])
def test_fake_semantics_reproduce_reference_arc_outputs_bit_for_bit(golden, name, is_w):
    g = golden("fake_arc_identity_4096.npz")
    KQ, KE = (int(v) for v in g["meta"])
    x = _bf16_as_f32(g[f"{name}_in"])
    want = _f32(g[f"{name}_q_fp32"])                      # model/quantize.py:219-268 run on this tensor in fp32
    _, _, dq = O.quantize_sem(x, KE, O.G16, is_weight=is_w, flags=FAKE, floor=O.FLOOR_MODEL_QUANTIZE)
    got = O.fake_layout(dq, KQ, KE)
    assert got.shape == want.shape == (x.shape[0], KQ + KE)
    assert _same_bits(got, want), f"{(got.view(np.uint32) != want.view(np.uint32)).sum()} elements differ"


@pytest.mark.parametrize("flavour,key", [(O.FLOOR_KERNELS_FAKE, "fake_fp32"), (O.FLOOR_MODEL_QUANTIZE, "model_fp32")])
def test_fake_semantics_reproduce_reference_single_tensor_bit_for_bit(golden, flavour, key):
    g = golden("fake_nvfp4_tensor.npz")
    x = _f32(g["in_fp32"])                                # arbitrary fp32 values, incl. a zero block, ties, an outlier
    want = _f32(g[key])                                   # kernels/fake.py:34-62 / model/quantize.py:65-92
    _, _, dq = O.quantize_sem(x, 0, O.G16, flags=FAKE, floor=flavour)
    assert _same_bits(dq, want)


def test_python_port_matches_reference_in_the_callers_dtype(golden):
    """The bf16 run (the dtype the reference's callers use) pins oracle/fake_quant.py, the CPU baseline, at the headline size."""
    g = golden("fake_arc_identity_4096.npz")
    KQ, KE = (int(v) for v in g["meta"])
    ident = torch.arange(KQ)
    qx, _, sx = FQ.fake_arc_x(from_bits(g["x_in"]).clone(), ident, KE)
    qw, _, sw = FQ.fake_arc_w(from_bits(g["w_in"]).clone(), ident, KE)
    assert float(sx) == 1.0 and float(sw) == 1.0
    assert np.array_equal(qx.view(torch.int16).numpy().view(np.uint16), g["x_q_bf16"])
    assert np.array_equal(qw.view(torch.int16).numpy().view(np.uint16), g["w_q_bf16"])


# ------------------------------------------------------------------------------------------------ 2. same code as the byte oracle
@pytest.mark.parametrize("variant", [O.G16, O.G32])
@pytest.mark.parametrize("is_w", [False, True])
def test_kernel_semantics_of_the_switchable_entry_equal_the_byte_oracle(golden, variant, is_w):
    g = golden("fake_arc_identity_4096.npz")
    KQ, KE = (int(v) for v in g["meta"])
    xb = g["w_in" if is_w else "x_in"]
    idx = np.arange(KQ, dtype=np.int16)
    q_ref, sf_ref = (O.quantize_w if is_w else O.quantize_x)(xb, idx, KE, variant)
    q, sff, dq = O.quantize_sem(_bf16_as_f32(xb), KE, variant, is_weight=is_w, flags=KERNEL)
    assert np.array_equal(q, q_ref)
    assert _same_bits(dq, O.dequant(q_ref, sf_ref))
    K = KQ + KE
    for r in range(xb.shape[0]):
        for p in range(0, K // 16, 7):
            assert sff[r, p] == O.ue4m3_decode(int(sf_ref[O.sf_offset(r, p, K)]))


# ------------------------------------------------------------------------------------------------ 3. every difference has its cause
# quotients on which RNE-to-even-code and first-minimum argmin disagree (fake grid ascending: the tie goes to the smaller value)
TIES_DIFFER = np.array([0.75, 1.75, 3.5, -0.25, -1.25, -2.5, -5.0], np.float32)


def _causes_for_group_inputs(v, sdec_k, sdec_f):
    """v [n,16] fp32 group inputs, per-group decoded scales in kernel / fake mode -> boolean masks S [n], D [n,16], T [n,16]."""
    amax = np.abs(v).max(1)
    s_raw = (amax / np.float32(6.0)).astype(np.float32)
    S = s_raw < np.float32(2.0 ** -6)                     # only there do the two scale rules pick different values
    s = sdec_k.astype(np.float32)[:, None]
    rcp = (1.0 / s.astype(np.float64)).astype(np.float32)  # (float)(1.0 / (double)s), reorder.cu:146
    t_mul = (v * rcp).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_div = (v / s).astype(np.float32)
    D = t_mul != t_div
    T = np.isin(t_mul, TIES_DIFFER) | np.isin(t_div, TIES_DIFFER)
    return S, D, T, (sdec_k != sdec_f)


@pytest.mark.parametrize("name,is_w", [("x", False), ("w", True)])
def test_every_difference_to_the_reference_output_has_a_named_cause(golden, name, is_w):
    g = golden("fake_arc_identity_4096.npz")
    KQ, KE = (int(v) for v in g["meta"])
    x = _bf16_as_f32(g[f"{name}_in"])
    rows, G, P = x.shape[0], KQ // 16, (KQ - KE) // 16
    want = _f32(g[f"{name}_q_fp32"])
    qk, sfk, dqk = O.quantize_sem(x, KE, O.G16, is_weight=is_w, flags=KERNEL)
    qf, sff, dqf = O.quantize_sem(x, KE, O.G16, is_weight=is_w, flags=FAKE, floor=O.FLOOR_MODEL_QUANTIZE)
    got_k = O.fake_layout(dqk, KQ, KE)
    assert _same_bits(O.fake_layout(dqf, KQ, KE), want)    # (the fake-mode side IS the reference output)
    ppos = np.array([O.primary_pos(gi, KQ, KE, O.G16) for gi in range(G)])
    rpos = np.array([O.residual_pos(gi, KQ, KE, O.G16) for gi in range(P, G)])

    # ---- primaries: [rows, G, 16]
    v = x.reshape(rows * G, 16)
    sk, sf = sfk[:, ppos].reshape(-1), sff[:, ppos].reshape(-1)
    S, D, T, scale_differs = _causes_for_group_inputs(v, sk, sf)
    mism = (got_k[:, :KQ] != want[:, :KQ]).reshape(rows * G, 16)       # numeric compare: -0.0 == +0.0
    assert not np.any(scale_differs & ~S), "a block scale differs outside the e4m3-subnormal range"
    unexplained = mism & ~S[:, None] & ~D & ~T
    assert not unexplained.any(), f"{unexplained.sum()} primary elements differ without a named cause"
    # the predicates are tight, not blanket excuses: where scale and quotient agree, a tie of the differing kind ALWAYS shows
    pure_tie = T & ~D & ~S[:, None] & (np.abs(v) > 0)
    assert np.all(mism[pure_tie]), "a differing-tie quotient did not produce a difference"
    # outside S and D blocks nothing but ties differs, and the outputs are within one e2m1 step of each other
    counts = {"S": int((mism & S[:, None]).sum()), "D": int((mism & ~S[:, None] & D).sum()), "T": int((mism & ~S[:, None] & ~D & T).sum())}
    assert counts["T"] > 0 and counts["S"] > 0, counts          # the fixture really exercises the rules
    prim_group_mism = mism.any(1).reshape(rows, G)

    # ---- residual groups (x) / duplicates (w)
    tail_k, tail_w = got_k[:, KQ:].reshape(rows, G - P, 16), want[:, KQ:].reshape(rows, G - P, 16)
    tmism = tail_k != tail_w
    if is_w:
        # duplicates: exactly the primary's differences again (reorder.cu:306-316 / model/quantize.py:241)
        assert np.array_equal(tmism, mism.reshape(rows, G, 16)[:, P:])
        return
    # the residual inputs of both modes, rebuilt here from the primaries: r = x - q*s (fp32), the kernel rounds it to bf16
    xt = x.reshape(rows, G, 16)[:, P:]
    prim_k = got_k[:, :KQ].reshape(rows, G, 16)[:, P:]
    prim_f = want[:, :KQ].reshape(rows, G, 16)[:, P:]
    r_f = (xt - prim_f).astype(np.float32)
    r_k_unrounded = (xt - prim_k).astype(np.float32)
    r_k = O.bf16_bits_to_f32(O.f32_to_bf16_bits(r_k_unrounded))
    R = (r_k != r_f).any(2)                                            # bf16 rounding changed an input of the group, or cascade
    cascade = prim_group_mism[:, P:]
    Sr, Dr, Tr, sdiff_r = _causes_for_group_inputs(r_k.reshape(-1, 16), sfk[:, rpos].reshape(-1), sff[:, rpos].reshape(-1))
    explained_group = (R | cascade).reshape(-1)
    tm = tmism.reshape(-1, 16)
    unexplained = tm & ~explained_group[:, None] & ~Sr[:, None] & ~Dr & ~Tr
    assert not unexplained.any(), f"{unexplained.sum()} residual elements differ without a named cause"
    assert not np.any(sdiff_r & ~Sr & ~explained_group), "a residual scale differs without a cause"
    # and the converse on the clean groups: same inputs, no S / D / T  =>  identical output
    clean = ~explained_group & ~Sr & ~(Dr | Tr).any(1)
    assert clean.sum() > 0 and not tm[clean].any()


def test_single_switch_runs_isolate_each_rule(golden):
    """Each switch alone changes only elements its predicate covers; all four together give the reference output."""
    g = golden("fake_arc_identity_4096.npz")
    KQ, KE = (int(v) for v in g["meta"])
    x = _bf16_as_f32(g["x_in"])
    G = KQ // 16
    ppos = np.array([O.primary_pos(gi, KQ, KE, O.G16) for gi in range(G)])
    _, sfk, dqk = O.quantize_sem(x, KE, O.G16, flags=KERNEL)
    base = O.fake_layout(dqk, KQ, KE)[:, :KQ].reshape(-1, 16)
    v = x.reshape(-1, 16)
    S, D, T, _ = _causes_for_group_inputs(v, sfk[:, ppos].reshape(-1), sfk[:, ppos].reshape(-1))
    for flag, allowed in ((O.SEM_TIE_FIRSTMIN, T), (O.SEM_DIV_TRUE, D), (O.SEM_SCALE_FAKE, np.broadcast_to(S[:, None], D.shape))):
        _, _, dq = O.quantize_sem(x, KE, O.G16, flags=flag, floor=O.FLOOR_MODEL_QUANTIZE)
        got = O.fake_layout(dq, KQ, KE)[:, :KQ].reshape(-1, 16)
        changed = got != base
        assert changed.any(), flag
        assert not np.any(changed & ~allowed), f"switch {flag} changed elements outside its predicate"
    _, _, dq = O.quantize_sem(x, KE, O.G16, flags=O.SEM_RESID_F32)
    assert np.array_equal(O.fake_layout(dq, KQ, KE)[:, :KQ].reshape(-1, 16), base)     # primaries untouched by the residual switch
