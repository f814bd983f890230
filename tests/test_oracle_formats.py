"""CPU tests: scalar formats and layouts of the oracle against independent statements.

* e2m1 ties follow IEEE RNE (0.25->0, 0.75->1, 1.25->1, 1.75->2, 2.5->2, 3.5->4, 5->4).
* ue4m3 is checked against torch's float8_e4m3fn CPU conversion (an implementation we did not write).
* the scale-factor swizzle is checked against a from-first-principles evaluation of the CUTLASS
  layout algebra (atom ((32,4),(16,4)):((16,4),(0,1)), tiled K-major) and the committed tables.
"""
import itertools

import numpy as np
import pytest
import torch

from oracle import oracle as O


# ------------------------------------------------------------------------------------------ e2m1
E2M1_VALUES = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]


def test_e2m1_decode_table():
    for code in range(16):
        v = O.e2m1_decode(code)
        assert abs(v) == E2M1_VALUES[code & 7]
        assert (np.signbit(v)) == bool(code & 8)


@pytest.mark.parametrize("x,want", [(0.25, 0.0), (0.75, 1.0), (1.25, 1.0), (1.75, 2.0), (2.5, 2.0), (3.5, 4.0),
                                    (5.0, 4.0), (0.2500001, 0.5), (0.7499999, 0.5), (5.0000005, 6.0), (6.0, 6.0),
                                    (7.5, 6.0), (1e9, 6.0), (0.0, 0.0), (1e-30, 0.0)])
def test_e2m1_rne_ties_and_saturation(x, want):
    assert O.e2m1_decode(O.e2m1_encode(x)) == want
    assert O.e2m1_decode(O.e2m1_encode(-x)) == -want
    # sign of zero is preserved (cvt.rn.satfinite semantics, oracle assumption A1)
    if want == 0.0:
        assert O.e2m1_encode(-x) == 0x8 and O.e2m1_encode(x) == 0x0


def test_e2m1_is_nearest_everywhere():
    xs = np.linspace(-7, 7, 28001, dtype=np.float32)
    grid = np.array(sorted(set(E2M1_VALUES + [-v for v in E2M1_VALUES])), np.float32)
    for x in xs[::7]:
        got = O.e2m1_decode(O.e2m1_encode(float(x)))
        best = np.min(np.abs(grid - x))
        assert abs(abs(got - x) - best) < 1e-7, (x, got)


# ------------------------------------------------------------------------------------------ ue4m3
def test_ue4m3_decode_matches_torch_all_codes():
    codes = torch.arange(0, 127, dtype=torch.uint8)
    want = codes.view(torch.float8_e4m3fn).float()
    for c, w in zip(codes.tolist(), want.tolist()):
        assert O.ue4m3_decode(c) == w


def test_ue4m3_encode_matches_torch_fp8():
    rng = np.random.default_rng(0)
    # log-uniform over the clamped range [2^-9, 448], plus every midpoint and every grid point
    xs = np.exp2(rng.uniform(-9, np.log2(448), 20000)).astype(np.float32)
    grid = torch.arange(1, 127, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    mids = (grid[:-1] + grid[1:]) / 2
    xs = np.concatenate([xs, grid, mids, np.nextafter(mids, 0).astype(np.float32),
                         np.nextafter(mids, 1e9).astype(np.float32)]).astype(np.float32)
    xs = np.clip(xs, 2.0 ** -9, 448.0)
    want = torch.from_numpy(xs).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = np.array([O.ue4m3_encode(float(x)) for x in xs], np.uint8)
    assert np.array_equal(got, want)


def test_reciprocal_float_division_equals_double_path():
    """r = (float)(1.0/(double)s) (reorder.cu:146) equals the correctly rounded fp32 quotient 1.0f/s for
    every ue4m3 value, so a HIP kernel may use an IEEE fp32 division."""
    for c in range(1, 127):
        s = np.float32(O.ue4m3_decode(c))
        via_double = np.float32(1.0 / np.float64(s))
        via_float = np.float32(1.0) / s
        assert via_double == via_float, c


# ------------------------------------------------------------------------------------------ layouts
def _cutlass_sf_offset(r, p, K):
    """Evaluate the CUTLASS layout directly: atom shape ((32,4),(16,4)) stride ((16,4),(0,1)),
    tile_to_shape(atom, (rows, K), Step<_2,_1>) i.e. atoms laid out K-fastest."""
    k = 16 * p                                   # element column; the SF vector covers 16 columns
    atom_r, atom_k = r // 128, k // 64
    ri, ki = r % 128, k % 64
    inner = (ri % 32) * 16 + (ri // 32) * 4 + (ki // 16) * 1 + (ki % 16) * 0
    return (atom_r * (K // 64) + atom_k) * 512 + inner


@pytest.mark.parametrize("K", [64, 128, 4160, 3648, 19008])
def test_sf_offset_matches_layout_algebra_and_is_bijective(K):
    rows = 300
    offs = np.array([[O.sf_offset(r, p, K) for p in range(K // 16)] for r in range(rows)])
    ref = np.array([[_cutlass_sf_offset(r, p, K) for p in range(K // 16)] for r in range(rows)])
    assert np.array_equal(offs, ref)
    assert len(np.unique(offs)) == offs.size
    assert offs.max() < O.sf_used_bytes(rows, K) <= O.sf_alloc_bytes(rows, K)


def test_sf_sizes_follow_bindings():
    # bindings.cpp:83-95: (rows/128 + 1) * 128 * K / 16, i.e. one spare tile when rows % 128 == 0
    assert O.sf_alloc_bytes(1, 4160) == 128 * 260
    assert O.sf_alloc_bytes(128, 4160) == 2 * 128 * 260
    assert O.sf_used_bytes(128, 4160) == 128 * 260
    assert O.sf_alloc_bytes(4096, 4160) == 1_098_240       # "1 098 240 alloc"
    assert O.sf_used_bytes(4096, 4160) == 1_064_960        # "1 064 960 B used"


def test_layout_tables_golden(golden):
    t = golden("layout_tables.npz")
    for K in (64, 128, 4160):
        want = t[f"sf_off_K{K}"]
        got = np.array([[O.sf_offset(r, p, K) for p in range(K // 16)] for r in range(256)])
        assert np.array_equal(got, want)
    for variant, vn in ((O.G16, "g16"), (O.G32, "g32")):
        for KE in (0, 64, 256):
            want = t[f"pos_{vn}_KE{KE}"]
            got = np.array([[O.primary_pos(g, 256, KE, variant), O.residual_pos(g, 256, KE, variant)]
                            for g in range(16)])
            assert np.array_equal(got, want)


@pytest.mark.parametrize("variant", [O.G16, O.G32])
@pytest.mark.parametrize("KQ,KE", [(256, 0), (256, 64), (256, 256), (4096, 64), (3584, 128)])
def test_augmented_k_map_is_a_bijection(variant, KQ, KE):
    G, P = KQ // 16, (KQ - KE) // 16
    used = []
    for g in range(G):
        p = O.primary_pos(g, KQ, KE, variant)
        r = O.residual_pos(g, KQ, KE, variant)
        used.append(p)
        assert (r >= 0) == (g >= P)
        if r >= 0:
            used.append(r)
        if g < P:
            assert p == g                       # the non-outlier prefix is not moved
    assert sorted(used) == list(range((KQ + KE) // 16))


def test_augmented_k_map_explicit_small_case():
    # KQ=128 (8 groups), KE=64 (last 4 groups carry residuals)
    g16 = [(O.primary_pos(g, 128, 64, O.G16), O.residual_pos(g, 128, 64, O.G16)) for g in range(8)]
    assert g16 == [(0, -1), (1, -1), (2, -1), (3, -1), (4, 5), (6, 7), (8, 9), (10, 11)]
    g32 = [(O.primary_pos(g, 128, 64, O.G32), O.residual_pos(g, 128, 64, O.G32)) for g in range(8)]
    assert g32 == [(0, -1), (1, -1), (2, -1), (3, -1), (4, 6), (5, 7), (8, 10), (9, 11)]


def test_bf16_roundtrip_helpers():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(4096).astype(np.float32) * 100
    want = torch.from_numpy(a).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(O.f32_to_bf16_bits(a), want)
    L = O.lib()
    for v in a[:512]:
        assert L.arcq_o_f32_to_bf16(float(v)) == int(O.f32_to_bf16_bits(np.array([v]))[0])
