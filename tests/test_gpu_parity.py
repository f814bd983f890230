"""GPU parity tests (``-m gpu``): the HIP kernels, called through the C-ABI, against the CPU oracle.

Bar: byte-exact for everything integer (packed codes, scale bytes, their swizzled positions, untouched
padding); for the GEMM, fp32 output within 1e-3 relative (north-star tolerance; measured error is ~1e-6)
of the fp64 oracle and bf16 output within one bf16 ulp of the oracle's rounding.
"""
import ctypes

import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.util import bits, from_bits, outlier_activations, prescale, random_perm

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _agemm():
    from arcquant_amd import agemm
    return agemm


def _lib():
    from arcquant_amd import _lib
    return _lib


def _raw_quantize(kind, x_bits, idx, KE, variant, wn_bits=None, eps=1e-6, fill=0xEE):
    """Call the C-ABI directly with poisoned output buffers so that untouched bytes are checked too."""
    L = _lib().lib()
    X = torch.from_numpy(x_bits.view(np.int16)).to(DEV)
    I = torch.from_numpy(idx).to(DEV)
    rows, KQ = x_bits.shape
    K = KQ + KE
    Q = torch.full((rows, K // 2), fill, dtype=torch.uint8, device=DEV)
    SF = torch.full((O.sf_alloc_bytes(rows, K),), fill, dtype=torch.uint8, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    if kind == "x":
        st = L.arcq_quantize_x(X.data_ptr(), I.data_ptr(), Q.data_ptr(), SF.data_ptr(), rows, KQ, KE, variant, s)
    elif kind == "w":
        st = L.arcq_quantize_w(X.data_ptr(), I.data_ptr(), Q.data_ptr(), SF.data_ptr(), rows, KQ, KE, variant, s)
    else:
        Wn = torch.from_numpy(wn_bits.view(np.int16)).to(DEV)
        st = L.arcq_rmsnorm_quantize_x(X.data_ptr(), Wn.data_ptr(), eps, I.data_ptr(), Q.data_ptr(), SF.data_ptr(), rows, KQ, KE,
                                       variant, s)
    _lib().check(st, kind)
    torch.cuda.synchronize()
    return Q.cpu().numpy(), SF.cpu().numpy()


QUANT_CASES = [
    # rows, KQ, KE, variant
    (1, 4096, 64, O.G16),          # BASELINE config[1] activation
    (3, 256, 64, O.G16),
    (130, 256, 0, O.G16),
    (257, 2048, 2048, O.G16),      # every group carries a residual
    (5, 3584, 64, O.G32),          # Qwen2.5-7B hidden
    (3, 256, 64, O.G32),
    (4, 18944, 128, O.G32),        # Qwen2.5-7B intermediate
    (2, 28672, 64, O.G32),         # the reference's "down" kernels (56 KB row in LDS)
    (7, 14336, 64, O.G16),         # Llama-3-8B intermediate
    (9, 1024, 64, O.G16),          # TP shard size outside the reference's closed list
    (300, 4096, 64, O.G16),
]


@pytest.mark.parametrize("rows,KQ,KE,variant", QUANT_CASES)
@pytest.mark.parametrize("kind", ["x", "w"])
def test_quantizers_byte_exact(rows, KQ, KE, variant, kind):
    x, _ = prescale(outlier_activations(rows, KQ, 1000 + rows + KQ))
    idx = random_perm(KQ, KQ + KE).numpy()
    xb = bits(x)
    want_q, want_sf = (O.quantize_x if kind == "x" else O.quantize_w)(xb, idx, KE, variant, sf_fill=0xEE)
    got_q, got_sf = _raw_quantize(kind, xb, idx, KE, variant)
    assert np.array_equal(got_q, want_q), f"packed codes differ in {(got_q != want_q).sum()} bytes"
    assert np.array_equal(got_sf, want_sf), f"scale bytes differ in {(got_sf != want_sf).sum()} places"


def test_quantizer_fuzz_byte_exact():
    """Forty seeded random (rows, KQ, KE, layout, kind) cases with RAW random bf16 bit patterns (all exponents, both
    zeros, subnormals; NaN / inf excluded) and random permutations: packed bytes and the whole scale buffer, untouched
    padding included, equal the kernel-text oracle."""
    rng = np.random.default_rng(4242)
    for case in range(40):
        variant = int(rng.integers(0, 2))
        unit = 32 if variant == O.G32 else 16
        KQ = int(rng.choice([64, 128, 192, 256, 448, 1024, 1536, 2048, 3584, 4096, 5120]))
        ke_opts = [k for k in range(0, KQ + 1, 64) if (KQ + k) % 64 == 0 and k % unit == 0 and KQ % unit == 0]
        if not ke_opts:
            continue
        KE = int(rng.choice(ke_opts[: max(1, min(len(ke_opts), 6))] + [ke_opts[-1]]))
        rows = int(rng.choice([1, 2, 3, 5, 16, 33, 127, 128, 129, 260]))
        kind = "x" if rng.integers(0, 2) else "w"
        xb = rng.integers(0, 1 << 16, size=(rows, KQ), dtype=np.uint16)
        if case % 2:                       # half of the cases: moderate magnitudes, so that codes other than 0 / +-6 dominate
            xb = (xb & 0x807F) | (rng.integers(120, 134, size=xb.shape, dtype=np.uint16) << 7)
        expo = (xb >> 7) & 0xFF
        xb = np.where(expo == 0xFF, xb & 0x807F, xb).astype(np.uint16)      # inf / nan -> subnormal / zero
        idx = rng.permutation(KQ).astype(np.int16)
        want_q, want_sf = (O.quantize_x if kind == "x" else O.quantize_w)(xb, idx, KE, variant, sf_fill=0xEE)
        got_q, got_sf = _raw_quantize(kind, xb, idx, KE, variant)
        assert np.array_equal(got_q, want_q), (case, rows, KQ, KE, variant, kind, int((got_q != want_q).sum()))
        assert np.array_equal(got_sf, want_sf), (case, rows, KQ, KE, variant, kind, int((got_sf != want_sf).sum()))


def test_quantizer_special_values_byte_exact():
    """zeros, negative zeros, tiny values (scale floor 2^-9), exact e2m1 ties, saturation at 448*6."""
    KQ = 256
    x = torch.zeros(8, KQ, dtype=torch.bfloat16)
    x[1] = -0.0
    x[2] = 1e-5
    x[3, :8] = torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 6.0])
    x[3, 8:16] = -x[3, :8]
    x[4] = 2688.0
    x[5, ::2] = 3e4                       # beyond 448*6: scale clamps at 448
    x[6] = torch.linspace(-7, 7, KQ).to(torch.bfloat16)
    x[7] = torch.linspace(0, 0.05, KQ).to(torch.bfloat16)
    idx = np.arange(KQ, dtype=np.int16)
    for variant in (O.G16, O.G32):
        for kind, fn in (("x", O.quantize_x), ("w", O.quantize_w)):
            want = fn(bits(x), idx, 64, variant, sf_fill=0xEE)
            got = _raw_quantize(kind, bits(x), idx, 64, variant)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("rows,KQ,KE,variant", [(4, 2048, 64, O.G16), (3, 3584, 64, O.G16), (3, 3584, 64, O.G32),
                                                 (130, 4096, 0, O.G16), (2, 8192, 128, O.G16), (5, 5120, 64, O.G16)])
def test_rmsnorm_quantizer_byte_exact(rows, KQ, KE, variant):
    x = outlier_activations(rows, KQ, 77 + KQ)
    wn = (torch.rand(KQ, generator=torch.Generator().manual_seed(KQ)) + 0.5).to(torch.bfloat16)
    idx = random_perm(KQ, 3 * KQ).numpy()
    want = O.rmsnorm_quantize_x(bits(x), bits(wn), 1e-6, idx, KE, variant, sf_fill=0xEE)
    got = _raw_quantize("rms", bits(x), idx, KE, variant, wn_bits=bits(wn))
    assert np.array_equal(got[0], want[0]), f"{(got[0] != want[0]).sum()} code bytes differ"
    assert np.array_equal(got[1], want[1])


def test_golden_pins_on_gpu(golden):
    g = golden("oracle_pins.npz")
    for tag in ("g16_a", "g16_b", "g16_c", "g32_a", "g32_b"):
        M, KQ, KE, variant = (int(v) for v in g[f"{tag}_meta"])
        qx, sfx = _raw_quantize("x", g[f"{tag}_x"], g[f"{tag}_idx"], KE, variant, fill=0)
        qw, sfw = _raw_quantize("w", g[f"{tag}_x"], g[f"{tag}_idx"], KE, variant, fill=0)
        assert np.array_equal(qx, g[f"{tag}_qx"]) and np.array_equal(sfx, g[f"{tag}_sfx"])
        assert np.array_equal(qw, g[f"{tag}_qw"]) and np.array_equal(sfw, g[f"{tag}_sfw"])
    ag = _agemm()
    A = torch.from_numpy(g["rms_qx"]).to(DEV)
    B = torch.from_numpy(g["rms_qw"]).to(DEV)
    SFA = torch.from_numpy(g["rms_sfx"]).to(DEV)
    SFB = torch.from_numpy(g["rms_sfw"]).to(DEV)
    d32 = ag.matmul(A, B, SFA, SFB, 0.0123, out_dtype=torch.float32).cpu().numpy().astype(np.float64)
    assert np.allclose(d32, g["rms_d_exact"], rtol=1e-3, atol=1e-6)
    d16 = bits(ag.matmul(A, B, SFA, SFB, 0.0123))
    assert _max_bf16_ulp_diff(d16, g["rms_d_bf16"]) <= 1


# ------------------------------------------------------------------------------------------------ GEMM
def _max_bf16_ulp_diff(a_bits, b_bits):
    def key(b):  # monotone integer key of a bf16 bit pattern
        b = b.astype(np.int32)
        return np.where(b & 0x8000, -(b & 0x7FFF), b & 0x7FFF)
    return int(np.abs(key(a_bits) - key(b_bits)).max()) if a_bits.size else 0


def _make_operands(M, N, KQ, KE, variant, seed):
    x, sx = prescale(outlier_activations(M, KQ, seed))
    w, sw = prescale((torch.rand(N, KQ, generator=torch.Generator().manual_seed(seed + 1)) * 3 - 1.0).to(torch.bfloat16))
    idx = random_perm(KQ, seed + 2).numpy()
    qx, sfx = O.quantize_x(bits(x), idx, KE, variant, sf_fill=0)
    qw, sfw = O.quantize_w(bits(w), idx, KE, variant, sf_fill=0)
    return qx, sfx, qw, sfw, float(sx * sw)


GEMM_CASES = [
    # M, N, KQ, KE, variant
    (1, 4096, 4096, 64, O.G16),       # BASELINE config[1]
    (1, 48, 256, 0, O.G16),
    (4, 1024, 2048, 64, O.G16),       # kv-proj-like: split-K path
    (16, 100, 256, 64, O.G16),        # N not a multiple of 16 / 4
    (3, 3584, 3584, 64, O.G32),
    (17, 128, 256, 64, O.G16),        # first M on the tile kernel
    (64, 256, 512, 64, O.G16),
    (130, 200, 256, 0, O.G16),        # ragged M and N on the tile kernel
    (128, 384, 1024, 128, O.G32),
    (300, 130, 320, 0, O.G16),        # N % 4 != 0 (scalar store path)
    (24, 260, 1984, 64, O.G16),       # 32-row tile, split-K x4, ragged 256-row weight tiles
    (33, 516, 2048, 64, O.G16),       # 64-row tile (33 live rows), split-K
    (100, 384, 2048, 128, O.G32),     # 128x128 tile, split-K
    (40, 130, 2048, 0, O.G16),        # N % 4 != 0: the split is refused, one pass over K
    (520, 256, 1024, 64, O.G16),      # several M tiles x split-K
    (4, 5120, 1088, 64, O.G16),       # 32-row-tile decode kernel (N >= 5120), partial tail slab
    (9, 5250, 512, 64, O.G16),        # ... ragged N, N % 4 != 0, two activation dwords per thread
    (16, 5120, 2048, 128, O.G32),     # ... four activation dwords per thread, three slabs
    (1, 6144, 256, 0, O.G16),         # ... a single short slab
    (4, 5120, 256, 256, O.G16),       # ... KE == KQ (every channel carries a residual)
    (4, 5120, 64, 0, O.G16),          # K = 64: a single scale atom, on each of the three kernels
    (3, 200, 64, 0, O.G16),
    (40, 256, 64, 0, O.G16),
    # M <= 16 now runs gemm_regtile.hip's 16 x 16 tiles by default (the cases above); the LDS-transposing decode kernels keep the shapes
    # where they measured faster -- M <= 8 with K > 8448 or N > 16384:
    (4, 384, 8512, 64, O.G16),        # 16-row decode kernel (gemm_skinny.hip), long K
    (2, 5120, 8512, 0, O.G16),        # 32-row decode kernel (gemm_decode.hip), long K
    (8, 16400, 256, 64, O.G16),       # 32-row decode kernel, N > 16384, ragged N
    # gemm_regtile.hip (16 < M, one round of <= 256 and >= 96 workgroups of its smallest fitting tile): every configuration its heuristic
    # picks, K tails of 0 / 1 / 2 / 3 atoms
    (24, 3104, 256, 64, O.G16),       # 32 x 16 tiles (2 x 1 MFMA tiles per wave, ring of 4 steps), one quad-step + 1 tail atom, 24 live token rows
    (64, 2112, 2112, 64, O.G16),      # 32 x 32 tiles (the 32 x 16 grid would exceed one round), 8 quad-steps over 8 waves + 2 tail atoms
    (32, 8200, 256, 64, O.G16),       # 32 x 64 tiles (the 32 x 32 grid would exceed one round)
    (100, 2110, 320, 64, O.G32),      # 64 x 32 tiles, ragged M and N (N % 16 != 0, N % 4 != 0: scalar stores), 2 tail atoms
    (256, 2048, 1024, 0, O.G16),      # ... K % 256 == 0: no tail, 4 quad-steps over 8 waves (some waves idle)
    (200, 2500, 384, 64, O.G16),      # 64 x 64 tiles (half-step prefetch, droppable requests), 3 tail atoms, N % 16 = 4
    (300, 4100, 576, 64, O.G16),      # 128 x 64 tiles (waves 2 x 1 x 4), 2 tail atoms
    (192, 10000, 256, 64, O.G16),     # 64 x 128 tiles (waves 1 x 2 x 4)
    (256, 10752, 256, 64, O.G16),     # 128 x 128 tiles (waves 2 x 2 x 2)
    (20, 16448, 256, 64, O.G16),      # very wide weight, no one-round tile: 32 x 64 tiles over two rounds
    (40, 16448, 256, 0, O.G16),       # ... 64 x 64 tiles
]


@pytest.mark.parametrize("M,N,KQ,KE,variant", GEMM_CASES)
def test_gemm_matches_oracle(M, N, KQ, KE, variant):
    ag = _agemm()
    qx, sfx, qw, sfw, alpha = _make_operands(M, N, KQ, KE, variant, 500 + M + N)
    want_bf16, want_exact, want_abs = O.gemm(qx, qw, sfx, sfw, alpha, want_abs=True)
    A, B = torch.from_numpy(qx).to(DEV), torch.from_numpy(qw).to(DEV)
    SFA, SFB = torch.from_numpy(sfx).to(DEV), torch.from_numpy(sfw).to(DEV)
    d32 = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32).cpu().numpy().astype(np.float64)
    # fp32 accumulation of exact products: error is bounded relative to sum|a*b|
    err = np.abs(d32 - want_exact)
    assert np.all(err <= 2e-6 * want_abs + 1e-30), float((err / (want_abs + 1e-30)).max())
    # north-star tolerance: 1e-3 relative (norm-wise, and element-wise away from cancellation)
    assert np.linalg.norm(d32 - want_exact) <= 1e-3 * np.linalg.norm(want_exact)
    big = np.abs(want_exact) > 1e-2 * want_abs
    assert np.all(err[big] <= 1e-3 * np.abs(want_exact[big]))
    # bf16 output: same rounding as the oracle up to accumulation-order noise
    d16 = bits(ag.matmul(A, B, SFA, SFB, alpha))
    assert _max_bf16_ulp_diff(d16, want_bf16) <= 1
    assert (d16 == want_bf16).mean() > 0.99
    # device-resident scale (no host sync) gives the same bits as the host float
    alpha_dev = torch.tensor([alpha], dtype=torch.float32, device=DEV)
    d16_dev = bits(ag.matmul(A, B, SFA, SFB, alpha_dev))
    assert np.array_equal(d16_dev, d16)


def test_gemm_bias_and_empty():
    ag = _agemm()
    qx, sfx, qw, sfw, alpha = _make_operands(5, 64, 256, 64, O.G16, 9)
    A, B = torch.from_numpy(qx).to(DEV), torch.from_numpy(qw).to(DEV)
    SFA, SFB = torch.from_numpy(sfx).to(DEV), torch.from_numpy(sfw).to(DEV)
    bias = torch.randn(64, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(DEV)
    base = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32)
    withb = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32, bias=bias)
    assert torch.allclose(withb, base + bias.float(), rtol=0, atol=1e-5 * float(base.abs().max()))
    empty = ag.matmul(A[:0], B, SFA, SFB, alpha)
    assert tuple(empty.shape) == (0, 64)


def test_full_pipeline_against_oracle_pipeline():
    """reorder_quantize_x + reorder_quantize_w + matmul through the agemm mirror == oracle end to end,
    on the reference's own check (kernels/main.py:7-48) at reduced M."""
    ag = _agemm()
    M, N, KQ = 32, 512, 4096
    x = outlier_activations(M, KQ, 45510)
    w = (torch.rand(N, KQ, generator=torch.Generator().manual_seed(45510)) * 3).to(torch.bfloat16)
    idx = torch.arange(KQ, dtype=torch.int16)
    xs, sx = prescale(x)
    ws, sw = prescale(w)
    ref = (x.float() @ w.float().T).numpy()
    prev = None
    for KE in (0, 256, 1024, 4096):
        A, SFA = ag.reorder_quantize_x(xs.to(DEV), idx.to(DEV), KE)
        B, SFB = ag.reorder_quantize_w(ws.to(DEV), idx.to(DEV), KE)
        C = ag.matmul(A, B, SFA, SFB, float(sx * sw), out_dtype=torch.float32).cpu().numpy()
        oq, osf = O.quantize_x(bits(xs), idx.numpy(), KE, O.G16)
        ow, owsf = O.quantize_w(bits(ws), idx.numpy(), KE, O.G16)
        assert np.array_equal(A.cpu().numpy(), oq) and np.array_equal(B.cpu().numpy(), ow)
        _, want = O.gemm(oq, ow, osf, owsf, float(sx * sw))
        assert np.linalg.norm(C - want) <= 1e-5 * np.linalg.norm(want)
        mse = float(np.mean((C - ref) ** 2))
        assert prev is None or mse < prev         # MSE(k) falls as KE grows, as the reference prints
        prev = mse


def test_absmax_scale_matches_torch():
    ag = _agemm()
    for n in (8, 1000, 4096 * 33 + 5):
        x = (torch.randn(n, generator=torch.Generator().manual_seed(n)) * 37).to(torch.bfloat16).to(DEV)
        got = ag.absmax_scale(x)
        want = torch.max(x.abs()).float() / (448.0 * 6.0)
        assert got.item() == want.item()


def test_unsupported_shapes_raise_runtime_error():
    ag = _agemm()
    x = torch.zeros(2, 4000, dtype=torch.bfloat16, device=DEV)         # (KQ+KE) % 64 != 0
    with pytest.raises(RuntimeError):
        ag.reorder_quantize_x(x, torch.arange(4000, dtype=torch.int16, device=DEV), 0)
    x = torch.zeros(2, 4096, dtype=torch.float16, device=DEV)          # wrong dtype (reference: c10::Error)
    with pytest.raises(RuntimeError):
        ag.reorder_quantize_x(x, torch.arange(4096, dtype=torch.int16, device=DEV), 0)
    with pytest.raises(NotImplementedError):
        ag.batch_decode_i4()


# ------------------------------------------------------------------------------------------------ full size
def _torch_dequant(Q, SF, K):
    """Independent (torch, on the GPU) statement of the format spec, validated against the oracle below."""
    rows = Q.shape[0]
    lut = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6, -0., -.5, -1, -1.5, -2, -3, -4, -6], dtype=torch.float64, device=Q.device)
    codes = torch.stack([Q & 0xF, Q >> 4], dim=-1).reshape(rows, K).long()
    r = torch.arange(rows, device=Q.device).unsqueeze(1)
    p = torch.arange(K // 16, device=Q.device).unsqueeze(0)
    off = ((r // 128) * (K // 64) + p // 4) * 512 + (r % 32) * 16 + ((r // 32) % 4) * 4 + p % 4
    sf = SF[off].view(torch.float8_e4m3fn).to(torch.float64)
    return lut[codes] * sf.repeat_interleave(16, dim=1)


def test_torch_dequant_helper_matches_oracle():
    qx, sfx, _, _, _ = _make_operands(130, 16, 256, 64, O.G16, 4)
    got = _torch_dequant(torch.from_numpy(qx).to(DEV), torch.from_numpy(sfx).to(DEV), 320).cpu().numpy()
    assert np.array_equal(got, O.dequant(qx, sfx).astype(np.float64))


@pytest.mark.parametrize("M", [1, 4096])
def test_baseline_size_gemm_against_fp64_matmul(M):
    """BASELINE.json sizes (N = KQ = 4096, KE = 64; M = 1 and M = 4096): quantise on the GPU (checked
    byte-exact against the oracle, which is fast enough for that), then compare the GEMM with an fp64
    matmul of the dequantised operands computed by torch on the GPU."""
    ag = _agemm()
    N = KQ = 4096
    KE = 64
    x, sx = prescale(outlier_activations(M, KQ, 45510))
    w, sw = prescale((torch.rand(N, KQ, generator=torch.Generator().manual_seed(7)) * 3).to(torch.bfloat16))
    idx = random_perm(KQ, 8)
    A, SFA = ag.reorder_quantize_x(x.to(DEV), idx.to(DEV), KE)
    B, SFB = ag.reorder_quantize_w(w.to(DEV), idx.to(DEV), KE)
    oq, osf = O.quantize_x(bits(x), idx.numpy(), KE, O.G16)
    ow, owsf = O.quantize_w(bits(w), idx.numpy(), KE, O.G16)
    assert np.array_equal(A.cpu().numpy(), oq) and np.array_equal(B.cpu().numpy(), ow)
    K = KQ + KE
    used_a, used_b = O.sf_used_bytes(M, K), O.sf_used_bytes(N, K)
    a64 = _torch_dequant(torch.from_numpy(oq).to(DEV), torch.from_numpy(osf).to(DEV), K)
    b64 = _torch_dequant(torch.from_numpy(ow).to(DEV), torch.from_numpy(owsf).to(DEV), K)
    assert np.array_equal(B.cpu().numpy(), ow) and used_a <= SFA.numel() and used_b <= SFB.numel()
    alpha = float(sx * sw)
    want = alpha * (a64 @ b64.T)
    got = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32).double()
    rel = (got - want).norm() / want.norm()
    assert rel < 1e-5, float(rel)
    # linearity in alpha (power-of-two factor: exact) and bf16 rounding of the same accumulators
    got2 = ag.matmul(A, B, SFA, SFB, 2.0 * alpha, out_dtype=torch.float32).double()
    assert torch.equal(got2, 2.0 * got)
    d16 = ag.matmul(A, B, SFA, SFB, alpha)
    assert torch.equal(d16, got.float().to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------ host mirror
def test_qlinear_layer_mirror_matches_oracle_pipeline():
    """QLinearLayer + NVFP4_reorder_quantize_x (the reference's operator protocol, model/qLinearLayer.py:30-78,
    model/qLlamaLayer.py:73-77) end to end against the oracle fed with the same torch pre-scaling."""
    from arcquant_amd import qlinear
    torch.manual_seed(0)
    bsz, q_len, KQ, N, KE = 2, 3, 2048, 256, 64
    lin = torch.nn.Linear(KQ, N, bias=True).to(torch.bfloat16).to(DEV)
    idx = random_perm(KQ, 12)
    layer = qlinear.QLinearLayer(lin, KE, idx)
    x = outlier_activations(bsz * q_len, KQ, 13).to(DEV)
    qx, scale_x, scale = qlinear.reorder_quantize_x(x, idx.to(DEV), KE)
    y = layer((qx, scale_x, scale, bsz, q_len))
    assert tuple(y.shape) == (bsz, q_len, N) and y.dtype == torch.bfloat16
    # oracle on identical pre-scaled inputs
    w = lin.weight.data
    sw = torch.max(w).float() / 2688.0
    sx = torch.max(x.abs()).float() / 2688.0
    ow, owsf = O.quantize_w(bits((w / sw).contiguous()), idx.numpy(), KE, O.G16)
    ox, oxsf = O.quantize_x(bits((x / sx).contiguous()), idx.numpy(), KE, O.G16)
    assert np.array_equal(layer.W.cpu().numpy(), ow) and np.array_equal(qx.cpu().numpy(), ox)
    want_bits, _ = O.gemm(ox, ow, oxsf, owsf, float(sx * sw))
    want = (from_bits(want_bits) + lin.bias.data.cpu()).reshape(bsz, q_len, N)      # bias added after rounding, as the reference
    assert _max_bf16_ulp_diff(bits(y), bits(want)) <= 2
    assert (bits(y) == bits(want)).mean() > 0.98
    # the optional decode copy of the weight gives the same layer output (up to fp32 accumulation order)
    fast = qlinear.QLinearLayer(lin, KE, idx, repack_for_decode=True)
    assert fast.RW is not None and torch.equal(fast.W, layer.W)
    y2 = fast((qx, scale_x, scale, bsz, q_len))
    assert _max_bf16_ulp_diff(bits(y2), bits(y)) <= 1 and (bits(y2) == bits(y)).mean() > 0.99


def test_gemm_with_subnormal_and_extreme_scales():
    """Scale bytes in the e4m3 subnormal range (tiny activations) and at 448 (saturating rows) are decoded exactly
    by the fp16-bit-pattern trick of gemm_common.hpp."""
    ag = _agemm()
    M, N, KQ = 4, 64, 256
    x = torch.zeros(M, KQ, dtype=torch.bfloat16)
    x[0] = 0.004                        # amax/6 below 2^-6 -> subnormal ue4m3 scale
    x[1] = torch.linspace(-0.05, 0.05, KQ).to(torch.bfloat16)
    x[2] = 2688.0                       # scale 448
    x[3, ::3] = 1e-3
    w = (torch.rand(N, KQ, generator=torch.Generator().manual_seed(2)) * 0.02 - 0.01).to(torch.bfloat16)
    idx = np.arange(KQ, dtype=np.int16)
    qx, sfx = O.quantize_x(bits(x), idx, 64, O.G16, sf_fill=0)
    qw, sfw = O.quantize_w(bits(w), idx, 64, O.G16, sf_fill=0)
    assert (sfx[:4 * 0 + 16] < 8).any() or (sfx < 8).any()      # subnormal scale bytes really occur
    _, want, wabs = O.gemm(qx, qw, sfx, sfw, 1.0, want_abs=True)
    got = ag.matmul(torch.from_numpy(qx).to(DEV), torch.from_numpy(qw).to(DEV), torch.from_numpy(sfx).to(DEV),
                    torch.from_numpy(sfw).to(DEV), 1.0, out_dtype=torch.float32).cpu().numpy().astype(np.float64)
    assert np.all(np.abs(got - want) <= 2e-6 * wabs + 1e-30)


def test_dynamic_quantizer_equals_torch_prescale_pipeline():
    """reorder_quantize_x_dynamic == the reference's three-step wrapper (model/qLlamaLayer.py:73-77:
    scale = max|x|/2688; x/scale in torch; reorder_quantize_x) byte for byte, for decode- and prefill-sized inputs,
    and repeated calls keep working (the abs-max scratch is rewritten by every call)."""
    ag = _agemm()
    for (M, KQ, KE) in [(4, 3584, 64), (1, 4096, 64), (300, 2048, 128), (4, 18944, 64), (64, 2048, 64), (16, 4096, 64)]:
        x = outlier_activations(M, KQ, 40 + M).to(DEV)
        idx = random_perm(KQ, 41).to(DEV)
        for _ in range(2):
            scale = torch.max(x.abs()).float() / (448.0 * 6.0)
            want_q, want_sf = ag.reorder_quantize_x((x / scale).contiguous(), idx, KE)
            got_q, got_sf, got_scale = ag.reorder_quantize_x_dynamic(x, idx, KE)
            assert got_scale.item() == scale.item()
            assert torch.equal(got_q, want_q)
            K = KQ + KE
            used = torch.zeros(got_sf.numel(), dtype=torch.bool)
            r = torch.arange(M).unsqueeze(1)
            p = torch.arange(K // 16).unsqueeze(0)
            off = ((r // 128) * (K // 64) + p // 4) * 512 + (r % 32) * 16 + ((r // 32) % 4) * 4 + p % 4
            used[off.reshape(-1)] = True
            assert torch.equal(got_sf.cpu()[used], want_sf.cpu()[used])
            x = -x * 0.5       # second round with a different maximum


def _used_sf_mask(M, K, numel):
    used = torch.zeros(numel, dtype=torch.bool)
    r = torch.arange(M).unsqueeze(1)
    p = torch.arange(K // 16).unsqueeze(0)
    off = ((r // 128) * (K // 64) + p // 4) * 512 + (r % 32) * 16 + ((r // 32) % 4) * 4 + p % 4
    used[off.reshape(-1)] = True
    return used


def test_silu_mul_quantizer_equals_torch_pipeline():
    """silu_mul_quantize_x_dynamic(gate|up) == the reference MLP's torch steps (model/qLlamaLayer.py:417
    `act_fn(gate) * up`, then :73-77) byte for byte: the bf16 roundings of torch's silu and mul kernels and its
    exp / divide are reproduced inside the quantiser."""
    import torch.nn.functional as F
    ag = _agemm()
    for (M, KQ, KE) in [(4, 18944, 64), (1, 3584, 64), (8, 512, 64), (300, 2048, 128)]:
        g = torch.Generator().manual_seed(7 * M + KQ)
        gu = (torch.randn(M, 2 * KQ, generator=g) * 3).to(torch.bfloat16)
        gu[0, :4] = torch.tensor([0.0, -0.0, 60.0, -60.0]).to(torch.bfloat16)      # exp under/overflow ends
        gu[0, 4:8] = torch.tensor([-100.0, 100.0, 1e-3, -1e-3]).to(torch.bfloat16)
        gu = gu.to(DEV)
        idx = random_perm(KQ, 43).to(DEV)
        act = F.silu(gu[:, :KQ]) * gu[:, KQ:]
        want_q, want_sf, want_scale = ag.reorder_quantize_x_dynamic(act.contiguous(), idx, KE)
        pairs = torch.stack((gu[:, :KQ], gu[:, KQ:]), dim=2).reshape(M, 2 * KQ).contiguous()      # g0, u0, g1, u1, ...
        for src, layout in ((gu, ag.GU_HALVES), (pairs, ag.GU_PAIRS), (gu, ag.GU_HALVES)):
            got_q, got_sf, got_scale = ag.silu_mul_quantize_x_dynamic(src, idx, KE, layout=layout)
            assert got_scale.item() == want_scale.item()
            assert torch.equal(got_q, want_q)
            used = _used_sf_mask(M, KQ + KE, got_sf.numel())
            assert torch.equal(got_sf.cpu()[used], want_sf.cpu()[used])
    with pytest.raises(RuntimeError):
        ag.silu_mul_quantize_x_dynamic(gu[:, :-1].contiguous(), idx, 64)


def test_gemm_residual_epilogue_matches_torch_add():
    ag = _agemm()
    for M, KQ in ((4, 512), (130, 512), (40, 1984), (90, 1984)):          # the last two finish through split-K
        qx, sfx, qw, sfw, alpha = _make_operands(M, 256, KQ, 64, O.G16, 77 + M)
        A, B = torch.from_numpy(qx).to(DEV), torch.from_numpy(qw).to(DEV)
        SFA, SFB = torch.from_numpy(sfx).to(DEV), torch.from_numpy(sfw).to(DEV)
        res = torch.randn(M, 256, generator=torch.Generator().manual_seed(M)).to(torch.bfloat16).to(DEV)
        want = res + ag.matmul(A, B, SFA, SFB, alpha)                     # torch: bf16 + bf16, rounded once more
        got = ag.matmul(A, B, SFA, SFB, alpha, residual=res)
        assert torch.equal(got, want)
        dev_scale = torch.tensor(alpha / 0.25, dtype=torch.float32, device=DEV)
        got2 = ag.matmul(A, B, SFA, SFB, dev_scale, scale_host=0.25, residual=res)
        assert torch.equal(got2, want)


def test_decode_and_tile_kernels_agree_and_full_residual_width():
    """The same 16 tokens through the decode kernel (M=16) and as the first rows of an M=48 call (tile kernel) give the
    same fp32 result up to accumulation order; KE == KQ (every channel carries a residual) is handled by both."""
    ag = _agemm()
    KQ = KE = 512
    qx, sfx, qw, sfw, alpha = _make_operands(48, 192, KQ, KE, O.G16, 321)
    A, B = torch.from_numpy(qx).to(DEV), torch.from_numpy(qw).to(DEV)
    SFA, SFB = torch.from_numpy(sfx).to(DEV), torch.from_numpy(sfw).to(DEV)
    big = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32)
    # rows 0..15 share the first 128-row scale tile, so the same scale buffer is valid for the 16-row call
    small = ag.matmul(A[:16].contiguous(), B, SFA, SFB, alpha, out_dtype=torch.float32)
    _, want, wabs = O.gemm(qx, qw, sfx, sfw, alpha, want_abs=True)
    assert np.all(np.abs(big.cpu().numpy() - want) <= 2e-6 * wabs + 1e-30)
    assert np.all(np.abs(small.cpu().numpy() - want[:16]) <= 2e-6 * wabs[:16] + 1e-30)
    assert torch.allclose(big[:16], small, rtol=1e-5, atol=1e-5 * float(big.abs().max()))


def test_repacked_weight_gemm_equals_the_reference_layout_gemm():
    """repack_w is pure data movement and matmul_repacked contracts the same exact products: its fp32 output equals the
    oracle within accumulation noise and matmul's within a few ulp, bf16 within one ulp; bias / residual / device scale
    behave alike; shapes cover one to eight waves per row block, ragged N, K % 256 in {0, 64, 128, 192}."""
    ag = _agemm()
    cases = [(1, 512, 256, 64, O.G16), (4, 100, 256, 64, O.G16), (3, 1000, 64, 0, O.G16), (16, 272, 1024, 64, O.G16),
             (8, 777, 512, 64, O.G16), (4, 3584, 3584, 64, O.G32), (2, 52000, 256, 64, O.G16), (5, 5120, 384, 0, O.G16),
             (4, 256, 18944, 64, O.G32),        # a 153.7 KB activation image (Qwen2.5-7B down-projection at bs=4)
             # decode batches, 16 < M <= 128: the weight is still read once; activations fetched per tile pair (no LDS: small weights, any
             # K) or resident packed in LDS (large weights while they fit) -- gemm_rowmid.hip, mid_kind
             (17, 128, 256, 64, O.G16), (32, 1000, 2048, 64, O.G16), (33, 516, 2048, 0, O.G16), (48, 777, 512, 64, O.G16),
             (64, 272, 1024, 64, O.G16), (40, 3584, 3584, 64, O.G32), (64, 4096, 4096, 64, O.G16), (32, 4096, 4096, 0, O.G16),
             (96, 520, 1088, 64, O.G16), (128, 4096, 4096, 64, O.G16), (100, 3584, 3584, 64, O.G32), (81, 1000, 64, 0, O.G16),
             (32, 256, 18944, 64, O.G32), (32, 10752, 3584, 64, O.G32), (64, 10752, 3584, 64, O.G32), (32, 37888, 3584, 64, O.G32)]
    for (M, N, KQ, KE, variant) in cases:
        K = KQ + KE
        if not ag.repacked_supported(M, N, K):             # only under the tuning override that switches the no-LDS kernel off
            assert M > 16 and os.environ.get("ARCQ_ROWTOK") == "0", (M, N, K)
            continue
        qx, sfx, qw, sfw, alpha = _make_operands(M, N, KQ, KE, variant, 77 + M + N)
        A, B = torch.from_numpy(qx).to(DEV), torch.from_numpy(qw).to(DEV)
        SFA, SFB = torch.from_numpy(sfx).to(DEV), torch.from_numpy(sfw).to(DEV)
        RW, RSF = ag.repack_w(B, SFB)
        ref32 = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32)
        got32 = ag.matmul_repacked(A, RW, SFA, RSF, alpha, N, out_dtype=torch.float32)
        assert got32.shape == (M, N)
        if N * K <= 4_000_000:
            _, want, wabs = O.gemm(qx, qw, sfx, sfw, alpha, want_abs=True)
            assert np.all(np.abs(got32.cpu().numpy() - want) <= 2e-6 * wabs + 1e-30), (M, N, K)
        assert torch.allclose(got32, ref32, rtol=2e-5, atol=2e-6 * float(ref32.abs().max())), (M, N, K)
        g = torch.Generator().manual_seed(N)
        bias = torch.randn(N, generator=g).to(torch.bfloat16).to(DEV)
        res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV)
        dev_scale = torch.tensor(alpha / 0.5, dtype=torch.float32, device=DEV)
        want16 = res + (got32.to(torch.bfloat16) + bias)          # the reference's op order: matmul -> + bias -> x + y, each in bf16
        got16 = ag.matmul_repacked(A, RW, SFA, RSF, dev_scale, N, scale_host=0.5, bias=bias, residual=res)
        assert torch.equal(got16, want16), (M, N, K)
    assert not ag.repacked_supported(129, 256, 256) and not ag.repacked_supported(5, 256, 19008)
    if "ARCQ_ROWTOK" not in os.environ:      # (the tuning override forces the no-LDS kernel onto every decode batch)
        assert not ag.repacked_supported(64, 37888, 3648)
    with pytest.raises(RuntimeError):
        ag.matmul_repacked(A, RW[:-1], SFA, RSF, alpha, N)


def test_repacked_gateup_gemm_leaves_the_absmax_of_silu_mul_for_a_one_launch_quantiser():
    """matmul_repacked_silu_absmax: y == matmul_repacked(...) bit for bit, max over its slots == max |silu(g) * u| of the
    torch pipeline, and silu_mul_quantize_x_dynamic(y, GU_PAIRS, absmax_slots=...) returns the bytes and the scale of the
    two-launch path (one to eight waves per row block, ragged N % 16, M = 1 ... 16)."""
    import torch.nn.functional as F
    ag = _agemm()
    for (M, IT, KQ, KE) in [(4, 2560, 512, 64), (1, 18, 256, 0), (3, 136, 256, 64), (16, 1000, 1024, 64), (4, 18944, 3584, 64)]:
        N = 2 * IT
        g = torch.Generator().manual_seed(M + IT)
        x, sx = prescale(outlier_activations(M, KQ, 5 + M))
        w = ((torch.rand(N, KQ, generator=g) * 2 - 1.0)).to(torch.bfloat16)               # rows: g0, u0, g1, u1, ...
        w, sw = prescale(w)
        idx = random_perm(KQ, 6).to(DEV)
        QX, SFX = ag.reorder_quantize_x(x.to(DEV), idx, KE)
        QW, SFW = ag.reorder_quantize_w(w.to(DEV), idx, KE)
        RW, RSF = ag.repack_w(QW, SFW)
        alpha = float(sx * sw)
        want_y = ag.matmul_repacked(QX, RW, SFX, RSF, alpha, N)
        y, slots = ag.matmul_repacked_silu_absmax(QX, RW, SFX, RSF, alpha, N)
        assert torch.equal(y, want_y) and slots.shape == ((N + 15) // 16,) and slots.dtype == torch.int32
        act = F.silu(y[:, 0::2]) * y[:, 1::2]
        amax_bits = int(act.abs().max().view(torch.int16).item()) & 0x7fff
        assert int(slots.max().item()) == amax_bits, (M, IT, KQ)
        idx_i = random_perm(IT, 9).to(DEV)
        ke_i = 64 if IT % 64 == 0 else 0
        if IT % 16 == 0 and (IT + ke_i) % 64 == 0:
            q2, s2, sc2 = ag.silu_mul_quantize_x_dynamic(y, idx_i, ke_i, layout=ag.GU_PAIRS)
            q1, s1, sc1 = ag.silu_mul_quantize_x_dynamic(y, idx_i, ke_i, layout=ag.GU_PAIRS, absmax_slots=slots)
            assert torch.equal(q1, q2) and float(sc1) == float(sc2), (M, IT, KQ)
            used = _used_sf_mask(M, IT + ke_i, s1.numel())
            assert torch.equal(s1.cpu()[used], s2.cpu()[used])
    with pytest.raises(RuntimeError):
        ag.matmul_repacked_silu_absmax(QX, RW, SFX, RSF, alpha, N + 2)
    with pytest.raises(RuntimeError):
        ag.silu_mul_quantize_x_dynamic(y, idx_i, ke_i, layout=ag.GU_PAIRS, absmax_slots=slots.float())


def test_silu_mul_gemm_epilogue_equals_the_unfused_steps():
    """matmul_silu_mul on row-interleaved gate/up weights == matmul, then torch's silu and mul, bit for bit; its abs-max
    slots give the dynamic quantiser the same scale and bytes as the abs-max pass (decode kernel, 64-row, 128x128 and
    256x256 tiles)."""
    import torch.nn.functional as F
    ag = _agemm()
    for (M, IT, KQ) in [(4, 2560, 512), (3, 136, 256), (40, 256, 512), (300, 512, 256), (3072, 2048, 256)]:
        g = torch.Generator().manual_seed(M + IT)
        x, sx = prescale(outlier_activations(M, KQ, 5 + M))
        w = ((torch.rand(2 * IT, KQ, generator=g) * 2 - 1.0)).to(torch.bfloat16)          # rows: g0, u0, g1, u1, ...
        w, sw = prescale(w)
        idx = random_perm(KQ, 6).to(DEV)
        qx, sfx = ag.reorder_quantize_x(x.to(DEV), idx, 64)
        qw, sfw = ag.reorder_quantize_w(w.to(DEV), idx, 64)
        alpha = float(sx * sw) * 40.0                                                   # activations of order 1..10
        y = ag.matmul(qx, qw, sfx, sfw, alpha)
        want = F.silu(y[:, 0::2]) * y[:, 1::2]
        act, slots = ag.matmul_silu_mul(qx, qw, sfx, sfw, alpha)
        assert act.shape == (M, IT) and torch.equal(act, want), (M, IT, KQ)
        dev_alpha = torch.tensor(alpha / 0.5, dtype=torch.float32, device=DEV)
        act2, _ = ag.matmul_silu_mul(qx, qw, sfx, sfw, dev_alpha, scale_host=0.5)
        assert torch.equal(act2, want)
        if IT % 16 == 0 and (IT + 64) % 64 == 0:
            idx2 = random_perm(IT, 8).to(DEV)
            q1, sf1, s1 = ag.reorder_quantize_x_dynamic(want.contiguous(), idx2, 64)
            q2, sf2, s2 = ag.reorder_quantize_x_dynamic(act, idx2, 64, absmax_slots=slots)
            assert s1.item() == s2.item() and torch.equal(q1, q2)
            used = _used_sf_mask(M, IT + 64, sf1.numel())
            assert torch.equal(sf1.cpu()[used], sf2.cpu()[used])
    with pytest.raises(RuntimeError):
        ag.matmul_silu_mul(qx, qw[:-4], sfx, sfw, 1.0)                                   # N % 8 != 0


def test_gemm_epilogue_operands_on_every_kernel():
    """bias + residual + device scale through the register-tiled kernel's decode tiles (M <= 16), the 16-row and the 32-row decode
    kernel (K > 8448), the register-tiled kernel
    (gemm_regtile.hip: 64 x 64, 32 x 32, 64 x 32, 64 x 64 and 128 x 64 tiles), the split-K tile path and the 128 x 128 tile: the fused epilogue equals the separate torch ops on the plain result, and doubling
    alpha doubles the fp32 output exactly (linearity in the per-tensor scale; a power of two commutes with every
    rounding)."""
    ag = _agemm()
    for (M, N, KQ) in [(4, 384, 512), (4, 5120, 512), (4, 384, 8512), (4, 5120, 8512), (40, 512, 1984), (300, 2048, 256), (60, 3200, 512), (128, 2100, 256), (250, 2500, 320), (300, 4100, 256), (1100, 256, 256)]:
        qx, sfx, qw, sfw, alpha = _make_operands(M, N, KQ, 64, O.G16, 3 * M + N)
        A, B = torch.from_numpy(qx).to(DEV), torch.from_numpy(qw).to(DEV)
        SFA, SFB = torch.from_numpy(sfx).to(DEV), torch.from_numpy(sfw).to(DEV)
        g = torch.Generator().manual_seed(N)
        bias = torch.randn(N, generator=g).to(torch.bfloat16).to(DEV)
        res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV)
        plain32 = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32)
        assert torch.equal(ag.matmul(A, B, SFA, SFB, 2 * alpha, out_dtype=torch.float32), 2 * plain32)
        want = res + (plain32.to(torch.bfloat16) + bias)                  # matmul -> + bias -> + residual, each rounded to bf16 (the reference's ops)
        dev_scale = torch.tensor(alpha / 0.5, dtype=torch.float32, device=DEV)
        got = ag.matmul(A, B, SFA, SFB, dev_scale, scale_host=0.5, bias=bias, residual=res)
        assert torch.equal(got, want), (M, N, KQ)


def test_gemm_epilogue_operands_as_misaligned_views():
    """bias / residual that are 2-byte-offset views (not 8-byte aligned): the kernels fall back from their 8-byte operand loads to
    element loads and give the same bits (tile GEMM 256 x 256 and 128 x 256, register-tiled kernel at M = 4 and M = 200)."""
    ag = _agemm()
    for (M, N, KQ) in [(4, 384, 512), (200, 2500, 320), (300, 256, 256), (1100, 256, 256)]:
        qx, sfx, qw, sfw, alpha = _make_operands(M, N, KQ, 64, O.G16, 5 * M + N)
        A, B = torch.from_numpy(qx).to(DEV), torch.from_numpy(qw).to(DEV)
        SFA, SFB = torch.from_numpy(sfx).to(DEV), torch.from_numpy(sfw).to(DEV)
        g = torch.Generator().manual_seed(N + 1)
        bias = torch.randn(N, generator=g).to(torch.bfloat16).to(DEV)
        res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV)
        want = ag.matmul(A, B, SFA, SFB, alpha, bias=bias, residual=res)
        bias_off = torch.empty(N + 1, dtype=torch.bfloat16, device=DEV)[1:]
        res_off = torch.empty(M * N + 1, dtype=torch.bfloat16, device=DEV)[1:].view(M, N)
        bias_off.copy_(bias)
        res_off.copy_(res)
        assert bias_off.data_ptr() % 8 == 2 and res_off.data_ptr() % 8 == 2
        got = ag.matmul(A, B, SFA, SFB, alpha, bias=bias_off, residual=res_off)
        assert torch.equal(got, want), (M, N, KQ)


@pytest.mark.parametrize("M,N", [(1, 1), (2, 16), (16, 17), (33, 1)])
def test_gemm_tiny_output_shapes(M, N):
    ag = _agemm()
    qx, sfx, qw, sfw, alpha = _make_operands(M, N, 256, 64, O.G16, 900 + M + N)
    _, want, wabs = O.gemm(qx, qw, sfx, sfw, alpha, want_abs=True)
    got = ag.matmul(torch.from_numpy(qx).to(DEV), torch.from_numpy(qw).to(DEV), torch.from_numpy(sfx).to(DEV),
                    torch.from_numpy(sfw).to(DEV), alpha, out_dtype=torch.float32).cpu().numpy()
    assert got.shape == (M, N)
    assert np.all(np.abs(got - want) <= 2e-6 * wabs + 1e-30)


def test_gemm_fuzz_random_shapes_against_fp64_matmul():
    """Sixty seeded random shapes across every dispatch boundary (16-row / 32-row decode kernels, 32- / 64- / 128- / 256-row
    tiles, split-K on and off, ragged M and N, N % 4 != 0, KE in {0, 64, KQ}) with random e2m1 codes and random ue4m3
    scale bytes straight in the operand buffers, against an fp64 matmul of the dequantised operands (torch, on the GPU)."""
    ag = _agemm()
    rng = np.random.default_rng(20261004)
    m_pool = [1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 18, 31, 32, 33, 47, 64, 65, 100, 128, 129, 200, 257, 300, 513, 700]
    n_pool = [1, 7, 16, 24, 100, 127, 128, 129, 255, 256, 384, 500, 1000, 1024, 2050, 4096, 5119, 5120, 5124, 6148]
    k_pool = [64, 128, 192, 320, 512, 1088, 2048, 3136]
    for case in range(60):
        M, N, K = int(rng.choice(m_pool)), int(rng.choice(n_pool)), int(rng.choice(k_pool))
        if M * N * K > 6e9:
            K = 320
        g = torch.Generator(device=DEV).manual_seed(1000 + case)

        def operand(rows):
            q = torch.randint(0, 256, (rows, K // 2), generator=g, device=DEV, dtype=torch.uint8)
            nbytes = ag.sf_buffer_bytes(rows, K)
            # scale bytes: any finite ue4m3 code from the subnormals up to 2^3 (keeps |products| far below fp32 overflow)
            sf = torch.randint(1, 0x58, (nbytes,), generator=g, device=DEV, dtype=torch.uint8)
            return q, sf

        A, SFA = operand(M)
        B, SFB = operand(N)
        alpha = float(rng.uniform(0.25, 4.0))
        a64, b64 = _torch_dequant(A, SFA, K), _torch_dequant(B, SFB, K)
        want = alpha * (a64 @ b64.t())
        wabs = abs(alpha) * (a64.abs() @ b64.abs().t())
        got = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32).double()
        assert got.shape == (M, N)
        err = (got - want).abs()
        assert bool((err <= 2e-6 * wabs + 1e-30).all()), (case, M, N, K, float((err / (wabs + 1e-30)).max()))
        got16 = ag.matmul(A, B, SFA, SFB, alpha).double()
        assert bool(((got16 - want).abs() <= want.abs() * 2.0 ** -8 + 2e-6 * wabs + 1e-30).all()), (case, M, N, K)
        if ag.repacked_supported(M, N, K):                 # M <= 16: fp16 image; 16 < M <= 64: packed activations in LDS
            RW, RSF = ag.repack_w(B, SFB)
            gotr = ag.matmul_repacked(A, RW, SFA, RSF, alpha, N, out_dtype=torch.float32).double()
            errr = (gotr - want).abs()
            assert bool((errr <= 2e-6 * wabs + 1e-30).all()), ("repacked", case, M, N, K, float((errr / (wabs + 1e-30)).max()))


# ------------------------------------------------------------------------------------------------ fused decode linears
def _oracle_dyn_quant(x_dev, idx_dev, KE, variant):
    """The reference's NVFP4_reorder_quantize_x on the CPU: torch's scale and division, the ORACLE's quantiser."""
    x = x_dev.cpu()
    # torch's GPU true-divide by a Python scalar multiplies by the fp32 reciprocal (BinaryDivTrueKernel): the scale of
    # `torch.max(x.abs()).float() / (448.0*6.0)` (model/qLlamaLayer.py:74) is amax * fl(1/2688) there; the CPU would divide
    scale = torch.max(x.abs()).float() * torch.tensor(1.0 / (448.0 * 6.0), dtype=torch.float32)
    # torch on the GPU divides a bf16 tensor by a 0-dim fp32 tensor in bf16: the scale is rounded to bf16, the quotient formed
    # in fp32 and rounded once (BinaryFunctor<BFloat16, BFloat16, BFloat16, DivFunctor>); stated explicitly here for the CPU
    xs = (x.float() / scale.to(torch.bfloat16).float()).to(torch.bfloat16)
    q, sf = O.quantize_x(bits(xs.contiguous()), idx_dev.cpu().numpy(), KE, variant, sf_fill=0)
    return q, sf, float(scale)


FUSED_CASES = [
    # M, N, KQ, KE : Qwen2.5-7B decode shapes (G32) and small / G16 / ragged ones
    (4, 10752, 3584, 64),      # q|k|v
    (4, 3584, 3584, 64),       # o_proj
    (1, 4096, 4096, 64),       # config[1]
    (16, 272, 2048, 64),
    (3, 1000, 2048, 128),
    (2, 48, 2048, 2048),       # every channel carries a residual
    (5, 520, 4096, 0),
]


@pytest.mark.parametrize("M,N,KQ,KE", FUSED_CASES)
def test_fused_rmsnorm_and_dynamic_linears_equal_the_separate_launches_and_the_oracle(M, N, KQ, KE):
    """rmsnorm_matmul_repacked == rmsnorm_quantize_x + matmul_repacked, dynamic_matmul_repacked == reorder_quantize_x_dynamic +
    matmul_repacked, BIT FOR BIT (bias, residual, bf16 and fp32 output), and both within the GEMM tolerance of the CPU oracle
    chain (O.rmsnorm_quantize_x / O.quantize_x on torch's x/scale, then O.gemm): the fused prologue is the quantiser."""
    ag = _agemm()
    variant = ag.variant_for_kq(KQ)
    g = torch.Generator().manual_seed(M * 1000 + N + KQ)
    x = outlier_activations(M, KQ, 900 + M + N).to(DEV)
    wn = (torch.rand(KQ, generator=g) + 0.5).to(torch.bfloat16).to(DEV)
    w, sw = prescale((torch.rand(N, KQ, generator=g) * 2 - 1.0).to(torch.bfloat16))
    sw = float(sw)
    idx = random_perm(KQ, 7 + KQ).to(DEV)
    QW, SFW = ag.reorder_quantize_w(w.to(DEV), idx, KE)
    RW, RSF = ag.repack_w(QW, SFW)
    bias = torch.randn(N, generator=g).to(torch.bfloat16).to(DEV)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV)
    ow, owsf = QW.cpu().numpy(), SFW.cpu().numpy()

    if 2048 <= KQ <= 8192:
        assert ag.fused_supported(ag.SRC_RMSNORM, M, N, KQ, KE)
        A, SFA = ag.rmsnorm_quantize_x(x, wn, 1e-6, idx, KE)
        for kw in (dict(), dict(bias=bias), dict(residual=res), dict(bias=bias, residual=res, out_dtype=torch.float32)):
            want = ag.matmul_repacked(A, RW, SFA, RSF, sw, N, kernel="stream", **kw)
            got = ag.rmsnorm_matmul_repacked(x, wn, 1e-6, idx, KE, RW, RSF, sw, N, **kw)
            assert torch.equal(got, want), (M, N, KQ, KE, list(kw))
        dev_scale = torch.tensor(sw / 0.5, dtype=torch.float32, device=DEV)
        assert torch.equal(ag.rmsnorm_matmul_repacked(x, wn, 1e-6, idx, KE, RW, RSF, dev_scale, N, scale_host=0.5),
                           ag.matmul_repacked(A, RW, SFA, RSF, sw, N, kernel="stream"))
        oq, osf = O.rmsnorm_quantize_x(bits(x), bits(wn), 1e-6, idx.cpu().numpy(), KE, variant, sf_fill=0)
        if N * (KQ + KE) <= 16_000_000:
            _, want_e, wabs = O.gemm(oq, ow, osf, owsf, sw, want_abs=True)
            got32 = ag.rmsnorm_matmul_repacked(x, wn, 1e-6, idx, KE, RW, RSF, sw, N, out_dtype=torch.float32).cpu().numpy()
            assert np.all(np.abs(got32 - want_e) <= 2e-6 * wabs + 1e-30)

    assert ag.fused_supported(ag.SRC_DYNAMIC, M, N, KQ, KE)
    qa, sfa, sa = ag.reorder_quantize_x_dynamic(x, idx, KE)
    slots = None
    for use_slots in (False, True):
        if use_slots:      # abs-max words as a producing kernel leaves them: one per 16 columns (bf16 magnitude bits), any split works
            mag = (x.view(torch.int16).to(torch.int32) & 0x7FFF)
            slots = torch.stack([c.max() for c in mag.reshape(-1).split(997)]).to(torch.int32).contiguous()
        for kw in (dict(), dict(bias=bias, residual=res), dict(residual=res, out_dtype=torch.float32)):
            want = ag.matmul_repacked(qa, RW, sfa, RSF, sa, N, scale_host=sw, kernel="stream", **kw)
            got, got_scale = ag.dynamic_matmul_repacked(x, idx, KE, RW, RSF, sw, N, absmax_slots=slots, **kw)
            assert float(got_scale) == float(sa)
            assert torch.equal(got, want), (M, N, KQ, KE, use_slots, list(kw))
    oq, osf, osc = _oracle_dyn_quant(x, idx, KE, variant)
    assert osc == float(sa) and np.array_equal(qa.cpu().numpy(), oq)
    if N * (KQ + KE) <= 16_000_000:
        _, want_e, wabs = O.gemm(oq, ow, osf, owsf, np.float32(osc) * np.float32(sw), want_abs=True)
        got32, _ = ag.dynamic_matmul_repacked(x, idx, KE, RW, RSF, sw, N, out_dtype=torch.float32)
        assert np.all(np.abs(got32.cpu().numpy() - want_e) <= 4e-6 * wabs + 1e-30)


@pytest.mark.parametrize("M,IT,KQ", [(4, 18944, 3584), (1, 2048, 2048), (16, 320, 2048), (3, 8192, 4096)])
def test_fused_mlp_pair_equals_the_separate_launches(M, IT, KQ):
    """The decode MLP in two launches -- rmsnorm_matmul_repacked_silu (RMSNorm + quantise + gate|up GEMM + SiLU*up + abs-max
    words) and dynamic_matmul_repacked(absmax_slots=...) (dynamic quantise + down GEMM + residual) -- against the seven-launch
    chain it replaces, bit for bit; the Qwen2.5-7B case gathers its 18944-wide activation from global memory (the 152 KB image
    leaves no LDS to stage it)."""
    import torch.nn.functional as F
    ag = _agemm()
    KE, N = 64, 2 * IT
    g = torch.Generator().manual_seed(IT + M)
    x = outlier_activations(M, KQ, 31 + M).to(DEV)
    wn = (torch.rand(KQ, generator=g) + 0.5).to(torch.bfloat16).to(DEV)
    wgu, sgu = prescale((torch.rand(N, KQ, generator=g) * 2 - 1.0).to(torch.bfloat16))       # rows: g0, u0, g1, u1, ...
    idx = random_perm(KQ, 3).to(DEV)
    RWg, RSFg = ag.repack_w(*ag.reorder_quantize_w(wgu.to(DEV), idx, KE))
    alpha = float(sgu) * 3e-3                                                                # activations of order 1
    y = ag.rmsnorm_matmul_repacked(x, wn, 1e-6, idx, KE, RWg, RSFg, alpha, N)
    y_auto = ag.matmul_repacked(*ag.rmsnorm_quantize_x(x, wn, 1e-6, idx, KE)[:1], RWg, ag.rmsnorm_quantize_x(x, wn, 1e-6, idx, KE)[1], RSFg, alpha, N)
    assert _max_bf16_ulp_diff(bits(y.cpu()), bits(y_auto.cpu())) <= 1            # the default packed-path kernel: same products, another summation order
    want_act = F.silu(y[:, 0::2]) * y[:, 1::2]
    act, slots = ag.rmsnorm_matmul_repacked_silu(x, wn, 1e-6, idx, KE, RWg, RSFg, alpha, N)
    assert act.shape == (M, IT) and torch.equal(act, want_act)
    assert slots.shape == ((N + 15) // 16,) and int(slots.max()) == (int(want_act.abs().max().view(torch.int16)) & 0x7FFF)
    # direct check of the SiLU epilogue against a CPU statement of torch's two roundings (the GEMM itself is oracle-checked above)
    yc = y.cpu().float()
    silu_cpu = (yc[:, 0::2] / (1.0 + torch.exp(-yc[:, 0::2]))).to(torch.bfloat16)
    act_cpu = (silu_cpu.float() * yc[:, 1::2]).to(torch.bfloat16)
    same = (bits(act_cpu) == bits(act.cpu()))
    assert same.mean() > 0.999 and _max_bf16_ulp_diff(bits(act_cpu), bits(act.cpu())) <= 1       # exp: ocml on the GPU, libm here
    # down projection
    Nd = 1024 if IT > 4096 else 256
    idx_i = random_perm(IT, 5).to(DEV)
    wd, sd = prescale((torch.rand(Nd, IT, generator=g) * 2 - 1.0).to(torch.bfloat16))
    RWd, RSFd = ag.repack_w(*ag.reorder_quantize_w(wd.to(DEV), idx_i, KE))
    res = torch.randn(M, Nd, generator=g).to(torch.bfloat16).to(DEV)
    if not ag.fused_supported(ag.SRC_DYNAMIC, M, Nd, IT, KE):
        return
    qa, sfa, sa = ag.reorder_quantize_x_dynamic(want_act.contiguous(), idx_i, KE)
    want = ag.matmul_repacked(qa, RWd, sfa, RSFd, sa, Nd, scale_host=float(sd), residual=res, kernel="stream")
    got, got_scale = ag.dynamic_matmul_repacked(act, idx_i, KE, RWd, RSFd, float(sd), Nd, absmax_slots=slots, residual=res)
    assert float(got_scale) == float(sa) and torch.equal(got, want)
    got2, _ = ag.dynamic_matmul_repacked(act, idx_i, KE, RWd, RSFd, float(sd), Nd, residual=res)       # abs-max recomputed in the kernel
    assert torch.equal(got2, want)


@pytest.mark.parametrize("M,IT,KQ,variant", [(4, 18944, 3584, None), (1, 1024, 2048, 0), (7, 1536, 2048, 1), (16, 320, 2048, None)])
def test_gateup_epilogue_scatter_plus_contiguous_quantiser_equals_the_gather(M, IT, KQ, variant):
    """act_scatter_index = inverse of the down projection's reorder_index: the gate|up epilogue stores act[:, reorder_index], and
    reorder_quantize_x_dynamic(reorder_index=None) on it gives the bytes the gathering quantiser gives on the natural-order
    activation -- codes, every swizzled scale byte (poisoned buffers are not used here: both sides come from torch.empty, so only
    the used bytes are compared) and the per-tensor scale; both variants, residual channels included."""
    ag = _agemm()
    KE, N = 64, 2 * IT
    g = torch.Generator().manual_seed(IT + M)
    x = outlier_activations(M, KQ, 77 + M).to(DEV)
    wn = (torch.rand(KQ, generator=g) + 0.5).to(torch.bfloat16).to(DEV)
    wgu, sgu = prescale((torch.rand(N, KQ, generator=g) * 2 - 1.0).to(torch.bfloat16))
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    idx = random_perm(KQ, 3).to(DEV)
    RWg, RSFg = ag.repack_w(*ag.reorder_quantize_w(wgu.to(DEV), idx, KE))
    alpha = float(sgu) * 3e-3
    idx_i = random_perm(IT, 5).to(DEV)
    inv = torch.argsort(idx_i.long()).to(torch.int16)
    act, slots = ag.rmsnorm_matmul_repacked_silu(x, wn, 1e-6, idx, KE, RWg, RSFg, alpha, N, bias=bias)
    act_s, slots_s = ag.rmsnorm_matmul_repacked_silu(x, wn, 1e-6, idx, KE, RWg, RSFg, alpha, N, bias=bias, act_scatter_index=inv)
    assert torch.equal(act_s, act[:, idx_i.long()]) and torch.equal(slots_s, slots)
    q0, sf0, s0 = ag.reorder_quantize_x_dynamic(act, idx_i, KE, variant=variant, absmax_slots=slots)
    q1, sf1, s1 = ag.reorder_quantize_x_dynamic(act_s, None, KE, variant=variant, absmax_slots=slots_s)
    assert float(s0) == float(s1) and torch.equal(q0, q1)
    v = ag.variant_for_kq(IT) if variant is None else variant
    want_q, want_sf = O.quantize_x(bits(ag_div(act, s0).cpu()), idx_i.cpu().numpy(), KE, v, sf_fill=0xEE)
    assert np.array_equal(q1.cpu().numpy(), want_q)                                       # ... and the CPU oracle's bytes
    used = want_sf != 0xEE
    got0, got1 = sf0.cpu().numpy(), sf1.cpu().numpy()
    assert np.array_equal(got0[used], got1[used]) and np.array_equal(got1[used], want_sf[used])
    with pytest.raises(RuntimeError):
        ag.reorder_quantize_x_dynamic(act_s, None, KE)                                   # identity order needs the abs-max words


def ag_div(x, scale):
    """torch's GPU `x / scale` for a bf16 tensor and a 0-dim fp32 scale (model/qLlamaLayer.py:74-76)."""
    return x / scale
