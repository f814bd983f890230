"""Every BASELINE.json configuration at FULL size through the C-ABI on the GPU (configs_untested).

Recipe of test_gpu_parity.test_baseline_size_gemm_against_fp64_matmul: the quantisers' bytes against the CPU oracle
(poisoned output buffers, so untouched padding counts), then the GEMM against an fp64 matmul of the dequantised ORACLE
operands computed by torch on the GPU -- fp32 output within 2e-6 * sum|a*b| element-wise and 1e-5 norm-wise (the
north-star tolerance is 1e-3), bf16 output = the rounding of the same accumulators.

  config[1]  M=1, N=KQ=4096, KE=64                       -> test_gpu_parity.py (kept there)
  config[2]  Llama-3-8B linears, bs=1 seqlen=1           -> (1,14336,4096) (1,4096,14336) (1,1024,4096) (1,4096,4096)
  config[3]  Qwen2.5-7B bs=4 prefill=1024 + decode       -> M=4096 and M=4 x {3584->3584, 3584->37888 (gate|up), 18944->3584}, G32
  config[4]  Llama-3-70B TP=8 per-rank shards            -> N=1024 / 128 / 3584 column shards, K=8256 and 28736 cut 8 ways
  north star 8192 x 8192 x 8256
"""
import numpy as np
import pytest
import torch

from arcquant_amd import tp
from oracle import oracle as O
from tests.test_gpu_parity import DEV, _raw_quantize, _torch_dequant
from tests.util import bits, outlier_activations, prescale, random_perm

pytestmark = pytest.mark.gpu


def _agemm():
    from arcquant_amd import agemm
    return agemm


def _weights(N, KQ, seed):
    g = torch.Generator().manual_seed(seed)
    w = (torch.rand(N, KQ, generator=g) * 3 - 1.0).to(torch.bfloat16)
    return prescale(w)


def _quantise_both_ways(x, w, idx, KE, variant):
    """GPU quantisers (raw C-ABI, poisoned buffers) == oracle, byte for byte; returns the device operands."""
    xb, wb, ib = bits(x), bits(w), idx.numpy()
    oq, osf = O.quantize_x(xb, ib, KE, variant, sf_fill=0xEE)
    ow, owsf = O.quantize_w(wb, ib, KE, variant, sf_fill=0xEE)
    gq, gsf = _raw_quantize("x", xb, ib, KE, variant)
    gw, gwsf = _raw_quantize("w", wb, ib, KE, variant)
    assert np.array_equal(gq, oq) and np.array_equal(gsf, osf), "activation quantiser differs from the oracle"
    assert np.array_equal(gw, ow) and np.array_equal(gwsf, owsf), "weight quantiser differs from the oracle"
    return tuple(torch.from_numpy(a).to(DEV) for a in (oq, osf, ow, owsf))


def _check_gemm(A, SFA, B, SFB, alpha, repacked=True):
    ag = _agemm()
    M, N, K = A.shape[0], B.shape[0], A.shape[1] * 2
    a64, b64 = _torch_dequant(A, SFA, K), _torch_dequant(B, SFB, K)
    want = alpha * (a64 @ b64.t())
    wabs = abs(alpha) * (a64.abs() @ b64.abs().t())
    del a64, b64
    got = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32)
    err = (got.double() - want).abs()
    assert bool((err <= 2e-6 * wabs + 1e-30).all()), (M, N, K, float((err / (wabs + 1e-30)).max()))
    assert float((got.double() - want).norm() / want.norm()) < 1e-5
    d16 = ag.matmul(A, B, SFA, SFB, alpha)
    assert torch.equal(d16, got.to(torch.bfloat16))
    dev_alpha = torch.tensor([alpha], dtype=torch.float32, device=DEV)
    assert torch.equal(ag.matmul(A, B, SFA, SFB, dev_alpha), d16)          # device-resident per-tensor scale: same bits
    if repacked and ag.repacked_supported(M, N, K):                         # the decode copy of the weight: same products
        RW, RSF = ag.repack_w(B, SFB)
        got_r = ag.matmul_repacked(A, RW, SFA, RSF, alpha, N, out_dtype=torch.float32)
        err = (got_r.double() - want).abs()
        assert bool((err <= 2e-6 * wabs + 1e-30).all()), ("repacked", M, N, K)
    return got


# (M, N, KQ, KE): Llama-3-8B q/o, k/v, gate/up, down at one token (config[2]); variant from the reference's dispatch
CONFIG2 = [(1, 4096, 4096, 64), (1, 1024, 4096, 64), (1, 14336, 4096, 64), (1, 4096, 14336, 64)]
# Qwen2.5-7B (hidden 3584, intermediate 18944): q/k/v/o, fused gate|up, down; prefill (bs 4 x 1024 tokens) and decode (bs 4)
CONFIG3 = [(4096, 3584, 3584, 64), (4096, 37888, 3584, 64), (4096, 3584, 18944, 64),
           (4, 3584, 3584, 64), (4, 37888, 3584, 64), (4, 3584, 18944, 64)]
# Llama-3-70B (hidden 8192, intermediate 28672, 64 q / 8 kv heads of 128) column shards of one of 8 ranks, decode and prefill
CONFIG4_COLUMN = [(4, 1024, 8192, 64), (4, 128, 8192, 64), (4, 3584, 8192, 64), (512, 1024, 8192, 64), (512, 128, 8192, 64),
                  (512, 3584, 8192, 64)]
NORTH_STAR = [(8192, 8192, 8192, 64)]


@pytest.mark.parametrize("M,N,KQ,KE", CONFIG2 + CONFIG3 + CONFIG4_COLUMN + NORTH_STAR)
def test_baseline_config_shape_full_size(M, N, KQ, KE):
    ag = _agemm()
    variant = ag.variant_for_kq(KQ)                          # bindings.cpp:141-160: G32 for 3584 / 18944, G16 otherwise
    x, sx = prescale(outlier_activations(M, KQ, 45510 + M + N))
    w, sw = _weights(N, KQ, 7 + N + KQ)
    idx = random_perm(KQ, KQ + N)
    A, SFA, B, SFB = _quantise_both_ways(x, w, idx, KE, variant)
    del x, w
    _check_gemm(A, SFA, B, SFB, float(sx * sw))


@pytest.mark.parametrize("M", [4, 256])
@pytest.mark.parametrize("N,KQ", [(8192, 8192), (8192, 28672)])          # o_proj, down_proj of Llama-3-70B
def test_llama3_70b_row_parallel_shards_over_8_ranks(M, N, KQ):
    """config[4], row-parallel half: the augmented K axis (8256 = 129 atoms; 28736 = 449 atoms, G32) cut into the 8 per-rank
    ranges of tp.k_slices (1024/1088 and 3584/3648 elements -- 1024 is not in the reference's template list); every
    rank's GEMM on its slice of packed bytes and swizzled scales, fp32 partials summed in rank order == the unsharded
    fp64 result, and each shard's own result matches the fp64 matmul of its K range."""
    ag = _agemm()
    KE, world = 64, 8
    variant = ag.variant_for_kq(KQ)
    x, sx = prescale(outlier_activations(M, KQ, 70 + M))
    w, sw = _weights(N, KQ, 71 + KQ)
    idx = random_perm(KQ, 72)
    A, SFA, B, SFB = _quantise_both_ways(x, w, idx, KE, variant)
    del x, w
    K, alpha = KQ + KE, float(sx * sw)
    a64, b64 = _torch_dequant(A, SFA, K), _torch_dequant(B, SFB, K)
    want = alpha * (a64 @ b64.t())
    wabs = abs(alpha) * (a64.abs() @ b64.abs().t())
    total = torch.zeros((M, N), dtype=torch.float32, device=DEV)
    sizes = set()
    for r, (k0, k1) in enumerate(tp.k_slices(K, world)):
        sizes.add(k1 - k0)
        a, sfa = tp.shard_k(A, SFA, k0, k1)
        b, sfb = tp.shard_k(B, SFB, k0, k1)
        part = ag.matmul(a, b, sfa, sfb, alpha, out_dtype=torch.float32)
        want_r = alpha * (a64[:, k0:k1] @ b64[:, k0:k1].t())
        wabs_r = abs(alpha) * (a64[:, k0:k1].abs() @ b64[:, k0:k1].abs().t())
        assert bool(((part.double() - want_r).abs() <= 2e-6 * wabs_r + 1e-30).all()), (r, k0, k1)
        total += part
    assert sizes == ({1024, 1088} if KQ == 8192 else {3584, 3648})
    assert bool(((total.double() - want).abs() <= 4e-6 * wabs + 1e-30).all())
    assert float((total.double() - want).norm() / want.norm()) < 1e-5


def test_llama3_70b_column_shards_concatenate_to_the_full_projection():
    """config[4], column-parallel half on a REAL shard of a wider weight: q_proj 8192 -> 8192 split into 8 x 1024 rows with
    tp.shard_n (128-row scale tiles stay self-contained); rank outputs side by side == the unsharded GEMM, bit for bit in
    bf16 (same kernel, same K order per output element)."""
    ag = _agemm()
    M, N, KQ, KE, world = 4, 8192, 8192, 64, 8
    x, sx = prescale(outlier_activations(M, KQ, 80))
    w, sw = _weights(N, KQ, 81)
    idx = random_perm(KQ, 82).to(DEV)
    A, SFA = ag.reorder_quantize_x(x.to(DEV), idx, KE)
    B, SFB = ag.reorder_quantize_w(w.to(DEV), idx, KE)
    alpha = float(sx * sw)
    full = ag.matmul(A, B, SFA, SFB, alpha, out_dtype=torch.float32)
    outs = []
    for r in range(world):
        cp = tp.ColumnParallelARCLinear(B, SFB, float(sw), r, world)
        assert cp.W.shape[0] == 1024
        outs.append(ag.matmul(A, cp.W, SFA, cp.SFW, alpha, out_dtype=torch.float32))
    got = torch.cat(outs, dim=1)
    assert torch.allclose(got, full, rtol=1e-5, atol=2e-6 * float(full.abs().max()))
