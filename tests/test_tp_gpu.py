"""Tensor-parallel shards through the HIP GEMM (one process plays every rank; the collectives themselves are covered
with gloo in test_tp_gloo.py): column shards concatenate to the unsharded result, row shards -- K ranges that are
multiples of 64 but of nothing larger -- sum to it."""
import numpy as np
import pytest
import torch

from arcquant_amd import tp
from oracle import oracle as O
from tests.util import bits, outlier_activations, prescale, random_perm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M", [3, 40, 300])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_reproduce_the_unsharded_gemm(M, world):
    from arcquant_amd import agemm
    N, KQ, KE = 1152, 1344, 64                       # 9 row tiles, 22 K atoms: uneven over 2, 3 and 8 ranks
    x, sx = prescale(outlier_activations(M, KQ, 30 + M))
    w, sw = prescale((torch.rand(N, KQ, generator=torch.Generator().manual_seed(31)) * 3 - 1).to(torch.bfloat16))
    idx = random_perm(KQ, 32).to(DEV)
    qx, sfx = agemm.reorder_quantize_x(x.to(DEV), idx, KE)
    qw, sfw = agemm.reorder_quantize_w(w.to(DEV), idx, KE)
    alpha = float(sx * sw)
    full = agemm.matmul(qx, qw, sfx, sfw, alpha, out_dtype=torch.float32)
    scale = float(full.abs().max())

    cols = []
    for r in range(world):
        cp = tp.ColumnParallelARCLinear(qw, sfw, float(sw), r, world)
        cols.append(agemm.matmul(qx, cp.W, sfx, cp.SFW, alpha, out_dtype=torch.float32))
        assert cols[-1].shape[1] == cp.ranges[r][1] - cp.ranges[r][0]
    got = torch.cat(cols, dim=1)
    assert got.shape == full.shape
    assert torch.allclose(got, full, rtol=1e-5, atol=2e-6 * scale)        # same products, possibly another tile shape

    total = torch.zeros_like(full)
    for r in range(world):
        rp = tp.RowParallelARCLinear(qw, sfw, float(sw), r, world)
        a, sfa = rp.shard_activation(qx, sfx)
        assert (rp.k1 - rp.k0) % 64 == 0 and a.shape[1] * 2 == rp.k1 - rp.k0
        if rp.k1 > rp.k0:
            total += agemm.matmul(a, rp.W, sfa, rp.SFW, alpha, out_dtype=torch.float32)
    assert torch.allclose(total, full, rtol=1e-5, atol=4e-6 * scale)


@pytest.mark.parametrize("world", [2, 8])
def test_handoff_local_scale_quantises_each_shard_with_the_global_scale(world):
    """Column -> row hand-off B on the GPU (one process plays every rank): the abs-max words of the shards reduce (max) to the word
    of the whole tensor; with it, ``reorder_quantize_x_dynamic(shard, local_index, KE, absmax_slots=word)`` returns for every
    shard the bytes of the ORACLE quantiser on torch's ``shard / scale`` with the GLOBAL scale -- what the rank would get from
    quantising the whole row, restricted to its columns and its own reorder_index.  Llama-3-70B: intermediate 28672 / 8."""
    from arcquant_amd import agemm
    M, NI, KE = 4, 28672, 64
    y = outlier_activations(M, NI, 90).to(DEV)
    shard = NI // world
    words = [tp.absmax_word(y[:, r * shard:(r + 1) * shard].contiguous()) for r in range(world)]
    word = torch.stack(words).max().reshape(1).to(torch.int32)                   # the all-reduce(MAX), played by hand
    assert int(word) == int(tp.absmax_word(y))
    amax = word.to(torch.int16).view(torch.bfloat16).float()[0].cpu()
    scale = amax * torch.tensor(1.0 / 2688.0, dtype=torch.float32)
    for r in (0, world - 1):
        ys = y[:, r * shard:(r + 1) * shard].contiguous()
        lidx = random_perm(shard, 91 + r)
        q, sf, s = agemm.reorder_quantize_x_dynamic(ys, lidx.to(DEV), KE, absmax_slots=word)
        assert float(s) == float(scale)
        xs = (ys.cpu().float() / scale.to(torch.bfloat16).float()).to(torch.bfloat16)
        oq, osf = O.quantize_x(bits(xs), lidx.numpy(), KE, agemm.variant_for_kq(shard), sf_fill=0)
        assert np.array_equal(q.cpu().numpy(), oq)
        K = shard + KE
        for row in range(M):
            for p in range(0, K // 16, 5):
                assert int(sf[O.sf_offset(row, p, K)]) == int(osf[O.sf_offset(row, p, K)])


def test_tp_decoder_layer_shard_fused_path_against_unfused_and_oracle():
    """One rank's shard of the Llama-3-70B layer at TP = 8 (hidden 8192, 8 query heads + 1 KV head of 128, intermediate 3584) as a
    world-1 TPDecoderLayer: every linear through its ONE-launch decode path (repack_for_decode: quantiser as the GEMM's prologue,
    fp32 partial for the row-parallel ones) against the same linear through the separate quantiser + reference-layout GEMM
    (another summation order: <= 1 bf16 ulp), and the o_proj hand-off B path against the oracle chain directly."""
    from arcquant_amd import agemm
    cfg = dict(hidden=8192, heads=8, kv_heads=1, head_dim=128, inter=3584)
    M, KE = 4, 64
    g = torch.Generator().manual_seed(70)
    h, hq, hk, it = cfg["hidden"], 8 * 128, 128, cfg["inter"]

    def rnd(n, k):
        return ((torch.rand(n, k, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16).to(DEV)

    shards = dict(wqkv=rnd(hq + 2 * hk, h), wo=rnd(h, hq), wgu=rnd(2 * it, h), wd=rnd(h, it))
    ln = (torch.rand(h, generator=g) + 0.5).to(torch.bfloat16).to(DEV)
    idx_h, idx_o, idx_d = random_perm(h, 1).to(DEV), random_perm(hq, 2).to(DEV), random_perm(it, 3).to(DEV)
    mk = lambda repack: tp.TPDecoderLayer.build(shards, ln, ln, idx_h, idx_o, idx_d, KE, KE, KE, 0, 1, 8, 1, 128, M, 16, repack=repack)   # noqa: E731
    Lf, Lu = mk(True), mk(False)
    assert Lf.qkv.RW is not None and Lu.qkv.RW is None
    x = (outlier_activations(M, h, 71) * 0.05).to(DEV)

    def close(a, b, what):
        a, b = a.float(), b.float()
        assert torch.all((a - b).abs() <= 2.0 ** -7 * b.abs() + 1e-4 * b.abs().max()), what

    # q|k|v and the MLP's first half: fused RMSNorm prologue (+ SiLU epilogue) vs quantiser + GEMM (+ torch SiLU)
    close(Lf.qkv.forward_rmsnorm(x, ln, 1e-5, idx_h, KE), Lu.qkv.forward_rmsnorm(x, ln, 1e-5, idx_h, KE), "qkv")
    act_f, word_f = Lf.gateup.forward_rmsnorm_silu(x, ln, 1e-5, idx_h, KE)
    act_u, word_u = Lu.gateup.forward_rmsnorm_silu(x, ln, 1e-5, idx_h, KE)
    assert act_f.shape == (M, it)
    af, au = act_f.float(), act_u.float()
    assert torch.all((af - au).abs() <= 2.0 ** -5 * au.abs() + 1e-3 * au.abs().max())          # SiLU*up of values one ulp apart
    assert int(word_f) == int(tp.absmax_word(act_f))
    # row-parallel linears with hand-off B on the SAME input: one launch (dynamic prologue, fp32 partial) vs two
    att = (outlier_activations(M, hq, 72) * 0.1).to(DEV)
    w_o = tp.handoff_local_scale(att)
    close(Lf.o.forward_local(att, w_o, residual=x), Lu.o.forward_local(att, w_o, residual=x), "o")
    w_d = tp.handoff_local_scale(word=word_u)
    close(Lf.down.forward_local(act_u, w_d, residual=x), Lu.down.forward_local(act_u, w_d, residual=x), "down")
    # ... and the fused o_proj partial against the oracle chain: torch's scale and division, the oracle's quantiser and fp64 GEMM
    part, _ = agemm.dynamic_matmul_repacked(att, idx_o, KE, Lf.o.RW, Lf.o.RSF, float(Lf.o.scale_w), h, absmax_slots=w_o, out_dtype=torch.float32)
    amax = w_o.to(torch.int16).view(torch.bfloat16).float()[0].cpu()
    scale = amax * torch.tensor(1.0 / 2688.0, dtype=torch.float32)
    xs = (att.cpu().float() / scale.to(torch.bfloat16).float()).to(torch.bfloat16)
    oq, osf = O.quantize_x(bits(xs), idx_o.cpu().numpy(), KE, agemm.variant_for_kq(hq), sf_fill=0)
    _, want, wabs = O.gemm(oq, Lf.o.W.cpu().numpy(), osf, Lf.o.SFW.cpu().numpy(), np.float32(float(scale)) * np.float32(Lf.o.scale_w), want_abs=True)
    assert np.all(np.abs(part.cpu().numpy() - want) <= 4e-6 * wabs + 1e-30)
    # the whole layer runs (two decode steps over its KV cache) and both variants agree to bf16 noise on the first stage output
    o1 = Lf.forward(x, 0, trace=True)
    o2 = Lf.forward(o1, 1)
    assert o1.shape == o2.shape == (M, h) and torch.isfinite(o2.float()).all()
    assert Lf.trace["qkv"].shape == (M, hq + 2 * hk) and Lf.trace["act"].shape == (M, it)
