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
