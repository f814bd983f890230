"""CPU tests of the tensor-parallel sharding (arcquant_amd/tp.py) with world_size-2 (and 4) gloo ranks.

The slicing helpers are device-agnostic tensor ops; the per-rank GEMM is done by the CPU oracle here (the
product GEMM needs a GPU), the exchange step is a real torch.distributed all-reduce / all-gather over gloo.
Checked: row-parallel partial sums == unsharded oracle result; column-parallel concatenation == unsharded.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from arcquant_amd import tp
from oracle import oracle as O
from tests.util import bits, outlier_activations, prescale, random_perm


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    M, N, KQ, KE = 5, 384, 512, 128
    x, sx = prescale(outlier_activations(M, KQ, 3))
    w, sw = prescale((torch.rand(N, KQ, generator=torch.Generator().manual_seed(4)) * 3 - 1).to(torch.bfloat16))
    idx = random_perm(KQ, 5).numpy()
    qx, sfx = O.quantize_x(bits(x), idx, KE, O.G16, sf_fill=0)
    qw, sfw = O.quantize_w(bits(w), idx, KE, O.G16, sf_fill=0)
    return M, N, KQ + KE, qx, sfx, qw, sfw, float(sx * sw)


def _worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        M, N, K, qx, sfx, qw, sfw, alpha = _problem()
        QX, SFX = torch.from_numpy(qx), torch.from_numpy(sfx)
        QW, SFW = torch.from_numpy(qw), torch.from_numpy(sfw)
        if mode == "row":
            k0, k1 = tp.k_slices(K, world)[rank]
            a, sfa = tp.shard_k(QX, SFX, k0, k1)
            b, sfb = tp.shard_k(QW, SFW, k0, k1)
            _, part = O.gemm(a.numpy(), b.numpy(), sfa.numpy(), sfb.numpy(), alpha)
            part = torch.from_numpy(part)
            dist.all_reduce(part, op=dist.ReduceOp.SUM)          # the one exchange step of the row-parallel linear
            if rank == 0:
                np.save(os.path.join(out_dir, "row.npy"), part.numpy())
        else:
            n0, n1 = tp.n_slices(N, world)[rank]
            b, sfb = tp.shard_n(QW, SFW, n0, n1)
            _, part = O.gemm(qx, b.numpy(), sfx, sfb.numpy(), alpha)
            part = torch.from_numpy(part)
            full = tp.all_gather_columns(part, [e - s for s, e in tp.n_slices(N, world)])
            if rank == 0:
                np.save(os.path.join(out_dir, "col.npy"), full.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("mode", ["row", "col"])
def test_sharded_linear_matches_unsharded(tmp_path, world, mode):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    M, N, K, qx, sfx, qw, sfw, alpha = _problem()
    _, want = O.gemm(qx, qw, sfx, sfw, alpha)
    got = np.load(os.path.join(str(tmp_path), f"{mode}.npy"))
    assert got.shape == want.shape
    # fp64 partial sums of exact products: equal up to summation order
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())


def test_k_slices_are_64_aligned_and_cover_k():
    for K, world in [(4160, 8), (640, 4), (8256, 8), (64, 2), (19008, 8)]:
        sl = tp.k_slices(K, world)
        assert sl[0][0] == 0 and sl[-1][1] == K
        assert all(a % 64 == 0 and b % 64 == 0 and a <= b for a, b in sl)
        assert all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))
        widths = [b - a for a, b in sl]
        assert max(widths) - min(widths) <= 64


def test_n_slices_are_128_aligned_and_cover_n():
    for N, world in [(4096, 8), (1024, 8), (384, 2), (200, 2), (28672, 8)]:
        sl = tp.n_slices(N, world)
        assert sl[0][0] == 0 and sl[-1][1] == N
        assert all(a % 128 == 0 or a == N for a, _ in sl)


def test_shard_k_is_a_valid_operand_for_the_oracle():
    """Dequantising a K shard == the matching columns of the dequantised full operand (layout-exact slicing)."""
    M, N, K, qx, sfx, qw, sfw, alpha = _problem()
    full = O.dequant(qw, sfw)
    for k0, k1 in tp.k_slices(K, 5 if K // 64 >= 5 else 2):
        q, sf = tp.shard_k(torch.from_numpy(qw), torch.from_numpy(sfw), k0, k1)
        if k1 > k0:
            assert np.array_equal(O.dequant(q.numpy(), sf.numpy()), full[:, k0:k1])


def test_shard_n_is_a_valid_operand_for_the_oracle():
    M, N, K, qx, sfx, qw, sfw, alpha = _problem()
    full = O.dequant(qw, sfw)
    for n0, n1 in tp.n_slices(N, 3):
        q, sf = tp.shard_n(torch.from_numpy(qw), torch.from_numpy(sfw), n0, n1)
        assert sf.numel() == O.sf_alloc_bytes(n1 - n0, K) or (n1 - n0) % 128 != 0
        assert np.array_equal(O.dequant(q.numpy(), sf.numpy()), full[n0:n1])
