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


# ---------------------------------------------------------------------------- column -> row hand-off
def _torch_gpu_scale_and_prescale(x, amax_bits):
    """scale = max|x| / 2688 and bf16(x / scale) with torch-on-GPU semantics (see tests/test_gpu_parity._oracle_dyn_quant)."""
    amax = torch.tensor([amax_bits], dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float()[0]
    scale = amax * torch.tensor(1.0 / 2688.0, dtype=torch.float32)
    xs = (x.float() / scale.to(torch.bfloat16).float()).to(torch.bfloat16)
    return float(scale), xs


def _mlp_problem(world):
    """A column-parallel linear (N_inter = 1024) followed by a row-parallel one (K = 1024 -> 192), M = 5 tokens."""
    M, H, NI, N2, KE = 5, 256, 1024, 192, 64
    g = torch.Generator().manual_seed(11)
    x, sx = prescale(outlier_activations(M, H, 12))
    w1, s1 = prescale((torch.rand(NI, H, generator=g) * 2 - 1).to(torch.bfloat16))
    w2 = (torch.rand(N2, NI, generator=g) * 2 - 1).to(torch.bfloat16)
    idx1 = random_perm(H, 13).numpy()
    idx2 = random_perm(NI, 14).numpy()                              # hand-off A: a reorder_index over the WHOLE intermediate row
    shard = NI // world
    local_idx = [random_perm(shard, 20 + r).numpy() for r in range(world)]       # hand-off B: one per shard
    qx, sfx = O.quantize_x(bits(x), idx1, KE, O.G16, sf_fill=0)
    qw1, sfw1 = O.quantize_w(bits(w1), idx1, KE, O.G16, sf_fill=0)
    return dict(M=M, H=H, NI=NI, N2=N2, KE=KE, qx=qx, sfx=sfx, qw1=qw1, sfw1=sfw1, alpha1=float(sx * s1), w2=w2, idx2=idx2,
                local_idx=local_idx, shard=shard)


def _handoff_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = _mlp_problem(world)
        M, NI, KE = P["M"], P["NI"], P["KE"]
        # column-parallel first linear on this rank's rows of W1 (scale tiles stay whole)
        n0, n1 = tp.n_slices(NI, world)[rank]
        b, sfb = tp.shard_n(torch.from_numpy(P["qw1"]), torch.from_numpy(P["sfw1"]), n0, n1)
        yb, _ = O.gemm(P["qx"], b.numpy(), P["sfx"], sfb.numpy(), P["alpha1"])
        y_local = torch.from_numpy(yb.view(np.int16)).view(torch.bfloat16)                      # bf16 [M, NI / world]
        widths = [e - s_ for s_, e in tp.n_slices(NI, world)]
        # ---- hand-off A: all-gather, quantise the full row replicated, slice K
        y_full = tp.handoff_gather(y_local, widths)
        scale_a, xs = _torch_gpu_scale_and_prescale(y_full, int(tp.absmax_word(y_full)))
        qa, sfa = O.quantize_x(bits(xs), P["idx2"], KE, O.G16, sf_fill=0)
        w2s, s2 = prescale(P["w2"])
        qw2, sfw2 = O.quantize_w(bits(w2s), P["idx2"], KE, O.G16, sf_fill=0)
        k0, k1 = tp.k_slices(NI + KE, world)[rank]
        a_sh, sfa_sh = tp.shard_k(torch.from_numpy(qa), torch.from_numpy(sfa), k0, k1)
        b_sh, sfb_sh = tp.shard_k(torch.from_numpy(qw2), torch.from_numpy(sfw2), k0, k1)
        _, part = O.gemm(a_sh.numpy(), b_sh.numpy(), sfa_sh.numpy(), sfb_sh.numpy(), np.float32(scale_a) * np.float32(float(s2)))
        out_a = torch.from_numpy(part)
        dist.all_reduce(out_a, op=dist.ReduceOp.SUM)
        # ---- hand-off B: only the abs-max word travels; shard-local reorder_index and residual channels
        word = tp.handoff_local_scale(y_local)
        scale_b, xs_l = _torch_gpu_scale_and_prescale(y_local, int(word))
        lidx = P["local_idx"][rank]
        ql, sfl = O.quantize_x(bits(xs_l), lidx, KE, O.G16, sf_fill=0)
        w2_l = (P["w2"][:, n0:n1].float() / float(s2)).to(torch.bfloat16)                       # this rank's input channels, the layer's scale
        qwl, sfwl = O.quantize_w(bits(w2_l.contiguous()), lidx, KE, O.G16, sf_fill=0)
        _, part_b = O.gemm(ql, qwl, sfl, sfwl, np.float32(scale_b) * np.float32(float(s2)))
        out_b = torch.from_numpy(part_b)
        dist.all_reduce(out_b, op=dist.ReduceOp.SUM)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), y_local=bits(y_local), qa=qa, sfa=sfa, scale_a=scale_a, out_a=out_a.numpy(),
                 word=int(word), ql=ql, sfl=sfl, scale_b=scale_b, out_b=out_b.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_column_to_row_handoff_both_ways_against_the_unsharded_oracle(tmp_path, world):
    port = _free_port()
    mp.spawn(_handoff_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    P = _mlp_problem(world)
    M, NI, KE = P["M"], P["NI"], P["KE"]
    # unsharded reference: the whole layer pair through the oracle
    yb, _ = O.gemm(P["qx"], P["qw1"], P["sfx"], P["sfw1"], P["alpha1"])
    y = torch.from_numpy(yb.view(np.int16)).view(torch.bfloat16)
    amax_bits = int(tp.absmax_word(y))
    scale, xs = _torch_gpu_scale_and_prescale(y, amax_bits)
    qa, sfa = O.quantize_x(bits(xs), P["idx2"], KE, O.G16, sf_fill=0)
    w2s, s2 = prescale(P["w2"])
    qw2, sfw2 = O.quantize_w(bits(w2s), P["idx2"], KE, O.G16, sf_fill=0)
    _, want_a = O.gemm(qa, qw2, sfa, sfw2, np.float32(scale) * np.float32(float(s2)))
    dense = (xs.float() * scale) @ P["w2"].float().t()
    want_b = np.zeros_like(want_a)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        n0, n1 = tp.n_slices(NI, world)[r]
        assert np.array_equal(g["y_local"], bits(y[:, n0:n1]))
        # A: EVERY rank holds the unsharded quantiser's bytes and scale
        assert np.array_equal(g["qa"], qa) and np.array_equal(g["sfa"], sfa) and float(g["scale_a"]) == scale
        assert np.allclose(g["out_a"], want_a, rtol=1e-12, atol=1e-12 * np.abs(want_a).max())
        # B: the global abs-max word, and this rank's bytes == the unsharded quantiser on its column slice with the global scale
        assert int(g["word"]) == amax_bits and float(g["scale_b"]) == scale
        ql, sfl = O.quantize_x(bits(xs[:, n0:n1].contiguous()), P["local_idx"][r], KE, O.G16, sf_fill=0)
        assert np.array_equal(g["ql"], ql) and np.array_equal(g["sfl"], sfl)
        w2_l = (P["w2"][:, n0:n1].float() / float(s2)).to(torch.bfloat16)
        qwl, sfwl = O.quantize_w(bits(w2_l.contiguous()), P["local_idx"][r], KE, O.G16, sf_fill=0)
        want_b += O.gemm(ql, qwl, sfl, sfwl, np.float32(scale) * np.float32(float(s2)))[1]
    g0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    assert np.allclose(g0["out_b"], want_b, rtol=1e-12, atol=1e-12 * np.abs(want_b).max())
    # both are ARC-NVFP4 approximations of the dense layer of the same quality (fp4 weights: ~10 % of the output norm on uniform weights)
    err_a = np.linalg.norm(want_a - dense.numpy()) / np.linalg.norm(dense.numpy())
    err_b = np.linalg.norm(want_b - dense.numpy()) / np.linalg.norm(dense.numpy())
    assert err_a < 0.2 and err_b < 0.2 and abs(err_a - err_b) < 0.02, (err_a, err_b)
    assert tp.handoff_bytes_per_rank(M, NI, world, "gather") == (world - 1) * M * (NI // world) * 2
    assert tp.handoff_bytes_per_rank(M, NI, world, "local_scale") == 4 * (world - 1)


# ---------------------------------------------------------------------------- sharded linears WITH a bias
def _bias_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import oracle_ops as OPS
        M, N, K, qx, sfx, qw, sfw, alpha = _problem()
        g = torch.Generator().manual_seed(99)
        bias = (torch.randn(N, generator=g) * 4).to(torch.bfloat16)
        res = (torch.randn(M, N, generator=g) * 4).to(torch.bfloat16)
        QX, SFX, QW, SFW = (torch.from_numpy(a) for a in (qx, sfx, qw, sfw))
        col = tp.ColumnParallelARCLinear(QW, SFW, 1.0, rank, world, bias=bias, ops=OPS)
        y_col = col.forward(QX, SFX, alpha, gather_output=True)
        row = tp.RowParallelARCLinear(QW, SFW, 1.0, rank, world, bias=bias, ops=OPS)
        y_row = row.forward(QX, SFX, alpha, residual=res)
        if rank == 0:
            np.savez(os.path.join(out_dir, "bias.npz"), col=bits(y_col), row=bits(y_row))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_linears_with_bias_round_like_the_unsharded_layer(tmp_path, world):
    """Column-parallel: the bias rides in each shard's GEMM epilogue; row-parallel: it is added AFTER the reduction, with the
    single-GPU epilogue's roundings bf16(bf16(sum) + bias), then bf16(residual + y) (model/qLinearLayer.py:74-76)."""
    mp.spawn(_bias_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    M, N, K, qx, sfx, qw, sfw, alpha = _problem()
    g = torch.Generator().manual_seed(99)
    bias = (torch.randn(N, generator=g) * 4).to(torch.bfloat16)
    res = (torch.randn(M, N, generator=g) * 4).to(torch.bfloat16)
    yb, _ = O.gemm(qx, qw, sfx, sfw, alpha)
    y = torch.from_numpy(yb.view(np.int16)).view(torch.bfloat16)
    want_col = y + bias
    want_row = res + (y + bias)
    got = np.load(os.path.join(str(tmp_path), "bias.npz"))
    assert np.array_equal(got["col"], bits(want_col))                      # same products per column, same epilogue: bit-exact
    d = np.abs(got["row"].astype(np.int32) - bits(want_row).astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.01                           # fp32 partials of the K slices: at most a rounding-boundary flip
