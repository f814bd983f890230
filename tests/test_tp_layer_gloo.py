"""A tensor-parallel decoder layer (arcquant_amd/tp.py::TPDecoderLayer: column-parallel q|k|v and gate|up, row-parallel o and down
with hand-off B, two all-reduce(MAX) + two all-reduce(SUM) per layer) over world_size-2 / 4 gloo ranks, against an UNSHARDED
computation from the dense weights, stage by stage.

The operator module is tests/oracle_ops.py (the CPU oracle behind agemm's function names; the product GEMM needs a GPU), the
collectives are real torch.distributed calls.  Each stage is checked on the TP layer's own previous-stage output (gathered over
the ranks), as tests/test_e2e_gpu.py does: requantisation amplifies a one-ulp difference, so chaining would only test noise.
  q|k|v, SiLU*up, abs-max words: bit-exact (row slices of the same quantised weight; element-wise torch ops)
  attention: harness glue, within bf16 rounding of an fp32 softmax over the unsharded heads
  o / down: the all-reduced fp32 partials against ONE oracle GEMM over the K-concatenated operands (the unsharded layer whose
            reorder_index is the concatenation of the local ones), within the rounding of the partials to fp32."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from arcquant_amd import tp
from oracle import oracle as O
from tests import oracle_ops as OPS
from tests.util import bits, from_bits, outlier_activations, random_perm

CFG = dict(hidden=2048, heads=16, kv_heads=4, head_dim=128, inter=1024)        # RMSNorm path: 2048 <= hidden <= 8192 (rmsnorm.cu:241-246)
M, KE, KE_O, KE_D, EPS = 3, 64, 64, 128, 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dense(world):
    g = torch.Generator().manual_seed(1234)
    h, hd = CFG["hidden"], CFG["head_dim"]
    hq, hk, it = CFG["heads"] * hd, CFG["kv_heads"] * hd, CFG["inter"]

    def rnd(n, k):
        return ((torch.rand(n, k, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)

    dense = dict(wq=rnd(hq, h), wk=rnd(hk, h), wv=rnd(hk, h), wo=rnd(h, hq), wg=rnd(it, h), wu=rnd(it, h), wd=rnd(h, it))
    scales = dict(wqkv=float(torch.cat([dense["wq"], dense["wk"], dense["wv"]]).max().float() / 2688.0), wo=float(dense["wo"].max().float() / 2688.0),
                  wgu=float(torch.cat([dense["wg"], dense["wu"]]).max().float() / 2688.0), wd=float(dense["wd"].max().float() / 2688.0))
    ln1 = (torch.rand(h, generator=g) + 0.5).to(torch.bfloat16)
    ln2 = (torch.rand(h, generator=g) + 0.5).to(torch.bfloat16)
    idx_h = random_perm(h, 77)
    idx_o = [random_perm(hq // world, 100 + r) for r in range(world)]
    idx_d = [random_perm(it // world, 200 + r) for r in range(world)]
    hs = [outlier_activations(M, h, 300 + s) * 0.1 for s in range(2)]       # the layer input of decode step 0 and 1
    return dense, scales, ln1, ln2, idx_h, idx_o, idx_d, hs


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dense, scales, ln1, ln2, idx_h, idx_o, idx_d, hs = _dense(world)
        shards = tp.TPDecoderLayer.shard_weights(dense, rank, world, CFG["heads"], CFG["kv_heads"], CFG["head_dim"])
        layer = tp.TPDecoderLayer.build(shards, ln1, ln2, idx_h, idx_o[rank], idx_d[rank], KE, KE_O, KE_D, rank, world, CFG["heads"], CFG["kv_heads"],
                                        CFG["head_dim"], M, 8, eps=EPS, scales=scales, ops=OPS, repack=False)
        rec = {}
        for step, h in enumerate(hs):
            out = layer.forward(h.to(torch.bfloat16), step, trace=True)
            for k, v in layer.trace.items():
                rec[f"s{step}_{k}"] = bits(v) if v.dtype == torch.bfloat16 else v.numpy()
            assert torch.equal(out, layer.trace["h2"])
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **rec)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _ulps(a_bits, b_bits):
    """distance in bf16 code points between two bf16 bit arrays of equal sign pattern (0 where bit-identical)"""
    a = a_bits.astype(np.int32)
    b = b_bits.astype(np.int32)
    a = np.where(a & 0x8000, 0x8000 - (a & 0x7FFF), 0x8000 + a)
    b = np.where(b & 0x8000, 0x8000 - (b & 0x7FFF), 0x8000 + b)
    return np.abs(a - b)


@pytest.mark.parametrize("world", [2, 4])
def test_tp_decoder_layer_matches_the_unsharded_layer_stage_by_stage(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    dense, scales, ln1, ln2, idx_h, idx_o, idx_d, hs = _dense(world)
    R = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    h, hd, nh, nkv, it = CFG["hidden"], CFG["head_dim"], CFG["heads"], CFG["kv_heads"], CFG["inter"]
    hq_l, hk_l, it_l = nh // world * hd, nkv // world * hd, it // world

    def qw(w, s, idx, ke):
        return O.quantize_w(bits((w.float() / s).to(torch.bfloat16)), idx.numpy(), ke, OPS.variant_for_kq(w.shape[1]), sf_fill=0)

    # the unsharded weights, quantised as a whole with the layer's per-tensor scales
    QKV = qw(torch.cat([dense["wq"], dense["wk"], dense["wv"]]), scales["wqkv"], idx_h, KE)
    G, U = qw(dense["wg"], scales["wgu"], idx_h, KE), qw(dense["wu"], scales["wgu"], idx_h, KE)
    k_cache = np.zeros((2, M, nkv, hd), np.float32)
    v_cache = np.zeros((2, M, nkv, hd), np.float32)
    for step, hin in enumerate(hs):
        hin = hin.to(torch.bfloat16)
        # ---- q|k|v: every rank's columns are columns of the unsharded projection, bit for bit
        A, SFA = O.rmsnorm_quantize_x(bits(hin), bits(ln1), EPS, idx_h.numpy(), KE, O.G16, sf_fill=0)
        full, _ = O.gemm(A, QKV[0], SFA, QKV[1], np.float32(scales["wqkv"]))
        qf, kf, vf = full[:, : nh * hd], full[:, nh * hd: (nh + nkv) * hd], full[:, (nh + nkv) * hd:]
        for r in range(world):
            got = R[r][f"s{step}_qkv"]
            assert np.array_equal(got[:, :hq_l], qf[:, r * hq_l:(r + 1) * hq_l])
            assert np.array_equal(got[:, hq_l: hq_l + hk_l], kf[:, r * hk_l:(r + 1) * hk_l])
            assert np.array_equal(got[:, hq_l + hk_l:], vf[:, r * hk_l:(r + 1) * hk_l])
        # ---- attention over the unsharded heads, fp32 softmax (GQA: nh / nkv query heads per KV head)
        q = from_bits(qf).float().reshape(M, nkv, nh // nkv, hd)
        k_cache[step] = from_bits(kf).float().reshape(M, nkv, hd).numpy()
        v_cache[step] = from_bits(vf).float().reshape(M, nkv, hd).numpy()
        kc = torch.from_numpy(k_cache[: step + 1]).permute(1, 2, 0, 3)            # [M, nkv, T, hd]
        vc = torch.from_numpy(v_cache[: step + 1]).permute(1, 2, 0, 3)
        p = torch.softmax(torch.einsum("bgrd,bgtd->bgrt", q, kc) * hd ** -0.5, dim=-1)
        att_ref = torch.einsum("bgrt,bgtd->bgrd", p, vc).reshape(M, nh * hd).to(torch.bfloat16)
        att = np.concatenate([R[r][f"s{step}_att"] for r in range(world)], axis=1)      # rank-major = head order
        af, rf = from_bits(att).float().numpy(), att_ref.float().numpy()
        assert np.all(np.abs(af - rf) <= 2.0 ** -7 * np.abs(rf) + 2.0 ** -9 * np.abs(rf).max())      # bf16 softmax weights inside torch's kernel
        att_t = from_bits(att)
        # ---- hand-off B + o_proj: the global word, then ONE GEMM over the K-concatenated shards
        word = int(tp.absmax_word(att_t))
        assert all(int(R[r][f"s{step}_word_o"][0]) == word for r in range(world))
        scale, xs = OPS.dyn_scale_and_prescale(att_t, word)
        xa, wa = [], []
        for r in range(world):
            ql, sfl = O.quantize_x(bits(xs[:, r * hq_l:(r + 1) * hq_l].contiguous()), idx_o[r].numpy(), KE_O, O.G16, sf_fill=0)
            xa.append((torch.from_numpy(ql), torch.from_numpy(sfl)))
            wl = qw(dense["wo"][:, r * hq_l:(r + 1) * hq_l].contiguous(), scales["wo"], idx_o[r], KE_O)
            wa.append((torch.from_numpy(wl[0]), torch.from_numpy(wl[1])))
        XA, WA = tp.concat_k(xa), tp.concat_k(wa)
        assert XA[0].shape[1] * 2 == nh * hd + world * KE_O
        yb, ye = O.gemm(XA[0].numpy(), WA[0].numpy(), XA[1].numpy(), WA[1].numpy(), np.float32(float(scale)) * np.float32(scales["wo"]))
        h1_ref = hin + from_bits(yb)
        for r in range(world):
            assert np.array_equal(R[r][f"s{step}_h1"], R[0][f"s{step}_h1"])             # replicated, bit-identical
        d = _ulps(R[0][f"s{step}_h1"], bits(h1_ref))
        assert d.max() <= 1 and (d > 0).mean() < 0.01      # fp32 partials: a sum may land on the other side of a bf16 rounding boundary
        h1 = from_bits(R[0][f"s{step}_h1"])
        # ---- MLP first half on the TP layer's own h1: SiLU*up and the local abs-max words, bit for bit
        A2, SFA2 = O.rmsnorm_quantize_x(bits(h1), bits(ln2), EPS, idx_h.numpy(), KE, O.G16, sf_fill=0)
        gb, _ = O.gemm(A2, G[0], SFA2, G[1], np.float32(scales["wgu"]))
        ub, _ = O.gemm(A2, U[0], SFA2, U[1], np.float32(scales["wgu"]))
        act_ref = torch.nn.functional.silu(from_bits(gb)) * from_bits(ub)
        act = np.concatenate([R[r][f"s{step}_act"] for r in range(world)], axis=1)
        assert np.array_equal(act, bits(act_ref))
        for r in range(world):
            assert int(R[r][f"s{step}_word_local"][0]) == int(tp.absmax_word(act_ref[:, r * it_l:(r + 1) * it_l].contiguous()))
        word_d = int(tp.absmax_word(act_ref))
        assert all(int(R[r][f"s{step}_word_d"][0]) == word_d for r in range(world))
        # ---- down_proj
        scale_d, xs_d = OPS.dyn_scale_and_prescale(act_ref, word_d)
        xa, wa = [], []
        for r in range(world):
            var = OPS.variant_for_kq(it_l)
            ql, sfl = O.quantize_x(bits(xs_d[:, r * it_l:(r + 1) * it_l].contiguous()), idx_d[r].numpy(), KE_D, var, sf_fill=0)
            xa.append((torch.from_numpy(ql), torch.from_numpy(sfl)))
            wl = qw(dense["wd"][:, r * it_l:(r + 1) * it_l].contiguous(), scales["wd"], idx_d[r], KE_D)
            wa.append((torch.from_numpy(wl[0]), torch.from_numpy(wl[1])))
        XA, WA = tp.concat_k(xa), tp.concat_k(wa)
        yb, _ = O.gemm(XA[0].numpy(), WA[0].numpy(), XA[1].numpy(), WA[1].numpy(), np.float32(float(scale_d)) * np.float32(scales["wd"]))
        h2_ref = h1 + from_bits(yb)
        for r in range(world):
            assert np.array_equal(R[r][f"s{step}_h2"], R[0][f"s{step}_h2"])
        d = _ulps(R[0][f"s{step}_h2"], bits(h2_ref))
        assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_concat_k_inverts_shard_k_and_gather_rows_moves_whole_tiles():
    g = torch.Generator().manual_seed(5)
    N, KQ, KE = 384, 512, 128
    w = ((torch.rand(N, KQ, generator=g) * 2 - 1)).to(torch.bfloat16)
    idx = random_perm(KQ, 6)
    q, sf = O.quantize_w(bits(w), idx.numpy(), KE, O.G16, sf_fill=0)
    Q, SF = torch.from_numpy(q), torch.from_numpy(sf)
    parts = [tp.shard_k(Q, SF, a, b) for a, b in tp.k_slices(KQ + KE, 3)]
    Q2, SF2 = tp.concat_k(parts)
    assert torch.equal(Q2, Q) and np.array_equal(O.dequant(Q2.numpy(), SF2.numpy()), O.dequant(q, sf))
    # GQA-style shard: rows [128, 256) and [0, 128) of the weight as one operand
    Qg, SFg = tp.gather_rows(Q, SF, [(128, 256), (0, 128)])
    full = O.dequant(q, sf)
    assert np.array_equal(O.dequant(Qg.numpy(), SFg.numpy()), np.concatenate([full[128:256], full[0:128]]))
    assert SFg.numel() == O.sf_alloc_bytes(256, KQ + KE)
    with pytest.raises(ValueError):
        tp.gather_rows(Q, SF, [(64, 128)])


def test_finish_row_parallel_rounds_like_the_single_gpu_epilogue():
    """bf16(sum), then bf16(y + bias), then bf16(residual + y): three roundings, in this order."""
    total = torch.tensor([[1.00390625, 3.0e-3, -2.5]], dtype=torch.float32)
    bias = torch.tensor([0.00390625, 1.0, 2.5], dtype=torch.bfloat16)
    res = torch.tensor([[0.0, 1.0, 1.0]], dtype=torch.bfloat16)
    y = tp.finish_row_parallel(total.clone(), bias, res)
    want = (res.float() + (total.to(torch.bfloat16).float() + bias.float()).to(torch.bfloat16).float()).to(torch.bfloat16)
    assert torch.equal(y, want)
    one_rounding = (total + bias.float() + res.float()).to(torch.bfloat16)
    assert not torch.equal(y, one_rounding)                  # the first element separates the two rules
