"""The decode harness (arcquant_amd/e2e.py) on a toy decoder: every variant runs through the HIP
operators, the fused K|V cache append equals the two separate appends, attention over the cache equals attention
over the current tokens at prefill."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu


def _toy():
    from arcquant_amd import e2e
    return e2e, e2e.ModelConfig("toy", num_layers=2, num_heads=4, hidden_size=2048, intermediate_size=4096, vocab_size=512)


def test_prefill_and_decode_variants_agree_where_they_must():
    e2e, cfg = _toy()
    dev = torch.device("cuda:0")
    tok = torch.randint(0, cfg.vocab_size, (2, 8), device=dev)
    nxt = torch.randint(0, cfg.vocab_size, (2, 1), device=dev)
    with torch.no_grad():
        cur = e2e.DecoderModel(cfg, 2, 16, dev, fused=True, attention="current")
        cache = e2e.DecoderModel(cfg, 2, 16, dev, fused=True, attention="cache")          # same seed -> same weights
        a, b = cur.forward(tok, 0), cache.forward(tok, 0)
        assert a.shape == (2, cfg.vocab_size) and torch.isfinite(a.float()).all()
        assert torch.equal(a, b)                      # at prefill the cache holds exactly the current tokens
        # the single strided K|V append wrote what two separate appends write
        L = cur.layers[0]
        assert torch.equal(L["kc"], L["kv"][0]) and torch.equal(L["vc"], L["kv"][1])
        assert float(L["kc"][:, :, :8].abs().sum()) > 0 and float(L["kc"][:, :, 8:].abs().sum()) == 0
        d = cache.forward(nxt, 8)
        assert d.shape == (2, cfg.vocab_size) and torch.isfinite(d.float()).all()
        assert float(cache.layers[1]["vc"][:, :, 8].abs().sum()) > 0
        ref = e2e.DecoderModel(cfg, 2, 16, dev, fused=False)
        r = ref.forward(tok, 0)
        assert r.shape == (2, cfg.vocab_size) and torch.isfinite(r.float()).all()


def test_bench_decode_reports_graph_and_eager_times():
    e2e, cfg = _toy()
    e2e.MODEL_CFGS["toy"] = cfg
    try:
        out = e2e.bench_decode("toy", batch=2, prefill=16, steps=2, repeats=1, fused=True)
    finally:
        del e2e.MODEL_CFGS["toy"]
    assert out["decode_tok_per_s"] > 0 and out["decode_ms_per_step_graph"] > 0 and out["layers"] == 2


def test_reference_protocol_benchmark_runs_graph_and_eager():
    """bench_protocol: the three modules of benchmarks/benchmark_e2e_arc.py (prefill, multi-step decode over a growing
    cache, both) with its warm-up / timed / repeat counts; the graph and the eager decode run the same launches."""
    e2e, cfg = _toy()
    e2e.MODEL_CFGS["toy"] = cfg
    try:
        for graph in (True, False):
            out = e2e.bench_protocol("toy", batch=2, prefill=16, decode_steps=3, repeats=2, warmup=1, steps=2, graph=graph)
            assert out["decode_from_hip_graph"] is graph and out["decode_steps"] == 3
            for key in ("prefill_ms", "decode_ms", "e2e_ms"):
                assert out[key][0] > 0 and out[key][1] >= 0
            assert out["e2e_ms"][0] > out["decode_ms"][0] * 0.5 and out["peak_memory_gb"] > 0
    finally:
        del e2e.MODEL_CFGS["toy"]


def test_fused_decoder_layer_matches_the_oracle_stage_by_stage():
    """One decoder layer of the harness on the fused decode path (three fused launches + the fused down projection), every
    arcq stage against the CPU ORACLE on the stage's own input, BIT FOR BIT:
        q|k|v   = O.rmsnorm_quantize_x -> O.gemm (bf16) + bias
        o_proj  = torch's abs-max / division (GPU semantics) + O.quantize_x -> O.gemm + bias, + residual
        act     = O.rmsnorm_quantize_x -> O.gemm + bias -> silu(gate) * up with torch's two roundings
        down    = abs-max / division + O.quantize_x -> O.gemm + bias, + residual
    and the stages chained by hand reproduce DecoderModel.forward exactly (so forward() IS these ops).  Only the attention
    between them is torch's (flash on the GPU against fp32 math on the CPU: compared within 1e-2).  A chained end-to-end
    tolerance would hide a wrong stage -- one bf16 bit of attention noise already moves the logits by several percent through
    two re-quantisations -- which is why each stage is pinned on its own."""
    import numpy as np
    import torch.nn.functional as F
    from arcquant_amd import agemm as ag
    from oracle import oracle as O
    from tests.util import bits, from_bits
    e2e, _ = _toy()
    cfg = e2e.ModelConfig("toy1", num_layers=1, num_heads=4, hidden_size=2048, intermediate_size=4096, vocab_size=256,
                          attention_bias=True, mlp_bias=True)
    dev = torch.device("cuda:0")
    bsz, q_len = 2, 3
    tok = torch.randint(0, cfg.vocab_size, (bsz, q_len), device=dev)
    with torch.no_grad():
        model = e2e.DecoderModel(cfg, bsz, 8, dev, fused=True, attention="cache")
        model.fuse = {"qkv", "o", "gateup", "down"}
        logits = model.forward(tok, 0)
    L = model.layers[0]
    h, it, ke, nh = cfg.hidden_size, cfg.intermediate_size, cfg.select_num, cfg.num_heads
    hd, T = h // nh, bsz * q_len
    idx_h, idx_i = model.idx_h.cpu().numpy(), model.idx_i.cpu().numpy()

    def oracle_linear(qx, sfx, alpha, lin, residual=None):
        db, _ = O.gemm(qx, lin.W.cpu().numpy(), sfx, lin.SFW.cpu().numpy(), np.float32(alpha) * np.float32(lin.scale_f))
        y = from_bits(db) + lin.bias.cpu()                          # bf16 product, then the bf16 bias add (qLinearLayer.py:74-76)
        return y if residual is None else residual.cpu() + y

    def oracle_dyn_quant(x, idx):                                  # qLlamaLayer.py:73-77 with torch-on-GPU semantics
        x = x.cpu()
        scale = torch.max(x.abs()).float() * torch.tensor(1.0 / 2688.0, dtype=torch.float32)
        xs = (x.float() / scale.to(torch.bfloat16).float()).to(torch.bfloat16)
        q, sf = O.quantize_x(bits(xs), idx, ke, O.G16, sf_fill=0)
        return q, sf, float(scale)

    def same(a, b):
        return torch.equal(a.cpu(), b.cpu())

    with torch.no_grad():
        hcur = model.embed[tok].reshape(T, h)
        Q = L["qkv"]
        qkv = ag.rmsnorm_matmul_repacked(hcur, L["ln1"], cfg.eps, model.idx_h, ke, Q.RW, Q.RSF, Q.scale, Q.out_f, bias=Q.bias)
        qx, sfx = O.rmsnorm_quantize_x(bits(hcur.cpu()), bits(L["ln1"].cpu()), cfg.eps, idx_h, ke, O.G16, sf_fill=0)
        assert same(qkv, oracle_linear(qx, sfx, 1.0, Q))
        q, k, v = (qkv[:, i * h:(i + 1) * h].reshape(bsz, q_len, nh, hd).transpose(1, 2) for i in range(3))
        att = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(T, h)
        att_cpu = F.scaled_dot_product_attention(q.float().cpu(), k.float().cpu(), v.float().cpu(), is_causal=True)
        att_cpu = att_cpu.to(torch.bfloat16).transpose(1, 2).reshape(T, h)
        assert float((att.cpu().float() - att_cpu.float()).norm() / att_cpu.float().norm()) < 1e-2
        O_ = L["o"]
        h2, _ = ag.dynamic_matmul_repacked(att, model.idx_h, ke, O_.RW, O_.RSF, O_.scale_f, O_.out_f, bias=O_.bias, residual=hcur)
        qa, sfa, sa = oracle_dyn_quant(att, idx_h)
        assert same(h2, oracle_linear(qa, sfa, sa, O_, residual=hcur))
        Gt = L["gateup"]
        act, slots = ag.rmsnorm_matmul_repacked_silu(h2, L["ln2"], cfg.eps, model.idx_h, ke, Gt.RW, Gt.RSF, Gt.scale, Gt.out_f, bias=Gt.bias)
        qx, sfx = O.rmsnorm_quantize_x(bits(h2.cpu()), bits(L["ln2"].cpu()), cfg.eps, idx_h, ke, O.G16, sf_fill=0)
        gu = oracle_linear(qx, sfx, 1.0, Gt).float()               # interleaved (g0, u0, g1, u1, ...)
        g_, u_ = gu[:, 0::2], gu[:, 1::2]
        act_cpu = ((g_ / (1.0 + torch.exp(-g_))).to(torch.bfloat16).float() * u_).to(torch.bfloat16)
        assert (bits(act.cpu()) == bits(act_cpu)).mean() > 0.999 and same(act, act_cpu) or \
            float((act.cpu().float() - act_cpu.float()).abs().max()) <= 2.0 ** -7 * float(act_cpu.float().abs().max())   # exp: ocml vs libm
        D_ = L["down"]
        h3, _ = ag.dynamic_matmul_repacked(act, model.idx_i, ke, D_.RW, D_.RSF, D_.scale_f, D_.out_f, absmax_slots=slots, bias=D_.bias, residual=h2)
        qa, sfa, sa = oracle_dyn_quant(act, idx_i)
        assert same(h3, oracle_linear(qa, sfa, sa, D_, residual=h2))
        # the chain above IS what forward() runs
        hn = F.rms_norm(h3.view(bsz, q_len, -1)[:, -1], (h,), model.norm, cfg.eps)
        assert torch.equal(hn @ model.lm_head.t(), logits)


def test_harness_decode_attention_kernel_matches_torch_sdpa():
    """The harness's streaming decode attention (include/arcq_harness.h; NOT part of the drop-in boundary) against torch on the
    same dense bf16 cache: appends k / v at `pos` exactly as the strided copy does and attends over [0, pos]; fp32 math here
    against flash attention's bf16 P: agreement to bf16 rounding.  Positions cover one slice, several slices, slice
    boundaries and the last cache slot."""
    import torch.nn.functional as F
    e2e, _ = _toy()
    cfg = e2e.ModelConfig("toyattn", num_layers=1, num_heads=4, hidden_size=512, intermediate_size=1024, vocab_size=64)
    dev = torch.device("cuda:0")
    bsz, tmax = 3, 1100
    model = e2e.DecoderModel(cfg, bsz, tmax, dev, fused=True, attention="cache")
    L = model.layers[0]
    g = torch.Generator(device=dev).manual_seed(3)
    nh, hd, h = 4, 128, 512
    for pos in (0, 1, 15, 16, 255, 256, 700, 1039, tmax - 1):
        L["kv"].copy_(torch.randn(L["kv"].shape, generator=g, device=dev).to(torch.bfloat16))
        before = L["kv"].clone()
        qkv = torch.randn(bsz, 3 * h, generator=g, device=dev).to(torch.bfloat16)
        got = model._attn_decode_stream(qkv, L, pos)
        # the append: only position `pos` of both caches changed, to this token's k / v
        want_kv = before.clone()
        want_kv[:, :, :, pos:pos + 1] = qkv[:, h:].reshape(bsz, 1, 2, nh, hd).permute(2, 0, 3, 1, 4)
        assert torch.equal(L["kv"], want_kv), pos
        q = qkv[:, :h].reshape(bsz, 1, nh, hd).transpose(1, 2)
        want = F.scaled_dot_product_attention(q.float(), L["kc"][:, :, :pos + 1].float(), L["vc"][:, :, :pos + 1].float())
        want = want.transpose(1, 2).reshape(bsz, h)
        err = float((got.float() - want).abs().max())
        assert err <= 2.0 ** -7 * float(want.abs().max()) + 1e-3, (pos, err)


def test_harness_decode_attention_window_is_the_current_token():
    """attention="current" (benchmarks/modeling_arc.py:169-198: a decode step attends over the tokens of the current call, i.e. the
    one new token): softmax over one key is 1, so the output IS this token's v, bit for bit, in ONE launch; k / v are still
    appended.  A window in the middle of the cache ([first, pos]) against torch over the same slice."""
    import torch.nn.functional as F
    from arcquant_amd import _lib
    e2e, _ = _toy()
    cfg = e2e.ModelConfig("toyattn", num_layers=1, num_heads=4, hidden_size=512, intermediate_size=1024, vocab_size=64)
    dev = torch.device("cuda:0")
    bsz, tmax, nh, hd, h = 3, 600, 4, 128, 512
    model = e2e.DecoderModel(cfg, bsz, tmax, dev, fused=True, attention="current")
    L = model.layers[0]
    g = torch.Generator(device=dev).manual_seed(4)
    for pos in (0, 17, 599):
        L["kv"].copy_(torch.randn(L["kv"].shape, generator=g, device=dev).to(torch.bfloat16))
        before = L["kv"].clone()
        qkv = torch.randn(bsz, 3 * h, generator=g, device=dev).to(torch.bfloat16)
        got = model._attn_decode_stream(qkv, L, pos)
        assert torch.equal(got, qkv[:, 2 * h:]), pos
        want_kv = before.clone()
        want_kv[:, :, :, pos:pos + 1] = qkv[:, h:].reshape(bsz, 1, 2, nh, hd).permute(2, 0, 3, 1, 4)
        assert torch.equal(L["kv"], want_kv), pos
    lib = _lib.lib()
    for first, pos in ((100, 420), (299, 300), (0, 599)):
        qkv = torch.randn(bsz, 3 * h, generator=g, device=dev).to(torch.bfloat16)
        out = torch.empty((bsz, h), dtype=torch.bfloat16, device=dev)
        st = lib.arcq_harness_attn_decode_window(qkv.data_ptr(), L["kc"].data_ptr(), L["vc"].data_ptr(), out.data_ptr(), model._attn_ws.data_ptr(),
                                                 bsz, nh, tmax, pos, first, torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "window")
        q = qkv[:, :h].reshape(bsz, 1, nh, hd).transpose(1, 2)
        want = F.scaled_dot_product_attention(q.float(), L["kc"][:, :, first:pos + 1].float(), L["vc"][:, :, first:pos + 1].float())
        want = want.transpose(1, 2).reshape(bsz, h)
        err = float((out.float() - want).abs().max())
        assert err <= 2.0 ** -7 * float(want.abs().max()) + 1e-3, (first, pos, err)


def test_harness_final_rmsnorm_matches_torch():
    """The harness's one-launch final RMSNorm (include/arcq_harness.h; a stock module in the reference, not an ARC operator) against
    an fp64 statement of the same formula: bf16 rounding of the exact value (<= 1 ulp), strided rows as the prefill's last-token
    slice has them."""
    from arcquant_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(9)
    for rows, H, ld_mult in ((4, 3584, 1), (4, 3584, 1024), (1, 4096, 1), (7, 512, 3)):
        full = (torch.randn(rows, ld_mult, H, generator=g, device=dev) * 3).to(torch.bfloat16)
        x = full[:, -1]                                                # [rows, H], row stride ld_mult * H
        w = (torch.rand(H, generator=g, device=dev) + 0.5).to(torch.bfloat16)
        out = torch.empty((rows, H), dtype=torch.bfloat16, device=dev)
        st = _lib.lib().arcq_harness_rmsnorm(x.data_ptr(), x.stride(0), w.data_ptr(), out.data_ptr(), rows, H, 1e-6, torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "harness rmsnorm")
        xd = x.double()
        want = xd * torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-6) * w.double()
        err = (out.double() - want).abs()
        assert bool((err <= want.abs() * 2.0 ** -7 + 1e-30).all()), (rows, H, float(err.max()))
