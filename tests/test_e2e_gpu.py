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
