#!/usr/bin/env python3
"""Decode-step tok/s of the harness on other model shapes of benchmarks/benchmark_e2e_arc.py:26-77 (fused path, HIP-graph replay,
attention over the full cache).  usage: python tools/e2e_models.py [name:batch ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import e2e  # noqa: E402

specs = sys.argv[1:] or ["llama-3.1-8b:1", "llama-3.1-8b:4", "llama-2-7b:4", "qwen2.5-14b:4"]
for spec in specs:
    name, batch = spec.split(":")
    r = e2e.bench_decode(name, batch=int(batch), prefill=1024, steps=16, fused=True, attention="cache")
    print(json.dumps({k: r[k] for k in ("model", "batch", "layers", "decode_ms_per_step_graph", "decode_tok_per_s", "decode_ms_per_step_eager", "prefill_ms",
                                        "weight_bytes", "hbm_floor_ms_at_8TBps")}), flush=True)
    torch.cuda.empty_cache()
