#!/usr/bin/env python3
"""A-B of the decode harness's fusion choices on one model build: which linears run their quantiser as the GEMM prologue.
usage: python tools/e2e_fuse_ab.py [layers] [attention ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import e2e  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 28
for fuse in ("", "o", "qkv", "gateup", "o,gateup", "qkv,o,gateup", "qkv,o,gateup,down"):
    os.environ["ARCQ_E2E_FUSE"] = fuse
    for att in (sys.argv[2:] or ["current"]):
        r = e2e.bench_decode("qwen2.5-7b", batch=4, prefill=1024, steps=16, layers=layers, fused=True, attention=att)
        print(json.dumps({"fuse": fuse, "attention": att, "ms_per_step": r["decode_ms_per_step_graph"], "tok_per_s": r["decode_tok_per_s"]}), flush=True)
        torch.cuda.empty_cache()
