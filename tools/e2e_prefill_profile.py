#!/usr/bin/env python3
"""Prefill of the Qwen2.5-7B-shape harness (bs = 4 x 1024 tokens) N times, so that `rocprofv3 --kernel-trace -- python3 tools/e2e_prefill_profile.py`
shows where a prefill's time goes (the kernels of the one-off model build appear 28 x 4 times, those of prefill N x 28 x k times).
usage: e2e_prefill_profile.py [repeats]"""
import dataclasses
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import e2e  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = dataclasses.replace(e2e.MODEL_CFGS["qwen2.5-7b"])
dev = torch.device("cuda:0")
with torch.no_grad():
    model = e2e.DecoderModel(cfg, 4, 1024 + 8, dev, fused=True, attention="cache")
    tok = torch.randint(100, 200, (4, 1024), device=dev)
    model.forward(tok, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.forward(tok, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
print(json.dumps({"prefill_ms": round(ms, 2), "prefill_tok_per_s": round(4096 / ms * 1e3, 0), "repeats": reps}))
