#!/usr/bin/env python3
"""The quantiser launches of one prefill layer of the Qwen2.5-7B-shape harness (M = 4096 tokens), each timed alone, sustained, us:
RMSNorm -> quantise (hidden 3584), dynamic quantise of the attention output (3584; abs-max pass + quantiser) and of the MLP activation
(18944, abs-max words from the GEMM epilogue), with the bytes each one moves."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import agemm  # noqa: E402
from bench import time_events_steady  # noqa: E402

dev = torch.device("cuda:0")
M, KE = 4096, 64
g = torch.Generator().manual_seed(3)
for name, kq in (("hidden", 3584), ("intermediate", 18944)):
    x = torch.randn(M, kq, generator=g).to(torch.bfloat16).to(dev)
    idx = torch.randperm(kq, generator=g).to(torch.int16).to(dev)
    wn = torch.rand(kq, generator=g).to(torch.bfloat16).to(dev)
    byt = M * kq * 2 + M * (kq + KE) * 9 // 16
    rec = {"row": name, "KQ": kq, "MB": round(byt / 1e6, 1)}
    if kq <= 8192:
        t = time_events_steady(lambda: agemm.rmsnorm_quantize_x(x, wn, 1e-6, idx, KE), 30, 20.0)
        rec["rmsnorm_quantize_x_us"] = round(t, 2); rec["rmsnorm_TBps"] = round(byt / t / 1e6, 2)
    t = time_events_steady(lambda: agemm.reorder_quantize_x(x, idx, KE), 30, 20.0)
    rec["reorder_quantize_x_us"] = round(t, 2); rec["static_TBps"] = round(byt / t / 1e6, 2)
    t = time_events_steady(lambda: agemm.reorder_quantize_x_dynamic(x, idx, KE), 30, 20.0)
    rec["reorder_quantize_x_dynamic_us (abs-max pass + quantiser)"] = round(t, 2)
    slots = torch.full((2368,), 0x4000, dtype=torch.int32, device=dev)
    t = time_events_steady(lambda: agemm.reorder_quantize_x_dynamic(x, idx, KE, absmax_slots=slots), 30, 20.0)
    rec["reorder_quantize_x_dynamic_us (abs-max words given)"] = round(t, 2); rec["dynamic_TBps"] = round(byt / t / 1e6, 2)
    t = time_events_steady(lambda: agemm.reorder_quantize_x_dynamic(x, None, KE, absmax_slots=slots), 30, 20.0)
    rec["... input already in reordered channel order (no gather)"] = round(t, 2)
    print(json.dumps(rec), flush=True)
