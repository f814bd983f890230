#!/usr/bin/env python3
"""What the tile GEMM's epilogue operands cost on the model's prefill shapes: plain / + bias / + residual / + both, sustained launches, us."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import agemm  # noqa: E402
from bench import make_problem, time_events_steady  # noqa: E402

dev = torch.device("cuda:0")
for (m, n, kq) in [(4096, 10752, 3584), (4096, 3584, 3584), (4096, 3584, 18944), (1024, 4096, 4096)]:
    q = make_problem(m, n, kq, 64, dev)
    g = torch.Generator().manual_seed(1)
    bias = torch.randn(n, generator=g).to(torch.bfloat16).to(dev)
    res = torch.randn(m, n, generator=g).to(torch.bfloat16).to(dev)
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    rec = {"shape": [m, n, kq]}
    for name, kw in (("plain", {}), ("bias", {"bias": bias}), ("residual", {"residual": res}), ("bias+residual", {"bias": bias, "residual": res})):
        rec[name] = round(time_events_steady(lambda: agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"], out=out, **kw), 30, 30.0), 2)
    print(json.dumps(rec), flush=True)
    del q, res, out
    torch.cuda.empty_cache()
# the gate|up GEMM with the SiLU * up epilogue (interleaved gate / up rows), without and with its bias; and the headline shape
q = make_problem(4096, 37888, 3584, 64, dev)
bias = torch.randn(37888, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16).to(dev)
rec = {"shape": [4096, 37888, 3584], "epilogue": "silu*up"}
for name, kw in (("plain", {}), ("bias", {"bias": bias})):
    rec[name] = round(time_events_steady(lambda: agemm.matmul_silu_mul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"], **kw), 20, 30.0), 2)
print(json.dumps(rec), flush=True)
del q
torch.cuda.empty_cache()
q = make_problem(4096, 4096, 4096, 64, dev)
out = torch.empty((4096, 4096), dtype=torch.bfloat16, device=dev)
print(json.dumps({"shape": [4096, 4096, 4096], "plain": round(time_events_steady(lambda: agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"], out=out), 100, 60.0), 2)}), flush=True)
