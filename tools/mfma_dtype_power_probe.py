#!/usr/bin/env python3
"""Does the matrix pipe sustain a higher clock on bf16 than on fp16 operands, and on operands with few significant bits?  The library GEMM
(hipBLASLt via torch.matmul) at the headline shape on (a) randn data, (b) the dequantised NVFP4 operands of the headline problem (<= 6
significant bits per value), in fp16 and in bf16 (both exact for (b)); sustained launches.  usage: python tools/mfma_dtype_power_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem, time_events_steady, gemm_flops  # noqa: E402

dev = torch.device("cuda:0")
M = N = 4096
KQ, KE = 4096, 64
K = KQ + KE
q = make_problem(M, N, KQ, KE, dev)


def dequant(Q, SF, rows):
    """packed e2m1 + swizzled ue4m3 -> fp32 [rows, K] (format spec, torch ops)."""
    lut = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6, -0., -.5, -1, -1.5, -2, -3, -4, -6], dtype=torch.float32, device=dev)
    codes = torch.stack([Q & 15, Q >> 4], dim=-1).reshape(rows, K).long()
    r = torch.arange(rows, device=dev).unsqueeze(1)
    g = torch.arange(K // 16, device=dev).unsqueeze(0)
    off = ((r // 128) * (K // 64) + g // 4) * 512 + (r % 32) * 16 + ((r // 32) % 4) * 4 + g % 4
    sc = SF[off].view(torch.float8_e4m3fn).float()
    return lut[codes] * sc.repeat_interleave(16, dim=1)


a32, b32 = dequant(q["qx"], q["sfx"], M), dequant(q["qw"], q["sfw"], N)
out = {}
for name, (a, b) in {"randn": (torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)), "dequantised_nvfp4": (a32, b32)}.items():
    for dt in (torch.float16, torch.bfloat16):
        x, w = a.to(dt), b.to(dt)
        exact = bool(torch.equal(x.float(), a)) and bool(torch.equal(w.float(), b)) if name != "randn" else None
        t = min(time_events_steady(lambda: torch.matmul(x, w.t()), 50, 60.0) for _ in range(2))
        out[f"{name}_{str(dt).split('.')[-1]}"] = {"us": round(t, 2), "TFLOPs": round(gemm_flops(M, N, K) / t / 1e6, 1), "operands_exact": exact}
print(json.dumps(out))
