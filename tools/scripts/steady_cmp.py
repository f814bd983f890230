"""Steady-state TFLOP/s of the tile GEMM at 4096^2 / 8192^2 under the ARCQ_* tuning knobs of the environment (tuning aid):
    ARCQ_TILE_STAGGER=1 python tools/scripts/steady_cmp.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from arcquant_amd import agemm
dev = torch.device("cuda:0")
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ARCQ_")) or "default"
for (M, N, KQ) in [(4096, 4096, 4096), (8192, 8192, 8192)]:
    p = bench.make_problem(M, N, KQ, 64, dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f = lambda: agemm.matmul(p["qx"], p["qw"], p["sfx"], p["sfw"], p["alpha"], out=out)
    us = bench.time_events_steady(f, 200 if M == 4096 else 50, 80.0)
    print(f"{tag}: M={M} {us:.1f} us {2.0*M*N*(KQ+64)/us/1e6:.0f} TFLOP/s", flush=True)
