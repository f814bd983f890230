"""Fixed cost and per-K-step cost of the tile GEMM at M=N=4096 in steady state (tuning aid): python tools/scripts/k_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from arcquant_amd import agemm
dev = torch.device("cuda:0")
M = N = 4096
res = []
for KQ in (448, 960, 1984, 4032, 8128):
    p = bench.make_problem(M, N, KQ, 64, dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f = lambda: agemm.matmul(p["qx"], p["qw"], p["sfx"], p["sfw"], p["alpha"], out=out)
    us = bench.time_events_steady(f, 200, 80.0)
    steps = (KQ + 64) // 64
    res.append((steps, us))
    print(f"K_aug={KQ+64} steps={steps}: {us:.1f} us  {2.0*M*N*(KQ+64)/us/1e6:.0f} TFLOP/s", flush=True)
(s0, t0), (s1, t1) = res[1], res[-1]
slope = (t1 - t0) / (s1 - s0)
print(f"slope {slope:.3f} us per K-step, intercept {t0 - slope * s0:.1f} us")
