"""How the duration of the headline GEMM (and of the fp16 rocBLAS GEMM of the same shape) evolves under sustained load
(tuning aid): back-to-back launches, one HIP-event pair per block of 50.  python tools/clock_transient.py [launches]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcquant_amd import agemm

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
M = N = KQ = 4096
p = bench.make_problem(M, N, KQ, 64, dev)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
a16 = torch.randn(M, KQ + 64, device=dev, dtype=torch.float16)
b16 = torch.randn(N, KQ + 64, device=dev, dtype=torch.float16)
o16 = torch.empty((M, N), dtype=torch.float16, device=dev)


def run(name, fn, flops):
    torch.cuda.synchronize()
    import time
    time.sleep(1.0)                                   # idle first: the transient starts from a cold power state
    blocks = n // 50
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    ev[0].record()
    for b in range(blocks):
        for _ in range(50):
            fn()
        ev[b + 1].record()
    torch.cuda.synchronize()
    us = [ev[b].elapsed_time(ev[b + 1]) * 1e3 / 50 for b in range(blocks)]
    print(name, "us per launch, blocks of 50:", " ".join(f"{u:.0f}" for u in us), flush=True)
    print(name, f"first 200: {flops / (sum(us[:4]) / 4) / 1e6:.0f} TFLOP/s, last 500: {flops / (sum(us[-10:]) / 10) / 1e6:.0f} TFLOP/s", flush=True)


fl = 2.0 * M * N * (KQ + 64)
run("arc-nvfp4", lambda: agemm.matmul(p["qx"], p["qw"], p["sfx"], p["sfw"], p["alpha"], out=out), fl)
run("fp16 rocBLAS", lambda: torch.matmul(a16, b16.t(), out=o16), fl)
run("arc-nvfp4", lambda: agemm.matmul(p["qx"], p["qw"], p["sfx"], p["sfw"], p["alpha"], out=out), fl)
