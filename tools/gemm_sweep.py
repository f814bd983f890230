"""Quick tile-GEMM timing sweep (tuning aid): python tools/gemm_sweep.py [M N KQ]"""
import os, sys, json, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(cfg, M, N, KQ):
    code = f"""
import torch, bench, numpy as np
from arcquant_amd import agemm
from oracle import oracle as O
p = bench.make_problem({M}, {N}, {KQ}, 64, torch.device('cuda:0'))
f = lambda: agemm.matmul(p['qx'], p['qw'], p['sfx'], p['sfw'], p['alpha'])
us = bench.time_events(f, 50, 10)
d = f().float()
# spot check 64 random outputs against fp64 dequant
torch.manual_seed(0)
K = {KQ} + 64
print('RESULT', round(us, 2), round(bench.gemm_flops({M}, {N}, K) / us / 1e6, 1), float(d.abs().mean()))
"""
    env = dict(os.environ, ARCQ_TILE_CFG=str(cfg))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for line in out.stdout.splitlines():
        if line.startswith("RESULT"):
            return line
    return "FAILED " + out.stderr[-400:]

CFGS = tuple(int(c) for c in os.environ.get("SWEEP_CFGS", "1,2,3,4").split(","))

if __name__ == "__main__":
    shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 18944, 3584)]
    if len(sys.argv) == 4:
        shapes = [tuple(int(v) for v in sys.argv[1:])]
    for shp in shapes:
        for cfg in CFGS:
            print(shp, "cfg", cfg, run(cfg, *shp), flush=True)
