"""Quantiser timings (tuning aid): python tools/quant_bench.py [perm]   (perm: a random reorder_index instead of identity)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcquant_amd import agemm

dev = torch.device("cuda:0")
for (M, KQ) in [(4096, 4096), (8192, 8192), (4096, 3584), (4096, 18944), (4, 3584), (4, 18944)]:
    x = bench.outlier_activations(M, KQ, dev)
    idx = torch.arange(KQ, dtype=torch.int16, device=dev)
    if "perm" in sys.argv[1:]:
        idx = torch.randperm(KQ, generator=torch.Generator().manual_seed(7)).to(torch.int16).to(dev)
    wn = torch.ones(KQ, dtype=torch.bfloat16, device=dev)
    K = KQ + 64
    byts = M * KQ * 2 + M * K * 9 / 16
    res = {}
    res["x"] = bench.time_events(lambda: agemm.reorder_quantize_x(x, idx, 64), 30, 5)
    res["w"] = bench.time_events(lambda: agemm.reorder_quantize_w(x, idx, 64), 30, 5)
    if 2048 <= KQ <= 8192:
        res["rms"] = bench.time_events(lambda: agemm.rmsnorm_quantize_x(x, wn, 1e-6, idx, 64), 30, 5)
    res["dyn"] = bench.time_events(lambda: agemm.reorder_quantize_x_dynamic(x, idx, 64), 30, 5)
    print(f"{'perm' if 'perm' in sys.argv[1:] else 'identity'} M={M} KQ={KQ}: " + "  ".join(f"{k}={v:.2f}us ({byts / v / 1e3:.0f} GB/s)" for k, v in res.items()), flush=True)
