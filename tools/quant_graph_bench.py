"""Quantiser device time by HIP-graph replay (no host pacing): python tools/quant_graph_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcquant_amd import agemm
from tools.decode_stream_bench import graph_time

dev = torch.device("cuda:0")
for (M, KQ) in [(4096, 4096), (8192, 8192), (4096, 3584), (4096, 18944)]:
    x = bench.outlier_activations(M, KQ, dev)
    idx = torch.arange(KQ, dtype=torch.int16, device=dev)
    wn = torch.ones(KQ, dtype=torch.bfloat16, device=dev)
    K = KQ + 64
    byts = M * KQ * 2 + M * K * 9 / 16
    res = {}
    res["x_graph"] = graph_time([lambda: agemm.reorder_quantize_x(x, idx, 64)] * 8)
    res["x_eager"] = bench.time_events(lambda: agemm.reorder_quantize_x(x, idx, 64), 50, 10)
    if 2048 <= KQ <= 8192:
        res["rms_graph"] = graph_time([lambda: agemm.rmsnorm_quantize_x(x, wn, 1e-6, idx, 64)] * 8)
    res["dyn_graph"] = graph_time([lambda: agemm.reorder_quantize_x_dynamic(x, idx, 64)] * 8)
    print(f"M={M} KQ={KQ}: " + "  ".join(f"{k}={v:.2f}us ({byts / v / 1e3:.0f} GB/s)" for k, v in res.items()), flush=True)
