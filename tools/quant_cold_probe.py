#!/usr/bin/env python3
"""RMSNorm / static quantiser at the prefill size with a cache-resident input (the same tensor every launch) against inputs
rotated through > 400 MB (HBM-cold), replayed from one HIP graph: why the e2e profile shows ~32 us where the bench shows ~18."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from arcquant_amd import agemm  # noqa: E402
from tools.decode_stream_bench import graph_time  # noqa: E402

dev = torch.device("cuda:0")
M, KQ = 4096, 3584
idx = torch.arange(KQ, dtype=torch.int16, device=dev)
wn = torch.ones(KQ, dtype=torch.bfloat16, device=dev)
xs = [bench.outlier_activations(M, KQ, dev, seed=i) for i in range(16)]          # 16 x 29 MB = 470 MB
hot_r = graph_time([lambda: agemm.rmsnorm_quantize_x(xs[0], wn, 1e-6, idx, 64)] * 16)
cold_r = graph_time([(lambda i=i: agemm.rmsnorm_quantize_x(xs[i], wn, 1e-6, idx, 64)) for i in range(16)])
hot_s = graph_time([lambda: agemm.reorder_quantize_x(xs[0], idx, 64)] * 16)
cold_s = graph_time([(lambda i=i: agemm.reorder_quantize_x(xs[i], idx, 64)) for i in range(16)])
print(f"M={M} KQ={KQ}  rmsnorm quantiser: same input {hot_r:.2f} us, rotating inputs {cold_r:.2f} us;  static quantiser: {hot_s:.2f} / {cold_s:.2f} us")
