"""Decode GEMM with MALL-warm weights (the same weight every launch) against HBM-cold (rotating copies): what a
weight prefetch on a side stream could buy (tuning aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcquant_amd import agemm

dev = torch.device("cuda:0")


def graph_time(launches, reps=10):
    for f in launches: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for f in launches: f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for f in launches: f()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(launches))


for (M, N, KQ) in [(4, 37888, 3584), (4, 3584, 18944), (4, 10752, 3584), (4, 3584, 3584), (1, 4096, 4096)]:
    p = bench.make_problem(M, N, KQ, 64, dev)
    K = KQ + 64
    rot = max(2, int(320e6 // (N * K * 9 / 16)) + 1)
    qws = [p["qw"].clone() for _ in range(rot)]
    sfws = [p["sfw"].clone() for _ in range(rot)]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    cold = graph_time([(lambda i=i: agemm.matmul(p["qx"], qws[i], p["sfx"], sfws[i], p["alpha"], out=out)) for i in range(rot)])
    warm = graph_time([(lambda: agemm.matmul(p["qx"], qws[0], p["sfx"], sfws[0], p["alpha"], out=out))] * 8)
    print(f"M={M} N={N} KQ={KQ}: cold {cold:.2f} us, warm {warm:.2f} us ({N * K * 9 / 16 / 1e6:.1f} MB)", flush=True)
    del qws, sfws, p
