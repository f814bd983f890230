"""Decode GEMM: reference-layout kernels against the repacked-weight kernel, weights rotated through > 320 MB (HBM-cold),
HIP-graph replay (tuning aid).  python tools/repacked_bench.py [M,N,KQ ...]"""
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


shapes = [(4, 37888, 3584), (4, 10752, 3584), (4, 3584, 3584), (1, 4096, 4096), (1, 14336, 4096), (1, 4096, 14336), (1, 1024, 4096),
          (4, 4096, 4096), (8, 37888, 3584), (16, 14336, 1984)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (M, N, KQ) in shapes:
    K = KQ + 64
    p = bench.make_problem(M, N, KQ, 64, dev)
    rot = max(2, int(320e6 // (N * K * 9 / 16)) + 1)
    qws = [p["qw"].clone() for _ in range(rot)]
    sfws = [p["sfw"].clone() for _ in range(rot)]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    t_ref = graph_time([(lambda i=i: agemm.matmul(p["qx"], qws[i], p["sfx"], sfws[i], p["alpha"], out=out)) for i in range(rot)])
    line = f"M={M} N={N} KQ={KQ}: reference layout {t_ref:.2f} us"
    if agemm.repacked_supported(M, N, K):
        rp = [agemm.repack_w(qws[i], sfws[i]) for i in range(rot)]
        t_rp = graph_time([(lambda i=i: agemm.matmul_repacked(p["qx"], rp[i][0], p["sfx"], rp[i][1], p["alpha"], N, out=out)) for i in range(rot)])
        gb = bench.gemm_bytes(M, N, K)
        line += f", repacked {t_rp:.2f} us ({gb / t_rp / 1e3:.0f} GB/s)"
        del rp
    print(line, flush=True)
    del qws, sfws, p
