"""ARC-GEMM time against the token count M (tuning aid): python tools/m_sweep.py [N KQ]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcquant_amd import agemm


def main():
    shapes = [(4096, 4096), (37888, 3584), (3584, 18944)]
    if len(sys.argv) == 3:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    ms = [int(v) for v in os.environ.get("SWEEP_M", "1,4,8,16,17,32,48,64,96,128,192,256,512,1024").split(",")]
    dev = torch.device("cuda:0")
    for N, KQ in shapes:
        for M in ms:
            p = bench.make_problem(M, N, KQ, 64, dev)
            f = lambda: agemm.matmul(p["qx"], p["qw"], p["sfx"], p["sfw"], p["alpha"])
            us = bench.time_graph(f, 20, 5) if hasattr(bench, "time_graph") else bench.time_events(f, 50, 10)
            K = KQ + 64
            byts = (M + N) * (K // 2 + K // 16) + M * N * 2
            print(f"N={N} KQ={KQ} M={M:5d}  {us:8.2f} us  {bench.gemm_flops(M, N, K) / us / 1e6:8.1f} TFLOP/s  {byts / us / 1e3:8.1f} GB/s", flush=True)
            del p


if __name__ == "__main__":
    main()
