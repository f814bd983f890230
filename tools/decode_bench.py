"""Decode-shape (M <= 16) GEMM timing with weights rotated through > 256 MiB (so they stream from HBM),
replayed from a HIP graph so that host launch overhead does not pace the GPU.
Run under `rocprofv3 --kernel-trace --stats` for pure kernel durations."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcquant_amd import agemm

dev = torch.device("cuda:0")
shapes = [(1, 4096, 4096), (4, 4096, 4096), (16, 4096, 4096), (1, 14336, 4096), (1, 4096, 14336), (1, 1024, 4096),
          (4, 3584, 3584), (4, 18944, 3584), (4, 3584, 18944)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
res = {}
for (M, N, KQ) in shapes:
    KE = 64
    K = KQ + KE
    p = bench.make_problem(M, N, KQ, KE, dev)
    rot = max(2, int(320e6 // (N * K * 9 / 16)) + 1)
    qws = [p["qw"].clone() for _ in range(rot)]
    sfws = [p["sfw"].clone() for _ in range(rot)]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ws_bytes = 0
    def run_all():
        for i in range(rot):
            agemm.matmul(p["qx"], qws[i], p["sfx"], sfws[i], p["alpha"], out=out)
    run_all(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run_all()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            run_all()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * rot)
    gb = bench.gemm_bytes(M, N, K)
    res[f"M{M}_N{N}_KQ{KQ}"] = {"us_per_launch_in_graph": round(us, 3), "GBps": round(gb / us / 1e3, 1), "rot": rot}
    print(f"M={M} N={N} KQ={KQ}: {us:.3f} us/launch (graph replay, {rot} weight copies) -> {gb/us/1e3:.0f} GB/s", flush=True)
    del qws, sfws, g
print(json.dumps(res))
