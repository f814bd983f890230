#!/usr/bin/env python3
"""A-B of the two plain repacked decode kernels (gemm_rowblock.hip): fp16 activation image in LDS (ARCQ_ROWBLOCK_DIRECT=0) against
per-lane activation loads with no LDS image and no barrier before the K loop (=1).  Each arm in its own process (the switch is read
once), HBM-cold weights, HIP-graph replay.  usage: python tools/direct_ab.py   (parent makes no GPU call)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(1, 4096, 4096), (4, 4096, 4096), (16, 4096, 4096), (1, 14336, 4096), (1, 4096, 14336), (1, 1024, 4096), (4, 3584, 3584), (4, 10752, 3584),
          (4, 37888, 3584), (4, 3584, 18944), (8, 3584, 18944), (4, 8192, 3584), (4, 8192, 1024)]


def child():
    import torch
    sys.path.insert(0, ROOT)
    from arcquant_amd import agemm
    from bench import make_problem
    from tools.decode_stream_bench import graph_time
    dev = torch.device("cuda:0")
    for (m, n, kq) in SHAPES:
        q = make_problem(m, n, kq, 64, dev)
        K = kq + 64
        rot = max(2, int(320e6 // (n * K * 9 / 16)) + 1)
        o = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        rps = [agemm.repack_w(q["qw"].clone(), q["sfw"].clone()) for _ in range(rot)]
        t = graph_time([(lambda i=i: agemm.matmul_repacked(q["qx"], rps[i][0], q["sfx"], rps[i][1], q["alpha"], n, out=o)) for i in range(rot)])
        print(json.dumps({"shape": [m, n, kq], "us": round(t, 2)}), flush=True)
        del rps, q
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    res = {}
    for rnd in range(2):
        for arm in ("0", "1"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, ARCQ_ROWBLOCK_DIRECT=arm), capture_output=True, text=True)
            for line in r.stdout.splitlines():
                if line.startswith("{"):
                    d = json.loads(line)
                    res.setdefault(tuple(d["shape"]), {"0": [], "1": []})[arm].append(d["us"])
            if r.returncode:
                print(r.stderr[-2000:], file=sys.stderr)
    for shape, v in res.items():
        print(json.dumps({"shape": list(shape), "image_us": v["0"], "direct_us": v["1"],
                          "direct_over_image": round(min(v["1"]) / min(v["0"]), 3) if v["0"] and v["1"] else None}), flush=True)
