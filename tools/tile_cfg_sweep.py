#!/usr/bin/env python3
"""Which tile wins where: the tile GEMM at N = K = 4096 (KE = 64) and the Qwen prefill widths for M = 128 ... 4096 under every forced
tile configuration (ARCQ_TILE_CFG: 0 heuristic, 1 = 128x128, 3 = 256x256 / 8 waves, 4 = 128x256, 7 = 64x256), each arm in its own
process (the switch is read once), sustained launches.  usage: python tools/tile_cfg_sweep.py   (parent makes no GPU call)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(m, 4096, 4096) for m in (128, 256, 512, 1024, 2048, 4096)] + [(m, 3584, 3584) for m in (1024, 2048, 4096)] + [(2048, 10752, 3584), (1024, 37888, 3584)]
if os.environ.get("SWEEP_SHAPES") == "big":
    SHAPES = [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 3584, 18944), (4096, 10752, 3584)]


def child():
    import torch
    sys.path.insert(0, ROOT)
    from arcquant_amd import agemm
    from bench import make_problem, time_events_steady, gemm_flops
    dev = torch.device("cuda:0")
    for (m, n, kq) in SHAPES:
        q = make_problem(m, n, kq, 64, dev)
        try:
            t = time_events_steady(lambda: agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"]), 50, 30.0)
            print(json.dumps({"shape": [m, n, kq], "us": round(t, 2), "TFLOPs": round(gemm_flops(m, n, kq + 64) / t / 1e6, 1)}), flush=True)
        except Exception as e:
            print(json.dumps({"shape": [m, n, kq], "error": repr(e)[:100]}), flush=True)
        del q
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    res = {}
    for cfg in os.environ.get("SWEEP_CFGS", "0,1,3,4,7").split(","):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, ARCQ_TILE_CFG=cfg), capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                res.setdefault(tuple(d["shape"]), {})[cfg] = d.get("us", d.get("error"))
    for shape, v in res.items():
        print(json.dumps({"shape": list(shape), **{f"cfg{k}_us": x for k, x in v.items()}}), flush=True)
