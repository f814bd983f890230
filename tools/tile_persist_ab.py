#!/usr/bin/env python3
"""A-B of the persistent tile schedule (ARCQ_TILE_PERSIST=0|1 -- a tuning switch of csrc/gemm_tile.hip in commit eb436bb only; the
schedule was removed again, DESIGN.md 3.2, so on the current source both arms run the same kernel): steady-state time per launch of
agemm.matmul (and of the SiLU*up epilogue variant) on shapes with more than 256 tiles of 256 x 256.  One subprocess per setting
(the switch is read once per process).  usage: tile_persist_ab.py [M N KQ [silu]] ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(4096, 4096, 4096, 0), (8192, 8192, 8192, 0), (4096, 10752, 3584, 0), (4096, 3584, 3584, 0), (4096, 3584, 18944, 0), (4096, 37888, 3584, 1),
          (8192, 4096, 4096, 0)]

CODE = """
import json, sys, torch
sys.path.insert(0, {root!r})
import bench
from arcquant_amd import agemm
dev = torch.device('cuda:0')
out = []
for (M, N, KQ, silu) in {shapes!r}:
    p = bench.make_problem(M, N, KQ, 64, dev)
    if silu:
        f = lambda: agemm.matmul_silu_mul(p['qx'], p['qw'], p['sfx'], p['sfw'], p['alpha'])
    else:
        f = lambda: agemm.matmul(p['qx'], p['qw'], p['sfx'], p['sfw'], p['alpha'])
    us = bench.time_events_steady(f, 100, warm_ms=60.0)
    out.append({{"shape": [M, N, KQ], "silu": silu, "us": round(us, 2), "TFLOPs": round(bench.gemm_flops(M, N, KQ + 64) / us / 1e6, 1)}})
    del p
    torch.cuda.empty_cache()
print("RESULT " + json.dumps(out))
"""


def run(persist, shapes):
    env = dict(os.environ, ARCQ_TILE_PERSIST=str(persist))
    r = subprocess.run([sys.executable, "-c", CODE.format(root=ROOT, shapes=shapes)], env=env, capture_output=True, text=True, cwd=ROOT)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    raise RuntimeError(r.stderr[-800:])


if __name__ == "__main__":
    shapes = SHAPES
    a, b = run(0, shapes), run(1, shapes)
    for x, y in zip(a, b):
        print(json.dumps({"shape": x["shape"], "silu": x["silu"], "one_tile_per_workgroup_us": x["us"], "persistent_us": y["us"],
                          "one_tile_TFLOPs": x["TFLOPs"], "persistent_TFLOPs": y["TFLOPs"], "gain": round(x["us"] / y["us"], 4)}), flush=True)
