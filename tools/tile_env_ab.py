#!/usr/bin/env python3
"""A-B of a tile-GEMM tuning switch (an environment variable read once per process by csrc/gemm_tile.hip, e.g. ARCQ_TILE_224):
steady-state time per launch of agemm.matmul / matmul_silu_mul with the switch at 0 and at 1, rounds interleaved, one subprocess
each.  usage: tile_env_ab.py ENVNAME [M,N,KQ[,silu] ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.tile_persist_ab import CODE  # noqa: E402

DEFAULT = [(4096, 3584, 3584, 0), (4096, 10752, 3584, 0), (4096, 3584, 18944, 0), (4096, 7168, 8192, 0), (4096, 14336, 4096, 0), (4096, 7168, 3584, 1),
           (1024, 3584, 3584, 0)]


def run(env_name, val, shapes):
    env = dict(os.environ)
    env[env_name] = str(val)
    r = subprocess.run([sys.executable, "-c", CODE.format(root=ROOT, shapes=shapes)], env=env, capture_output=True, text=True, cwd=ROOT)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    raise RuntimeError(r.stderr[-800:])


if __name__ == "__main__":
    name = sys.argv[1]
    shapes = [tuple(int(v) for v in a.split(",")) + ((0,) if a.count(",") == 2 else ()) for a in sys.argv[2:]] or DEFAULT
    acc = {0: [], 1: []}
    for rnd in range(2):
        for v in (0, 1):
            acc[v].append(run(name, v, shapes))
    for i, shp in enumerate(shapes):
        off = [r[i]["us"] for r in acc[0]]
        on = [r[i]["us"] for r in acc[1]]
        print(json.dumps({"shape": shp, name + "=0_us": off, name + "=1_us": on, "gain": round(min(off) / min(on), 4)}), flush=True)
