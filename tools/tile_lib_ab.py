#!/usr/bin/env python3
"""A-B of two builds of the library on the tile GEMM (tuning aid): steady-state time per launch, each build in its own subprocess,
rounds interleaved.  usage: tile_lib_ab.py name=path[:ENV=VAL] ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.tile_persist_ab import CODE  # noqa: E402

SHAPES = [(4096, 4096, 4096, 0), (8192, 8192, 8192, 0), (4096, 3584, 18944, 0), (4096, 37888, 3584, 1)]


def run(spec):
    path, *envs = spec.split(":")
    env = dict(os.environ, ARCQ_HIP_LIB=os.path.join(ROOT, path))
    for e in envs:
        k, v = e.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CODE.format(root=ROOT, shapes=SHAPES)], env=env, capture_output=True, text=True, cwd=ROOT)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    raise RuntimeError(r.stderr[-800:])


if __name__ == "__main__":
    specs = dict(a.split("=", 1) for a in sys.argv[1:])
    acc = {n: [] for n in specs}
    for rnd in range(2):
        for n, s in specs.items():
            acc[n].append(run(s))
    for i, shp in enumerate(SHAPES):
        print(json.dumps({"shape": shp, **{n: [r[i]["us"] for r in acc[n]] for n in specs}}), flush=True)
