#!/usr/bin/env python3
"""Same-box A-B of library builds on the decode harness: tok/s of the Qwen2.5-7B-shape decode step (full-cache and current-token
attention) per build, rounds interleaved, each run in its own process.  usage: e2e_lib_ab.py name=path.so [name=path.so ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(path, att):
    env = dict(os.environ, ARCQ_HIP_LIB=os.path.join(ROOT, path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_profile.py"), "28", att], env=env, capture_output=True, text=True, cwd=ROOT)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)["decode_tok_per_s"]
    raise RuntimeError(r.stderr[-800:])


if __name__ == "__main__":
    specs = dict(a.split("=", 1) for a in sys.argv[1:])
    acc = {n: {"cache": [], "current": []} for n in specs}
    for rnd in range(2):
        for n, p in specs.items():
            for att in ("cache", "current"):
                acc[n][att].append(run(p, att))
    for n in specs:
        print(json.dumps({"build": n, **acc[n]}), flush=True)
