#!/usr/bin/env python3
"""Decode batches 16 < M <= 128: LDS-resident packed activations (gemm_rowmid, M <= 64) / tiled GEMM (above) against the no-LDS kernel
that fetches its activation operands per tile pair (gemm_rowtok, ARCQ_ROWTOK=1).  Each arm in its own process, HBM-cold, graph replay.
usage: python tools/rowtok_ab.py [lib.so ...]   (parent makes no GPU call; extra libraries are timed as further arms with ARCQ_ROWTOK=0)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(m, 4096, 4096, 0) for m in (17, 32, 48, 64, 96, 128)] + [(m, 3584, 3584, 64) for m in (32, 64, 128)] + \
         [(m, 10752, 3584, 64) for m in (32, 64, 128)] + [(m, 37888, 3584, 64) for m in (32, 64)] + [(32, 3584, 18944, 64)]


def child():
    import torch
    sys.path.insert(0, ROOT)
    from arcquant_amd import agemm
    from bench import make_problem
    from tools.decode_stream_bench import graph_time
    dev = torch.device("cuda:0")
    for (m, n, kq, ke) in SHAPES:
        q = make_problem(m, n, kq, ke, dev)
        K = kq + ke
        rot = max(2, int(320e6 // (n * K * 9 / 16)) + 1)
        o = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        qws, sfws = [q["qw"].clone() for _ in range(rot)], [q["sfw"].clone() for _ in range(rot)]
        if agemm.repacked_supported(m, n, K):
            rps = [agemm.repack_w(qws[i], sfws[i]) for i in range(rot)]
            t = graph_time([(lambda i=i: agemm.matmul_repacked(q["qx"], rps[i][0], q["sfx"], rps[i][1], q["alpha"], n, out=o)) for i in range(rot)])
            kind = "repacked"
            del rps
        else:
            t = graph_time([(lambda i=i: agemm.matmul(q["qx"], qws[i], q["sfx"], sfws[i], q["alpha"], out=o)) for i in range(rot)])
            kind = "tile"
        print(json.dumps({"shape": [m, n, kq, ke], "us": round(t, 2), "kind": kind}), flush=True)
        del qws, sfws, q
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    arms = {"lds_or_tile": {"ARCQ_ROWTOK": "0"}, "rowtok": {"ARCQ_ROWTOK": "1"}}
    for lib in sys.argv[1:]:
        arms[os.path.basename(lib)] = {"ARCQ_ROWTOK": "0", "ARCQ_HIP_LIB": os.path.join(ROOT, lib)}
    res = {}
    for rnd in range(2):
        for name, env in arms.items():
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True)
            for line in r.stdout.splitlines():
                if line.startswith("{"):
                    d = json.loads(line)
                    res.setdefault(tuple(d["shape"]), {}).setdefault(name, []).append(d["us"])
                    res[tuple(d["shape"])][name + "_kind"] = d["kind"]
            if r.returncode:
                print(r.stderr[-1500:], file=sys.stderr)
    for shape, v in res.items():
        print(json.dumps({"shape": list(shape), **v}), flush=True)
