#!/usr/bin/env python3
"""Decode-batch GEMMs (M = 8 ... 128) on both weight layouts, HBM-cold (weights rotated through > 320 MB), HIP-graph replay:
the reference-layout kernels (M <= 16: skinny / decode; above: split-K tile GEMM + finish pass) against the repacked
weight-streaming kernels (M <= 16: gemm_rowblock; 16 < M <= 64: gemm_rowmid).  usage: [MIDM_NK=NxKQxKE,...] [MIDM_MS=1,4,16] [MIDM_REPACKED=0] python tools/midm_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import agemm  # noqa: E402
from bench import make_problem, gemm_bytes  # noqa: E402
from tools.decode_stream_bench import graph_time  # noqa: E402

dev = torch.device("cuda:0")

SHAPES = [(4096, 4096, 0), (4096, 4096, 64), (3584, 3584, 64), (10752, 3584, 64), (37888, 3584, 64)]
if os.environ.get("MIDM_NK"):          # "4096x4096x64,14336x4096x64"
    SHAPES = [tuple(int(y) for y in x.split("x")) for x in os.environ["MIDM_NK"].split(",")]
MS = [int(x) for x in os.environ.get("MIDM_MS", "8,16,17,32,48,64,128").split(",")]
REPACKED = os.environ.get("MIDM_REPACKED", "1") == "1"
for (n, kq, ke) in SHAPES:
    K = kq + ke
    rot = max(2, int(320e6 // (n * K * 9 / 16)) + 1)
    for m in MS:
        q = make_problem(m, n, kq, ke, dev)
        o = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        qws, sfws = [q["qw"].clone() for _ in range(rot)], [q["sfw"].clone() for _ in range(rot)]
        rec = {"N": n, "KQ": kq, "KE": ke, "M": m}
        rec["reference_layout_us"] = round(graph_time([(lambda i=i: agemm.matmul(q["qx"], qws[i], q["sfx"], sfws[i], q["alpha"], out=o)) for i in range(rot)]), 2)
        if REPACKED and agemm.repacked_supported(m, n, K):
            rps = [agemm.repack_w(qws[i], sfws[i]) for i in range(rot)]
            rec["repacked_us"] = round(graph_time([(lambda i=i: agemm.matmul_repacked(q["qx"], rps[i][0], q["sfx"], rps[i][1], q["alpha"], n, out=o)) for i in range(rot)]), 2)
            del rps
        best = min(v for k, v in rec.items() if k.endswith("_us"))
        rec["GBps"] = round(gemm_bytes(m, n, K) / best / 1e3, 1)
        print(json.dumps(rec), flush=True)
        del qws, sfws, q
        torch.cuda.empty_cache()
