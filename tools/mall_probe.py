#!/usr/bin/env python3
"""Does a decode linear run faster when its weight already sits in the Infinity Cache (MALL)?  Times every decode kernel form
(plain repacked, stream packed, fused RMSNorm / dynamic prologue) as HIP-graph replays
   cold : weight copies rotated through > 320 MB (what bench.py reports: every byte comes from HBM)
   hot  : ONE weight copy replayed back to back (<= 80 MB: resident in the 256 MB Infinity Cache after the first replay)
The hot/cold ratio bounds what a cross-kernel weight prefetch (the previous kernel touching the next kernel's weight lines while
its own prologue leaves HBM idle) could gain.  usage: python tools/mall_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import agemm  # noqa: E402
from bench import make_problem  # noqa: E402
from tools.decode_stream_bench import graph_time  # noqa: E402

dev = torch.device("cuda:0")


def main():
    KE = 64
    for (m, n, kq) in [(1, 4096, 4096), (4, 3584, 3584), (4, 10752, 3584), (4, 37888, 3584), (4, 3584, 18944)]:
        q = make_problem(m, n, kq, KE, dev)
        K = kq + KE
        rot = max(2, int(320e6 // (n * K * 9 / 16)) + 1)
        o = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        rps = [agemm.repack_w(q["qw"].clone(), q["sfw"].clone()) for _ in range(rot)]
        x, sw = q["x"], float(q["sw"])
        wn = torch.ones(kq, dtype=torch.bfloat16, device=dev)
        forms = {
            "rowblock": lambda i: agemm.matmul_repacked(q["qx"], rps[i][0], q["sfx"], rps[i][1], q["alpha"], n, out=o),
            "stream": lambda i: agemm.matmul_repacked(q["qx"], rps[i][0], q["sfx"], rps[i][1], q["alpha"], n, out=o, kernel="stream"),
        }
        if agemm.fused_supported(agemm.SRC_DYNAMIC, m, n, kq, KE):
            forms["fused_dyn"] = lambda i: agemm.dynamic_matmul_repacked(x, q["idx"], KE, rps[i][0], rps[i][1], sw, n, out=o)
        if 2048 <= kq <= 8192 and agemm.fused_supported(agemm.SRC_RMSNORM, m, n, kq, KE):
            forms["fused_rms"] = lambda i: agemm.rmsnorm_matmul_repacked(x, wn, 1e-6, q["idx"], KE, rps[i][0], rps[i][1], sw, n, out=o)
        rec = {"shape": [m, n, kq], "weight_MB": round(n * K * 9 / 16 / 1e6, 1)}
        for name, f in forms.items():
            cold = graph_time([(lambda i=i: f(i)) for i in range(rot)])
            hot = graph_time([(lambda: f(0))] * 8)
            rec[name] = {"cold_us": round(cold, 2), "hot_us": round(hot, 2), "hot_over_cold": round(hot / cold, 3)}
        print(json.dumps(rec), flush=True)
        del rps, q
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
