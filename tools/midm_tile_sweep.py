#!/usr/bin/env python3
"""The tile GEMM between decode and prefill (M = 128 ... 2048): every tile configuration x forced split-K factor, each arm in its own
process (the switches are read once), sustained launches replayed from a HIP graph (SWEEP_GRAPH=0: eager, host-bound below ~9.5 us).  usage: python tools/midm_tile_sweep.py   (parent makes no GPU call)
  SWEEP_ARMS="0:0,1:0,10:0,12:2,r1:0"  (ARCQ_TILE_CFG:ARCQ_TILE_SPLIT; split 0 = the launcher's own choice; rN = ARCQ_REGTILE_CFG=N, gemm_regtile.hip)   SWEEP_MS, SWEEP_NK"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MS = [int(x) for x in os.environ.get("SWEEP_MS", "128,256,512,1024,2048").split(",")]
NK = [tuple(int(y) for y in x.split("x")) for x in os.environ.get("SWEEP_NK", "4096x4096,3584x3584,10752x3584,3584x18944").split(",")]


def child():
    import torch
    sys.path.insert(0, ROOT)
    from arcquant_amd import agemm
    from bench import make_problem, time_events_steady
    dev = torch.device("cuda:0")
    bufs = {}

    def out_buf(m, n):
        if (m, n) not in bufs:
            bufs.clear()
            bufs[(m, n)] = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        return bufs[(m, n)]

    def graph_us(fn, per_graph=20):
        """us per launch, `per_graph` launches replayed from one HIP graph (an eager launch through the ctypes mirror costs ~9.5 us of
        host time: below that an eager loop measures the host)"""
        fn()
        torch.cuda.synchronize()
        g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st):
                for _ in range(per_graph):
                    fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(30):
            g.replay()
        torch.cuda.synchronize()
        e0.record()
        reps = 40
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * per_graph)

    for (n, kq) in NK:
        for m in MS:
            q = make_problem(m, n, kq, 64, dev)
            try:
                if os.environ.get("SWEEP_GRAPH", "1") == "1":
                    t = graph_us(lambda: agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"], out=out_buf(m, n)))
                else:
                    t = time_events_steady(lambda: agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"]), 50, 20.0)
                print(json.dumps({"shape": [m, n, kq], "us": round(t, 2)}), flush=True)
            except Exception as e:
                print(json.dumps({"shape": [m, n, kq], "error": repr(e)[:100]}), flush=True)
            del q
            torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    res = {}
    arms = os.environ.get("SWEEP_ARMS", "0:0,1:0,10:0,11:0,12:0,13:0,14:0,15:0,16:0,17:0,12:2,14:2,15:2,16:2,17:2,15:4,14:4").split(",")
    for arm in arms:
        cfg, split = arm.split(":")
        env = dict(os.environ, ARCQ_REGTILE_CFG=cfg[1:]) if cfg.startswith("r") else dict(os.environ, ARCQ_TILE_CFG=cfg, ARCQ_TILE_SPLIT=split, ARCQ_REGTILE_CFG="-1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env,
                           capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                res.setdefault(tuple(d["shape"]), {})[arm] = d.get("us", d.get("error"))
        if r.returncode != 0:
            print(json.dumps({"arm": arm, "rc": r.returncode, "stderr": r.stderr[-300:]}), flush=True)
    for shape, v in res.items():
        best = min((x, k) for k, x in v.items() if isinstance(x, float))
        print(json.dumps({"shape": list(shape), "best": best[1], **v}), flush=True)
