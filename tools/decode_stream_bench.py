#!/usr/bin/env python3
"""Decode GEMM / fused decode linear timings, HIP-graph replay, weights rotated through > 320 MB (HBM-cold), with the
equal-shape fp16 torch.matmul (hipBLASLt / rocBLAS) beside each.  usage: python tools/decode_stream_bench.py [quick]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import agemm  # noqa: E402
from bench import make_problem, gemm_bytes  # noqa: E402

dev = torch.device("cuda:0")


def graph_time(launches, warm_ms=30.0, min_ms=10.0):
    for f in launches:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for f in launches:
            f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for f in launches:
                f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    per = max(e0.elapsed_time(e1) / 3, 1e-3)
    for _ in range(min(20000, int(warm_ms / per))):
        g.replay()
    reps = max(10, min(20000, int(min_ms / per)))
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(launches))


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    KE = 64
    shapes = [(1, 4096, 4096), (4, 4096, 4096), (1, 14336, 4096), (1, 4096, 14336), (1, 1024, 4096), (4, 3584, 3584), (4, 10752, 3584),
              (4, 37888, 3584), (4, 3584, 18944), (16, 4096, 4096)]
    if quick:
        shapes = [(1, 4096, 4096), (4, 37888, 3584), (4, 3584, 18944)]
    for (m, n, kq) in shapes:
        q = make_problem(m, n, kq, KE, dev)
        K = kq + KE
        rot = max(2, int(320e6 // (n * K * 9 / 16)) + 1)
        o = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        out = {"shape": [m, n, kq]}
        qws = [q["qw"].clone() for _ in range(rot)]
        sfws = [q["sfw"].clone() for _ in range(rot)]
        out["ref_layout_us"] = round(graph_time([(lambda i=i: agemm.matmul(q["qx"], qws[i], q["sfx"], sfws[i], q["alpha"], out=o)) for i in range(rot)]), 2)
        del qws, sfws
        if agemm.repacked_supported(m, n, K):
            rps = [agemm.repack_w(q["qw"].clone(), q["sfw"].clone()) for _ in range(rot)]
            out["repacked_us"] = round(graph_time([(lambda i=i: agemm.matmul_repacked(q["qx"], rps[i][0], q["sfx"], rps[i][1], q["alpha"], n, out=o)) for i in range(rot)]), 2)
            x = q["x"]
            sw = float(q["sw"])
            if agemm.fused_supported(agemm.SRC_DYNAMIC, m, n, kq, KE):
                out["fused_dynamic_us"] = round(graph_time([(lambda i=i: agemm.dynamic_matmul_repacked(x, q["idx"], KE, rps[i][0], rps[i][1], sw, n, out=o)) for i in range(rot)]), 2)
                slots = torch.full((max(1, kq // 8),), 0x4000, dtype=torch.int32, device=dev)
                out["fused_dynamic_slots_us"] = round(graph_time([(lambda i=i: agemm.dynamic_matmul_repacked(x, q["idx"], KE, rps[i][0], rps[i][1], sw, n, out=o, absmax_slots=slots)) for i in range(rot)]), 2)

                def unfused(i):
                    qa, sfa, sa = agemm.reorder_quantize_x_dynamic(x, q["idx"], KE)
                    agemm.matmul_repacked(qa, rps[i][0], sfa, rps[i][1], sa, n, scale_host=sw, out=o)
                out["unfused_dynamic_pair_us"] = round(graph_time([(lambda i=i: unfused(i)) for i in range(rot)]), 2)
            if 2048 <= kq <= 8192 and agemm.fused_supported(agemm.SRC_RMSNORM, m, n, kq, KE):
                wn = torch.ones(kq, dtype=torch.bfloat16, device=dev)
                out["fused_rmsnorm_us"] = round(graph_time([(lambda i=i: agemm.rmsnorm_matmul_repacked(x, wn, 1e-6, q["idx"], KE, rps[i][0], rps[i][1], sw, n, out=o)) for i in range(rot)]), 2)

                def unfused_r(i):
                    a, sfa = agemm.rmsnorm_quantize_x(x, wn, 1e-6, q["idx"], KE)
                    agemm.matmul_repacked(a, rps[i][0], sfa, rps[i][1], sw, n, out=o)
                out["unfused_rmsnorm_pair_us"] = round(graph_time([(lambda i=i: unfused_r(i)) for i in range(rot)]), 2)
            del rps
        # equal-shape fp16 library GEMM, weights rotated the same way (4x the bytes)
        rot16 = max(2, int(320e6 // (n * K * 2)) + 1)
        a16 = torch.randn(m, K, dtype=torch.float16, device=dev)
        b16 = [torch.randn(n, K, dtype=torch.float16, device=dev) for _ in range(rot16)]
        o16 = torch.empty((m, n), dtype=torch.float16, device=dev)
        out["fp16_rocblas_us"] = round(graph_time([(lambda i=i: torch.matmul(a16, b16[i].t(), out=o16)) for i in range(rot16)]), 2)
        del b16
        gb = gemm_bytes(m, n, K)
        best = min(v for k, v in out.items() if k in ("ref_layout_us", "repacked_us"))
        out["GBps_best"] = round(gb / best / 1e3, 1)
        out["speedup_vs_fp16_rocblas"] = round(out["fp16_rocblas_us"] / best, 2)
        print(json.dumps(out), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
