#!/usr/bin/env python3
"""Host time per call of the agemm wrappers (Python + ctypes + allocations), GPU left to run asynchronously: what paces an EAGER
decode step (benchmarks/benchmark_e2e_arc.py runs eagerly).  usage: python tools/host_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import agemm  # noqa: E402
from bench import make_problem  # noqa: E402

dev = torch.device("cuda:0")
M, N, KQ, KE = 4, 3584, 3584, 64
q = make_problem(M, N, KQ, KE, dev)
RW, RSF = agemm.repack_w(q["qw"], q["sfw"])
wn = torch.ones(KQ, dtype=torch.bfloat16, device=dev)
res = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
calls = {
    "rmsnorm_matmul_repacked": lambda: agemm.rmsnorm_matmul_repacked(q["x"], wn, 1e-6, q["idx"], KE, RW, RSF, 1.0, N),
    "dynamic_matmul_repacked(residual)": lambda: agemm.dynamic_matmul_repacked(q["x"], q["idx"], KE, RW, RSF, 1.0, N, residual=res),
    "matmul_repacked": lambda: agemm.matmul_repacked(q["qx"], RW, q["sfx"], RSF, q["alpha"], N),
    "reorder_quantize_x_dynamic": lambda: agemm.reorder_quantize_x_dynamic(q["x"], q["idx"], KE),
    "matmul (reference layout)": lambda: agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"]),
    "torch.empty((4, 3584))": lambda: torch.empty((M, N), dtype=torch.bfloat16, device=dev),
}
try:                                                           # the pybind11 extension form of the same boundary (csrc/agemm_ext.cpp)
    from arcquant_amd import _build_ext
    ext = _build_ext.import_agemm_extension()
    xs = (q["x"] / q["sx"]).contiguous()
    calls["EXT matmul (reference layout)"] = lambda: ext.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"])
    calls["EXT reorder_quantize_x"] = lambda: ext.reorder_quantize_x(xs, q["idx"], KE)
    calls["ctypes reorder_quantize_x"] = lambda: agemm.reorder_quantize_x(xs, q["idx"], KE)
    calls["EXT rmsnorm_quantize_x"] = lambda: ext.rmsnorm_quantize_x(q["x"], wn, 1e-6, q["idx"], KE)
    calls["ctypes rmsnorm_quantize_x"] = lambda: agemm.rmsnorm_quantize_x(q["x"], wn, 1e-6, q["idx"], KE)
except ImportError as e:
    print("extension not built:", e)
for name, f in calls.items():
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:40s} host {1e6 * (t1 - t0) / n:6.2f} us per call   (with the GPU drained: {1e6 * (t2 - t0) / n:6.2f} us)", flush=True)
