#!/usr/bin/env python3
"""Where a launch of gemm_regtile.hip spends its time: DIAGNOSTIC library only (make -C arcquant_amd/csrc diag); per wave s_memrealtime at
kernel entry, first operands arrived, K loop done, tail done, epilogue done.  Medians over waves, in us, and the launch's span.
usage: ARCQ_HIP_LIB=$PWD/arcquant_amd/lib/libarcq_hip_diag.so [ARCQ_REGTILE_CFG=n] python tools/regtile_stamps.py [M N KQ KE]..."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import _lib, agemm  # noqa: E402
from bench import make_problem  # noqa: E402

dev = torch.device("cuda:0")
setter = ctypes.CDLL(_lib.LIB_PATH).arcq_debug_set_regtile_stamps
setter.argtypes = [ctypes.c_void_p]
args = [int(x) for x in sys.argv[1:]] or [128, 4096, 4096, 0, 256, 4096, 4096, 0, 512, 4096, 4096, 0, 64, 3584, 18944, 64]
for i in range(0, len(args), 4):
    M, N, KQ, KE = args[i:i + 4]
    q = make_problem(M, N, KQ, KE, dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    stamps = torch.zeros((4096 * 16 * 8,), dtype=torch.int64, device=dev)
    setter(None)
    for _ in range(200):
        agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"], out=out)
    torch.cuda.synchronize()
    setter(stamps.data_ptr())
    agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"], out=out)
    torch.cuda.synchronize()
    setter(None)
    t = stamps.cpu().numpy().reshape(-1, 8)[:, :5].astype(np.float64)
    t = t[t[:, 0] > 0]
    if len(t) == 0:
        print(json.dumps({"shape": [M, N, KQ, KE], "note": "no stamps: this shape does not run gemm_regtile"}))
        continue
    d = np.diff(t, axis=1) / 100.0
    names = ["entry->first operands", "K loop", "tail atoms", "LDS reduce + epilogue"]
    rec = {"shape": [M, N, KQ, KE], "waves": int(len(t)), "span_us": round(float(t[:, 4].max() - t[:, 0].min()) / 100.0, 2),
           "entry_skew_us_p90": round(float(np.percentile(t[:, 0] - t[:, 0].min(), 90)) / 100.0, 2),
           "wave_total_us_median": round(float(np.median(t[:, 4] - t[:, 0])) / 100.0, 2)}
    for j, n in enumerate(names):
        rec[n] = {"median": round(float(np.median(d[:, j])), 2), "p90": round(float(np.percentile(d[:, j], 90)), 2)}
    print(json.dumps(rec), flush=True)
