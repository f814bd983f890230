#!/usr/bin/env python3
"""In-kernel clock of the tile GEMM's K loop (MI355X_MICROARCH.md "DVFS give-back" item 6): DIAGNOSTIC library only
(make -C arcquant_amd/csrc diag), >= 2 s of back-to-back launches on random data, then delta s_memtime / delta s_memrealtime
x 100 MHz per workgroup, median over workgroups.  The fp16 library GEMM cannot be stamped; its clock is read from
GRBM_GUI_ACTIVE in a separate rocprofv3 pass (profiles/).
usage: ARCQ_HIP_LIB=$PWD/arcquant_amd/lib/libarcq_hip_diag.so python tools/tile_clock.py"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import _lib, agemm  # noqa: E402
from bench import make_problem  # noqa: E402

dev = torch.device("cuda:0")
setter = ctypes.CDLL(_lib.LIB_PATH).arcq_debug_set_tile_stamps
setter.argtypes = [ctypes.c_void_p]
for S in (4096, 8192):
    q = make_problem(S, S, S, 64, dev)
    out = torch.empty((S, S), dtype=torch.bfloat16, device=dev)
    stamps = torch.zeros((4096 * 4,), dtype=torch.int64, device=dev)
    setter(None)
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.5:                         # >= 2 s of continuous load before the stamped launches
        for _ in range(200):
            agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"], out=out)
        n += 200
        torch.cuda.synchronize()
    setter(stamps.data_ptr())
    for _ in range(50):
        agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"], out=out)
    torch.cuda.synchronize()
    setter(None)
    t = stamps.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 0] > 0]
    cyc, rt = (t[:, 2] - t[:, 0]).astype(np.float64), (t[:, 3] - t[:, 1]).astype(np.float64)
    ghz = cyc / rt * 0.1
    print(json.dumps({"gemm": f"{S}x{S}x{S + 64}", "workgroups": int(len(t)), "k_loop_us_median": round(float(np.median(rt)) / 100.0, 2),
                      "in_kernel_clock_GHz_median": round(float(np.median(ghz)), 3), "p10": round(float(np.percentile(ghz, 10)), 3),
                      "p90": round(float(np.percentile(ghz, 90)), 3), "launches_before_stamp": n}), flush=True)
