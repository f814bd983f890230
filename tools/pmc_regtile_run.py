#!/usr/bin/env python3
"""Workload for PMC passes over gemm_regtile.hip: `agemm.matmul` on the reference layout at one shape (argv: M N KQ KE), 60 plain stream
launches (counters are per dispatch; pmc_summarize.py averages the last 20).  usage under rocprofv3: -- python3 tools/pmc_regtile_run.py 512 4096 4096 0"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import agemm  # noqa: E402
from bench import make_problem  # noqa: E402

M, N, KQ, KE = (int(x) for x in sys.argv[1:5])
dev = torch.device("cuda:0")
q = make_problem(M, N, KQ, KE, dev)
o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
for _ in range(60):
    agemm.matmul(q["qx"], q["qw"], q["sfx"], q["sfw"], q["alpha"], out=o)
torch.cuda.synchronize()
