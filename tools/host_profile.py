#!/usr/bin/env python3
"""cProfile of one agemm wrapper (where the ~10 us of host time per call go).  usage: python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import agemm  # noqa: E402
from bench import make_problem  # noqa: E402

dev = torch.device("cuda:0")
M, N, KQ, KE = 4, 3584, 3584, 64
q = make_problem(M, N, KQ, KE, dev)
RW, RSF = agemm.repack_w(q["qw"], q["sfw"])
wn = torch.ones(KQ, dtype=torch.bfloat16, device=dev)
f = lambda: agemm.rmsnorm_matmul_repacked(q["x"], wn, 1e-6, q["idx"], KE, RW, RSF, 1.0, N)  # noqa: E731
for _ in range(200):
    f()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5000):
    f()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
