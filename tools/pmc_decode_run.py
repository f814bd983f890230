#!/usr/bin/env python3
"""Workload for the decode-GEMM PMC passes (tools/scripts/pmc_r02_decode.sh): the Qwen2.5-7B gate|up decode shape (M=4, N=37888,
KQ=3584, KE=64) on the reference-layout kernel, the repacked rowblock kernel and the fused RMSNorm + gate|up + SiLU kernel, weights
rotated through > 320 MB, 6 rounds each, plain stream launches (counters are per dispatch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import agemm  # noqa: E402
from bench import make_problem  # noqa: E402

dev = torch.device("cuda:0")
M, N, KQ, KE = 4, 37888, 3584, 64
q = make_problem(M, N, KQ, KE, dev)
K = KQ + KE
rot = int(320e6 // (N * K * 9 / 16)) + 1
qws = [(q["qw"].clone(), q["sfw"].clone()) for _ in range(rot)]
rps = [agemm.repack_w(*w) for w in qws]
wn = torch.ones(KQ, dtype=torch.bfloat16, device=dev)
o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
for _ in range(6):
    for i in range(rot):
        agemm.matmul(q["qx"], qws[i][0], q["sfx"], qws[i][1], q["alpha"], out=o)
    for i in range(rot):
        agemm.matmul_repacked(q["qx"], rps[i][0], q["sfx"], rps[i][1], q["alpha"], N, out=o)
    for i in range(rot):
        agemm.rmsnorm_matmul_repacked_silu(q["x"], wn, 1e-6, q["idx"], KE, rps[i][0], rps[i][1], 1.0, N)
torch.cuda.synchronize()
print("done")
