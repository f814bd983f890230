#!/usr/bin/env python3
"""Phase shares of the decode stream kernel from in-kernel s_memtime stamps (DIAGNOSTIC library libarcq_hip_diag.so, built by
`make -C arcquant_amd/csrc diag`; the product library has no stamps).  Prints, per shape, the median over waves of each phase
in shader cycles (s_memtime ticks at 100 MHz * ...: reported as microseconds via s_memrealtime-free assumption of 100 MHz
reference -> we print raw ticks and the share of the total).
usage: ARCQ_HIP_LIB=arcquant_amd/lib/libarcq_hip_diag.so python tools/stream_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import _lib, agemm  # noqa: E402
from bench import make_problem  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
setter = ctypes.CDLL(_lib.LIB_PATH).arcq_debug_set_stream_stamps
setter.argtypes = [ctypes.c_void_p]
PH = ["args->ring issued", "ring->image built", "image->barrier", "K loop", "loop end->slots written+2 barriers", "reduce+epilogue"]


def run(m, n, kq, mode):
    KE = 64
    q = make_problem(m, n, kq, KE, dev)
    K = kq + KE
    rot = max(2, int(320e6 // (n * K * 9 / 16)) + 1)
    rps = [agemm.repack_w(q["qw"].clone(), q["sfw"].clone()) for _ in range(rot)]
    stamps = torch.zeros((4096 * 16 * 16,), dtype=torch.int64, device=dev)
    setter(stamps.data_ptr())
    o = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    wn = torch.ones(kq, dtype=torch.bfloat16, device=dev)
    for i in range(rot):
        if mode == "packed":
            agemm.matmul_repacked(q["qx"], rps[i][0], q["sfx"], rps[i][1], q["alpha"], n, out=o, kernel="stream")
        elif mode == "rms":
            agemm.rmsnorm_matmul_repacked(q["x"], wn, 1e-6, q["idx"], KE, rps[i][0], rps[i][1], 1.0, n, out=o)
        else:
            agemm.dynamic_matmul_repacked(q["x"], q["idx"], KE, rps[i][0], rps[i][1], 1.0, n, out=o)
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().reshape(4096, 16, 16)
    rb = (n + 15) // 16
    G = max(min(rb, 256), (rb + 15) // 16)
    t = t[:G].astype(np.int64)
    assert (t[:, :, :7] > 0).all(), "a wave left no stamp"
    start = t[:, :, 0].min()
    d = np.diff(t[:, :, :7], axis=2)                      # [G, 16, 6]
    tot = t[:, :, 6] - t[:, :, 0]
    print(f"M={m} N={n} KQ={kq} {mode}: kernel span {t[:, :, 6].max() - start} ticks; per-wave total median {int(np.median(tot))}")
    for i, name in enumerate(PH):
        print(f"   {name:40s} median {int(np.median(d[:, :, i])):7d}  p90 {int(np.percentile(d[:, :, i], 90)):7d}  share {np.median(d[:, :, i]) / np.median(tot):5.1%}")
    rt = (t[:, :, 11] - t[:, :, 10]).astype(np.float64) / 100.0        # us (s_memrealtime: 100 MHz)
    print(f"   shader clock over the kernel: {np.median(tot / (rt * 1e3)):.2f} GHz (median wave: {np.median(rt):.2f} us in-kernel)")
    if mode != "packed":
        print(f"   prologue: loads+stage+scale {int(np.median(t[:, :, 8] - t[:, :, 1]))}, rms sums/tree {int(np.median(t[:, :, 9] - t[:, :, 8]))}, quantise {int(np.median(t[:, :, 2] - t[:, :, 9]))}")
    print(f"   first stamp spread over workgroups: {int(t[:, 0, 0].max() - t[:, 0, 0].min())} ticks; last end - first start: {int(t[:, :, 6].max() - start)}")
    stamps.zero_()


if __name__ == "__main__":
    for (m, n, kq) in [(1, 4096, 4096), (4, 3584, 3584), (4, 10752, 3584), (4, 37888, 3584), (4, 3584, 18944)]:
        for mode in ("packed", "dyn") + (("rms",) if kq <= 8192 else ()):
            run(m, n, kq, mode)
