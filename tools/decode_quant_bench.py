"""Decode-step quantiser latencies in a replayed HIP graph (tuning aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcquant_amd import agemm

dev = torch.device("cuda:0")


def graph_time(fn, n=16, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


x = bench.outlier_activations(4, 3584, dev)
gu = (torch.randn(4, 2 * 18944, device=dev) * 2).to(torch.bfloat16)
act = (torch.randn(4, 18944, device=dev) * 2).to(torch.bfloat16)
ih = torch.arange(3584, dtype=torch.int16, device=dev)
ii = torch.arange(18944, dtype=torch.int16, device=dev)
wn = torch.ones(3584, dtype=torch.bfloat16, device=dev)
xs = (x / 3).contiguous()
print("rmsnorm_quantize_x   M=4 KQ=3584 : %.2f us" % graph_time(lambda: agemm.rmsnorm_quantize_x(x, wn, 1e-6, ih, 64)))
print("reorder_quantize_x   M=4 KQ=3584 : %.2f us" % graph_time(lambda: agemm.reorder_quantize_x(xs, ih, 64)))
print("quantize_x_dynamic   M=4 KQ=3584 : %.2f us" % graph_time(lambda: agemm.reorder_quantize_x_dynamic(x, ih, 64)))
print("quantize_x_dynamic   M=4 KQ=18944: %.2f us" % graph_time(lambda: agemm.reorder_quantize_x_dynamic(act, ii, 64)))
print("silu_mul_quantize    M=4 KQ=18944: %.2f us (two launches)" % graph_time(lambda: agemm.silu_mul_quantize_x_dynamic(gu, ii, 64)))

# instruction-cache effect: the same four kernels interleaved (every launch follows a different kernel) against the sum of their times alone
fns = [lambda: agemm.rmsnorm_quantize_x(x, wn, 1e-6, ih, 64), lambda: agemm.reorder_quantize_x(xs, ih, 64),
       lambda: agemm.reorder_quantize_x_dynamic(x, ih, 64), lambda: agemm.reorder_quantize_x_dynamic(act, ii, 64)]
alone = [graph_time(f) for f in fns]
def mixed():
    for f in fns:
        f()
tm = graph_time(mixed, n=4)
print("alone: %s  sum %.2f us;  interleaved round of the four: %.2f us" % (["%.2f" % t for t in alone], sum(alone[:3]) + alone[3], tm))
