"""Calibration: what does a plain streaming read of ~10 MB / ~33 MB cost on this GPU in a graph replay?
(torch copy_ and sum kernels over rotating buffers > 256 MiB)."""
import torch, sys
dev = torch.device("cuda:0")
for mb in (9.6, 33.5, 134.0):
    n = int(mb * 1e6) // 16 * 16
    rot = max(2, int(320e6 // n) + 1)
    srcs = [torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255) for _ in range(rot)]
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    acc = torch.zeros(1, dtype=torch.float32, device=dev)
    def run_copy():
        for s in srcs: dst.copy_(s)
    def run_sum():
        for s in srcs: torch.sum(s.view(torch.float32), dim=0, keepdim=True, out=acc)
    for name, fn in (("copy", run_copy), ("sum", run_sum)):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st): fn()
        torch.cuda.synchronize()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (10 * rot)
        traffic = n * (2 if name == "copy" else 1)
        print(f"{name} {mb} MB: {us:.2f} us/launch -> {traffic/us/1e3:.0f} GB/s", flush=True)
