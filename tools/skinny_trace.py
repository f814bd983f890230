"""In-kernel timeline of the decode GEMM (debug stamps, 100 MHz wall clock)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcquant_amd import agemm, _lib
dev = torch.device("cuda:0")
M, N, KQ = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1, 4096, 4096)))
p = bench.make_problem(M, N, KQ, 64, dev)
rot = 40
qws = [p["qw"].clone() for _ in range(rot)]; sfws = [p["sfw"].clone() for _ in range(rot)]
tr = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
for i in range(rot): agemm.matmul(p["qx"], qws[i], p["sfx"], sfws[i], p["alpha"])
torch.cuda.synchronize()
_lib.lib().arcq_debug_set_trace(tr.data_ptr())
allst = []
for i in range(rot):
    tr.zero_(); torch.cuda.synchronize()
    agemm.matmul(p["qx"], qws[i], p["sfx"], sfws[i], p["alpha"])
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0]
    allst.append(t)
_lib.lib().arcq_debug_set_trace(None)
t = allst[-1].astype(np.float64)
t0 = t[:, 0].min()
print(f"workgroups: {len(t)}  (times in us from the first workgroup's start; 10 ns ticks)")
for k, name in enumerate(["start", "prefetch issued", "first item landed", "last item multiplied", "last tile stored"]):
    v = (t[:, k] - t0) / 100.0
    print(f"  {name:22s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
d = (t[:, 2] - t[:, 0]) / 100.0
print(f"  per-WG start->first landed: median {np.median(d):.2f} max {d.max():.2f};  first landed->stored: median {np.median((t[:,4]-t[:,2])/100):.2f}")
