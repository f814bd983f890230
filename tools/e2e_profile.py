"""Run a few graph-replayed decode steps of the Qwen2.5-7B-shape model (for rocprofv3 --kernel-trace)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd.e2e import bench_decode
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 28
fused = len(sys.argv) > 2 and sys.argv[2] == "fused"
print(json.dumps(bench_decode("qwen2.5-7b", batch=4, prefill=1024, steps=8, repeats=1, layers=layers, fused=fused)))
