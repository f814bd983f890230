"""Which torch ops move data in one prefill of the harness (two layers): aten::copy_ / SDPA with shapes and device time (torch.profiler).
Finding (round 3): per layer one K|V append copy [2, 4, 28, 1024, 128] = 47 us and SDPA 177 us; reading k / v from the qkv views instead of the
cache slices changes nothing (SDPA makes no internal copy).  usage: python tools/prefill_copies_probe.py"""
import dataclasses, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import e2e
from torch.profiler import profile, ProfilerActivity
cfg = dataclasses.replace(e2e.MODEL_CFGS["qwen2.5-7b"]); cfg.num_layers = 2
dev = torch.device("cuda:0")
with torch.no_grad():
    model = e2e.DecoderModel(cfg, 4, 1032, dev, fused=True, attention="cache")
    tok = torch.randint(100, 200, (4, 1024), device=dev)
    model.forward(tok, 0); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        model.forward(tok, 0); torch.cuda.synchronize()
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if ev.key in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::fill_", "aten::zero_", "aten::_scaled_dot_product_flash_attention", "aten::scaled_dot_product_attention", "aten::reshape", "aten::repeat_interleave", "aten::expand"):
        print(ev.key, ev.count, round(ev.device_time_total, 1), ev.input_shapes, [s for s in ev.stack if "e2e.py" in s or "agemm.py" in s][:3])
