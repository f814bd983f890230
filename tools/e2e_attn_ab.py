import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcquant_amd import e2e
for mode in ("sdpa", "bmm"):
    os.environ["ARCQ_E2E_DECODE_ATTENTION"] = mode
    r = e2e.bench_decode("qwen2.5-7b", batch=4, prefill=1024, steps=16, fused=True, attention="cache")
    print(json.dumps({"decode_attention": mode, "ms_per_step": r["decode_ms_per_step_graph"], "tok_per_s": r["decode_tok_per_s"]}), flush=True)
    torch.cuda.empty_cache()
