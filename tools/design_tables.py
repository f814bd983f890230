#!/usr/bin/env python3
"""Regenerate the two measured tables of DESIGN.md (between the <!-- decode-table --> / <!-- results --> markers) from
profiles/r03_bench_full.json and profiles/r03_bench_profiled_stdout.json, so the document quotes the committed run."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_full.json")))
prof = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_profiled_stdout.json")))
trace = open(os.path.join(ROOT, "profiles", "r03_bench_kernel_trace_summary.txt")).read()
e = d["extra"]
names = {"decode_gemm_M1_N4096_KQ4096": "config[1] M=1 N=4096 KQ=4096", "decode_gemm_M4_N4096_KQ4096": "M=4 N=4096 KQ=4096",
         "decode_gemm_M16_N4096_KQ4096": "M=16 N=4096 KQ=4096", "decode_gemm_M1_N14336_KQ4096": "config[2] gate/up M=1 N=14336 KQ=4096",
         "decode_gemm_M1_N4096_KQ14336": "config[2] down M=1 N=4096 KQ=14336", "decode_gemm_M1_N1024_KQ4096": "config[2] k/v M=1 N=1024 KQ=4096",
         "decode_gemm_M4_N3584_KQ3584": "config[3] o M=4 N=3584 KQ=3584", "decode_gemm_M4_N10752_KQ3584": "config[3] q\\|k\\|v M=4 N=10752 KQ=3584",
         "decode_gemm_M4_N37888_KQ3584": "config[3] gate\\|up M=4 N=37888 KQ=3584", "decode_gemm_M4_N3584_KQ18944": "config[3] down M=4 N=3584 KQ=18944",
         "decode_gemm_M4_N1280_KQ8192": "config[4] rank q\\|k\\|v M=4 N=1280 KQ=8192", "decode_gemm_M4_N7168_KQ8192": "config[4] rank gate\\|up M=4 N=7168 KQ=8192",
         "decode_gemm_M4_N8192_KQ1024": "config[4] rank o M=4 N=8192 K slice 1088", "decode_gemm_M4_N8192_KQ3584": "config[4] rank down M=4 N=8192 K slice 3648"}
t = ("| shape (KE = 64) | reference layout µs | repacked µs | TB/s (best) | of 8 TB/s | fp16 `torch.matmul` µs | fastest fp16 call µs (TB/s) | speed-up vs fastest |\n"
     "|---|---|---|---|---|---|---|---|\n")
for k, n in names.items():
    if k not in e:
        continue
    v = e[k]
    t += (f"| {n} | {v['reference_layout_us']:.2f} | {v['repacked_us']:.2f} | {v['GBps'] / 1000:.2f} | {v['frac_hbm_peak']:.2f} | "
          f"{v['fp16_rocblas_us']:.1f} | {v['fp16_best_us']:.1f} `{v['fp16_best_call']}` ({v['fp16_best_GBps'] / 1000:.2f}) | {v['speedup_vs_fp16_best']:.2f}× |\n")
l = e["llama3_8b_layer_linears_decode"]
t += f"| Llama-3-8B layer, 7 linears, M=1 | | {l['us']:.1f} | {l['GBps'] / 1000:.2f} | {l['GBps'] / 8000:.2f} | {l['fp16_rocblas_us']:.1f} | | {l['speedup_vs_fp16_rocblas']:.2f}× (vs `torch.matmul`) |\n"
if "llama3_70b_tp8_rank_linears_decode" in e:
    l = e["llama3_70b_tp8_rank_linears_decode"]
    t += f"| Llama-3-70B layer on one of 8 ranks, 4 sharded linears, M=4 | | {l['us']:.1f} | {l['GBps'] / 1000:.2f} | {l['GBps'] / 8000:.2f} | {l['fp16_rocblas_us']:.1f} | | {l['speedup_vs_fp16_rocblas']:.2f}× (vs `torch.matmul`) |\n"
p = e["qwen2.5-7b_e2e_reference_protocol"]
hg, ea = p["hip_graph"], p["eager"]
m = re.search(r"last 200 \(the timed steps\) average ([0-9.]+) us", trace)
cpu_s = d["cpu_baseline"]["sample"].split("best of 2 = ")[1].split(" ")[0]
sw = e["reference_m_sweep"]
sweep = "| M | reference layout µs | repacked µs | TFLOP/s | GB/s | bound | fraction of the bounding roof |\n|---|---|---|---|---|---|---|\n"
for k, v in sw.items():
    if k == "note":
        continue
    sweep += (f"| {k[1:]} | {v.get('reference_layout_us', v['us']):.2f} | {('%.2f' % v['repacked_us']) if 'repacked_us' in v else ''} | {v['TFLOPs']:.1f} | "
              f"{v['GBps']:.0f} | {v['bound']} | {v['frac_of_bounding_roof']:.3f} |\n")
pg = e["prefill_gemms"]
pre = "| GEMM (M = 4096, KE = 64) | µs | TFLOP/s | of the fp16 roof | µs with the model's bias / residual |\n|---|---|---|---|---|\n"
for k, v in pg.items():
    if isinstance(v, dict):
        tail = f"{v['us_as_in_model']:.1f} ({v.get('epilogue_operand', 'all four')})" if "us_as_in_model" in v else ""
        pre += f"| {k} | {v['us']:.1f} | {v['TFLOPs']:.0f} | {v['frac']:.3f} | {tail} |\n"
res = f"""**Results of the committed run** (`profiles/r03_bench_full.json`, one MI355X; the same command under `rocprofv3 --kernel-trace --stats`:
`profiles/r03_bench_kernel_trace_summary.txt`, `r03_bench_kernel_stats.csv`). Box to box the sustained headline ranged 1342–1379 TFLOP/s this
round (`roofline.frac` 0.538–0.553), the library GEMM beside it 1426–1447 on randn operands:

| what | value |
|---|---|
| headline: ARC-NVFP4 GEMM M=N=KQ=4096, KE=64 | **{d['value']:.0f} TFLOP/s**, {d['roofline']['kernel_us']:.1f} µs per launch, `roofline.frac` **{d['roofline']['frac']:.3f}** of the 2.5 PFLOP/s fp16 roof ({d['roofline']['frac_of_fp4_peak']:.3f} of the fp4 roof) |
| same launch under rocprofv3 | the 200 timed dispatches average {m.group(1)} µs in the kernel trace against {prof['roofline']['kernel_us']:.2f} µs from that run's own events |
| fp16 library GEMM, same shape, same state | on randn operands {e['gemm_4096']['fp16_rocblas_TFLOPs']:.0f} TFLOP/s (ours {e['gemm_4096']['speedup_vs_fp16_rocblas']:.3f}×), on the SAME dequantised operand values {e['gemm_4096']['fp16_rocblas_same_values_TFLOPs']:.0f} (ours {e['gemm_4096']['speedup_vs_fp16_rocblas_same_values']:.3f}×); 8192²: ours {e['gemm_8192']['TFLOPs']:.0f} vs {e['gemm_8192']['fp16_rocblas_TFLOPs']:.0f} ({e['gemm_8192']['speedup_vs_fp16_rocblas']:.3f}×) / {e['gemm_8192']['fp16_rocblas_same_values_TFLOPs']:.0f} ({e['gemm_8192']['speedup_vs_fp16_rocblas_same_values']:.3f}×) |
| `cpu_baseline` (port of the reference's fake path, {d['cpu_baseline']['cores']} host threads, full workload) | {d['cpu_baseline']['value'] * 1000:.1f} GFLOP/s-equivalent ({cpu_s} s per step) |
| quantiser, static, graph replay, inputs rotated through > 320 MB | 4096²: {e['quantize_x_4096']['us']:.1f} µs = {e['quantize_x_4096']['GBps'] / 1000:.2f} TB/s ({e['quantize_x_4096'].get('us_same_input', float('nan')):.1f} µs with the same input every launch); 8192²: {e['quantize_x_8192']['us']:.1f} µs = {e['quantize_x_8192']['GBps'] / 1000:.2f} TB/s |
| Qwen2.5-7B shape, bs = 4, reference protocol (prefill 1024 + 128 decode steps over the growing cache, biases on) | HIP graph: prefill {hg['prefill_ms'][0]:.1f} ms ({hg['prefill_tok_per_s']:.0f} tok/s), decode {hg['decode_ms'][0]:.1f} ± {hg['decode_ms'][1]:.1f} ms = **{hg['decode_tok_per_s']:.0f} tok/s**, e2e {hg['e2e_ms'][0]:.1f} ms, peak {hg['peak_memory_gb']:.1f} GB; eager launches: decode {ea['decode_ms'][0]:.1f} ms = {ea['decode_tok_per_s']:.0f} tok/s |
| one decode step at 1040 cached tokens | full-cache attention {e['qwen2.5-7b_decode_step_full_cache']['decode_tok_per_s']:.0f} tok/s (round 2: 1703–1715; with torch SDPA instead of the harness kernel {e['qwen2.5-7b_decode_step_full_cache_torch_sdpa_attention']['decode_tok_per_s']:.0f}, round 2: 1152–1160); current-token attention (the reference harness's quirk) {e['qwen2.5-7b_decode_step_current_token_attention_harness_quirk']['decode_tok_per_s']:.0f} (round 2: 2066–2101); the reference's unfused call structure on the same kernels {e['qwen2.5-7b_decode_step_reference_call_structure']['decode_tok_per_s']:.0f} |

The reference's own kernel benchmark (`kernels/bench.py:8-49`: `agemm.matmul` at N = K = 4096, KE = 0; M ≤ 512 HBM-cold under graph replay):

{sweep}
The prefill GEMMs of the model (Qwen2.5-7B, sustained launches):

{pre}"""
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
s = re.sub(r"<!-- decode-table -->.*?<!-- /decode-table -->", "<!-- decode-table -->\n" + t + "<!-- /decode-table -->", s, flags=re.S)
s = re.sub(r"<!-- results -->.*?<!-- /results -->", "<!-- results -->\n" + res + "<!-- /results -->", s, flags=re.S)
open(path, "w").write(s)
print("DESIGN.md tables regenerated")
