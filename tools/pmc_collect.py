#!/usr/bin/env python3
"""Collect the per-group JSON files of tools/scripts/prof_r03.sh (gpurun_out/<tag>_pmc_tile_<group>.json and
gpurun_out/<tag>_pmcd_<group>_<kernel>.json) into profiles/<tag>_pmc_tile_gemm.json and profiles/<tag>_pmc_decode_gemm.json:
FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md), WRITE_SIZE, wave-state shares, instruction counts.
usage: python tools/pmc_collect.py <tag>   (run in the build container on the merged gpurun_out/)"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def merged(pattern):
    rec = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern))):
        rec.update({a: b for a, b in json.load(open(f)).items() if not a.endswith("_dispatches")})
    return rec


def derive(rec, alg_bytes=None):
    r = {"FETCH_SIZE_KB_raw": rec.get("FETCH_SIZE"), "WRITE_SIZE_KB": rec.get("WRITE_SIZE"),
         "fetch_bytes_corrected_x2": rec.get("FETCH_SIZE", 0) * 1024 * 2, "write_bytes": rec.get("WRITE_SIZE", 0) * 1024}
    r["hbm_traffic_bytes"] = r["fetch_bytes_corrected_x2"] + r["write_bytes"]
    if alg_bytes:
        r["algorithmic_bytes"] = alg_bytes
    wc = rec.get("SQ_WAVE_CYCLES", 0)
    if wc:
        r["share_parked_waitcnt_or_barrier"] = round(rec["SQ_WAIT_ANY"] / wc, 3)
        r["share_issue_stalled"] = round(rec["SQ_WAIT_INST_ANY"] / wc, 3)
        r["share_issuing"] = round(rec["SQ_ACTIVE_INST_ANY"] / wc, 3)
    if rec.get("SQ_INSTS_MFMA"):
        r["valu_per_mfma"] = round(rec["SQ_INSTS_VALU"] / rec["SQ_INSTS_MFMA"], 3)
    if rec.get("SQ_WAVES"):
        r["vector_insts_per_wave"] = round(rec["SQ_INSTS_VALU"] / rec["SQ_WAVES"], 1)
    for c, v in rec.items():
        if c.startswith("SQ_") or c.startswith("GRBM_"):
            r[c] = v
    return r


tile = merged(f"{tag}_pmc_tile_*.json")
if tile:
    M = N = 4096
    K = 4160
    out = {"source": "rocprofv3 --pmc <group> --kernel-trace, one group per pass (FETCH_SIZE | WRITE_SIZE | SQ wave-state | SQ instruction counts + GRBM), "
                     "command: python3 bench.py --no-extra --no-cpu (60 ms pre-warm + 500 warm-up + 200 timed launches); means over the LAST 200 "
                     "dispatches of the kernel (tools/scripts/prof_r03.sh, tools/pmc_summarize.py, tools/pmc_collect.py); MI355X, the kernel as shipped",
           "kernel": "arcq::gemm_tile_kernel<256,256,2,4,false,0,false,true>  M=4096 N=4096 K_aug=4160",
           "per_launch": derive(tile, N * K * 9 / 16 + M * K * 9 / 16 + 2 * M * N)}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_tile_gemm.json"), "w"), indent=1)
    print("tile:", {k: out["per_launch"].get(k) for k in ("hbm_traffic_bytes", "algorithmic_bytes", "valu_per_mfma", "share_issuing")})
dec = {}
for k in ("gemm_decode_kernel", "gemm_rowblock_kernel", "gemm_stream_kernel"):
    rec = merged(f"{tag}_pmcd_*_{k}.json")
    if rec:
        dec[k] = derive(rec)
if dec:
    out = {"source": "rocprofv3 --pmc <group> --kernel-trace (one group per pass) -- python3 tools/pmc_decode_run.py: Qwen2.5-7B gate|up decode shape "
                     "(M=4, N=37888, KQ=3584, KE=64; 77.7 MB of weights, 83.4 MB repacked incl. the K padding) on the reference-layout kernel, the "
                     "repacked rowblock kernel and the fused RMSNorm + gate|up + SiLU stream kernel; means over the last 20 dispatches",
           "kernels": dec}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_decode_gemm.json"), "w"), indent=1)
    print("decode:", {k: {a: v.get(a) for a in ("fetch_bytes_corrected_x2", "share_parked_waitcnt_or_barrier", "share_issue_stalled", "share_issuing")}
                      for k, v in dec.items()})
