#!/usr/bin/env python3
"""Collect the per-kernel JSON files of tools/scripts/pmc_r02_decode.sh (gpurun_out/r02pmcd_<group>_<kernel>.json) into
profiles/r02_pmc_decode_gemm.json: FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE, wave-state shares, instructions per wave."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for k in ("gemm_decode_kernel", "gemm_rowblock_kernel", "gemm_stream_kernel"):
    rec = {}
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"r02pmcd_*_{k}.json")):
        rec.update({a: b for a, b in json.load(open(f)).items() if not a.endswith("_dispatches")})
    if not rec:
        continue
    r = {"FETCH_SIZE_KB_raw": rec.get("FETCH_SIZE"), "fetch_bytes_corrected_x2": rec.get("FETCH_SIZE", 0) * 1024 * 2,
         "write_bytes": rec.get("WRITE_SIZE", 0) * 1024}
    wc = rec.get("SQ_WAVE_CYCLES", 0)
    if wc:
        r["share_parked_waitcnt_or_barrier"] = round(rec["SQ_WAIT_ANY"] / wc, 3)
        r["share_issue_stalled"] = round(rec["SQ_WAIT_INST_ANY"] / wc, 3)
        r["share_issuing"] = round(rec["SQ_ACTIVE_INST_ANY"] / wc, 3)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES",
              "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"):
        if c in rec:
            r[c] = rec[c]
    if rec.get("SQ_WAVES"):
        r["vector_insts_per_wave"] = round(rec["SQ_INSTS_VALU"] / rec["SQ_WAVES"], 1)
    out[k] = r
path = os.path.join(ROOT, "profiles", "r02_pmc_decode_gemm.json")
old = json.load(open(path)) if os.path.exists(path) else {}
old["kernels"] = out
json.dump(old, open(path, "w"), indent=1)
print(json.dumps({k: {a: v[a] for a in ("fetch_bytes_corrected_x2", "share_parked_waitcnt_or_barrier", "share_issue_stalled", "share_issuing",
                                         "vector_insts_per_wave") if a in v} for k, v in out.items()}, indent=1))
