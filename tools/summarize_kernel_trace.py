#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name (shortened) count, mean, median, total, sorted by total.
usage: summarize_kernel_trace.py <dir> [top_n] [--last N substring]: the mean over the LAST N dispatches of the kernels whose name
contains `substring` (bench.py's timed steps come last: compare with roofline.kernel_us of the same run)."""
import csv
import glob
import statistics
import sys
from collections import defaultdict

path = sys.argv[1]
last_n, last_sub = None, None
if "--last" in sys.argv:
    i = sys.argv.index("--last")
    last_n, last_sub = int(sys.argv[i + 1]), sys.argv[i + 2]
    del sys.argv[i:i + 3]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = defaultdict(list)
picked = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if last_sub and last_sub in name:
            picked.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        short = name.split("(")[0][-90:]
        rows[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print(f"{'kernel':92s} {'n':>6s} {'mean us':>9s} {'med us':>9s} {'total ms':>9s} {'share':>6s}")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{k:92s} {len(v):6d} {statistics.mean(v):9.2f} {statistics.median(v):9.2f} {sum(v) / 1e3:9.3f} {sum(v) / tot:6.1%}")
if last_sub:
    picked.sort()
    tail = [d for _, d in picked[-last_n:]]
    head = [d for _, d in picked[:-last_n]]
    print(f"\n'{last_sub}': {len(picked)} dispatches; the last {len(tail)} (the timed steps) average {statistics.mean(tail):.2f} us "
          f"(median {statistics.median(tail):.2f}); the {len(head)} before them (pre-warm + warm-up) average "
          f"{statistics.mean(head) if head else float('nan'):.2f} us")
