#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name (shortened) count, mean, median, total, sorted by total."""
import csv
import glob
import statistics
import sys
from collections import defaultdict

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.split("(")[0][-90:]
        rows[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print(f"{'kernel':92s} {'n':>6s} {'mean us':>9s} {'med us':>9s} {'total ms':>9s} {'share':>6s}")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{k:92s} {len(v):6d} {statistics.mean(v):9.2f} {statistics.median(v):9.2f} {sum(v) / 1e3:9.3f} {sum(v) / tot:6.1%}")
