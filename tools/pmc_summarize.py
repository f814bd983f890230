#!/usr/bin/env python3
"""Per-launch means of rocprofv3 --pmc counters for one kernel over its LAST `n` dispatches (the timed, clock-settled ones).
usage: pmc_summarize.py <dir with *counter_collection.csv> <kernel name substring> [n]   -> JSON on stdout"""
import csv
import glob
import json
import sys
from collections import defaultdict

root, needle = sys.argv[1], sys.argv[2]
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
per = defaultdict(lambda: defaultdict(float))        # counter -> dispatch id -> value (summed over dimensions / XCDs)
names = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if needle not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        per[r["Counter_Name"]][d] += float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
out = {}
for c, byd in per.items():
    ids = sorted(byd)[-last_n:]
    out[c] = sum(byd[i] for i in ids) / len(ids)
    out[c + "_dispatches"] = len(ids)
print(json.dumps(out, indent=1))
