"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) median/min/max duration in us."""
import csv, collections, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if filt and filt not in n:
            continue
        key = (n.split("(")[0][-60:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:60s} grid=({k[1]},{k[2]}) wg={k[3]:>4s} n={len(v):5d} med={v[len(v)//2]:9.2f} min={v[0]:9.2f} max={v[-1]:9.2f} us")
