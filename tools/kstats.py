#!/usr/bin/env python3
"""Print the kernels of this library from a rocprofv3 --stats kernel_stats.csv:  python tools/kstats.py FILE [substring ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
for r in rows:
    name = r["Name"]
    if any(t in name for t in ("at::", "elementwise", "vectorized", "rocprim", "hipcub")) and not pats:
        continue
    if pats and not any(p in name for p in pats):
        continue
    print(f'{name[:64]:64s} calls {int(r["Calls"]):7d}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
