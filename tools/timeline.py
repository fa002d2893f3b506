#!/usr/bin/env python3
"""Per-kernel timeline of the last N launches of this library in a rocprofv3 kernel trace (single chunk runs):
    python tools/timeline.py TRACE.csv [N]   -> start offset, duration, gap to the previous end (per stream)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))
        if not any(t in r["Kernel_Name"] for t in ("at::", "elementwise", "vectorized", "Memcpy", "rocprim", "hipcub", "fillBuffer"))]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
last_end = {}
for r in rows:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    name = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "")[:34]
    print(f"{(s - t0) / 1e3:9.1f} us  q{q:>2s} +{gap:6.1f}  {(e - s) / 1e3:7.1f} us  {name}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
