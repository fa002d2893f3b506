#!/usr/bin/env python3
"""Per-kernel timeline of the last N launches of this library in a rocprofv3 kernel trace (single chunk runs):
    python tools/timeline.py TRACE.csv [N] [SKIP]  -> start offset, queue, gap to the previous end on that queue, duration
    python tools/timeline.py TRACE.csv --gaps      -> histogram of the gaps between consecutive launches of a queue"""
import csv, re, sys
from collections import Counter, defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1]))
        if not any(t in r["Kernel_Name"] for t in ("at::", "elementwise", "vectorized", "Memcpy", "rocprim", "hipcub", "fillBuffer"))]
rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
def short(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    return re.sub(r"^void ", "", n).split("(")[0][:28]
if len(sys.argv) > 2 and sys.argv[2] == "--gaps":
    last, gaps, per = {}, [], defaultdict(list)
    for r in rows[len(rows) // 3:]:      # steady state
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
        if q in last:
            g = (s - last[q][0]) / 1e3
            gaps.append(g)
            per[(last[q][1], short(r))].append(g)
        last[q] = (e, short(r))
    edges = [0.5, 1, 2, 3, 5, 8, 12, 20, 50, 100, 1e9]
    hist = Counter()
    for g in gaps:
        hist[next(i for i, e in enumerate(edges) if g <= e)] += 1
    lo = 0
    print(f"{len(gaps)} boundaries between consecutive launches of one queue (steady-state two thirds of the trace)")
    for i, e in enumerate(edges):
        print(f"  {lo:>5} .. {e if e < 1e9 else 'inf':>5} us: {hist[i]:6d}  {'#' * int(60 * hist[i] / max(1, len(gaps)))}")
        lo = e
    print("largest mean gaps by (previous kernel -> next kernel):")
    for (a, b), v in sorted(per.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]))[:12]:
        print(f"  {sum(v) / len(v):7.1f} us x{len(v):4d}  {a} -> {b}")
    sys.exit(0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[-(n + skip):len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
last_end = {}
for r in rows:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"{(s - t0) / 1e3:9.1f} us  q{q:>2s} +{gap:6.1f}  {(e - s) / 1e3:7.1f} us  {short(r):28s} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
