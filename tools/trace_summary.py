#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace CSV of a bench.py run: per-kernel time inside the timed region (the
last N tracking launches), waves per launch, concurrency of the tracking launches.

    python tools/trace_summary.py trace.csv [n_lk_launches_in_region]
"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^>]*>)?)", n)
    return m.group(1) if m else n


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n_lk = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    ev = []
    for r in rows:
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), g // wg * ((wg + 63) // 64)))
    ev.sort()
    lk = [e for e in ev if e[2].startswith("lk_track")]
    t0 = lk[-n_lk][0]
    t1 = max(e[1] for e in ev)
    reg = [e for e in ev if e[0] >= t0]
    print(f"region {(t1 - t0) / 1e6:.2f} ms, {len(reg)} launches, {len([e for e in reg if e[2].startswith('lk_track')])} tracking launches")
    pts = []
    for a, b, n, _ in reg:
        if n.startswith("lk_track"):
            pts += [(a, 1), (b, -1)]
    pts.sort()
    c, last, hist = 0, pts[0][0], collections.Counter()
    for t, d in pts:
        hist[c] += t - last
        last = t
        c += d
    print("tracking launches in flight (share of the region):", {k: round(v / (t1 - t0), 3) for k, v in sorted(hist.items())})
    agg = collections.defaultdict(lambda: [0, 0, 0, 1 << 60])
    for a, b, n, w in reg:
        x = agg[n]
        x[0] += b - a
        x[1] += 1
        x[2] += w
        x[3] = min(x[3], b - a)
    tot = sum(v[0] for v in agg.values())
    print(f"{'kernel':30s} {'ms':>8s} {'launches':>8s} {'avg us':>8s} {'min us':>8s} {'waves':>8s} {'share':>6s}")
    for n, (d, k, w, mn) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"{n:30s} {d / 1e6:8.2f} {k:8d} {d / k / 1e3:8.1f} {mn / 1e3:8.1f} {w // k:8d} {100 * d / tot:5.1f}%")
    print(f"non-tracking share of kernel time: {100 * (1 - agg[lk[0][2]][0] / tot):.1f} %")


if __name__ == "__main__":
    main()
