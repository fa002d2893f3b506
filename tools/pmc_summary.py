#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 counter-collection output (profiles/r01_pmc_*.csv are made with it).

    python tools/pmc_summary.py sq    gpurun_out/pmc_sq/*_counter_collection.csv      > profiles/..._sq.csv
    python tools/pmc_summary.py hbm   fetch_counter_collection.csv write_counter_collection.csv > profiles/..._hbm.csv

With SVO_PMC_LAST=N only the last N dispatches of every kernel are averaged (the timed region of a
bench run: the chunks' initial keyframes and the warm-up steps come first).

`hbm` joins the FETCH_SIZE and WRITE_SIZE passes (the two counters cannot share a pass) and adds
(2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950 correction of MI355X_MICROARCH.md for kernels that
read 16 bytes per lane.  Kernel names are shortened to the function name.
"""
import csv
import re
import sys
from collections import OrderedDict, defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def load(paths):
    import os
    last = int(os.environ.get("SVO_PMC_LAST", "0"))
    rows = []
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                rows.append((p, short(row["Kernel_Name"]), int(row["Dispatch_Id"]), row["Counter_Name"],
                             float(row["Counter_Value"])))
    keep = None
    if last > 0:
        ids = defaultdict(set)
        for p, k, d, _, _ in rows:
            ids[(p, k)].add(d)
        keep = {pk: set(sorted(v)[-last:]) for pk, v in ids.items()}
    per = defaultdict(lambda: defaultdict(float))     # kernel -> counter -> sum
    disp = defaultdict(set)
    for p, k, d, c, v in rows:
        if keep is not None and d not in keep[(p, k)]:
            continue
        per[k][c] += v
        disp[k].add((p, d))
    return per, disp


def main():
    mode, paths = sys.argv[1], sys.argv[2:]
    w = csv.writer(sys.stdout, lineterminator="\n")
    if mode == "sq":
        per, disp = load(paths)
        names = OrderedDict()
        for k in per:
            for c in per[k]:
                names[c] = 1
        names = list(names)
        w.writerow(["kernel", "dispatches"] + [c + "_per_dispatch" for c in names])
        for k in sorted(per, key=lambda k: -per[k].get("SQ_INSTS_VALU", 0)):
            n = len(disp[k])
            w.writerow([k, n] + [int(round(per[k].get(c, 0) / n)) for c in names])
    elif mode == "hbm":
        fetch, dF = load(paths[:1])
        write, dW = load(paths[1:2])
        w.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_per_dispatch", "WRITE_SIZE_KB_per_dispatch",
                    "hbm_bytes_per_dispatch_2F_plus_W"])
        rows = []
        for k in fetch:
            n = len(dF[k])
            f = fetch[k].get("FETCH_SIZE", 0) / n
            wr = write[k].get("WRITE_SIZE", 0) / max(1, len(dW[k])) if k in write else 0.0
            rows.append((k, n, f, wr, (2 * f + wr) * 1024))
        for k, n, f, wr, b in sorted(rows, key=lambda r: -r[4] * r[1]):
            w.writerow([k, n, "%.1f" % f, "%.1f" % wr, int(round(b))])
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main()
