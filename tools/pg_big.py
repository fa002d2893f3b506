import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from pg_fixtures import drifting_loop
from ros_stereo_slam_amd import capi
from oracle import orc
ctx = capi.Context(0)
def build(cls, est, closures):
    g = cls()
    cl = {}
    for a,b in closures: cl.setdefault(a, []).append(b)
    for i in range(1, len(est)):
        g.augment_node(est[i])
        for to in cl.get(i, []): g.add_loop_closure(to)
    return g
for n, laps, ncl in ((3,1,0),(4,1,1),(12000,6,150),(20000,10,300)):
    gt, est = drifting_loop(n, radius=300.0, yaw_drift=1e-5, scale_drift=1.0001, laps=max(laps,1))
    per = n // max(laps,1)
    if n == 4: closures = [(3, 1)]
    elif ncl == 0: closures = []
    else:
        rng = np.random.default_rng(5)
        at = sorted(rng.choice(np.arange(per + 10, n), size=ncl, replace=False).tolist())
        closures = [(a, a - per * int(rng.integers(1, a // per + 1))) for a in at]
        closures = [(a, b) for a, b in closures if b >= 0]
    g = build(lambda: capi.PoseGraph(ctx), est, closures)
    t0 = time.perf_counter(); cg = g.optimize(10); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); cg2 = None
    msg = f"n {n} closures {len(closures)}: GPU {tg*1e3:.1f} ms chi2 {cg[0]:.4g} -> {cg[-1]:.4g}"
    if n <= 12000:
        o = build(orc.PoseGraph, est, closures)
        t0 = time.perf_counter(); co = o.optimize(10); to = time.perf_counter() - t0
        d = np.abs(g.estimates()[:, :3] - o.estimates()[:, :3]).max()
        msg += f" | oracle {to*1e3:.0f} ms chi2 -> {co[-1]:.4g}, max |dt| {d:.2e}"
    print(msg, flush=True)
    g.close()
