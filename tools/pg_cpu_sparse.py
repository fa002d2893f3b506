#!/usr/bin/env python3
"""A fair CPU comparator for the pose-graph solve (the oracle factorises in time order, which flatters
the GPU): the same Gauss-Newton normal equations of the benchmark graph, assembled from the oracle's edge linearisation
(orc_se3_edge_error) and solved by a FILL-REDUCING sparse direct solver -- scipy's SuperLU with its default COLAMD column
ordering -- on the host.  Test infrastructure (it uses oracle/); prints the time of one factorisation + solve.

    python tools/pg_cpu_sparse.py [V] [C]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import orc
from ros_stereo_slam_amd import chunked, synth
from ros_stereo_slam_amd.chunked import pose7

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 40
poses = synth.loop_trajectory(V, **synth.BENCH_LOOP)
R0, t0 = poses[0]
rng = np.random.default_rng(1)
traj, drift = [], np.zeros(3)
for R, t in poses:
    drift = drift + rng.normal(0, 0.002, 3)
    traj.append((R0.T @ R, R0.T @ (t - t0) + drift))
matches = synth.loop_closures(poses, max_dist=0.3, max_angle_deg=10.0, min_gap=100, pick="nearest")
closures = dict(list(chunked.gate_closures([m if m >= 1 else -1 for m in matches]).items())[:Cn])
g = orc.PoseGraph()
for q in range(1, len(traj)):
    m = closures.get(q, -1)
    if m >= 0:
        g.add_loop_closure(max(m - 1, 0))
    g.augment_node(pose7(*traj[q]))
X = g.estimates()
edges = list(g.edges())
lib = orc.load()
fn = lib.orc_se3_edge_error
fn.restype = None
dp = C.POINTER(C.c_double)
nb = V - 1
rows, cols, vals = [], [], []
b = np.zeros(6 * nb)
t_lin = time.perf_counter()
e6, Ji, Jj = np.zeros(6), np.zeros(36), np.zeros(36)
for (i, j, Z) in edges:
    Zc = np.ascontiguousarray(Z, np.float64)
    fn(X[i].ctypes.data_as(dp), X[j].ctypes.data_as(dp), Zc.ctypes.data_as(dp), e6.ctypes.data_as(dp),
       Ji.ctypes.data_as(dp), Jj.ctypes.data_as(dp))
    A, B = Ji.reshape(6, 6), Jj.reshape(6, 6)
    for (u, Ju) in ((i, A), (j, B)):
        if u == 0:
            continue
        b[6 * (u - 1):6 * u] -= Ju.T @ e6
        for (v, Jv) in ((i, A), (j, B)):
            if v == 0:
                continue
            blk = Ju.T @ Jv
            r, c = np.meshgrid(np.arange(6 * (u - 1), 6 * u), np.arange(6 * (v - 1), 6 * v), indexing="ij")
            rows.append(r.ravel()); cols.append(c.ravel()); vals.append(blk.ravel())
t_lin = time.perf_counter() - t_lin
H = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * nb, 6 * nb))
print(f"{V} vertices, {len(closures)} closures: H {H.shape[0]} x {H.shape[0]}, {H.nnz} nonzeros "
      f"(python assembly {t_lin * 1e3:.0f} ms, not part of the figures)")
# scipy has no sparse Cholesky (CHOLMOD is not installed): SuperLU in its symmetric mode (minimum degree on A^T + A, no
# pivoting off the diagonal) is the closest a stock install offers; COLAMD is its default for unsymmetric matrices
for spec, kw in (("MMD_AT_PLUS_A", dict(diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))), ("COLAMD", {})):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        lu = spla.splu(H, permc_spec=spec, **kw)
        dx = lu.solve(b)
        best = min(best, time.perf_counter() - t0)
    print(f"  SuperLU, {spec}: factor + solve {best * 1e3:.1f} ms on one host thread; fill {lu.L.nnz + lu.U.nnz} nonzeros; "
          f"residual {np.abs(H @ dx - b).max():.2e}")
t0 = time.perf_counter()
g.optimize(1)
print(f"the oracle's time-ordered skyline Cholesky, one Gauss-Newton iteration incl. linearisation: {(time.perf_counter() - t0) * 1e3:.1f} ms")
