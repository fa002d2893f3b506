"""Generates tests/golden/posegraph_loop40.npz: the 40-vertex drifting loop with one identity
loop-closure edge (last vertex -> vertex 0), 4 Gauss-Newton iterations of the CPU oracle.
test_oracle_posegraph.py cross-checks the same iterates against an independent numpy/scipy
Gauss-Newton (numeric Jacobians, dense solve) before trusting them.
Run from the repo root:  python tests/golden/make_posegraph_golden.py"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import orc  # noqa: E402
from pg_fixtures import drifting_loop  # noqa: E402

gt, est = drifting_loop(40)
g = orc.PoseGraph()
for i in range(1, 40):
    g.augment_node(est[i])
g.add_loop_closure(0)
start = g.estimates()
chi2 = g.optimize(4)
np.savez(pathlib.Path(__file__).with_name("posegraph_loop40.npz"), start=start, after4=g.estimates(), chi2=chi2)
print("chi2", chi2)
