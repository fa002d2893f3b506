#!/usr/bin/env python3
"""Bitwise agreement of every geometry stage, GPU vs oracle, on the parity tests' fixtures (diagnostic: which stage
is not yet the same program).   python tools/stage_bits.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.is_available()
from scipy.spatial.transform import Rotation as Rot
from geom_fixtures import BASELINE, K4, project, scene_points, two_view
from oracle import orc
from ros_stereo_slam_amd import capi

ctx = capi.Context(0)

def bits(a, b, name):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    same = a.tobytes() == b.tobytes()
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    nd = int((a != b).sum())
    print(f"  {name}: {'BIT-IDENTICAL' if same else f'{nd} of {a.size} differ, max abs {d.max():.3e}, rel {float((d / np.maximum(np.abs(b), 1e-300)).max()):.3e}'}")

print("F-RANSAC")
for seed, thr in ((1, 1.0), (42, 3.0), (77, 1.0)):
    x1, x2, gt, *_ = two_view(n=4096, n_out=700, seed=seed, noise=0.15)
    gc, gm, gF, git = ctx.fransac(x1, x2, thr, seed=seed)
    oc, om, oF, oit = orc.fransac(x1, x2, thr, seed=seed)
    print(f" seed {seed} thr {thr}: iters {git}/{oit} count {gc}/{oc} mask flips {(gm != om).sum()}")
    bits(gF, oF, "F")
print("triangulate")
P1, P2 = capi.stereo_projections(*K4, BASELINE)
X = scene_points(4428, 3)
rng = np.random.default_rng(0)
a = project(X).astype(np.float32)
b = (project(X, np.eye(3), np.array([-BASELINE, 0, 0])) + rng.normal(0, 0.3, (4428, 2))).astype(np.float32)
gx, gh = ctx.triangulate(P1, P2, a, b)
ox, oh = orc.triangulate(P1, P2, a, b)
bits(gh, oh, "homogeneous (f32)")
bits(gx, ox, "xyz (f32)")
print("PnP-RANSAC")
def noisy(n, n_out, seed, noise=0.15):
    rng = np.random.default_rng(seed)
    X = scene_points(n, seed)
    R, t = Rot.from_rotvec([0.02, -0.05, 0.01]).as_matrix(), np.array([0.1, -0.05, -0.8])
    x = project(X, R, t).astype(np.float32) + rng.normal(0, noise, (n, 2)).astype(np.float32)
    out = rng.choice(n, n_out, replace=False)
    x[out] += rng.uniform(10, 50, (n_out, 2)).astype(np.float32)
    return X.astype(np.float32), x
for seed in (1, 2, 3):
    X, x = noisy(3000, 800, seed)
    for iters in (1, 100):
        gc, grv, gtv, ginl, git = ctx.pnp_ransac(X, x, K4, iterations=iters, seed=seed)
        oc, orv, otv, oinl, oit = orc.pnp_ransac(X, x, K4, iterations=iters, seed=seed)
        print(f" seed {seed} iterations {iters}: iters {git}/{oit} count {gc}/{oc} inlier xor {len(np.setxor1d(ginl, oinl))}")
        bits(grv, orv, "rvec"); bits(gtv, otv, "tvec")
