"""ctypes loader of the CPU oracle (oracle/_build/libsvo_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under ros_stereo_slam_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "_build" / "libsvo_oracle.so"
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", os.fspath(_HERE)], check=True)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        _lib = C.CDLL(os.fspath(LIB_PATH))
    return _lib


def _p(a):
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def pyr_sizes(w, h, levels):
    ws, hs = [w], [h]
    for _ in range(1, levels):
        ws.append((ws[-1] + 1) // 2)
        hs.append((hs[-1] + 1) // 2)
    return ws, hs


def pyr_down(img: np.ndarray) -> np.ndarray:
    h, w, c = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2, c), np.uint8)
    load().orc_pyr_down(_p(np.ascontiguousarray(img)), w, h, c, _p(out))
    return out


def scharr(img: np.ndarray) -> np.ndarray:
    h, w, c = img.shape
    out = np.empty((h, w, c, 2), np.int16)
    load().orc_scharr(_p(np.ascontiguousarray(img)), w, h, c, _p(out))
    return out


def lk_track(prev: np.ndarray, nxt: np.ndarray, pts: np.ndarray):
    h, w, c = prev.shape
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = pts.shape[0]
    out = np.zeros_like(pts)
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    mineig = np.zeros(n, np.float32)
    rc = load().orc_lk_track(_p(np.ascontiguousarray(prev)), _p(np.ascontiguousarray(nxt)), w, h, c,
                             _p(pts), n, _p(out), _p(status), _p(err), _p(mineig), None)
    if rc != 0:
        raise ValueError("orc_lk_track: bad arguments")
    return out, status, err, mineig


def grid_keypoints(rows, cols, step) -> np.ndarray:
    n = load().orc_grid_keypoints(rows, cols, step, None, 0)
    out = np.empty((n, 2), np.float32)
    load().orc_grid_keypoints(rows, cols, step, _p(out), n)
    return out
