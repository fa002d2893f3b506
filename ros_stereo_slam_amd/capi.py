"""ctypes binding of ``libsvo_hip.so`` (``include/svo.h``).

The shared library is the product; this module is plumbing.  It never falls back to a CPU
implementation: if the library is missing it raises, and ``Context()`` raises when no
gfx950 device is present.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import re

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "libsvo_hip.so"
HEADER_PATH = _HERE.parent / "include" / "svo.h"

SVO_OK = 0
SVO_ERR_NO_DEVICE = -4
SVO_ERR_TRACKING_LOST = -5
MEM_HOST, MEM_DEVICE = 0, 1
K_PYRAMID, K_LK, K_FRANSAC, K_TRIANGULATE, K_PNP, K_POSEGRAPH, K_ANMS = range(7)


class SvoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsvo_hip error {code}: {msg}")
        self.code = code


_lib = None


def declared_symbols() -> list[str]:
    """Every function ``include/svo.h`` declares (used by the export test)."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svo_[a-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    """Load the library (building nothing).  Raises if it is absent -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C ros_stereo_slam_amd/csrc` "
            "(or __graft_entry__.build()); the HIP path has no CPU fallback")
    lib = C.CDLL(os.fspath(LIB_PATH))
    lib.svo_last_error.restype = C.c_char_p
    lib.svo_ctx_stream.restype = C.c_void_p
    _lib = lib
    return lib


def _check(rc: int):
    if rc != SVO_OK:
        raise SvoError(rc, load().svo_last_error().decode(errors="replace"))


def _ptr(a):
    """Pointer of a numpy array (host) or an int / object with data_ptr() (device)."""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


class Pyramid:
    def __init__(self, ctx: "Context", w: int, h: int, c: int, levels: int = 4):
        self.ctx, self.w, self.h, self.c, self.levels = ctx, w, h, c, levels
        self._h = C.c_void_p()
        _check(ctx.lib.svo_pyramid_create(ctx._h, w, h, c, levels, C.byref(self._h)))

    def build(self, image, mem: int = MEM_HOST):
        if isinstance(image, np.ndarray):
            assert image.dtype == np.uint8 and image.shape == (self.h, self.w, self.c)
        _check(self.ctx.lib.svo_pyramid_build(self.ctx._h, self._h, _ptr(image), mem))
        return self

    def level(self, l: int) -> np.ndarray:
        w, h = C.c_int(), C.c_int()
        _check(self.ctx.lib.svo_pyramid_get_level(self.ctx._h, self._h, l, None, MEM_HOST,
                                                  C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value, self.c), np.uint8)
        _check(self.ctx.lib.svo_pyramid_get_level(self.ctx._h, self._h, l, _ptr(out), MEM_HOST,
                                                  C.byref(w), C.byref(h)))
        return out

    def close(self):
        if self._h:
            self.ctx.lib.svo_pyramid_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One per device: owns the HIP stream, scratch and kernel timers (``svo_ctx``)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self._h = C.c_void_p()
        _check(self.lib.svo_ctx_create(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            self.lib.svo_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(self.lib.svo_ctx_sync(self._h))

    @property
    def stream(self) -> int:
        return self.lib.svo_ctx_stream(self._h)

    def enable_kernel_timing(self, on: bool = True):
        _check(self.lib.svo_ctx_enable_kernel_timing(self._h, int(on)))

    def reset_kernel_time(self):
        _check(self.lib.svo_ctx_reset_kernel_time(self._h))

    def kernel_time(self, kid: int):
        ms, n = C.c_double(), C.c_int()
        _check(self.lib.svo_ctx_kernel_time(self._h, kid, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- hot-path entry points (host-array convenience forms) ----
    def pyramid(self, w, h, c, levels=4) -> Pyramid:
        return Pyramid(self, w, h, c, levels)

    def grid_keypoints(self, rows: int, cols: int, step: int) -> np.ndarray:
        cnt = C.c_int()
        _check(self.lib.svo_grid_keypoints(self._h, rows, cols, step, None, 0, MEM_HOST, C.byref(cnt)))
        out = np.empty((cnt.value, 2), np.float32)
        _check(self.lib.svo_grid_keypoints(self._h, rows, cols, step, _ptr(out), cnt.value, MEM_HOST,
                                           C.byref(cnt)))
        return out

    def lk_track(self, prev: Pyramid, nxt: Pyramid, pts: np.ndarray):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = pts.shape[0]
        out = np.empty_like(pts)
        status = np.empty(n, np.uint8)
        err = np.empty(n, np.float32)
        mineig = np.empty(n, np.float32)
        _check(self.lib.svo_lk_track(self._h, prev._h, nxt._h, _ptr(pts), n, _ptr(out), _ptr(status),
                                     _ptr(err), _ptr(mineig), MEM_HOST))
        return out, status, err, mineig

    def lk_track_device(self, prev: Pyramid, nxt: Pyramid, pts, n, out, status, err=None, mineig=None):
        """Device-pointer form (torch tensors or raw addresses); asynchronous on the ctx stream."""
        _check(self.lib.svo_lk_track(self._h, prev._h, nxt._h, _ptr(pts), n, _ptr(out), _ptr(status),
                                     _ptr(err), _ptr(mineig), MEM_DEVICE))
