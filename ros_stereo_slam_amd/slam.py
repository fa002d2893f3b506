"""Host-side frame loop with the pose graph in it: the body of visualSLAM::initSequence
(``src/VisualSLAM.cpp:54-169``) over a front-end (``VisualOdometry``) and a ``PoseGraph``.

The loop-closure detector of the reference (ORB + DLoopDetector,
``src/optimizationStuff.cpp:49-64``) is either a ``detector`` object (``capi.LoopDetector`` or the
oracle's ``LoopDetector``: ``detect(left)`` is called once per frame, frame 0 included) or, without
one, an externally supplied ``loop_match`` (the matched earlier frame id, or -1).  The reference's
gating is applied here: accept iff ``query - match > 100`` and the cooldown is 0, then
``LCidx = match - 1`` and ``cooldown = 100`` (``src/optimizationStuff.cpp:58-63``).

On an accepted closure the reference (``src/VisualSLAM.cpp:76-86``) adds the identity-measurement
loop edge from the PREVIOUS vertex, adds the current vertex, optimises the whole graph for 10
Gauss-Newton iterations and re-anchors only the translation of the current pose (the rotation
keeps its un-optimised value); the frame is then forced to be a keyframe (``:120``).

The same class drives the GPU objects of ``capi`` and the CPU oracle objects in the tests, so
the policy is exercised identically on both.
"""
from __future__ import annotations

import numpy as np

from .chunked import pose7


class StereoSlam:
    def __init__(self, vo, pose_graph, min_gap: int = 100, cooldown: int = 100, optimize_iters: int = 10,
                 detector=None):
        self.vo, self.pg, self.detector = vo, pose_graph, detector
        self.min_gap, self.cooldown_frames, self.optimize_iters = min_gap, cooldown, optimize_iters
        self.frame = 0
        self.cooldown = 0
        self.shutdown = False
        self.trajectory: list[tuple[np.ndarray, np.ndarray]] = []  # pose per frame as produced (VisualSLAM.cpp:93-97)
        self.keyframes: list[int] = []
        self.closures: list[tuple[int, int]] = []
        self.chi2: list[np.ndarray] = []

    def start(self, left, right) -> int:
        """Frame 0: stereoTriangulate + initializeGraph (src/VisualSLAM.cpp:22-41)."""
        n = self.vo.init(left, right)
        if self.detector is not None:
            self.detector.detect(left)  # the reference's detector sees every frame from the first
        self.frame = 0
        self.trajectory = [(np.eye(3), np.zeros(3))]
        self.keyframes = [0]
        return n

    def step(self, left, right, loop_match: int = -1):
        """One frame.  Returns (ok, R, t, info)."""
        if self.shutdown:
            return False, None, None, {}
        # a detector with a queue (capi.LoopDetector on a context of its own) works on this frame BESIDE the
        # front-end's localisation: queued here, its verdict collected once the pose is there
        queued = self.detector is not None and hasattr(self.detector, "submit")
        if queued:
            self.detector.submit(left)
        res = self.vo.localize(left)
        rc, R, t, n_inl, n_trk = res
        self.frame += 1
        if rc:
            self.shutdown = True  # SHUTDOWN_FLAG, src/VisualSLAM.cpp:65-67
            if queued:
                self.detector.collect()
            return False, R, t, {"inliers": n_inl, "tracked": n_trk}
        if self.detector is not None:
            r = self.detector.collect() if queued else self.detector.detect(left)
            loop_match = r["match"] if r["status"] == 0 else -1
        lc = False
        if loop_match >= 0 and (self.frame - loop_match) > self.min_gap and self.cooldown == 0:
            lc = True
            lc_idx = max(loop_match - 1, 0)  # the reference indexes vertices[-1] when match == 0
            self.cooldown = self.cooldown_frames
        if lc:
            self.pg.add_loop_closure(lc_idx)            # stageForPGO(..., true)
            self.pg.augment_node(pose7(R, t))           # stageForPGO(..., false)
            chi2 = self.pg.optimize(self.optimize_iters)
            est = self.pg.estimates()
            t = est[-1][:3].copy()                      # only t is re-anchored (VisualSLAM.cpp:81-82)
            self.closures.append((self.frame, lc_idx))
            self.chi2.append(chi2)
        else:
            self.pg.augment_node(pose7(R, t))
        upd = self.vo.update(right, R, t, n_inl, lc)
        kf = upd[-1] if isinstance(upd, tuple) else upd
        if kf:
            self.keyframes.append(self.frame)
        if self.cooldown:
            self.cooldown -= 1
        self.trajectory.append((np.array(R), np.array(t)))
        return True, R, t, {"inliers": n_inl, "tracked": n_trk, "keyframe": bool(kf), "loop_closure": lc}

    def optimized_translations(self) -> np.ndarray:
        """Translations of every vertex of the pose graph (what updateOdometry consumes,
        src/optimizationStuff.cpp:17-25)."""
        return self.pg.estimates()[:, :3].copy()
