"""MI355X-native stereo-VO + pose-graph hot path (drop-in for the hot loop of
Gautham-JS/ROS_Stereo_SLAM).  The product is ``libsvo_hip.so`` (HIP kernels for gfx950
behind the C ABI of ``include/svo.h``); this package is the thin host binding plus the
synthetic-sequence generator the benchmark and the tests use.
"""
from . import capi, chunked, slam, synth  # noqa: F401
from .capi import Context, Pyramid, SvoError  # noqa: F401

from .capi import PoseGraph, VisualOdometry  # noqa: F401
from .slam import StereoSlam  # noqa: F401

__all__ = ["capi", "chunked", "slam", "synth", "Context", "Pyramid", "SvoError", "PoseGraph", "VisualOdometry",
           "StereoSlam"]
