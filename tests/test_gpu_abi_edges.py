"""Error behaviour and degenerate sizes of the C ABI (include/svo.h): every entry point returns a
status instead of faulting -- empty inputs are SVO_OK with empty outputs, bad arguments are
SVO_ERR_ARG with a message in svo_last_error(), and nothing touches the GPU for them."""
import ctypes as C

import numpy as np
import pytest

from ros_stereo_slam_amd import capi

pytestmark = pytest.mark.gpu
K4 = (718.856, 718.856, 607.1928, 185.2157)


def test_empty_inputs_are_ok(ctx):
    e2, e3 = np.zeros((0, 2), np.float32), np.zeros((0, 3), np.float32)
    a = capi.Pyramid(ctx, 128, 96, 1).build(np.zeros((96, 128, 1), np.uint8))
    out, st, err, eig = ctx.lk_track(a, a, e2)
    assert len(out) == 0 and len(st) == 0
    cnt, mask, F, it = ctx.fransac(e2, e2, 1.0)
    assert cnt == 0 and len(mask) == 0
    assert len(ctx.triangulate(*capi.stereo_projections(*K4, 0.54), e2, e2)[0]) == 0
    assert len(ctx.transform_points(np.eye(3, 4), e3)) == 0
    assert len(ctx.anms(e2, np.zeros(0, np.float32), 10)) == 0
    assert ctx.pnp_ransac(e3, e2, K4)[0] == 0
    assert len(ctx.compact(np.zeros(0, np.uint8), e2)[0]) == 0
    a.close()


def test_too_few_points_for_a_model(ctx):
    rng = np.random.default_rng(0)
    p = rng.uniform(0, 300, (6, 2)).astype(np.float32)
    cnt, mask, F, it = ctx.fransac(p, p + 1, 1.0)          # 6 < 7 correspondences: no model, empty mask
    assert cnt == 0 and not mask.any()
    X = rng.uniform(1, 5, (4, 3)).astype(np.float32)
    assert ctx.pnp_ransac(X, p[:4], K4)[0] == 0             # 4 < 5


def test_bad_arguments_are_refused_with_a_message(ctx):
    lib = ctx.lib
    buf = np.zeros((8, 2), np.float32)
    n = C.c_int()
    assert lib.svo_grid_keypoints(None, 100, 100, 10, None, 0, capi.MEM_HOST, C.byref(n)) == capi.SVO_ERR_ARG
    assert b"bad argument" in lib.svo_last_error()
    assert lib.svo_transform_points(ctx._h, None, capi._ptr(buf), 4, capi._ptr(buf), capi.MEM_HOST) == capi.SVO_ERR_ARG
    assert lib.svo_transform_points(ctx._h, capi._ptr(np.eye(3, 4)), capi._ptr(buf), 4, capi._ptr(buf), 7) == capi.SVO_ERR_ARG
    h = C.c_void_p()
    assert lib.svo_pyramid_create(ctx._h, 1, 1, 3, 4, C.byref(h)) == capi.SVO_ERR_ARG       # image too small
    assert lib.svo_pyramid_create(ctx._h, 640, 480, 2, 4, C.byref(h)) == capi.SVO_ERR_ARG   # 1 or 3 channels
    assert lib.svo_vo_create(ctx._h, None, 0, 0, 3, C.byref(h)) == capi.SVO_ERR_ARG
    assert lib.svo_ctx_create(9999, C.byref(h)) in (capi.SVO_ERR_ARG, capi.SVO_ERR_NO_DEVICE)
    with pytest.raises(capi.SvoError):
        capi.LoopDetector(ctx, 640, 480, 3, n_features=4)                                   # 8..2048
    with pytest.raises(capi.SvoError):
        ctx.orb_extract(np.zeros((40, 40), np.uint8))                                        # smaller than 4 margins


def test_pose_graph_refuses_bad_ids_and_reports_state(ctx):
    g = capi.PoseGraph(ctx)
    assert g.num_vertices == 1 and g.num_edges == 0
    assert np.all(g.optimize(3) == 0)                       # nothing to optimise: chi2 zeros
    with pytest.raises(capi.SvoError):
        g.add_loop_closure(5)                               # no such vertex
    g.augment_node([1, 0, 0, 0, 0, 0, 1])
    with pytest.raises(capi.SvoError):
        g.add_loop_closure(-1)
    g.close()
