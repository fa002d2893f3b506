"""Parity on COLOUR input.  The reference feeds the tracker 3-channel BGR images whose channels
differ (KITTI image_2 / image_3 through cv::imread's default flag, the grey conversion commented out:
src/keyFrameManagement.cpp:52-54,64-66; colours read back as B, G, R: include/monoUtils.h:180-193), and OpenCV's LK
sums the normal equations over the channels.  Until round 4 every 3-channel image a parity test fed the tracker had
R = G = B, so a channel-stride or channel-order slip in the tile staging of lk_track_kernel<3>, in the packed
derivative levels or in the oracle's own loops would have been invisible.  Here every stage that reads pixels runs on
images whose channels are different textures (synth.Scene(colour=True), synth.textured_pair(colour=True)) -- against
the oracle through the C ABI, bit for bit.  tests/test_lk_independent.py holds the CPU leg: the numpy restatement of
calcOpticalFlowPyrLK against oracle/lk.c on the same images."""
import numpy as np
import pytest

from ros_stereo_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _channels_differ(img):
    d = [np.abs(img[..., i].astype(int) - img[..., j].astype(int)).mean() for i, j in ((0, 1), (0, 2), (1, 2))]
    return min(d) > 5.0


def _compare_lk(ctx, orc, a, b, pts):
    h, w, c = a.shape
    pa, pb = ctx.pyramid(w, h, c).build(a), ctx.pyramid(w, h, c).build(b)
    out, st, err, me = ctx.lk_track(pa, pb, pts)
    ro, rs, re, rm = orc.lk_track(a, b, pts)
    pa.close()
    pb.close()
    assert np.array_equal(st, rs), f"status differs at {np.nonzero(st != rs)[0][:10]}"
    assert np.array_equal(_bits(me), _bits(rm)), "minEig bits differ"
    bad = np.nonzero((_bits(out) != _bits(ro)).any(1))[0]
    assert bad.size == 0, f"{bad.size} points differ, first {bad[:5]}: {out[bad[:5]]} vs {ro[bad[:5]]}"
    assert np.array_equal(_bits(err), _bits(re)), "err bits differ"
    return out, st


def _colour_loop_frames(n):
    import torch

    poses = synth.loop_trajectory(n, **synth.BENCH_LOOP)
    lefts, rights = synth.stereo_torch(synth.bench_scene(colour=True), poses, device="cuda", batch=8)
    torch.cuda.synchronize()
    R0, t0 = poses[0]
    rel = [(R0.T @ R, R0.T @ (t - t0)) for R, t in poses]
    return rel, [(l.cpu().numpy(), r.cpu().numpy()) for l, r in zip(lefts, rights)], list(zip(lefts, rights))


def test_pyramid_and_derivative_levels_of_a_colour_image(ctx, orc):
    sc = synth.bench_scene(colour=True)
    R, t = synth.loop_trajectory(1, **synth.BENCH_LOOP)[0]
    img = sc.render(R, t)[0]
    assert _channels_differ(img)
    pyr = ctx.pyramid(1241, 376, 3).build(img)
    ref = img
    for l in range(4):
        assert np.array_equal(pyr.level(l), ref), f"level {l}"
        ref = orc.pyr_down(ref)
    pyr.close()


@pytest.mark.parametrize("shift", [(0.0, 0.0), (2.3, -1.4), (-9.6, 5.2)])
def test_lk_colour_texture_bit_exact(ctx, orc, shift):
    a, b = synth.textured_pair(320, 200, 3, shift=shift, seed=3, colour=True)
    assert _channels_differ(a)
    pts = orc.grid_keypoints(200, 320, 10)
    out, st = _compare_lk(ctx, orc, a, b, pts)
    inner = (pts[:, 0] > 45) & (pts[:, 0] < 275) & (pts[:, 1] > 45) & (pts[:, 1] < 155)
    assert np.abs((out - pts)[inner] - np.array(shift, np.float32)).max() < 0.12


def test_lk_colour_channel_order_matters(ctx, orc):
    """A tracker that mixed up channels between the two images would still pass on R = G = B.  Here the SECOND image's
    channels are rotated: GPU and oracle must agree on that input as well (bit for bit), and the result must differ from
    the aligned pair's -- the test images really distinguish the channels."""
    a, b = synth.textured_pair(320, 200, 3, shift=(1.5, 0.75), seed=5, colour=True)
    pts = orc.grid_keypoints(200, 320, 10)
    out0, st0 = _compare_lk(ctx, orc, a, b, pts)
    b_rot = np.ascontiguousarray(b[..., [1, 2, 0]])
    out1, st1 = _compare_lk(ctx, orc, a, b_rot, pts)
    assert (st0 != st1).any() or np.abs(out0 - out1)[st0 == 1].max() > 0.05


def test_lk_colour_full_size_bit_exact(ctx, orc):
    """1241x376x3, grid step 10 (4428 points): left -> right and t -> t+1 on the colour form of the benchmark's scene."""
    sc = synth.bench_scene(colour=True)
    poses = synth.loop_trajectory(2, **synth.BENCH_LOOP)
    l0, r0, depth = sc.stereo(*poses[0])
    l1 = sc.render(*poses[1])[0]
    assert _channels_differ(l0)
    pts = orc.grid_keypoints(376, 1241, 10)
    out, st = _compare_lk(ctx, orc, l0, r0, pts)
    assert st.sum() > 0.7 * len(pts)          # the sky and the far walls of the yard carry no texture
    z = depth[pts[:, 1].astype(int), pts[:, 0].astype(int)]
    ok = (st == 1) & (z > 0)
    disp = synth.KITTI_K[0] * synth.KITTI_BASELINE / z[ok]
    assert np.median(np.abs(-(out - pts)[ok, 0] - disp)) < 0.3
    out, st = _compare_lk(ctx, orc, l0, l1, pts)
    assert st.sum() > 0.6 * len(pts)


def test_lk_colour_edge_cases_bit_exact(ctx, orc):
    a, b = synth.textured_pair(200, 160, 3, shift=(1.0, 0.5), seed=9, colour=True)
    a[60:110, 70:130] = (100, 30, 220)
    b[60:110, 70:130] = (100, 30, 220)
    rng = np.random.default_rng(2)
    pts = np.concatenate([
        np.array([[100, 85], [30, 30], [-40, 50], [199, 159], [400, 80], [0, 0], [-10.5, -10.5],
                  [199.99, 0.01], [10.0, 159.5], [-11.0, 80.0], [210.0, 80.0], [100.0, 170.5]], np.float32),
        rng.uniform([-15, -15], [215, 175], (500, 2)).astype(np.float32)])
    _compare_lk(ctx, orc, a, b, pts)


def test_frontend_matches_oracle_on_the_colour_stream(ctx, orc):
    """The first 40 frames of the 200-frame front-end test on the colour form of the benchmark stream, 4096 keypoints:
    every tracked count, inlier count, keyframe decision, pose and the final reference sets equal to the oracle's,
    bit for bit; then the same frames as ONE pipelined chunk (four streams) and as a one-stream chunk."""
    nframes = 41
    poses, frames, dev = _colour_loop_frames(nframes)
    assert _channels_differ(frames[0][0])
    orc.set_num_threads(16)
    kw = dict(grid_step=10, anms_keep=4096, keyframe_min_inliers=2000, seed=20261003)
    g = capi.VisualOdometry(ctx, 1241, 376, 3, **kw)
    o = orc.VO(1241, 376, 3, **kw)
    assert g.init(*frames[0]) == o.init(*frames[0])
    ref = []
    n_kf = 0
    for i in range(1, nframes):
        rg, Rg, tg, ig, kg, ng = g.track(*frames[i])
        ro, Ro, to, io, ko, no = o.track(*frames[i])
        assert rg == 0 and ro == 0
        assert ng == no, f"frame {i}: tracked {ng} vs {no}"
        assert ig == io, f"frame {i}: inliers {ig} vs {io}"
        assert kg == ko, f"frame {i}: keyframe decision {kg} vs {ko}"
        assert np.array_equal(tg, to) and np.array_equal(Rg, Ro), f"frame {i}: {np.linalg.norm(tg - to):.2e} m"
        assert np.linalg.norm(tg - poses[i][1]) < 0.02 * i + 0.05
        n_kf += kg
        ref.append((Rg, tg, ig, kg, ng))
    a2, a3 = g.reference()
    b2, b3 = o.ref()
    assert np.array_equal(a2, b2) and np.array_equal(a3, b3)
    assert 0 < n_kf < nframes - 1
    col = g.keyframe_colors()
    assert len(col) == len(g.keyframe_cloud()) and len(col) > 1000
    for pipeline in (True, False):
        p = capi.VisualOdometry(ctx, 1241, 376, 3, **kw)
        p.init(*dev[0])
        rc, done, R, t, inl, trk, kf = p.run_chunk([d[0] for d in dev[1:]], [d[1] for d in dev[1:]], pipeline=pipeline)
        assert rc == 0 and done == nframes - 1
        for i, (R_i, t_i, inl_i, kf_i, trk_i) in enumerate(ref):
            assert np.array_equal(R[i], R_i) and np.array_equal(t[i], t_i), f"pipeline={pipeline}, frame {i + 1}"
            assert inl[i] == inl_i and trk[i] == trk_i and bool(kf[i]) == kf_i
        p2, p3 = p.reference()
        assert np.array_equal(p2, a2) and np.array_equal(p3, a3)
        assert np.array_equal(p.keyframe_colors(), col)
        p.close()
    g.close()
    o.close()


def test_keyframe_colours_are_bgr_of_the_left_image(ctx, orc):
    """getColors (include/monoUtils.h:180-193): img.at<Vec3b>(int(y), int(x)) as three floats in memory order
    (B, G, R) -- on images whose channels differ everywhere, at initialisation and at a forced keyframe."""
    poses, frames, dev = _colour_loop_frames(3)
    g = capi.VisualOdometry(ctx, 1241, 376, 3, grid_step=30, keyframe_min_inliers=200, seed=4)
    n0 = g.init(*dev[0])
    ref2d, _ = g.reference()
    col = g.keyframe_colors()
    want = orc.get_colors(frames[0][0], ref2d)
    assert col.shape == (n0, 3) and np.array_equal(col, want)
    xi, yi = ref2d[:, 0].astype(int), ref2d[:, 1].astype(int)
    assert np.array_equal(col, frames[0][0][yi, xi].astype(np.float32))       # the definition itself
    assert not np.array_equal(col, col[:, ::-1])                              # and the order is observable
    rc, R, t, inl, kf, trk = g.track(*dev[1], force_keyframe=True)
    assert rc == 0 and kf
    ref2d, _ = g.reference()
    assert np.array_equal(g.keyframe_colors(), orc.get_colors(frames[1][0], ref2d))
    # stand-alone entry point
    pts = np.array([[0.0, 0.0], [1240.9, 375.9], [607.2, 185.2], [33.7, 12.2]], np.float32)
    pyr = ctx.pyramid(1241, 376, 3).build(frames[2][0])
    assert np.array_equal(ctx.get_colors(pyr, pts), orc.get_colors(frames[2][0], pts))
    pyr.close()
    g.close()


@pytest.mark.parametrize("n_features,fast_t", [(500, 20), (2000, 10)])
def test_orb_grey_conversion_on_colour_images(ctx, orc, n_features, fast_t):
    """The loop detector's extractor converts BGR -> grey with OpenCV's fixed-point weights (B 1868, G 9617, R 4899
    of 2^14): on R = G = B images any weights summing to 2^14 give the same grey image.  On a colour image they do not."""
    sc = synth.bench_scene(colour=True)
    for R, t in synth.loop_trajectory(2, **synth.BENCH_LOOP):
        img = sc.render(R, t)[0]
        grey = orc.bgr_to_gray(img)
        i64 = img.astype(np.int64)
        assert np.array_equal(grey, ((1868 * i64[..., 0] + 9617 * i64[..., 1] + 4899 * i64[..., 2] + 8192) >> 14).astype(np.uint8))
        assert not np.array_equal(grey, orc.bgr_to_gray(np.ascontiguousarray(img[..., ::-1])))   # order matters here
        o = orc.orb_extract(img, n_features, fast_t)
        g = ctx.orb_extract(img, n_features, fast_t)
        assert len(g[0]) == len(o[0]) and len(o[0]) > 50
        for a, b, name in zip(g, o, ("xy", "octave", "response", "dir", "desc")):
            assert np.array_equal(a, b), name
