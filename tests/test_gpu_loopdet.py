"""GPU loop-closure detector (svo_lc, src/optimizationStuff.cpp:49-64) against the oracle's
line-by-line restatement of DLoopDetector::detectLoop on the same synthetic loop: the status and
the matched entry of every frame, and the detection visualSLAM accepts (query - match > 100)."""
import numpy as np
import pytest

from oracle.loop_detector import LoopDetector as OracleDetector, Params
from ros_stereo_slam_amd import capi, synth

pytestmark = pytest.mark.gpu
SIZE, K4 = (480, 160), (270.0, 270.0, 240.0, 80.0)


def _loop_images(n=134):
    poses = synth.loop_trajectory(n, half_x=6, half_z=10, radius=4, step=0.5)
    sc = synth.Scene(wall_x=14, z_min=-18, z_max=18)
    return poses, [sc.stereo(R, t, K=K4, size=SIZE)[0] for R, t in poses]


@pytest.mark.parametrize("alpha", [0.9, 0.3])
def test_detector_matches_oracle_frame_by_frame(ctx, alpha):
    poses, imgs = _loop_images()
    g = capi.LoopDetector(ctx, SIZE[0], SIZE[1], 3, alpha=alpha, seed=5)
    o = OracleDetector(Params(alpha=alpha, seed=5))
    accepted_g, accepted_o = [], []
    for i, img in enumerate(imgs):
        rg, ro = g.detect(img), o.detect(img)
        assert rg["query"] == ro["query"] == i
        assert rg["status"] == ro["status"], (i, capi.LC_STATUS[rg["status"]], capi.LC_STATUS[ro["status"]])
        assert rg["match"] == ro["match"], i
        for r, acc in ((rg, accepted_g), (ro, accepted_o)):
            if r["status"] == 0 and r["query"] - r["match"] > 100:      # src/optimizationStuff.cpp:58
                acc.append((r["query"], r["match"]))
    assert len(g) == len(imgs)
    assert accepted_g == accepted_o and accepted_g
    # the first accepted detection is the true revisit of the start of the loop
    gt = synth.loop_closures(poses, min_gap=100)
    first_true = next(i for i, m in enumerate(gt) if m >= 0)
    q, m = accepted_g[0]
    assert abs(q - first_true) <= 8 and m <= 8
    g.close()


def test_device_images_and_capacity(ctx):
    import torch
    _, imgs = _loop_images(30)
    a = capi.LoopDetector(ctx, SIZE[0], SIZE[1], 3, max_entries=26)
    b = capi.LoopDetector(ctx, SIZE[0], SIZE[1], 3, max_entries=26)
    for i, img in enumerate(imgs[:26]):
        d = torch.from_numpy(img).cuda()
        torch.cuda.synchronize()
        assert a.detect(img) == b.detect(d)
    with pytest.raises(capi.SvoError):
        a.detect(imgs[26])                                       # database full
    a.close()
    b.close()


@pytest.mark.parametrize("max_db_results", [50, 4])
def test_queued_detector_on_a_context_of_its_own_equals_the_synchronous_one(ctx, max_db_results):
    """svo_lc_submit / svo_lc_collect: every frame queued on the detector's own context before anything is collected
    (the scores reduced on the device to the max_db_results candidates the host logic reads -- with 4 the cut falls
    inside runs of equal scores), then frames two deep in flight: the verdicts of the synchronous calls, and the
    oracle's."""
    import torch
    poses, imgs = _loop_images()
    dev = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    own = capi.Context(0)
    a = capi.LoopDetector(ctx, SIZE[0], SIZE[1], 3, seed=5, max_db_results=max_db_results)
    b = capi.LoopDetector(own, SIZE[0], SIZE[1], 3, seed=5, max_db_results=max_db_results)
    c = capi.LoopDetector(own, SIZE[0], SIZE[1], 3, seed=5, max_db_results=max_db_results)
    o = OracleDetector(Params(seed=5, max_db_results=max_db_results))
    ref = [a.detect(d) for d in dev]
    for i, (r, img) in enumerate(zip(ref, imgs)):
        ro = o.detect(img)
        assert (r["status"], r["match"]) == (ro["status"], ro["match"]), i
    for d in dev:
        b.submit(d)
    assert b.pending() == len(dev) and len(b) == len(dev)
    assert [b.collect() for _ in dev] == ref
    assert b.pending() == 0
    got = []
    c.submit(dev[0])
    for d in dev[1:]:
        c.submit(d)                      # frame i + 1 is queued ...
        got.append(c.collect())          # ... before frame i is collected
    got.append(c.collect())
    assert got == ref
    with pytest.raises(capi.SvoError):
        c.collect()                      # nothing queued
    c.submit(dev[0])
    with pytest.raises(capi.SvoError):
        c.detect(dev[1])                 # the synchronous call refuses to overtake a queued frame
    assert any(r["status"] == 0 for r in ref)
    for x in (a, b, c):
        x.close()
    own.close()
