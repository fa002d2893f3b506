"""GPU feature extractor of the loop detector (svo_orb_extract) against the oracle on the same
images: keypoints, responses, orientation vectors and 256-bit descriptors bit for bit."""
import numpy as np
import pytest

from ros_stereo_slam_amd import synth

pytestmark = pytest.mark.gpu


def _images():
    sc = synth.Scene()
    poses = synth.corridor_trajectory(3)
    return [sc.stereo(R, t)[0] for R, t in poses]


@pytest.mark.parametrize("n_features,fast_t", [(500, 20), (120, 35), (2000, 10)])
def test_features_match_oracle_bit_for_bit(ctx, orc, n_features, fast_t):
    for img in _images()[:2]:
        o = orc.orb_extract(img, n_features, fast_t)
        g = ctx.orb_extract(img, n_features, fast_t)
        assert len(g[0]) == len(o[0]) and len(o[0]) > 50
        for a, b, name in zip(g, o, ("xy", "octave", "response", "dir", "desc")):
            assert np.array_equal(a, b), name


def test_grey_input_small_image_and_flat_image(ctx, orc):
    img = _images()[2]
    grey = orc.bgr_to_gray(img)[40:260, 100:500]          # 400 x 220, single channel
    o = orc.orb_extract(grey, 300)
    g = ctx.orb_extract(grey, 300)
    assert len(o[0]) > 20
    for a, b in zip(g, o):
        assert np.array_equal(a, b)
    flat = np.full((200, 300, 3), 77, np.uint8)             # nothing to detect
    assert len(ctx.orb_extract(flat, 500)[0]) == 0 and len(orc.orb_extract(flat, 500)[0]) == 0
