"""The device evaluates include/svo_math.h to the SAME BITS as the oracle's host build: the precondition for the
front-end being decision-identical to the oracle over a whole sequence (tests/test_gpu_frontend.py)."""
import numpy as np
import pytest

from oracle import orc
from test_svo_math import samples

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fn,kind", [("sin", "angle"), ("cos", "angle"), ("acos", "unit"), ("cbrt", "positive"),
                                     ("log", "positive")])
def test_device_bits_equal_host_bits(ctx, fn, kind):
    x = samples(kind, n=300_000, seed=11)
    g, o = ctx.math_eval(fn, x), orc.math_eval(fn, x)
    assert np.array_equal(g.view(np.uint64), o.view(np.uint64)), f"{fn}: {(g != o).sum()} of {x.size} differ"
