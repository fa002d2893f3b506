"""include/svo_math.h -- the one software implementation of sin / cos / acos / cbrt / log that the HIP library and
the oracle share.  Here: its accuracy against numpy's long double (it claims about 1 ulp, not
equality with any libm) and its behaviour at the edges the solvers reach.  tests/test_gpu_math.py checks that the
device evaluates it to the same bits."""
import numpy as np
import pytest

from oracle import orc

LD = np.longdouble


def ulp_error(got, ref_ld):
    ref64 = ref_ld.astype(np.float64)
    return np.abs(got.astype(LD) - ref_ld) / np.spacing(np.abs(ref64)).astype(LD)


def samples(kind, n=200_000, seed=5):
    rng = np.random.default_rng(seed)
    if kind == "angle":
        return np.concatenate([rng.uniform(-10, 10, n), rng.uniform(-1e-3, 1e-3, n // 10), rng.uniform(-400, 400, n // 10)])
    if kind == "unit":
        return np.concatenate([rng.uniform(-1, 1, n), 1 - 10 ** rng.uniform(-14, 0, n // 4), -1 + 10 ** rng.uniform(-14, 0, n // 4)])
    if kind == "positive":
        return np.concatenate([10 ** rng.uniform(-30, 30, n), rng.uniform(0.5, 2.0, n // 4), 1 - 10 ** rng.uniform(-9, -1, n // 4)])
    raise ValueError(kind)


@pytest.mark.parametrize("fn,kind,ref,bound", [
    ("sin", "angle", np.sin, 2.0), ("cos", "angle", np.cos, 2.0), ("acos", "unit", np.arccos, 1.0),
    ("cbrt", "positive", np.cbrt, 1.0), ("log", "positive", np.log, 1.0)])
def test_accuracy_against_long_double(fn, kind, ref, bound):
    x = samples(kind)
    if fn in ("sin", "cos"):
        # near a zero of the function the error is absolute (the reduction carries pi/2 to 1e-31 * k): bound it there
        r = ref(x.astype(LD))
        big = np.abs(r) > 1e-3
        err = ulp_error(orc.math_eval(fn, x), r)
        assert float(err[big].max()) <= bound
        assert float(np.abs(orc.math_eval(fn, x)[~big].astype(LD) - r[~big]).max()) < 1e-18
    else:
        assert float(ulp_error(orc.math_eval(fn, x), ref(x.astype(LD))).max()) <= bound


def test_edges():
    assert orc.math_eval("acos", [1.0, 1.5])[0] == 0.0 and orc.math_eval("acos", [1.5])[0] == 0.0
    assert abs(orc.math_eval("acos", [-1.0])[0] - np.pi) < 1e-15 and abs(orc.math_eval("acos", [0.0])[0] - np.pi / 2) < 1e-16
    assert orc.math_eval("sin", [0.0])[0] == 0.0 and orc.math_eval("cos", [0.0])[0] == 1.0
    assert orc.math_eval("cbrt", [0.0, 27.0, -8.0, 1e-300]).tolist()[:3] == [0.0, 3.0, -2.0]
    assert orc.math_eval("log", [1.0])[0] == 0.0 and abs(orc.math_eval("log", [np.e])[0] - 1) < 3e-16
    # odd / even symmetry holds exactly (the solvers rely on Rodrigues(-r) = Rodrigues(r)^T)
    x = samples("angle", 20000)
    assert np.array_equal(orc.math_eval("sin", -x), -orc.math_eval("sin", x))
    assert np.array_equal(orc.math_eval("cos", -x), orc.math_eval("cos", x))


def test_the_solvers_call_no_libm_transcendentals():
    """sin / cos / acos / cbrt / pow / log reach the device code and the oracle only through svo_math.h."""
    import pathlib
    import re

    root = pathlib.Path(__file__).resolve().parents[1]
    files = [root / "ros_stereo_slam_amd" / "csrc" / f for f in
             ("fransac.hip", "pnp.hip", "ba.hip", "geometry.hip", "posegraph.hip", "ransac_common.hip.h", "frontend.hip")]
    files += [root / "oracle" / f for f in ("geometry.c", "pnp.c", "ba.c", "posegraph.c", "vo.c")]
    pat = re.compile(r"(?<![\w.])(sin|cos|acos|asin|atan2?|cbrt|pow|log|exp|sincos)\s*\(")
    for f in files:
        code = re.sub(r"/\*.*?\*/", "", f.read_text(), flags=re.S)
        code = "\n".join(ln.split("//")[0] for ln in code.splitlines())
        assert not pat.findall(code), f"{f.name} calls libm: {pat.findall(code)}"
