import os
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure only)."""
    from oracle import orc as _orc

    _orc.load()
    return _orc


@pytest.fixture(scope="session")
def ctx():
    """A libsvo_hip context on device 0.  GPU tests only; fails loudly without the library."""
    # torch ships its own HIP runtime: let it load first (as bench.py does) so that one copy
    # serves both torch and libsvo_hip in this process
    import torch

    torch.cuda.is_available()
    from ros_stereo_slam_amd import capi

    c = capi.Context(0)
    yield c
    c.close()
