"""The C-ABI library loads on a CPU-only box and exports every symbol include/svo.h declares."""
import ctypes

import pytest

from ros_stereo_slam_amd import capi


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = capi.declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/svo.h but not exported: {missing}"


def test_version():
    assert capi.load().svo_version() == 100


def test_no_cpu_fallback_without_device():
    """Without a GPU the context cannot be created -- there is no CPU path to fall back to."""
    lib = capi.load()
    h = ctypes.c_void_p()
    rc = lib.svo_ctx_create(0, ctypes.byref(h))
    if rc == 0:  # a GPU box: fine, clean up
        lib.svo_ctx_destroy(h)
        pytest.skip("a device is present")
    assert rc == capi.SVO_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.svo_last_error() or b"gfx950" in lib.svo_last_error()
    with pytest.raises(capi.SvoError):
        capi.Context(0)


def test_product_does_not_import_oracle():
    """Nothing under ros_stereo_slam_amd/ may reference the oracle (it is test infrastructure)."""
    import pathlib

    root = pathlib.Path(capi.__file__).resolve().parent
    for p in list(root.rglob("*.py")) + list(root.rglob("*.hip")) + list(root.rglob("*.h")) + list(
            root.rglob("*.cpp")):
        text = p.read_text(errors="replace")
        assert "svo_oracle" not in text and "from oracle" not in text and "import oracle" not in text, p
