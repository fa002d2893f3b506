"""The tracking kernel keeps hand-issued loads in flight in registers across arithmetic (lk.hip: tile_issue /
dtile_issue ... commit).  The compiler must not touch those registers in between; this reads the generated
gfx950 assembly and checks that it did not (tools/check_lk_inflight.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_no_instruction_touches_a_register_with_a_load_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lk_inflight.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 problem(s)" in r.stdout
