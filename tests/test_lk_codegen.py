"""The tracking kernel keeps hand-issued loads in flight in registers across arithmetic (lk.hip: tile_issue /
dtile_issue ... commit).  The compiler must not touch those registers in between.  csrc/Makefile checks the linked
library on every build (tools/check_lk_inflight.py --so) and refuses to install a library that violates it; this
test runs the same check on the SHIPPED libsvo_hip.so and makes sure the checker can still see a violation."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_lk_inflight.py")


def test_shipped_library_has_no_instruction_touching_a_register_with_a_load_in_flight():
    so = os.path.join(ROOT, "ros_stereo_slam_amd", "libsvo_hip.so")
    assert os.path.exists(so), "libsvo_hip.so is not built"
    r = subprocess.run([sys.executable, TOOL, "--so", so], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 problem(s)" in r.stdout and " 0 tracking kernel(s)" not in r.stdout


def test_the_checker_sees_a_violation(tmp_path):
    lst = tmp_path / "bad.s"
    lst.write_text("""
0000000000001200 <_ZN12_GLOBAL__N_115lk_track_kernelILi3EEEv7LkBatchNS_8LkParamsE>:
	global_load_dwordx4 v[24:27], v22, s[2:3]                  // 000000001628: DC5C8000 18020016
	v_add_u32_e32 v1, v25, v2                                  // touches a destination in flight
	s_waitcnt vmcnt(0)
	v_add_u32_e32 v1, v25, v2
	global_load_dwordx4 v[28:31], v2, s[2:3]
<L3>:
	s_waitcnt vmcnt(0)
	s_endpgm
""")
    r = subprocess.run([sys.executable, TOOL, str(lst)], capture_output=True, text=True)
    assert r.returncode == 1
    assert "touches v[25]" in r.stdout and "label at line" in r.stdout and " 2 problem(s)" in r.stdout


def test_the_makefile_runs_the_checker_with_the_library_it_links():
    mk = open(os.path.join(ROOT, "ros_stereo_slam_amd", "csrc", "Makefile")).read()
    assert "check_lk_inflight.py --so $@.tmp" in mk and "mv $@.tmp $@" in mk
