"""The C++ adaptors of include/svo_compat/ (the reference's visualSLAM / globalPoseGraph /
StereoProcess member surface on top of the C ABI) compile without OpenCV/Eigen/g2o/ROS and,
on a GPU box, run."""
import pathlib
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "cpp" / "compat_smoke.cpp"
EXE = ROOT / "tests" / "cpp" / "compat_smoke"


def _build():
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(SRC),
           f"-L{ROOT / 'ros_stereo_slam_amd'}", "-l:libsvo_hip.so",
           f"-Wl,-rpath,{ROOT / 'ros_stereo_slam_amd'}", "-o", str(EXE)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_compat_headers_compile_and_link():
    _build()
    assert EXE.exists()


def test_each_header_is_self_contained(tmp_path):
    for h in ("types.hpp", "poseGraph.hpp", "visualSLAM.hpp", "stereoCV.hpp"):
        tu = tmp_path / f"tu_{h}.cpp"
        tu.write_text(f'#include "svo_compat/{h}"\nint main() {{ return 0; }}\n')
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", f"-I{ROOT / 'include'}", str(tu)],
                       check=True, capture_output=True, text=True)


def test_c_header_is_plain_c(tmp_path):
    tu = tmp_path / "abi.c"
    tu.write_text('#include "svo.h"\nint main(void) { return svo_version() == SVO_VERSION ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", f"-I{ROOT / 'include'}", str(tu)],
                   check=True, capture_output=True, text=True)


@pytest.mark.gpu
def test_compat_smoke_runs_on_gpu():
    _build()
    out = subprocess.run([str(EXE)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "compat smoke ok" in out.stdout
