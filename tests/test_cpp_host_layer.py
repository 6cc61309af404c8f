"""The C++ host layer over the C ABI (include/sadvio_optimizer.hpp: the solve entry points of isae::AOptimizer on plain
structs). CPU: it compiles with g++ -std=c++17 against include/ and links the library; GPU: the C++ test program
(tests/cpp/test_optimizer.cpp) recovers perturbed local maps through localMapBA / landmarkOptimization /
singleFrameOptimization."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_optimizer.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "test_optimizer")


def build():
    import __graft_entry__ as g
    g.build_hip()
    lib_dir = os.path.join(ROOT, "sadvio_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-L", lib_dir, "-lsadvio_ba",
           "-Wl,-rpath," + lib_dir, "-o", BIN]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return BIN


def test_cpp_host_layer_compiles_and_links():
    assert os.path.exists(build())


@pytest.mark.gpu
def test_cpp_host_layer_recovers_perturbed_maps():
    r = subprocess.run([build()], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout


def test_camera_models_round_trip(tmp_path):
    """include/sadvio_cameras.hpp (getRayCamera / project of Camera, Fisheye x3, Omni, DoubleSphere): pure host code."""
    exe = str(tmp_path / "test_cameras")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "test_cameras.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout + r.stderr
