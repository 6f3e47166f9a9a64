"""The C++ mirror of the reference's class API (admm-elastic_amd/host): builds on CPU; on the GPU the
re-stated reference test (tests/cpp/test_lineartet.cpp) must print SUCCESS."""
import os
import subprocess

import pytest

import admm_elastic_amd as pkg
from admm_elastic_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_lineartet")


def _build_exe():
    build.build_host_library()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_lineartet.cpp")
    pk = os.path.join(ROOT, "admm-elastic_amd")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(build.OUT_HOST)):
        subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(pk, "host", "include"), src, "-L" + pk, "-ladmm_elastic",
                        "-ladmm_hip", "-Wl,-rpath," + pk, "-o", EXE], check=True)
    return EXE


def test_cpp_api_builds_and_fails_loudly_without_gpu():
    exe = _build_exe()
    if pkg.device_count() > 0:
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_lineartet_known_answers():
    exe = _build_exe()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SUCCESS" in r.stdout, r.stdout + r.stderr
