"""Builds libadmm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["csrc/admm_hip.hip", "csrc/host_setup.cpp", "csrc/oc_plan.cpp"]
HEADERS = sorted("csrc/" + f for f in os.listdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")) if f.endswith(".hpp")) + ["../include/admm_hip.h"]
OUT = os.path.join(HERE, "libadmm_hip.so")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> admm-elastic_amd/libadmm_hip.so.  Returns the path."""
    if not force and not needs_build():
        return OUT
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result"] + os.environ.get("ADMM_HIP_EXTRA_FLAGS", "").split() \
        + [os.path.join(HERE, s) for s in SOURCES] + ["-o", OUT]   # extra flags: kernel experiments only
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


OUT_HOST = os.path.join(HERE, "libadmm_elastic.so")
HOST_SOURCES = ["host/src/Solver.cpp"]


def build_host_library(force=False, verbose=False):
    """g++ -> admm-elastic_amd/libadmm_elastic.so: the C++ mirror of the reference's class API
    (admm::Solver, EnergyTerm, ...) on top of libadmm_hip.so."""
    build_library(force=force, verbose=verbose)
    import glob
    deps = [os.path.join(HERE, f) for f in HOST_SOURCES] + glob.glob(os.path.join(HERE, "host", "include", "*.hpp")) + [OUT]
    if not force and os.path.exists(OUT_HOST) and all(os.path.getmtime(d) <= os.path.getmtime(OUT_HOST) for d in deps):
        return OUT_HOST
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + os.path.join(HERE, "host", "include")] + \
          [os.path.join(HERE, f) for f in HOST_SOURCES] + ["-L" + HERE, "-ladmm_hip", "-Wl,-rpath,$ORIGIN", "-o", OUT_HOST]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT_HOST
