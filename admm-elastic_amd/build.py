"""Builds libadmm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

A rebuild is keyed on a HASH of the sources (+ compiler flags) stored beside the library (libadmm_hip.so.srchash), not on
file times: a binary that does not belong to the sources in the tree -- a leftover experiment, a checkout with fresh
mtimes -- is never used silently.  A/B variants of the library live OUTSIDE the package (ADMM_HIP_LIB, capi.py)."""
import hashlib
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


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"] \
        + os.environ.get("ADMM_HIP_EXTRA_FLAGS", "").split()   # extra flags: kernel experiments only


def source_hash(files=None, extra=()):
    """sha256 over the given files (default: every source and header of the HIP library) and the compiler flags."""
    h = hashlib.sha256()
    for f in (files if files is not None else SOURCES + HEADERS):
        h.update(f.encode()); h.update(b"\0")
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    for x in (extra if files is not None else _flags()):
        h.update(x.encode()); h.update(b"\0")
    return h.hexdigest()


def _stored_hash(out):
    try:
        with open(out + ".srchash") as fh:
            return fh.read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(OUT) or _stored_hash(OUT) != source_hash()


def build_library(force=False, verbose=False, out=None):
    """hipcc --offload-arch=gfx950 -> admm-elastic_amd/libadmm_hip.so.  Returns the path.
    `out`: build a VARIANT (ADMM_HIP_EXTRA_FLAGS) somewhere else, for ADMM_HIP_LIB -- the in-tree library is not touched."""
    if out is None and not force and not needs_build():
        return OUT
    target = out or OUT
    cmd = [_hipcc()] + _flags() + [os.path.join(HERE, s) for s in SOURCES] + ["-o", target]
    if verbose:
        print(" ".join(cmd))
    h = source_hash()
    subprocess.run(cmd, check=True)
    with open(target + ".srchash", "w") as fh:
        fh.write(h + "\n")
    return target


OUT_HOST = os.path.join(HERE, "libadmm_elastic.so")
HOST_SOURCES = ["host/src/Solver.cpp"]


def build_host_library(force=False, verbose=False):
    """g++ -> admm-elastic_amd/libadmm_elastic.so: the C++ mirror of the reference's class API
    (admm::Solver, EnergyTerm, ...) on top of libadmm_hip.so."""
    build_library(force=force, verbose=verbose)
    import glob
    deps = HOST_SOURCES + sorted(os.path.relpath(f, HERE) for f in glob.glob(os.path.join(HERE, "host", "include", "*.hpp"))) + ["../include/admm_hip.h"]
    h = source_hash(deps, extra=(_stored_hash(OUT) or "",))     # (relinked when the HIP library it links against changes)
    if not force and os.path.exists(OUT_HOST) and _stored_hash(OUT_HOST) == h:
        return OUT_HOST
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + os.path.join(HERE, "host", "include")] + \
          [os.path.join(HERE, f) for f in HOST_SOURCES] + ["-L" + HERE, "-ladmm_hip", "-Wl,-rpath,$ORIGIN", "-o", OUT_HOST]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(OUT_HOST + ".srchash", "w") as fh:
        fh.write(h + "\n")
    return OUT_HOST
