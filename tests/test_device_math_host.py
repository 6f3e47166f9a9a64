"""The per-element math of the HIP kernels (csrc/device_math.hpp) compiled for the HOST, with the gfx950 hardware
approximations (v_rcp / v_rsq / v_sqrt, 2^-24; measured by experiments/hw_prec.hip) emulated with random errors of that size:
the numerics of the mixed-precision signed SVD and of the stretch minimisation can be checked without a GPU.  tests/hostmath/
holds the stand-in for <hip/hip_runtime.h> (macros and the handful of builtins the header uses) -- it is test scaffolding for
OUR device code, nothing of the reference is involved.  The same functions run on the GPU in tests/test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hostmath") / "libhostmath.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(HERE, "hostmath"), "-o", out,
                           os.path.join(HERE, "hostmath", "hostmath.cpp")])
    return C.CDLL(out)


def _svd(L, F):
    n = len(F)
    Fc = np.ascontiguousarray(np.transpose(F, (0, 2, 1)).reshape(n, 9))      # column-major
    U = np.zeros((n, 9)); S = np.zeros((n, 3)); V = np.zeros((n, 9))
    cnt = np.zeros((n, 4), np.int32); rec = np.zeros((n, 32))
    L.hm_svd_counted(n, Fc.ctypes.data_as(dp), U.ctypes.data_as(dp), S.ctypes.data_as(dp), V.ctypes.data_as(dp),
                     cnt.ctypes.data_as(C.POINTER(C.c_int)), rec.ctypes.data_as(dp))
    return np.transpose(U.reshape(n, 3, 3), (0, 2, 1)), S, np.transpose(V.reshape(n, 3, 3), (0, 2, 1)), cnt


def _rot(rng, n):
    q, _ = np.linalg.qr(rng.standard_normal((n, 3, 3)))
    return q * np.sign(np.linalg.det(q))[:, None, None]


def _diag(rng, n, d):
    return _rot(rng, n) @ (d[:, :, None] * np.eye(3)) @ np.transpose(_rot(rng, n), (0, 2, 1))


def _cases():
    rng = np.random.default_rng(0)
    n = 4000
    sets = {"random": rng.standard_normal((n, 3, 3))}
    for e in (1e-1, 1e-2, 1e-3, 1e-5, 1e-8, 1e-12):      # elements near rest at every strain scale
        sets["rest+%g" % e] = _rot(rng, n) @ (np.eye(3) + e * rng.standard_normal((n, 3, 3)))
    sets["rest"] = np.repeat(np.eye(3)[None], 64, 0)
    sets["rotation"] = _rot(rng, n)
    d = np.ones((n, 3)); d[:, 0] = 1.0 + rng.uniform(0, 1, n); d[:, 2] = d[:, 1] * (1.0 + 1e-9 * rng.standard_normal(n))
    sets["two equal"] = _diag(rng, n, d)
    d = rng.uniform(0.2, 2.0, (n, 3)); d[:, 2] *= -1
    sets["inverted"] = _diag(rng, n, d)
    for e in (0.0, 1e-6, 1e-10, 1e-13, 1e-15):            # flat / nearly flat, the thin direction in every column position
        for col in (0, 1, 2):
            d = rng.uniform(0.2, 2.0, (n // 4, 3)); d[:, col] = e * rng.uniform(-1, 1, n // 4)
            sets["thin %g @%d" % (e, col)] = _diag(rng, n // 4, d)
    d = rng.uniform(0.2, 2.0, (n, 3)); d[:, 1:] = 0
    sets["rank one"] = _diag(rng, n, d)
    sets["zero"] = np.zeros((8, 3, 3))
    sets["tiny"] = 1e-60 * rng.standard_normal((n, 3, 3))
    sets["huge"] = 1e60 * rng.standard_normal((n, 3, 3))
    sets["stretched 1e4"] = _diag(rng, n, np.tile(np.array([1e4, 1.0, 1e-3]), (n, 1)))
    return sets


def test_signed_svd_is_a_factorisation_to_3e14_on_every_kind_of_element(hm):
    worst = {}
    for name, F in _cases().items():
        U, S, V, cnt = _svd(hm, F)
        nF = np.linalg.norm(F, axis=(1, 2)) + 1e-300
        rec = np.linalg.norm(U @ (S[:, :, None] * np.transpose(V, (0, 2, 1))) - F, axis=(1, 2)) / nF
        ou = np.linalg.norm(np.transpose(U, (0, 2, 1)) @ U - np.eye(3), axis=(1, 2))
        ov = np.linalg.norm(np.transpose(V, (0, 2, 1)) @ V - np.eye(3), axis=(1, 2))
        assert rec.max() < 5e-14, (name, rec.max())                      # the documented bound (ADMM_SVD_TOL2 = 1e-27)
        assert ou.max() < 1e-14 and ov.max() < 1e-14, (name, ou.max(), ov.max())
        assert np.linalg.det(U).min() > 0.999 and np.linalg.det(V).min() > 0.999, name      # rotations, not reflections
        sv = np.linalg.svd(F, compute_uv=False)
        mine = np.sort(np.abs(S), axis=1)[:, ::-1]
        assert (np.abs(mine - sv).max(axis=1) / nF).max() < 1e-14, name
        # the reference's sign convention (src/FastSVD.hpp:43-68): at most one negative stretch, and it is the smallest one
        big = np.abs(S) > 1e-12 * nF[:, None]                            # (round-off sized stretches have no sign to speak of)
        neg = (S < 0) & big
        assert (neg.sum(axis=1) <= 1).all(), name
        rows = neg.any(axis=1)
        assert (np.abs(S)[neg] <= np.abs(S).min(axis=1)[rows] * (1 + 1e-9)).all(), name
        if name in ("random", "inverted", "rest+0.1"):
            assert (np.sign(np.prod(S, axis=1)) == np.sign(np.linalg.det(F))).all(), name
        worst[name] = cnt[:len(F) // 64 * 64].reshape(-1, 64, 4).max(axis=1).mean(axis=0) if len(F) >= 64 else None
    # cost model of the kernel: a wave (64 elements) runs as long as its slowest lane -- moderate strains take the fast path:
    # three or four FP32 seed sweeps and ONE FP64 sweep
    for name in ("rest+0.01", "rest+0.001", "rest+1e-05", "rest+1e-08", "rest+1e-12"):
        assert worst[name][0] <= 3.5 and worst[name][1] <= 1.05, (name, worst[name])


@pytest.mark.parametrize("kind", [1, 2])
def test_stretch_minimisation_meets_the_oracles_optimality_condition(hm, kind):
    """prox_stretches (NH = 1, StVK = 2): at the returned stretches the gradient of the ORACLE's objective
    Psi(s) + k/2 |s - x0|^2 (oracle/admm_oracle.c: prox_gradient, src/TetEnergyTerm.cpp:173-237) vanishes -- in the interior;
    StVK components on the s = 0 boundary carry an outward gradient."""
    rng = np.random.default_rng(kind)
    mu, la, _ = orc.lame(1e6, 0.3)
    L = orc.lib()
    hm.hm_prox.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, dp]
    worst = 0.0
    for k in (0.1 * mu, mu, 30.0 * mu):
        n = 3000
        x0 = np.concatenate([1.0 + 0.05 * rng.standard_normal((n, 3)), rng.uniform(0.3, 2.5, (n, 3)),
                             np.abs(1.0 + 0.3 * rng.standard_normal((n, 3))) * np.array([1, 1, -1.0]),
                             rng.uniform(0.15, 6.0, (n, 3)),                                                   # extreme: det up to 200
                             np.array([[4.91949492, 2.87407946, 2.38054277]])])      # near rest, large strain, inverted, extreme,
        # and the element that exposed the convexified Hessian model (one negative diagonal entry, Hessian still convex): the
        # Newton then crawled to its iteration cap 5e-5 short of the minimiser
        s = np.ascontiguousarray(x0.copy())
        hm.hm_prox(kind, len(s), mu, la, k, s.ctypes.data_as(dp))
        assert np.isfinite(s).all()
        for i in range(0, len(s), 7):
            g = np.zeros(3)
            L.orc_prox_gradient(kind, mu, la, k, np.ascontiguousarray(x0[i]).ctypes.data_as(dp), np.ascontiguousarray(s[i]).ctypes.data_as(dp), g.ctypes.data_as(dp))
            scale = (mu + la + k) * max(1.0, np.abs(s[i]).max())
            if kind == 1:
                assert (s[i] > 0).all()
                worst = max(worst, np.abs(g).max() / scale)
            else:
                assert (s[i] >= 0).all()
                interior = s[i] > 0
                worst = max(worst, np.abs(g[interior]).max() / scale if interior.any() else 0.0)
                assert (g[~interior] >= -1e-7 * scale).all()
    assert worst < 1e-8, worst
