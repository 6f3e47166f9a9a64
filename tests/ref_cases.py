"""Seeded input cases and the outputs of the REAL reference code for them.

Every case has (i) a deterministic input builder and (ii) a function that runs the reference's own code on it through
oracle/_ref/libadmm_ref.so (the reference sources compiled in place by oracle/Makefile; only possible where
/root/reference exists).  tests/golden/make_golden.py stores those outputs in tests/golden/ref_vectors.npz;
`ref_out(name)` returns them LIVE when the library is present (and checks them against the stored file, so a stale file
is caught in the build container) and FROM THE FILE otherwise -- a clean clone, or the GPU box, pins the oracle and the
HIP path on exactly the same reference numbers."""
import ctypes as C
import os

import numpy as np

from oracle import oracle as orc
from admm_elastic_amd import meshes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz")


def _p(a):
    return a.ctypes.data_as(orc.dp)


def _i(a):
    return a.ctypes.data_as(orc.ip)


# ---- inputs ---------------------------------------------------------------------------------------------------------
def svd_cases():
    rng = np.random.default_rng(7)
    cases = [rng.standard_normal((3, 3)) for _ in range(40)]
    cases += [np.eye(3), np.diag([2.0, 1.0, 0.5]), -np.eye(3), np.diag([1.0, 1.0, -1.0]), np.diag([3.0, 3.0, 1.0]),
              np.diag([1.0, 1e-9, 1e-9]), np.zeros((3, 3)), 1e-8 * rng.standard_normal((3, 3))]
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    cases += [q, -q, q @ np.diag([1.5, 1.5, 0.2]), q @ np.diag([1.0, 0.7, -0.3]) @ q.T]
    return cases


TRI_LIMITS = [(0.95, 1.05), (-100.0, 100.0), (0.5, 1.01)]


def tri_case(m=6, seed=3, limits=(0.95, 1.05)):
    verts, tris = meshes.cloth_grid(m, 1.0, 0.5)
    rng = np.random.default_rng(seed)
    verts = verts + 0.02 * rng.standard_normal(verts.shape)       # non-trivial rest shapes
    x = verts + 0.1 * rng.standard_normal(verts.shape)
    u = 0.05 * rng.standard_normal(6 * len(tris))
    mu, la, _ = orc.lame(100.0, 0.1)
    return verts, tris, x, u, mu, la, limits


def pin_case():
    rng = np.random.default_rng(5)
    nv = 10
    x = rng.standard_normal(3 * nv)
    vidx = np.array([2, 7, 4], np.int32); pins = rng.standard_normal((3, 3)); act = np.array([1, 0, 1], np.int32)
    u = np.zeros(18); u[[0, 1, 2, 6, 7, 8, 12, 13, 14]] = rng.standard_normal(9)
    return nv, x, vidx, pins, act, u


FLOOR_CW = (1.0, 37.5)


def floor_case():
    return np.random.default_rng(9).standard_normal((50, 3)), -0.2


def ldlt_case():
    verts, tets = meshes.kuhn_cube(3)
    mu, la, _ = orc.lame(1e7, 0.399)
    m = np.repeat(meshes.lumped_masses_tets(verts, tets), 3)
    o = orc.OracleSolver(verts, m, tets=dict(idx=tets, verts=verts, kind=1, mu=mu, la=la))
    A = o.A.tocsr(); A.sort_indices()
    b = np.random.default_rng(1).standard_normal(A.shape[0])
    return o, A, b


SPLINE_KAPPAS = (0.0, 2.5e5)
SPLINE_X = np.concatenate([np.linspace(0.05, 0.95, 7), np.linspace(1.0, 2.6, 9)])
LAME_CASES = ((1e7, 0.499), (1e7, 0.399), (1e6, 0.299), (100.0, 0.1))


# ---- the reference's outputs ------------------------------------------------------------------------------------------
def _ref_svd(L):
    S, U, V = [], [], []
    for F in svd_cases():
        a = np.ascontiguousarray(F.T).copy()
        u = np.zeros(9); s = np.zeros(3); v = np.zeros(9)
        L.ref_signed_svd(_p(a), _p(s), _p(u), _p(v))
        S.append(s); U.append(u.reshape(3, 3).T.copy()); V.append(v.reshape(3, 3).T.copy())
    return dict(F=np.array(svd_cases()), S=np.array(S), U=np.array(U), V=np.array(V))


def _ref_tri(L, limits):
    verts, tris, x, u, mu, la, _ = tri_case(limits=limits)
    n, nv = len(tris), len(verts)
    z = np.zeros(6 * n); uu = u.copy(); w = np.zeros(n)
    tr, tc, tv = np.zeros(18 * n, np.int32), np.zeros(18 * n, np.int32), np.zeros(18 * n)
    nnz = L.ref_tri_local_step(n, _i(np.ascontiguousarray(tris)), nv, _p(np.ascontiguousarray(verts)), mu, la, limits[0], limits[1],
                               _p(np.ascontiguousarray(x)), _p(z), _p(uu), _p(w), _i(tr), _i(tc), _p(tv))
    return dict(z=z, u=uu, w=w, nnz=np.array([nnz]), D_row=tr, D_col=tc, D_val=tv)


def _ref_pin(L):
    nv, x, vidx, pins, act, u = pin_case()
    z = np.zeros(18); uu = u.copy()
    L.ref_pin_local_step.restype = C.c_double
    w = L.ref_pin_local_step(3, _i(vidx), _p(pins), _i(act), nv, _p(x), _p(z), _p(uu))
    for blk in range(3):      # rows 3..5 of a SpringPin block are never populated by the reference: not data
        z[6 * blk + 3:6 * blk + 6] = 0.0; uu[6 * blk + 3:6 * blk + 6] = 0.0
    return dict(z=z, u=uu, w=np.array([w]))


def _ref_floor(L, cw):
    x, y0 = floor_case()
    nv = len(x)
    rv = np.zeros(nv, np.int32); rc = np.zeros(nv); coef = np.zeros(3 * nv)
    rows = L.ref_floor_constraints(nv, _p(np.ascontiguousarray(x)), C.c_double(y0), C.c_double(cw), nv, _i(rv), _p(rc), _p(coef))
    # the reference's hit order is thread order: store sorted by vertex
    order = np.argsort(rv[:rows], kind="stable")
    return dict(rows=np.array([rows]), vert=rv[:rows][order], c=rc[:rows][order], coef=coef.reshape(-1, 3)[:rows][order])


def _ref_ldlt(L):
    o, A, b = ldlt_case()
    x = np.zeros_like(b)
    rc = L.ref_ldlt_solve(A.shape[0], _i(A.indptr.astype(np.int32)), _i(A.indices.astype(np.int32)), _p(A.data), 1, _p(b), _p(x))
    assert rc == 0
    return dict(x=x)


def _ref_spline(L, which, kappa):
    mu, la, _ = orc.lame(1e6, 0.3)
    out = np.zeros(6)
    rows = []
    for x in SPLINE_X:
        L.ref_xu_spline(which, C.c_double(mu), C.c_double(la), C.c_double(kappa), C.c_double(float(x)), _p(out))
        rows.append(out.copy())
    return dict(fgh=np.array(rows))      # [len(SPLINE_X)][f, g, h, df, dg, dh]


def _ref_lame(L):
    mu, la, k = C.c_double(), C.c_double(), C.c_double()
    rows = []
    for E, nu in LAME_CASES:
        L.ref_lame(C.c_double(E), C.c_double(nu), C.byref(mu), C.byref(la), C.byref(k))
        rows.append([mu.value, la.value, k.value])
    return dict(mlk=np.array(rows))


CASES = {"svd": _ref_svd, "pin": _ref_pin, "ldlt": _ref_ldlt, "lame": _ref_lame}
for _k, _lim in enumerate(TRI_LIMITS):
    CASES["tri%d" % _k] = (lambda L, lim=_lim: _ref_tri(L, lim))
for _k, _cw in enumerate(FLOOR_CW):
    CASES["floor%d" % _k] = (lambda L, cw=_cw: _ref_floor(L, cw))
for _w in range(3):
    for _k, _kap in enumerate(SPLINE_KAPPAS):
        CASES["spline%d_%d" % (_w, _k)] = (lambda L, w=_w, kap=_kap: _ref_spline(L, w, kap))

_gold = None
_live = {}


def gold():
    global _gold
    if _gold is None:
        _gold = np.load(GOLD)
    return _gold


def have_live_reference():
    return orc.ref_lib() is not None


def ref_out(name):
    """Outputs of the reference for case `name`: live (checked against the stored file) or from the stored file."""
    if name in _live:
        return _live[name]
    g = gold()
    stored = {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "/")}
    L = orc.ref_lib()
    if L is None:
        assert stored, "golden vectors for %r missing from %s (run tests/golden/make_golden.py where /root/reference exists)" % (name, GOLD)
        _live[name] = stored
        return stored
    out = CASES[name](L)
    assert stored, "tests/golden/ref_vectors.npz is stale: no %r (run tests/golden/make_golden.py)" % name
    for k, v in out.items():
        assert np.allclose(np.asarray(v, dtype=float), np.asarray(stored[k], dtype=float), rtol=1e-12, atol=1e-13), \
            "stored golden %s/%s differs from the live reference" % (name, k)
    _live[name] = out
    return out
