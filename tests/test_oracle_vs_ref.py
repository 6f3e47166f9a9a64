"""Pins the C restatement (oracle/admm_oracle.c) against the REAL reference: FastSVD.hpp signed_svd, TriEnergyTerm.cpp,
SpringEnergyTerm.hpp, EnergyTerm::update, ConstraintSet::make_matrix, Collider::detect, XuSpline.hpp,
Eigen::SimplicialLDLT, Lame.  The reference's outputs come from tests/ref_cases.py: LIVE from the reference sources
compiled in place (oracle/_ref/libadmm_ref.so) where /root/reference exists -- and then also checked against the stored
file -- and from tests/golden/ref_vectors.npz (tests/golden/make_golden.py) everywhere else, so a clean clone pins
exactly as much as the build container."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc
import ref_cases as R
from ref_cases import _p, _i, ref_out, svd_cases, tri_case  # noqa: F401


def test_golden_file_is_complete_and_current():
    """Every case has stored vectors; where the live reference exists they agree with it (ref_out asserts that)."""
    stored = set(k.split("/")[0] for k in R.gold().files)
    assert stored == set(R.CASES), stored ^ set(R.CASES)
    for name in R.CASES:
        assert ref_out(name)


def test_signed_svd_matches_reference():
    g = ref_out("svd")
    for F, Sr, Ur, Vr in zip(g["F"], g["S"], g["U"], g["V"]):
        U, S, V = orc.signed_svd3(F)
        scale = max(1.0, np.abs(F).max())
        assert np.allclose(S, Sr, atol=1e-12 * scale), (F, S, Sr)
        assert np.allclose(U @ np.diag(S) @ V.T, F, atol=1e-12 * scale)
        assert np.allclose(Ur @ np.diag(Sr) @ Vr.T, F, atol=1e-12 * scale)
        assert abs(np.linalg.det(U) - 1) < 1e-10 and abs(np.linalg.det(V) - 1) < 1e-10
        if S[1] - abs(S[2]) > 1e-6 and S[0] - S[1] > 1e-6 and abs(S[2]) > 1e-6:  # unique up to paired signs
            assert np.allclose(np.abs(np.sum(U * Ur, axis=0)), 1.0, atol=1e-9)


@pytest.mark.parametrize("case", range(len(R.TRI_LIMITS)))
def test_tri_local_step_matches_reference(case):
    limits = R.TRI_LIMITS[case]
    verts, tris, x, u, mu, la, _ = tri_case(limits=limits)
    n, nv = len(tris), len(verts)
    g = ref_out("tri%d" % case)
    zr, ur, w, tr, tc, tv = g["z"], g["u"], g["w"], g["D_row"], g["D_col"], g["D_val"]
    assert int(g["nnz"][0]) == 18 * n
    o = orc.OracleSolver(x, np.ones(3 * nv), tris=dict(idx=tris, verts=verts, mu=mu, la=la, limit_min=limits[0], limit_max=limits[1]))
    z = np.zeros(6 * n); uo = u.copy()
    o.local_step(x.ravel(), z, uo)
    assert np.allclose(o.r_w, w, rtol=1e-13)
    Dref = sp.csr_matrix((tv, (tr, tc)), shape=(6 * n, 3 * nv))
    assert abs(Dref - o.D).max() < 1e-12
    assert np.allclose(z, zr, atol=1e-12) and np.allclose(uo, ur, atol=1e-12)


def test_pin_local_step_matches_reference():
    nv, x, vidx, pins, act, u = R.pin_case()
    g = ref_out("pin")
    zr, ur = g["z"], g["u"]
    assert abs(float(g["w"][0]) - orc.PIN_WEIGHT) < 1e-9
    z = np.zeros(18); uo = u.copy()
    orc.lib().orc_local_pins(3, _i(vidx), _p(pins), _i(act), _p(x), _p(z), _p(uo))
    rows = [0, 1, 2, 6, 7, 8, 12, 13, 14]  # rows 3..5 of a SpringPin block are never populated
    assert np.allclose(z[rows], zr[rows], atol=1e-14) and np.allclose(uo[rows], ur[rows], atol=1e-14)


def test_floor_constraints_match_reference():
    x, y0 = R.floor_case()
    nv = len(x)
    for case, cw in enumerate(R.FLOOR_CW):
        g = ref_out("floor%d" % case)
        rows, rv, rc, coef = int(g["rows"][0]), g["vert"], g["c"], g["coef"].ravel()
        o = orc.OracleSolver(x, np.ones(3 * nv), linsolver=2, constraint_w=cw, obstacles=[(0, [-0.2, 0, 0, 0])])
        hits = o.detect_passive(x.ravel())
        Cm, c = o.make_matrix(hits)
        assert rows == len(hits) == int((x[:, 1] < -0.2).sum())
        # the reference's hit order is thread order: compare as sets keyed by vertex
        got = {h[0]: (c[i], Cm[i].toarray().ravel()[3 * h[0]:3 * h[0] + 3]) for i, h in enumerate(hits)}
        for r in range(rows):
            cv, co = got[int(rv[r])]
            assert abs(cv - rc[r]) < 1e-13 and np.allclose(co, coef[3 * r:3 * r + 3], atol=1e-14)


def test_exact_solve_matches_eigen_ldlt():
    """scipy SuperLU (oracle's LDLT stand-in) vs the real Eigen::SimplicialLDLT on an assembled A."""
    o, A, b = R.ldlt_case()
    x = ref_out("ldlt")["x"]
    xo = o.solve_ldlt(b)
    assert np.linalg.norm(x - xo) <= 1e-9 * np.linalg.norm(xo)
    # A = M + Ahat (x) I3: no cross-axis coupling
    coo = A.tocoo()
    assert np.all(coo.row % 3 == coo.col % 3)


@pytest.mark.parametrize("which", [0, 1, 2])
def test_xu_splines_match_reference(which):
    """f, g, h, df, dg, dh of xu::NeoHookean / StVK / CoRotated (src/XuSpline.hpp:48-96) with and without the compression
    term (kappa != 0, :44-45): the Python restatement against the real header's values."""
    mu, la, _ = orc.lame(1e6, 0.3)
    for case, kappa in enumerate(R.SPLINE_KAPPAS):
        tab = ref_out("spline%d_%d" % (which, case))["fgh"]
        for x, row in zip(R.SPLINE_X, tab):
            got = np.array(orc.xu_spline(which, mu, la, kappa, float(x)))
            assert np.allclose(got, row, rtol=1e-13, atol=1e-9 * (1.0 + np.abs(row).max())), (which, kappa, x, got, row)


def test_xu_neohookean_spline_is_the_nh_model():
    """SplineTet's default spline (TetEnergyTerm.hpp:191-195) equals NeoHookeanTet's density."""
    mu, la, k = orc.lame(1e6, 0.3)
    for s in ([1.1, 0.9, 1.3], [0.5, 0.6, 2.0]):
        e = sum(orc.xu_spline(0, mu, la, 0.0, si)[0] for si in s) + orc.xu_spline(0, mu, la, 0.0, s[0] * s[1] * s[2])[2]
        x0 = np.array(s)
        v1 = orc.lib().orc_prox_value(1, mu, la, k, _p(x0), _p(x0))
        v3 = orc.lib().orc_prox_value(3, mu, la, k, _p(x0), _p(x0))
        assert abs(e - v1) < 1e-9 * abs(v1) + 1e-6 and abs(v3 - v1) < 1e-9 * abs(v1) + 1e-6


@pytest.mark.parametrize("which,kind", [(0, 3), (1, 4), (2, 5)])
@pytest.mark.parametrize("kappa", R.SPLINE_KAPPAS)
def test_xu_spline_prox_objective_matches_reference(which, kind, kappa):
    """SplineTet::SplineProx::value / gradient (TetEnergyTerm.cpp:243-265) assembled from the xu:: spline functions (pinned
    on the real src/XuSpline.hpp above, kappa = 0 and kappa != 0) against the oracle's kinds 3 / 4 / 5; and xu::StVK is the
    StVK model."""
    mu, la, k = orc.lame(1e6, 0.3)

    def spl(x):
        return orc.xu_spline(which, mu, la, kappa, float(x))     # f, g, h, df, dg, dh
    rng = np.random.default_rng(5)
    for _ in range(20):
        x = rng.uniform(0.4, 1.8, 3); x0 = rng.uniform(0.4, 1.8, 3)
        val = sum(spl(x[i])[0] for i in range(3)) + spl(x[0] * x[1])[1] + spl(x[1] * x[2])[1] + spl(x[2] * x[0])[1] \
            + spl(x[0] * x[1] * x[2])[2] + 0.5 * k * np.sum((x - x0) ** 2)
        hp = spl(x[0] * x[1] * x[2])[5]
        grad = np.array([
            spl(x[0])[3] + spl(x[0] * x[1])[4] * x[1] + spl(x[2] * x[0])[4] * x[2] + hp * x[1] * x[2] + k * (x[0] - x0[0]),
            spl(x[1])[3] + spl(x[1] * x[2])[4] * x[2] + spl(x[0] * x[1])[4] * x[0] + hp * x[2] * x[0] + k * (x[1] - x0[1]),
            spl(x[2])[3] + spl(x[2] * x[0])[4] * x[0] + spl(x[1] * x[2])[4] * x[1] + hp * x[0] * x[1] + k * (x[2] - x0[2])])
        vo = orc.lib().orc_prox_value_k(kind, mu, la, k, kappa, _p(x0), _p(x))
        go = np.zeros(3)
        orc.lib().orc_prox_gradient_k(kind, mu, la, k, kappa, _p(x0), _p(x), _p(go))
        assert abs(vo - val) <= 1e-12 * abs(val) + 1e-9
        assert np.abs(go - grad).max() <= 1e-12 * np.abs(grad).max() + 1e-9
        if kind == 4 and kappa == 0.0:       # the same numbers as StVKTet::StVKProx (TetEnergyTerm.cpp:210-237)
            assert abs(orc.lib().orc_prox_value(2, mu, la, k, _p(x0), _p(x)) - vo) <= 1e-9 * abs(vo)


def test_lame_matches_reference():
    tab = ref_out("lame")["mlk"]
    for (E, nu), row in zip(R.LAME_CASES, tab):
        assert np.allclose(orc.lame(E, nu), row, rtol=1e-15)
