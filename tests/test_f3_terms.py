"""SURVEY section 8 row f3, the part the reference never shipped (README.md:23-28 lists them as TODOs): stable Neo-Hookean tets, a bending
term for cloth, slide constraints.  There is NO reference code for any of them, so parity is "unpinned": the oracle restates each from
its paper / its definition in include/admm_hip.h (oracle/admm_oracle.c: kind 7, oracle/oracle.py: bends, slides), and the HIP path is
held to the oracle exactly as for the terms the reference does ship -- the local step at 1e-10, whole steps at 1e-7 of the bounding box.

  stable Neo-Hookean   Smith, de Goes, Kim 2018: Psi = mu_s/2 (I_C - 3) + la_s/2 (J - alpha)^2 - mu_s/2 log(I_C + 1); one more
                       HyperElasticTet (src/TetEnergyTerm.cpp:114-136 is its prox), finite for inverted elements
  bending              Bergou et al. 2006 quadratic bending: one hinge term per interior edge, D_i x = sum_k c_k x_k, prox = q / 2
  slide constraints    a SpringPin (src/SpringEnergyTerm.hpp:31-73) whose prox projects onto a plane (linsolver 0 / 2); the plane-constrained
                       update of src/NodalMultiColorGS.hpp:218-262 on the pin's own plane inside the sweeps (linsolver 1)"""
import numpy as np
import pytest

import admm_elastic_amd as pkg
import scenes
from admm_elastic_amd import capi, meshes
from admm_elastic_amd.solver import Lame
from oracle import oracle as orc


# ---- CPU: oracle restatements and host set-up ------------------------------------------------------------------------------------
def test_stable_nh_oracle_gradient_hessian_and_rest_state():
    """The oracle's stable Neo-Hookean objective: gradient = finite differences of the value, zero gradient at the rest state (alpha is
    chosen for exactly that), finite for inverted stretches, and the prox of an inverted element comes back un-inverted."""
    mu, la, k = orc.lame(1.0e6, 0.3)
    L = orc.lib()
    x0 = np.array([1.0, 1.0, 1.0]); g = np.zeros(3)
    L.orc_prox_gradient_k(7, mu, la, k, 0.0, orc._p(x0), orc._p(x0), orc._p(g))
    assert np.abs(g).max() < 1e-9 * mu
    rng = np.random.default_rng(3)
    for _ in range(20):
        x = rng.uniform(-0.5, 2.0, 3); c = rng.uniform(0.2, 1.8, 3)
        L.orc_prox_gradient_k(7, mu, la, k, 0.0, orc._p(c), orc._p(x), orc._p(g))
        fd = np.zeros(3)
        for i in range(3):
            e = np.zeros(3); e[i] = 1e-6
            fd[i] = (L.orc_prox_value_k(7, mu, la, k, 0.0, orc._p(c), orc._p(x + e)) - L.orc_prox_value_k(7, mu, la, k, 0.0, orc._p(c), orc._p(x - e))) / 2e-6
        assert np.abs(g - fd).max() < 1e-6 * (np.abs(fd).max() + mu), (x, g, fd)
    F = np.diag([1.1, 0.9, -0.4]).T.ravel().copy()          # an inverted element
    L.orc_prox_tet_hyper_k(7, mu, la, k, 0.0, orc._p(F), 1)
    assert np.isfinite(F).all() and np.linalg.det(F.reshape(3, 3)) > -0.4 * 1.1 * 0.9     # pulled towards the un-inverted side


def test_bend_hinges_host_builder_matches_the_oracle_and_vanishes_on_flat_shapes():
    verts, tris = meshes.cloth_grid(7, size=1.0, y=0.5)
    rng = np.random.default_rng(0)
    verts = verts + 0.03 * rng.standard_normal(verts.shape) * np.array([1.0, 0.0, 1.0])     # irregular, still flat
    idx, coef, area = capi.bend_hinges(verts, tris)
    idx_o, coef_o, area_o = orc.bend_hinges(verts, tris)
    n_interior = 3 * 7 * 7 - 2 * 7            # edges of the grid (3 n^2 + 2 n) minus its 4 n boundary edges
    assert idx.shape == (n_interior, 4) and np.array_equal(idx, idx_o)
    assert np.abs(coef - coef_o).max() < 1e-12 and np.abs(area - area_o).max() < 1e-14
    assert np.abs(coef.sum(axis=1)).max() < 1e-12                                            # translation invariance
    assert np.abs(np.einsum("hk,hkj->hj", coef, verts[idx])).max() < 1e-12                   # no bending force in a flat configuration
    bent = verts.copy(); bent[:, 1] += 0.3 * np.sin(3.0 * verts[:, 0])
    assert np.abs(np.einsum("hk,hkj->hj", coef, bent[idx])).max() > 1e-2
    e_idx, e_coef, e_area = capi.bend_hinges(np.zeros((0, 3)), np.zeros((0, 3), np.int32))
    assert e_idx.shape == (0, 4)                                                              # empty mesh


@pytest.mark.parametrize("ls", [0, 1])
def test_assembled_matrix_with_bends_and_slides_matches_oracle(ls):
    sc = scenes.cloth_scene(6, limits=None, admm_iters=5, linsolver=ls)
    sc.bends.append((sc.tris[0][0], sc.tris[0][1], 0.5, 0))
    sc.slides[20] = (sc.x[20].copy(), np.array([0.0, 2.0, 0.0]))
    o = sc.make_oracle(mode=1, gs_colors=np.zeros(len(sc.x), np.int32) if ls == 1 else None)
    s = sc.make_solver(init=False)
    rp, ci, va = s.host_matrix(sc.product_settings)
    import scipy.sparse as sp
    K = sp.csr_matrix((va, ci, rp), shape=(len(sc.x),) * 2) + sp.diags(sc.m)
    Ko = o.A[0::3, :][:, 0::3]
    assert np.abs((K - Ko)).max() < 1e-12 * np.abs(Ko).max()
    assert o.R == 6 * len(sc.tris[0][1]) + 3 * 96 + (6 * 3 if ls == 0 else 0)


def test_bad_bend_and_slide_descriptions_are_rejected():
    sc = scenes.cloth_scene(4, limits=None, admm_iters=5, linsolver=0)
    s = sc.make_solver(init=False)
    with pytest.raises(pkg.AdmmHipError, match="zero normal"):
        s.set_slide_pins([3], [sc.x[3]], [np.zeros(3)])
    s.add_bends(sc.tris[0][0], sc.tris[0][1], 1.0)
    s._bends[0][0][0, 0] = 10 ** 6
    with pytest.raises(pkg.AdmmHipError, match="bend index out of range"):
        s.host_matrix(sc.product_settings)


# ---- GPU: the kernels against the oracle -----------------------------------------------------------------------------------------
def _deformed(sc, amp, seed):
    rng = np.random.default_rng(seed)
    x = sc.x @ np.array([[1.15, 0.1, 0.0], [0.0, 0.9, 0.05], [0.02, 0.0, 1.05]]).T
    return (x + amp * rng.standard_normal(x.shape)).ravel()


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [0.0, 0.01, 0.12, 0.3])   # 0.12 inverts a few tets, 0.3 many
def test_stable_nh_local_step_vs_oracle(amp):
    sc = scenes.cube_scene(4, pkg.TET_STABLE_NH, pin_face=False)
    s = sc.make_solver(); o = sc.make_oracle(mode=1)
    x = _deformed(sc, amp, 11)
    u0 = 0.05 * np.random.default_rng(12).standard_normal(o.R)
    Mxbar = np.random.default_rng(7).standard_normal(x.size)
    z, u, b = s.local_step(x, u0, Mxbar)
    zo = np.zeros(o.R); uo = u0.copy()
    o.local_step(x, zo, uo)
    # both sides minimise the same smooth objective from the same start with their own Newton iterations
    assert np.abs(z - zo).max() < 1e-9, np.abs(z - zo).max()
    assert np.abs(u - uo).max() < 1e-9
    bo = o.rhs(Mxbar, zo, uo)
    assert np.abs(b - bo).max() <= 1e-8 * np.abs(bo).max()
    if amp >= 0.12:
        assert (np.linalg.det((uo - u0 + zo).reshape(-1, 3, 3)) < 0).any(), "case meant to contain inverted elements"


@pytest.mark.gpu
@pytest.mark.parametrize("ls", [0, 1, 2])
def test_stable_nh_whole_steps_vs_oracle_and_next_to_neo_hookean(ls):
    """A cantilever of stable Neo-Hookean tets, three frames against the oracle; at small strain the model meets linear elasticity with the
    tet's Lame constants (the paper's re-parametrisation), so it stays close to the Neo-Hookean cantilever -- and unlike it, it takes
    a frame that starts from a crushed state."""
    sc = scenes.cube_scene(4, pkg.TET_STABLE_NH, admm_iters=12, linsolver=ls)
    s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=800)
    colors = s.gs_colors()[0] if ls == 1 else None
    o = sc.make_oracle(mode=1, gs_colors=colors)
    for f in range(3):
        s.step(); o.step()
        assert scenes.rel_err(s.m_x, o.x) < 1e-7, (f, scenes.rel_err(s.m_x, o.x))
    nh = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, admm_iters=12, linsolver=ls).make_solver(pcg_tol=1e-12, pcg_max_iters=800)
    for f in range(3):
        nh.step()
    assert 0.0 < scenes.rel_err(s.m_x, nh.m_x) < 2e-2
    # crushed start: every vertex on the plane x = 0.5 +- tiny (all tets degenerate or inverted)
    sc2 = scenes.cube_scene(3, pkg.TET_STABLE_NH, admm_iters=12, linsolver=0)
    s2 = sc2.make_solver(pcg_tol=1e-11, pcg_max_iters=800); o2 = sc2.make_oracle(mode=1)
    xc = sc2.x.copy(); xc[:, 0] = 0.5 - 0.3 * (xc[:, 0] - 0.5)          # mirrored and squeezed: det F < 0 everywhere
    for k in sc2.pins:
        xc[k] = sc2.x[k]
    s2.m_x = xc.ravel().copy(); o2.x = xc.ravel().copy()
    for f in range(2):
        s2.step(); o2.step()
    assert np.isfinite(s2.m_x).all()
    assert scenes.rel_err(s2.m_x, o2.x) < 1e-6


def _cloth_with_bends(n, ls, k_bend=2.0, **kw):
    sc = scenes.cloth_scene(n, limits=None, linsolver=ls, **kw)
    verts, tris = sc.tris[0][0], sc.tris[0][1]
    sc.bends.append((verts, tris, k_bend, 0))
    return sc


@pytest.mark.gpu
def test_bending_local_step_vs_oracle():
    sc = _cloth_with_bends(9, 0, admm_iters=5)
    s = sc.make_solver(); o = sc.make_oracle(mode=1)
    rng = np.random.default_rng(5)
    x = (sc.x + 0.05 * rng.standard_normal(sc.x.shape)).ravel()
    u0 = 0.05 * rng.standard_normal(o.R)
    Mxbar = rng.standard_normal(x.size)
    z, u, b = s.local_step(x, u0, Mxbar)
    zo = np.zeros(o.R); uo = u0.copy()
    o.local_step(x, zo, uo)
    nb = 3 * o.nbend
    r0 = o.h_row
    assert o.nbend == 3 * 81 - 2 * 9
    assert np.abs(z[r0:r0 + nb] - zo[r0:r0 + nb]).max() < 1e-13 and np.abs(u[r0:r0 + nb] - uo[r0:r0 + nb]).max() < 1e-13    # (a linear prox: exact)
    assert np.abs(z - zo).max() < 1e-11 and np.abs(u - uo).max() < 1e-11
    bo = o.rhs(Mxbar, zo, uo)
    assert np.abs(b - bo).max() <= 1e-11 * np.abs(bo).max()


@pytest.mark.gpu
@pytest.mark.parametrize("ls", [0, 1, 2])
def test_bending_whole_steps_vs_oracle_and_it_stiffens_the_cloth(ls):
    """A cloth hanging from two corners, with and without the bending term: frames against the oracle; the bending energy makes the
    sheet sag less sharply (its mean hinge curvature |sum_k c_k x_k| stays smaller)."""
    sc = _cloth_with_bends(10, ls, k_bend=5.0, admm_iters=15)
    s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=1500)
    colors = s.gs_colors()[0] if ls == 1 else None
    o = sc.make_oracle(mode=1, gs_colors=colors)
    for f in range(4):
        s.step(); o.step()
        assert scenes.rel_err(s.m_x, o.x) < 1e-7, (f, scenes.rel_err(s.m_x, o.x))
    plain = scenes.cloth_scene(10, limits=None, linsolver=ls, admm_iters=15).make_solver(pcg_tol=1e-12, pcg_max_iters=1500)
    for f in range(4):
        plain.step()
    idx, coef, _ = capi.bend_hinges(sc.tris[0][0], sc.tris[0][1])
    curv = lambda x: np.linalg.norm(np.einsum("hk,hkj->hj", coef, x.reshape(-1, 3)[idx]), axis=1).mean()
    assert curv(s.m_x) < 0.8 * curv(plain.m_x), (curv(s.m_x), curv(plain.m_x))


@pytest.mark.gpu
@pytest.mark.parametrize("ls", [0, 1, 2])
def test_slide_pins_vs_oracle(ls):
    """A block whose x = 0 face may only slide in its own plane (normal (1, 0, 0)) and whose bottom edge of that face is pinned: under gravity
    the face's free vertices move down IN the plane.  Against the oracle; with the GS the constraint holds exactly after every sweep,
    with the energy-term form (linsolver 0 / 2) like a SpringPin's: to the accuracy the ADMM iterations give it."""
    sc = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, pin_face=False, admm_iters=15, linsolver=ls)
    verts = sc.x
    face = [int(i) for i in np.nonzero(verts[:, 0] < 1e-9)[0]]
    for v in face:
        if verts[v, 1] < 1e-9:
            sc.pins[v] = verts[v].copy()
        else:
            sc.slides[v] = (verts[v].copy(), np.array([3.0, 0.0, 0.0]))
    s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=1500)
    colors = s.gs_colors()[0] if ls == 1 else None
    o = sc.make_oracle(mode=1, gs_colors=colors)
    for f in range(4):
        s.step(); o.step()
        assert scenes.rel_err(s.m_x, o.x) < 1e-7, (f, scenes.rel_err(s.m_x, o.x))
    X = s.m_x.reshape(-1, 3)
    sl = list(sc.slides)
    off = np.abs(X[sl, 0] - verts[sl, 0]).max()
    moved = np.abs(X[sl, 1] - verts[sl, 1]).max()
    assert moved > 1e-4                                   # they slide ...
    assert off < (1e-12 if ls == 1 else 1e-3 * moved + 1e-6), (off, moved)      # ... in the plane
    # moving the planes afterwards (set_slide_pins after initialize): push the face by 1 cm along its normal
    s.set_slide_pins(sl, [verts[v] + np.array([0.01, 0.0, 0.0]) for v in sl], [np.array([1.0, 0.0, 0.0])] * len(sl))
    o.slides = {v: (verts[v] + np.array([0.01, 0.0, 0.0]), np.array([1.0, 0.0, 0.0])) for v in sl}
    if ls != 1:
        o.p_xyz[len(sc.pins):] = np.array([o.slides[v][0] for v in sl])
    for f in range(2):
        s.step(); o.step()
    assert scenes.rel_err(s.m_x, o.x) < 1e-7
    assert np.abs(s.m_x.reshape(-1, 3)[sl, 0] - 0.01).max() < (1e-12 if ls == 1 else 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("ls", [0, 1, 2])
def test_slide_normal_does_not_outlive_its_constraint(ls):
    """Slide, then drop the slide constraints, then set_pins on the same vertices: they must be HELD (an ordinary pin), not slide on in
    the plane they once had (a stale normal kept by the context made them slide again).  A twin that keeps its slide constraints gives the
    distance a sliding vertex covers in the same two frames."""
    def make():
        sc = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, pin_face=False, admm_iters=15, linsolver=ls)
        verts = sc.x
        for v in (int(i) for i in np.nonzero(verts[:, 0] < 1e-9)[0]):
            if verts[v, 1] < 1e-9:
                sc.pins[v] = verts[v].copy()
            else:
                sc.slides[v] = (verts[v].copy(), np.array([1.0, 0.0, 0.0]))
        return sc, sc.make_solver(pcg_tol=1e-12, pcg_max_iters=1500)
    sc, s = make()
    _, twin = make()
    for f in range(2):
        s.step(); twin.step()
    sl = list(sc.slides)
    here = s.m_x.reshape(-1, 3)[sl].copy()
    s.set_slide_pins([], [], [])
    s.set_pins(list(sc.pins) + sl, [sc.pins[v] for v in sc.pins] + [here[i] for i in range(len(sl))])
    for f in range(2):
        s.step(); twin.step()
    held = np.abs(s.m_x.reshape(-1, 3)[sl] - here).max()
    slid = np.abs(twin.m_x.reshape(-1, 3)[sl] - here).max()
    assert slid > 1e-4, slid
    assert held < (1e-12 if ls == 1 else 0.05 * slid), (held, slid)
    # and back: the same vertices slide again when the constraints return
    s.set_pins(list(sc.pins), [sc.pins[v] for v in sc.pins])
    s.set_slide_pins(sl, [sc.x[v] for v in sl], [np.array([1.0, 0.0, 0.0])] * len(sl))
    for f in range(2):
        s.step()
    X = s.m_x.reshape(-1, 3)[sl]
    assert np.abs(X - here).max() > 0.2 * slid and np.abs(X[:, 0]).max() < (1e-12 if ls == 1 else 2e-3)


@pytest.mark.gpu
def test_f3_terms_in_the_multi_rank_partitions():
    """Element-block partition (every rank its block of hinges, partial right-hand sides summed) and the component partition (sub-scenes
    carry their hinges and slide normals): rank contexts on one device reproduce the single context."""
    sc = _cloth_with_bends(8, 0, admm_iters=6)
    sc.slides[40] = (sc.x[40].copy(), np.array([0.0, 1.0, 0.0]))
    single = sc.make_solver(pcg_tol=1e-12)
    x = (sc.x + 0.03 * np.random.default_rng(2).standard_normal(sc.x.shape)).ravel()
    R = single.num_rows()
    u0 = 0.02 * np.random.default_rng(3).standard_normal(R)
    Mxbar = np.random.default_rng(4).standard_normal(x.size)
    z1, u1, b1 = single.local_step(x, u0, Mxbar)
    import os
    os.environ["ADMM_HIP_PARTITION"] = "elements"
    try:
        parts = [sc.make_solver(pcg_tol=1e-12, rank=r, world_size=3) for r in range(3)]
    finally:
        os.environ.pop("ADMM_HIP_PARTITION")
    bsum = np.zeros_like(b1); zsum = np.zeros(R); usum = np.zeros(R)
    for p in parts:
        z, u, b = p.local_step(x, u0, Mxbar)
        bsum += b
        own = (z != 0.0) | (u != u0)
        zsum[own] = z[own]; usum[own] = u[own]
    assert np.abs(bsum - b1).max() <= 1e-11 * np.abs(b1).max()
    assert np.abs(zsum - z1).max() < 1e-12
