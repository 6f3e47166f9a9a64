"""Dynamic (self-)collision -- SURVEY 8(f) item 2: TetMeshCollision (src/DynamicObject.hpp:31-121) found by
Collider::detect (src/Collider.hpp:152-212) and turned into the dynamic rows of ConstraintSet::make_matrix
(src/ConstraintSet.hpp:92-110) for UzawaCG.

The reference holds no test for this path and its BVH library (mclscene) is absent, so parity is against the oracle's
brute-force restatement (oracle/admm_oracle.c: orc_detect_dynamic), whose own invariants are checked first on the CPU.
Tolerances: detection is exact in the discrete outputs (hit set, faces) and ~1e-12 in the FP64 payloads; solves agree
to 1e-7 like the passive Uzawa tests."""
import numpy as np
import pytest

import admm_elastic_amd as pkg
from admm_elastic_amd import meshes
from admm_elastic_amd.capi import AdmmHipError
from oracle import oracle as orc
import scenes


# ------------------------------------------------------------------------------------------ CPU: oracle + host logic
def test_surface_faces_of_a_cube_are_closed_and_outward():
    n = 3
    verts, tets = meshes.kuhn_cube(n)
    faces = meshes.surface_faces(tets)
    assert faces.shape == (12 * n * n, 3)
    c = verts[faces].mean(axis=1)
    nrm = np.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]])
    assert (np.einsum("ij,ij->i", nrm, c - 0.5) > 0).all()          # outward
    edges = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(edges, axis=0, return_counts=True)
    assert (cnt == 2).all()                                         # closed 2-manifold
    assert len(meshes.surface_inds(tets)) == (n + 1) ** 3 - (n - 1) ** 3


def _probe_scene(p):
    """unit cube (6 tets) + a far-away tet whose vertex 8 is moved to p: the only candidate that can be inside"""
    sc = scenes.Scene()
    verts, tets = meshes.kuhn_cube(1)
    off = sc.add_tet_mesh(verts, tets, scenes.Lame(1e6, 0.3), pkg.TET_LINEAR)
    sc.add_self_collision(verts, tets, off)
    pv = np.array([[5.0, 5, 5], [6, 5, 5], [5, 6, 5], [5, 5, 6]])
    sc.add_tet_mesh(pv, np.array([[0, 1, 2, 3]], np.int32), scenes.Lame(1e6, 0.3), pkg.TET_LINEAR)
    sc.surface_inds = []          # every vertex is a candidate (Collider.hpp:157)
    x = sc.x.copy(); x[8] = p
    return sc, x.ravel()


def test_oracle_detect_hand_worked():
    """DynamicObject.hpp:72-119 on a case done by hand: a vertex at (0.3, 0.5, 0.4) inside the unit cube is 0.3 from
    the x = 0 face: dx = -0.3, normal -x, proj = (0, 0.5, 0.4) = sum_j bary_j face_j."""
    sc, x = _probe_scene([0.3, 0.5, 0.4])
    sc.settings.update(linsolver=2)
    o = sc.make_oracle()
    hits = o.detect_dynamic(x)
    assert len(hits) == 1
    v, dx, face, bary, n = hits[0]
    assert v == 8 and abs(dx + 0.3) < 1e-15
    assert np.allclose(n, [-1, 0, 0], atol=1e-15)
    assert np.allclose(bary @ sc.x[face], [0.0, 0.5, 0.4], atol=1e-15) and abs(bary.sum() - 1) < 1e-15 and (bary >= 0).all()
    # a vertex outside, or exactly on the surface (dx = 0 is not < 0, Collider.hpp:203), is no hit
    for p in ([1.2, 0.5, 0.5], [0.0, 0.5, 0.4]):
        assert o.detect_dynamic(_probe_scene(p)[1]) == []
    # the cube's own vertices never collide with the cube: every tet / face touching the vertex is skipped (:78,:100)
    assert all(h[0] == 8 for h in o.detect_dynamic(_probe_scene([0.5, 0.45, 0.7])[1]))


def test_oracle_detect_uses_the_rest_shape():
    """The depth is measured in REST space (:92-104): deform the cube by x -> 2x; the vertex at (0.6, 1.0, 0.8)
    maps back to (0.3, 0.5, 0.4): dx = -0.3 (not -0.6) and the normal is the rest normal."""
    sc, _ = _probe_scene([0, 0, 0])
    sc.settings.update(linsolver=2)
    x = sc.x.copy(); x[:8] *= 2.0; x[8] = [0.6, 1.0, 0.8]
    (v, dx, face, bary, n), = sc.make_oracle().detect_dynamic(x.ravel())
    assert v == 8 and abs(dx + 0.3) < 1e-14 and np.allclose(n, [-1, 0, 0], atol=1e-15)


def test_oracle_dynamic_rows():
    """ConstraintSet.hpp:92-110: row = ck n^T (x_v - sum_j bary_j x_face_j), rhs 0; a vertex that already holds a passive
    row keeps it and the dynamic row stays empty (:96-99) but still counts as a row."""
    sc = scenes.two_blocks_scene(3, floor=None)
    o = sc.make_oracle()
    dh = o.detect_dynamic(o.x)
    assert len(dh) > 5
    Cm, c = o.make_matrix([], dh)
    assert Cm.shape[0] == len(dh) and np.all(c == 0)
    assert np.diff(Cm.indptr).tolist() == [12] * len(dh)
    t = np.tile([0.3, -1.1, 0.7], o.nv)
    assert np.abs(Cm @ t).max() < 1e-14                      # rigid translations are not resisted: 1 - sum bary = 0
    X = o.x.reshape(-1, 3)
    for i, (v, dx, face, bary, n) in enumerate(dh):
        assert abs((Cm @ o.x)[i] - n @ (X[v] - bary @ X[face])) < 1e-13
    v0 = dh[0][0]
    ph = [(v0, -0.1, X[v0].copy(), np.array([0.0, 1.0, 0.0]))]
    Cm2, c2 = o.make_matrix(ph, dh)
    assert Cm2.shape[0] == 1 + len(dh) and Cm2[1].nnz == 0 and Cm2[0].nnz == 3


def test_oracle_gs_recolouring_rule():
    """NodalMultiColorGS.hpp:80-86 re-colours A + C^T C (library absent).  The build's rule: untouched nodes keep their
    colour, touched nodes get new colours after the old ones, first fit in node order; the result is a proper colouring
    of the node graph of A + C^T C."""
    sc = scenes.two_blocks_scene(3, linsolver=1)
    s = sc.make_solver(init=False)
    from admm_elastic_amd import capi
    rp, ci, _ = s.host_matrix(sc.product_settings)
    base, _ = capi.greedy_coloring(rp, ci)
    o = sc.make_oracle(gs_colors=base)
    dh = o.detect_dynamic(o.x)
    col = orc.recolor_touched(base, o.Ah, dh)
    touched = sorted({int(v) for h in dh for v in [h[0], *h[2]]})
    K = base.max() + 1
    assert (col[touched] >= K).all() and (np.delete(col, touched) == np.delete(base, touched)).all()
    Cm, _ = o.make_matrix([], dh)
    M = (o.A + Cm.T @ Cm).tocoo()
    a, b = M.row // 3, M.col // 3
    off = (a != b) & (M.data != 0)
    assert (col[a[off]] != col[b[off]]).all()


# ------------------------------------------------------------------------------------------ GPU: parity with the oracle
def _same_hits(hg, ho):
    assert [h[0] for h in hg] == [h[0] for h in ho]
    for g, o in zip(hg, ho):
        assert (g[2] == o[2]).all(), (g, o)
        assert abs(g[1] - o[1]) < 1e-12 and np.abs(g[3] - o[3]).max() < 1e-11 and np.abs(g[4] - o[4]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n,amp", [(2, 0.0), (3, 0.0), (3, 0.03), (6, 0.02)])
def test_detect_parity(n, amp):
    """refit + point-in-tet + nearest rest triangle on the GPU = the oracle's brute force, on the rest shape and on
    deformed states (amp: Gaussian noise on every vertex: tets tilt, a few may even invert)."""
    sc = scenes.two_blocks_scene(n)
    s = sc.make_solver()
    o = sc.make_oracle()
    x = scenes.perturb(sc.x, amp, seed=3).ravel() if amp else sc.x.ravel()
    ho = o.detect_dynamic(x)
    assert len(ho) >= (n - 1) ** 2
    _same_hits(s.detect_dynamic(x), ho)
    # without Solver::surface_inds every vertex is a candidate (Collider.hpp:157)
    sc.surface_inds = []
    _same_hits(sc.make_solver().detect_dynamic(x), sc.make_oracle().detect_dynamic(x))


@pytest.mark.gpu
def test_detect_tiny_meshes():
    """Smallest possible objects: two single tets (trees of one padded leaf group), one corner of the second inside the
    first; and a mesh that touches nothing."""
    sc = scenes.Scene()
    t0 = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    t1 = np.array([[0.2, 0.25, 0.15], [1.2, 1.3, 0.2], [1.3, 0.2, 1.1], [0.3, 1.4, 1.2]])
    one = np.array([[0, 1, 2, 3]], np.int32)
    for v in (t0, t1, t0 + 10.0):
        if meshes.tet_volumes(v, one)[0] < 0:
            v = v[[0, 2, 1, 3]]
        off = sc.add_tet_mesh(v, one, scenes.Lame(1e6, 0.3), pkg.TET_LINEAR)
        sc.add_self_collision(v, one, off)
    sc.settings.update(linsolver=2)
    ho = sc.make_oracle().detect_dynamic(sc.x.ravel())
    assert [h[0] for h in ho] == [4]                       # the corner of the second tet that sits inside the first
    _same_hits(sc.make_solver().detect_dynamic(sc.x.ravel()), ho)


@pytest.mark.gpu
def test_detect_three_meshes_first_object_keeps_the_payload():
    """A vertex inside two other meshes at once: the object added first answers (DynamicObject.hpp:73)."""
    sc = scenes.two_blocks_scene(2, floor=None)
    verts, tets = meshes.kuhn_cube(2)
    verts = verts * 0.5 + np.array([0.33, 0.62, 0.31]) + 0.003 * np.random.default_rng(1).uniform(-1, 1, verts.shape)
    off = sc.add_tet_mesh(verts, tets, scenes.Lame(1e6, 0.3), pkg.TET_LINEAR)
    sc.add_self_collision(verts, tets, off)
    ho = sc.make_oracle().detect_dynamic(sc.x.ravel())
    assert sum(1 for h in ho if h[0] >= off) >= 4
    _same_hits(sc.make_solver().detect_dynamic(sc.x.ravel()), ho)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["persistent", "two_launches", "list_by_many_blocks"])
@pytest.mark.parametrize("floor", [None, 0.02])
def test_global_solve_uzawa_dynamic_rows(floor, variant, monkeypatch):
    """UzawaCG::solve with dynamic rows (and passive ones next to them) at the SOLVE level.  34 rows need ~31 Schur-CG
    iterations; the reference's cap is 20 (UzawaCG.hpp:44), where both sides stop unconverged (|Cx-c| ~ 7e-5) and CG's
    sensitivity shows (1.6e-6) -- so the tight comparison runs with the cap lifted, the default cap loosely.
    Variants: the persistent Schur kernel on the coupled rows (default: uz_persist.hpp + k_uzc_schur), two launches per Schur
    iteration (ADMM_HIP_UZ_PERSIST=0), the active / row lists by the many-block kernels of large scenes (ADMM_HIP_UZ_LIST_BLOCKS=0)."""
    if variant == "two_launches":
        monkeypatch.setenv("ADMM_HIP_UZ_PERSIST", "0")
    if variant == "list_by_many_blocks":
        monkeypatch.setenv("ADMM_HIP_UZ_LIST_BLOCKS", "0")
    for cap, tol in ((100, 1e-8), (20, 1e-5)):
        sc = scenes.two_blocks_scene(3, floor=floor)
        s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=600, uzawa_max_iters=cap)
        o = sc.make_oracle(uzawa_max_iters=cap)
        rng = np.random.default_rng(2)
        x = sc.x.ravel().copy()
        b = o.A @ (x + 0.001 * rng.standard_normal(x.size))
        hits = o.detect_passive(x)
        o._dhits = o.detect_dynamic(x)
        assert len(o._dhits) > 5 and (floor is None or len(hits) > 3)
        for rep in range(2):      # second call: multiplier warm start (UzawaCG.hpp:74)
            xo, ito = o.solve_uzawa(x, b, hits)
            xg, itg = s.global_solve(b, x)
            assert np.abs(xg - xo).max() < tol, (cap, rep, np.abs(xg - xo).max())
            assert abs(itg - ito) <= 5
        if cap == 100:            # the constraint rows are satisfied by the solution: C x = c
            Cm, c = o.make_matrix(hits, o._dhits)
            assert np.abs(Cm @ xg - c).max() < 1e-8


@pytest.mark.gpu
def test_step_with_self_collision():
    """Whole steps: the upper block starts 0.2 inside the lower one, is pushed out in the first frame, both fall on the
    floor and the upper one lands on the lower one again.  First frame (same hit set on both sides): tight parity;
    later frames: loose, the active set is chaotic by construction (see test_step_uzawa_collisions_loose)."""
    sc = scenes.two_blocks_scene(3, floor=-0.3)
    s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=600)
    o = sc.make_oracle()
    s.step(); o.step()
    assert s.runtime_data().inner_iters > 5
    assert s.runtime_data().collision_ms > 0.0          # RuntimeData::collision_ms = detect + constraint rows
    assert scenes.rel_err(s.m_x, o.x) < 1e-6
    dyn_frames = 0
    for _ in range(14):
        s.step(); o.step()
        dyn_frames += 1 if len(o._dhits) else 0
    assert dyn_frames >= 1, "scene meant to collide again"
    assert np.isfinite(s.m_x).all()
    assert scenes.rel_err(s.m_x, o.x) < 5e-2
    # the blocks do not pass through each other: the upper block's lowest vertex stays above the lower block's centre
    X = s.m_x.reshape(-1, 3); nvb = len(X) // 2
    assert X[nvb:, 1].min() > X[:nvb, 1].mean()


@pytest.mark.gpu
def test_self_collision_steps_are_bit_reproducible():
    """C^T y of the dynamic rows scatters to face vertices that several rows share; it is summed in fixed-point integers
    (csrc/dyn_collide.hpp: k_uz_ct_dyn), so two runs of a colliding scene are identical to the last bit -- with FP64 atomic
    adds (rounds 1 and 2a) they differed by round-off and, the active set being chaotic, soon by more."""
    runs = []
    for rep in range(2):
        sc = scenes.two_blocks_scene(3, floor=-0.3)
        s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=600)
        hits = 0
        for _ in range(6):
            s.step()
            hits += s.runtime_data().inner_iters
        runs.append(s.m_x.copy()); s.close()
        assert hits > 5
    assert np.array_equal(runs[0], runs[1])


def _gs_pair(n, floor, **kw):
    """GPU solver + oracle on the two-blocks scene with the multi-colour GS, sharing the base colouring"""
    sc = scenes.two_blocks_scene(n, floor=floor, linsolver=1, **kw)
    s = sc.make_solver()
    o = sc.make_oracle(gs_colors=s.gs_colors()[0])
    return sc, s, o


@pytest.mark.gpu
@pytest.mark.parametrize("floor", [None, 0.02])
def test_global_solve_gs_dynamic_rows(floor):
    """NodalMultiColorGS::solve with dynamic hits (NodalMultiColorGS.hpp:75-146): A + C^T C swept colour by colour --
    the GPU never forms the matrix (untouched rows from the SELL, touched rows from their hits), the oracle does;
    sweep for sweep on the shared colouring, with and without in-sweep floor projection."""
    sc, s, o = _gs_pair(3, floor)
    rng = np.random.default_rng(4)
    x = sc.x.ravel().copy()
    b = o.A @ (x + 0.001 * rng.standard_normal(x.size))
    o._dhits = o.detect_dynamic(x)
    assert len(o._dhits) > 5
    xo, ito = o.solve_gs(x, b)
    xg, itg = s.global_solve(b, x)
    assert itg == ito == 30
    assert np.abs(xg - xo).max() < 1e-9, np.abs(xg - xo).max()
    # the penalty acts: the same solve without the proxies leaves the blocks interpenetrating
    o._dhits = []
    xfree, _ = o.solve_gs(x, b)
    assert np.abs(xg - xfree).max() > 1e-3
    if floor is not None:      # vertices projected on the floor never satisfy their row: no convergence to test
        return
    # a converging case: the residual test of A + C^T C stops both at the same sweep
    sc2 = scenes.two_blocks_scene(3, floor=floor, linsolver=1)
    s2 = sc2.make_solver(gs_tol=1e-3, gs_max_iters=200)
    o2 = sc2.make_oracle(gs_colors=s2.gs_colors()[0], gs_tol=1e-3, gs_max_iters=200)
    o2._dhits = o2.detect_dynamic(x)
    xo2, ito2 = o2.solve_gs(x, b)
    xg2, itg2 = s2.global_solve(b, x)
    assert 0 < ito2 < 200 and itg2 == ito2
    assert np.abs(xg2 - xo2).max() < 1e-9


@pytest.mark.gpu
def test_step_gs_with_self_collision():
    """Whole steps with -ls 1 (the reference's boxes.cpp set-up).  First frame tight, later frames loose (active set)."""
    sc, s, o = _gs_pair(3, -0.3)
    s.step(); o.step()
    assert scenes.rel_err(s.m_x, o.x) < 1e-7
    dyn_frames = 0
    for _ in range(14):
        s.step(); o.step()
        dyn_frames += 1 if len(o._dhits) else 0
    assert dyn_frames >= 1 and np.isfinite(s.m_x).all()
    assert scenes.rel_err(s.m_x, o.x) < 5e-2
    X = s.m_x.reshape(-1, 3); nvb = len(X) // 2
    assert X[nvb:, 1].min() > X[:nvb, 1].mean()


@pytest.mark.gpu
def test_surface_inds_restrict_passive_detection():
    """Collider.hpp:157,163: with Solver::surface_inds set, only those vertices are tested against the obstacles."""
    sc = scenes.cube_scene(3, pkg.TET_LINEAR, pin_face=False, admm_iters=4, linsolver=2, size=0.5)
    sc.obstacles.append((0, [0.03, 0.0, 0.0, 0.0]))
    verts, tets = sc.tets[0][0], sc.tets[0][1]
    sc.surface_inds = [int(i) for i in meshes.surface_inds(tets) if verts[i, 0] > 0.2]
    s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=400)
    o = sc.make_oracle()
    x = sc.x.copy(); x[:, 1] -= 0.05; x = x.ravel()
    hits = o.detect_passive(x)
    assert 0 < len(hits) < (verts[:, 1] < 0.08).sum()
    b = o.A @ x
    xo, _ = o.solve_uzawa(x, b, hits)
    xg, _ = s.global_solve(b, x)
    assert np.abs(xg - xo).max() < 1e-7


@pytest.mark.gpu
def test_dynamic_collider_errors():
    sc = scenes.two_blocks_scene(2, linsolver=0, floor=None)
    with pytest.raises(AdmmHipError, match="No collisions with LDLT solver"):      # Solver.cpp:249-254
        sc.make_solver()
    sc = scenes.two_blocks_scene(2, floor=None)
    sc.dynamic[0]["faces"] = np.zeros((0, 3), np.int32)
    with pytest.raises(AdmmHipError, match="needs surface faces"):                 # DynamicObject.hpp:49-51
        sc.make_solver()
