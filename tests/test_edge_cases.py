"""Edge cases the reference guards (or would crash on): empty / minimal scenes, ragged term mixes,
degenerate elements, error paths.  GPU-marked where a context is needed."""
import numpy as np
import pytest

import admm_elastic_amd as pkg
from admm_elastic_amd import capi, meshes
from admm_elastic_amd.solver import Lame, Settings, Solver
from oracle import oracle as orc
import scenes


def test_oracle_prox_degenerate_elements():
    """collapsed-to-a-point and flat deformation gradients go through the reference's fix-ups
    (TetEnergyTerm.cpp:128-133) without NaNs"""
    mu, la, k = orc.lame(1e6, 0.3)
    for F in (np.zeros((3, 3)), np.diag([1.0, 1.0, 0.0]), np.diag([1.0, 1e-9, -1e-9]), 1e-9 * np.ones((3, 3))):
        for kind in (1, 2, 3):
            z = np.ascontiguousarray(F.T).ravel().copy()
            orc.lib().orc_prox_tet_hyper(kind, mu, la, k, orc._p(z), 1)
            assert np.isfinite(z).all()
        z = np.ascontiguousarray(F.T).ravel().copy()
        orc.lib().orc_prox_tet_linear(orc._p(z))
        assert np.isfinite(z).all()


def test_host_helpers_accept_empty_inputs():
    B, v = capi.tet_rest(np.zeros((4, 3)), np.zeros((0, 4), np.int32))
    assert B.shape == (0, 9) and v.shape == (0,)
    R, a = capi.tri_rest(np.zeros((3, 3)), np.zeros((0, 3), np.int32))
    assert R.shape == (0, 4)
    assert capi.partition(0, 4, 2) == (0, 0)


@pytest.mark.gpu
def test_scene_without_energy_terms_is_free_fall():
    """no terms: A = M, every solve returns x_bar (the reference would do the same with an empty D)"""
    s = Solver()
    x = np.random.default_rng(0).standard_normal((5, 3))
    s.add_nodes(x, np.ones(15))
    assert s.initialize(Settings(gravity=-9.8, admm_iters=3, linsolver=0))
    s.step()
    dt = 1.0 / 24.0
    assert np.allclose(s.m_x.reshape(-1, 3)[:, 1], x[:, 1] + dt * dt * -9.8, atol=1e-12)
    assert np.allclose(s.m_x.reshape(-1, 3)[:, [0, 2]], x[:, [0, 2]], atol=1e-12)
    assert s.num_rows() == 0


@pytest.mark.gpu
def test_single_element_scenes_and_zero_iterations():
    for kind in (pkg.TET_LINEAR, pkg.TET_NEOHOOKEAN, pkg.TET_STVK, pkg.TET_SPLINE_NH):
        s = Solver()
        verts = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0]], float)
        s.add_nodes(verts, np.ones(12))
        s.add_tets(verts, [[0, 1, 2, 3]], Lame.soft_rubber(), kind)
        assert s.initialize(Settings(gravity=0.0, admm_iters=0))
        s.step()                       # zero ADMM iterations: x = x_bar = x
        assert np.allclose(s.m_x, verts.ravel())
        s.close()


@pytest.mark.gpu
def test_ragged_mix_of_all_term_types():
    """tets of every model + triangles + pins + unused vertices in one scene (ragged SELL slices,
    isolated rows) against the oracle"""
    sc = scenes.mixed_cube_scene(3, admm_iters=6, linsolver=0)
    v2, tris = meshes.cloth_grid(3, 0.7, 1.4)
    sc.add_tri_mesh(v2, tris, Lame(100.0, 0.1))
    sc.x = np.concatenate([sc.x, np.array([[5.0, 5.0, 5.0], [6.0, 5.0, 5.0]])])   # two isolated nodes
    sc.m = np.concatenate([sc.m, [0.3, 0.4]])
    s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=300)
    o = sc.make_oracle()
    for _ in range(3):
        s.step(); o.step()
    assert scenes.rel_err(s.m_x, o.x) < 1e-7


@pytest.mark.gpu
def test_degenerate_deformations_through_the_kernel():
    """collapsed / flat / inverted inputs: finite outputs that match the oracle where the problem is
    well-posed (linear + NH), finite everywhere"""
    sc = scenes.cube_scene(2, pkg.TET_NEOHOOKEAN, pin_face=False)
    s = sc.make_solver()
    o = sc.make_oracle(mode=1)
    R = o.R
    x = sc.x.copy()
    x[:, 2] = 0.0                       # flatten the whole mesh into a plane
    z, u = s.local_step(x.ravel(), np.zeros(R))
    assert np.isfinite(z).all() and np.isfinite(u).all()
    x = np.zeros_like(sc.x)             # collapse to a point
    z, u = s.local_step(x.ravel(), np.zeros(R))
    assert np.isfinite(z).all() and np.isfinite(u).all()
    x = -sc.x                           # point reflection: every tet inverted
    z, u = s.local_step(x.ravel(), np.zeros(R))
    zo = np.zeros(R); uo = np.zeros(R)
    o.local_step(x.ravel(), zo, uo)
    assert np.abs(z - zo).max() < 1e-7


@pytest.mark.gpu
def test_error_paths_keep_the_reference_messages():
    sc = scenes.cube_scene(2, pkg.TET_LINEAR)
    s = sc.make_solver()
    with pytest.raises(pkg.AdmmHipError, match="Constraint for"):
        s.set_pins([26], [np.zeros(3)])       # vertex 26 (x = 1 face) was never a pin: Solver.cpp:147-151
    s2 = Solver()
    with pytest.raises(pkg.AdmmHipError, match="not initialized"):
        s2.step_device()
    bad = scenes.cube_scene(2, pkg.TET_LINEAR)
    bad.m[:] = 0.0
    with pytest.raises(pkg.AdmmHipError, match="mass"):
        bad.make_solver()


@pytest.mark.gpu
def test_barrier_timeout_falls_back_and_replays(monkeypatch):
    """the persistent PCG kernel needs its blocks co-resident; when a grid barrier cannot complete (here:
    injected with the test hook ADMM_HIP_TEST_ABORT_SOLVE) the context must not fail the step -- it switches to the
    launch-per-iteration PCG, restores the last good state and replays the steps issued since, also when those steps
    were issued without statistics (asynchronously)."""
    import scenes
    sc = scenes.mixed_cube_scene(8, admm_iters=6, linsolver=0)
    ref = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=2000)
    for _ in range(4):
        ref.step()
    monkeypatch.setenv("ADMM_HIP_TEST_ABORT_SOLVE", "15")     # 3rd frame, 3rd ADMM iteration
    s = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=2000)
    monkeypatch.delenv("ADMM_HIP_TEST_ABORT_SOLVE")
    s.upload()
    for _ in range(3):
        s.step_device(stats=False)          # asynchronous: the time-out is only seen at the next synchronisation
    s.step_device(stats=True)
    s.download()
    assert np.isfinite(s.m_x).all()
    assert scenes.rel_err(s.m_x, ref.m_x) < 1e-8, scenes.rel_err(s.m_x, ref.m_x)
    assert s.runtime_data().unconverged_solves == 0
    # after the abort the context must not go back to the persistent kernel -- neither in the replay nor later (the
    # frame-history branch of the recycled start used to launch it unconditionally)
    n_oc = s.persistent_launches()["pcg"]
    assert n_oc == 24          # (the four frames were all issued on the on-chip kernel before the abort was seen; launches behind the abort leave at once)
    for _ in range(2):
        s.step_device(stats=True); ref.step()
    s.download()
    assert s.persistent_launches()["pcg"] == n_oc
    assert scenes.rel_err(s.m_x, ref.m_x) < 1e-8
    assert s.runtime_data().unconverged_solves == 0
    ref = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=2000)
    for _ in range(4):
        ref.step()
    # the same with the time-out inside a step that asks for statistics
    monkeypatch.setenv("ADMM_HIP_TEST_ABORT_SOLVE", "8")
    s2 = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=2000)
    monkeypatch.delenv("ADMM_HIP_TEST_ABORT_SOLVE")
    for _ in range(4):
        s2.step()
    assert scenes.rel_err(s2.m_x, ref.m_x) < 1e-8


@pytest.mark.gpu
def test_pin_in_place_after_device_resident_steps_uses_the_current_positions():
    """Solver::set_pins(inds) without points pins at the CURRENT m_x (src/Solver.cpp:121-125); after
    step_device() the host copy is stale, so the binding has to fetch the device state first."""
    import scenes
    sc = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, admm_iters=5, linsolver=1)
    s = sc.make_solver()
    s.upload()
    for _ in range(5):
        s.step_device()
    stale = s.m_x.copy()
    free = [i for i in range(len(sc.x)) if i not in sc.pins][:3]
    s.set_pins(list(sc.pins.keys()) + free)
    now = s.m_x.reshape(-1, 3)
    assert np.abs(s.m_x - stale).max() > 1e-4          # the body moved while the host copy was stale
    for v in free:
        assert np.array_equal(s._pins[v], now[v])


# ---- user-defined passive obstacles (src/Collider.hpp:66-83: PassiveCollision is an interface; round 4) -----------------------------
class _UserSphere:
    """What a user's PassiveCollision subclass looks like to the Python mirror: signed_distance(x) -> (dx, point, normal)."""

    def __init__(self, c, r):
        self.c, self.r = np.asarray(c, dtype=np.float64), float(r)

    def signed_distance(self, x):
        d = x - self.c
        l = np.linalg.norm(d)
        return l - self.r, self.c + d / l * self.r, d / l


@pytest.mark.gpu
@pytest.mark.parametrize("ls", [1, 2])
def test_plane_obstacle_equals_floor_and_oracle(ls, monkeypatch):
    """ADMM_OBJ_PLANE: the half space n.x < d.  With n = (0, 1, 0) it is the reference's Floor (src/PassiveObject.hpp:32-45) up to the
    rounding of x - dx n; a tilted plane is checked against the oracle's restatement of the same object, whole steps.
    UzawaCG (ls = 2): the active set is frozen per step (one Collider::detect per step, as in test_step_uzawa_frozen_active_set_is_tight).
    Free-running, a vertex that rests ON the floor is detected or not by the last bit of its height, and the one-ulp difference between the
    two contact points decides that differently from some frame on in EVERY solver variant (experiments/plane_floor_sensitivity.py: floor
    heights -0.01 / -0.0123, cached columns or inner solves, persistent Schur kernel or two launches per iteration: 3e-5 at step 2-4)."""
    from admm_elastic_amd.solver import Plane
    if ls == 2:
        monkeypatch.setenv("ADMM_HIP_UZ_FREEZE", "1")
    sc = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, pin_face=False, admm_iters=8, linsolver=ls, size=0.5)
    sc.pins.clear()
    res = []
    for obst in ("floor", "plane"):
        sc2 = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, pin_face=False, admm_iters=8, linsolver=ls, size=0.5)
        sc2.pins.clear()
        sol = sc2.make_solver(init=False)
        sol.add_obstacle(pkg.Floor(-0.01) if obst == "floor" else Plane([0.0, 2.0, 0.0], -0.02))     # (normalised by the library)
        assert sol.initialize(sc2.product_settings)
        for _ in range(4):
            sol.step()
        res.append(sol.m_x.copy()); sol.close()
    assert np.abs(res[0] - res[1]).max() < 1e-9 and np.abs(res[0] - sc.x.ravel()).max() > 1e-3
    if ls == 1:       # tilted plane, GS with the plane projection inside the sweeps, against the oracle
        n = np.array([0.3, 1.0, -0.2]); n /= np.linalg.norm(n)
        sc3 = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, pin_face=False, admm_iters=8, linsolver=1, size=0.5)
        sc3.pins.clear()
        sc3.obstacles.append((2, [n[0], n[1], n[2], -0.03]))
        sol = sc3.make_solver(init=False)
        sol._obstacles = [Plane(n, -0.03)]
        assert sol.initialize(sc3.product_settings)
        o = sc3.make_oracle(mode=1, gs_colors=sol.gs_colors()[0])
        closest = 1.0
        for _ in range(5):
            sol.step(); o.step()
            closest = min(closest, (sol.m_x.reshape(-1, 3) @ n + 0.03).min())
        assert scenes.rel_err(sol.m_x, o.x) < 1e-7
        assert -1e-3 < closest < 1e-6        # it came down on the plane (and bounced)


@pytest.mark.gpu
def test_sampled_user_obstacle_matches_the_analytic_object():
    """ADMM_OBJ_GRID: a user-defined PassiveCollision (here: a sphere written by the user) is sampled at initialize and interpolated on
    the device.  Against the oracle evaluating the SAME object analytically (its Sphere): the trajectories agree to the interpolation
    error of the grid, O(h^2) -- stated: 2e-3 of the bounding box at 48^3 nodes (spacing 0.03), 4x smaller at 96^3 (measured 9.8e-4 / 2.4e-4)."""
    from admm_elastic_amd.solver import SampledObstacle
    c, r = np.array([0.25, -0.32, 0.25]), 0.3
    errs = []
    for nodes in (48, 96):
        sc = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, pin_face=False, admm_iters=8, linsolver=1, size=0.5)
        sc.pins.clear()
        sc.obstacles.append((1, [c[0], c[1], c[2], r]))
        sol = sc.make_solver(init=False)
        sol._obstacles = [SampledObstacle(_UserSphere(c, r), [-0.5, -0.5, -0.5], [1.0, 0.8, 1.0], (nodes, nodes, nodes))]
        assert sol.initialize(sc.product_settings)
        o = sc.make_oracle(mode=1, gs_colors=sol.gs_colors()[0])
        for _ in range(6):
            sol.step(); o.step()
        errs.append(scenes.rel_err(sol.m_x, o.x))
        assert len(o.detect_passive(o.x)) > 0 or np.linalg.norm(o.x.reshape(-1, 3) - c, axis=1).min() < r + 1e-6     # it does touch
        sol.close()
    assert errs[0] < 2e-3 and errs[1] < 0.4 * errs[0], errs


def test_plane_normal_is_normalised_and_failing_obstacle_callbacks_raise():
    """the Python Plane stores the unit normal and the offset scaled with it (as the C++ mirror's Plane and the library
    do), so the oracle's kind-2 plane sees the same half space; a user obstacle whose signed_distance raises (or returns something
    malformed) must fail the sampling loudly instead of leaving distance 0 / normal 0 in the grid."""
    from admm_elastic_amd.solver import Plane, SampledObstacle
    p = Plane((0.0, 2.0, 0.0), 1.0)
    assert p.params == [0.0, 1.0, 0.0, 0.5]

    class Bad:
        def __init__(self):
            self.calls = 0

        def signed_distance(self, x):
            self.calls += 1
            if self.calls == 5:
                raise ValueError("user obstacle failed at a node")
            return x[1], (x[0], 0.0, x[2]), (0.0, 1.0, 0.0)
    with pytest.raises(ValueError, match="failed at a node"):
        SampledObstacle(Bad(), (-1, -1, -1), (1, 1, 1), dims=(4, 4, 4)).sample()

    class Malformed:
        def signed_distance(self, x):
            return 0.0, (0.0, 0.0)          # two components instead of three
    with pytest.raises((IndexError, TypeError, ValueError)):
        SampledObstacle(Malformed(), (-1, -1, -1), (1, 1, 1), dims=(4, 4, 4)).sample()


@pytest.mark.gpu
def test_solver_params_after_initialize():
    """admm_hip_set_solver_params (round 4 review, boundary item): the reference reads NodalMultiColorGS::max_iters / m_tol / m_omega and
    UzawaCG::max_iters / m_tol on every solve (src/NodalMultiColorGS.hpp:40-46,100; src/UzawaCG.hpp:44-45,92)."""
    import scenes
    sc = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, admm_iters=6, linsolver=1)
    s = sc.make_solver()
    assert s.solver_params(pkg.LS_NCMCGS) == (30, 1e-10, 1.9)
    s.step()
    assert s.runtime_data().inner_iters == 6 * 30
    s.set_solver_params(pkg.LS_NCMCGS, max_iters=15)
    s.step()
    assert s.runtime_data().inner_iters == 6 * 15
    assert s.solver_params(pkg.LS_NCMCGS) == (15, 1e-10, 1.9)
    # omega and the tolerance against the oracle run with the same members
    colors, nc = s.gs_colors()
    o = sc.make_oracle(mode=1, gs_colors=colors, gs_max_iters=12, gs_omega=1.5, gs_tol=1e-3)
    s2 = sc.make_solver()
    s2.set_solver_params(pkg.LS_NCMCGS, max_iters=12, tol=1e-3, omega=1.5)
    for _ in range(2):
        s2.step(); o.step()
        assert s2.runtime_data().inner_iters == o.inner_iters        # (the loose tolerance stops sweeps early on both sides)
    assert scenes.rel_err(s2.m_x, o.x) < 1e-8
    # the colour-kernel path (captured hipGraph) picks the new values up as well
    import os
    os.environ["ADMM_HIP_GS_PERSIST"] = "0"
    try:
        s3 = sc.make_solver()
    finally:
        os.environ.pop("ADMM_HIP_GS_PERSIST")
    s3.step()
    assert s3.runtime_data().inner_iters == 6 * 30
    s3.set_solver_params(pkg.LS_NCMCGS, max_iters=15)
    s3.step()
    assert s3.runtime_data().inner_iters == 6 * 15
    with pytest.raises(pkg.AdmmHipError):
        s3.set_solver_params(pkg.LS_UZAWACG, max_iters=5)            # this context does not run UzawaCG
    # PCG standing for the prefactored solve: a tolerance change is honoured by the next solve
    sc0 = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, admm_iters=6, linsolver=0)
    a = sc0.make_solver(pcg_tol=1e-4, pcg_max_iters=500); b = sc0.make_solver(pcg_tol=1e-12, pcg_max_iters=500)
    a.set_solver_params(pkg.LS_LDLT, tol=1e-12)
    for _ in range(2):
        a.step(); b.step()
    assert scenes.rel_err(a.m_x, b.m_x) < 1e-10
