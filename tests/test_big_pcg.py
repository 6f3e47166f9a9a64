"""The LAUNCH-PATH two-level PCG (csrc/pcg_big.hpp, oc_plan.cpp: build_big_plan; round-4 review, "missing" item 2): the global solve of
bodies beyond the chip's LDS (more than 262 144 vertices) and the fall-back of the on-chip kernel -- the same preconditioner
M^-1 = D^-1 + P (P^T A P)^-1 P^T with the affine coarse space, in kernels that stream the matrix.  Replaces the prefactored LDLT solve
of src/LinearSolver.hpp:87-90 like every PCG path.  Small scenes reach it with ADMM_HIP_PCG_LAUNCHES=1 (no on-chip kernel);
ADMM_HIP_BIG=0 keeps the Jacobi PCG of rounds 1-4 (the A/B)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import admm_elastic_amd as pkg
import scenes

pytestmark = pytest.mark.gpu


def _solver(sc, monkeypatch, big=True, **kw):
    monkeypatch.setenv("ADMM_HIP_PCG_LAUNCHES", "1")
    if not big:
        monkeypatch.setenv("ADMM_HIP_BIG", "0")
    try:
        return sc.make_solver(**kw)
    finally:
        monkeypatch.delenv("ADMM_HIP_PCG_LAUNCHES")
        monkeypatch.delenv("ADMM_HIP_BIG", raising=False)


@pytest.mark.parametrize("n", [6, 20])     # 6: one aggregate with dummy rows; 20: 12 aggregates
def test_big_pcg_solve_is_exact_and_needs_far_fewer_iterations_than_jacobi(n, monkeypatch):
    sc = scenes.blob_scene(n, admm_iters=5, linsolver=0)
    s = _solver(sc, monkeypatch, pcg_tol=1e-12, pcg_max_iters=4000)
    j = _solver(sc, monkeypatch, big=False, pcg_tol=1e-12, pcg_max_iters=4000)
    rp, ci, va = s.system_matrix()
    nv = len(sc.x)
    K = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsc()
    rng = np.random.default_rng(1)
    xs = rng.standard_normal((nv, 3))
    b = (K @ xs).ravel()
    x, it = s.global_solve(b, np.zeros(3 * nv))
    xj, itj = j.global_solve(b, np.zeros(3 * nv))
    assert np.abs(x - xs.ravel()).max() < 1e-8 * np.abs(xs).max(), np.abs(x - xs.ravel()).max()
    assert np.abs(xj - xs.ravel()).max() < 1e-8 * np.abs(xs).max()
    lu = spla.splu(K)
    assert np.abs(x.reshape(-1, 3) - lu.solve(b.reshape(-1, 3))).max() < 1e-8 * np.abs(xs).max()
    assert s.persistent_launches()["pcg"] == 0              # no on-chip kernel was involved
    if n == 20:
        assert it < itj, (it, itj)      # (a body this small is well conditioned; the coarse space's factor at size: test_onchip_pcg_preconditioner_modes, the size curve)
    # warm start from the solution: no iteration needed
    x2, it2 = s.global_solve(b, x)
    assert it2 <= 1 and np.abs(x2 - x).max() < 1e-9 * np.abs(xs).max()


@pytest.mark.parametrize("ls", [0, 2])
def test_big_pcg_whole_steps_match_the_on_chip_solver_and_the_oracle(ls, monkeypatch):
    sc = scenes.blob_scene(16, admm_iters=8, linsolver=ls)
    if ls == 2:
        sc.obstacles.append((0, [-0.05, 0.0, 0.0, 0.0]))
    a = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=3000)
    b = _solver(sc, monkeypatch, pcg_tol=1e-12, pcg_max_iters=3000)
    o = sc.make_oracle(mode=1, big=True) if ls == 0 else None
    for f in range(3):
        a.step(); b.step()
        assert b.runtime_data().unconverged_solves == 0
        assert scenes.rel_err(b.m_x, a.m_x) < (1e-8 if ls == 0 else 1e-5), (f, scenes.rel_err(b.m_x, a.m_x))
        if o is not None:
            o.step()
            assert scenes.rel_err(b.m_x, o.x) < 1e-7
    assert b.persistent_launches()["pcg"] == 0 and a.persistent_launches()["pcg"] > 0
    tot = b.solve_totals()
    assert tot[0] > 0 and tot[0] == tot[1]                   # its own totals: every solve converged


def test_big_pcg_is_the_fall_back_after_a_barrier_time_out(monkeypatch):
    """After an aborted on-chip solve the context replays on the launch path: now the two-level PCG, with the iteration counts of the
    on-chip solver's order of magnitude instead of Jacobi's."""
    sc = scenes.mixed_cube_scene(8, admm_iters=6, linsolver=0)
    ref = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=2000)
    monkeypatch.setenv("ADMM_HIP_TEST_ABORT_SOLVE", "3")
    s = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=2000)
    monkeypatch.delenv("ADMM_HIP_TEST_ABORT_SOLVE")
    for _ in range(3):
        s.step(); ref.step()
    assert scenes.rel_err(s.m_x, ref.m_x) < 1e-8
    assert s.runtime_data().unconverged_solves == 0
    assert s.runtime_data().inner_iters < 4 * max(ref.runtime_data().inner_iters, 60)


def test_launch_path_end_projection_takes_the_solves_own_residual(monkeypatch):
    """Round 6: after a launch-path solve the soft-mode Galerkin step forms Z^T r from the PCG's own final residual (k_big_scatter leaves D^-1 r
    in memory) instead of a second matrix-vector product.  Same step: the residual of the returned iterate is orthogonal to the modes, and the
    iterate equals the one of the product form (ADMM_HIP_DEFL_RESID=0 at create) to the solver's round-off; whole frames agree too."""
    sc = scenes.blob_scene(16, admm_iters=8, linsolver=0)
    s = _solver(sc, monkeypatch, pcg_tol=1e-6, pcg_max_iters=2000, soft_modes=16)
    monkeypatch.setenv("ADMM_HIP_DEFL_RESID", "0")
    p = _solver(sc, monkeypatch, pcg_tol=1e-6, pcg_max_iters=2000, soft_modes=16)
    monkeypatch.delenv("ADMM_HIP_DEFL_RESID")
    plain = _solver(sc, monkeypatch, pcg_tol=1e-6, pcg_max_iters=2000)
    rp, ci, va = s.system_matrix()
    nv = len(sc.x)
    K = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
    Z = s.get_soft_modes()
    rng = np.random.default_rng(4)
    xs = rng.standard_normal((nv, 3)) + 30.0 * (Z[:3].T @ rng.standard_normal((3, 3)))
    b = K @ xs
    x, _ = s.global_solve(b.ravel(), np.zeros(b.size))
    xp, _ = p.global_solve(b.ravel(), np.zeros(b.size))
    x0, _ = plain.global_solve(b.ravel(), np.zeros(b.size))
    proj = np.abs(Z @ (b - K @ x.reshape(-1, 3))).max(); proj_plain = np.abs(Z @ (b - K @ x0.reshape(-1, 3))).max()
    assert proj < 1e-6 * max(proj_plain, 1e-300) or proj < 1e-9 * np.abs(b).max(), (proj, proj_plain)
    assert np.abs(x - xp).max() < 1e-9 * np.abs(xs).max(), np.abs(x - xp).max()
    assert np.abs(x.reshape(-1, 3) - xs).max() < np.abs(x0.reshape(-1, 3) - xs).max()
    for f in range(3):
        s.step(); p.step()
        assert s.runtime_data().unconverged_solves == 0
    assert scenes.rel_err(s.m_x, p.m_x) < 1e-7, scenes.rel_err(s.m_x, p.m_x)
    assert s.persistent_launches()["pcg"] == 0


def test_body_beyond_the_chip_runs_the_two_level_launch_path():
    """A single body of ~1.6 M tets (more than 262 144 vertices): no on-chip plan exists; the launch-path two-level PCG serves it.  Two frames
    at the bench tolerance against the same path at 1e-12 (bar 1e-5, as for every workload), every solve converged."""
    n = int(os.environ.get("ADMM_TEST_HUGE_BLOB_N", "140"))
    import bench
    sc, nt, nv = bench.build_scene(bench.WORKLOADS["blob1m_mix"], n)
    if n == 140:
        assert nv > 262144, nv
    loose = sc.make_solver(pcg_tol=bench.PCG_TOL, pcg_max_iters=1500)
    tight = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=4000)
    for f in range(2):
        loose.step(); tight.step()
        assert loose.runtime_data().unconverged_solves == 0 and tight.runtime_data().unconverged_solves == 0
        assert scenes.rel_err(loose.m_x, tight.m_x) < 1e-5
    assert loose.persistent_launches()["pcg"] == 0
    its = loose.runtime_data().inner_iters / 20.0
    print("body of %d tets / %d verts: %.1f PCG iterations per solve on the launch path, %.2f ms per frame" % (nt, nv, its, loose.runtime_data().step_ms))
    assert its < 60
    loose.close(); tight.close()
