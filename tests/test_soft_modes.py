"""End projection of every PCG solve on the soft modes of the system matrix (round 5: admm_hip_compute_soft_modes / admm_hip_set_soft_modes;
csrc/pcg_onchip2.hpp: the epilogue of k_pcg2, csrc/kernels.hpp: k_defl_* as separate launches).  The reference solves exactly
(src/LinearSolver.hpp:87-90); a PCG stopped on a residual norm leaves its error where the eigenvalues are small, and that error is what
drifts over hundreds of frames (profiles/r05_drift_*).  The projection x += Z (Z^T K Z)^-1 Z^T (b - A x) is an exact Galerkin step:
afterwards the residual is orthogonal to the modes whatever their accuracy."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import admm_elastic_amd as pkg
import scenes

pytestmark = pytest.mark.gpu


def _K(s, sc):
    rp, ci, va = s.system_matrix()
    nv = len(sc.x)
    return (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()


def test_library_modes_are_the_lowest_eigenvectors():
    sc = scenes.blob_scene(16, admm_iters=6, linsolver=0)
    s = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=2000)
    s.compute_soft_modes(12)
    Z = s.get_soft_modes()
    K = _K(s, sc)
    assert Z.shape == (12, len(sc.x))
    G = Z @ Z.T
    assert np.abs(G - np.eye(12)).max() < 1e-6                       # orthonormal (to the single precision the fused projection stores them in)
    ritz = np.array([z @ (K @ z) for z in Z])
    assert (np.diff(ritz) > -1e-9 * ritz[-1]).all()                  # ascending
    ew = np.linalg.eigvalsh(K.toarray())[:12]
    assert np.abs(ritz[:8] - ew[:8]).max() < 1e-3 * ew[7], (ritz[:8], ew[:8])      # the lowest ones have converged
    for z, lam in zip(Z[:8], ritz[:8]):
        assert np.linalg.norm(K @ z - lam * z) < 3e-2 * lam


@pytest.mark.parametrize("fused", ["1", "0"])
def test_end_projection_makes_the_residual_orthogonal_to_the_modes(fused, monkeypatch):
    sc = scenes.blob_scene(16, admm_iters=6, linsolver=0)
    K = None
    monkeypatch.setenv("ADMM_HIP_DEFL_FUSED", fused)
    s = sc.make_solver(pcg_tol=1e-6, pcg_max_iters=2000, soft_modes=16)      # a LOOSE tolerance: the projection is what makes the modes exact
    plain = sc.make_solver(pcg_tol=1e-6, pcg_max_iters=2000)
    monkeypatch.delenv("ADMM_HIP_DEFL_FUSED")
    K = _K(s, sc)
    Z = s.get_soft_modes()
    rng = np.random.default_rng(4)
    xs = rng.standard_normal((len(sc.x), 3)) + 30.0 * (Z[:3].T @ rng.standard_normal((3, 3)))      # a solution with a large soft part
    b = K @ xs
    # the global solve of the ADMM loop is what carries the projection: one frame's worth of solves through the kernel-level entry point
    x, it = s.global_solve(b.ravel(), np.zeros(b.size))
    xp, itp = plain.global_solve(b.ravel(), np.zeros(b.size))
    r = b - K @ x.reshape(-1, 3); rp_ = b - K @ xp.reshape(-1, 3)
    proj = np.abs(Z @ r).max(); proj_plain = np.abs(Z @ rp_).max()
    assert proj < 1e-6 * max(proj_plain, 1e-300) or proj < 1e-9 * np.abs(b).max(), (proj, proj_plain)
    if fused == "1":      # the two forms of the step on ONE solve at a tight tolerance: equal to round-off level
        a = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=3000, soft_modes=16)
        monkeypatch.setenv("ADMM_HIP_DEFL_FUSED", "0")
        b2 = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=3000, soft_modes=16)
        monkeypatch.delenv("ADMM_HIP_DEFL_FUSED")
        xa, _ = a.global_solve(b.ravel(), np.zeros(b.size)); xb, _ = b2.global_solve(b.ravel(), np.zeros(b.size))
        assert np.abs(xa - xb).max() < 1e-8 * np.abs(xs).max()
    err = np.abs(x.reshape(-1, 3) - xs).max(); err_plain = np.abs(xp.reshape(-1, 3) - xs).max()
    assert err < err_plain                                            # and the iterate is closer to the solution
    assert (s.persistent_launches()["pcg"] > 0)


def test_fused_and_separate_projection_agree_over_frames_and_help_the_drift(monkeypatch):
    sc = scenes.blob_scene(16, admm_iters=10, linsolver=2)
    tight = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=3000)
    fused = sc.make_solver(pcg_tol=1e-7, pcg_max_iters=3000, soft_modes=16)
    monkeypatch.setenv("ADMM_HIP_DEFL_FUSED", "0")
    sep = sc.make_solver(pcg_tol=1e-7, pcg_max_iters=3000, soft_modes=16)
    monkeypatch.delenv("ADMM_HIP_DEFL_FUSED")
    plain = sc.make_solver(pcg_tol=1e-7, pcg_max_iters=3000)
    for f in range(12):
        for q in (tight, fused, sep, plain):
            q.step()
    # (the same step, in the kernel's epilogue -- on the recursive residual -- or as three launches on the true one: equal to the solver's
    # tolerance per solve, which twelve frames of a swaying body amplify)
    d_fs = scenes.rel_err(fused.m_x, sep.m_x)
    print("fused vs separate after 12 frames at pcg_tol 1e-7: %.2e" % d_fs)
    assert d_fs < 2e-5
    e_f, e_p = scenes.rel_err(fused.m_x, tight.m_x), scenes.rel_err(plain.m_x, tight.m_x)
    print("12 frames at pcg_tol 1e-7: rel_err %.2e with the end projection, %.2e without" % (e_f, e_p))
    assert e_f < 0.5 * e_p
    assert fused.runtime_data().unconverged_solves == 0


def test_block_smoother_survives_the_mode_computation(monkeypatch):
    """Round 6: from their second round on the solves of admm_hip_compute_soft_modes have (nearly) eigenvectors as right-hand sides -- CG is done
    after a step or two and runs on into rounding noise, where r . u of the recurrences can turn negative; k_pcg2 reads that as a preconditioner
    that is not positive definite and gives up its block smoother for the context.  That is what every context with library-computed modes ran
    with until round 6 (the bench body: 8.7 instead of 6.7 iterations per solve).  The findings of those solves are put back now: the smoother
    is alive afterwards (admm_hip_pcg_findings), and at work (fewer iterations than with ADMM_HIP_OC_CHEB=0)."""
    import bench
    sc, nt, nv = bench.build_scene(bench.WORKLOADS["blob1m_mix"], int(os.environ.get("ADMM_TEST_SMOOTHER_N", "118")))
    s = sc.make_solver(pcg_tol=7e-10, pcg_max_iters=1500, soft_modes=24)
    assert s.persistent_launches()["pcg"] > 0
    f = s.pcg_findings()
    assert not f["smoother_given_up"] and not f["trust_revoked"], f
    monkeypatch.setenv("ADMM_HIP_OC_CHEB", "0")
    j = sc.make_solver(pcg_tol=7e-10, pcg_max_iters=1500, soft_modes=24)
    monkeypatch.delenv("ADMM_HIP_OC_CHEB")
    t0s, t0j = s.solve_totals(), j.solve_totals()
    for _ in range(4):
        s.step(); j.step()
    its_s, its_j = s.solve_totals()[2] - t0s[2], j.solve_totals()[2] - t0j[2]
    print("%d tets: %d PCG iterations in 4 frames with the block smoother, %d with S = D^-1" % (nt, its_s, its_j))
    assert its_s < 0.93 * its_j, (its_s, its_j)
    assert not s.pcg_findings()["smoother_given_up"]
    assert scenes.rel_err(s.m_x, j.m_x) < 1e-6
    s.close(); j.close()


def test_soft_modes_argument_checks():
    sc = scenes.cube_scene(3, pkg.TET_NEOHOOKEAN, admm_iters=4, linsolver=1)
    s = sc.make_solver()
    with pytest.raises(pkg.AdmmHipError):
        s.compute_soft_modes(4)                                       # the GS context runs no PCG
    sc0 = scenes.cube_scene(3, pkg.TET_NEOHOOKEAN, admm_iters=4, linsolver=0)
    s0 = sc0.make_solver()
    with pytest.raises(pkg.AdmmHipError):
        s0.set_soft_modes(np.ones((2, len(sc0.x))))                   # linearly dependent
    s0.compute_soft_modes(4); assert s0.get_soft_modes().shape[0] == 4
    s0.set_soft_modes(None); assert s0.get_soft_modes().shape[0] == 0
    s0.step(); assert np.isfinite(s0.m_x).all()


def test_rank_contexts_compute_the_same_modes_as_the_single_context():
    """Multi-rank contexts (element-block partition: the solve is replicated): every rank computes the modes of the system it solves with
    the same deterministic code -- identical on all ranks, equal to the single context's.  (bench.py --gpus N runs its PCG workloads with
    the same settings as --gpus 1: tolerance AND soft modes.)"""
    sc = scenes.blob_scene(16, admm_iters=6, linsolver=0)
    single = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=2000, soft_modes=6)
    Z1 = single.get_soft_modes()
    assert Z1.shape[0] == 6
    single.close()
    for r in range(2):
        s = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=2000, soft_modes=6, rank=r, world_size=2)
        Z = s.get_soft_modes()
        assert Z.shape == Z1.shape
        assert np.array_equal(Z, Z1), np.abs(Z - Z1).max()
        s.close()


def test_start_step_in_front_of_the_second_solve_changes_iterations_not_results(monkeypatch):
    """ADMM_HIP_DEFL_START (default mask 2): the soft-mode Galerkin step in front of the second solve of a frame is one more exact projection --
    the converged trajectory moves within the tolerance, the PCG iterations of a frame do not go up (on the bench body: 65 -> 20 for that solve,
    profiles/r05_drift_start_projection.txt)."""
    sc = scenes.blob_scene(20, admm_iters=10, linsolver=0)
    monkeypatch.setenv("ADMM_HIP_DEFL_START", "0")
    off = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=3000, soft_modes=8)
    monkeypatch.delenv("ADMM_HIP_DEFL_START")
    on = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=3000, soft_modes=8)
    it_on = it_off = 0
    for f in range(8):
        on.step(); off.step()
        if f >= 3:
            it_on += on.runtime_data().inner_iters; it_off += off.runtime_data().inner_iters
    d = scenes.rel_err(on.m_x, off.m_x)
    print("start step: %d vs %d PCG iterations over 5 frames, trajectories differ by %.2e" % (it_on, it_off, d))
    assert d < 1e-8      # (measured 9e-11)
    assert it_on <= 1.02 * it_off
    assert on.runtime_data().unconverged_solves == 0
    on.close(); off.close()
