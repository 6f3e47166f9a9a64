"""Parity at the BENCHMARKED settings and sizes (round-3 review, item 1): every workload bench.py quotes a number on has a twin
here that runs it under a checker at the size, solver settings and frame count the driver times.

  blob1m_mix            configs[2]  1 012 608 tets, bench.PCG_TOL + bench.SOFT_MODES, recycled warm start, unverified short passes:
                                    two frames of the 1e-12 path against the ORACLE's exact solves at full size; 200 frames (the
                                    driver's timed region is frames 5-24, its statistics frames 25-44) of the bench settings against
                                    that 1e-12 path (kernel vs kernel), rel_err < 1e-5 at EVERY frame; the 52 k-tet twin at bench
                                    settings against the oracle's exact (SuperLU) solves for 25 frames
  cube1m_nh, cube1m_mix             998 250 tets, their own settings (bench.workload_settings): 200 frames like the blob
  cube100k_gs           configs[1]  105 456 tets, 30-sweep multi-colour GS: whole frames against the oracle, shared colouring
  cloth200k_gs_floor    configs[4]  199 712 triangles, limits, pins, floor inside the sweeps: whole frames against the oracle until
                                    the cloth lies on the floor
  cube100k_uzawa_floor              105 456 tets on a floor, UzawaCG with 729 active rows, active set frozen per step on both sides

Reference: src/Solver.cpp:80-101 (the ADMM loop), src/NodalMultiColorGS.hpp:60-146, src/UzawaCG.hpp:57-125.
Tolerance: 1e-5 of the bounding-box diagonal per vertex (BASELINE.json north_star); tighter where the test says so."""
import os

import numpy as np
import pytest

import admm_elastic_amd as pkg
import scenes

pytestmark = pytest.mark.gpu


def _bench_scene(name, n=None):
    import bench
    return bench.build_scene(bench.WORKLOADS[name], n)


def test_blob1m_two_frames_vs_oracle_exact_solves():
    """Round-4 review, item 1(a): the parity chain closed AT THE HEADLINE SIZE.  Two whole frames (40 ADMM iterations) of the 1 012 608-tet
    body: HIP at 1e-12 with every pass verified against the ORACLE -- its OpenMP local step (exact minimiser) and exact global solves.
    The exact solve at this size is oracle.BigExactSolve (CPU conjugate gradients in numpy / scipy to a TRUE relative residual of 1e-13
    per axis, re-formed from scratch; SuperLU needs a 12-minute factorisation and 345 M non-zeros of fill here -- measured -- and agrees
    with it to 3e-10 of the bounding box on the 52 k-tet twin).  Together with test_blob1m_drift_* (bench settings vs this 1e-12 path
    at every frame) the chain bench settings -> oracle holds at 1 M tets.  Bound: 1e-7 of the bounding box (measured ~1e-9)."""
    n = int(os.environ.get("ADMM_TEST_BIG_BLOB_N", "118"))
    sc, nt, nv = _bench_scene("blob1m_mix", n)
    if n == 118:
        assert nt == 1012608
    os.environ["ADMM_HIP_OC_VERIFY"] = "1"
    try:
        s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=1500)
    finally:
        os.environ.pop("ADMM_HIP_OC_VERIFY", None)
    o = sc.make_oracle(mode=1, big=True, exact="pcg")
    errs = []
    for f in range(2):
        s.step(); o.step()
        assert s.runtime_data().unconverged_solves == 0
        errs.append(scenes.rel_err(s.m_x, o.x))
    print("blob1m vs oracle (exact solves) rel_err per frame:", " ".join("%.2e" % e for e in errs),
          " oracle CG: %d iterations in %d solves, worst true relative residual %.1e" % (o._big.iterations, o._big.solves, o._big.worst_residual))
    assert o._big.worst_residual <= 1e-13
    assert max(errs) < 1e-7, errs
    assert np.abs(s.m_x - sc.x.ravel()).max() > 1e-4
    s.close()


def _drift_200_frames(workload, n=None):
    """One quoted PCG workload with ITS OWN bench settings (bench.workload_settings: tolerance, soft modes, the library's default start step)
    against the same path converged to 1e-12 with every pass verified, frame by frame -- through the driver's whole run (warm-up + timed +
    statistics frames) and far beyond it: 200 frames (ADMM_TEST_DRIFT_FRAMES overrides).  NOTE what this is: kernel against kernel -- the
    oracle enters the chain at 1 M tets for two frames (test_blob1m_two_frames_vs_oracle_exact_solves: the 1e-12 path vs exact solves,
    ~1e-9) and at 52 k tets for 25 frames at bench settings (test_blob52k_drift_25_frames_bench_settings_vs_oracle).  The per-frame record
    goes to gpurun_out/drift_<workload>_frames.txt (committed under profiles/ per round)."""
    import bench
    sc, nt, nv = _bench_scene(workload, n)
    tol, soft = bench.workload_settings(workload)
    frames = int(os.environ.get("ADMM_TEST_DRIFT_FRAMES", "200"))
    os.environ["ADMM_HIP_OC_VERIFY"] = "1"
    try:
        tight = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=1500)
    finally:
        os.environ.pop("ADMM_HIP_OC_VERIFY", None)
    loose = sc.make_solver(pcg_tol=tol, pcg_max_iters=600, soft_modes=soft)        # what `python bench.py --workload <workload>` runs
    errs = []
    for f in range(frames):
        tight.step(); loose.step()
        assert tight.runtime_data().unconverged_solves == 0 and loose.runtime_data().unconverged_solves == 0, f
        errs.append(scenes.rel_err(loose.m_x, tight.m_x))
    print("%s drift rel_err, %d frames at pcg_tol %g, %d soft modes: max %.2e at frame %d; every 10th:" % (workload, frames, tol, soft, max(errs), int(np.argmax(errs))),
          " ".join("%.2e" % e for e in errs[9::10]))
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "drift_%s_frames.txt" % workload), "w") as fh:
            fh.write("# %s (%d tets), bench settings (pcg_tol %g, soft modes %d) vs the same path at 1e-12 verified: rel_err per frame\n" % (workload, nt, tol, soft))
            fh.write("\n".join("%d %.3e" % (i, e) for i, e in enumerate(errs)) + "\n")
    except OSError:
        pass
    assert max(errs) < 1e-5, errs
    assert np.abs(tight.m_x - sc.x.ravel()).max() > 1e-3      # the body actually moves
    tight.close(); loose.close()


def test_blob1m_drift_200_frames_bench_tolerance_vs_tight_solve():
    """configs[2], the driver's default workload (round-4 review, item 1(b): "nobody knows the error at frame 300")."""
    n = int(os.environ.get("ADMM_TEST_BIG_BLOB_N", "118"))
    _drift_200_frames("blob1m_mix", n)


@pytest.mark.parametrize("workload", ["cube1m_nh", "cube1m_mix"])
def test_cube1m_drift_200_frames_bench_settings_vs_tight_solve(workload):
    """Round-5 review, weak item 1: the other two quoted 1 M-tet PCG workloads -- cube1m_nh is the north-star's own mesh -- had no drift record
    at the settings bench.py runs them with.  They have their OWN settings now (bench.WORKLOADS: a looser tolerance, no soft-mode step: the
    pinned-face cube has no soft global modes), asserted here like the blob's."""
    n = int(os.environ.get("ADMM_TEST_BIG_N", "55"))
    _drift_200_frames(workload, n)


def test_blob52k_drift_25_frames_bench_settings_vs_oracle():
    """The same body at 52 464 tets, bench settings, against the ORACLE (exact minimiser + SuperLU direct solves) for 25 frames."""
    sc, nt, nv = _bench_scene("blob1m_mix", 44)
    assert nt == 52464
    import bench
    s = sc.make_solver(pcg_tol=bench.PCG_TOL, pcg_max_iters=600, soft_modes=bench.SOFT_MODES)
    o = sc.make_oracle(mode=1, big=True)
    errs = []
    for f in range(25):
        s.step(); o.step()
        assert s.runtime_data().unconverged_solves == 0
        errs.append(scenes.rel_err(s.m_x, o.x))
    print("blob52k vs oracle rel_err per frame:", " ".join("%.2e" % e for e in errs))
    assert max(errs) < 1e-5, errs
    s.close()


def test_cube100k_gs_full_size_vs_oracle():
    """configs[1] at its benchmarked size: 105 456 NH tets, 20 ADMM iterations x 30 sweeps, multi-block colour kernels + hipGraph
    replay + fused residual test, against the oracle's sweep-for-sweep GS on the shared colouring."""
    sc, nt, nv = _bench_scene("cube100k_gs")
    assert nt == 105456
    s = sc.make_solver()
    colors, nc = s.gs_colors()
    o = sc.make_oracle(mode=1, gs_colors=colors, big=True)
    for f in range(3):
        s.step(); o.step()
        err = scenes.rel_err(s.m_x, o.x)
        assert err < 1e-6, (f, err)
        assert s.runtime_data().inner_iters == o.inner_iters == 20 * 30
    assert np.abs(s.m_x - sc.x.ravel()).max() > 1e-3
    s.close()


@pytest.mark.parametrize("floor", [0.3, 0.47])
def test_cloth200k_gs_floor_full_size_vs_oracle(floor):
    """configs[4] at its benchmarked size: 199 712 strain-limited triangles, two pins, Floor handled inside the GS sweeps (three
    colours, 256 blocks of the persistent kernel).  floor 0.3 = the bench scene: the cloth swings on its two pins and only comes down
    on the floor in frame 29 (oracle), after the driver's timed frames; floor 0.47 = the same scene with the floor raised into the
    cloth's first dip (frames 2-4: without it the lowest vertex passes 0.449, 0.398, 0.330): the plane projection inside the sweeps is
    active there and no vertex ever ends a frame below the floor."""
    import bench
    sc, nt, nv = _bench_scene("cloth200k_gs_floor")
    assert nt == 199712
    sc.obstacles[:] = [(0, [floor, 0.0, 0.0, 0.0])]
    s = sc.make_solver()
    colors, nc = s.gs_colors()
    o = sc.make_oracle(mode=1, gs_colors=colors, big=True)
    errs, on_floor = [], []
    free = np.ones(nv, bool); free[list(sc.pins)] = False
    for f in range(8):
        s.step(); o.step()
        errs.append(scenes.rel_err(s.m_x, o.x))
        assert s.runtime_data().inner_iters == o.inner_iters
        y = s.m_x.reshape(-1, 3)[free, 1]
        assert y.min() >= floor - 1e-12                  # never below the floor: the plane projection is exact
        on_floor.append(float(y.min() - floor))
    print("cloth200k floor %.2f rel_err per frame:" % floor, " ".join("%.1e" % e for e in errs), " lowest vertex above the floor:", " ".join("%.1e" % g for g in on_floor))
    assert max(errs) < 1e-6, errs
    if floor > 0.4:
        assert min(on_floor) < 2e-2                      # it was stopped by the floor (and swings up again on its two pins)
    s.close()


@pytest.mark.parametrize("n, n_rows", [(26, 729), (30, 961)])
def test_cube100k_uzawa_floor_full_size_frozen_active_set(n, n_rows, monkeypatch):
    """The bench's contact workload at its size (105 456 NH tets dropped on a Floor, 729 active rows) with the chaos of the
    free-running active set taken out as in test_step_uzawa_frozen_active_set_is_tight: Collider::detect in the first ADMM
    iteration of a step on both sides.  Cached K^-1 columns, the Schur CG of a solve as one persistent launch, bench tolerance.
    n = 30 (162 000 tets, 961 rows): the persistent kernel's layout for more than 800 rows (8 rows of the Schur matrix per block,
    121 blocks) at a real size."""
    sc, nt, nv = _bench_scene("cube100k_uzawa_floor", n)
    monkeypatch.setenv("ADMM_HIP_UZ_FREEZE", "1")
    import bench
    s = sc.make_solver(pcg_tol=bench.PCG_TOL, pcg_max_iters=600)
    monkeypatch.delenv("ADMM_HIP_UZ_FREEZE")
    o = sc.make_oracle(mode=1, big=True)
    o.freeze_active = True
    rows = 0
    for f in range(3):
        s.step(); o.step()
        rows = max(rows, len(o._hits))
        err = scenes.rel_err(s.m_x, o.x)
        assert err < 1e-6, (f, err)
    assert rows == n_rows, rows
    st = s.uzawa_cache_stats()
    assert st["schur_from_columns"] > 0 and st["schur_by_pcg"] == 0
    assert s.m_x.reshape(-1, 3)[:, 1].min() > -0.02 - 1e-6
    s.close()


def test_contact_counters_cloth_and_cube():
    """admm_hip_contact_totals (round-4 review, item 1(c): bench.py states `rows_projected_in_timed_region`): rows projected onto the
    floor inside the GS sweeps -- counted by the persistent kernel (LDS counter, one atomic per block and solve) exactly like by the
    colour kernels -- and rows of C over the UzawaCG solves."""
    sc = scenes.cloth_scene(40, limits=(0.95, 1.05), floor=0.47, admm_iters=10, linsolver=1)
    counts = []
    for persist in ("1", "0"):
        os.environ["ADMM_HIP_GS_PERSIST"] = persist
        try:
            s = sc.make_solver()
        finally:
            os.environ.pop("ADMM_HIP_GS_PERSIST")
        assert s.contact_totals() == 0
        per_frame = []
        for f in range(6):
            s.step(); per_frame.append(s.contact_totals())
        assert per_frame[-1] > 0                       # the cloth dips into the raised floor within the first frames
        counts.append(per_frame)
        s.close()
    assert counts[0] == counts[1], counts              # bit-identical sweeps: identical projection counts
    scu = scenes.cube_scene(6, pkg.TET_NEOHOOKEAN, pin_face=False, admm_iters=8, linsolver=2)
    scu.pins.clear(); scu.obstacles.append((0, [-0.02, 0.0, 0.0, 0.0]))
    su = scu.make_solver(pcg_tol=1e-10)
    tot = 0
    for f in range(8):
        su.step()
    assert su.contact_totals() >= 49                   # the bottom face (7 x 7 vertices) rests on the floor
    su.close()
