"""Randomised GPU-vs-oracle parity sweep (run on a GPU box): random jittered Kuhn meshes, materials, Lame parameters,
pins, gravity, initial velocities, ADMM iteration counts and global solvers.  Prints the worst position error."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "8")   # the oracle's GS opens one tiny OpenMP region per colour: 256 host threads make it 100x slower
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import admm_elastic_amd as pkg
from admm_elastic_amd import meshes
from admm_elastic_amd.solver import Lame
import scenes

KINDS = [pkg.TET_LINEAR, pkg.TET_NEOHOOKEAN, pkg.TET_STVK, pkg.TET_SPLINE_NH, pkg.TET_SPLINE_STVK, pkg.TET_SPLINE_COROTATED]
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0; fails = 0; t0 = time.time()
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(2, 8))
    verts, tets = meshes.kuhn_cube(n, float(rng.uniform(0.3, 2.0)))
    h = verts[:, 0].max() / n
    interior = np.all((verts > 1e-9) & (verts < verts.max() - 1e-9), axis=1)
    verts = verts + interior[:, None] * rng.uniform(-0.15, 0.15, verts.shape) * h          # jitter, tets stay positive
    assert meshes.tet_volumes(verts, tets).min() > 0
    sc = scenes.Scene()
    sc.x = verts; sc.m = meshes.lumped_masses_tets(verts, tets, float(rng.uniform(500, 3000)))
    nk = int(rng.integers(1, 4))
    part = rng.integers(0, nk, len(tets))
    for k in range(nk):
        sel = tets[part == k]
        if len(sel):
            lame = Lame(float(10 ** rng.uniform(4.5, 7.2)), float(rng.uniform(0.1, 0.45)))
            sc.tets.append((verts, sel, lame, int(rng.choice(KINDS)), 0))
    ls = int(rng.choice([0, 0, 1]))
    if rng.random() < 0.8:
        for i in np.nonzero(verts[:, 0] < 1e-9)[0]:
            sc.pins[int(i)] = verts[i].copy() + (rng.uniform(-0.02, 0.02, 3) if rng.random() < 0.3 else 0.0)
    if ls == 1 and rng.random() < 0.5:
        sc.obstacles.append((0, [float(verts[:, 1].min() - rng.uniform(0.0, 0.05)), 0.0, 0.0, 0.0]))
    sc.settings.update(admm_iters=int(rng.integers(2, 9)), linsolver=ls, gravity=float(rng.uniform(-15, 2)),
                       timestep_s=float(rng.choice([1 / 24, 1 / 60, 1 / 100])))
    s = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=3000)
    o = sc.make_oracle(mode=1, gs_colors=s.gs_colors()[0] if ls == 1 else None)
    v0 = rng.uniform(-0.5, 0.5, verts.size) * (rng.random() < 0.5)
    s.m_v[:] = v0; o.v[:] = v0
    for _ in range(int(rng.integers(1, 4))):
        s.step(); o.step()
    err = scenes.rel_err(s.m_x, o.x)
    worst = max(worst, err)
    bad = not (err < 1e-6) or s.runtime_data().unconverged_solves > 0
    fails += bad
    print("seed %3d n=%d nt=%5d models=%d ls=%d iters=%d pins=%3d floor=%d  err %.2e %s" % (seed, n, len(tets), nk, ls, sc.settings["admm_iters"], len(sc.pins), len(sc.obstacles), err, "FAIL" if bad else ""))
    s.close()
print("fuzz: %d scenes, %d failures, worst error %.2e, %.0f s" % (seeds, fails, worst, time.time() - t0))
