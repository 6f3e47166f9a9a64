"""Randomised parity sweep: jittered Kuhn meshes, random constitutive models / Lame parameters / densities, pins
(some displaced), gravity, time step, initial velocities, ADMM iteration counts, PCG or multi-colour GS (with and
without a floor) -- the HIP step against the oracle on every seed."""
import numpy as np
import pytest

import admm_elastic_amd as pkg
from admm_elastic_amd import meshes
from admm_elastic_amd.solver import Lame
import scenes

pytestmark = pytest.mark.gpu
KINDS = [pkg.TET_LINEAR, pkg.TET_NEOHOOKEAN, pkg.TET_STVK, pkg.TET_SPLINE_NH]


@pytest.mark.parametrize("block", range(4))
def test_random_scenes_match_the_oracle(block):
    for seed in range(10 * block, 10 * block + 10):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(2, 8))
        verts, tets = meshes.kuhn_cube(n, float(rng.uniform(0.3, 2.0)))
        h = verts[:, 0].max() / n
        interior = np.all((verts > 1e-9) & (verts < verts.max() - 1e-9), axis=1)
        verts = verts + interior[:, None] * rng.uniform(-0.15, 0.15, verts.shape) * h     # tets stay positively oriented
        assert meshes.tet_volumes(verts, tets).min() > 0
        sc = scenes.Scene()
        sc.x = verts; sc.m = meshes.lumped_masses_tets(verts, tets, float(rng.uniform(500, 3000)))
        nk = int(rng.integers(1, 4))
        part = rng.integers(0, nk, len(tets))
        for k in range(nk):
            sel = tets[part == k]
            if len(sel):
                sc.tets.append((verts, sel, Lame(float(10 ** rng.uniform(4.5, 7.2)), float(rng.uniform(0.1, 0.45))), int(rng.choice(KINDS)), 0))
        ls = int(rng.choice([0, 0, 1]))
        if rng.random() < 0.8:
            for i in np.nonzero(verts[:, 0] < 1e-9)[0]:
                sc.pins[int(i)] = verts[i].copy() + (rng.uniform(-0.02, 0.02, 3) if rng.random() < 0.3 else 0.0)
        if ls == 1 and rng.random() < 0.6:
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
        assert err < 1e-6, (seed, err)
        assert s.runtime_data().unconverged_solves == 0, seed
        s.close()


@pytest.mark.parametrize("block", range(2))
def test_random_scenes_with_the_round5_terms_match_the_oracle(block):
    """The same sweep with the terms of round 5 in the draw: stable Neo-Hookean tets next to the reference's models, slide pins (a pinned face
    free to move in a random plane) next to fixed pins, and -- on the PCG path -- the end projection on a few soft modes, which must leave a
    converged trajectory where it is."""
    kinds = KINDS + [pkg.TET_STABLE_NH, pkg.TET_STABLE_NH]
    for seed in range(8 * block, 8 * block + 8):
        rng = np.random.default_rng(5000 + seed)
        n = int(rng.integers(2, 7))
        verts, tets = meshes.kuhn_cube(n, float(rng.uniform(0.3, 2.0)))
        h = verts[:, 0].max() / n
        interior = np.all((verts > 1e-9) & (verts < verts.max() - 1e-9), axis=1)
        verts = verts + interior[:, None] * rng.uniform(-0.15, 0.15, verts.shape) * h
        sc = scenes.Scene()
        sc.x = verts; sc.m = meshes.lumped_masses_tets(verts, tets, float(rng.uniform(500, 3000)))
        nk = int(rng.integers(1, 4))
        part = rng.integers(0, nk, len(tets))
        for k in range(nk):
            sel = tets[part == k]
            if len(sel):
                sc.tets.append((verts, sel, Lame(float(10 ** rng.uniform(4.5, 7.0)), float(rng.uniform(0.1, 0.45))), int(rng.choice(kinds)), 0))
        ls = int(rng.choice([0, 1, 2]))
        face = np.nonzero(verts[:, 0] < 1e-9)[0]
        slide = rng.random() < 0.6
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        for i in face:
            if slide and verts[i, 1] > 0.5 * verts[:, 1].max(): sc.slides[int(i)] = (verts[i].copy(), nrm.copy())
            else: sc.pins[int(i)] = verts[i].copy()
        sc.settings.update(admm_iters=int(rng.integers(2, 9)), linsolver=ls, gravity=float(rng.uniform(-15, 2)),
                           timestep_s=float(rng.choice([1 / 24, 1 / 60])))
        soft = int(rng.choice([0, 4])) if ls != 1 else 0
        s = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=3000, soft_modes=soft)
        o = sc.make_oracle(mode=1, gs_colors=s.gs_colors()[0] if ls == 1 else None)
        for _ in range(int(rng.integers(1, 4))):
            s.step(); o.step()
        err = scenes.rel_err(s.m_x, o.x)
        assert err < 1e-6, (seed, ls, soft, err)
        assert s.runtime_data().unconverged_solves == 0, seed
        s.close()
