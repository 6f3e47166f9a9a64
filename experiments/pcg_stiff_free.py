"""PCG on a FREE (no pins) nearly incompressible body: cond(D^-1 A) ~ 1e7.  Solve-level check against the exact solve,
for dense and sparse right-hand sides, on-chip and launch path, several tolerances."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scenes
import admm_elastic_amd as pkg
from admm_elastic_amd import meshes
from admm_elastic_amd.solver import Lame
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = scenes.Scene()
for i in range(2):
    verts, tets = meshes.tet_blocks(cells, cells, cells)
    verts = verts / cells + np.array([-0.5 + 0.013 * i, -0.5 + i * 1.3, -0.5 + 0.007 * i])
    sc.add_tet_mesh(verts, tets, Lame.rubber(), pkg.TET_LINEAR)
sc.settings.update(linsolver=0, admm_iters=10)
o = sc.make_oracle()
rng = np.random.default_rng(0)
x0 = sc.x.ravel().copy()
rhs = {'dense A(x+noise)': o.A @ (x0 + 1e-3 * rng.standard_normal(x0.size)), 'sparse (3 entries)': np.zeros(x0.size)}
rhs['sparse (3 entries)'][[5, 100, 301]] = [1.0, -2.0, 0.5]
for launches in ('0', '1'):
    os.environ['ADMM_HIP_PCG_LAUNCHES'] = launches
    for tol in (1e-8, 1e-10, 1e-12, 1e-14):
        s = sc.make_solver(pcg_tol=tol, pcg_max_iters=3000)
        for name, b in rhs.items():
            xe = o._lu.solve(b)
            for start, xs in (('x0=0', np.zeros_like(b)), ('x0=x', x0)):
                xg, it = s.global_solve(b, xs)
                print('launch' if launches == '1' else 'onchip', 'tol %.0e' % tol, '%-18s' % name, start, 'its', it,
                      'rel err %.3g' % (np.abs(xg - xe).max() / max(np.abs(xe).max(), 1e-300)),
                      'rel res %.3g' % (np.linalg.norm(o.A @ xg - b) / np.linalg.norm(b)), flush=True)
        s.close()
