"""expand_check with Lame::rubber() (nu = 0.499), GPU next to the oracle (exact solves, tight prox)."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scenes
import admm_elastic_amd as pkg
from admm_elastic_amd import meshes
from admm_elastic_amd.solver import Lame
n = 4
sc = scenes.cube_scene(n, pkg.TET_NEOHOOKEAN, lame=Lame.rubber(), pin_face=False, admm_iters=20, gravity=0.0)
s = sc.make_solver(pcg_tol=1e-10, pcg_max_iters=3000)
o = sc.make_oracle(mode=1)
rng = np.random.default_rng(0)
X0 = sc.x.copy(); tets = sc.tets[0][1]
x0 = rng.uniform(-0.75, 0.75, X0.shape).ravel()
s.m_x = x0.copy(); o.x = x0.copy()
e = np.array([[a, b] for t in tets for a, b in ((t[0], t[1]), (t[0], t[2]), (t[0], t[3]), (t[1], t[2]), (t[1], t[3]), (t[2], t[3]))])
L0 = np.linalg.norm(X0[e[:, 0]] - X0[e[:, 1]], axis=1)
for f in range(100):
    s.step(); o.step()
    if f % 10 == 9 or f < 3:
        out = []
        for X in (s.m_x.reshape(-1, 3), o.x.reshape(-1, 3)):
            vol = meshes.tet_volumes(X, tets); L = np.linalg.norm(X[e[:, 0]] - X[e[:, 1]], axis=1)
            out.append('inv %3d err %.3g' % ((vol <= 0).sum(), np.abs(L / L0 - 1).max()))
        print(f, 'gpu', out[0], '| oracle', out[1], '| unconv', s.runtime_data().unconverged_solves, 'rel diff %.2g' % scenes.rel_err(s.m_x, o.x), flush=True)
