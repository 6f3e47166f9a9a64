"""Debug helper: the boxes sample in Python (GPU vs oracle), frame by frame."""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, scenes
import admm_elastic_amd as pkg
from admm_elastic_amd import meshes
from admm_elastic_amd.solver import Lame
cells, gap = 4, 1.3
sc = scenes.Scene()
for i in range(2):
    verts, tets = meshes.tet_blocks(cells, cells, cells)
    verts = verts / cells + np.array([-0.5 + 0.013 * i, -0.5 + i * gap, -0.5 + 0.007 * i])
    off = sc.add_tet_mesh(verts, tets, Lame.rubber(), pkg.TET_LINEAR)
    sc.add_self_collision(verts, tets, off)
sc.obstacles.append((0, [-1.0, 0, 0, 0]))
sc.settings.update(linsolver=2, admm_iters=10)
s = sc.make_solver(pcg_tol=1e-10, pcg_max_iters=500)
o = sc.make_oracle()
nv = len(sc.x) // 2
for f in range(0):
    s.step(); o.step()
    X = s.m_x.reshape(-1, 3); Y = o.x.reshape(-1, 3)
    print(f, 'gpu low[%.4f %.4f] up %.4f | orc low[%.4f %.4f] up %.4f | inner %d %d  hits p%d d%d unconv %d' % (
        X[:nv, 1].min(), X[:nv, 1].max(), X[nv:, 1].min(), Y[:nv, 1].min(), Y[:nv, 1].max(), Y[nv:, 1].min(),
        s.runtime_data().inner_iters, o.inner_iters, len(o._hits), len(o._dhits), s.runtime_data().unconverged_solves), flush=True)
    if not np.isfinite(X).all():
        break

# ---- second part: replay the oracle's frame with dynamic hits solve by solve on the GPU
print('replay')
sc2 = sc
s = sc2.make_solver(pcg_tol=1e-10, pcg_max_iters=500)
o = sc2.make_oracle()
for f in range(9):
    o.step()
xprev = o.x + o.dt * (o.v + np.tile([0, o.dt * o.gravity, 0], o.nv))   # x_bar of the next frame
tr = []
o.step(trace=tr)
import os
tol = float(os.environ.get('PCG_TOL', '1e-10'))
s = sc2.make_solver(pcg_tol=tol, pcg_max_iters=3000)
print('pcg_tol', tol)
xin = xprev
for k, (z, u, b, curr) in enumerate(tr):
    dh = o.detect_dynamic(xin); ph = o.detect_passive(xin)
    xg, it = s.global_solve(b, xin)
    o._dhits = dh
    Cm, c = o.make_matrix(ph, dh)
    o.y = np.zeros(0); o.uz_max_iters = 300
    xc, itc = o.solve_uzawa(xin, b, ph)
    o.uz_max_iters = 20
    print(k, 'rows p%d d%d' % (len(ph), len(dh)), 'gpu its', it, 'diff', np.abs(xg - curr).max(), 'max|xg| %.3g max|curr| %.3g' % (np.abs(xg).max(), np.abs(curr).max()),
          'res gpu %.3g orc %.3g' % (np.linalg.norm(Cm @ xg - c), np.linalg.norm(Cm @ curr - c)), 'converged: its', itc, 'max|x| %.3g' % np.abs(xc).max(),
          'dist gpu %.3g orc %.3g' % (np.abs(xg - xc).max(), np.abs(curr - xc).max()), flush=True)
    if not np.isfinite(xg).all():
        np.savez('gpurun_out/dbg_boxes_fail.npz', b=b, xin=xin, curr=curr)
        break
    xin = curr
