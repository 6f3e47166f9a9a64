import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, scenes
import admm_elastic_amd as pkg
sc = scenes.cloth_scene(8, floor=0.46, admm_iters=8, linsolver=2)
s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=400)
o = sc.make_oracle(mode=1)
rng = np.random.default_rng(1)
x = sc.x.copy(); x[:,1] -= 0.02 + 0.03*rng.random(len(x)); x = x.ravel()
b = o.A @ (x + 0.001*rng.standard_normal(x.size))
hits = o.detect_passive(x)
print("hits", len(hits))
xo, ito = o.solve_uzawa(x, b, hits)
xg, itg = s.global_solve(b, x)
print("iters oracle", ito, "gpu", itg, "diff", np.abs(xg-xo).max())
yo = xo.reshape(-1,3)[:,1]; yg = xg.reshape(-1,3)[:,1]
hv = [h[0] for h in hits]
print("oracle y at hits min/max", yo[hv].min(), yo[hv].max(), " gpu", yg[hv].min(), yg[hv].max())
# second call: warm start y
xo2, ito2 = o.solve_uzawa(x, b, hits); xg2, itg2 = s.global_solve(b, x)
print("2nd: iters", ito2, itg2, "diff", np.abs(xg2-xo2).max())
