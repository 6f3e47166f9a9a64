import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, scenes
sc = scenes.cloth_scene(40)
o = sc.make_oracle()
b = o.A @ np.random.default_rng(3).standard_normal(o.dof)
xo = o.solve_ldlt(b)
for aff in ("1", "0"):
    for cheb in ("1", "0"):
        for tol in (1e-8, 1e-11):
            os.environ["ADMM_HIP_OC_AFFINE"] = aff; os.environ["ADMM_HIP_OC_CHEB"] = cheb
            s = sc.make_solver(pcg_tol=tol, pcg_max_iters=5000)
            x, it = s.global_solve(b, np.zeros(o.dof))
            print("affine", aff, "cheb", cheb, "tol", tol, "iters", it, "err", np.linalg.norm(x - xo) / np.linalg.norm(xo), flush=True)
            s.close()
