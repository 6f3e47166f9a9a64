import os, sys
os.environ["ADMM_HIP_OC_DEBUG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import admm_elastic_amd as pkg
import scenes
sc = scenes.cube_scene(3, pkg.TET_NEOHOOKEAN, admm_iters=4, linsolver=0)
s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=400)
for f in range(2):
    s.step()
    print("frame", f, "nan", np.isnan(s.m_x).sum())
