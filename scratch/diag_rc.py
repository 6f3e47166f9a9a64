import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, scenes
import admm_elastic_amd as pkg
def run(iters, frames=6):
    sc = scenes.cube_scene(3, pkg.TET_NEOHOOKEAN, admm_iters=iters, linsolver=0)
    s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=400)
    o = sc.make_oracle(mode=1)
    errs=[]
    for f in range(frames):
        s.step(); o.step(); errs.append(scenes.rel_err(s.m_x, o.x))
    return errs, s.runtime_data().pcg_iters_per_solve
for iters in (5, 10, 30):
    e, its = run(iters)
    print("recycle", "off" if os.environ.get("ADMM_HIP_NO_RECYCLE")=="1" else "on ", "admm_iters", iters, "err/frame", ["%.1e"%x for x in e], "its", its[:12])
