import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, scenes
sc = scenes.blob_scene(20, admm_iters=10, linsolver=0)
def run(tol, soft, start, verify=False):
    os.environ["ADMM_HIP_DEFL_START"] = str(start)
    if verify: os.environ["ADMM_HIP_OC_VERIFY"] = "1"
    s = sc.make_solver(pcg_tol=tol, pcg_max_iters=3000, soft_modes=soft)
    os.environ.pop("ADMM_HIP_OC_VERIFY", None)
    for f in range(8): s.step()
    x = s.m_x.copy(); s.close(); return x
ref = run(1e-13, 0, 0, True)
for tol, soft, start, ver in ((1e-10, 8, 2, False), (1e-10, 8, 0, False), (1e-10, 0, 0, False), (1e-10, 8, 2, True), (1e-11, 8, 2, False), (1e-12, 8, 2, False), (1e-12, 0, 0, False)):
    print("tol %g soft %d start %d verify %d: rel_err vs 1e-13 %.2e" % (tol, soft, start, ver, scenes.rel_err(run(tol, soft, start, ver), ref)))
