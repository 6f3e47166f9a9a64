import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import bench, scenes
w = bench.WORKLOADS["cube1m_mix"]
n = int(sys.argv[1]) if len(sys.argv)>1 else 55
sc, nt, nv = bench.build_scene(w, n)
def run(tol, mx, frames=2):
    s = sc.make_solver(pcg_tol=tol, pcg_max_iters=mx)
    its=[]
    for f in range(frames):
        s.step(); its.append(s.runtime_data().inner_iters/20.0)
    x = s.m_x.copy(); rd = s.runtime_data(); s.close()
    return x, its, rd
xt, it_t, rd = run(1e-12, 1500)
print("truth its/admm", it_t, "conv", rd.last_solve_converged, "global ms/it", rd.global_ms/20)
for tol in (1e-4, 1e-5, 1e-6, 1e-7, 1e-8):
    x, its, rd = run(tol, 600)
    print("tol %.0e  its/admm %s conv %d  rel_err vs truth %.3e   global ms/admm-it %.3f" % (tol, np.round(its,1), rd.last_solve_converged, scenes.rel_err(x, xt), rd.global_ms/20))
