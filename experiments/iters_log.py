import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, bench, scenes
w = bench.WORKLOADS["cube1m_mix"]
sc, nt, nv = bench.build_scene(w, int(sys.argv[1]) if len(sys.argv)>1 else 55)
for tol in (1e-8, 1e-10):
    s = sc.make_solver(pcg_tol=tol, pcg_max_iters=1500)
    for f in range(4):
        s.step()
        print("tol", tol, "frame", f, "iters/solve", s.runtime_data().pcg_iters_per_solve)
    s.close()
