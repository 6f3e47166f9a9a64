import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, bench, scenes
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube100k_uzawa_floor"], None)
s = sc.make_solver(pcg_tol=bench.PCG_TOL, pcg_max_iters=600)
for f in range(26):
    s.step()
    x = s.m_x.reshape(-1, 3)
    st = s.uzawa_cache_stats()
    print(f, "ymin %.5f" % x[:, 1].min(), "n below floor+1e-9:", int((x[:, 1] < -0.02 + 1e-9).sum()) if True else 0, st, "inner", s.runtime_data().inner_iters, flush=True)
