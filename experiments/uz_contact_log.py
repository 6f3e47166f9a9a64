"""The UzawaCG bench workload frame by frame: lowest point, vertices on the floor, cached K^-1 columns / column solves so far, Schur iterations,
wall time of the frame (a frame in which new vertices touch pays their column solves).   python experiments/uz_contact_log.py [frames=26]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, bench, scenes
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube100k_uzawa_floor"], None)
s = sc.make_solver(pcg_tol=bench.PCG_TOL, pcg_max_iters=600)
last = (0, 0, 0)
for f in range(int(sys.argv[1]) if len(sys.argv) > 1 else 26):
    t = time.perf_counter(); s.step(); dt = time.perf_counter() - t
    x = s.m_x.reshape(-1, 3)
    st = s.uzawa_cache_stats()
    tot = s.solve_totals(); pcg = (tot[0] - (last[0] if f else 0), tot[2] - (last[2] if f else 0)); last = tot
    print(f, "%.1f ms" % (1e3 * dt), "ymin %.5f" % x[:, 1].min(), "on the floor:", int((x[:, 1] < -0.02 + 1e-9).sum()), st, "inner", s.runtime_data().inner_iters, "| PCG solves %d, iterations %d (%.1f per solve)" % (pcg[0], pcg[1], pcg[1] / max(1, pcg[0])), flush=True)
