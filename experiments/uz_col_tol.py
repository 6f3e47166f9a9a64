"""Tolerance of the cached K^-1 columns of UzawaCG (ADMM_HIP_UZ_COL_TOL = fraction of pcg_tol): whole-step error against the oracle with
the active set frozen (the full-size parity test's set-up, 3 frames) for each fraction.   python experiments/uz_col_tol.py 0.01 0.2 1"""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
if len(sys.argv) > 2:
    for f in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, f], env=dict(os.environ, ADMM_HIP_UZ_COL_TOL=f))
    raise SystemExit
import bench, scenes
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube100k_uzawa_floor"], None)
os.environ["ADMM_HIP_UZ_FREEZE"] = "1"
s = sc.make_solver(pcg_tol=bench.PCG_TOL, pcg_max_iters=600)
o = sc.make_oracle(mode=1, big=True); o.freeze_active = True
errs = []
for f in range(3):
    s.step(); o.step(); errs.append(scenes.rel_err(s.m_x, o.x))
print("ADMM_HIP_UZ_COL_TOL", os.environ.get("ADMM_HIP_UZ_COL_TOL"), "rel_err per frame", " ".join("%.2e" % e for e in errs), s.uzawa_cache_stats(), flush=True)
