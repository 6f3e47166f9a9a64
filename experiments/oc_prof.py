"""Per-phase timing of the on-chip PCG (block 0's view): ADMM_HIP_OC_PROF=1 python experiments/oc_prof.py"""
import os, sys
os.environ["ADMM_HIP_OC_PROF"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import bench
sc, nt, nv = bench.build_scene(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cube1m_mix"], None)
s = sc.make_solver(pcg_tol=float(os.environ.get("ADMM_PROF_TOL", bench.PCG_TOL)), pcg_max_iters=600, soft_modes=int(os.environ.get("ADMM_PROF_SOFT", bench.SOFT_MODES)))
for _ in range(6):
    s.step()
