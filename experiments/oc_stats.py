"""ADMM_HIP_OC_DEBUG=1: prints every solve's verdict (iterations, pipelined iterations, gamma)."""
import os, sys
os.environ["ADMM_HIP_OC_DEBUG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import bench
sc, nt, nv = bench.build_scene(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cube1m_mix"], None)
s = sc.make_solver(pcg_tol=float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8, pcg_max_iters=1500)
for _ in range(2):
    s.step()
