"""PCG iterations of every solve of a frame (bench workload, bench tolerance): python experiments/iters_log.py [workload] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "blob1m_mix"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc, nt, nv = bench.build_scene(dict(bench.WORKLOADS[wl], linsolver=0) if bench.WORKLOADS[wl]["linsolver"] == 2 else bench.WORKLOADS[wl])   # (per-solve counts are reported for linsolver 0)
s = sc.make_solver(pcg_tol=float(os.environ.get("TOL", bench.PCG_TOL)), pcg_max_iters=600, soft_modes=int(os.environ.get("SOFT", bench.SOFT_MODES)))
s.upload()
for f in range(frames):
    s.step_device(stats=True)
    rd = s.runtime_data()
    print("frame %2d: %s  sum %d  global %.3f ms" % (f, " ".join("%2d" % i for i in rd.pcg_iters_per_solve), sum(rd.pcg_iters_per_solve), rd.global_ms))
