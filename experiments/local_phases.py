"""Where the waves of the local step spend their time (library built with -DADMM_LOCAL_PHASES): mean microseconds per wave
between the marks of tet_compute_store.  python experiments/local_phases.py [workload]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["ADMM_HIP_EXTRA_FLAGS"] = "-DADMM_LOCAL_PHASES"
from admm_elastic_amd import build
build.build_library(force=True)
import bench
from admm_elastic_amd import capi
wl = sys.argv[1] if len(sys.argv) > 1 else "blob1m_mix"
sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], None)
s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=2000)
s.upload()
L = capi.lib()
out = (C.c_ulonglong * 8)()
for f in range(3): s.step_device(stats=False)
s.download()
L.admm_debug_local_phases(out)
for f in range(3): s.step_device(stats=False)
s.download()
L.admm_debug_local_phases(out)
w = out[7]
names = ["issue loads + wait for them", "F, SVD", "prox (Newton)", "u+, forces, park", "wait at the block barrier", "reduce + store records", "drain stores"]
tot = 0
for k in range(7):
    us = out[k] / w * 0.01
    tot += us
    print("%-30s %7.2f us per wave" % (names[k], us))
print("wave lifetime %.2f us; %d wave-launches" % (tot, w))
s.close()
