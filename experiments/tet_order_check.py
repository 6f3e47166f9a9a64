"""Does the ORDER of the tets matter for the local step?  cube1m_mix with the tets in the generator's cell-major order
vs Morton order of their centroids vs random order (the library keeps the caller's order inside a model group)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import bench

def morton(c, bits=10):
    lo, hi = c.min(axis=0), c.max(axis=0)
    q = np.minimum(((c - lo) / (hi - lo) * (2 ** bits - 1)).astype(np.uint64), 2 ** bits - 1)
    code = np.zeros(len(c), dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return code

for order in ("cell-major", "morton", "random"):
    sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube1m_mix"], None)
    new = []
    for verts, tets, lame, kind, off in sc.tets:
        if order == "morton":
            p = np.argsort(morton(verts[tets].mean(axis=1)), kind="stable")
        elif order == "random":
            p = np.random.default_rng(0).permutation(len(tets))
        else:
            p = np.arange(len(tets))
        new.append((verts, tets[p], lame, kind, off))
    sc.tets = new
    s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=600)
    s.upload()
    loc = rhs = 0.0
    for f in range(5):
        s.step_device(stats=True)
        if f >= 2:
            rd = s.runtime_data(); loc += rd.local_ms; rhs += rd.rhs_ms
    print("%-10s local %.1f us  rhs (gather) %.1f us per ADMM iteration" % (order, 1e3 * loc / 60, 1e3 * rhs / 60), flush=True)
    s.close()
