"""UzawaCG touchdown cost: the K^-1 columns of the bottom layer of the cube100k_uzawa_floor scene (729 vertices = 243 launches of
k_pcg2) on the main stream alone (ADMM_HIP_UZ_LANES=1) and side by side on the lanes the chip holds.  Prints wall-clock per batch and
the largest difference between the two Schur solves.  Usage (GPU box): python experiments/uz_lanes_ab.py [lanes ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench

w = bench.WORKLOADS["cube100k_uzawa_floor"]
sc, ne, nv = bench.build_scene(w)
x0 = sc.x.copy()
floor_h = sc.obstacles[0][1][0]
x = x0.copy()
low = x[:, 1] < x[:, 1].min() + 1e-9
x[low, 1] = floor_h - 0.003          # the bottom layer 3 mm under the floor
x = x.ravel()
rng = np.random.default_rng(0)
b = None
out = {}
for lanes in (sys.argv[1:] or ["1", "0", "2", "4"]):
    if lanes != "0": os.environ["ADMM_HIP_UZ_LANES"] = lanes
    else: os.environ.pop("ADMM_HIP_UZ_LANES", None)
    s = sc.make_solver(pcg_tol=bench.PCG_TOL, pcg_max_iters=600)
    if b is None: b = np.repeat(sc.m, 3) * x      # (any right-hand side: the columns do not depend on it)
    t0 = time.perf_counter()
    xg, it = s.global_solve(b, x)
    t1 = time.perf_counter()
    xg2, it2 = s.global_solve(b, x)
    t2 = time.perf_counter()
    st = s.uzawa_cache_stats()
    out[lanes] = xg
    print("lanes=%s (in use %d): first solve with %d column launches %.1f ms, same solve again %.2f ms; Schur its %d; unconverged %d" %
          (lanes, st["lanes"], st["column_solves"], 1e3 * (t1 - t0), 1e3 * (t2 - t1), it, st["unconverged_columns"]), flush=True)
    del s
k0 = list(out)[0]
for k in out:
    print("max |x(lanes=%s) - x(lanes=%s)| = %.3e" % (k, k0, np.abs(out[k] - out[k0]).max()))
