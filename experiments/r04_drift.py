"""Round 4, review item 1(a): which pcg_tol keeps blob1m_mix within 1e-5 of the converged trajectory over the driver's 25 frames?
Reference = the same path at 1e-12 (+ 1e-13 as the reference's own noise floor); per tolerance: rel_err per frame, PCG iterations per
solve, ADMM it/s of frames 5..24 (stats frames).  ADMM_DRIFT_N=44 for a quick run.  Optional: ADMM_DRIFT_SCHEDULE=k,tol : the LAST k
solves of every frame at `tol` (ADMM_HIP_TOL_LAST / ADMM_HIP_TOL_LAST_N), the others at the listed tolerance."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import scenes

n = int(os.environ.get("ADMM_DRIFT_N", "118"))
frames = int(os.environ.get("ADMM_DRIFT_FRAMES", "25"))
wl = os.environ.get("ADMM_DRIFT_WORKLOAD", "blob1m_mix")
sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], n if wl.startswith("blob") else (n if n != 118 else None))
print("%s n=%d: %d tets %d verts" % (wl, n, nt, nv), flush=True)


def run(tol, mx, verify, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    os.environ["ADMM_HIP_OC_VERIFY"] = "1" if verify else "0"
    try:
        s = sc.make_solver(pcg_tol=tol, pcg_max_iters=mx)
    finally:
        os.environ.pop("ADMM_HIP_OC_VERIFY", None)
        for k in (env or {}):
            os.environ.pop(k, None)
    xs, its, unconv = [], 0, 0
    t_frames = 0.0
    s.upload()
    for f in range(frames):
        t0 = time.perf_counter()
        s.step_device(stats=True)
        dt = time.perf_counter() - t0
        rd = s.runtime_data()
        if f >= 5:
            t_frames += rd.step_ms
        s.download()
        xs.append(s.m_x.copy())
    tot = s.solve_totals()
    s.close()
    return xs, tot, 1e3 * sc.settings["admm_iters"] * (frames - 5) / t_frames


ref, tot, _ = run(1e-12, 1500, True)
if os.environ.get("ADMM_DRIFT_SKIP_FLOOR") != "1":
    ref2, _, _ = run(1e-13, 3000, True)
    e = [scenes.rel_err(a, b) for a, b in zip(ref2, ref)]
    print("reference noise floor (1e-13 vs 1e-12): max %.2e" % max(e), flush=True)
tols = [float(t) for t in os.environ.get("ADMM_DRIFT_TOLS", "1e-8,3e-9,1e-9,3e-10,1e-10,3e-11").split(",")]
for tol in tols:
    xs, tot, rate = run(tol, 800, False)
    e = [scenes.rel_err(a, b) for a, b in zip(xs, ref)]
    print("tol %.0e: max rel_err %.2e (frame %d)  iterations/solve %.2f  unconverged %d  ADMM it/s (stats frames) %.0f | %s" %
          (tol, max(e), int(np.argmax(e)), tot[2] / max(tot[0], 1), tot[0] - tot[1], rate, " ".join("%.1e" % v for v in e)), flush=True)
sched = os.environ.get("ADMM_DRIFT_SCHEDULE")
if sched:
    for item in sched.split(";"):
        base, k, last = item.split(",")
        xs, tot, rate = run(float(base), 800, False, env={"ADMM_HIP_TOL_LAST": last, "ADMM_HIP_TOL_LAST_N": k})
        e = [scenes.rel_err(a, b) for a, b in zip(xs, ref)]
        print("tol %s, last %s solves of a frame at %s: max rel_err %.2e (frame %d)  iterations/solve %.2f  ADMM it/s %.0f | %s" %
              (base, k, last, max(e), int(np.argmax(e)), tot[2] / max(tot[0], 1), rate, " ".join("%.1e" % v for v in e)), flush=True)
