"""Round 5: drift of blob1m_mix against the converged trajectory (the same path at 1e-12, every pass verified) over MANY frames, for a
list of solver variants -- tolerance, tolerance schedule of the first solves (ADMM_HIP_TOL_SCHED), any other environment switch.
    ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="5e-10;5e-10:ADMM_HIP_TOL_SCHED=20,20;7e-10" python experiments/r05_drift.py
Per variant: rel_err per frame (max, where, every 10th), PCG iterations per solve, ADMM it/s of the statistics frames."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import scenes

n = int(os.environ.get("ADMM_DRIFT_N", "118"))
frames = int(os.environ.get("ADMM_DRIFT_FRAMES", "200"))
wl = os.environ.get("ADMM_DRIFT_WORKLOAD", "blob1m_mix")
sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], n if wl.startswith("blob") else (n if n != 118 else None))
print("%s n=%d: %d tets %d verts, %d frames" % (wl, n, nt, nv, frames), flush=True)


MODES = None


def run(tol, mx, verify, env=None, ref=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    os.environ["ADMM_HIP_OC_VERIFY"] = "1" if verify else "0"
    try:
        s = sc.make_solver(pcg_tol=tol, pcg_max_iters=mx, soft_modes=int((env or {}).get("SOFTSET", "0")))      # SOFTSET=k: Settings.soft_modes (the product path, environment switches in effect)
    finally:
        os.environ.pop("ADMM_HIP_OC_VERIFY", None)
        for k in (env or {}):
            os.environ.pop(k, None)
    xs, errs = [], []
    t_frames = 0.0
    k_soft = int((env or {}).get("SOFT", "0"))       # pseudo-switch of this script: end projection of every solve on the k lowest modes
    if k_soft:
        global MODES
        if MODES is None or MODES[1].shape[0] < k_soft:
            t0 = time.time(); MODES = s.soft_modes(max(k_soft, int(os.environ.get("ADMM_DRIFT_MODES_MAX", "64")))); print("soft modes: lowest eigenvalues %s ... %.3g (%.0f s)" % (" ".join("%.3g" % e for e in MODES[0][:6]), MODES[0][-1], time.time() - t0), flush=True)
        s.set_soft_modes(MODES[1][:k_soft])
    if "SOFTLIB" in (env or {}):      # the product path: the library computes the modes, the projection runs inside k_pcg2
        t0 = time.time(); s.compute_soft_modes(int(env["SOFTLIB"])); print("  (library modes: %.1f s)" % (time.time() - t0), flush=True)
    tot0 = s.solve_totals()
    s.upload()
    for f in range(frames):
        s.step_device(stats=True)
        rd = s.runtime_data()
        if f >= 5:
            t_frames += rd.step_ms
        s.download()
        if ref is None:
            xs.append(s.m_x.astype(np.float64).copy())
        else:
            errs.append(scenes.rel_err(s.m_x, ref[f]))
    tot = tuple(a - b for a, b in zip(s.solve_totals(), tot0))      # (without the solves that computed the modes)
    s.close()
    return xs, errs, tot, 1e3 * sc.settings["admm_iters"] * (frames - 5) / t_frames


t0 = time.time()
ref, _, tot, _ = run(1e-12, 1500, True)
print("reference (1e-12, verified): %.1f PCG iterations per solve, %.0f s" % (tot[2] / max(tot[0], 1), time.time() - t0), flush=True)
for item in os.environ.get("ADMM_DRIFT_VARIANTS", "5e-10").split(";"):
    parts = item.split(":")
    tol = float(parts[0])
    env = dict(kv.split("=", 1) for kv in parts[1:])
    _, e, tot, rate = run(tol, 800, False, env=env, ref=ref)
    print("tol %-7s %-40s max rel_err %.2e (frame %d)  its/solve %5.2f  unconverged %d  ADMM it/s %.0f | %s" %
          (parts[0], " ".join("%s=%s" % kv for kv in env.items()), max(e), int(np.argmax(e)), tot[2] / max(tot[0], 1), tot[0] - tot[1], rate,
           " ".join("%.1e" % v for v in e[9::10])), flush=True)
    if os.environ.get("ADMM_DRIFT_DUMP"):
        np.savetxt(os.environ["ADMM_DRIFT_DUMP"] + "_" + item.replace(":", "_").replace("=", "").replace(",", "-") + ".txt", np.array(e), fmt="%.3e")
