"""Soak run of the on-chip PCG: many frames, checks every frame for aborts / non-finite state / unconverged solves."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import bench, scenes
import admm_elastic_amd as pkg

def run(name, sc, frames, **kw):
    s = sc.make_solver(**kw)
    s.upload()
    t0 = time.time(); unconv = 0; inner = 0
    for f in range(frames):
        s.step_device(stats=True)
        rd = s.runtime_data()
        unconv += rd.unconverged_solves; inner += rd.inner_iters
    s.download()
    ok = np.isfinite(s.m_x).all()
    print("%s: %d frames in %.1f s, finite %s, unconverged solves %d, inner iterations %d, max |x| %.3f" % (name, frames, time.time() - t0, ok, unconv, inner, np.abs(s.m_x).max()))
    s.close()
    return ok and unconv == 0

good = True
sc, nt, nv = bench.build_scene(bench.WORKLOADS["blob1m_mix"], None)
good &= run("blob1m_mix pcg at the bench tolerance", sc, int(sys.argv[1]) if len(sys.argv) > 1 else 40, pcg_tol=bench.PCG_TOL, pcg_max_iters=2000)
sc, nt, nv = bench.build_scene(bench.WORKLOADS["blob1m_mix"], 50)
good &= run("blob77k pcg 1e-12, asynchronous steps", sc, 150, pcg_tol=1e-12, pcg_max_iters=3000)
for n in (7, 13, 31, 47):     # block counts / waves per block around the plan's break points
    sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube1m_mix"], n)
    good &= run("cube n=%d pcg 1e-10" % n, sc, 30, pcg_tol=1e-10, pcg_max_iters=3000)
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube1m_mix"], None)
good &= run("cube1m_mix pcg 1e-8", sc, int(sys.argv[1]) if len(sys.argv) > 1 else 40, pcg_tol=1e-8, pcg_max_iters=2000)
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube1m_mix"], 20)
good &= run("cube48k pcg 1e-12", sc, 200, pcg_tol=1e-12, pcg_max_iters=3000)
sc = scenes.cloth_scene(60, floor=0.3, admm_iters=10, linsolver=2)
good &= run("cloth60 uzawa floor", sc, 60, pcg_tol=1e-10, pcg_max_iters=2000)
for wl, frames in (("cube100k_gs", 80), ("cloth200k_gs_floor", 120)):     # the persistent multi-colour GS kernel (one launch per solve)
    sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], None)
    good &= run(wl + " (k_gs_persist)", sc, frames, pcg_tol=bench.PCG_TOL, pcg_max_iters=600)
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube100k_uzawa_floor"], None)      # UzawaCG with the persistent Schur kernel at the bench size
good &= run("cube100k_uzawa_floor (k_uz_persist)", sc, 60, pcg_tol=bench.PCG_TOL, pcg_max_iters=600)
print("SOAK", "OK" if good else "FAILED")
