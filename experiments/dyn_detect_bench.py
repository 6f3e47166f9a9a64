"""Self-collision detection at scale: two n-cell cubes (2 x 6 n^3 tets) with TetMeshCollision proxies, the upper one
pushed slightly into the lower one, UzawaCG.  Run under rocprofv3 --kernel-trace --stats to get the per-kernel times of
k_dyn_refit0 / k_dyn_refit_up / k_dyn_query (profiles/r01_m_*).  usage: dyn_detect_bench.py [n=44] [frames=2] [overlap]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
overlap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
sc = scenes.two_blocks_scene(n, overlap=overlap, floor=None, jitter=0.05, admm_iters=5)
nt = sum(len(t[1]) for t in sc.tets)
print("tets", nt, "verts", len(sc.x), "candidates", len(sc.surface_inds), flush=True)
t0 = time.time()
s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=600)
print("setup %.1f s" % (time.time() - t0), flush=True)
x = sc.x.ravel()
t0 = time.time(); h = s.detect_dynamic(x); t1 = time.time()
print("detect_dynamic (incl. H2D/D2H of the test hook): %d hits, %.1f ms" % (len(h), 1e3 * (t1 - t0)), flush=True)
s.upload()
for f in range(frames):
    t0 = time.time(); s.step_device(stats=True); rd = s.runtime_data()
    print("frame %d: %.1f ms, inner %d, local %.3f global %.3f collision %.3f ms" % (f, 1e3 * (time.time() - t0), rd.inner_iters, rd.local_ms, rd.global_ms, rd.collision_ms), flush=True)
s.download()
print("finite", np.isfinite(s.m_x).all())
