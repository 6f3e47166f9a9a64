"""Self-collision inside the multi-colour GS at scale: two n-cell cubes with TetMeshCollision proxies, -ls 1.
usage: dyn_gs_bench.py [n=44] [frames=3]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = scenes.two_blocks_scene(n, overlap=0.01, floor=-0.2, jitter=0.05, admm_iters=10, linsolver=1)
print("tets", sum(len(t[1]) for t in sc.tets), "verts", len(sc.x), "candidates", len(sc.surface_inds), flush=True)
s = sc.make_solver()
print("colours", s.gs_colors()[1], flush=True)
s.upload()
for f in range(frames):
    t0 = time.time(); s.step_device(stats=True); rd = s.runtime_data()
    print("frame %d: %.1f ms wall, sweeps %d, local %.3f global %.3f collision %.3f ms" % (f, 1e3 * (time.time() - t0), rd.inner_iters, rd.local_ms, rd.global_ms, rd.collision_ms), flush=True)
s.download()
X = s.m_x.reshape(-1, 3); nv = len(X) // 2
print("finite", np.isfinite(X).all(), "lower y [%.4f %.4f] upper min y %.4f" % (X[:nv, 1].min(), X[:nv, 1].max(), X[nv:, 1].min()))
